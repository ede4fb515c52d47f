#!/usr/bin/env python
"""Which stream bounds a training step, and what runs on it — from a
`rocprofv3 --kernel-trace --output-format csv` trace of bench.py.

    python tools/timeline.py <N_kernel_trace.csv[.gz]> out.md [steps_to_average]

Steps are delimited by the optimizer (`sgd_kernel` launches); per step and stream: kernels, busy
time, idle gaps; per stream and kernel group: launches and microseconds per step; which kernels the
main stream's idle gaps follow."""
import csv
import gzip
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_rocprof import group, short  # noqa: E402


def _open(path):
    return gzip.open(path, "rt") if path.endswith(".gz") else open(path)


def main():
    src, dst = sys.argv[1], sys.argv[2]
    nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    ev = []
    for r in csv.DictReader(_open(src)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", r.get("Queue_Id")),
                   r["Kernel_Name"]))
    ev.sort()
    sgd = [i for i, e in enumerate(ev) if "sgd_kernel" in e[3]]
    steps = []
    for i in sgd:                      # an optimizer step is a few sgd launches close together
        if steps and i - steps[-1][-1] < 50:
            steps[-1].append(i)
        else:
            steps.append([i])
    bounds = [ev[s[-1]][1] for s in steps]
    nsteps = min(nsteps, len(bounds) - 1)
    a, b = bounds[-nsteps - 1], bounds[-1]
    es = [e for e in ev if a < e[0] <= b]
    streams = sorted({e[2] for e in es}, key=lambda s: -sum(1 for e in es if e[2] == s))
    out = ["# Stream timeline of the training step (rocprofv3 --kernel-trace)", "",
           "source: `%s`, last %d optimizer steps; wall %.2f ms per step" % (
               os.path.basename(src), nsteps, (b - a) / 1e6 / nsteps), ""]
    out += ["| stream | kernels / step | busy ms / step | idle gaps ms / step (count, median us) |",
            "|---|---|---|---|"]
    for st in streams:
        q = [e for e in es if e[2] == st]
        busy = sum(e[1] - e[0] for e in q) / 1e6 / nsteps
        gaps = [(y[0] - x[1]) / 1e3 for x, y in zip(q, q[1:]) if y[0] > x[1]]
        med = sorted(gaps)[len(gaps) // 2] if gaps else 0.0
        out.append("| %s | %.0f | %.2f | %.2f (%.0f, %.1f) |" % (
            st, len(q) / nsteps, busy, sum(gaps) / 1e3 / nsteps, len(gaps) / nsteps, med))
    cur, union = None, 0
    for s, e, *_ in es:
        if cur is None or s > cur[1]:
            if cur:
                union += cur[1] - cur[0]
            cur = [s, e]
        else:
            cur[1] = max(cur[1], e)
    union += cur[1] - cur[0]
    out += ["", "GPU busy on any stream: %.2f ms per step." % (union / 1e6 / nsteps), "",
            "## kernel groups per stream", "", "| stream | group | launches / step | us / step | % of kernel time |",
            "|---|---|---|---|---|"]
    g = defaultdict(lambda: [0, 0.0])
    for s, e, st, name in es:
        g[(st, group(name))][0] += 1
        g[(st, group(name))][1] += (e - s) / 1e3
    tot = sum(v[1] for v in g.values())
    for key, v in sorted(g.items(), key=lambda kv: -kv[1][1]):
        out.append("| %s | %s | %.1f | %.0f | %.1f |" % (key[0], key[1], v[0] / nsteps, v[1] / nsteps,
                                                       100 * v[1] / tot))
    main_q = [e for e in es if e[2] == streams[0]]
    before = defaultdict(lambda: [0, 0.0])
    for x, y in zip(main_q, main_q[1:]):
        gap = (y[0] - x[1]) / 1e3
        if gap > 2:
            before[short(x[3])[:70]][0] += 1
            before[short(x[3])[:70]][1] += gap
    out += ["", "## idle gaps of the main stream by preceding kernel (> 2 us)", "",
            "| kernel | gaps / step | idle us / step |", "|---|---|---|"]
    for k, v in sorted(before.items(), key=lambda kv: -kv[1][1])[:10]:
        out.append("| `%s` | %.1f | %.0f |" % (k, v[0] / nsteps, v[1] / nsteps))
    open(dst, "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
