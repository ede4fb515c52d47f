#!/usr/bin/env python
"""Kernel-by-kernel listing of ONE training step from a `rocprofv3 --kernel-trace` CSV:

    python tools/step_sequence.py <N_kernel_trace.csv[.gz]> out.txt [step_from_end]

One line per kernel in start order: start offset (us) within the step, duration (us), the gap to the
previous kernel END on the same stream, stream, workgroups, short kernel name.  Steps are delimited by
the optimizer (`sgd_kernel` launches), like tools/timeline.py."""
import csv
import gzip
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_rocprof import short  # noqa: E402


def main():
    src, dst = sys.argv[1], sys.argv[2]
    back = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    ev = []
    op = gzip.open(src, "rt") if src.endswith(".gz") else open(src)
    for r in csv.DictReader(op):
        wg = int(r.get("Workgroup_Size", 0) or 0) or 1
        grid = int(r.get("Grid_Size", 0) or 0)
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                   r.get("Stream_Id", r.get("Queue_Id")), r["Kernel_Name"], grid // wg))
    ev.sort()
    sgd = [i for i, e in enumerate(ev) if "sgd_kernel" in e[3]]
    steps = []
    for i in sgd:
        if steps and i - steps[-1][-1] < 50:
            steps[-1].append(i)
        else:
            steps.append([i])
    bounds = [ev[s[-1]][1] for s in steps]
    a, b = bounds[-back - 1], bounds[-back]
    es = [e for e in ev if a < e[0] <= b]
    last_end = {}
    with open(dst, "w") as f:
        f.write("# step wall %.1f us, %d kernels\n# start_us  dur_us  gap_us  stream  wgs  kernel\n"
                % ((b - a) / 1e3, len(es)))
        for s, e, st, name, wgs in es:
            gap = (s - last_end[st]) / 1e3 if st in last_end else 0.0
            last_end[st] = e
            f.write("%9.1f %7.1f %7.1f  %s %6d  %s\n" % ((s - a) / 1e3, (e - s) / 1e3, gap, st, wgs,
                                                       short(name)[:110]))


if __name__ == "__main__":
    main()
