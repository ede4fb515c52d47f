#!/usr/bin/env python
"""Per-queue busy time and idle gaps from a `rocprofv3 --kernel-trace --output-format csv` trace:
which stream bounds a training step, and how much of it is launch gaps rather than kernels.

    python tools/timeline.py gpurun_out/prof/x/N_kernel_trace.csv [skip_fraction]
"""
import csv
import sys
from collections import defaultdict


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
    ev = []
    for r in rows:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"),
                   r.get("Stream_Id", r.get("Queue_Id", "0")), r["Kernel_Name"]))
    ev.sort()
    t0, t1 = ev[0][0], ev[-1][1]
    cut = t0 + skip * (t1 - t0)           # drop warm-up
    ev = [e for e in ev if e[0] >= cut]
    span = (ev[-1][1] - ev[0][0]) / 1e6
    print("window %.1f ms, %d dispatches" % (span, len(ev)))
    byq = defaultdict(list)
    for e in ev:
        byq[(e[2], e[3])].append(e)
    for q, es in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        busy = sum(e[1] - e[0] for e in es) / 1e6
        gaps = [(b[0] - a[1]) / 1e3 for a, b in zip(es, es[1:]) if b[0] > a[1]]
        small = sum(g for g in gaps if g < 20)
        print("queue %s stream %s: %5d kernels, busy %.1f ms (%.0f%% of window), gaps<20us total %.1f ms "
              "(median %.1f us), gaps>=20us total %.1f ms" % (
                  q[0], q[1], len(es), busy, 100 * busy / span, small / 1e3,
                  sorted(gaps)[len(gaps) // 2] if gaps else 0.0, sum(g for g in gaps if g >= 20) / 1e3))
    # union busy (any queue)
    cur_s, cur_e, union = None, None, 0
    for s, e, *_ in ev:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    print("GPU busy (any queue) %.1f ms = %.0f%% of the window" % (union / 1e6, 100 * union / 1e6 / span))


if __name__ == "__main__":
    main()
