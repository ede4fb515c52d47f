#!/usr/bin/env python
"""HBM traffic of the headline kernel (K3 = bottleneck conv2 forward, role-1 instantiations) from
two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py, per K3 launch.

    python tools/k3_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> out.json

Corrections per MI355X_MICROARCH.md (HBM section): FETCH_SIZE x2 on gfx950 for 16-B/lane reads;
WRITE_SIZE exact.  Both counters are reported in KiB."""
import csv
import json
import sys


def collect(path, counter):
    conv = red = 0.0
    n_conv = n_red = 0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        v = float(r["Counter_Value"])
        if "igemm_rows_fast_kernel" in name and ", 0, 1, true, " in name:
            conv += v
            n_conv += 1
        elif "splitk_reduce_kernel<false, 1>" in name:
            red += v
            n_red += 1
    return conv, n_conv, red, n_red


def main():
    fetch, write, out = sys.argv[1:4]
    fc, n1, fr, nr1 = collect(fetch, "FETCH_SIZE")
    wc, n2, wr, nr2 = collect(write, "WRITE_SIZE")
    assert n1 == n2 and n1 > 0, (n1, n2)
    per = lambda v: v * 1024.0 / n1
    res = {
        "kernel": "igemm_rows_fast_kernel<64,BN,false,3,0,1,true,PAIR> + splitk_reduce_kernel<false,1> "
                  "(bottleneck conv2 forward of the R50 anchor, bs 2, 512x1024)",
        "k3_launches": n1, "reduce_launches": nr1,
        "conv_fetch_bytes_per_launch": round(2 * per(fc)), "conv_write_bytes_per_launch": round(per(wc)),
        "reduce_fetch_bytes_per_launch": round(2 * per(fr)), "reduce_write_bytes_per_launch": round(per(wr)),
        "hbm_bytes_per_launch": round(2 * per(fc) + per(wc) + 2 * per(fr) + per(wr)),
        "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B requests as 64 B, MI355X_MICROARCH.md HBM "
                      "section); WRITE_SIZE exact; both in KiB",
        "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 "
                   "bench.py --steps 4 --warmup 2 --arch R50 --no-cpu-baseline",
    }
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
