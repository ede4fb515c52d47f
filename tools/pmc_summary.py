#!/usr/bin/env python
"""Per-kernel-group counter summary from `rocprofv3 --pmc ...` counter_collection CSVs.

    python tools/pmc_summary.py mfma  <csv with SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES> out.md
    python tools/pmc_summary.py hbm   <csv with FETCH_SIZE> <csv with WRITE_SIZE> out.json

mfma: busy fraction of the MFMA pipes per kernel group.  SQ_VALU_MFMA_BUSY_CYCLES is summed over the
      1024 SIMDs (32 cycles per v_mfma_f32_16x16x4_f32: checked, busy / 32 x 2048 FLOP = the padded FLOPs
      of the launch); SQ_BUSY_CYCLES is summed over the 32 shader engines (checked: = 32 x duration x
      clock).  Utilisation = MFMA_BUSY / (1024 x kernel cycles) = (MFMA_BUSY / SQ_BUSY_CYCLES) / 32.
hbm : bytes per launch of the headline kernel (role-1 bottleneck conv2 forward + its split-K reduce);
      FETCH_SIZE x2 on gfx950 for 16-B/lane reads, WRITE_SIZE exact, both in KiB (HBM section)."""
import csv
import gzip
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_rocprof import group  # noqa: E402


def _open(path):
    return gzip.open(path, "rt") if path.endswith(".gz") else open(path)


def read(path):
    per = defaultdict(dict)   # dispatch -> {counter: value, name}
    for r in csv.DictReader(_open(path)):
        d = per[r["Dispatch_Id"]]
        d["name"] = r["Kernel_Name"]
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return per


def mfma(path, out):
    per = read(path)
    g = defaultdict(lambda: [0, 0.0, 0.0])
    for d in per.values():
        if "SQ_BUSY_CYCLES" not in d:
            continue
        k = group(d["name"])
        g[k][0] += 1
        g[k][1] += d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        g[k][2] += d["SQ_BUSY_CYCLES"]
    lines = ["# MFMA pipe busy per kernel group (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES)", "",
             "source: `%s`" % path, "",
             "MFMA busy = (sum SQ_VALU_MFMA_BUSY_CYCLES / sum SQ_BUSY_CYCLES) / 32 over the group's "
             "launches: MFMA_BUSY is summed over the 1024 SIMDs (32 cycles per fp32 16x16x4 MFMA), "
             "SQ_BUSY_CYCLES over the 32 shader engines, so the quotient / 32 is the fraction of the "
             "kernel's cycles in which a SIMD's MFMA pipe was executing, averaged over all SIMDs.  "
             "Counter collection serialises the kernels (no stream overlap).  It includes MFMA work on "
             "tile padding; `useful GF / launch` = MFMA_BUSY / 32 x 2048 FLOP.", "",
             "| group | launches | MFMA busy | MFMA GFLOP issued / launch |", "|---|---|---|---|"]
    for k, (n, m, b) in sorted(g.items(), key=lambda kv: -kv[1][1]):
        if m > 0:
            lines.append("| %s | %d | %.3f | %.2f |" % (k, n, m / b / 32.0 if b else 0.0,
                                                       m / n / 32.0 * 2048 / 1e9))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


def hbm(fetch, write, out):
    pf, pw = read(fetch), read(write)

    def collect(per, counter):
        conv = red = 0.0
        n_conv = n_red = 0
        for d in per.values():
            name = d["name"]
            if counter not in d:
                continue
            if "igemm_rows_fast_kernel" in name and ", 0, 1, true, " in name:
                conv += d[counter]
                n_conv += 1
            elif "splitk_reduce_kernel<false, 1>" in name or ("splitk_reduce_stats_kernel<1>" in name):
                red += d[counter]
                n_red += 1
        return conv, n_conv, red, n_red
    fc, n1, fr, nr1 = collect(pf, "FETCH_SIZE")
    wc, n2, wr, nr2 = collect(pw, "WRITE_SIZE")
    assert n1 == n2 and n1 > 0, (n1, n2)
    per_launch = lambda v: v * 1024.0 / n1
    res = {
        "kernel": "igemm_rows_fast_kernel<64,BN,false,3,0,1,...> (split-K slabs combined inside the launch; with "
                  "GS_SPLITK_INKERNEL=0 + its reduce launch splitk_reduce_kernel<false,1> / splitk_reduce_stats_kernel<1>): "
                  "bottleneck conv2 forward, sampled subnet mix of bench.py, bs 2, 512x1024",
        "k3_launches": n1, "reduce_launches": nr1,
        "conv_fetch_bytes_per_launch": round(2 * per_launch(fc)),
        "conv_write_bytes_per_launch": round(per_launch(wc)),
        "reduce_fetch_bytes_per_launch": round(2 * per_launch(fr)),
        "reduce_write_bytes_per_launch": round(per_launch(wr)),
        "hbm_bytes_per_launch": round(2 * per_launch(fc) + per_launch(wc) + 2 * per_launch(fr) + per_launch(wr)),
        "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B requests as 64 B, MI355X_MICROARCH.md HBM "
                      "section); WRITE_SIZE exact; both in KiB; separate --pmc passes",
        "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 "
                   + (os.environ.get("GS_PMC_COMMAND") or
                      "bench.py --steps 20 --warmup 0 --repeats 1 --no-cpu-baseline --no-k3-timer --no-check"),
    }
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "mfma":
        mfma(sys.argv[2], sys.argv[3])
    else:
        hbm(sys.argv[2], sys.argv[3], sys.argv[4])
