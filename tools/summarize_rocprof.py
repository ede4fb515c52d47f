#!/usr/bin/env python
"""Turn a `rocprofv3 --kernel-trace --stats --output-format csv` kernel_stats CSV into the compact
per-kernel table committed under profiles/ (raw CSV next to it).

    python tools/summarize_rocprof.py gpurun_out/prof/x/N_kernel_stats.csv profiles/r01_bench [steps]
"""
import csv
import re
import shutil
import sys


K3 = "K3: bottleneck conv2 3x3 forward (role 1)"
K3_RED = "K3: its split-K reduce (role 1)"


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("gs::", "")
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("at::native::", "")
    return name[:96]


def group(name):
    if "igemm_rows_fast_kernel" in name:     # <BM, BN, BTRANS, KS, ABL, ROLE, PIPE, PAIR>
        t = name.split("<")[1].split(">")[0].split(", ")
        if len(t) > 5 and t[5] == "1":
            return K3
        kind = "conv dgrad" if t[2] == "true" else "conv forward"
        return "%s %s" % (kind, {"1": "1x1", "3": "3x3"}.get(t[3], "other"))
    if "stem7x7_fwd_kernel" in name:
        return "conv forward stem/other"
    if "stem7x7_wgrad_kernel" in name:
        return "conv wgrad stem/other"
    if "conv1x1_stream_kernel" in name:      # <BNW, BTRANS, AFF>
        t = name.split("<")[1].split(">")[0].split(", ")
        return "%s 1x1" % ("conv dgrad" if t[1] == "true" else "conv forward")
    if "igemm_rows_kernel" in name:          # <BM, BN, BTRANS, DIVS, SCALAR, KS>
        t = name.split("<")[1].split(">")[0].split(", ")
        kind = "conv dgrad" if t[2] == "true" else "conv forward"
        return "%s %s" % (kind, {"1": "1x1", "3": "3x3"}.get(t[5], "stem/other"))
    if "igemm_wgrad_fast" in name:           # <BM, BN, KS, PAIR>
        ks = name.split("<")[1].split(">")[0].split(", ")[2]
        return "conv wgrad %s" % {"1": "1x1", "3": "3x3"}.get(ks, "stem/other")
    if "igemm_wgrad" in name:                # general kernel <BM, BN, SCALAR, KS>
        ks = name.split("<")[1].split(">")[0].split(", ")[-1]
        return "conv wgrad %s" % {"1": "1x1", "3": "3x3"}.get(ks, "stem/other")
    if "splitk_reduce_kernel<false, 1>" in name or "splitk_reduce_stats_kernel<1>" in name:
        return K3_RED      # (the fused reduce + BN statistics pass when the K3 timer is off)
    if "splitk_reduce" in name:
        return "split-K reduce"
    if re.search(r"bn_|sum_partials|colsum", name):
        return "batchnorm / reductions"
    if re.search(r"ce_|resize_argmax", name):
        return "fused resize + cross entropy"
    if re.search(r"slide_fuse|seg_augment", name):
        return "inference epilogue / input pipeline"
    if re.search(r"maxpool|avgpool|bilinear|copy2d|scale_nc", name):
        return "pool / resize / copy"
    if "sgd_kernel" in name:
        return "fused SGD"
    return "torch / runtime (fill, copy, rng)"


def main():
    src, dst = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else None
    rows = list(csv.DictReader(open(src)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    shutil.copyfile(src, dst + "_kernel_stats.csv")
    groups = {}
    for r in rows:
        g = groups.setdefault(group(r["Name"]), [0.0, 0])
        g[0] += float(r["TotalDurationNs"])
        g[1] += int(r["Calls"])
    with open(dst + "_summary.md", "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary\n\nsource: `%s`\n\n" % src)
        f.write("total kernel time %.3f ms" % (total / 1e6))
        if steps:
            f.write(" over %d steps = %.3f ms/step" % (steps, total / 1e6 / steps))
        if K3 in groups:
            t, c = groups[K3]
            tr, cr = groups.get(K3_RED, (0.0, 0))
            f.write("\n\n## headline kernel (bench.py `roofline`)\n\nK3 launches %d, conv kernel avg %.2f us, "
                    "split-K reduce avg %.2f us over %d reduces; **conv + reduce per K3 launch = %.2f us**"
                    % (c, t / c / 1e3, (tr / cr / 1e3) if cr else 0.0, cr, (t + tr) / c / 1e3))
        f.write("\n\n## by operator group\n\n| group | % | total ms | calls |\n|---|---|---|---|\n")
        for k, (t, c) in sorted(groups.items(), key=lambda kv: -kv[1][0]):
            f.write("| %s | %.2f | %.3f | %d |\n" % (k, 100 * t / total, t / 1e6, c))
        f.write("\n## by kernel (top 40)\n\n| kernel | calls | avg us | total ms | % |\n|---|---|---|---|---|\n")
        for r in rows[:40]:
            f.write("| `%s` | %s | %.1f | %.3f | %.2f |\n" % (
                short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
    print("wrote", dst + "_summary.md")


if __name__ == "__main__":
    main()
