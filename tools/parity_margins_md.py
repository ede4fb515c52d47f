#!/usr/bin/env python
"""profiles/rNN_parity_margins.md from the JSON lines tests/parity.py appends under
GS_PARITY_MARGINS=<file> (one record per compared HIP-vs-oracle training step):

    GS_PARITY_MARGINS=gpurun_out/margins.jsonl python -m pytest tests -m gpu -q
    python tools/parity_margins_md.py gpurun_out/margins.jsonl profiles/r03_parity_margins.md
"""
import json
import sys


def main(src, dst):
    recs = [json.loads(l) for l in open(src) if l.strip()]
    lines = ["# Parity margins of the model-level tests (HIP train step vs the fp64 oracle on the HIP "
             "path's branch pattern)", "",
             "Every parameter gradient is compared in the max norm at 1e-3 of that parameter's largest "
             "oracle gradient.  Where it misses, the fp32 oracle (PyTorch-CPU, same branches) is run too and "
             "the parameter is accepted only if `hip_err <= 3 x fp32_err` (tests/parity.py); per step the "
             "MEDIAN ratio over the conditioned parameters must stay <= 1.5.  Columns: parameters with a "
             "gradient / conditioned / ratio median, p90, max / largest HIP error among the conditioned / "
             "largest error among the unconditioned (all < 1e-3) / loss error.", "",
             "| test | params | conditioned | ratio median | p90 | max | worst conditioned hip err | worst "
             "unconditioned | loss rel err |", "|---|---|---|---|---|---|---|---|---|"]
    for r in recs:
        lines.append("| %s | %d | %d | %.2f | %.2f | %.2f | %.2e | %.2e | %.1e |" % (
            r["test"].split("::")[-1], r["parameters"], r["conditioned"], r["ratio_median"], r["ratio_p90"],
            r["ratio_max"], r["hip_err_max_conditioned"], r["worst_unconditioned"], r["loss_err"]))
    nat = [r for r in recs if r.get("native_fp32")]
    if nat:
        lines += ["", "## HIP forward vs the oracle's OWN fp32 forward (no shared ReLU masks)", "",
                  "`north_star`: \"the same logits / loss as the reference CPU path within 1e-3 rel fp32\".  Logits in "
                  "the max norm relative to the largest logit; `witness` = the fp32 oracle against the fp64 pass "
                  "(what two fp32 evaluations of this network may differ by); bound 1e-3 where witness <= 1e-3 / 3, "
                  "else 3 x witness (tests/parity.py `check_native_fp32`).", "",
                  "| test | head | HIP vs fp32 oracle | witness (fp32 oracle vs fp64) | loss_seg rel err |",
                  "|---|---|---|---|---|"]
        for r in nat:
            n = r["native_fp32"]
            for k, v in n.items():
                if k.startswith("logits."):
                    head = k[len("logits."):]
                    loss = n.get("loss.%s.loss_seg" % head)
                    lines.append("| %s | %s | %.2e | %.2e | %s |" % (
                        r["test"].split("::")[-1], head, v[0], v[1], "%.1e" % loss if loss is not None else "-"))
    single = [r["test"].split("::")[-1] for r in recs if r.get("single_source")]
    if single:
        lines += ["", "Single-source steps (p90 / p10 of the ratios < 1.25: every conditioned gradient inherits one "
                  "upstream error, so the step's median is one random draw; held to the factor 3 and to the pooled "
                  "median): " + ", ".join(single) + "."]
    n_c = sum(r["conditioned"] for r in recs)
    lines += ["", "%d compared steps, %d conditioned parameters in all; largest ratio %.2f, largest median %.2f."
              % (len(recs), n_c, max((r["ratio_max"] for r in recs), default=0.0),
                 max((r["ratio_median"] for r in recs), default=0.0))]
    open(dst, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
