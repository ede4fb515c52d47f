#!/usr/bin/env python
"""Per-subnet FLOPs / params table (the reference's tools/count_flops.py:63-179 writes the same
kind of model-space file, `flops.json`, sharded over ranks)."""
import argparse
import json
import os
import os.path as osp
import sys

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))

from gaia_seg_amd.core.config import Config  # noqa: E402
from gaia_seg_amd.core.dynamic import fold_dict  # noqa: E402
from gaia_seg_amd.core.flops import model_flops  # noqa: E402
from gaia_seg_amd.core.model_space import build_model_sampler  # noqa: E402
from gaia_seg_amd.models import build_segmentor  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--shape", type=int, nargs=2, default=[512, 2048])  # count_flops.py:139-140
    ap.add_argument("--out", default="flops.json")
    ap.add_argument("--sampler", default="val_sampler")
    args = ap.parse_args()
    cfg = Config.fromfile(args.config)
    model = build_segmentor(cfg.model, train_cfg=cfg.get("train_cfg"), test_cfg=cfg.get("test_cfg"))
    metas = build_model_sampler(cfg[args.sampler]).traverse()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    rows = []
    for meta in metas[rank::world]:
        model.manipulate_arch(fold_dict(meta)["arch"])
        f = model_flops(model, *args.shape)
        rows.append(dict(meta, **{"overhead.flops": f["total"], "overhead.backbone_flops": f["backbone"],
                                  "overhead.params": f["backbone_params"]}))
        print("%-8s total %.1f GF  backbone %.1f GF (3x3 %.1f)  params %.2f M" % (
            meta.get("name", "?"), f["total"] / 1e9, f["backbone"] / 1e9, f["backbone_3x3"] / 1e9,
            f["backbone_params"] / 1e6))
    with open(args.out, "w") as fh:
        json.dump(rows, fh, indent=1)


if __name__ == "__main__":
    main()
