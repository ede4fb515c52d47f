"""Generates tests/golden/bench_check.json: the CPU oracle's losses for the step bench.py verifies
before it times anything (the first timed step's subnet and batch, evaluated from the initial
weights of seed S with dropout off).  Run in the build container:

    python tests/golden/make_bench_check.py            # default bench configuration
    python tests/golden/make_bench_check.py --arch R50 # extra entries for `bench.py --arch R50`

The product model is only BUILT here (on the CPU, to reproduce bench.py's seeded initialisation); the
numbers come from oracle/model.py in float64."""
import argparse
import json
import os
import random
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def _plain(obj):
    if isinstance(obj, dict):
        return {k: _plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_plain(v) for v in obj]
    return obj


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default=os.path.join(ROOT, "configs/supernet/fcn_ar50to101v2.py"))
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--arch", default="sample")
    args = ap.parse_args()
    from gaia_seg_amd.core.config import Config
    from gaia_seg_amd.core.dynamic import fold_dict
    from gaia_seg_amd.core.model_space import build_model_sampler
    from gaia_seg_amd.core.synthetic import make_batch
    from gaia_seg_amd.models import build_segmentor
    from oracle.model import OEncoderDecoder

    cfg = Config.fromfile(args.config)
    torch.manual_seed(args.seed)
    random.seed(args.seed)
    prod = build_segmentor(cfg.model, train_cfg=cfg.get("train_cfg"), test_cfg=cfg.get("test_cfg"))
    sampler = build_model_sampler(cfg.train_sampler)
    sampler.seed(args.seed)
    if args.arch == "sample":
        meta = sampler.sample()
    else:
        meta = dict({a["name"]: a for a in sampler.model_samplers[0].anchors}[args.arch])
        if cfg.get("stem_anchors"):   # deep-stem (v1c / OS8) supernet: a width per stem conv (as bench.py)
            meta["arch.backbone.stem.width"] = list(cfg["stem_anchors"][args.arch])
    bs = cfg.data["samples_per_gpu"]
    h, w = cfg.crop_size
    mcfg = {k: v for k, v in _plain(cfg.model).items() if k != "type"}
    for head in ("decode_head", "auxiliary_head"):
        if mcfg.get(head):
            mcfg[head]["dropout_ratio"] = 0.0
    orc = OEncoderDecoder(**mcfg)
    orc.load_state_dict({k: v.detach().clone().contiguous() for k, v in prod.state_dict().items()},
                        strict=True)
    csum = float(sum(p.detach().double().abs().sum() for p in prod.parameters()))
    orc.double().train()
    orc.manipulate_arch(fold_dict(meta)["arch"])
    batch = make_batch(bs, h, w, 19, args.seed * 1000003)
    with torch.no_grad():
        losses = orc.forward_train(batch["img"].double(), batch["gt_semantic_seg"])
        loss, log_vars = orc.parse_losses(losses)
    log_vars = {k: float(v) for k, v in log_vars.items()}
    log_vars["loss"] = float(loss)
    key = "%s|seed%d|%dx%d|bs%d|%s" % (os.path.basename(args.config), args.seed, h, w, bs,
                                       meta.get("name", "random"))
    path = os.path.join(HERE, "bench_check.json")
    data = {}
    if os.path.exists(path):
        with open(path) as f:
            data = json.load(f)
    data[key] = dict(log_vars=log_vars, param_abs_sum=csum, arch=_plain(meta),
                     torch=torch.__version__.split("+")[0], oracle="oracle/model.py float64, dropout 0")
    with open(path, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
    print(key, log_vars)


if __name__ == "__main__":
    main()
