#!/usr/bin/env python
"""Extract standalone subnets from a supernet checkpoint — same flow as the reference's
tools/extract_subnet.py:54-152: swap every norm to DynBN, build, load the supernet checkpoint,
``model.eval(); model.deploy()``, then for every meta of ``train_sampler.traverse()`` (sharded
over ranks): ``manipulate_arch`` -> deepcopy -> one dummy forward, during which the deploy-mode
modules physically prune themselves (leading slices, first ``depth`` blocks) -> save a
checkpoint named by the md5 of the meta.
"""
import argparse
import copy
import hashlib
import json
import os
import os.path as osp
import sys

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))

import torch  # noqa: E402

from gaia_seg_amd.core.checkpoint import load_checkpoint, save_checkpoint  # noqa: E402
from gaia_seg_amd.core.config import Config, DictAction  # noqa: E402
from gaia_seg_amd.core.dynamic import fold_dict  # noqa: E402
from gaia_seg_amd.core.model_space import build_model_sampler  # noqa: E402
from gaia_seg_amd.models import build_segmentor  # noqa: E402


def prepare_cfg(cfg):
    """tools/extract_subnet.py:54-62: every norm becomes a (rank-local) DynBN."""
    def swap(d):
        if isinstance(d, dict):
            for k, v in d.items():
                if k == "norm_cfg" and isinstance(v, dict):
                    d[k] = dict(type="DynBN", requires_grad=v.get("requires_grad", True))
                else:
                    swap(v)
        elif isinstance(d, (list, tuple)):
            for v in d:
                swap(v)
    swap(cfg.model)
    return cfg


def meta_hash(meta):
    return hashlib.md5(json.dumps(meta, sort_keys=True).encode()).hexdigest()[:8]


def extract(model, meta, input_size=64):
    """One pruned copy of ``model`` (already in deploy mode) for ``meta``."""
    model.manipulate_arch(fold_dict(meta)["arch"])
    sub = copy.deepcopy(model)
    sub.deploy()
    dev = next(sub.parameters()).device
    with torch.no_grad():
        sub.forward_dummy(torch.zeros(1, 3, input_size, input_size, device=dev))
    sub.deploy(False)
    return sub


def main():
    ap = argparse.ArgumentParser(description="Extract subnets from a supernet checkpoint")
    ap.add_argument("config")
    ap.add_argument("checkpoint")
    ap.add_argument("--work-dir", default="./work_dirs/extract")
    ap.add_argument("--cfg-options", nargs="+", default=None)
    ap.add_argument("--input-size", type=int, default=64)
    args = ap.parse_args()
    cfg = Config.fromfile(args.config)
    if args.cfg_options:
        cfg.merge_from_dict(DictAction.parse(args.cfg_options))
    cfg = prepare_cfg(cfg)
    model = build_segmentor(cfg.model, train_cfg=cfg.get("train_cfg"), test_cfg=cfg.get("test_cfg"))
    ck = load_checkpoint(model, args.checkpoint, strict=False)
    model = model.cuda().eval()
    model.deploy()
    sampler = build_model_sampler(cfg.train_sampler)
    metas = sampler.traverse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    os.makedirs(args.work_dir, exist_ok=True)
    for meta in metas[rank::world]:
        sub = extract(model, meta, args.input_size)
        name = meta_hash(meta)
        save_checkpoint(sub, osp.join(args.work_dir, "%s.pth" % name),
                        meta=dict(ck.get("meta", {}), **{k: v for k, v in meta.items()}))
        print("saved %s (%s): %.2f M parameters" % (
            name, meta.get("name", "subnet"), sum(p.numel() for p in sub.parameters()) / 1e6))


if __name__ == "__main__":
    main()
