"""Generates tests/golden/ref_loss_utils.npz by RUNNING the reference's own loss utilities.

Only two reference modules are importable without the absent mmcv / mmseg / gaiavision stack:
gaiaseg/models/losses/utils.py and gaiaseg/models/losses/accuracy.py (pure torch).  They are loaded
by file path (the package __init__ chain would import mmcv).  Run in the build container:

    python tests/golden/make_ref_loss_fixtures.py

The reference never travels to the GPU box; only the .npz (inputs + expected outputs) does.
"""
import importlib.util
import os

import numpy as np
import torch

REF = "/root/reference/gaiaseg/models/losses"
HERE = os.path.dirname(os.path.abspath(__file__))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    utils = _load(os.path.join(REF, "utils.py"), "ref_loss_utils")
    acc = _load(os.path.join(REF, "accuracy.py"), "ref_accuracy")
    g = torch.Generator().manual_seed(1234)
    out = {}
    # accuracy: logits [N,C,H,W], labels with ignore value 255 and an exact-argmax region
    for i, (n, c, h, w) in enumerate([(2, 19, 7, 9), (1, 5, 4, 4), (3, 19, 16, 8)]):
        pred = torch.randn(n, c, h, w, generator=g)
        target = torch.randint(0, c, (n, h, w), generator=g)
        target[0, 0, :3] = 255
        target[-1, -1, :] = pred[-1, :, -1, :].argmax(0)
        out["acc%d_pred" % i] = pred.numpy()
        out["acc%d_target" % i] = target.numpy()
        out["acc%d_top1" % i] = acc.accuracy(pred, target).numpy()
        # (topk > 1 raises in the reference itself on torch >= 1.8: accuracy.py:47 uses .view on
        # a non-contiguous slice — only top-1, the one the heads use, can be pinned)
    # weight_reduce_loss: mean over all elements / weighted / avg_factor / sum / none
    for i, shape in enumerate([(2, 7, 9), (4, 3, 3)]):
        loss = torch.rand(shape, generator=g)
        weight = (torch.rand(shape, generator=g) > 0.4).float()
        out["wrl%d_loss" % i] = loss.numpy()
        out["wrl%d_weight" % i] = weight.numpy()
        out["wrl%d_mean" % i] = utils.weight_reduce_loss(loss).numpy()
        out["wrl%d_wmean" % i] = utils.weight_reduce_loss(loss, weight).numpy()
        out["wrl%d_sum" % i] = utils.weight_reduce_loss(loss, weight, reduction="sum").numpy()
        out["wrl%d_none" % i] = utils.weight_reduce_loss(loss, weight, reduction="none").numpy()
        out["wrl%d_avg" % i] = utils.weight_reduce_loss(loss, weight, avg_factor=17.0).numpy()
    np.savez_compressed(os.path.join(HERE, "ref_loss_utils.npz"), **out)
    print("wrote", os.path.join(HERE, "ref_loss_utils.npz"), len(out), "arrays")


if __name__ == "__main__":
    main()
