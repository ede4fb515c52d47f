"""Deploy / subnet extraction (reference tools/extract_subnet.py:65-152):
the physically pruned copy has the subnet's parameter shapes, reproduces the supernet-slice
forward exactly, loads into a network BUILT at subnet size, and the supernet itself is untouched."""
import copy
import os
import sys

import pytest
import torch

from conftest import rel_err
from util_models import ARCHS, arch_meta, fcn_head, make_batch, make_pair, model_cfg

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_extract_subnet_prunes_and_matches(hip_lib, tmp_path):
    from extract_subnet import extract, meta_hash
    from gaia_seg_amd.core.checkpoint import load_checkpoint, save_checkpoint
    from gaia_seg_amd.models import build_segmentor
    cfg = model_cfg(fcn_head(), aux=True)
    sup, _ = make_pair(cfg)
    sup = sup.cuda().eval()
    n_sup = sum(p.numel() for p in sup.parameters())
    a = ARCHS["sub"]
    meta = {"name": "sub", "arch.backbone.stem.width": a["stem"],
            "arch.backbone.body.width": list(a["width"]), "arch.backbone.body.depth": list(a["depth"])}
    img, _ = make_batch(2, 64, 96)
    sup.manipulate_arch(arch_meta("sub"))
    with torch.no_grad():
        want = sup.encode_decode(img.cuda(), None).clone()
    sup.deploy()
    sub = extract(sup, meta)
    sup.deploy(False)
    # pruned shapes
    assert sum(p.numel() for p in sub.parameters()) < n_sup
    assert sum(p.numel() for p in sup.parameters()) == n_sup          # supernet untouched
    assert len(sub.backbone.layer3) == a["depth"][2]
    assert tuple(sub.backbone.layer2[0].conv2.weight.shape) == (a["width"][1], a["width"][1], 3, 3)
    assert tuple(sub.decode_head.convs[0].conv.weight.shape)[1] == 4 * a["width"][3]
    with torch.no_grad():
        got = sub.encode_decode(img.cuda(), None)
    assert rel_err(got, want) < 1e-6
    # round trip through the reference checkpoint format into a standalone network
    path = os.path.join(str(tmp_path), meta_hash(meta) + ".pth")
    save_checkpoint(sub, path, meta=meta)
    sub_cfg = copy.deepcopy(cfg)
    sub_cfg["backbone"].update(stem_width=a["stem"], body_width=list(a["width"]), body_depth=list(a["depth"]))
    sub_cfg["decode_head"]["in_channels"] = 4 * a["width"][3]
    # like the reference, the dummy forward (forward_dummy -> encode_decode) never runs the
    # auxiliary head, so it is exported at supernet size
    alone = build_segmentor(sub_cfg)
    load_checkpoint(alone, path, strict=True)
    alone = alone.cuda().eval()
    with torch.no_grad():
        got2 = alone.encode_decode(img.cuda(), None)
    assert rel_err(got2, want) < 1e-6
