"""BASELINE config 5 pieces on the GPU: OHEM pixel sampling + aux head in training, and
whole-image / sliding-window inference with eval-mode BatchNorm, against the oracle."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from util_models import arch_meta, fcn_head, make_batch, make_pair, model_cfg

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("thresh", [0.7, None])
def test_ohem_sampler_matches_mmseg_rule(hip_lib, thresh):
    from oracle.ops import ohem_pixel_weights
    from gaia_seg_amd.models.pixel_samplers import OHEMPixelSampler

    class Ctx:
        ignore_index, align_corners = 255, False

    torch.manual_seed(0)
    logits = torch.randn(2, 19, 16, 32) * 2
    label = torch.randint(0, 19, (2, 1, 64, 128))
    label[:, :, :4] = 255
    up = F.interpolate(logits, size=(64, 128), mode="bilinear", align_corners=False)
    want = ohem_pixel_weights(up, label, thresh=thresh, min_kept=1000)
    s = OHEMPixelSampler(Ctx(), thresh=thresh, min_kept=1000)
    lg = logits.cuda().contiguous(memory_format=torch.channels_last)
    got = s.sample(lg, label.cuda()).cpu()
    # ties at the k-th probability are measure-zero for random logits
    assert int((got != want).sum()) <= 2, int((got != want).sum())
    assert int(want.sum()) > 0


def test_ohem_train_step_matches_oracle(hip_lib):
    from oracle.ops import ohem_pixel_weights
    head = fcn_head()
    head["sampler"] = dict(type="OHEMPixelSampler", thresh=0.7, min_kept=500)
    cfg = model_cfg(head, aux=True)
    ocfg = copy.deepcopy(cfg)
    ocfg["decode_head"].pop("sampler")
    prod, orc = make_pair(cfg, ocfg=ocfg)
    orc.decode_head.sampler = lambda lg, lb: ohem_pixel_weights(lg, lb, thresh=0.7, min_kept=500)
    prod = prod.cuda().train()
    orc.train()
    meta = arch_meta("sub")
    prod.manipulate_arch(meta)
    orc.manipulate_arch(meta)
    img, gt = make_batch(2, 64, 96)
    lo, _ = orc.parse_losses(orc.forward_train(img, gt))
    metas = [dict(ori_shape=(64, 96, 3), flip=False)] * 2
    out = prod.train_step(dict(img=img.cuda(), img_metas=metas, gt_semantic_seg=gt.cuda()), None)
    assert abs(float(out["loss"]) - float(lo)) < 2e-3 * abs(float(lo))


@pytest.mark.parametrize("mode", ["whole", "slide"])
def test_inference_matches_oracle(hip_lib, mode):
    cfg = model_cfg(fcn_head(), aux=True)
    if mode == "slide":
        cfg["test_cfg"] = dict(mode="slide", crop_size=(64, 64), stride=(40, 40))
    prod, orc = make_pair(cfg)
    from gaia_seg_amd.core.config import ConfigDict
    prod.test_cfg = ConfigDict(cfg["test_cfg"])
    prod = prod.cuda().eval()
    orc.eval()
    meta = arch_meta("sub")
    prod.manipulate_arch(meta)
    orc.manipulate_arch(meta)
    img, _ = make_batch(1, 96, 128)
    with torch.no_grad():
        if mode == "whole":
            want = orc.encode_decode(img).argmax(1)
        else:
            preds = torch.zeros(1, 19, 96, 128)
            cnt = torch.zeros(1, 1, 96, 128)
            for y1, x1 in [(0, 0), (0, 40), (0, 64), (32, 0), (32, 40), (32, 64)]:
                crop = img[:, :, y1:y1 + 64, x1:x1 + 64]
                preds[:, :, y1:y1 + 64, x1:x1 + 64] += orc.encode_decode(crop)
                cnt[:, :, y1:y1 + 64, x1:x1 + 64] += 1
            want = (preds / cnt).argmax(1)
        metas = [dict(ori_shape=(96, 128, 3), img_shape=(96, 128, 3), flip=False)]
        got = prod(img=[img.cuda()], img_metas=[metas], return_loss=False)
    got = torch.from_numpy(np.stack(got))
    mismatch = float((got != want).float().mean())
    assert mismatch < 2e-3, mismatch   # argmax ties / near-ties only
