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


# ---- K17: the fused slide / whole epilogue kernel against the oracle's restatement ------------
class _FakeSegmentor:
    """encode_decode() hands out pre-drawn low-resolution logits, window after window, up-sampled
    to the crop like the reference's encode_decode (dynamic_distiller.py:252-262)."""

    class decode_head:
        align_corners = False

    def __init__(self, low, align):
        self.low, self.k = low, 0
        self.decode_head = type("H", (), {"align_corners": align})

    def encode_decode(self, crop):
        out = F.interpolate(self.low[self.k], size=crop.shape[2:], mode="bilinear",
                            align_corners=self.decode_head.align_corners)
        self.k += 1
        return out


@pytest.mark.parametrize("case", [
    dict(h=96, w=128, crop=(64, 64), stride=(40, 40), low=(8, 8), C=19),
    dict(h=100, w=100, crop=(33, 47), stride=(17, 29), low=(5, 6), C=19, ori=(75, 131)),
    dict(h=50, w=70, crop=(64, 64), stride=(32, 32), low=(7, 9), C=19, flip="horizontal"),
    dict(h=65, w=129, crop=(64, 64), stride=(64, 64), low=(8, 8), C=150, flip="vertical", ori=(97, 200)),
    dict(h=64, w=96, crop=None, stride=None, low=(8, 12), C=19, align=True, ori=(128, 100)),
    dict(h=64, w=96, crop=None, stride=None, low=(2, 3), C=37, probs_in=True),
])
def test_slide_fuse_kernel_matches_oracle(hip_lib, case):
    from gaia_seg_amd.core.inference import FusedInference, window_axes
    from oracle import inference as OI
    h, w, C = case["h"], case["w"], case["C"]
    align = case.get("align", False)
    mode = "slide" if case["crop"] else "whole"
    if mode == "slide":
        ys, xs, hc, wc = window_axes(h, w, case["crop"], case["stride"])
    else:
        ys, xs, hc, wc = [0], [0], h, w
    n = 2
    g = torch.Generator().manual_seed(3)
    low = torch.randn(len(ys) * len(xs), n, C, *case["low"], generator=g) * 2
    img = torch.zeros(n, 3, h, w)
    ori = case.get("ori", (h, w))
    meta = dict(ori_shape=ori + (3,), flip=bool(case.get("flip")), flip_direction=case.get("flip"))
    test_cfg = dict(mode=mode)
    if mode == "slide":
        test_cfg.update(crop_size=case["crop"], stride=case["stride"])
    want = OI.inference(_FakeSegmentor(low, align), img, meta, test_cfg, rescale=True)
    probs_in = None
    if case.get("probs_in"):
        probs_in = torch.rand(n, C, *ori, generator=g)
        want = want + probs_in

    calls = []

    def logits_fn(batch):
        # windows arrive as one window-major batch: hand back the matching low-resolution maps
        assert batch.shape[0] == low.shape[0] * n and tuple(batch.shape[2:]) == (hc, wc)
        calls.append(1)
        return low.reshape(-1, C, *case["low"]).cuda()
    eng = FusedInference(C, align)
    labels, probs = eng(logits_fn, img.cuda(), mode=mode, crop_size=case["crop"], stride=case["stride"],
                        out_size=ori, flip=case.get("flip"),
                        probs_in=probs_in.cuda() if probs_in is not None else None, want_probs=True)
    assert len(calls) == 1
    assert tuple(probs.shape) == tuple(want.shape)
    assert float((probs.cpu() - want).abs().max()) < 2e-6
    differ = labels.cpu() != want.argmax(1)
    if bool(differ.any()):
        top2 = want.topk(2, dim=1).values
        assert float((top2[:, 0] - top2[:, 1])[differ].max()) < 1e-5
    # labels-only launch (no probabilities in memory) gives the same map
    labels2, none = eng(logits_fn, img.cuda(), mode=mode, crop_size=case["crop"], stride=case["stride"],
                        out_size=ori, flip=case.get("flip"))
    assert none is None
    if probs_in is None:
        d2 = labels2.cpu() != want.argmax(1)
        if bool(d2.any()):
            top2 = want.topk(2, dim=1).values
            assert float((top2[:, 0] - top2[:, 1])[d2].max()) < 1e-5


@pytest.mark.parametrize("case", [
    dict(h=96, w=128, crop=(64, 64), stride=(40, 40), low=(8, 8)),                       # 2 x 3 windows, overlaps
    dict(h=50, w=71, crop=(33, 47), stride=(17, 24), low=(5, 6)),                        # ragged last strip (71 % 4 = 3)
    dict(h=50, w=70, crop=(64, 64), stride=(32, 32), low=(7, 9), flip="horizontal"),     # mirrored strips, ragged
    dict(h=65, w=129, crop=(64, 64), stride=(64, 64), low=(8, 8), flip="vertical"),
    dict(h=64, w=96, crop=None, stride=None, low=(8, 12), align=True),                   # whole mode, align_corners
    dict(h=128, w=256, crop=(64, 128), stride=(43, 85), low=(8, 16)),                    # config 5's 3 x 3 grid, scaled down
])
def test_label_strip_kernel_equals_the_per_pixel_kernel(hip_lib, case):
    """The label-only, un-rescaled epilogue walks strips of four pixels that share a low-resolution
    cell (csrc/inference.hip slide_label_strip_kernel); its label map equals the per-pixel kernel's bit
    for bit -- same expressions, same window order, same division by the cover count."""
    from gaia_seg_amd.core.inference import FusedInference
    h, w = case["h"], case["w"]
    mode = "slide" if case["crop"] else "whole"
    n, C = 2, 19
    from gaia_seg_amd.core.inference import window_axes
    nwin = 1
    if mode == "slide":
        ys, xs, hc, wc = window_axes(h, w, case["crop"], case["stride"])
        nwin = len(ys) * len(xs)
    g = torch.Generator().manual_seed(11)
    low = (torch.randn(nwin * n, C, *case["low"], generator=g) * 2).cuda()
    img = torch.zeros(n, 3, h, w, device="cuda")
    eng = FusedInference(C, case.get("align", False))
    maps = []
    try:
        for strip in (1, 0):
            assert hip_lib.gs_debug_set_slide_strip(strip) == 0
            labels, none = eng(lambda batch: low, img, mode=mode, crop_size=case["crop"],
                               stride=case["stride"], out_size=(h, w), flip=case.get("flip"))
            assert none is None
            maps.append(labels.cpu())
    finally:
        hip_lib.gs_debug_set_slide_strip(-1)
    assert torch.equal(maps[0], maps[1])
    assert int(maps[0].min()) >= 0 and int(maps[0].max()) < C and maps[0].unique().numel() > 3


def test_slide_fuse_rejects_uncovered_images(hip_lib):
    import ctypes
    from gaia_seg_amd.hip import lib
    L = lib.load()
    d = lib.SlideDesc()
    d.N, d.C, d.ld, d.hl, d.wl, d.hc, d.wc = 1, 19, 20, 4, 4, 32, 32
    d.H, d.W, d.Ho, d.Wo, d.ny, d.nx = 64, 64, 64, 64, 1, 2
    low = torch.zeros(2, 1, 4, 4, 20, device="cuda")
    out = torch.zeros(1, 64, 64, dtype=torch.int64, device="cuda")
    ys, xs = (ctypes.c_int32 * 1)(0), (ctypes.c_int32 * 2)(0, 32)
    rc = L.gs_slide_fuse(ctypes.byref(d), ys, xs, low.data_ptr(), None, None, out.data_ptr(), None)
    assert rc == -1      # rows 32..63 are covered by no window: GS_E_BADARG


def test_aug_test_averages_view_probabilities(hip_lib):
    """aug_test over (plain, horizontally flipped) views == argmax of the mean probabilities, the
    flipped view flipped back (dynamic_distiller.py:523-540)."""
    from oracle import inference as OI
    cfg = model_cfg(fcn_head(), aux=True)
    prod, orc = make_pair(cfg)
    prod = prod.cuda().eval()
    orc.eval()
    meta = arch_meta("sub")
    prod.manipulate_arch(meta)
    orc.manipulate_arch(meta)
    img, _ = make_batch(1, 64, 96)
    imgs = [img, img.flip(3)]
    metas = [[dict(ori_shape=(80, 120, 3), flip=False)],
             [dict(ori_shape=(80, 120, 3), flip=True, flip_direction="horizontal")]]
    with torch.no_grad():
        got = prod(img=[i.cuda() for i in imgs], img_metas=metas, return_loss=False)
        want = OI.aug_test(orc, imgs, [m[0] for m in metas], dict(mode="whole"))
    got = torch.from_numpy(np.stack(got))
    assert tuple(got.shape) == (1, 80, 120)
    assert float((got != want).float().mean()) < 2e-3
