"""End-to-end parity of the MI355X model path (HIP kernels through the C-ABI) against the CPU
oracle on the same seeded inputs: features, losses, accuracy, parameter gradients, BN running
statistics — for the three decode heads, 7x7 and deep stems, OS32 and OS8 (dilated) backbones and
several subnets of one supernet.  Tolerance from BASELINE.json: 1e-3 relative (fp32), max norm, for
forward quantities and gradients alike (see tests/parity.py for how the gradient comparison is made
independent of rounding-level ReLU branch flips).  BASELINE configs at their stated sizes:
tests/test_baseline_configs_gpu.py."""
import copy

import pytest
import torch

from conftest import rel_err
from parity import train_step_parity
from util_models import (arch_meta, fcn_head, make_batch, make_pair, model_cfg, psp_head, uper_head)

pytestmark = pytest.mark.gpu
TOL = 1e-3        # forward quantities AND gradients, max norm (BASELINE.json: 1e-3 rel fp32)


def _run_pair(cfg, arch, deep_stem=False, size=(64, 96), check_grads=True):
    """One HIP train step against one fp64 oracle pass on the HIP path's ReLU branch pattern
    (tests/parity.py): losses, accuracy, BN statistics and every parameter gradient at 1e-3 max norm."""
    prod, orc = make_pair(cfg)
    prod = prod.cuda().train()
    orc.train()
    meta = arch_meta(arch, deep_stem)
    prod.manipulate_arch(meta)
    orc.manipulate_arch(meta)
    img, gt = make_batch(2, *size)
    return train_step_parity(prod, orc, img, gt, check_grads=check_grads)


@pytest.mark.parametrize("arch", ["max", "sub", "min"])
def test_fcn_supernet_train_step(hip_lib, arch):
    _run_pair(model_cfg(fcn_head(), aux=True), arch)


def test_fcn_deep_stem_os8(hip_lib):
    _run_pair(model_cfg(fcn_head(), aux=True, deep_stem=True, os8=True), "sub", deep_stem=True)


@pytest.mark.parametrize("arch", ["max", "sub"])
def test_psp_supernet_train_step(hip_lib, arch):
    _run_pair(model_cfg(psp_head(), aux=True), arch)


def test_psp_os8(hip_lib):
    _run_pair(model_cfg(psp_head(), aux=True, os8=True), "sub", size=(64, 64))


@pytest.mark.parametrize("arch", ["max", "sub"])
def test_uper_supernet_train_step(hip_lib, arch):
    _run_pair(model_cfg(uper_head(), aux=False), arch, size=(97, 97))


def test_fcn_avg_down_shortcuts(hip_lib):
    """avg_down=True (dynamic_res_layer.py:75-82): AvgPool2d(ceil_mode, count_include_pad=False)
    + stride-1 1x1 shortcut; 63x95 makes every pooled border window ragged."""
    _run_pair(model_cfg(fcn_head(), aux=True, deep_stem=True, avg_down=True), "sub", deep_stem=True,
              size=(63, 95))


def test_backbone_features_and_depth_prefix(hip_lib):
    """depth d == first d blocks; features equal the oracle's on every output level."""
    prod, orc = make_pair(model_cfg(fcn_head(), aux=False))
    prod = prod.cuda().eval()
    orc.eval()
    img, _ = make_batch(2, 64, 64)
    for arch in ["max", "sub", "min"]:
        meta = arch_meta(arch)
        prod.manipulate_arch(meta)
        orc.manipulate_arch(meta)
        with torch.no_grad():
            fp = prod.extract_feat(img.cuda())
            fo = orc.backbone(img)
        for a, b in zip(fp, fo):
            assert a.shape == b.shape
            assert rel_err(a, b) < TOL


def test_slice_equals_standalone_subnet(hip_lib):
    """Metamorphic (the equivalence tools/extract_subnet.py relies on): a subnet run through the
    supernet with manipulate_arch equals a network BUILT at subnet size whose weights are the
    leading slices.  Both sides are the HIP path with identical GEMM shapes, so the match is
    exact up to nothing: we require 1e-6."""
    import copy
    from gaia_seg_amd.models import build_segmentor
    from util_models import ARCHS
    cfg = model_cfg(fcn_head(), aux=True)
    sup, _ = make_pair(cfg)
    a = ARCHS["sub"]
    sub_cfg = copy.deepcopy(cfg)
    sub_cfg["backbone"].update(stem_width=a["stem"], body_width=list(a["width"]),
                               body_depth=list(a["depth"]))
    sub_cfg["decode_head"]["in_channels"] = 4 * a["width"][3]
    sub_cfg["auxiliary_head"]["in_channels"] = 4 * a["width"][2]
    sub = build_segmentor(sub_cfg)
    sd_sup = sup.state_dict()
    sd = {}
    for k, v in sub.state_dict().items():
        src = sd_sup[k]
        sd[k] = src[tuple(slice(0, s) for s in v.shape)].clone() if v.dim() else src.clone()
    sub.load_state_dict(sd)
    sup, sub = sup.cuda().train(), sub.cuda().train()
    sup.manipulate_arch(arch_meta("sub"))
    img, gt = make_batch(2, 64, 96)
    metas = [dict(ori_shape=(64, 96, 3), flip=False)] * 2
    batch = dict(img=img.cuda(), img_metas=metas, gt_semantic_seg=gt.cuda())
    o1 = sup.train_step(batch, None)
    o2 = sub.train_step(batch, None)
    o1["loss"].backward()
    o2["loss"].backward()
    assert abs(float(o1["loss"]) - float(o2["loss"])) <= 1e-6 * abs(float(o2["loss"]))
    p_sup = dict(sup.named_parameters())
    for name, p in sub.named_parameters():
        g_sup = p_sup[name].grad[tuple(slice(0, s) for s in p.shape)]
        assert rel_err(g_sup, p.grad) < 1e-6, name


def test_overfits_a_learnable_batch(hip_lib):
    """End-to-end sanity of the optimised training loop (fused conv+BN calls, statistics from the
    conv, side-stream weight gradients, fused SGD over the arena): on a fixed batch whose labels are
    a function of the image, the loss must fall well below chance within a few dozen steps."""
    from gaia_seg_amd.core.dist import GradReducer
    from gaia_seg_amd.core.param_arena import ParamArena
    from gaia_seg_amd.core.runner import ArenaOptimizerHook, IterBasedRunner
    from gaia_seg_amd.models import build_segmentor
    torch.manual_seed(0)
    model = build_segmentor(copy.deepcopy(model_cfg(fcn_head(), aux=True))).cuda().train()
    model.manipulate_arch(arch_meta("sub"))
    arena = ParamArena(model)
    runner = IterBasedRunner(model, arena, GradReducer(arena.flat_grad, arena.segments), base_lr=0.02,
                             momentum=0.9, weight_decay=1e-4, max_iters=100)
    runner.set_arch(None)   # active parameter ranges of the arch set above
    runner.register_hook(ArenaOptimizerHook())
    runner.call_hook("before_run")
    n, h, w = 2, 64, 96
    img = torch.randn(n, 3, h, w)
    # label = coarse quantisation of a smoothed image channel: 6 classes, spatially coherent
    sm = torch.nn.functional.avg_pool2d(img[:, :1], 9, 1, 4)
    gt = ((sm - sm.min()) / (sm.max() - sm.min() + 1e-6) * 5.999).long()
    metas = [dict(ori_shape=(h, w, 3), img_shape=(h, w, 3), flip=False) for _ in range(n)]
    batch = dict(img=img.cuda(), img_metas=metas, gt_semantic_seg=gt.cuda())
    losses = []
    for _ in range(60):
        out = runner.train_iter(batch)
        losses.append(float(out["log_vars"]["decode.loss_seg"]))
    assert all(l == l for l in losses)            # finite
    assert losses[-1] < 0.5 * losses[0], losses[::10]
