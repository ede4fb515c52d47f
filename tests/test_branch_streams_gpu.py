"""The projection-shortcut block on the branch stream and with its BatchNorm applied inside norm3's apply
pass (gs_bn_args.residual_coeffs) against (1) the same block on ONE stream with the shortcut's
normalised output written out — every output, running statistic and gradient bit for bit: the kernels,
operands and summation orders are the same, only the queue and one fused expression differ — and
(2) PyTorch on the CPU (gaiaseg/models/utils/dynamic_res_layer.py:70-125: first block of a stage, stride
on conv2, 1x1 projection shortcut or avg_down).  Also the auxiliary head on its branch stream at the
segmentor level: train_step + backward twice, with and without branches, bit for bit."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _block(inplanes, planes, stride, dilation, avg_down, seed):
    from gaia_seg_amd.core.bricks import DynamicBottleneck
    from gaia_seg_amd.models.utils.dynamic_res_layer import DynamicResLayer
    torch.manual_seed(seed)
    conv_cfg, norm_cfg = dict(type="DynConv2d"), dict(type="DynBN")
    ds = DynamicResLayer._shortcut(DynamicBottleneck, inplanes, planes, stride, avg_down, conv_cfg, norm_cfg)
    blk = DynamicBottleneck(inplanes, planes, stride=stride, dilation=dilation, downsample=ds,
                            conv_cfg=conv_cfg, norm_cfg=norm_cfg)
    for m in blk.modules():
        if isinstance(m, nn.modules.batchnorm._BatchNorm):
            nn.init.uniform_(m.weight, 0.5, 1.5)
            nn.init.normal_(m.bias, 0.0, 0.2)
    return blk


def _run(blk, x, gz, branch, defer, masks=None):
    import gaia_seg_amd.hip.ops as ops
    from gaia_seg_amd.hip.runtime import tape_function
    ops.BRANCH_SHORTCUT, ops.DEFER_SHORTCUT_BN = branch, defer
    ops.RELU_TRACE = [] if masks is not None else None
    for m in blk.modules():
        if isinstance(m, nn.modules.batchnorm._BatchNorm):
            m.running_mean.zero_()
            m.running_var.fill_(1.0)
    for p in blk.parameters():
        p.grad = None
    xg = x.clone().requires_grad_(True)
    z = tape_function(lambda tape, acts: [blk.forward_act(tape, acts[0])], [xg], True)[0]
    z.backward(gz)
    torch.cuda.synchronize()
    if masks is not None:      # the ReLU branches this run took, by BatchNorm (NHWC bool -> NCHW, CPU)
        by_gamma = {id(g): m for g, m in ops.RELU_TRACE}
        for name in ("norm1", "norm2", "norm3"):
            masks[name] = by_gamma[id(getattr(blk, name).weight)].permute(0, 3, 1, 2).cpu()
        ops.RELU_TRACE = None
    out = {"z": z.detach().clone(), "dx": xg.grad.clone()}
    for n, p in blk.named_parameters():
        out["g." + n] = p.grad.clone()
    for n, b in blk.named_buffers():
        if "running" in n:
            out["b." + n] = b.clone()
    return out


def _reference(blk, x, gz, masks):
    """The block in plain torch on the CPU (fp64), from the module's own parameters, evaluated on the
    ReLU branches the HIP run took (a pre-activation within fp32 rounding of zero may fall either way;
    the two sides must differentiate the same smooth function — tests/parity.py protocol)."""
    blk = blk.double()
    xr = x.double().requires_grad_(True)

    def cbn(conv, bn, t):
        y = F.conv2d(t, conv.weight, None, conv.stride, conv.padding, conv.dilation)
        return F.batch_norm(y, None, None, bn.weight, bn.bias, True, 0.1, bn.eps)

    def relu(y, name):
        own = y > 0
        flipped = own != masks[name]
        if flipped.any():      # only ties may differ
            assert float(y[flipped].abs().max() / y.pow(2).mean().sqrt()) < 1e-4, name
        return y * masks[name].double()
    ident = xr
    mods = list(blk.downsample)
    if isinstance(mods[0], nn.AvgPool2d):
        ident = mods.pop(0)(ident)
    ident = cbn(mods[0], mods[1], ident)
    o = relu(cbn(blk.conv1, blk.norm1, xr), "norm1")
    o = relu(cbn(blk.conv2, blk.norm2, o), "norm2")
    o = relu(cbn(blk.conv3, blk.norm3, o) + ident, "norm3")
    o.backward(gz.double())
    out = {"z": o.detach(), "dx": xr.grad}
    for n, p in blk.named_parameters():
        out["g." + n] = p.grad
    return out


CASES = [
    # inplanes planes stride dil avg_down  n   h   w
    (64, 64, 1, 1, False, 2, 40, 48),      # stage 1: stride-1 projection 64 -> 256
    (256, 128, 2, 1, False, 2, 32, 48),    # stage 2: strided projection (sparse data gradient)
    (128, 48, 2, 1, True, 2, 31, 37),      # avg_down with ragged sizes (ceil_mode pooling)
    (96, 80, 1, 2, False, 1, 24, 24),      # OS8: dilated conv2, stride-1 projection, 80-wide tiles
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_shortcut_on_the_branch_stream_and_in_norm3_apply(hip_lib, case):
    import copy
    import gaia_seg_amd.hip.ops as ops
    inplanes, planes, stride, dil, avg_down, n, h, w = case
    blk_cpu = _block(inplanes, planes, stride, dil, avg_down, seed=3)
    torch.manual_seed(17)
    x = torch.randn(n, inplanes, h, w)
    blk = copy.deepcopy(blk_cpu).to(DEV).train()
    xg = x.to(DEV).contiguous(memory_format=torch.channels_last)
    ho = (h + stride - 1) // stride if avg_down else (h - 1) // stride + 1
    wo = (w + stride - 1) // stride if avg_down else (w - 1) // stride + 1
    gz = torch.randn(n, planes * 4, ho, wo)
    keep = (ops.BRANCH_SHORTCUT, ops.DEFER_SHORTCUT_BN)
    try:
        masks = {}
        base = _run(blk, xg, gz.to(DEV), branch=False, defer=False, masks=masks)
        for branch, defer in ((True, False), (False, True), (True, True)):
            got = _run(blk, xg, gz.to(DEV), branch=branch, defer=defer)
            for k, v in base.items():
                assert torch.equal(got[k], v), (k, branch, defer)
    finally:
        ops.BRANCH_SHORTCUT, ops.DEFER_SHORTCUT_BN = keep
    ref = _reference(blk_cpu, x, gz, masks)
    assert rel_err(base["z"], ref["z"]) < 1e-4
    assert rel_err(base["dx"], ref["dx"]) < 2e-4
    for k, v in ref.items():
        if k.startswith("g."):
            assert rel_err(base[k], v) < 3e-4, k


def test_auxiliary_head_on_its_branch_stream_bitwise(hip_lib):
    """One training step (forward, both losses, backward) of a small FCN + aux-FCN supernet with all
    branches on equals the single-stream run bit for bit: losses, every parameter gradient, every
    BatchNorm running statistic."""
    import gaia_seg_amd.hip.ops as ops
    from util_models import fcn_head, model_cfg, randomize
    from gaia_seg_amd.models import build_segmentor
    torch.manual_seed(0)
    model = build_segmentor(model_cfg(fcn_head(), aux=True))
    randomize(model, 4)
    model = model.to(DEV).train()
    img = torch.randn(2, 3, 96, 128, device=DEV)
    gt = torch.randint(0, 19, (2, 1, 96, 128), device=DEV)
    gt[0, 0, :7] = 255
    state = {k: v.clone() for k, v in model.state_dict().items()}
    keep = (ops.BRANCH_SHORTCUT, ops.BRANCH_AUX, ops.DEFER_SHORTCUT_BN)
    runs = []
    try:
        for on in (False, True, True):
            ops.BRANCH_SHORTCUT = ops.BRANCH_AUX = on
            model.load_state_dict(state)
            for p in model.parameters():
                p.grad = None
            out = model.train_step(dict(img=img, img_metas=[{}, {}], gt_semantic_seg=gt))
            out["loss"].backward()
            torch.cuda.synchronize()
            rec = {"loss": out["loss"].detach().clone()}
            rec.update({"lv." + k: v.clone() for k, v in out["log_vars"].items()})
            rec.update({"g." + n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
            rec.update({"b." + n: b.clone() for n, b in model.named_buffers() if "running" in n})
            runs.append(rec)
    finally:
        ops.BRANCH_SHORTCUT, ops.BRANCH_AUX, ops.DEFER_SHORTCUT_BN = keep
    assert len(runs[0]) > 20 and any(k.startswith("g.auxiliary_head") for k in runs[0])
    for other in runs[1:]:
        assert other.keys() == runs[0].keys()
        for k, v in runs[0].items():
            assert torch.equal(other[k], v), k


def test_two_weight_gradient_streams_knob():
    """GS_SIDE_STREAMS=2 (an experiment knob, default 1: profiles/r04_stream_experiments.md) deals the
    weight-gradient jobs to two streams; the cases above and a model-level parity test pass under it
    (child interpreter: the stream count is read at import)."""
    import os
    import subprocess
    import sys
    if os.environ.get("GS_SIDE_STREAMS", "1") != "1":
        pytest.skip("already the GS_SIDE_STREAMS run")
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, GS_SIDE_STREAMS="2")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__),
                        os.path.join(here, "test_model_gpu.py"), "-q", "-m", "gpu", "-x", "-k",
                        "shortcut_on_the_branch or auxiliary_head_on or fcn_supernet_train_step"],
                       env=env, capture_output=True, text=True, cwd=os.path.dirname(here))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
