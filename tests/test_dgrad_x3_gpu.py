"""Operator-level parity of the bf16x3 data-gradient K loop (csrc/igemm_core.h: x3_k_loop), the default
contraction of every stride-1 DynConv2d data gradient whose grid fills the chip
(gaiaseg/models/utils/dynamic_res_layer.py:105-125: the autograd dgrad of conv1 / conv2 / conv3).

Every case here is sized past the dispatch gate (64-row tiles, 64- or 48-wide columns, >= 4 K steps
per workgroup, >= 512 workgroups, split-K only with >= 48 K steps per split) and ASSERTS through
gs_debug_last_conv_launch that the launch really ran on GS_KLOOP_BF16X3 before its result is compared
with PyTorch's conv2d gradient on the CPU at the fp32 tolerance of the other operator tests (3e-5).
The calls go through the C-ABI (gs_conv2d_dgrad) on raw buffers, so pixel strides wider than the
channel count (concat slices), leading slices of wider weights and accumulation are under test too.

With GS_X3=0 the same cases must dispatch the fp32 MFMA loop and pass the same bar:
test_same_cases_on_the_fp32_loop re-runs this file in a child interpreter (the switch is read once)."""
import ctypes
import os
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 3e-5
X3_ON = os.environ.get("GS_X3", "4") != "0"


def _desc(lib, n, h, w, ci, co, k, dil, ci_max, co_ld, ldx, ldy):
    p = dil * (k // 2)
    return lib.ConvDesc(N=n, H=h, W=w, Ci=ci, Co=co, Ci_max=ci_max, Co_ld=co_ld, KH=k, KW=k, stride=1,
                        pad=p, dil=dil, Ho=h, Wo=w, x_sn=h * w * ldx, x_sh=w * ldx, x_sw=ldx, x_sc=1,
                        ldy=ldy, ld_add=0, role=0, reserved=0, in_affine=None)


def _launch_record(lib, L):
    rec = lib.DebugLaunch()
    assert L.gs_debug_last_conv_launch(ctypes.byref(rec)) == 0
    return rec


def _expect_kloop(lib, rec):
    if X3_ON:
        assert rec.kloop == lib.KLOOP_BF16X3, "case does not reach the bf16x3 loop (kloop %d)" % rec.kloop
    else:
        assert rec.kloop in (lib.KLOOP_FP32, lib.KLOOP_FP32_PAIRS), rec.kloop


# n  h   w   ci  co  k dil ci_max co_ld ldx ldy acc  force_plan            what it covers
X3_CASES = [
    # (1x1 data gradients over >= 16384 rows with a short contraction go to the streaming kernel,
    # tests/test_stream_1x1_gpu.py; the bf16x3 loop keeps the 1x1s of stages 3-4: few rows, wide outputs)
    (2, 63, 65, 512, 64, 1, 1, 512, 64, 512, 64, 0, None),       # 1x1, bn 64, ragged M tail (8190 % 64 = 62), 4 K steps = 2 bf16 steps
    (2, 128, 136, 48, 48, 3, 1, 48, 48, 48, 48, 0, None),        # 3x3, bn 48, odd nk16 = 27 (zero-padded last half step), step pairs cross taps (3 per tap)
    (2, 128, 128, 64, 80, 3, 2, 64, 80, 64, 80, 0, None),        # dilation 2 (OS8 stages), Co = 80: 5 K steps per tap, odd total 45
    (2, 64, 64, 1024, 256, 1, 1, 1024, 256, 1024, 256, 1, None), # stage-3 conv1's dgrad: 16 column tiles, accumulate onto the identity gradient
    (2, 64, 64, 512, 256, 1, 1, 640, 320, 512, 256, 0, None),    # 1x1 from a leading slice of a wider weight
    (2, 64, 64, 128, 256, 3, 1, 128, 256, 128, 256, 0, None),    # split-K 3 x 48 K steps: slabs + fixed-order reduce
    (2, 64, 128, 192, 192, 3, 1, 192, 192, 192, 192, 1, None),   # long unsplit K (108 steps), 3 column tiles, accumulate
    (2, 128, 136, 48, 144, 3, 1, 96, 160, 112, 208, 1, (64, 48, 1)),   # dx and dy are channel slices of wider buffers (ld > C), bn 48, accumulate
    (1, 181, 183, 64, 48, 3, 2, 64, 48, 80, 48, 0, None),        # odd image size, dilated, ragged tail, sliced dx
    (2, 96, 176, 64, 144, 3, 1, 64, 144, 64, 144, 0, (64, 64, 1)),   # forced unsplit plan: 81 K steps, odd
    (2, 64, 64, 48, 144, 3, 1, 48, 144, 48, 144, 0, (64, 48, 4)),    # forced 4-way split x 21 K steps is refused by the gate -> see test below
]


def _run_case(hip_lib, case, expect=None):
    from gaia_seg_amd.hip import lib
    from gaia_seg_amd.hip.runtime import current_stream_ptr
    n, h, w, ci, co, k, dil, ci_max, co_ld, ldx, ldy, acc, force = case
    torch.manual_seed(1234)
    w_log = torch.randn(co_ld, ci_max, k, k) * 0.1                 # logical OIHW, max size
    dy = torch.randn(n, h, w, co)
    x_ref = torch.zeros(n, ci, h, w, requires_grad=True)
    y = F.conv2d(x_ref, w_log[:co, :ci], None, 1, dil * (k // 2), dil)
    y.backward(dy.permute(0, 3, 1, 2))
    dx_ref = x_ref.grad.permute(0, 2, 3, 1)                        # NHWC

    w_phys = w_log.permute(2, 3, 1, 0).contiguous().to(DEV)        # [KH][KW][Ci_max][Co_ld]
    dy_buf = torch.full((n, h, w, ldy), float("nan"), device=DEV)  # an over-read past Co poisons dx
    dy_buf[..., :co] = dy.to(DEV)
    prior = torch.randn(n, h, w, ldx)
    dx_buf = prior.to(DEV).clone()
    d = _desc(lib, n, h, w, ci, co, k, dil, ci_max, co_ld, ldx, ldy)
    if force:
        assert hip_lib.gs_debug_force_plan(*force) == 0
    try:
        need = hip_lib.gs_conv2d_workspace_bytes(ctypes.byref(d))
        ws = torch.empty(max(need, 16), dtype=torch.uint8, device=DEV)
        q = lib.DebugLaunch()
        assert hip_lib.gs_debug_query_conv_launch(ctypes.byref(d), lib.OP_DGRAD, ctypes.byref(q)) == 0
        lib.check(hip_lib.gs_conv2d_dgrad(ctypes.byref(d), dy_buf.data_ptr(), w_phys.data_ptr(),
                                          dx_buf.data_ptr(), acc, ws.data_ptr(), need,
                                          current_stream_ptr()), "dgrad")
    finally:
        if force:
            hip_lib.gs_debug_force_plan(0, 0, 0)
    torch.cuda.synchronize()
    rec = _launch_record(lib, hip_lib)
    assert rec.op == lib.OP_DGRAD
    # the host-only query describes the launch that really happened
    assert (q.kloop, q.bm, q.bn, q.splits, q.ksteps_per_split) == \
        (rec.kloop, rec.bm, rec.bn, rec.splits, rec.ksteps_per_split)
    (expect or _expect_kloop)(lib, rec)
    got = dx_buf.cpu()
    want = dx_ref + prior[..., :ci] if acc else dx_ref
    err = float((got[..., :ci].double() - want.double()).abs().max() / dx_ref.double().abs().max())
    assert err < TOL, (case, err, rec.kloop)
    if ldx > ci:   # columns beyond the slice are not touched
        assert torch.equal(got[..., ci:], prior[..., ci:])
    return rec


@pytest.mark.parametrize("case", X3_CASES[:-1], ids=lambda c: "x".join(str(v) for v in c[:7]))
def test_dgrad_on_the_bf16x3_loop_matches_conv2d_backward(hip_lib, case):
    rec = _run_case(hip_lib, case)
    if case == X3_CASES[5]:
        assert rec.splits == 3 and rec.ksteps_per_split == 48     # the split-K form of the loop
    if case[3] == 48:
        assert rec.bn == 48


def test_short_split_ranges_keep_the_fp32_loop(hip_lib):
    """The gate's other side: a 4-way split with 21 K steps per split must NOT take the bf16x3 loop
    (its fill would not amortise), with GS_X3 on or off — and is exact to the same tolerance."""
    from gaia_seg_amd.hip import lib

    def expect(lib_, rec):
        assert rec.kloop in (lib_.KLOOP_FP32, lib_.KLOOP_FP32_PAIRS) and rec.splits == 4
    _run_case(hip_lib, X3_CASES[-1], expect)


def test_old_conv_cases_stay_on_the_fp32_loops(hip_lib):
    """The small shapes of tests/test_hip_ops_gpu.py::CONV_CASES launch fewer than 512 workgroups:
    they test the fp32 loops (and say so now)."""
    from gaia_seg_amd.hip import lib
    from test_hip_ops_gpu import CONV_CASES
    for (ci_max, co_max, ci, co, k, s, p, d, n, h, w) in CONV_CASES:
        if ci == 3 or s != 1:
            continue
        desc = lib.ConvDesc(N=n, H=h, W=w, Ci=ci, Co=co, Ci_max=ci_max, Co_ld=co_max, KH=k, KW=k,
                            stride=s, pad=p, dil=d, Ho=(h + 2 * p - d * (k - 1) - 1) // s + 1,
                            Wo=(w + 2 * p - d * (k - 1) - 1) // s + 1, x_sn=h * w * ci, x_sh=w * ci,
                            x_sw=ci, x_sc=1, ldy=co, ld_add=0, role=0, reserved=0, in_affine=None)
        q = lib.DebugLaunch()
        assert hip_lib.gs_debug_query_conv_launch(ctypes.byref(desc), lib.OP_DGRAD, ctypes.byref(q)) == 0
        assert q.kloop != lib.KLOOP_BF16X3


def test_bn_backward_epilogue_on_the_bf16x3_loop(hip_lib, monkeypatch, stream_all=False):
    """gs_bn_bwd_fuse in the epilogues of the large-grid data-gradient kernels: a stage of three
    bottlenecks (planes 64 at 2 x 128 x 136).  conv2's dgrad runs on the bf16x3 loop carrying mode 1
    (bn1); the 1x1 dgrads run on the streaming kernel carrying mode 1 (bn2) and mode 2 (bn3 +
    accumulate).  The fused and the unfused backward must agree, both must match the CPU oracle's
    OResLayer, and the launch counters must show exactly that dispatch."""
    import gaia_seg_amd.hip.ops as ops
    from gaia_seg_amd.core.bricks import DynamicBottleneck
    from gaia_seg_amd.hip import lib
    from gaia_seg_amd.models.utils import DynamicResLayer
    from oracle.model import OResLayer
    n, h, w, planes, inpl = 2, 128, 136, 64, 32
    torch.manual_seed(2)
    layer = DynamicResLayer(DynamicBottleneck, inpl, planes, depth=3, stride=1,
                            conv_cfg=dict(type="DynConv2d"), norm_cfg=dict(type="DynBN"))
    for m in layer.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            torch.nn.init.uniform_(m.weight, 0.5, 1.5)
            torch.nn.init.normal_(m.bias, 0, 0.2)
        elif getattr(m, "weight", None) is not None and m.weight.dim() == 4:
            torch.nn.init.normal_(m.weight, 0, (2.0 / (m.weight.shape[1] * m.weight.shape[2] ** 2)) ** 0.5)
    ref = OResLayer(inpl, planes, 3, stride=1)
    ref.load_state_dict(layer.state_dict())
    ref.train()
    x = torch.randn(n, inpl, h, w)
    gz = torch.randn(n, 4 * planes, h, w)
    xr = x.clone().requires_grad_(True)
    zr = ref(xr)
    zr.backward(gz)
    ref_grads = dict(ref.named_parameters())

    layer = layer.to(DEV).train()
    results = {}
    # (True: fused, bn3's ReLU mask as bytes -- mode 3; "act": fused, mask from the activation -- mode 2)
    for fuse in (True, "act", False):
        monkeypatch.setattr(ops, "BNBWD_FUSE", bool(fuse))
        monkeypatch.setattr(ops, "RELU_MASK_BYTES", fuse is True)
        ops.BNBWD_FUSED_MASK_COUNT = 0
        for p in layer.parameters():
            p.grad = None
        for m in layer.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.running_mean.zero_()
                m.running_var.fill_(1)
        xg = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        z = layer(xg)
        hip_lib.gs_debug_conv_launch_counts(None, 1)
        z.backward(gz.to(DEV).contiguous(memory_format=torch.channels_last))
        torch.cuda.synchronize()
        counts = (ctypes.c_int64 * (3 * lib.KLOOP_COUNT * 3))()
        hip_lib.gs_debug_conv_launch_counts(counts, 1)
        c = lambda kloop, mode: counts[(lib.OP_DGRAD * lib.KLOOP_COUNT + kloop) * 3 + mode]
        x3 = [c(lib.KLOOP_BF16X3, m) for m in range(3)]
        f32 = [c(lib.KLOOP_FP32, m) + c(lib.KLOOP_FP32_PAIRS, m) for m in range(3)]
        stm = [c(lib.KLOOP_STREAM, m) for m in range(3)]
        # Production dispatch: conv2 (3x3, owns bn1) and conv3 (1x1, K = 256, owns bn2) of the three
        # blocks run on the bf16x3 loop carrying mode 1; conv1 of blocks 1, 2 (K = 64 -> 256 columns,
        # owns the previous block's bn3, accumulating) runs on the streaming kernel carrying mode 2;
        # block 0's conv1 and the shortcut conv (32 columns) stay on the fp32 tile loop, nothing fused.
        # With every eligible shape streamed (tests/test_stream_1x1_gpu.py) all seven 1x1s stream.
        big, other = (x3, f32) if X3_ON else (f32, x3)
        narrow = [0, 0, 0] if stream_all else [2, 0, 0]
        if X3_ON:
            assert f32 == narrow, list(counts)
        else:
            assert other == [0, 0, 0], list(counts)
            big = [b - n for b, n in zip(big, narrow)]
        n_big = 3 if stream_all else 6
        assert big == ([0, n_big, 0] if fuse else [n_big, 0, 0]), list(counts)
        if stream_all:
            assert stm == ([2, 3, 2] if fuse else [7, 0, 0]), list(counts)
        else:
            assert stm == ([0, 0, 2] if fuse else [2, 0, 0]), list(counts)
        assert ops.BNBWD_FUSED_MASK_COUNT == (2 if fuse is True else 0)
        results[fuse] = (z.detach().clone(), xg.grad.clone(),
                         {k: p.grad.clone() for k, p in layer.named_parameters()})
    assert torch.equal(results[True][0], results["act"][0])
    assert torch.equal(results[True][1], results["act"][1])      # mask bytes == mask from the activation
    for k in results[True][2]:
        assert torch.equal(results[True][2][k], results["act"][2][k]), k
    assert torch.equal(results[True][0], results[False][0])
    assert rel_err(results[True][1], results[False][1]) < 2e-5
    for k in results[True][2]:
        assert rel_err(results[True][2][k], results[False][2][k]) < 2e-5, k
    # against PyTorch on the CPU (nine BatchNorms deep: summation-order noise, not 3e-5)
    assert rel_err(results[True][0], zr) < 1e-4
    # (dx per pixel: a ReLU whose pre-activation is a rounding error away from zero may fall on the
    # other side than on the CPU and moves that pixel's gradient by O(1) -- tests/test_grad_criterion.py;
    # so the input gradient is held in the L2 norm and by the share of pixels off by more than 1e-3)
    dxe = (results[True][1].double().cpu() - xr.grad.double())
    assert float(dxe.norm() / xr.grad.double().norm()) < 3e-3
    assert float((dxe.abs() > 1e-3 * float(xr.grad.abs().max())).double().mean()) < 5e-3
    # (parameter gradients sum those pixels: the same handful of rounding-level ReLU flips shows as a
    # few 1e-3 of the largest entry, so they are held in the L2 norm; the sharp statements are fused == unfused above and the direct
    # dgrad cases at 3e-5)
    for k, g in results[True][2].items():
        want = ref_grads[k].grad.double()
        assert float((g.double().cpu() - want).norm() / want.norm()) < 3e-3, k
        assert rel_err(g, want) < 2e-2, k


def test_same_cases_on_the_fp32_loop(hip_lib):
    """GS_X3=0 must swap the kernel under exactly these cases (asserted by _expect_kloop in the child)
    and keep the same tolerance."""
    if not X3_ON:
        pytest.skip("already the GS_X3=0 child")
    env = dict(os.environ, GS_X3="0")
    here = os.path.dirname(os.path.abspath(__file__))
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu",
                          "-k", "matches_conv2d_backward or bn_backward_epilogue or short_split"],
                         env=env, capture_output=True, text=True, timeout=900, cwd=os.path.dirname(here))
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    assert " passed" in res.stdout
