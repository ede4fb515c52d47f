"""Operator-level parity of the FORWARD on the bf16x3 K loop (csrc/igemm_core.h x3_k_loop<BFWD>: the
[k][n] weights are staged one k row per thread and step, the two steps of a 32-channel slab
interleaved so that a column's (step 1, step 2) pair is one 32-bit LDS store per bf16 piece) --
DynConv2d forward of the bottleneck conv2 / conv1 / conv3 and the head convs
(gaiaseg/models/utils/dynamic_res_layer.py:84-125).

Direct gs_conv2d_forward calls on raw buffers, the dispatch asserted (GS_KLOOP_BF16X3 with the switch
on, the fp32 loops with it off), against F.conv2d on the CPU at 3e-5."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 3e-5

# n  h    w   ci   co  k dil ci_max co_ld ldx ldy  force_plan     what it covers
FWD_X3_CASES = [
    (2, 128, 256, 64, 64, 3, 1, 64, 64, 64, 64, None),           # K3 at stage 1: 36 K steps, 1024 tiles
    (2, 127, 131, 64, 64, 3, 1, 64, 64, 64, 64, None),           # ragged M tail
    (2, 64, 128, 128, 128, 3, 1, 128, 128, 128, 128, None),      # K3 at stage 2: two column tiles, 72 K steps
    (2, 128, 136, 48, 64, 3, 1, 48, 64, 48, 64, (64, 64, 1)),    # Ci = 48: odd K-step count 27, step pairs cross taps
    (2, 128, 136, 80, 128, 3, 2, 96, 160, 112, 176, (64, 64, 1)),  # dilation 2, leading weight slice, x / y slices (ld > C), Ci = 80
    (2, 32, 64, 256, 256, 3, 1, 256, 256, 256, 256, (64, 64, 3)),  # split-K 3 x 48 steps (slabs + reduce)
    (2, 64, 64, 256, 1024, 1, 1, 256, 1024, 256, 1024, None),    # 1x1 conv3 at stage 3: 16 column tiles
    (2, 128, 128, 64, 200, 1, 1, 64, 256, 64, 200, (64, 64, 1)),   # ragged last column tile (200 = 3 x 64 + 8)
    (2, 128, 256, 48, 48, 3, 1, 48, 48, 48, 48, None),           # MIN widths: 48-wide tiles (12 of 16 column quads staged)
    (2, 64, 128, 96, 96, 3, 1, 96, 96, 96, 96, (64, 48, 1)),     # two 48-wide column tiles
]


def _desc(lib, n, h, w, ci, co, k, dil, ci_max, co_ld, ldx, ldy):
    p = dil * (k // 2)
    return lib.ConvDesc(N=n, H=h, W=w, Ci=ci, Co=co, Ci_max=ci_max, Co_ld=co_ld, KH=k, KW=k, stride=1,
                        pad=p, dil=dil, Ho=h, Wo=w, x_sn=h * w * ldx, x_sh=w * ldx, x_sw=ldx, x_sc=1,
                        ldy=ldy, ld_add=0, role=0, reserved=0, in_affine=None)


@pytest.mark.parametrize("x3", [True, False], ids=["bf16x3", "fp32"])
@pytest.mark.parametrize("case", FWD_X3_CASES, ids=lambda c: "x".join(str(v) for v in c[:7]))
def test_forward_on_the_bf16x3_loop_matches_conv2d(hip_lib, case, x3):
    from gaia_seg_amd.hip import lib
    from gaia_seg_amd.hip.runtime import current_stream_ptr
    n, h, w, ci, co, k, dil, ci_max, co_ld, ldx, ldy, force = case
    torch.manual_seed(4321)
    w_log = torch.randn(co_ld, ci_max, k, k) * 0.1
    x = torch.randn(n, h, w, ci)
    y_ref = F.conv2d(x.permute(0, 3, 1, 2), w_log[:co, :ci], None, 1, dil * (k // 2), dil).permute(0, 2, 3, 1)
    w_phys = w_log.permute(2, 3, 1, 0).contiguous().to(DEV)
    x_buf = torch.full((n, h, w, ldx), float("nan"), device=DEV)
    x_buf[..., :ci] = x.to(DEV)
    prior = torch.randn(n, h, w, ldy)
    y_buf = prior.to(DEV).clone()
    d = _desc(lib, n, h, w, ci, co, k, dil, ci_max, co_ld, ldx, ldy)
    hip_lib.gs_debug_set_x3_fwd(2 if x3 else 0)
    hip_lib.gs_debug_set_stream_mode(0)        # (the 1x1 cases are about the tile kernel's loops)
    if force:
        assert hip_lib.gs_debug_force_plan(*force) == 0
    try:
        need = hip_lib.gs_conv2d_workspace_bytes(ctypes.byref(d))
        ws = torch.empty(max(need, 16), dtype=torch.uint8, device=DEV)
        lib.check(hip_lib.gs_conv2d_forward(ctypes.byref(d), x_buf.data_ptr(), w_phys.data_ptr(), None, None,
                                            y_buf.data_ptr(), ws.data_ptr(), need, current_stream_ptr()),
                  "forward")
    finally:
        hip_lib.gs_debug_force_plan(0, 0, 0)
        hip_lib.gs_debug_set_x3_fwd(-1)
        hip_lib.gs_debug_set_stream_mode(-1)
    torch.cuda.synchronize()
    rec = lib.DebugLaunch()
    assert hip_lib.gs_debug_last_conv_launch(ctypes.byref(rec)) == 0 and rec.op == lib.OP_FORWARD
    if x3:
        assert rec.kloop == lib.KLOOP_BF16X3, "case does not reach the bf16x3 loop (kloop %d)" % rec.kloop
    else:
        assert rec.kloop in (lib.KLOOP_FP32, lib.KLOOP_FP32_PAIRS)
    got = y_buf.cpu()
    assert rel_err(got[..., :co], y_ref) < TOL, (case, rec.kloop)
    if ldy > co:
        assert torch.equal(got[..., co:], prior[..., co:])
