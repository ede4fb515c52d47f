"""The stem kernels (csrc/stem.hip): 7x7 stride-2 pad-3 convolution of the 3-channel NCHW image
(gaiaseg/models/backbones/dynamic_resnet.py:290-297), forward and weight gradient, against F.conv2d on
the CPU at the fp32 tolerance of the other operator tests (3e-5), with the dispatch asserted through
gs_debug_last_conv_launch (128-row tiles = the stem kernels; 64 = the generic implicit GEMM); then the
fused conv + BatchNorm call, whose statistics are the stem epilogue's per-tile partials, against
F.batch_norm.  Widths 32 / 48 / 64 (the sampler's stem range), leading slice of the max-size weight,
odd H (the last output row's patch hangs over the image), several tiles per output row."""
import ctypes
import os

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 3e-5
WGRAD_ON = os.environ.get("GS_STEM_WGRAD", "1") != "0"   # GS_STEM_WGRAD=0: generic weight-gradient kernel

# co_max co  n  h    w     stem kernels?
CASES = [
    (64, 64, 2, 64, 256, True),      # Wo = 128: one tile per output row
    (64, 48, 1, 37, 512, True),      # leading slice 48 of 64, odd H, two tiles per row
    (64, 32, 2, 18, 256, True),      # MIN width
    (48, 48, 1, 20, 768, True),      # three tiles per row, weight pitch 48
    (64, 64, 2, 32, 200, False),     # Wo = 100: not a multiple of 128 -> generic kernels
]


def _last(hip_lib, op):
    from gaia_seg_amd.hip import lib
    rec = lib.DebugLaunch()
    assert hip_lib.gs_debug_last_conv_launch(ctypes.byref(rec)) == 0
    return rec


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c[:5])))
def test_stem_forward_and_weight_gradient(hip_lib, case):
    from gaia_seg_amd.core.bricks import DynamicConv2d
    from gaia_seg_amd.hip import lib, ops
    co_max, co, n, h, w, stem = case
    torch.manual_seed(3)
    m = DynamicConv2d(3, co_max, 7, stride=2, padding=3, bias=False)
    torch.nn.init.normal_(m.weight, 0, 0.1)
    m.manipulate_width(co)
    x = torch.randn(n, 3, h, w)
    w_ref = m.weight.detach().clone().contiguous().requires_grad_(True)
    y_ref = F.conv2d(x, w_ref[:co], None, 2, 3)
    gy = torch.randn_like(y_ref)
    y_ref.backward(gy)
    m = m.to(DEV)
    keep = ops.SIDE_WGRAD
    ops.SIDE_WGRAD = False          # the weight gradient on this stream: its launch record is the last one
    from gaia_seg_amd.hip.runtime import Act, Tape
    try:   # (the tape is replayed on THIS thread: the launch record is thread-local)
        tape = Tape()
        ya = m.forward_act(tape, Act.from_nchw(x.to(DEV)))
        torch.cuda.synchronize()
        rec = _last(hip_lib, lib.OP_FORWARD)
        assert rec.op == lib.OP_FORWARD and (rec.bm == 128) == stem, (rec.bm, stem)
        assert rel_err(ya.as_nchw(), y_ref) < TOL
        ya.set_grad_from_nchw(gy.to(DEV).contiguous(memory_format=torch.channels_last))
        tape.backward()
        torch.cuda.synchronize()
        rec = _last(hip_lib, lib.OP_WGRAD)
        assert rec.op == lib.OP_WGRAD and (rec.bm == 128) == (stem and WGRAD_ON), (rec.bm, stem)
    finally:
        ops.SIDE_WGRAD = keep
    g = m.weight.grad.detach().cpu()
    assert rel_err(g[:co], w_ref.grad[:co]) < TOL
    if co < co_max:
        assert float(g[co:].abs().max()) == 0.0
    q = lib.DebugLaunch()
    d = ops._conv_desc(ops.Act.from_nchw(x.to(DEV)), m.weight, co, 2, 3, 1, co)
    for op in (lib.OP_FORWARD, lib.OP_WGRAD):
        assert hip_lib.gs_debug_query_conv_launch(ctypes.byref(d), op, ctypes.byref(q)) == 0
        assert (q.bm == 128) == (stem and (op == lib.OP_FORWARD or WGRAD_ON))


def test_same_cases_with_the_generic_weight_gradient():
    """The same cases with GS_STEM_WGRAD=0 (child interpreter: the switch is read once)."""
    import subprocess
    import sys
    if not WGRAD_ON:
        pytest.skip("already the GS_STEM_WGRAD=0 run")
    env = dict(os.environ, GS_STEM_WGRAD="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-x",
                        "-k", "forward_and_weight_gradient"], env=env, capture_output=True, text=True,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("co", [64, 48])
def test_stem_conv_bn_statistics_from_the_epilogue(hip_lib, co):
    """conv1 -> norm1 -> ReLU through the fused call (what DynamicResNet.forward_act issues): batch
    statistics merged from the stem kernel's tile partials, running statistics, gradients."""
    from gaia_seg_amd.core.bricks import DynamicBatchNorm2d, DynamicConv2d, conv_bn_act
    from gaia_seg_amd.hip.runtime import tape_function
    torch.manual_seed(9)
    n, h, w = 2, 48, 256
    conv = DynamicConv2d(3, 64, 7, stride=2, padding=3, bias=False)
    torch.nn.init.normal_(conv.weight, 0, 0.1)
    conv.manipulate_width(co)
    bn = DynamicBatchNorm2d(64)
    torch.nn.init.uniform_(bn.weight, 0.5, 1.5)
    torch.nn.init.normal_(bn.bias, 0, 0.2)
    x = torch.randn(n, 3, h, w) * 2 + 0.5
    w_ref = conv.weight.detach().clone().contiguous().requires_grad_(True)
    g_ref = bn.weight.detach().clone().requires_grad_(True)
    b_ref = bn.bias.detach().clone().requires_grad_(True)
    rm, rv = torch.zeros(co), torch.ones(co)
    y_ref = F.conv2d(x, w_ref[:co], None, 2, 3)
    z_ref = F.relu(F.batch_norm(y_ref, rm, rv, g_ref[:co], b_ref[:co], True, 0.1, 1e-5))
    gz = torch.randn_like(z_ref)
    z_ref.backward(gz)
    conv, bn = conv.to(DEV), bn.to(DEV).train()
    z = tape_function(lambda tape, acts: [conv_bn_act(tape, conv, bn, acts[0], relu=True)],
                      [x.to(DEV)], True)[0]
    assert rel_err(z, z_ref) < 1e-4
    z.backward(gz.to(DEV))
    torch.cuda.synchronize()
    assert rel_err(bn.running_mean[:co], rm) < 1e-4 and rel_err(bn.running_var[:co], rv) < 1e-4
    assert rel_err(conv.weight.grad[:co], w_ref.grad[:co]) < 2e-4
    assert rel_err(bn.weight.grad[:co], g_ref.grad[:co]) < 2e-4
    assert rel_err(bn.bias.grad[:co], b_ref.grad[:co]) < 2e-4
