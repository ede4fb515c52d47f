"""Split-K combined inside the launch (csrc/igemm_core.h: splitk_publish / slabs_to_lds).

The forward and data-gradient row kernels of the small-spatial layers (stages 3-4 of the supernet
backbone, gaiaseg/models/utils/dynamic_res_layer.py:84-125, and the heads' bottleneck convs) split
their contraction over several workgroups per output tile.  Each workgroup publishes its partial tile
to a slab with write-through stores; the tile's last-arriving workgroup sums the slabs in split order
and runs the ordinary epilogue.  gs_debug_set_splitk_inkernel(0) puts the separate reduce launch back;
both sum in the same fixed order (the reduce launch: as long as it walks the slabs sequentially, i.e.
below 16 splits; its wide form for many splits sums 16 interleaved groups first), so y / dx must agree
BIT FOR BIT, on any placement of the workgroups and under any arrival order — that is what these
tests check, through the C-ABI, together with the fp32 tolerance of the other operator tests against
PyTorch's conv2d on the CPU:

  * forward: plain, bias, addend, output slice of a wider buffer, two LDS column chunks, ragged tiles;
  * data gradient: plain, accumulate, sliced dx, stride 2 (one launch per parity class);
  * the fused conv + BatchNorm calls, whose last arriver also produces the BatchNorm tile statistics
    (forward) and the BatchNorm-backward partial sums (data gradient);
  * a back-to-back sequence of launches with changing inputs and shapes, alternating with
    separate-reduce launches that leave the slab's lines cached wherever their reduce ran: a stale
    slab line (L1 or a remote L2) anywhere would show as a bit difference.
Every case asserts through gs_debug_splitk_combined that it really took the in-launch path.

The same hand-off one level up (column_arrive / column_finalize_*): a fused conv + BatchNorm launch with
few row tiles merges its per-tile partials itself — the last workgroup of every column tile writes the
BatchNorm coefficients (forward) or the BatchNorm-backward sums (data gradient) — instead of leaving
them to a bn_tile_finalize / sum_partials launch.  gs_debug_set_col_finalize(0) puts those launches
back; test_tile_partials_merged_in_the_launch compares the two (and float64 on the CPU) on split and
unsplit shapes, ragged last tiles, 80-wide column tiles, and back to back with changing inputs."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 3e-5


@pytest.fixture()
def col_mode(hip_lib):
    def set_mode(mode):
        assert hip_lib.gs_debug_set_col_finalize(mode) == 0
    yield set_mode
    hip_lib.gs_debug_set_col_finalize(-1)


@pytest.fixture()
def splitk_mode(hip_lib):
    """set(mode) switches the process-global hook; always restored to the default afterwards."""
    def set_mode(mode):
        assert hip_lib.gs_debug_set_splitk_inkernel(mode) == 0
    yield set_mode
    hip_lib.gs_debug_set_splitk_inkernel(-1)


def _desc(lib, n, h, w, ci, co, k, stride=1, dil=1, ci_max=None, co_ld=None, ldx=None, ldy=None,
          ld_add=0, role=0):
    p = dil * (k // 2)
    ho = (h + 2 * p - dil * (k - 1) - 1) // stride + 1
    wo = (w + 2 * p - dil * (k - 1) - 1) // stride + 1
    ldx = ldx or ci
    return lib.ConvDesc(N=n, H=h, W=w, Ci=ci, Co=co, Ci_max=ci_max or ci, Co_ld=co_ld or co, KH=k, KW=k,
                        stride=stride, pad=p, dil=dil, Ho=ho, Wo=wo, x_sn=h * w * ldx, x_sh=w * ldx,
                        x_sw=ldx, x_sc=1, ldy=ldy or co, ld_add=ld_add, role=role, reserved=0,
                        in_affine=None), ho, wo


def _same(a, b, splits):
    """bit equality where the reduce launch sums in split order too (its sequential form)"""
    if splits < 16:
        assert torch.equal(a, b)
    else:
        assert rel_err(a, b) < 2e-6


def _planned_splits(hip_lib, lib, d, op):
    q = lib.DebugLaunch()
    assert hip_lib.gs_debug_query_conv_launch(ctypes.byref(d), op, ctypes.byref(q)) == 0
    return q.splits


# n  h   w   ci   co   k  bias addend ldy_extra role   what it covers
FWD_CASES = [
    (2, 24, 32, 64, 64, 3, 0, 0, 0, 1),        # K3-shaped (role 1): 24 tiles x 3 splits on the bf16x3 loop
    (1, 9, 11, 128, 48, 3, 0, 0, 0, 0),        # 99 rows: two ragged tiles, many splits
    (2, 33, 33, 512, 128, 3, 0, 0, 0, 1),      # stage-4 conv2 at a 513 crop: 2178 rows (ragged), K = 4608
    (2, 17, 17, 1024, 512, 1, 0, 0, 0, 0),     # stage-4 conv1: 1x1, 8 column tiles
    (2, 20, 20, 256, 80, 3, 1, 0, 0, 0),       # 80-wide tiles: two LDS column chunks; bias
    (2, 24, 24, 128, 96, 3, 0, 1, 32, 0),      # addend (FPN lateral + top-down), y is a slice of a wider buffer
]


def _fwd_once(hip_lib, lib, case, x, w_phys, bias, addend, y_buf):
    from gaia_seg_amd.hip.runtime import current_stream_ptr
    n, h, w, ci, co, k, has_b, has_a, ldy_extra, role = case
    d, ho, wo = _desc(lib, n, h, w, ci, co, k, ldy=co + ldy_extra, ld_add=co if has_a else 0, role=role)
    need = hip_lib.gs_conv2d_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=DEV)
    lib.check(hip_lib.gs_conv2d_forward(ctypes.byref(d), x.data_ptr(), w_phys.data_ptr(),
                                        bias.data_ptr() if has_b else None,
                                        addend.data_ptr() if has_a else None, y_buf.data_ptr(),
                                        ws.data_ptr(), need, current_stream_ptr()), "forward")
    return d


@pytest.mark.parametrize("case", FWD_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_forward_combined_in_the_launch_is_bit_identical_to_the_reduce_launch(hip_lib, splitk_mode, case):
    from gaia_seg_amd.hip import lib
    n, h, w, ci, co, k, has_b, has_a, ldy_extra, role = case
    torch.manual_seed(99)
    x = torch.randn(n, h, w, ci)
    w_log = torch.randn(co, ci, k, k) * 0.1
    bias = torch.randn(co)
    addend = torch.randn(n, h, w, co)
    y_ref = F.conv2d(x.permute(0, 3, 1, 2), w_log, bias if has_b else None, 1, k // 2).permute(0, 2, 3, 1)
    if has_a:
        y_ref = y_ref + addend
    xg, wg = x.to(DEV), w_log.permute(2, 3, 1, 0).contiguous().to(DEV)
    bg, ag = bias.to(DEV), addend.to(DEV)
    outs = []
    for mode in (1, 0):
        splitk_mode(mode)
        hip_lib.gs_debug_splitk_combined(1)
        y_buf = torch.full((n, h, w, co + ldy_extra), 7.0, device=DEV)
        d = _fwd_once(hip_lib, lib, case, xg, wg, bg, ag, y_buf)
        torch.cuda.synchronize()
        splits = _planned_splits(hip_lib, lib, d, lib.OP_FORWARD)
        assert splits > 1, "case does not split K"
        assert hip_lib.gs_debug_splitk_combined(1) == (1 if mode == 1 else 0)
        outs.append(y_buf.cpu())
    _same(outs[0], outs[1], splits)
    assert rel_err(outs[0][..., :co], y_ref) < TOL
    if ldy_extra:
        assert bool((outs[0][..., co:] == 7.0).all())


@pytest.mark.parametrize("bn", [48, 64, 80])
@pytest.mark.parametrize("splits", [2, 3, 4, 5, 7, 8, 9, 14])
def test_forced_split_counts_and_column_widths(hip_lib, splitk_mode, bn, splits):
    """Every split count from 2 to 14 that sits on an edge of the combine's load batches (four slabs
    in flight per output quad on 48 / 64-wide tiles, two on 80-wide ones: 2, 3 | 4, 5 | 7, 8, 9 | 14), on
    each column width of the planner, with a ragged last row tile (33 x 33 pixels) and a ragged last
    column tile: bit-identical to the reduce launch, bit-identical between two runs."""
    from gaia_seg_amd.hip import lib
    n, h, w, k = 2, 33, 33, 3
    ci, co = 192, {48: 112, 64: 144, 80: 176}[bn]     # co: two full column tiles + a ragged third
    case = (n, h, w, ci, co, k, 0, 0, 0, 0)
    torch.manual_seed(splits * 100 + bn)
    x = torch.randn(n, h, w, ci)
    w_log = torch.randn(co, ci, k, k) * 0.1
    y_ref = F.conv2d(x.permute(0, 3, 1, 2), w_log, None, 1, 1).permute(0, 2, 3, 1)
    xg, wg = x.to(DEV), w_log.permute(2, 3, 1, 0).contiguous().to(DEV)
    assert hip_lib.gs_debug_force_plan(64, bn, splits) == 0
    try:
        outs = []
        for mode in (1, 1, 0):
            splitk_mode(mode)
            hip_lib.gs_debug_splitk_combined(1)
            y = torch.full((n, h, w, co), 3.0, device=DEV)
            d = _fwd_once(hip_lib, lib, case, xg, wg, None, None, y)
            torch.cuda.synchronize()
            rec = lib.DebugLaunch()
            assert hip_lib.gs_debug_last_conv_launch(ctypes.byref(rec)) == 0
            assert (rec.bm, rec.bn) == (64, bn) and rec.splits > 1
            assert hip_lib.gs_debug_splitk_combined(1) == (1 if mode == 1 else 0)
            outs.append(y.cpu())
    finally:
        hip_lib.gs_debug_force_plan(0, 0, 0)
    assert torch.equal(outs[0], outs[1])          # two runs of the in-launch combine
    _same(outs[0], outs[2], rec.splits)           # against the reduce launch
    assert rel_err(outs[0], y_ref) < TOL


# n  h   w   ci   co   k stride acc ldx_extra
DGRAD_CASES = [
    (2, 24, 32, 64, 64, 3, 1, 0, 0),
    (2, 33, 33, 128, 512, 3, 1, 1, 0),         # stage-4 conv2's data gradient, accumulating
    (2, 17, 17, 512, 2048, 1, 1, 1, 0),        # conv3's: accumulate onto the identity gradient
    (1, 9, 11, 48, 128, 3, 1, 0, 16),          # ragged, dx a slice of a wider buffer
    (2, 32, 32, 128, 256, 3, 2, 0, 0),         # stride 2: four parity classes, each its own split launch
    (2, 34, 34, 256, 512, 1, 2, 1, 0),         # 1x1 stride-2 shortcut, accumulate
]


@pytest.mark.parametrize("case", DGRAD_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_dgrad_combined_in_the_launch_is_bit_identical_to_the_reduce_launch(hip_lib, splitk_mode, case):
    from gaia_seg_amd.hip import lib
    from gaia_seg_amd.hip.runtime import current_stream_ptr
    n, h, w, ci, co, k, stride, acc, ldx_extra = case
    torch.manual_seed(5)
    w_log = torch.randn(co, ci, k, k) * 0.1
    d, ho, wo = _desc(lib, n, h, w, ci, co, k, stride=stride, ldx=ci + ldx_extra)
    dy = torch.randn(n, ho, wo, co)
    x_ref = torch.zeros(n, ci, h, w, requires_grad=True)
    F.conv2d(x_ref, w_log, None, stride, k // 2).backward(dy.permute(0, 3, 1, 2))
    dx_ref = x_ref.grad.permute(0, 2, 3, 1)
    prior = torch.randn(n, h, w, ci + ldx_extra)
    wg, dyg = w_log.permute(2, 3, 1, 0).contiguous().to(DEV), dy.to(DEV)
    need = hip_lib.gs_conv2d_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=DEV)
    outs = []
    for mode in (1, 0):
        splitk_mode(mode)
        hip_lib.gs_debug_splitk_combined(1)
        dx = prior.to(DEV).clone()
        lib.check(hip_lib.gs_conv2d_dgrad(ctypes.byref(d), dyg.data_ptr(), wg.data_ptr(), dx.data_ptr(),
                                          acc, ws.data_ptr(), need, current_stream_ptr()), "dgrad")
        torch.cuda.synchronize()
        rec = lib.DebugLaunch()
        assert hip_lib.gs_debug_last_conv_launch(ctypes.byref(rec)) == 0
        assert rec.op == lib.OP_DGRAD and rec.splits > 1, "case does not split K"
        combined = hip_lib.gs_debug_splitk_combined(1)
        assert combined >= 1 if mode == 1 else combined == 0
        outs.append(dx.cpu())
    _same(outs[0], outs[1], rec.splits)
    want = dx_ref + prior[..., :ci] if acc else dx_ref
    assert rel_err(outs[0][..., :ci] - (prior[..., :ci] if acc else 0), dx_ref) < TOL
    assert rel_err(outs[0][..., :ci], want) < TOL
    if ldx_extra:
        assert torch.equal(outs[0][..., ci:], prior[..., ci:])


def test_back_to_back_launches_never_read_a_stale_slab(hip_lib, splitk_mode):
    """60 launches in a row on one stream, no host synchronisation in between, inputs and shapes
    changing from launch to launch so that every slab line is rewritten by a different tile every time.
    Between them, separate-reduce launches (plain loads of the same slab addresses) leave those lines
    cached on whichever CUs ran the reduce.  Each in-launch result must equal, bit for bit, what the
    separate-reduce path gives for the same input."""
    from gaia_seg_amd.hip import lib
    shapes = [(2, 24, 32, 64, 64, 3, 0, 0, 0, 1), (2, 33, 33, 512, 128, 3, 0, 0, 0, 1),
              (2, 17, 17, 1024, 512, 1, 0, 0, 0, 0), (1, 9, 11, 128, 48, 3, 0, 0, 0, 0)]
    torch.manual_seed(3)
    data = []
    for case in shapes:
        n, h, w, ci, co, k = case[:6]
        data.append((torch.randn(n, h, w, ci, device=DEV),
                     (torch.randn(co, ci, k, k) * 0.1).permute(2, 3, 1, 0).contiguous().to(DEV)))
    got, inputs = [], []
    hip_lib.gs_debug_splitk_combined(1)
    for i in range(60):
        case = shapes[i % len(shapes)]
        n, h, w, ci, co, k = case[:6]
        x0, wg = data[i % len(shapes)]
        x = x0 * (1.0 + 0.03125 * i) + 0.25 * (i % 5)
        y = torch.empty(n, h, w, co, device=DEV)
        splitk_mode(0)      # a reduce launch over this slab: its lines are now cached where it ran
        _fwd_once(hip_lib, lib, case, x0, wg, None, None, y)
        splitk_mode(1)
        _fwd_once(hip_lib, lib, case, x, wg, None, None, y)
        got.append(y)
        inputs.append(x)
    torch.cuda.synchronize()
    assert hip_lib.gs_debug_splitk_combined(1) == 60
    splitk_mode(0)
    for i in range(60):
        case = shapes[i % len(shapes)]
        n, h, w, ci, co, k = case[:6]
        y = torch.empty(n, h, w, co, device=DEV)
        d = _fwd_once(hip_lib, lib, case, inputs[i], data[i % len(shapes)][1], None, None, y)
        torch.cuda.synchronize()
        _same(y, got[i], _planned_splits(hip_lib, lib, d, lib.OP_FORWARD))
    assert hip_lib.gs_debug_splitk_combined(1) == 0


CONV_BN_SPLIT_CASES = [
    # ci  co   k  n  h   w   relu residual
    (64, 64, 3, 2, 24, 32, True, False),
    (256, 256, 3, 2, 33, 33, True, False),     # stage-3 conv2 at a 513 crop: ragged rows
    (1024, 256, 1, 2, 17, 17, True, False),
    (128, 80, 3, 2, 20, 20, True, True),       # two column chunks, residual
]


@pytest.mark.parametrize("case", CONV_BN_SPLIT_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_fused_conv_bn_statistics_and_bn_backward_sums_from_the_last_arriver(hip_lib, splitk_mode, case):
    """gs_conv_bn_forward / gs_conv_bn_backward on split-K shapes: with the in-launch combine the
    BatchNorm tile statistics (forward) and the BatchNorm-backward partial sums (data gradient of the
    consumer, fused epilogue) come from each tile's last-arriving workgroup.  Against PyTorch on the
    CPU, and against the separate-reduce path of the same build."""
    from gaia_seg_amd.core.bricks import DynamicBatchNorm2d, DynamicConv2d, conv_bn_act
    from gaia_seg_amd.hip.runtime import tape_function
    ci, co, k, n, h, w, relu, use_res = case
    torch.manual_seed(7)
    conv_a = DynamicConv2d(ci, ci, 1, bias=False)            # producer: its BN backward is fused into
    bn_a = DynamicBatchNorm2d(ci)                            # the data gradient of `conv`
    conv = DynamicConv2d(ci, co, k, padding=k // 2, bias=False)
    bn = DynamicBatchNorm2d(co)
    for c in (conv_a, conv):
        torch.nn.init.normal_(c.weight, 0, 1.0 / (c.weight[0].numel() ** 0.5))
    for b in (bn_a, bn):
        torch.nn.init.uniform_(b.weight, 0.5, 1.5)
        torch.nn.init.normal_(b.bias, 0, 0.3)
    x = torch.randn(n, ci, h, w) + 0.5
    res = torch.randn(n, co, h, w) if use_res else None

    # reference in float64 on the CPU (an fp32 reference flips ReLU masks of its own: at the 256-channel
    # case its parameter gradients are 4e-3 off the float64 ones, the HIP path 6e-7)
    params = [conv_a.weight, bn_a.weight, bn_a.bias, conv.weight, bn.weight, bn.bias]
    refs = [p.detach().clone().double().requires_grad_(True) for p in params]
    x_ref = x.clone().double().requires_grad_(True)
    a_ref = F.relu(F.batch_norm(F.conv2d(x_ref, refs[0]), None, None, refs[1], refs[2], True, 0.1, 1e-5))
    z_ref = F.batch_norm(F.conv2d(a_ref, refs[3], None, 1, k // 2), None, None, refs[4], refs[5], True,
                         0.1, 1e-5)
    if use_res:
        z_ref = z_ref + res.double()
    if relu:
        z_ref = F.relu(z_ref)
    gz = torch.randn(z_ref.shape)
    z_ref.backward(gz.double())

    mods = [m.to(DEV) for m in (conv_a, bn_a, conv, bn)]
    mods[1].train(), mods[3].train()
    results = []
    for mode in (1, 0):
        splitk_mode(mode)
        hip_lib.gs_debug_splitk_combined(1)
        for p in params:
            p.grad = None
        xg = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        inputs = [xg]
        if use_res:
            inputs.append(res.to(DEV).contiguous(memory_format=torch.channels_last))

        def run(tape, acts):
            a = conv_bn_act(tape, mods[0], mods[1], acts[0], relu=True)
            return [conv_bn_act(tape, mods[2], mods[3], a, relu=relu,
                                residual=acts[1] if use_res else None)]
        z = tape_function(run, inputs, True)[0]
        z.backward(gz.to(DEV))
        torch.cuda.synchronize()
        combined = hip_lib.gs_debug_splitk_combined(1)
        assert combined >= 2 if mode == 1 else combined == 0, combined
        results.append([z.detach().cpu(), xg.grad.cpu()] + [p.grad.detach().cpu() for p in params])
    # (gradients: mean error — a ReLU mask that flips on a pre-activation within rounding of zero moves
    # single elements by a whole gradient value, on either path and against the CPU alike)
    def mean_err(a, b):
        return float((a.double() - b.double()).abs().mean() / b.double().abs().mean())
    for got in results:
        assert rel_err(got[0], z_ref) < 1e-4
        assert mean_err(got[1], x_ref.grad) < 1e-5
        for g, r in zip(got[2:], refs):
            assert mean_err(g, r.grad) < 1e-5
    # the two paths differ only in where the BatchNorm sums are rounded
    assert rel_err(results[0][0], results[1][0]) < 2e-5
    for a, b in zip(results[0][1:], results[1][1:]):
        assert mean_err(a, b) < 2e-5


COL_CASES = [
    # ci   co   k  n  h   w   residual   (row tiles, split?)
    (64, 64, 3, 2, 24, 32, False),        # 24 row tiles, split-K: slabs combined, then partials merged
    (256, 256, 3, 2, 33, 33, False),      # 35 row tiles, ragged last tile (2178 rows)
    (1024, 256, 1, 2, 32, 64, False),     # stage-3 conv1 at 512x1024: 64 row tiles, unsplit, 4 column tiles
    (256, 1024, 1, 2, 32, 64, False),     # stage-3 conv3: 16 column tiles
    # (no residual cases: with 4M outputs, relu(bn(y) + r) flips a mask bit against float64 about once per
    # draw, on either path, and that moves the producer's gradients by 1e-4; the apply pass is not under
    # test here — tests/test_hip_ops_gpu.py covers it)
    (320, 320, 3, 2, 32, 64, False),      # MAX stage 3: 80-wide column tiles (two LDS chunks, 20 quads)
    (128, 128, 3, 2, 64, 79, False),      # 158 row tiles: the largest the default limit (160) admits
]


@pytest.mark.parametrize("case", COL_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_tile_partials_merged_in_the_launch(hip_lib, col_mode, case):
    from gaia_seg_amd.core.bricks import DynamicBatchNorm2d, DynamicConv2d, conv_bn_act
    from gaia_seg_amd.hip.runtime import tape_function
    ci, co, k, n, h, w, use_res = case
    torch.manual_seed(17)
    conv_a = DynamicConv2d(ci, ci, 1, bias=False)
    bn_a = DynamicBatchNorm2d(ci)
    conv = DynamicConv2d(ci, co, k, padding=k // 2, bias=False)
    bn = DynamicBatchNorm2d(co)
    for c in (conv_a, conv):
        torch.nn.init.normal_(c.weight, 0, 1.0 / (c.weight[0].numel() ** 0.5))
    for b in (bn_a, bn):
        torch.nn.init.uniform_(b.weight, 0.5, 1.5)
        torch.nn.init.normal_(b.bias, 0, 0.3)
    x = torch.randn(n, ci, h, w) + 0.5
    res = torch.randn(n, co, h, w) if use_res else None
    params = [conv_a.weight, bn_a.weight, bn_a.bias, conv.weight, bn.weight, bn.bias]
    refs = [p.detach().clone().double().requires_grad_(True) for p in params]
    x_ref = x.clone().double().requires_grad_(True)
    rm_a, rv_a = torch.zeros(ci, dtype=torch.float64), torch.ones(ci, dtype=torch.float64)
    rm_b, rv_b = torch.zeros(co, dtype=torch.float64), torch.ones(co, dtype=torch.float64)
    a_ref = F.relu(F.batch_norm(F.conv2d(x_ref, refs[0]), rm_a, rv_a, refs[1], refs[2], True, 0.1, 1e-5))
    z_ref = F.batch_norm(F.conv2d(a_ref, refs[3], None, 1, k // 2), rm_b, rv_b, refs[4], refs[5], True,
                         0.1, 1e-5)
    if use_res:
        z_ref = z_ref + res.double()
    z_ref = F.relu(z_ref)
    gz = torch.randn(z_ref.shape)
    z_ref.backward(gz.double())

    mods = [m.to(DEV) for m in (conv_a, bn_a, conv, bn)]
    mods[1].train(), mods[3].train()

    def mean_err(a, b):
        return float((a.double() - b.double()).abs().mean() / b.double().abs().mean())
    results = []
    for mode in (3, 0, 3):           # (the third run: counters back at zero after the first)
        col_mode(mode)
        hip_lib.gs_debug_col_finalized(1)
        for p in params:
            p.grad = None
        for b in (mods[1], mods[3]):
            b.running_mean.zero_()
            b.running_var.fill_(1.0)
        xg = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        inputs = [xg]
        if use_res:
            inputs.append(res.to(DEV).contiguous(memory_format=torch.channels_last))

        def run(tape, acts):
            a = conv_bn_act(tape, mods[0], mods[1], acts[0], relu=True)
            return [conv_bn_act(tape, mods[2], mods[3], a, relu=True,
                                residual=acts[1] if use_res else None)]
        z = tape_function(run, inputs, True)[0]
        z.backward(gz.to(DEV))
        torch.cuda.synchronize()
        merged = hip_lib.gs_debug_col_finalized(1)
        assert merged >= 2 if mode == 3 else merged == 0, merged   # `conv` forward + its data gradient
        results.append([z.detach().cpu(), xg.grad.cpu()] + [p.grad.detach().cpu() for p in params] +
                       [mods[3].running_mean.cpu().clone(), mods[3].running_var.cpu().clone()])
    for got in results:
        assert rel_err(got[0], z_ref) < 1e-4
        assert mean_err(got[1], x_ref.grad) < 1e-5
        for g, r in zip(got[2:8], refs):
            assert mean_err(g, r.grad) < 1e-5
        assert rel_err(got[8], rm_b) < 1e-5 and rel_err(got[9], rv_b) < 1e-5
    # merged in the launch vs the separate launches: the same sums, rounded in another order
    assert rel_err(results[0][0], results[1][0]) < 2e-5
    for a, b in zip(results[0][1:], results[1][1:]):
        assert mean_err(a, b) < 2e-5
    # and reproducible: the second in-launch run is bit-identical to the first
    for a, b in zip(results[0], results[2]):
        assert torch.equal(a, b)


def test_merged_partials_back_to_back_with_changing_inputs(hip_lib, col_mode):
    """40 fused conv + BatchNorm forwards in a row on one stream (no host synchronisation), inputs
    changing every time and two shapes alternating, so that the partials of every tile are rewritten by
    each launch; every output must match the separate-finalize path for the same input (a stale
    partial read by a column's last workgroup would move that column's normalisation)."""
    from gaia_seg_amd.core.bricks import DynamicBatchNorm2d, DynamicConv2d, conv_bn_act
    from gaia_seg_amd.hip.runtime import Act, Tape
    torch.manual_seed(23)
    layers = []
    for ci, co, k, h, w in [(256, 256, 3, 32, 64), (1024, 256, 1, 32, 64)]:
        conv = DynamicConv2d(ci, co, k, padding=k // 2, bias=False).to(DEV)
        bn = DynamicBatchNorm2d(co).to(DEV).train()
        layers.append((conv, bn, torch.randn(2, h, w, ci, device=DEV)))
    outs, inputs = [], []
    col_mode(1)
    hip_lib.gs_debug_col_finalized(1)
    for i in range(40):
        conv, bn, x0 = layers[i % 2]
        x = x0 * (1.0 + 0.0625 * i) + 0.125 * (i % 7)
        outs.append(conv_bn_act(Tape(enabled=False), conv, bn, Act(x, True), relu=True).t)
        inputs.append(x)
    torch.cuda.synchronize()
    assert hip_lib.gs_debug_col_finalized(1) == 40
    col_mode(0)
    for i in range(40):
        conv, bn, _ = layers[i % 2]
        want = conv_bn_act(Tape(enabled=False), conv, bn, Act(inputs[i], True), relu=True).t
        assert rel_err(outs[i], want) < 2e-6, i
    assert hip_lib.gs_debug_col_finalized(1) == 0
