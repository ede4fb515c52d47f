"""GPU parity of every HIP operator against plain PyTorch on the CPU (the oracle's primitives),
forward and backward, through the C-ABI.  fp32 MFMA is an exact fmaf chain, so tolerances only
cover summation-order differences.

The shapes of this file are small (fewer than 512 workgroups per launch), so every convolution here
runs on the fp32 MFMA K loops.  The bf16x3 data-gradient loop — the default for large grids — has its
own operator tests sized past its dispatch gate in tests/test_dgrad_x3_gpu.py, which assert the loop
that ran (gs_debug_last_conv_launch) and also re-run under GS_X3=0."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 3e-5


def _conv_case(ci_max, co_max, ci, co, k, s, p, d, n, h, w, bias=False, seed=0):
    from gaia_seg_amd.core.bricks import DynamicConv2d
    torch.manual_seed(seed)
    m = DynamicConv2d(ci_max, co_max, k, stride=s, padding=p, dilation=d, bias=bias)
    torch.nn.init.normal_(m.weight, 0, 0.1)
    if bias:
        torch.nn.init.normal_(m.bias, 0, 0.5)
    m.manipulate_width(co)
    x = torch.randn(n, ci, h, w)
    w_ref = m.weight.detach().clone().contiguous().requires_grad_(True)
    b_ref = m.bias.detach().clone().requires_grad_(True) if bias else None
    x_ref = x.clone().requires_grad_(ci != 3)
    y_ref = F.conv2d(x_ref, w_ref[:co, :ci], b_ref[:co] if bias else None, s, p, d)
    gy = torch.randn_like(y_ref)
    y_ref.backward(gy)

    m = m.to(DEV)
    xg = x.to(DEV)
    if ci != 3:
        xg = xg.contiguous(memory_format=torch.channels_last)
    xg.requires_grad_(ci != 3)
    y = m(xg)
    assert y.shape == y_ref.shape
    e_y = rel_err(y, y_ref)
    y.backward(gy.to(DEV))
    e_w = rel_err(m.weight.grad, w_ref.grad)
    out = {"y": e_y, "dw": e_w}
    # gradient outside the active slice must be exactly zero
    gfull = m.weight.grad.detach().cpu()
    assert float(gfull[co:].abs().max()) == 0.0 if co < co_max else True
    assert float(gfull[:, ci:].abs().max()) == 0.0 if ci < ci_max else True
    if ci != 3:
        out["dx"] = rel_err(xg.grad, x_ref.grad)
    if bias:
        out["db"] = rel_err(m.bias.grad, b_ref.grad)
    return out


CONV_CASES = [
    # ci_max co_max ci  co  k s p d  n  h   w
    (80, 80, 64, 48, 1, 1, 0, 1, 2, 16, 16),
    (80, 80, 48, 48, 3, 1, 1, 1, 2, 13, 17),
    (64, 64, 64, 64, 3, 2, 1, 1, 2, 16, 16),
    (32, 32, 32, 32, 3, 1, 2, 2, 1, 12, 12),
    (64, 320, 64, 256, 1, 2, 0, 1, 2, 15, 15),
    (3, 64, 3, 32, 7, 2, 3, 1, 2, 32, 40),
    # 3x3 stride 1, wide / ragged channel blocks, leading slices of wider weights
    (64, 64, 64, 64, 3, 1, 1, 1, 2, 24, 32),       # one 64-channel block, 2 column strips
    (80, 80, 80, 80, 3, 1, 1, 1, 2, 12, 16),       # ci blocks 48 + 32, co blocks 64 + 16
    (160, 96, 160, 48, 3, 1, 1, 1, 1, 9, 48),      # 3 ci blocks (64, 64, 32), ragged co
    (320, 320, 192, 320, 3, 1, 1, 1, 2, 6, 16),    # leading slice of a wider weight, 5 co blocks
    (48, 48, 48, 48, 3, 1, 1, 1, 2, 130, 64),      # many rows: several row ranges per strip
    (3, 32, 3, 24, 3, 2, 1, 1, 2, 17, 19),
    (24, 48, 24, 48, 3, 1, 1, 1, 1, 9, 9),
    (512, 128, 512, 128, 3, 1, 1, 1, 2, 8, 8),
    (160, 160, 128, 80, 3, 1, 1, 1, 1, 10, 10),
    (96, 96, 96, 96, 1, 1, 0, 1, 1, 7, 7),
    (320, 640, 192, 320, 1, 1, 0, 1, 1, 6, 6),
    (16, 64, 16, 64, 1, 1, 0, 1, 2, 128, 256),
    (128, 128, 128, 128, 3, 2, 1, 1, 2, 9, 11),
    (64, 64, 64, 64, 3, 1, 4, 4, 1, 20, 20),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(map(str, c)))
def test_dyn_conv2d_fwd_bwd(hip_lib, case):
    errs = _conv_case(*case)
    assert max(errs.values()) < TOL, errs


def test_conv_seg_19_classes_with_bias(hip_lib):
    errs = _conv_case(64, 19, 64, 19, 1, 1, 0, 1, 2, 9, 11, bias=True)
    assert max(errs.values()) < TOL, errs


def _bn_case(c_max, c, n, h, w, relu, residual, training=True, seed=0):
    from gaia_seg_amd.core.bricks import DynamicBatchNorm2d
    from gaia_seg_amd.hip import ops
    from gaia_seg_amd.hip.runtime import Act, Tape
    torch.manual_seed(seed)
    bn = DynamicBatchNorm2d(c_max)
    torch.nn.init.uniform_(bn.weight, 0.5, 1.5)
    torch.nn.init.normal_(bn.bias, 0, 0.3)
    bn.running_mean.normal_(0, 0.2)
    bn.running_var.uniform_(0.5, 1.5)
    bn.train(training)
    x = torch.randn(n, c, h, w) * 2 + 0.7
    r = torch.randn(n, c, h, w) if residual else None
    # --- reference ---
    ref = torch.nn.BatchNorm2d(c_max)
    ref.load_state_dict(bn.state_dict())
    ref.train(training)
    xr = x.clone().requires_grad_(True)
    rr = r.clone().requires_grad_(True) if residual else None
    wr, br = ref.weight, ref.bias
    yr = F.batch_norm(xr, ref.running_mean[:c], ref.running_var[:c], wr[:c], br[:c], training,
                      0.1, 1e-5)
    if residual:
        yr = yr + rr
    if relu:
        yr = torch.relu(yr)
    gy = torch.randn_like(yr)
    yr.backward(gy)
    # --- HIP ---
    bn = bn.to(DEV)
    tape = Tape()
    xa = Act.from_nchw(x.to(DEV).contiguous(memory_format=torch.channels_last), requires_grad=True)
    ra = Act.from_nchw(r.to(DEV).contiguous(memory_format=torch.channels_last), requires_grad=True) if residual else None
    ya = bn.forward_act(tape, xa, relu=relu, residual=ra)
    errs = {"y": rel_err(ya.as_nchw(), yr)}
    ya.set_grad_from_nchw(gy.to(DEV).contiguous(memory_format=torch.channels_last).clone())
    tape.backward()
    errs["dx"] = rel_err(xa.g.permute(0, 3, 1, 2), xr.grad)
    errs["dgamma"] = rel_err(bn.weight.grad[:c], wr.grad[:c])
    errs["dbeta"] = rel_err(bn.bias.grad[:c], br.grad[:c])
    if residual:
        errs["dres"] = rel_err(ra.g.permute(0, 3, 1, 2), rr.grad)
    if training:
        errs["rm"] = rel_err(bn.running_mean, ref.running_mean)
        errs["rv"] = rel_err(bn.running_var, ref.running_var)
    if c < c_max:
        assert float(bn.weight.grad[c:].abs().max()) == 0.0
    return errs


BN_CASES = [
    (80, 48, 2, 9, 11, True, False),
    (64, 64, 2, 16, 16, True, True),
    (80, 80, 1, 5, 7, False, False),
    (320, 320, 2, 6, 6, True, True),
    (640, 640, 2, 4, 4, True, False),
    (2560, 2048, 2, 3, 3, False, True),
    (512, 512, 2, 1, 1, True, False),   # PPM scale 1: two samples per channel
    (512, 512, 2, 6, 6, True, False),
    (64, 64, 2, 64, 128, True, False),
]


@pytest.mark.parametrize("case", BN_CASES, ids=lambda c: "x".join(map(str, c)))
def test_dyn_batchnorm_train(hip_lib, case):
    errs = _bn_case(*case)
    assert max(errs.values()) < 2e-4, errs


def test_dyn_batchnorm_eval_mode(hip_lib):
    errs = _bn_case(80, 64, 2, 7, 9, True, True, training=False)
    assert max(errs.values()) < 1e-4, errs


def test_bn_near_constant_two_samples(hip_lib):
    """s=1 PPM branch: variance from two nearly equal samples must not cancel catastrophically."""
    from gaia_seg_amd.core.bricks import DynamicBatchNorm2d
    bn = DynamicBatchNorm2d(8).to(DEV)
    x = torch.tensor([1.0, 1.001]).view(2, 1, 1, 1).repeat(1, 8, 1, 1)
    y = bn(x.to(DEV).contiguous(memory_format=torch.channels_last))
    ref = F.batch_norm(x, None, None, None, None, True, 0.1, 1e-5)
    assert rel_err(y, ref) < 1e-3


@pytest.mark.parametrize("rows,C,ld,res", [(1000, 64, 64, True), (333, 1028, 1032, True), (77, 8, 8, False)])
def test_bn_apply_writes_the_relu_mask_bytes(hip_lib, rows, C, ld, res):
    """gs_bn_apply_mask (gs_bn_args::relu_mask): the same z as gs_bn_apply, plus one byte per channel
    quad with bit e set <=> z[r][4q+e] > 0 -- the ReLU mask of relu(bn3(y) + identity)
    (gaiaseg/models/utils/dynamic_res_layer.py:113-123) that gs_bn_bwd_fuse mode 3 consumes."""
    torch.manual_seed(rows)
    x = torch.randn(rows, ld, device=DEV)
    r = torch.randn(rows, ld, device=DEV) if res else None
    coeffs = torch.cat([torch.rand(C) + 0.5, torch.randn(C) * 0.3, torch.randn(C) * 0.2, torch.ones(C)]).to(DEV)
    z0 = torch.full((rows, ld), 7.0, device=DEV)
    z1 = torch.full((rows, ld), 7.0, device=DEV)
    mask = torch.full((rows, C // 4), 255, dtype=torch.uint8, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    rp, rl = (r.data_ptr(), ld) if res else (None, 0)
    assert hip_lib.gs_bn_apply(x.data_ptr(), rows, C, ld, coeffs.data_ptr(), rp, rl, 1, z0.data_ptr(), ld, st) == 0
    assert hip_lib.gs_bn_apply_mask(x.data_ptr(), rows, C, ld, coeffs.data_ptr(), rp, rl, z1.data_ptr(), ld,
                                    mask.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert torch.equal(z0, z1)
    assert torch.equal(z1[:, C:], torch.full((rows, ld - C), 7.0, device=DEV))     # padding untouched
    bits = (z1[:, :C] > 0).view(rows, C // 4, 4).to(torch.uint8)
    want = bits[..., 0] | (bits[..., 1] << 1) | (bits[..., 2] << 2) | (bits[..., 3] << 3)
    assert torch.equal(mask, want)
    # a NULL mask is refused
    assert hip_lib.gs_bn_apply_mask(x.data_ptr(), rows, C, ld, coeffs.data_ptr(), rp, rl, z1.data_ptr(), ld,
                                    None, st) != 0


def _nhwc(t):
    return t.to(DEV).contiguous(memory_format=torch.channels_last)


def test_maxpool_fwd_bwd(hip_lib):
    from gaia_seg_amd.hip import ops
    from gaia_seg_amd.hip.runtime import Act, Tape
    torch.manual_seed(0)
    for (n, c, h, w) in [(2, 32, 17, 19), (1, 64, 32, 64)]:
        x = torch.relu(torch.randn(n, c, h, w))  # many exact ties at zero
        xr = x.clone().requires_grad_(True)
        yr = F.max_pool2d(xr, 3, 2, 1)
        gy = torch.randn_like(yr)
        yr.backward(gy)
        tape = Tape()
        xa = Act.from_nchw(_nhwc(x), requires_grad=True)
        ya = ops.maxpool(tape, xa, 3, 2, 1)
        assert torch.equal(ya.as_nchw().cpu(), yr.detach())
        ya.set_grad_from_nchw(_nhwc(gy))
        tape.backward()
        assert rel_err(xa.g.permute(0, 3, 1, 2), xr.grad) < 1e-6


def test_adaptive_avgpool_fwd_bwd(hip_lib):
    from gaia_seg_amd.hip import ops
    from gaia_seg_amd.hip.runtime import Act, Tape
    torch.manual_seed(0)
    for (n, c, h, w) in [(2, 64, 16, 32), (1, 2048, 7, 9), (2, 32, 64, 128), (2, 16, 2, 3), (1, 8, 4, 4)]:
        scales = [1, 2, 3, 6]
        x = torch.randn(n, c, h, w)
        xr = x.clone().requires_grad_(True)
        yrs = [F.adaptive_avg_pool2d(xr, s) for s in scales]
        gys = [torch.randn_like(y) for y in yrs]
        torch.autograd.backward(yrs, gys)
        tape = Tape()
        xa = Act.from_nchw(_nhwc(x), requires_grad=True)
        yas = ops.adaptive_avgpool(tape, xa, scales)
        for ya, yr, gy in zip(yas, yrs, gys):
            assert rel_err(ya.as_nchw(), yr) < 1e-5
            ya.set_grad_from_nchw(_nhwc(gy))
        tape.backward()
        assert rel_err(xa.g.permute(0, 3, 1, 2), xr.grad) < 1e-5


@pytest.mark.parametrize("align", [False, True])
def test_bilinear_fwd_bwd(hip_lib, align):
    from gaia_seg_amd.hip import ops
    from gaia_seg_amd.hip.runtime import Act, Tape
    torch.manual_seed(0)
    cases = [((2, 16, 6, 6), (16, 32)), ((1, 32, 1, 1), (16, 32)), ((2, 8, 25, 25), (49, 49)),
             ((1, 8, 49, 49), (97, 97)), ((1, 16, 2, 3), (64, 128)), ((1, 8, 16, 16), (8, 8))]
    for shape, size in cases:
        x = torch.randn(*shape)
        xr = x.clone().requires_grad_(True)
        yr = F.interpolate(xr, size=size, mode="bilinear", align_corners=align)
        gy = torch.randn_like(yr)
        yr.backward(gy)
        tape = Tape()
        xa = Act.from_nchw(_nhwc(x), requires_grad=True)
        ya = ops.bilinear(tape, xa, size, align)
        assert rel_err(ya.as_nchw(), yr) < 1e-5, (shape, size)
        ya.set_grad_from_nchw(_nhwc(gy))
        tape.backward()
        assert rel_err(xa.g.permute(0, 3, 1, 2), xr.grad) < 1e-5, (shape, size)


def test_bilinear_accumulate_and_slice(hip_lib):
    """resize-add (UPer top-down) and writing into a channel slice of a concat buffer."""
    from gaia_seg_amd.hip import ops
    from gaia_seg_amd.hip.runtime import Act, Tape
    torch.manual_seed(0)
    coarse, fine = torch.randn(2, 16, 5, 7), torch.randn(2, 16, 9, 13)
    ref = fine + F.interpolate(coarse, size=(9, 13), mode="bilinear", align_corners=False)
    tape = Tape(enabled=False)
    fa = Act.from_nchw(_nhwc(fine), requires_grad=False)
    ops.bilinear(tape, Act.from_nchw(_nhwc(coarse), False), (9, 13), False, out=fa, accumulate=True)
    assert rel_err(fa.as_nchw(), ref) < 1e-5
    cat = Act.empty(2, 9, 13, 48, torch.device(DEV))
    cat.t.zero_()
    ops.bilinear(tape, Act.from_nchw(_nhwc(coarse), False), (9, 13), False, out=cat.slice(16, 32))
    up = F.interpolate(coarse, size=(9, 13), mode="bilinear", align_corners=False)
    assert rel_err(cat.as_nchw()[:, 16:32], up) < 1e-5
    assert float(cat.t[..., :16].abs().max()) == 0 and float(cat.t[..., 32:].abs().max()) == 0


def test_dropout2d_mask_semantics(hip_lib):
    from gaia_seg_amd.hip import ops
    from gaia_seg_amd.hip.runtime import Act, Tape
    torch.manual_seed(0)
    x = torch.randn(2, 64, 5, 7)
    tape = Tape()
    xa = Act.from_nchw(_nhwc(x), requires_grad=True)
    ya = ops.dropout2d(tape, xa, 0.5, True)
    y = ya.as_nchw().cpu()
    ratio = y / x
    # one Bernoulli draw per (n, channel): the ratio is constant over the spatial dims, 0 or 2
    per = ratio.flatten(2)
    assert torch.allclose(per, per[..., :1].expand_as(per), atol=1e-6)
    vals = per[..., 0].unique()
    assert set(vals.round().tolist()) <= {0.0, 2.0}
    ya.set_grad_from_nchw(_nhwc(torch.ones_like(x)))
    tape.backward()
    assert torch.allclose(xa.g.permute(0, 3, 1, 2).cpu(), per[..., :1].view(2, 64, 1, 1).expand_as(x))


def _ce_ref(logits, label, size, ignore=255, weight=None, cw=None, lw=1.0, align=False):
    up = F.interpolate(logits, size=size, mode="bilinear", align_corners=align)
    loss = F.cross_entropy(up, label, weight=cw, reduction="none", ignore_index=ignore)
    if weight is not None:
        loss = loss * weight
    acc = (up.argmax(1) == label).float().sum() * (100.0 / label.numel())
    return lw * loss.mean(), acc


@pytest.mark.parametrize("shape", [((2, 19, 16, 32), (64, 128)), ((2, 19, 5, 7), (33, 41)),
                                   ((1, 19, 25, 25), (97, 97)), ((2, 150, 4, 4), (16, 16)),
                                   ((1, 19, 8, 8), (8, 8)), ((2, 19, 4, 8), (64, 128)),
                                   ((1, 19, 3, 2), (96, 64)), ((1, 45, 2, 3), (16, 24))])
def test_fused_resize_ce_fwd_bwd(hip_lib, shape):
    from gaia_seg_amd.hip.runtime import Act
    from gaia_seg_amd.models.losses import seg_loss_and_accuracy
    (n, c, h, w), size = shape
    torch.manual_seed(0)
    logits = torch.randn(n, c, h, w) * 3
    label = torch.randint(0, c, (n, *size))
    label[:, :3, :] = 255
    label[0, 5:9, 2:7] = 255
    pw = (torch.rand(n, *size) > 0.3).float()
    for weight, cw in [(None, None), (pw, torch.rand(c) + 0.5)]:
        lr = logits.clone().requires_grad_(True)
        loss_r, acc_r = _ce_ref(lr, label, size, weight=weight, cw=cw, lw=0.4)
        loss_r.backward()
        a = Act.empty(n, h, w, c, torch.device(DEV))
        a.t.copy_(logits.permute(0, 2, 3, 1))
        lg = a.as_nchw().requires_grad_(True)
        loss, acc = seg_loss_and_accuracy(lg, label.to(DEV),
                                          weight.to(DEV) if weight is not None else None,
                                          cw.to(DEV) if cw is not None else None, 255, False, 0.4)
        assert abs(float(loss) - float(loss_r)) < 2e-5 * max(1.0, abs(float(loss_r)))
        assert abs(float(acc) - float(acc_r)) < 1e-3
        (loss * 2.5).backward()
        assert rel_err(lg.grad, lr.grad * 2.5) < 5e-5


@pytest.mark.parametrize("shape,align", [
    (((1, 19, 193, 193), (769, 769)), False),   # config 4's loss: ratio 3.98, ~16-pixel tiles, ragged borders
    (((2, 19, 25, 25), (97, 97)), True),        # align_corners: the last source row / column is hit exactly
    (((2, 45, 5, 7), (21, 30)), False),         # 45 classes: three class passes per tile; ratios 4.2 x 4.29
    (((1, 19, 9, 6), (10, 47)), True),          # nearly 1:1 along H (tiles of one row), 7.8 along W
    (((2, 19, 4, 4), (31, 29)), False),         # ~56 pixels per tile: four passes of a 16-lane row
])
def test_fused_resize_ce_backward_row_tiles(hip_lib, shape, align):
    """The any-scale tile form of the loss backward (csrc/loss.hip ce_bwd_rowtile_kernel: one 16-lane
    row per tile, every softmax term once) against F.interpolate + F.cross_entropy on the CPU, with
    pixel weights, class weights and ignored pixels."""
    from gaia_seg_amd.hip.runtime import Act
    from gaia_seg_amd.models.losses import seg_loss_and_accuracy
    (n, c, h, w), size = shape
    torch.manual_seed(1)
    logits = torch.randn(n, c, h, w) * 3
    label = torch.randint(0, c, (n, *size))
    label[:, :2, :] = 255
    label[0, 3:8, 1:6] = 255
    pw = (torch.rand(n, *size) > 0.3).float()
    for weight, cw in [(None, None), (pw, torch.rand(c) + 0.5)]:
        lr = logits.clone().requires_grad_(True)
        loss_r, acc_r = _ce_ref(lr, label, size, weight=weight, cw=cw, lw=0.4, align=align)
        loss_r.backward()
        a = Act.empty(n, h, w, c, torch.device(DEV))
        a.t.copy_(logits.permute(0, 2, 3, 1))
        lg = a.as_nchw().requires_grad_(True)
        loss, acc = seg_loss_and_accuracy(lg, label.to(DEV),
                                          weight.to(DEV) if weight is not None else None,
                                          cw.to(DEV) if cw is not None else None, 255, align, 0.4)
        assert abs(float(loss) - float(loss_r)) < 2e-5 * max(1.0, abs(float(loss_r)))
        (loss * 2.5).backward()
        assert rel_err(lg.grad, lr.grad * 2.5) < 5e-5


def test_sgd_step_matches_torch(hip_lib):
    import ctypes
    from gaia_seg_amd.hip import lib
    from gaia_seg_amd.hip.runtime import current_stream_ptr
    torch.manual_seed(0)
    n = 4096 + 8
    p = torch.randn(n)
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.SGD([pr], lr=0.01, momentum=0.9, weight_decay=5e-4)
    pg, buf = p.to(DEV), torch.zeros(n, device=DEV)
    for step in range(3):
        g = torch.randn(n)
        pr.grad = g.clone()
        opt.step()
        gg = g.to(DEV)
        zero = step % 2
        lib.check(hip_lib.gs_sgd_step(pg.data_ptr(), gg.data_ptr(), buf.data_ptr(), n, 0.01, 0.9,
                                      5e-4, 1.0, zero, current_stream_ptr()), "sgd")
        # zero_grad = 1 clears the consumed gradient, 0 leaves it
        assert float(gg.abs().max()) == 0.0 if zero else torch.equal(gg.cpu(), g)
    assert rel_err(pg, pr) < 1e-6


# conv -> BN (+ residual) (+ ReLU) through the one-call-per-direction entry points
# (gs_conv_bn_forward / gs_conv_bn_backward), which also fuse the batch statistics into the conv:
# no split-K -> per-tile partials in the conv epilogue; split-K -> slab reduction + statistics pass.
CONV_BN_CASES = [
    # ci  co  k  n   h   w  relu  residual     (rows, K) decide the statistics mode
    (32, 64, 3, 2, 40, 48, True, False),      # 3840 rows, 60 tiles: split-K, fused reduce+stats
    (64, 64, 1, 2, 121, 125, True, True),     # 30250 rows (ragged last tile), epilogue partials
    (64, 256, 1, 2, 96, 96, False, False),    # 18432 rows x 4 column tiles: epilogue partials
    (128, 48, 3, 1, 9, 11, True, False),      # 99 rows: tiny, ragged, split-K
    (16, 80, 3, 2, 64, 64, True, True),       # BN = 80 tiles, two LDS column chunks
]


@pytest.mark.parametrize("case", CONV_BN_CASES)
def test_conv_bn_fused_calls(hip_lib, case):
    from gaia_seg_amd.core.bricks import DynamicBatchNorm2d, DynamicConv2d, conv_bn_act
    from gaia_seg_amd.hip.runtime import tape_function
    ci, co, k, n, h, w, relu, use_res = case
    torch.manual_seed(7)
    conv = DynamicConv2d(ci, co, k, padding=k // 2, bias=False)
    bn = DynamicBatchNorm2d(co)
    torch.nn.init.normal_(conv.weight, 0, 0.2)
    torch.nn.init.uniform_(bn.weight, 0.5, 1.5)
    torch.nn.init.normal_(bn.bias, 0, 0.3)
    x = torch.randn(n, ci, h, w) + 0.5
    res = torch.randn(n, co, h, w) if use_res else None

    # reference: plain PyTorch on the CPU
    w_ref = conv.weight.detach().clone().contiguous().requires_grad_(True)
    g_ref = bn.weight.detach().clone().requires_grad_(True)
    b_ref = bn.bias.detach().clone().requires_grad_(True)
    x_ref = x.clone().requires_grad_(True)
    r_ref = res.clone().requires_grad_(True) if use_res else None
    rm, rv = torch.zeros(co), torch.ones(co)
    y_ref = F.conv2d(x_ref, w_ref, None, 1, k // 2)
    z_ref = F.batch_norm(y_ref, rm, rv, g_ref, b_ref, True, 0.1, 1e-5)
    if use_res:
        z_ref = z_ref + r_ref
    if relu:
        z_ref = F.relu(z_ref)
    gz = torch.randn_like(z_ref)
    z_ref.backward(gz)

    conv, bn = conv.to(DEV), bn.to(DEV).train()
    xg = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    inputs = [xg]
    if use_res:
        rg = res.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        inputs.append(rg)

    def run(tape, acts):
        return [conv_bn_act(tape, conv, bn, acts[0], relu=relu,
                            residual=acts[1] if use_res else None)]
    z = tape_function(run, inputs, True)[0]
    assert rel_err(z, z_ref) < 1e-4
    assert rel_err(bn.running_mean, rm) < 1e-4 and rel_err(bn.running_var, rv) < 1e-4
    z.backward(gz.to(DEV))
    assert rel_err(conv.weight.grad, w_ref.grad) < 2e-4
    assert rel_err(bn.weight.grad, g_ref.grad) < 2e-4
    assert rel_err(bn.bias.grad, b_ref.grad) < 2e-4
    assert rel_err(xg.grad, x_ref.grad) < 2e-4
    if use_res:
        assert rel_err(rg.grad, r_ref.grad) < 1e-5


def test_conv_bn_eval_mode_and_frozen_params(hip_lib):
    """Fused conv+BN entry with running statistics (norm_eval) and with frozen conv / BN parameters:
    no statistics update, no gradient for frozen tensors, input gradient still correct."""
    from gaia_seg_amd.core.bricks import DynamicBatchNorm2d, DynamicConv2d, conv_bn_act
    from gaia_seg_amd.hip.runtime import tape_function
    torch.manual_seed(3)
    ci, co, n, h, w = 32, 48, 2, 20, 24
    conv = DynamicConv2d(ci, co, 3, padding=1, bias=False)
    bn = DynamicBatchNorm2d(co)
    torch.nn.init.normal_(conv.weight, 0, 0.2)
    bn.running_mean.normal_(0, 0.5)
    bn.running_var.uniform_(0.5, 2.0)
    torch.nn.init.uniform_(bn.weight, 0.5, 1.5)
    torch.nn.init.normal_(bn.bias, 0, 0.3)
    x = torch.randn(n, ci, h, w)
    x_ref = x.clone().requires_grad_(True)
    y_ref = F.conv2d(x_ref, conv.weight.detach(), None, 1, 1)
    z_ref = F.relu(F.batch_norm(y_ref, bn.running_mean.clone(), bn.running_var.clone(),
                                bn.weight.detach(), bn.bias.detach(), False, 0.1, 1e-5))
    gz = torch.randn_like(z_ref)
    z_ref.backward(gz)
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()

    conv, bn = conv.to(DEV), bn.to(DEV).eval()
    for p in list(conv.parameters()) + list(bn.parameters()):
        p.requires_grad_(False)
    xg = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    z = tape_function(lambda tape, acts: [conv_bn_act(tape, conv, bn, acts[0], relu=True)], [xg], True)[0]
    assert rel_err(z, z_ref) < 1e-4
    z.backward(gz.to(DEV))
    assert rel_err(xg.grad, x_ref.grad) < 2e-4
    assert conv.weight.grad is None and bn.weight.grad is None and bn.bias.grad is None
    assert torch.equal(bn.running_mean.cpu(), rm0) and torch.equal(bn.running_var.cpu(), rv0)


def test_conv_bn_fused_equals_module_by_module(hip_lib, monkeypatch):
    """The one-call path and the module-by-module path launch the same kernels except for where the
    batch statistics are accumulated: outputs agree to rounding, split-K shapes bit for bit."""
    from gaia_seg_amd.core.bricks import DynamicBatchNorm2d, DynamicConv2d, conv_bn_act
    from gaia_seg_amd.hip.runtime import Act, Tape
    torch.manual_seed(11)
    for (ci, co, k, n, h, w, exact) in [(64, 64, 3, 2, 24, 32, True), (32, 128, 1, 2, 96, 128, False)]:
        conv = DynamicConv2d(ci, co, k, padding=k // 2, bias=False).to(DEV)
        bn = DynamicBatchNorm2d(co).to(DEV).train()
        x = Act(torch.randn(n, h, w, ci, device=DEV), True)
        outs = []
        for fused in (True, False):
            if fused:
                monkeypatch.delenv("GS_NO_FUSED_CALLS", raising=False)
            else:
                monkeypatch.setenv("GS_NO_FUSED_CALLS", "1")
            bn.running_mean.zero_()
            bn.running_var.fill_(1.0)
            outs.append(conv_bn_act(Tape(enabled=False), conv, bn, x, relu=True).t.clone())
        if exact:   # 1536 rows -> 24 tiles: split-K, fused reduce+stats is bit-identical
            assert torch.equal(outs[0], outs[1])
        else:
            assert rel_err(outs[0], outs[1]) < 1e-5


def test_syncbn_exchange_kernels_match_the_formula(hip_lib):
    """gs_bn_sync_local / gs_bn_sync_merge (the device path of the SyncBN statistics exchange)
    against the tensor formula (Chan et al.) on a fake three-rank gather with unequal counts."""
    import ctypes
    from gaia_seg_amd.hip import lib
    L = lib.load()
    torch.manual_seed(5)
    C, counts = 24, [40.0, 25.0, 70.0]
    st = torch.cuda.current_stream().cuda_stream
    gathered = torch.empty((3, 2 * C + 1), dtype=torch.float64, device=DEV)
    ref_mean, ref_var = [], []
    for r, n in enumerate(counts):
        x = torch.randn(int(n), C, dtype=torch.float64) * (1 + r) + r
        shift = x[0].clone()
        d = x - shift
        sums = torch.cat([d.sum(0), (d * d).sum(0), shift]).float().to(DEV)
        lib.check(L.gs_bn_sync_local(sums.data_ptr(), n, C, gathered[r].data_ptr(), st), "local")
        ref_mean.append(x.mean(0))
        ref_var.append(x.var(0, unbiased=False))
    g = gathered.cpu()
    for r in range(3):
        assert torch.allclose(g[r, :C], ref_mean[r], atol=1e-5)
        assert torch.allclose(g[r, C:2 * C], ref_var[r], rtol=1e-4, atol=1e-5)
        assert float(g[r, 2 * C]) == counts[r]
    merged = torch.empty(3 * C, dtype=torch.float32, device=DEV)
    lib.check(L.gs_bn_sync_merge(gathered.data_ptr(), 3, C, merged.data_ptr(), st), "merge")
    total = sum(counts)
    cnt = torch.tensor(counts, dtype=torch.float64)[:, None]
    gmean = (g[:, :C] * cnt).sum(0) / total
    gvar = ((g[:, C:2 * C] + (g[:, :C] - gmean) ** 2) * cnt).sum(0) / total
    m = merged.cpu().double()
    assert float(m[:C].abs().max()) == 0.0
    assert torch.allclose(m[C:2 * C] / total, gvar, rtol=1e-5)
    assert torch.allclose(m[2 * C:], gmean, atol=1e-5)


# BN + ReLU evaluated in the consumer's operand loaders (gs_conv_desc.in_affine): conv_a -> bn_a ->
# relu -> conv_b -> bn_b with the intermediate activation deferred must equal the written-out form
# bit for bit (same expression, same MFMA order), forward and backward, and match PyTorch on the CPU.
DEFER_CASES = [
    # ci  mid  co  kb  stride dil  n   h   w
    (64, 64, 256, 1, 1, 1, 2, 40, 56),     # bn2 -> conv3 (1x1), several column tiles
    (64, 48, 48, 3, 1, 1, 2, 33, 37),      # bn1 -> conv2 (3x3): padding must be zero AFTER the BN
    (32, 96, 96, 3, 2, 1, 2, 32, 48),      # strided 3x3 (first block of a stage)
    (32, 64, 64, 3, 1, 2, 1, 24, 24),      # dilated 3x3 (OS8 stages)
    (16, 640, 80, 1, 1, 1, 2, 8, 12),      # widest operand (640 channels), long K: split-K wgrad
    (16, 128, 64, 3, 1, 1, 2, 8, 8),       # 128 rows: split-K forward + paired K loop
    (32, 80, 64, 3, 1, 1, 2, 20, 32),      # ragged 80-channel operand with the affine loader
    (32, 64, 48, 3, 1, 1, 2, 40, 16),      # narrow image, many rows
]


@pytest.mark.parametrize("case", DEFER_CASES)
def test_deferred_bn_relu_in_operand_loaders(hip_lib, case, monkeypatch, cpu_norm="max"):
    import gaia_seg_amd.hip.ops as ops
    from gaia_seg_amd.core.bricks import DynamicBatchNorm2d, DynamicConv2d, conv_bn_act
    from gaia_seg_amd.hip.runtime import tape_function
    ci, mid, co, kb, stride, dil, n, h, w = case
    torch.manual_seed(5)
    ca = DynamicConv2d(ci, mid, 1, bias=False)
    ba = DynamicBatchNorm2d(mid)
    cb = DynamicConv2d(mid, co, kb, stride=stride, padding=dil * (kb // 2), dilation=dil, bias=False)
    bb = DynamicBatchNorm2d(co)
    for c in (ca, cb):
        torch.nn.init.normal_(c.weight, 0, 0.2)
    for b in (ba, bb):
        torch.nn.init.uniform_(b.weight, 0.5, 1.5)
        torch.nn.init.normal_(b.bias, 0, 0.3)
    x = torch.randn(n, ci, h, w) + 0.3

    # CPU reference
    xr = x.clone().requires_grad_(True)
    pr = [p.detach().clone().contiguous().requires_grad_(True)
          for p in (ca.weight, ba.weight, ba.bias, cb.weight, bb.weight, bb.bias)]
    t = F.relu(F.batch_norm(F.conv2d(xr, pr[0]), None, None, pr[1], pr[2], True, 0.1, 1e-5))
    z_ref = F.relu(F.batch_norm(F.conv2d(t, pr[3], None, stride, dil * (kb // 2), dil), None, None,
                                pr[4], pr[5], True, 0.1, 1e-5))
    gz = torch.randn_like(z_ref)
    z_ref.backward(gz)

    mods = [m.to(DEV) for m in (ca, ba, cb, bb)]
    ca, ba, cb, bb = mods
    ba.train(), bb.train()
    results = []
    for defer in (True, False):
        monkeypatch.setattr(ops, "DEFER_BN", defer)
        for m in mods:
            for p in m.parameters():
                p.grad = None
        xg = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        seen = {}

        def run(tape, acts):
            mid_act = conv_bn_act(tape, ca, ba, acts[0], relu=True, defer=True)  # (explicit: not via DEFER_EDGES)
            seen["deferred"] = mid_act.affine is not None
            return [conv_bn_act(tape, cb, bb, mid_act, relu=True)]
        z = tape_function(run, [xg], True)[0]
        assert seen["deferred"] == defer
        z.backward(gz.to(DEV))
        results.append([z.detach().clone(), xg.grad.clone()] +
                       [p.grad.clone() for m in mods for p in m.parameters()])
    for a, b in zip(*results):
        assert torch.equal(a, b)            # loader fusion == written-out activation, bit for bit
    z, dx, g_ca, g_baw, g_bab, g_cb, g_bbw, g_bbb = results[0]
    assert rel_err(z, z_ref) < 1e-4
    if cpu_norm == "l2":
        # large cases (tests/test_stream_1x1_gpu.py: millions of activations): a ReLU within rounding
        # of zero may fall on the other side than on the CPU and moves single entries by O(1e-2); the
        # bitwise equality above is the sharp statement, the CPU comparison is held in the L2 norm
        def err(a, b):
            return float((a.double().cpu() - b.double()).norm() / b.double().norm())
        tol = 3e-3
    else:
        err, tol = rel_err, 3e-4
    assert err(dx, xr.grad) < tol
    for got, ref in zip((g_ca, g_baw, g_bab, g_cb, g_bbw, g_bbb), pr):
        assert err(got, ref.grad) < tol


# BatchNorm-backward reduction folded into the consumer's dgrad epilogue (gs_bn_bwd_fuse): a stage of
# two bottlenecks (the second without projection shortcut, so conv1's dgrad accumulates onto the
# identity gradient and owns its input's gradient) must give the same gradients with and without the
# fusion, and the fusion must actually have run (mode 1 for bn1 / bn2, mode 2 for bn3).
@pytest.mark.parametrize("shape", [(2, 64, 96, 16), (1, 40, 72, 48)])
def test_bn_backward_reduction_in_dgrad_epilogue(hip_lib, shape, monkeypatch):
    import gaia_seg_amd.hip.ops as ops
    from gaia_seg_amd.core.bricks import DynamicBottleneck
    from gaia_seg_amd.models.utils import DynamicResLayer
    n, h, w, planes = shape
    torch.manual_seed(2)
    layer = DynamicResLayer(DynamicBottleneck, 32, planes, depth=3, stride=1,
                            conv_cfg=dict(type="DynConv2d"), norm_cfg=dict(type="DynBN"))
    for m in layer.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            torch.nn.init.uniform_(m.weight, 0.5, 1.5)
            torch.nn.init.normal_(m.bias, 0, 0.2)
        elif hasattr(m, "weight") and getattr(m, "weight", None) is not None and m.weight.dim() == 4:
            torch.nn.init.normal_(m.weight, 0, (2.0 / (m.weight.shape[1] * m.weight.shape[2] ** 2)) ** 0.5)
    layer = layer.to(DEV).train()
    x = torch.randn(n, 32, h, w)
    gz = torch.randn(n, 4 * planes, h, w)
    results, counts, mask_counts = [], [], []
    # (fused with the ReLU mask of bn3 as bytes -- gs_bn_bwd_fuse mode 3, the default; fused with the
    # mask read from the activation -- mode 2; unfused)
    for fuse, mask_bytes in ((True, True), (True, False), (False, True)):
        monkeypatch.setattr(ops, "BNBWD_FUSE", fuse)
        monkeypatch.setattr(ops, "RELU_MASK_BYTES", mask_bytes)
        ops.BNBWD_FUSED_COUNT = 0
        ops.BNBWD_FUSED_MASK_COUNT = 0
        for p in layer.parameters():
            p.grad = None
        for m in layer.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.running_mean.zero_()
                m.running_var.fill_(1)
        xg = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        z = layer(xg)
        z.backward(gz.to(DEV).contiguous(memory_format=torch.channels_last))
        counts.append(ops.BNBWD_FUSED_COUNT)
        mask_counts.append(ops.BNBWD_FUSED_MASK_COUNT)
        results.append([z.detach().clone(), xg.grad.clone()] + [p.grad.clone() for p in layer.parameters()])
    # block 0: bn1 (conv2 owns), bn2 (conv3 owns); blocks 1, 2: + bn3 of the block before (conv1 owns).
    # (second shape: some of the dgrads are split along K — there the slab reduce does the fusion)
    assert counts == [2 + 3 + 3, 2 + 3 + 3, 0]
    assert mask_counts == [2, 0, 0]          # the two bn3 masks came from the bytes
    for a, b in zip(results[0], results[1]):
        assert torch.equal(a, b)             # the same mask, the same arithmetic: bit for bit
    assert torch.equal(results[0][0], results[2][0])
    for a, b in zip(results[0][1:], results[2][1:]):
        assert rel_err(a, b) < 2e-5
