"""Parity instrument shared by the GPU model tests: one HIP train step against ONE fp64 pass of the
CPU oracle evaluated on the HIP path's ReLU branch pattern.

Forward quantities (losses, accuracy, BN running statistics, logits) are compared in the max norm
at BASELINE.json's 1e-3; gradients in the max norm at the same 1e-3 — per parameter, relative to
that parameter's largest oracle gradient — which is possible because the oracle is given the HIP
path's ReLU masks (oracle/ops.py ReluMasks): with equal branch patterns both sides evaluate the same
smooth function and differ by rounding only.  Where a supplied mask disagrees with the oracle's own
sign the pre-activation must be within rounding of zero (|x| <= FLIP_TOL * rms(x)) and such
positions must be rare; otherwise the test fails.  tests/test_grad_criterion.py shows what a single
unchecked flip does to a max-norm comparison, which is why r01's loose L2 criterion existed and why
it is gone.
"""
import torch

from conftest import rel_err

TOL = 1e-3          # BASELINE.json: 1e-3 rel fp32, forward and gradients alike
FLIP_TOL = 1e-3     # a mask disagreement is legitimate only if |pre-activation| <= this * rms
FLIP_FRAC = 1e-4    # ... and at most this fraction of a layer's activations may disagree


def hip_train_step(prod, img, gt, metas=None):
    """One forward+backward of the product model on cuda.  Returns (train_step output, masks) with
    masks = {BN module name: bool NCHW cpu tensor 'post-ReLU output > 0'}."""
    from gaia_seg_amd.hip import ops
    n, _, h, w = img.shape
    metas = metas or [dict(ori_shape=(h, w, 3), img_shape=(h, w, 3), flip=False) for _ in range(n)]
    ops.RELU_TRACE = []
    try:
        out = prod.train_step(dict(img=img.cuda(), img_metas=metas, gt_semantic_seg=gt.cuda()), None)
        trace = ops.RELU_TRACE
    finally:
        ops.RELU_TRACE = None
    out["loss"].backward()
    torch.cuda.synchronize()
    names = {id(p): k for k, p in prod.named_parameters()}
    masks = {}
    for w_, m in trace:
        key = names[id(w_)]
        assert key.endswith(".weight"), key
        key = key[:-len(".weight")]
        assert key not in masks, "ReLU after %s traced twice" % key
        masks[key] = m.permute(0, 3, 1, 2).cpu()
    return out, masks


def oracle_step_fp64(orc, img, gt, masks):
    """fp64 forward+backward of the oracle on the given ReLU masks; returns (losses, loss, ctx)."""
    from oracle import ops as O
    orc.double()
    with O.ReluMasks(masks) as ctx:
        losses = orc.forward_train(img.double(), gt)
        loss, _ = orc.parse_losses(losses)
        loss.backward()
    unused = set(masks) - ctx.used
    assert not unused, "HIP path applied ReLUs the oracle did not: %s" % sorted(unused)[:5]
    return losses, loss, ctx


def check_flips(ctx, masks):
    """Mask disagreements are only allowed within rounding of zero, and must be rare."""
    total = 0
    for key, (n, rel) in ctx.flips.items():
        numel = masks[key].numel()
        assert rel <= FLIP_TOL, ("ReLU after %s: %d sign disagreement(s) at |x|/rms = %.2e — not a "
                                 "rounding-level flip" % (key, n, rel))
        assert n <= max(2, FLIP_FRAC * numel), "ReLU after %s: %d of %d signs disagree" % (key, n, numel)
        total += n
    return total


def compare_step(prod, orc, out, losses_o, loss_o, gt, check_grads=True, check_buffers=True,
                 min_checked=10):
    """HIP step vs oracle step: log vars, total loss, BN buffers, every parameter gradient."""
    errs = {}
    for k, v in losses_o.items():
        if k.endswith("acc_seg"):
            # accuracy (percent) is a step function of the logits: allow two argmax ties
            npix = float(gt.numel())
            flips = abs(float(out["log_vars"][k]) - float(v)) / 100.0 * npix
            errs[k] = 0.0 if flips <= 2.01 else flips
        else:
            errs[k] = abs(float(out["log_vars"][k]) - float(v)) / max(abs(float(v)), 1e-6)
    errs["loss"] = abs(float(out["loss"]) - float(loss_o)) / abs(float(loss_o))
    if check_grads:
        op = dict(orc.named_parameters())
        n_checked = 0
        for name, p in prod.named_parameters():
            go = op[name].grad
            if go is None or float(go.abs().max()) == 0.0:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, \
                    "%s: gradient for a parameter the subnet does not use" % name
                continue
            assert p.grad is not None, name
            errs["grad:" + name] = rel_err(p.grad, go)   # max norm
            n_checked += 1
        assert n_checked > min_checked
    if check_buffers:
        ob = dict(orc.named_buffers())
        for name, b in prod.state_dict().items():  # state_dict() folds the host-side BN counters
            if name.endswith("running_mean") or name.endswith("running_var"):
                errs["buf:" + name] = rel_err(b, ob[name])
            elif name.endswith("num_batches_tracked"):
                assert int(b) == int(ob[name]), name
    bad = sorted([(k, v) for k, v in errs.items() if not v < TOL], key=lambda kv: -kv[1])
    assert not bad, "%d mismatches, worst: %s" % (len(bad), [(k, "%.2e" % v) for k, v in bad[:12]])
    return errs


def train_step_parity(prod, orc, img, gt, check_grads=True):
    """The whole protocol; prod is on cuda / train mode with its arch set, orc likewise (CPU)."""
    out, masks = hip_train_step(prod, img, gt)
    losses_o, loss_o, ctx = oracle_step_fp64(orc, img, gt, masks)
    nflips = check_flips(ctx, masks)
    errs = compare_step(prod, orc, out, losses_o, loss_o, gt, check_grads=check_grads)
    errs["_relu_flips"] = nflips
    return errs
