"""ORACLE (test infrastructure — never imported by the product path).

Plain-PyTorch CPU restatement of the operators on the GAIA-seg supernet hot path.  Each function
cites the reference lines it follows.  Operators whose source is NOT in /root/reference
(gaiavision DynConv2d / DynBN, mmseg resize / OHEM) follow the contract reconstructed from the
reference call sites in SURVEY.md Appendix A and are therefore PARITY UNPINNED (no reference test,
golden vector or runnable reference exists for them); `weight_reduce_loss` and `accuracy` ARE
pinned: tests/golden/ref_loss_utils.npz holds outputs of the reference's own
gaiaseg/models/losses/utils.py and accuracy.py run in the build container
(tests/golden/make_ref_loss_fixtures.py).
"""
import torch
import torch.nn.functional as F


# ---- gaivision DynConv2d: SURVEY.md Appendix A1; call site dynamic_fcn_head.py:76 -------------
def dyn_conv2d(x, weight, bias, width, stride=1, padding=0, dilation=1):
    """F.conv2d(x, weight[:width, :x.size(1)], bias[:width], ...): leading-slice convention
    (in-tree evidence of the convention: gaiaseg/models/backbones/dynamic_convnext.py:95)."""
    w = weight[:width, :x.size(1)]
    b = bias[:width] if bias is not None else None
    return F.conv2d(x, w, b, stride, padding, dilation, 1)


# ---- gaiavision DynBN / DynSyncBN(group_size=1): SURVEY.md Appendix A2 -------------------------
def dyn_batch_norm(x, running_mean, running_var, weight, bias, training, momentum=0.1, eps=1e-5):
    """F.batch_norm on the leading x.size(1) channels; running stats updated on the slice."""
    c = x.size(1)
    rm = running_mean[:c] if running_mean is not None else None
    rv = running_var[:c] if running_var is not None else None
    w = weight[:c] if weight is not None else None
    b = bias[:c] if bias is not None else None
    return F.batch_norm(x, rm, rv, w, b, training or rm is None, momentum, eps)


# ---- ReLU with an optional externally supplied mask (gradient-parity instrument) ---------------
class ReluMasks:
    """Context for `relu(x, key)`: every ReLU of the model path follows a BatchNorm, `key` is that
    BN's module name.

    Why: two correct fp32 implementations can disagree on the sign of a pre-activation that is
    within rounding of zero; that single ReLU branch flip changes every upstream gradient by a
    finite amount (tests/test_grad_criterion.py demonstrates it), so gradients of the HIP path and of
    this restatement can only be compared tightly on the SAME branch pattern.  With `masks` given
    (key -> bool tensor, the HIP path's `z > 0`), relu(x) = x * mask: the function is then smooth
    around the evaluation point and gradients must agree to rounding.  Every position where the
    given mask differs from this side's own `x > 0` is recorded with |x| relative to the tensor's
    RMS, so the test can REQUIRE that disagreements occur only within rounding of zero.
    With masks=None the context only records this side's own masks (self.own).

    `witness`: masks of a second fp32 implementation (this oracle run natively in fp32).  Random
    deep ReLU networks amplify rounding noise with depth (40+ residual blocks take 6e-8 to ~1e-3 of
    the activation RMS), so "within rounding of zero" is measured against what the reference's own
    fp32 arithmetic does at the same layer: witness_flips records the witness's disagreements with
    this (fp64) side in the same units as flips.  For small tensors (<= SMALL_PRE elements, e.g. the
    PPM's pool-scale-1 branch: N x 512 x 1 x 1, BatchNorm over N values per channel) sign
    disagreements are too few to be a statistic, so the witness also hands over its pre-activations
    (`witness_pre`) and witness_noise[key] = max |x_fp32 - x_fp64| / rms is the fp32 arithmetic's own
    error level at that layer."""
    SMALL_PRE = 1 << 20

    def __init__(self, masks=None, keep_own=False, pools=None, witness=None, keep_pre=False,
                 witness_pre=None):
        self.masks = masks
        self.witness = witness     # key -> bool mask of ANOTHER fp32 implementation (the fp32 oracle):
        self.witness_flips = {}    #   its disagreements with this side's signs, recorded like flips
        self.witness_pre = witness_pre or {}   # key -> the witness's pre-activation (small tensors)
        self.witness_noise = {}    # key -> max |x_witness - x| / rms(x)
        self.small_pre = {}        # keep_own: pre-activations of the small tensors
        self.keep_pre = keep_pre
        self.pools = pools         # key -> uint8 [N, C, Ho, Wo] tap index (kh * k + kw) of the maximum
        self.pool_flips = {}       # key -> (count, max (own max - chosen value) / rms(x))
        self.keep_own = keep_own
        self.own = {}
        self.pre = {}          # keep_own: the pre-activations themselves
        self.flips = {}        # key -> (count, max |x| / rms(x) over the flipped positions)
        self.used = set()

    def apply(self, x, key):
        own = x.detach() > 0
        if self.keep_own:
            self.own[key] = own
            if x.numel() <= self.SMALL_PRE:
                self.small_pre[key] = x.detach().clone()
        if key in self.witness_pre:
            xd = x.detach().double()
            rms = float(xd.pow(2).mean().sqrt().clamp_min(1e-30))
            self.witness_noise[key] = float((self.witness_pre[key].double() - xd).abs().max()) / rms
        if self.keep_pre:
            self.pre[key] = x.detach()
        if self.witness is not None and key in self.witness:
            wd = own != self.witness[key]
            nw = int(wd.sum())
            if nw:
                xd = x.detach()
                rms = float(xd.double().pow(2).mean().sqrt().clamp_min(1e-30))
                self.witness_flips[key] = (nw, float(xd[wd].abs().max()) / rms)
        if self.masks is None:
            return torch.relu(x)
        if key not in self.masks:
            raise KeyError("no ReLU mask supplied for %r" % key)
        if key in self.used:
            raise RuntimeError("ReLU key %r used twice in one forward" % key)
        self.used.add(key)
        m = self.masks[key]
        if m.shape != x.shape:
            raise ValueError("mask %s vs activation %s at %r" % (tuple(m.shape), tuple(x.shape), key))
        diff = own != m
        n = int(diff.sum())
        if n:
            xd = x.detach()
            rms = float(xd.double().pow(2).mean().sqrt().clamp_min(1e-30))
            self.flips[key] = (n, float(xd[diff].abs().max()) / rms)
        return x * m.to(x.dtype)

    def apply_pool(self, x, key, k, s, p):
        """max_pool2d with the argmax given: out = x[chosen tap].  Where the given tap is not this
        side's own maximum the two values must tie within rounding (recorded like ReLU flips)."""
        idx = self.pools[key].long()
        n, c, h, w = x.shape
        ho, wo = idx.shape[2], idx.shape[3]
        oy = torch.arange(ho).view(1, 1, ho, 1)
        ox = torch.arange(wo).view(1, 1, 1, wo)
        iy = oy * s - p + idx // k
        ix = ox * s - p + idx % k
        assert bool(((iy >= 0) & (iy < h) & (ix >= 0) & (ix < w)).all()), "pool tap outside the input"
        out = torch.gather(x.flatten(2), 2, (iy * w + ix).flatten(2)).view(n, c, ho, wo)
        own = F.max_pool2d(x.detach(), k, s, p)
        gap = own - out.detach()
        nflip = int((gap != 0).sum())
        if nflip:
            rms = float(x.detach().double().pow(2).mean().sqrt().clamp_min(1e-30))
            self.pool_flips[key] = (nflip, float(gap.max()) / rms)
        return out

    def __enter__(self):
        global _RELU_CTX
        self._prev = _RELU_CTX
        _RELU_CTX = self
        return self

    def __exit__(self, *exc):
        global _RELU_CTX
        _RELU_CTX = self._prev
        return False


_RELU_CTX = None


def relu(x, key=None):
    """torch.relu, or the masked form when a ReluMasks context is active and the call is keyed."""
    if _RELU_CTX is None or key is None:
        return torch.relu(x)
    return _RELU_CTX.apply(x, key)


def max_pool2d(x, k, s, p, key=None):
    """F.max_pool2d, or the gather form on externally supplied argmax taps (ReluMasks.pools)."""
    if _RELU_CTX is None or key is None or _RELU_CTX.pools is None or key not in _RELU_CTX.pools:
        return F.max_pool2d(x, k, s, p)
    return _RELU_CTX.apply_pool(x, key, k, s, p)


# ---- mmseg.ops.resize == F.interpolate: call sites dynamic_fcn_head.py:141-145 -----------------
def resize(input, size=None, scale_factor=None, mode="nearest", align_corners=None):
    return F.interpolate(input, size, scale_factor, mode, align_corners)


# ---- losses: gaiaseg/models/losses/utils.py:7-55 -----------------------------------------------
def reduce_loss(loss, reduction):
    """utils.py:7-23"""
    if reduction == "none":
        return loss
    if reduction == "mean":
        return loss.mean()
    if reduction == "sum":
        return loss.sum()
    raise ValueError(reduction)


def weight_reduce_loss(loss, weight=None, reduction="mean", avg_factor=None):
    """utils.py:26-55: elementwise weight, then mean over ALL elements unless avg_factor."""
    if weight is not None:
        assert weight.dim() == loss.dim()
        if weight.dim() > 1:
            assert weight.size(1) == 1 or weight.size(1) == loss.size(1)
        loss = loss * weight
    if avg_factor is None:
        loss = reduce_loss(loss, reduction)
    else:
        if reduction == "mean":
            loss = loss.sum() / avg_factor
        elif reduction != "none":
            raise ValueError('avg_factor can not be used with reduction="sum"')
    return loss


def cross_entropy(pred, label, weight=None, class_weight=None, reduction="mean", avg_factor=None,
                  ignore_index=255):
    """gaiaseg/models/losses/cross_entropy_loss.py:67-94"""
    loss = F.cross_entropy(pred, label, weight=class_weight, reduction="none",
                           ignore_index=ignore_index)
    if weight is not None:
        weight = weight.float()
    return weight_reduce_loss(loss, weight=weight, reduction=reduction, avg_factor=avg_factor)


def accuracy(pred, target, topk=1, thresh=None):
    """gaiaseg/models/losses/accuracy.py:4-49 (top-k, % of target.numel())."""
    assert isinstance(topk, (int, tuple))
    if isinstance(topk, int):
        topk = (topk,)
        return_single = True
    else:
        return_single = False
    maxk = max(topk)
    if pred.size(0) == 0:
        accu = [pred.new_tensor(0.) for _ in range(len(topk))]
        return accu[0] if return_single else accu
    assert pred.ndim == target.ndim + 1
    assert pred.size(0) == target.size(0)
    assert maxk <= pred.size(1)
    pred_value, pred_label = pred.topk(maxk, dim=1)
    pred_label = pred_label.transpose(0, 1)
    correct = pred_label.eq(target.unsqueeze(0).expand_as(pred_label))
    if thresh is not None:
        correct = correct & (pred_value > thresh).t()
    res = []
    for k in topk:
        correct_k = correct[:k].reshape(-1).float().sum(0, keepdim=True)
        res.append(correct_k.mul_(100.0 / target.numel()))
    return res[0] if return_single else res


def seg_losses(seg_logit, seg_label, loss_weight=1.0, ignore_index=255, align_corners=False,
               sampler=None, class_weight=None):
    """The `losses` recipe shared by the heads: dynamic_fcn_head.py:137-159
    (== dynamic_psp_head.py:149-173 minus the 'resize_logit' entry)."""
    loss = dict()
    seg_logit = resize(seg_logit, size=seg_label.shape[2:], mode="bilinear",
                       align_corners=align_corners)
    seg_weight = sampler(seg_logit, seg_label) if sampler is not None else None
    seg_label = seg_label.squeeze(1)
    loss["loss_seg"] = loss_weight * cross_entropy(seg_logit, seg_label, weight=seg_weight,
                                                   class_weight=class_weight,
                                                   ignore_index=ignore_index)
    loss["acc_seg"] = accuracy(seg_logit, seg_label)
    return loss


# ---- mmseg OHEMPixelSampler: SURVEY.md Appendix A11; call sites dynamic_fcn_head.py:70-71,147-148
def ohem_pixel_weights(seg_logit, seg_label, thresh=None, min_kept=100000, ignore_index=255):
    with torch.no_grad():
        assert seg_logit.shape[2:] == seg_label.shape[2:]
        assert seg_label.shape[1] == 1
        seg_label = seg_label.squeeze(1).long()
        batch_kept = min_kept * seg_label.size(0)
        valid_mask = seg_label != ignore_index
        seg_weight = seg_logit.new_zeros(size=seg_label.size())
        valid_seg_weight = seg_weight[valid_mask]
        if thresh is not None:
            seg_prob = F.softmax(seg_logit, dim=1)
            tmp_seg_label = seg_label.clone().unsqueeze(1)
            tmp_seg_label[tmp_seg_label == ignore_index] = 0
            seg_prob = seg_prob.gather(1, tmp_seg_label).squeeze(1)
            sort_prob, sort_indices = seg_prob[valid_mask].sort()
            if sort_prob.numel() > 0:
                min_threshold = sort_prob[min(batch_kept, sort_prob.numel() - 1)]
            else:
                min_threshold = 0.0
            threshold = max(min_threshold, thresh)
            valid_seg_weight[seg_prob[valid_mask] < threshold] = 1.
        else:
            losses = F.cross_entropy(seg_logit, seg_label, reduction="none",
                                     ignore_index=ignore_index)
            _, sort_indices = losses[valid_mask].sort(descending=True)
            valid_seg_weight[sort_indices[:batch_kept]] = 1.
        seg_weight[valid_mask] = valid_seg_weight
        return seg_weight
