"""CrossEntropyLoss / accuracy with mmseg's contract, computed by ONE fused HIP kernel pair.

Reference semantics (in-tree copies of the mmseg code the live path calls through
``build_loss(loss_decode)``, gaiaseg/models/decode_heads/dynamic_fcn_head.py:67,137-159):
  * cross_entropy       gaiaseg/models/losses/cross_entropy_loss.py:67-94
  * weight_reduce_loss  gaiaseg/models/losses/utils.py:26-55  (mean over ALL elements, ignored
                        pixels contribute 0 but count in the denominator)
  * accuracy            gaiaseg/models/losses/accuracy.py:4-49 (top-1, % of label.numel())

The heads call ``seg_loss_and_accuracy`` with the LOW-resolution logits: the bilinear resize to the
label size (dynamic_fcn_head.py:141-145) happens inside the kernel, the [N,C,H,W] tensor is never
materialised.
"""
import ctypes

import torch
import torch.nn as nn

from ...hip import lib as _lib
from ...hip.runtime import WORKSPACE, current_stream_ptr, require_gpu_tensor, round_up
from ..builder import LOSSES


def _ce_desc(logits, label_hw, ignore_index, align_corners):
    n, c, h, w = logits.shape
    d = _lib.CeDesc()
    d.N, d.h, d.w, d.Cls = n, h, w, c
    d.H, d.W = int(label_hw[0]), int(label_hw[1])
    d.l_sn, d.l_sc, d.l_sh, d.l_sw = logits.stride()
    d.ignore_index = -100 if ignore_index is None else int(ignore_index)
    d.align_corners = 1 if align_corners else 0
    return d


class _FusedResizeCE(torch.autograd.Function):
    """(loss_scale * sum_i w_i*ce_i, acc_scale * #correct) of bilinearly resized logits, both fp32
    scalars written by the loss kernel's last launch; backward gathers into the low-resolution
    logits."""

    @staticmethod
    def forward(ctx, logits, label, pixel_weight, class_weight, ignore_index, align_corners,
                loss_scale, acc_scale):
        require_gpu_tensor(logits, "seg_logit")
        L = _lib.load()
        dev = logits.device
        label = label.contiguous()
        if label.dtype != torch.int64:
            label = label.long()
        n, hh, ww = label.shape
        d = _ce_desc(logits, (hh, ww), ignore_index, align_corners)
        if pixel_weight is not None:
            pixel_weight = pixel_weight.contiguous().float()
        if class_weight is not None:
            class_weight = class_weight.contiguous().float()
        lse = torch.empty((n, hh, ww), dtype=torch.float32, device=dev)
        out = torch.empty(2, dtype=torch.float32, device=dev)
        nb = L.gs_ce_workspace_bytes(ctypes.byref(d))
        ws = WORKSPACE.get(nb, dev)
        _lib.check(L.gs_ce_forward_scaled(
            ctypes.byref(d), logits.data_ptr(), label.data_ptr(),
            pixel_weight.data_ptr() if pixel_weight is not None else None,
            class_weight.data_ptr() if class_weight is not None else None,
            lse.data_ptr(), loss_scale, acc_scale, out.data_ptr(), ws.data_ptr(), ws.numel(),
            current_stream_ptr()), "gs_ce_forward_scaled")
        ctx.desc = d
        ctx.loss_scale = loss_scale
        ctx.save_for_backward(logits, label, lse)
        ctx.pixel_weight, ctx.class_weight = pixel_weight, class_weight
        # (two views of the kernel's output: no launches; r01 produced them with five tiny kernels)
        loss, acc = out[0], out[1]
        ctx.mark_non_differentiable(acc)
        return loss, acc

    @staticmethod
    def backward(ctx, grad_loss, grad_acc):
        from ...hip import ops as _ops
        prev_slot = _ops.adopt_current_stream()   # (the auxiliary head's loss runs on the branch stream)
        try:
            return _FusedResizeCE._backward(ctx, grad_loss)
        finally:
            _ops.restore_stream_slot(prev_slot)

    @staticmethod
    def _backward(ctx, grad_loss):
        logits, label, lse = ctx.saved_tensors
        L = _lib.load()
        d = ctx.desc
        n, c, h, w = logits.shape
        ld = round_up(c, 4)
        buf = torch.empty((n, h, w, ld), dtype=torch.float32, device=logits.device)
        pw, cw = ctx.pixel_weight, ctx.class_weight
        nb = L.gs_ce_backward_workspace_bytes(ctypes.byref(d), ld)
        ws = WORKSPACE.get(nb, logits.device)
        _lib.check(L.gs_ce_backward_ws(ctypes.byref(d), logits.data_ptr(), label.data_ptr(),
                                       pw.data_ptr() if pw is not None else None,
                                       cw.data_ptr() if cw is not None else None, lse.data_ptr(),
                                       ctx.loss_scale, buf.data_ptr(), ld, ws.data_ptr(), ws.numel(),
                                       current_stream_ptr()), "gs_ce_backward_ws")
        # scale by the upstream scalar on device (no host sync); the tensor is low resolution
        buf.mul_(grad_loss)
        dlogits = buf[..., :c].permute(0, 3, 1, 2)
        return dlogits, None, None, None, None, None, None, None


def seg_loss_and_accuracy(seg_logit, seg_label, weight=None, class_weight=None, ignore_index=255,
                          align_corners=False, loss_weight=1.0):
    """Returns (loss_seg, acc_seg) exactly as the reference's ``losses`` does after resizing:
    loss = loss_weight * sum_i(w_i * ce_i) / label.numel();  acc = 100 * correct / label.numel()."""
    if seg_label.dim() == 4:
        seg_label = seg_label.squeeze(1)
    numel = seg_label.numel()
    return _FusedResizeCE.apply(seg_logit, seg_label, weight, class_weight, ignore_index,
                                align_corners, loss_weight / numel, 100.0 / numel)


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    """mmseg CrossEntropyLoss(use_sigmoid=False): kwargs as in the configs
    (configs/_dynamic_/models/pspnet_ar50to101v2_gsync.py:36-37,50-51)."""

    def __init__(self, use_sigmoid=False, use_mask=False, reduction="mean", class_weight=None,
                 loss_weight=1.0):
        super().__init__()
        if use_sigmoid or use_mask:
            raise NotImplementedError("only the softmax cross entropy is on the supernet path")
        if reduction != "mean":
            raise NotImplementedError("only reduction='mean' is used by the decode heads")
        self.use_sigmoid, self.use_mask = use_sigmoid, use_mask
        self.reduction, self.loss_weight = reduction, loss_weight
        self.class_weight = class_weight
        self.align_corners = False  # set by the head: the resize is fused into the loss kernel

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None,
                ignore_index=255, **kwargs):
        """``cls_score`` may be at any resolution; it is (virtually) resized to ``label``'s."""
        assert reduction_override in (None, "mean")
        if avg_factor is not None:
            raise NotImplementedError("avg_factor is never passed by the decode heads")
        cw = None
        if self.class_weight is not None:
            cw = cls_score.new_tensor(self.class_weight)
        loss, _ = seg_loss_and_accuracy(cls_score, label, weight, cw, ignore_index,
                                        kwargs.get("align_corners", self.align_corners),
                                        self.loss_weight)
        return loss


def accuracy(pred, target, topk=1, thresh=None, align_corners=False, ignore_index=255):
    """Top-1 accuracy in % of target.numel() (losses/accuracy.py:38-49); pred may be low-res."""
    if topk != 1 or thresh is not None:
        raise NotImplementedError("only top-1 accuracy is used on the supernet path")
    with torch.no_grad():
        _, acc = seg_loss_and_accuracy(pred.detach(), target, None, None, ignore_index,
                                       align_corners, 1.0)
    return acc
