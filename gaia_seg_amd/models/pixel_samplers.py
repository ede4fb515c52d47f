"""OHEMPixelSampler with mmseg's contract (``build_pixel_sampler(sampler, context=head)``,
``sample(seg_logit, seg_label) -> seg_weight``; call sites
gaiaseg/models/decode_heads/dynamic_fcn_head.py:70-71,147-148; semantics SURVEY.md Appendix A11).

The sampler receives the LOW-resolution logits: the resize to the label size is fused into the
probability kernel (``gs_ce_label_prob``), and the sorted-probability threshold is an exact radix
select on the device (``gs_ohem_weights``) — no [N,C,H,W] tensor and no sort are materialised."""
import ctypes

import torch

from ..hip import lib as _lib
from ..hip.runtime import WORKSPACE, current_stream_ptr, require_gpu_tensor
from .builder import PIXEL_SAMPLERS
from .losses.cross_entropy_loss import _ce_desc


@PIXEL_SAMPLERS.register_module()
class OHEMPixelSampler:
    def __init__(self, context, thresh=None, min_kept=100000):
        assert min_kept > 1
        self.context = context
        self.thresh = thresh
        self.min_kept = min_kept

    def sample(self, seg_logit, seg_label):
        """seg_logit [N,C,h,w] (any resolution), seg_label [N,1,H,W] -> float weights [N,H,W]."""
        with torch.no_grad():
            require_gpu_tensor(seg_logit, "seg_logit")
            assert seg_label.shape[1] == 1
            L = _lib.load()
            label = seg_label.squeeze(1).contiguous()
            if label.dtype != torch.int64:
                label = label.long()
            n, hh, ww = label.shape
            d = _ce_desc(seg_logit.detach(), (hh, ww), self.context.ignore_index,
                         self.context.align_corners)
            dev = seg_logit.device
            prob = torch.empty((n, hh, ww), dtype=torch.float32, device=dev)
            st = current_stream_ptr()
            _lib.check(L.gs_ce_label_prob(ctypes.byref(d), seg_logit.data_ptr(), label.data_ptr(),
                                          prob.data_ptr(), st), "gs_ce_label_prob")
            weight = torch.empty_like(prob)
            ws = WORKSPACE.get(L.gs_ohem_workspace_bytes(), dev)
            batch_kept = self.min_kept * n
            _lib.check(L.gs_ohem_weights(prob.data_ptr(), prob.numel(), batch_kept,
                                         float(self.thresh) if self.thresh is not None else 0.0,
                                         1 if self.thresh is not None else 0, weight.data_ptr(),
                                         ws.data_ptr(), ws.numel(), st), "gs_ohem_weights")
            return weight
