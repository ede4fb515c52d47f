"""DynamicPSPHead / DynamicPPM — host-side mirror of
gaiaseg/models/decode_heads/dynamic_psp_head.py:25-147 and the inherited PSPHead.forward
(gaiaseg/models/decode_heads/psp_head.py:228-241)."""
import torch.nn as nn

from ...core.bricks import DynamicConvModule
from ...hip import ops
from ...hip.runtime import Act
from ..builder import HEADS
from .decode_head import DynamicBaseDecodeHead


class DynamicPPM(nn.ModuleList):
    """Pooling Pyramid Module: for each scale AdaptiveAvgPool2d(s) -> 1x1 DynConvModule -> bilinear
    up to the input size (dynamic_psp_head.py:48-73).  All scales are pooled in one read of x and
    every upsampled branch is written straight into its slice of the concat buffer."""

    def __init__(self, pool_scales, in_channels, channels, conv_cfg, norm_cfg, act_cfg,
                 align_corners):
        super().__init__()
        self.pool_scales = pool_scales
        self.align_corners = align_corners
        self.in_channels, self.channels = in_channels, channels
        self.conv_cfg, self.norm_cfg, self.act_cfg = conv_cfg, norm_cfg, act_cfg
        for pool_scale in pool_scales:
            # index 0 of each Sequential is the parameter-free pool: state_dict keys stay
            # psp_modules.{i}.1.conv.weight
            self.append(nn.Sequential(
                nn.AdaptiveAvgPool2d(pool_scale),
                DynamicConvModule(self.in_channels, self.channels, 1, conv_cfg=self.conv_cfg,
                                  norm_cfg=self.norm_cfg, act_cfg=self.act_cfg)))

    def forward_acts(self, tape, x, outs=None):
        """outs: optional list of destination Acts (concat slices), one per scale."""
        pooled = ops.adaptive_avgpool(tape, x, list(self.pool_scales))
        results = []
        for i, (ppm, p) in enumerate(zip(self, pooled)):
            y = ppm[1].forward_act(tape, p)
            results.append(ops.bilinear(tape, y, (x.H, x.W), self.align_corners,
                                        out=None if outs is None else outs[i]))
        return results


def psp_concat(tape, ppm, x, channels):
    """cat([x] + ppm(x), dim=1) without materialising the pieces (psp_head.py:231-238)."""
    ns = len(ppm.pool_scales)
    cat = Act.empty(x.N, x.H, x.W, x.C + ns * channels, x.t.device)
    ops.copy_into(tape, x, cat.slice(0, x.C))
    outs = [cat.slice(x.C + i * channels, x.C + (i + 1) * channels) for i in range(ns)]
    ppm.forward_acts(tape, x, outs)
    return cat


@HEADS.register_module()
class DynamicPSPHead(DynamicBaseDecodeHead):
    def __init__(self, in_channels, channels, num_classes, pool_scales=(1, 2, 3, 6),
                 dropout_ratio=0.1, conv_cfg=None, norm_cfg=None, act_cfg=dict(type="ReLU"),
                 in_index=-1, input_transform=None,
                 loss_decode=dict(type="CrossEntropyLoss", use_sigmoid=False, loss_weight=1.0),
                 ignore_index=255, sampler=None, align_corners=False, keep_resize_logit=False):
        super().__init__(in_channels, channels, num_classes=num_classes,
                         dropout_ratio=dropout_ratio, conv_cfg=conv_cfg, norm_cfg=norm_cfg,
                         act_cfg=act_cfg, in_index=in_index, input_transform=input_transform,
                         loss_decode=loss_decode, ignore_index=ignore_index, sampler=sampler,
                         align_corners=align_corners)
        assert isinstance(pool_scales, (list, tuple))
        self.pool_scales = pool_scales
        self.psp_modules = DynamicPPM(self.pool_scales, self.in_channels, self.channels,
                                      conv_cfg=self.conv_cfg, norm_cfg=self.norm_cfg,
                                      act_cfg=self.act_cfg, align_corners=self.align_corners)
        self.bottleneck = DynamicConvModule(
            self.in_channels + len(pool_scales) * self.channels, self.channels, 3, padding=1,
            conv_cfg=self.conv_cfg, norm_cfg=self.norm_cfg, act_cfg=self.act_cfg)
        # The reference also puts the full-resolution logits into the loss dict
        # (dynamic_psp_head.py:160, 'resize_logit'); it never reaches the gradient (key lacks
        # 'loss') but costs a mean + all-reduce of an 80 MB tensor per step.  Off by default
        # (SURVEY.md Appendix D5); keep_resize_logit=True restores the key.
        self.keep_resize_logit = keep_resize_logit

    def forward_acts(self, tape, x):
        cat = psp_concat(tape, self.psp_modules, x, self.channels)
        out = self.bottleneck.forward_act(tape, cat)
        return self.cls_seg_act(tape, out)

    def losses(self, seg_logit, seg_label):
        loss = super().losses(seg_logit, seg_label)
        if self.keep_resize_logit:
            # the full-resolution logits of dynamic_psp_head.py:152-160, through the HIP resize kernel
            # (gs_bilinear_forward) like every other resize of the path; a log variable only
            from ...core.inference import _padded_nhwc
            from ...hip import ops
            from ...hip.runtime import Act, Tape
            c = seg_logit.shape[1]
            h, w = seg_label.shape[2:]
            src = Act(_padded_nhwc(seg_logit), False)      # class stride padded to a float4 multiple
            up = ops.bilinear(Tape(enabled=False), src, (int(h), int(w)), self.align_corners)
            loss["resize_logit"] = up.as_nchw()[:, :c]
        return loss
