"""DynamicUPerHead — host-side mirror of gaiaseg/models/decode_heads/dynamic_uper_head.py:16-131."""
import torch.nn as nn

from ...core.bricks import DynamicConvModule
from ...hip import ops
from ...hip.runtime import Act
from ..builder import HEADS
from .decode_head import DynamicBaseDecodeHead
from .dynamic_psp_head import DynamicPPM, psp_concat


@HEADS.register_module()
class DynamicUPerHead(DynamicBaseDecodeHead):
    def __init__(self, pool_scales=(1, 2, 3, 6), **kwargs):
        kwargs.pop("input_transform", None)
        in_channels = kwargs.pop("in_channels")
        channels = kwargs.pop("channels")
        super().__init__(in_channels, channels, input_transform="multiple_select", **kwargs)
        self.psp_modules = DynamicPPM(pool_scales, self.in_channels[-1], self.channels,
                                      conv_cfg=self.conv_cfg, norm_cfg=self.norm_cfg,
                                      act_cfg=self.act_cfg, align_corners=self.align_corners)
        self.bottleneck = DynamicConvModule(
            self.in_channels[-1] + len(pool_scales) * self.channels, self.channels, 3, padding=1,
            conv_cfg=self.conv_cfg, norm_cfg=self.norm_cfg, act_cfg=self.act_cfg)
        self.lateral_convs = nn.ModuleList()
        self.fpn_convs = nn.ModuleList()
        for in_ch in self.in_channels[:-1]:  # skip the top layer (dynamic_uper_head.py:51-70)
            self.lateral_convs.append(DynamicConvModule(
                in_ch, self.channels, 1, conv_cfg=self.conv_cfg, norm_cfg=self.norm_cfg,
                act_cfg=self.act_cfg, inplace=False))
            self.fpn_convs.append(DynamicConvModule(
                self.channels, self.channels, 3, padding=1, conv_cfg=self.conv_cfg,
                norm_cfg=self.norm_cfg, act_cfg=self.act_cfg, inplace=False))
        self.fpn_bottleneck = DynamicConvModule(
            len(self.in_channels) * self.channels, self.channels, 3, padding=1,
            conv_cfg=self.conv_cfg, norm_cfg=self.norm_cfg, act_cfg=self.act_cfg)

    def psp_forward_act(self, tape, x):
        cat = psp_concat(tape, self.psp_modules, x, self.channels)  # dynamic_uper_head.py:81-89
        return self.bottleneck.forward_act(tape, cat)

    def forward_acts(self, tape, inputs):
        # laterals (dynamic_uper_head.py:97-102)
        laterals = [lc.forward_act(tape, inputs[i]) for i, lc in enumerate(self.lateral_convs)]
        laterals.append(self.psp_forward_act(tape, inputs[-1]))
        n = len(laterals)
        # top-down path: laterals[i-1] += resize(laterals[i])  (:104-112) — one fused
        # resize-add kernel per level, in place on the finer map
        for i in range(n - 1, 0, -1):
            fine = laterals[i - 1]
            ops.bilinear(tape, laterals[i], (fine.H, fine.W), self.align_corners, out=fine,
                         accumulate=True)
        # fpn convs on levels 0..n-2 (:115-120); resize all to level 0 and concat (:122-128):
        # every branch is produced directly inside its slice of the concat buffer
        l0 = laterals[0]
        cat = Act.empty(l0.N, l0.H, l0.W, n * self.channels, l0.t.device)
        ch = self.channels
        for i in range(n - 1):
            if i == 0:
                self.fpn_convs[0].forward_act(tape, laterals[0], out=cat.slice(0, ch))
            else:
                y = self.fpn_convs[i].forward_act(tape, laterals[i])
                ops.bilinear(tape, y, (l0.H, l0.W), self.align_corners,
                             out=cat.slice(i * ch, (i + 1) * ch))
        ops.bilinear(tape, laterals[-1], (l0.H, l0.W), self.align_corners,
                     out=cat.slice((n - 1) * ch, n * ch))
        out = self.fpn_bottleneck.forward_act(tape, cat)
        return self.cls_seg_act(tape, out)
