"""DynamicFCNHead — host-side mirror of gaiaseg/models/decode_heads/dynamic_fcn_head.py:23-231
(same registered name / constructor; base-class helpers fcn_head.py:139-253)."""
import torch.nn as nn

from ...core.bricks import DynamicConvModule
from ...hip import ops
from ...hip.runtime import Act
from ..builder import HEADS
from .decode_head import DynamicBaseDecodeHead


@HEADS.register_module()
class DynamicFCNHead(DynamicBaseDecodeHead):
    def __init__(self, in_channels, channels, num_classes, num_convs=2, kernel_size=3,
                 concat_input=True, dropout_ratio=0.1, conv_cfg=None, norm_cfg=None,
                 act_cfg=dict(type="ReLU"), in_index=-1, input_transform=None,
                 loss_decode=dict(type="CrossEntropyLoss", use_sigmoid=False, loss_weight=1.0),
                 ignore_index=255, sampler=None, align_corners=False):
        super().__init__(in_channels, channels, num_classes=num_classes,
                         dropout_ratio=dropout_ratio, conv_cfg=conv_cfg, norm_cfg=norm_cfg,
                         act_cfg=act_cfg, in_index=in_index, input_transform=input_transform,
                         loss_decode=loss_decode, ignore_index=ignore_index, sampler=sampler,
                         align_corners=align_corners)
        assert num_convs >= 0
        self.num_convs, self.concat_input, self.kernel_size = num_convs, concat_input, kernel_size
        if num_convs == 0:
            assert self.in_channels == self.channels
        convs = []
        for i in range(num_convs):  # dynamic_fcn_head.py:91-112
            convs.append(DynamicConvModule(
                self.in_channels if i == 0 else self.channels, self.channels,
                kernel_size=kernel_size, padding=kernel_size // 2, conv_cfg=self.conv_cfg,
                norm_cfg=self.norm_cfg, act_cfg=self.act_cfg))
        self.convs = nn.Identity() if num_convs == 0 else nn.Sequential(*convs)
        if self.concat_input:  # dynamic_fcn_head.py:117-126
            self.conv_cat = DynamicConvModule(
                self.in_channels + self.channels, self.channels, kernel_size=kernel_size,
                padding=kernel_size // 2, conv_cfg=self.conv_cfg, norm_cfg=self.norm_cfg,
                act_cfg=self.act_cfg)

    def forward_acts(self, tape, x):
        # dynamic_fcn_head.py:128-135; torch.cat([x, output]) is fused: the last conv's BN/ReLU
        # writes straight into the channel slice of the concat buffer
        mods = [] if self.num_convs == 0 else list(self.convs)
        if self.concat_input:
            cat = Act.empty(x.N, x.H, x.W, x.C + self.channels, x.t.device)
            ops.copy_into(tape, x, cat.slice(0, x.C))
            out = x
            for i, m in enumerate(mods):
                last = i == len(mods) - 1
                out = m.forward_act(tape, out, out=cat.slice(x.C, x.C + self.channels) if last else None)
            if not mods:
                ops.copy_into(tape, x, cat.slice(x.C, x.C + self.channels))
            out = self.conv_cat.forward_act(tape, cat)
        else:
            out = x
            for m in mods:
                out = m.forward_act(tape, out)
        return self.cls_seg_act(tape, out)
