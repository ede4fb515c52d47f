"""One stage of the dynamic ResNet: ``depth_max`` bottlenecks, only the first ``depth_state`` run.

Host-side mirror of gaiaseg/models/utils/dynamic_res_layer.py:16-172 (same constructor, states and
manipulate_* methods); the blocks execute through HIP kernels (core/bricks.py).
"""
import warnings

import torch.nn as nn

from ...core.bricks import DynamicBottleneck, build_conv_layer, build_norm_layer
from ...core.dynamic import DynamicMixin
from ...hip.runtime import tape_function


class DynamicResLayer(nn.ModuleList, DynamicMixin):
    search_space = {"depth", "width"}

    def init_state(self, depth=None, width=None, **kwargs):
        # reference: dynamic_res_layer.py:36-44 (it stores `depth` into width_state by mistake,
        # :41; the value is never read before manipulate_width overwrites it — we store width)
        if depth is not None:
            self.depth_state = depth
        if width is not None:
            self.width_state = width
        for k, v in kwargs.items():
            setattr(self, "%s_state" % k, v)

    def __init__(self, block, inplanes, planes, depth, stride=1, dilation=1, avg_down=False,
                 conv_cfg=None, norm_cfg=None, downsample_first=True, contract_dilation=False,
                 **kwargs):
        # reference: dynamic_res_layer.py:60-63
        if conv_cfg is None or conv_cfg.get("type") != "DynConv2d":
            warnings.warn("Non-dynamic-conv detected in dynamic block.")
        if norm_cfg is None or "Dyn" not in norm_cfg.get("type", ""):
            warnings.warn("Non-dynamic-bn detected in dynamic block.")
        self.block = block
        self.avg_down = avg_down
        self.init_state(depth=depth, width=planes)

        downsample = self._shortcut(block, inplanes, planes, stride, avg_down, conv_cfg, norm_cfg)

        # contract_dilation: dynamic_res_layer.py:98-102
        first_dilation = dilation // 2 if (dilation > 1 and contract_dilation) else dilation
        if not downsample_first:
            raise AssertionError("downsample_first=False (Hourglass) is disabled in the reference "
                                 "(dynamic_res_layer.py:128)")
        layers = [block(inplanes=inplanes, planes=planes, stride=stride, dilation=first_dilation,
                        downsample=downsample, conv_cfg=conv_cfg, norm_cfg=norm_cfg, **kwargs)]
        inplanes = planes * block.expansion
        for _ in range(1, depth):
            layers.append(block(inplanes=inplanes, planes=planes, stride=1, dilation=dilation,
                                conv_cfg=conv_cfg, norm_cfg=norm_cfg, **kwargs))
        super().__init__(layers)

    @staticmethod
    def _shortcut(block, inplanes, planes, stride, avg_down, conv_cfg, norm_cfg):
        """Projection shortcut of the first block (dynamic_res_layer.py:70-94): needed when the block
        changes resolution or width.  avg_down moves the stride into an AvgPool2d(ceil_mode,
        count_include_pad=False) in front of a stride-1 1x1 conv."""
        out_planes = planes * block.expansion
        if stride == 1 and inplanes == out_planes:
            return None
        mods = []
        if avg_down:
            mods.append(nn.AvgPool2d(kernel_size=stride, stride=stride, ceil_mode=True,
                                     count_include_pad=False))
        mods.append(build_conv_layer(conv_cfg, inplanes, out_planes, kernel_size=1, padding=0,
                                     stride=1 if avg_down else stride, bias=False))
        mods.append(build_norm_layer(norm_cfg, out_planes)[1])
        return nn.Sequential(*mods)

    def manipulate_depth(self, depth):
        assert depth >= 1, "Depth must be greater than 0, skipping stage is not supported yet."
        if depth > len(self):
            raise ValueError("depth %d exceeds the %d blocks of this stage" % (depth, len(self)))
        self.depth_state = depth

    def manipulate_width(self, width):
        self.width_state = width
        for m in self:  # fan-out to every block, inactive ones included (dynamic_res_layer.py:154-157)
            m.manipulate_width(width)

    def forward_act(self, tape, x):
        if getattr(self, "_deploying", False):
            del self[self.depth_state:]  # deploy_forward, dynamic_res_layer.py:159-164
        for i in range(self.depth_state):
            x = self[i].forward_act(tape, x)
        return x

    def forward(self, x):
        needs = any(p.requires_grad for p in self.parameters())
        return tape_function(lambda tape, acts: [self.forward_act(tape, acts[0])], [x], needs)[0]

    def active_blocks(self):
        return [self[i] for i in range(self.depth_state)]
