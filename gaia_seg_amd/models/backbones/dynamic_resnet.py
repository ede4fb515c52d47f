"""DynamicResNet backbone — host-side mirror of gaiaseg/models/backbones/dynamic_resnet.py:25-421.

Same registered name, constructor signature, states, ``manipulate_stem`` / ``manipulate_body``,
init, freezing and ``train()`` behaviour.  ``forward`` returns logical NCHW feature maps (stored
channels-last) and runs the whole backbone as ONE autograd node over hand-written HIP kernels.
"""
from collections.abc import Sequence

import torch.nn as nn
from torch.nn.modules.batchnorm import _BatchNorm

from ...core.bricks import (DynamicBatchNorm2d, DynamicBottleneck, DynamicConv2d,
                            build_conv_layer, build_norm_layer, constant_init, conv_bn_act, kaiming_init)
from ...core.dynamic import DynamicMixin, freeze, unzip_meta
from ...hip import ops
from ...hip.runtime import tape_function
from ..builder import BACKBONES
from ..utils import DynamicResLayer


@BACKBONES.register_module()
class DynamicResNet(nn.Module, DynamicMixin):
    search_space = {"stem", "body"}

    def init_state(self, stem=None, body=None, **kwargs):
        if stem is not None:
            self.stem_state = stem
        if body is not None:
            self.body_state = body
        for k, v in kwargs.items():
            setattr(self, "%s_state" % k, v)

    def __init__(self, in_channels, stem_width, body_width, body_depth, num_stages=4,
                 strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1), out_indices=(0, 1, 2, 3),
                 style="pytorch", deep_stem=False, avg_down=False, frozen_stages=-1,
                 frozen_layers=None, conv_cfg=None, norm_cfg=dict(type="DynSyncBN"),
                 act_cfg=dict(type="ReLU"), norm_eval=False, dcn=None,
                 stage_with_dcn=(False, False, False, False), plugins=None, with_cp=False,
                 zero_init_residual=True, contract_dilation=False):
        super().__init__()
        # -- what the in-tree configs cannot ask for (reference asserts: dynamic_resnet.py:109-114) --
        if not 1 <= num_stages <= 4:
            raise AssertionError("num_stages must be in 1..4")
        if not (len(strides) == len(dilations) == num_stages) or max(out_indices) >= num_stages:
            raise AssertionError("strides / dilations / out_indices do not match num_stages")
        if dcn is not None or plugins is not None:
            raise NotImplementedError("dcn / plugins are not used by the in-tree seg configs")
        # -- the constructor arguments stay readable under their own names (tools and hooks read them) --
        for name, value in dict(
                stem_width=stem_width, body_width=body_width, num_stages=num_stages, strides=strides,
                dilations=dilations, out_indices=out_indices, style=style, deep_stem=deep_stem,
                avg_down=avg_down, frozen_stages=frozen_stages, frozen_layers=frozen_layers,
                conv_cfg=conv_cfg, norm_cfg=norm_cfg, act_cfg=act_cfg, with_cp=with_cp,
                norm_eval=norm_eval, dcn=dcn, stage_with_dcn=stage_with_dcn, plugins=plugins,
                zero_init_residual=zero_init_residual, contract_dilation=contract_dilation).items():
            setattr(self, name, value)
        self.block = DynamicBottleneck  # the reference has no BasicBlock (dynamic_resnet.py:132-133)
        self.body_depth = list(body_depth[:num_stages])
        self.init_state(stem={"width": stem_width}, body={"depth": body_depth, "width": body_width})

        # -- modules, all at their MAXIMUM size; names fix the state_dict keys --
        self._make_stem_layer(in_channels, stem_width)
        width_in = stem_width[-1] if deep_stem else stem_width
        self.res_layers = []
        for stage, (depth, planes) in enumerate(zip(self.body_depth, body_width), start=1):
            self.res_layers.append("layer%d" % stage)
            self.add_module(self.res_layers[-1], self.make_res_layer(
                block=self.block, inplanes=width_in, planes=planes, depth=depth,
                stride=strides[stage - 1], dilation=dilations[stage - 1], style=style,
                avg_down=avg_down, with_cp=with_cp, conv_cfg=conv_cfg, norm_cfg=norm_cfg, dcn=None,
                contract_dilation=contract_dilation, plugins=None))
            width_in = planes * self.block.expansion
        self.inplanes = width_in
        self.feat_dim = self.active_feat_dim = \
            self.block.expansion * body_width[0] * 2 ** (len(self.body_depth) - 1)
        self._freeze_stages()

    def make_res_layer(self, **kwargs):
        return DynamicResLayer(**kwargs)

    @property
    def norm1(self):
        return getattr(self, self.norm1_name)

    def _make_stem_layer(self, in_channels, stem_width):
        # Stem (dynamic_resnet.py:255-302).  Deep stem: three 3x3 conv + norm + ReLU triples in ONE
        # Sequential, so the convs sit at indices 0 / 3 / 6 and the norms at 1 / 4 / 7 (the indices
        # manipulate_stem relies on and the checkpoint keys "stem.0.weight" ...); else the 7x7 conv.
        if self.deep_stem:
            assert isinstance(stem_width, Sequence) and len(stem_width) == 3
            mods, cin = [], in_channels
            for i, cout in enumerate(stem_width):
                mods += [build_conv_layer(self.conv_cfg, cin, cout, kernel_size=3,
                                          stride=2 if i == 0 else 1, padding=1, bias=False),
                         build_norm_layer(self.norm_cfg, cout)[1], nn.ReLU(inplace=True)]
                cin = cout
            self.stem = nn.Sequential(*mods)
        else:
            self.conv1 = build_conv_layer(self.conv_cfg, in_channels, stem_width, kernel_size=7,
                                          stride=2, padding=3, bias=False)
            self.norm1_name, norm1 = build_norm_layer(self.norm_cfg, stem_width, postfix=1)
            self.add_module(self.norm1_name, norm1)
            self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)

    # ---- freezing (frozen_stages / frozen_layers, dynamic_resnet.py:304-334) ----
    def _frozen_modules(self):
        """frozen_stages = k >= 0 freezes the stem and stages 1..k; frozen_layers[i] = m freezes the
        first m blocks of stage i."""
        mods = []
        if self.frozen_stages >= 0:
            mods += [self.stem] if self.deep_stem else [self.conv1, self.norm1]
            mods += [getattr(self, name) for name in self.res_layers[:self.frozen_stages]]
        if self.frozen_layers is not None:
            for name, count in zip(self.res_layers, self.frozen_layers):
                layer = getattr(self, name)
                if count > len(layer):
                    raise ValueError("frozen_layers asks for %d blocks of %s, which has %d"
                                     % (count, name, len(layer)))
                mods += list(layer)[:count]   # (ModuleList slicing would re-run the constructor)
        return mods

    def _freeze_stages(self):
        for m in self._frozen_modules():
            freeze(m)
        if self.frozen_stages >= 0 and not self.deep_stem:
            # the reference puts only norm1 into eval mode here (dynamic_resnet.py:311-315); a conv has
            # no mode-dependent behaviour, the flag is mirrored so module.training reads the same
            self.conv1.training = self.training

    _freeze_layers = _freeze_stages   # one pass covers both options (kept for API parity)

    def init_weights(self, pretrained=None):
        """He-normal convs, unit norms, zero last norm of every bottleneck (zero_init_residual), or a
        checkpoint path (dynamic_resnet.py:336-367)."""
        if isinstance(pretrained, str):
            from ...core.checkpoint import load_checkpoint
            load_checkpoint(self, pretrained, strict=False)
            return
        if pretrained is not None:
            raise TypeError("pretrained must be a str or None")
        last_norms = {id(m.norm3) for m in self.modules()
                      if isinstance(m, DynamicBottleneck)} if self.zero_init_residual else set()
        for m in self.modules():
            if isinstance(m, DynamicConv2d):
                kaiming_init(m)
            elif isinstance(m, (_BatchNorm, nn.GroupNorm)):
                constant_init(m, 0 if id(m) in last_norms else 1)

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:   # BatchNorm keeps its running statistics while training
            for m in self.modules():
                if isinstance(m, _BatchNorm):
                    m.eval()
        return self

    # ---- arch manipulation (dynamic_resnet.py:381-403) ----
    def _stem_convs(self):
        return [self.stem[0], self.stem[3], self.stem[6]] if self.deep_stem else [self.conv1]

    def manipulate_stem(self, arch_meta):
        """{'width': 32}, or {'width': [16, 16, 32]} for the three convs of a deep stem."""
        self.stem_state = arch_meta
        metas = unzip_meta(arch_meta) if self.deep_stem else [arch_meta]
        for conv, meta in zip(self._stem_convs(), metas):
            conv.manipulate_arch(meta)

    def manipulate_body(self, arch_meta):
        """{'width': [w1..w4], 'depth': [d1..d4]}: one entry per stage."""
        self.body_state = arch_meta
        for name, meta in zip(self.res_layers, unzip_meta(arch_meta)):
            getattr(self, name).manipulate_arch(meta)

    # ---- execution ----
    def forward_act(self, tape, x):
        if self.deep_stem:
            mods = list(self.stem)
            for i in range(0, len(mods), 3):  # conv, norm, relu triples
                x = conv_bn_act(tape, mods[i], mods[i + 1], x, relu=True)
        else:
            # (one library call: the stem kernel's epilogue leaves the BatchNorm tile partials, so the
            # separate statistics pass over the 67 MB output is gone — csrc/stem.hip)
            x = conv_bn_act(tape, self.conv1, self.norm1, x, relu=True)
        mp = self.maxpool
        x = ops.maxpool(tape, x, mp.kernel_size, mp.stride, mp.padding)
        outs = []
        fork_stage = self.__dict__.get("aux_fork_stage")   # set by the segmentor (auxiliary heads)
        self.__dict__["_aux_forked"] = False
        last = len(self.res_layers) - 1
        for i, layer_name in enumerate(self.res_layers):
            x = getattr(self, layer_name).forward_act(tape, x)
            if i == fork_stage and i < last and tape.enabled and ops.BRANCH_AUX and ops.AUX_PREFORK:
                # the auxiliary heads' inputs are complete: their branch stream starts from here
                ops.prefork_branch(x.t.device, ops.SLOT_AUX)
                self.__dict__["_aux_forked"] = True
            if i == 1 and last >= 2:
                # backward crosses this point when stages 3.. (and the heads) are done: their
                # parameters can be updated while backward goes on (runner: early optimizer step)
                ops.backward_mark(tape, "stage2|stage3")
            if i == 0:
                # backward crosses this point last-but-one: parameters of every later layer can be
                # updated while the side stream finishes the stem / stage-1 weight gradients
                ops.side_checkpoint(tape)
            if i in self.out_indices:
                outs.append(x)
        return outs

    def late_gradient_parameters(self):
        """Parameters whose weight gradients are produced after the side-stream checkpoint (the
        stem and stage 1): the optimizer updates them last."""
        mods = [self.stem] if self.deep_stem else [self.conv1, self.norm1]
        mods.append(getattr(self, self.res_layers[0]))
        return [p for m in mods for p in m.parameters()]

    def early_gradient_parameters(self):
        """Parameters whose gradients are final when backward crosses the "stage2|stage3" mark:
        stages 3 and later (the segmentor adds its heads)."""
        return [p for name in self.res_layers[2:] for p in getattr(self, name).parameters()]

    def forward(self, x):
        needs = any(p.requires_grad for p in self.parameters())
        return tuple(tape_function(lambda tape, acts: self.forward_act(tape, acts[0]), [x], needs))

    def active_modules(self):
        """Modules whose parameters take part in the current subnet (depth-skipped blocks are
        'unused parameters': no gradient, no optimizer update — SURVEY.md Appendix A13)."""
        mods = [self.stem] if self.deep_stem else [self.conv1, self.norm1]
        for layer_name in self.res_layers:
            mods.extend(getattr(self, layer_name).active_blocks())
        return mods
