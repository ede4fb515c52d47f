"""ORACLE (test infrastructure — never imported by the product path).

Plain-PyTorch CPU restatement of the supernet model path with the reference's module / state_dict
names, so a state_dict of the MI355X model loads into it unchanged and the two can be compared
on the same seeded inputs.  Wiring follows the reference files line by line:

  ODynamicResNet / OResLayer   gaiaseg/models/backbones/dynamic_resnet.py:255-302,381-421
                               gaiaseg/models/utils/dynamic_res_layer.py:70-147,149-172
  OFCNHead                     gaiaseg/models/decode_heads/dynamic_fcn_head.py:91-135 (+ fcn_head.py:179-253)
  OPPM / OPSPHead              gaiaseg/models/decode_heads/dynamic_psp_head.py:48-73,130-147; psp_head.py:228-241
  OUPerHead                    gaiaseg/models/decode_heads/dynamic_uper_head.py:28-131
  OEncoderDecoder              gaiaseg/models/segmentors/dynamic_encoder_decoder.py:9-42 and the
                               EncoderDecoder flow restated in
                               "dynamic_encoder_decoder-distill-backup (1).py":85-143
The gaiavision bricks (OConv / OBN / OBottleneck / OConvModule) follow SURVEY.md Appendix A1-A4 —
PARITY UNPINNED, see oracle/ops.py.
"""
import torch
import torch.nn as nn

from . import ops as O


def _key(bn):
    """Name of the BatchNorm a ReLU follows (every ReLU on this path follows one): the key under
    which oracle.ops.relu records / is given that ReLU's mask.  Set by OEncoderDecoder."""
    return getattr(bn, "_gs_name", None)


class OReLU(nn.Module):
    """nn.ReLU(inplace=True) of the deep stem (dynamic_resnet.py:258-288), keyed by its BN."""

    def __init__(self, bn):
        super().__init__()
        self._bn = [bn]   # not a submodule: keeps the Sequential indices / state_dict unchanged

    def forward(self, x):
        return O.relu(x, key=_key(self._bn[0]))


class OConv(nn.Conv2d):
    """DynConv2d (A1): max-size nn.Conv2d parameters, forward on the leading slice."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.width_state = self.out_channels

    def manipulate_width(self, w):
        self.width_state = w

    def forward(self, x):
        return O.dyn_conv2d(x, self.weight, self.bias, self.width_state, self.stride, self.padding,
                            self.dilation)


class OBN(nn.BatchNorm2d):
    """DynBN / DynSyncBN(group_size=1) / SyncBN at world size 1 (A2)."""

    def forward(self, x):
        if self.training and self.track_running_stats and self.num_batches_tracked is not None:
            self.num_batches_tracked += 1
        return O.dyn_batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias,
                                self.training, self.momentum, self.eps)


class OConvModule(nn.Module):
    """DynamicConvModule (A4): conv(bias = no norm) -> bn -> ReLU."""

    def __init__(self, cin, cout, k, padding=0, norm=True, act=True):
        super().__init__()
        self.conv = OConv(cin, cout, k, padding=padding, bias=not norm)
        if norm:
            self.bn = OBN(cout)
        self.with_norm, self.with_act = norm, act

    def forward(self, x):
        x = self.conv(x)
        if self.with_norm:
            x = self.bn(x)
        if self.with_act:
            x = O.relu(x, key=_key(self.bn) if self.with_norm else None)
        return x


class OBottleneck(nn.Module):
    """DynamicBottleneck (A3), style='pytorch': stride on the 3x3."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, style="pytorch"):
        super().__init__()
        s1, s2 = (1, stride) if style == "pytorch" else (stride, 1)   # 'caffe': stride on the first 1x1
        self.conv1 = OConv(inplanes, planes, 1, stride=s1, bias=False)
        self.bn1 = OBN(planes)
        self.conv2 = OConv(planes, planes, 3, stride=s2, padding=dilation, dilation=dilation,
                           bias=False)
        self.bn2 = OBN(planes)
        self.conv3 = OConv(planes, planes * 4, 1, bias=False)
        self.bn3 = OBN(planes * 4)
        self.downsample = downsample

    def manipulate_width(self, w):
        self.conv1.manipulate_width(w)
        self.conv2.manipulate_width(w)
        self.conv3.manipulate_width(4 * w)
        if self.downsample is not None:
            for m in self.downsample:
                if isinstance(m, OConv):
                    m.manipulate_width(4 * w)

    def forward(self, x):
        identity = x
        out = O.relu(self.bn1(self.conv1(x)), key=_key(self.bn1))
        out = O.relu(self.bn2(self.conv2(out)), key=_key(self.bn2))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return O.relu(out + identity, key=_key(self.bn3))


class OResLayer(nn.ModuleList):
    def __init__(self, inplanes, planes, depth, stride=1, dilation=1, contract_dilation=False,
                 avg_down=False, style="pytorch"):
        downsample = None
        if stride != 1 or inplanes != planes * 4:  # dynamic_res_layer.py:70-94
            mods, conv_stride = [], stride
            if avg_down:                            # :75-82
                conv_stride = 1
                mods.append(nn.AvgPool2d(kernel_size=stride, stride=stride, ceil_mode=True,
                                         count_include_pad=False))
            mods += [OConv(inplanes, planes * 4, 1, stride=conv_stride, bias=False), OBN(planes * 4)]
            downsample = nn.Sequential(*mods)
        first_dilation = dilation // 2 if (dilation > 1 and contract_dilation) else dilation
        layers = [OBottleneck(inplanes, planes, stride, first_dilation, downsample, style)]
        for _ in range(1, depth):
            layers.append(OBottleneck(planes * 4, planes, 1, dilation, style=style))
        super().__init__(layers)
        self.depth_state = depth

    def forward(self, x):
        for i in range(self.depth_state):  # dynamic_res_layer.py:170-172
            x = self[i](x)
        return x


class ODynamicResNet(nn.Module):
    def __init__(self, in_channels, stem_width, body_width, body_depth, strides=(1, 2, 2, 2),
                 dilations=(1, 1, 1, 1), out_indices=(0, 1, 2, 3), deep_stem=False,
                 contract_dilation=False, avg_down=False, num_stages=4, style="pytorch", **unused):
        super().__init__()
        body_depth = body_depth[:num_stages]   # dynamic_resnet.py:134
        self.deep_stem = deep_stem
        self.out_indices = out_indices
        if deep_stem:  # dynamic_resnet.py:258-288
            sw = stem_width
            b0, b1, b2 = OBN(sw[0]), OBN(sw[1]), OBN(sw[2])
            self.stem = nn.Sequential(
                OConv(in_channels, sw[0], 3, stride=2, padding=1, bias=False), b0, OReLU(b0),
                OConv(sw[0], sw[1], 3, stride=1, padding=1, bias=False), b1, OReLU(b1),
                OConv(sw[1], sw[2], 3, stride=1, padding=1, bias=False), b2, OReLU(b2))
            inplanes = sw[-1]
        else:  # :290-301
            self.conv1 = OConv(in_channels, stem_width, 7, stride=2, padding=3, bias=False)
            self.bn1 = OBN(stem_width)
            inplanes = stem_width
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.res_layers = []
        for i, depth in enumerate(body_depth):
            layer = OResLayer(inplanes, body_width[i], depth, strides[i], dilations[i],
                              contract_dilation, avg_down, style)
            inplanes = body_width[i] * 4
            name = "layer%d" % (i + 1)
            self.add_module(name, layer)
            self.res_layers.append(name)

    def manipulate_arch(self, arch):
        """arch = {'stem': {'width': ..}, 'body': {'width': [..], 'depth': [..]}} (:381-403)."""
        if "stem" in arch:
            w = arch["stem"]["width"]
            if self.deep_stem:
                for idx, wi in zip((0, 3, 6), w):
                    self.stem[idx].manipulate_width(wi)
            else:
                self.conv1.manipulate_width(w)
        if "body" in arch:
            body = arch["body"]
            for i, name in enumerate(self.res_layers):
                layer = getattr(self, name)
                if "depth" in body:
                    assert body["depth"][i] >= 1
                    layer.depth_state = body["depth"][i]
                if "width" in body:
                    for blk in layer:
                        blk.manipulate_width(body["width"][i])

    def forward(self, x):
        if self.deep_stem:
            x = self.stem(x)
        else:
            x = O.relu(self.bn1(self.conv1(x)), key=_key(self.bn1))
        mp = self.maxpool   # dynamic_resnet.py:302 MaxPool2d(3, 2, 1)
        x = O.max_pool2d(x, mp.kernel_size, mp.stride, mp.padding, key="backbone.maxpool")
        outs = []
        for i, name in enumerate(self.res_layers):
            x = getattr(self, name)(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)


class _OHeadBase(nn.Module):
    def __init__(self, channels, num_classes, in_index, dropout_ratio, loss_weight, ignore_index,
                 align_corners):
        super().__init__()
        self.in_index = in_index
        self.conv_seg = OConv(channels, num_classes, 1)
        self.dropout = nn.Dropout2d(dropout_ratio) if dropout_ratio > 0 else None
        self.loss_weight, self.ignore_index, self.align_corners = loss_weight, ignore_index, align_corners
        self.sampler = None
        self.class_weight = None

    def cls_seg(self, feat):  # fcn_head.py:248-253
        if self.dropout is not None:
            feat = self.dropout(feat)
        return self.conv_seg(feat)

    def losses(self, seg_logit, seg_label):
        return O.seg_losses(seg_logit, seg_label, self.loss_weight, self.ignore_index,
                            self.align_corners, self.sampler, self.class_weight)

    def forward_train(self, inputs, gt):
        return self.losses(self.forward(inputs), gt)


class OFCNHead(_OHeadBase):
    def __init__(self, in_channels, channels, num_classes, num_convs=2, kernel_size=3,
                 concat_input=True, dropout_ratio=0.1, in_index=-1, loss_weight=1.0,
                 ignore_index=255, align_corners=False, input_transform=None, **unused):
        super().__init__(channels, num_classes, in_index, dropout_ratio, loss_weight, ignore_index,
                         align_corners)
        self.input_transform = input_transform
        if input_transform == "resize_concat":     # fcn_head.py:139-173
            in_channels = sum(in_channels)
        convs = [OConvModule(in_channels if i == 0 else channels, channels, kernel_size,
                             padding=kernel_size // 2) for i in range(num_convs)]
        self.convs = nn.Identity() if num_convs == 0 else nn.Sequential(*convs)
        self.concat_input = concat_input
        if concat_input:
            self.conv_cat = OConvModule(in_channels + channels, channels, kernel_size,
                                        padding=kernel_size // 2)

    def forward(self, inputs):  # dynamic_fcn_head.py:128-135
        if self.input_transform == "resize_concat":   # fcn_head.py:187-195
            sel = [inputs[i] for i in self.in_index]
            x = torch.cat([O.resize(t, size=sel[0].shape[2:], mode="bilinear",
                                    align_corners=self.align_corners) for t in sel], dim=1)
        else:
            x = inputs[self.in_index]
        output = self.convs(x)
        if self.concat_input:
            output = self.conv_cat(torch.cat([x, output], dim=1))
        return self.cls_seg(output)


class OPPM(nn.ModuleList):
    def __init__(self, pool_scales, in_channels, channels, align_corners):
        super().__init__()
        self.align_corners = align_corners
        for s in pool_scales:  # dynamic_psp_head.py:48-59
            self.append(nn.Sequential(nn.AdaptiveAvgPool2d(s), OConvModule(in_channels, channels, 1)))

    def forward(self, x):  # dynamic_psp_head.py:62-73
        outs = []
        for ppm in self:
            outs.append(O.resize(ppm(x), size=x.size()[2:], mode="bilinear",
                                 align_corners=self.align_corners))
        return outs


class OPSPHead(_OHeadBase):
    def __init__(self, in_channels, channels, num_classes, pool_scales=(1, 2, 3, 6),
                 dropout_ratio=0.1, in_index=-1, loss_weight=1.0, ignore_index=255,
                 align_corners=False, **unused):
        super().__init__(channels, num_classes, in_index, dropout_ratio, loss_weight, ignore_index,
                         align_corners)
        self.psp_modules = OPPM(pool_scales, in_channels, channels, align_corners)
        self.bottleneck = OConvModule(in_channels + len(pool_scales) * channels, channels, 3,
                                      padding=1)

    def forward(self, inputs):  # psp_head.py:228-241
        x = inputs[self.in_index]
        psp_outs = [x]
        psp_outs.extend(self.psp_modules(x))
        return self.cls_seg(self.bottleneck(torch.cat(psp_outs, dim=1)))


class OUPerHead(_OHeadBase):
    def __init__(self, in_channels, channels, num_classes, in_index, pool_scales=(1, 2, 3, 6),
                 dropout_ratio=0.1, loss_weight=1.0, ignore_index=255, align_corners=False,
                 **unused):
        super().__init__(channels, num_classes, in_index, dropout_ratio, loss_weight, ignore_index,
                         align_corners)
        self.psp_modules = OPPM(pool_scales, in_channels[-1], channels, align_corners)
        self.bottleneck = OConvModule(in_channels[-1] + len(pool_scales) * channels, channels, 3,
                                      padding=1)
        self.lateral_convs = nn.ModuleList()
        self.fpn_convs = nn.ModuleList()
        for c in in_channels[:-1]:
            self.lateral_convs.append(OConvModule(c, channels, 1))
            self.fpn_convs.append(OConvModule(channels, channels, 3, padding=1))
        self.fpn_bottleneck = OConvModule(len(in_channels) * channels, channels, 3, padding=1)

    def forward(self, inputs):  # dynamic_uper_head.py:91-131
        inputs = [inputs[i] for i in self.in_index]
        laterals = [lc(inputs[i]) for i, lc in enumerate(self.lateral_convs)]
        psp_outs = [inputs[-1]]
        psp_outs.extend(self.psp_modules(inputs[-1]))
        laterals.append(self.bottleneck(torch.cat(psp_outs, dim=1)))
        n = len(laterals)
        for i in range(n - 1, 0, -1):
            prev_shape = laterals[i - 1].shape[2:]
            # out-of-place form of `laterals[i-1] += resize(...)` (:108): same values; autograd
            # needs the pre-add ReLU output intact
            laterals[i - 1] = laterals[i - 1] + O.resize(laterals[i], size=prev_shape,
                                                         mode="bilinear",
                                                         align_corners=self.align_corners)
        fpn_outs = [self.fpn_convs[i](laterals[i]) for i in range(n - 1)]
        fpn_outs.append(laterals[-1])
        for i in range(n - 1, 0, -1):
            fpn_outs[i] = O.resize(fpn_outs[i], size=fpn_outs[0].shape[2:], mode="bilinear",
                                   align_corners=self.align_corners)
        return self.cls_seg(self.fpn_bottleneck(torch.cat(fpn_outs, dim=1)))


_HEADS = {"DynamicFCNHead": OFCNHead, "DynamicPSPHead": OPSPHead, "DynamicUPerHead": OUPerHead}


def _build_head(cfg):
    cfg = dict(cfg)
    cls = _HEADS[cfg.pop("type")]
    loss = cfg.pop("loss_decode", dict(loss_weight=1.0))
    cfg.pop("conv_cfg", None)
    cfg.pop("norm_cfg", None)
    cfg.pop("act_cfg", None)
    return cls(loss_weight=loss.get("loss_weight", 1.0), **cfg)


class OEncoderDecoder(nn.Module):
    """EncoderDecoder.forward_train + loss aggregation (A12)."""

    def __init__(self, backbone, decode_head, auxiliary_head=None, **unused):
        super().__init__()
        b = dict(backbone)
        b.pop("type", None)
        for k in ("conv_cfg", "norm_cfg", "style", "num_stages"):
            b.pop(k, None)
        self.backbone = ODynamicResNet(**b)
        self.decode_head = _build_head(decode_head)
        self.auxiliary_head = _build_head(auxiliary_head) if auxiliary_head is not None else None
        for name, m in self.named_modules():
            if isinstance(m, OBN):
                m._gs_name = name

    def manipulate_arch(self, arch):
        if "backbone" in arch:
            self.backbone.manipulate_arch(arch["backbone"])

    def forward_train(self, img, gt):
        x = self.backbone(img)
        losses = {"decode." + k: v for k, v in self.decode_head.forward_train(x, gt).items()}
        if self.auxiliary_head is not None:
            losses.update({"aux." + k: v for k, v in self.auxiliary_head.forward_train(x, gt).items()})
        return losses

    @staticmethod
    def parse_losses(losses):
        log_vars = {k: v.mean() for k, v in losses.items()}
        loss = sum(v for k, v in log_vars.items() if "loss" in k)
        return loss, log_vars

    def encode_decode(self, img):
        x = self.backbone(img)
        out = self.decode_head(x)
        return O.resize(out, size=img.shape[2:], mode="bilinear",
                        align_corners=self.decode_head.align_corners)
