"""Generates tests/golden/ref_ctors.json by RUNNING the reference's own CONSTRUCTORS, `init_weights`,
`_freeze_stages` / `_freeze_layers` and `train` on recording stand-in children.  Run in the build
container only:

    python tests/golden/make_ref_ctor_fixtures.py

The class nodes are cut out of the reference files by AST and compiled in memory (the file text is
never copied); the names they import from the absent gaiavision / mmcv / mmseg packages are bound to
RECORDING stand-ins defined here: `build_conv_layer`, `build_norm_layer`, the `block` class
(DynamicBottleneck), `DynamicConvModule`, `DynamicConv2d`, `build_loss`, `build_pixel_sampler`,
`kaiming_init`, `constant_init`.  A stand-in logs the arguments the REFERENCE'S code handed it and is a
small real torch module (so module paths, `requires_grad` flags and `.training` flags are real).
What such a child does inside is this build's reading of the absent package
and is marked `"standin_internal": true` in the dump -- not pinned; everything the reference's own
code decides IS pinned: per-stage stride / dilation / planes, `contract_dilation` -> first-block
dilation, the projection shortcut's form (stride on the conv or AvgPool2d(ceil_mode,
count_include_pad=False) under avg_down), stem layout and indices, module names and order, head
conv counts / channel arithmetic / kernel sizes / padding, PPM / FPN structure, which init goes to
which module (`norm3` zeroing), what `frozen_stages` / `frozen_layers` / `norm_eval` freeze.

  DynamicResLayer.__init__                    gaiaseg/models/utils/dynamic_res_layer.py:46-147
  DynamicResNet.__init__ / _make_stem_layer   gaiaseg/models/backbones/dynamic_resnet.py:81-182,255-302
  DynamicResNet.init_weights                  ...:336-367
  DynamicResNet._freeze_stages/_freeze_layers/train   ...:304-334,369-379
  DynamicFCNHead.__init__                     gaiaseg/models/decode_heads/dynamic_fcn_head.py:36-126
  DynamicPPM.__init__ / DynamicPSPHead.__init__   gaiaseg/models/decode_heads/dynamic_psp_head.py:38-60,85-147
  DynamicUPerHead.__init__                    gaiaseg/models/decode_heads/dynamic_uper_head.py:28-79
     (its mmseg base class is absent; the in-tree copy DynamicBaseDecodeHead,
      gaiaseg/models/decode_heads/dynamic_decode_head.py:56-98, stands in for it)

tests/test_ref_ctors.py builds the same configurations through this repo's registries (product
modules, CPU, no GPU needed) and through oracle/model.py and requires the same dump."""
import ast
import json
import os
import warnings
from abc import ABCMeta, abstractmethod
from collections.abc import Sequence

import torch
import torch.nn as nn
from torch.nn.modules.batchnorm import _BatchNorm

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _class(rel, cls, namespace):
    """the class `cls` of reference file `rel`, compiled from its AST node inside `namespace`"""
    with open(os.path.join(REF, rel)) as f:
        tree = ast.parse(f.read())
    node = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls)
    node.decorator_list = []                       # registry decorators
    for n in node.body:
        if isinstance(n, ast.FunctionDef):         # @auto_fp16 / @force_fp32: no-ops at fp16_enabled=False
            n.decorator_list = [d for d in n.decorator_list
                                if isinstance(d, ast.Name) and d.id in ("property", "staticmethod", "abstractmethod")]
    ns = dict(namespace)
    exec(compile(ast.Module(body=[node], type_ignores=[]), rel, "exec"), ns)
    return ns[cls]


def _plain(v):
    if isinstance(v, dict):
        return {str(k): _plain(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    if isinstance(v, (int, float, str, bool)) or v is None:
        return v
    if isinstance(v, type):
        return v.__name__
    return repr(v)


# ---- recording stand-ins -----------------------------------------------------------------------
class DynamicMixin:
    """[3P] gaiavision.core.DynamicMixin: only its presence as a base class matters to a constructor"""


class RecConv(nn.Conv2d):
    """what build_conv_layer / DynamicConv2d returned: a real (small) conv that remembers the call"""


def _conv(cfg, args, kwargs, how):
    m = RecConv(*args, **kwargs)
    m.spec = dict(kind="conv", how=how, cfg=_plain(cfg), args=_plain(args), kwargs=_plain(kwargs))
    return m


def build_conv_layer(cfg, *args, **kwargs):
    return _conv(cfg, args, kwargs, "build_conv_layer")


def DynamicConv2d(*args, **kwargs):
    return _conv(None, args, kwargs, "DynamicConv2d")


class RecNorm(nn.BatchNorm2d):
    pass


def build_norm_layer(cfg, num_features, postfix=""):
    m = RecNorm(num_features)
    m.spec = dict(kind="norm", cfg=_plain(cfg), num_features=num_features, postfix=_plain(postfix))
    for p in m.parameters():                       # [3P] mmcv: cfg['requires_grad'] (default True)
        p.requires_grad = cfg.get("requires_grad", True)
    return "bn%s" % postfix, m                     # [3P] mmcv abbreviation of the BN family


class DynamicBottleneck(nn.Module):
    """`block`: records the keyword arguments DynamicResLayer hands it.  Internals = this build's
    reading of gaiavision's DynamicBottleneck, marked standin_internal."""
    expansion = 4

    def __init__(self, **kwargs):
        super().__init__()
        ds = kwargs.get("downsample")
        self.spec = dict(kind="block", kwargs={k: _plain(v) for k, v in kwargs.items() if k != "downsample"},
                         has_downsample=ds is not None)
        inplanes, planes = kwargs["inplanes"], kwargs["planes"]
        cc, nc = kwargs.get("conv_cfg"), kwargs.get("norm_cfg")
        self.conv1 = build_conv_layer(cc, inplanes, planes, kernel_size=1, bias=False)
        self.add_module(*build_norm_layer(nc, planes, postfix=1))
        self.conv2 = build_conv_layer(cc, planes, planes, kernel_size=3, bias=False)
        self.add_module(*build_norm_layer(nc, planes, postfix=2))
        self.conv3 = build_conv_layer(cc, planes, planes * 4, kernel_size=1, bias=False)
        self.add_module(*build_norm_layer(nc, planes * 4, postfix=3))
        for m in (self.conv1, self.bn1, self.conv2, self.bn2, self.conv3, self.bn3):
            m.spec["standin_internal"] = True
        self.downsample = ds

    @property
    def norm3(self):
        return self.bn3


class DynamicConvModule(nn.Module):
    """records the call; internals ([3P] mmcv ConvModule: conv -> norm -> act) marked standin_internal"""

    def __init__(self, in_channels, out_channels, kernel_size, **kwargs):
        super().__init__()
        self.spec = dict(kind="conv_module", in_channels=in_channels, out_channels=out_channels,
                         kernel_size=_plain(kernel_size), kwargs=_plain(kwargs))
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, padding=kwargs.get("padding", 0),
                              bias=kwargs.get("norm_cfg") is None)
        if kwargs.get("norm_cfg") is not None:
            self.bn = nn.BatchNorm2d(out_channels)


class RecLeaf(nn.Module):
    def __init__(self, kind, **spec):
        super().__init__()
        self.spec = dict(kind=kind, **_plain(spec))


def build_loss(cfg):
    return RecLeaf("loss", cfg=cfg)


def build_pixel_sampler(cfg, context=None):
    return RecLeaf("pixel_sampler", cfg=cfg, context=type(context).__name__)


def dump(mod):
    """[path, class, spec-or-repr] for every module of the tree, in registration order"""
    rows = []
    for path, m in mod.named_modules():
        if path == "":
            continue
        spec = getattr(m, "spec", None)
        if spec is not None:
            rows.append([path, spec["kind"], spec])
        elif isinstance(m, (nn.Sequential, nn.ModuleList)):
            rows.append([path, "container", dict(cls="Sequential" if isinstance(m, nn.Sequential)
                                                 else "ModuleList", len=len(m))])
        elif type(m).__module__.startswith("torch.nn"):
            inside = any(isinstance(p, (DynamicBottleneck, DynamicConvModule))
                         for p in _parents(mod, path))
            rows.append([path, "torch", dict(repr=repr(m), standin_internal=inside)])
        else:
            rows.append([path, "ref_class", dict(cls=type(m).__name__)])
    return rows


def _parents(root, path):
    parts, cur, out = path.split(".")[:-1], root, []
    for p in parts:
        cur = getattr(cur, p) if not p.isdigit() else cur[int(p)]
        out.append(cur)
    return out


def _attrs(obj, names):
    return {n: _plain(getattr(obj, n)) for n in names if hasattr(obj, n)}


def _flags(mod):
    return dict(training={p: m.training for p, m in mod.named_modules() if p},
                requires_grad={n: p.requires_grad for n, p in mod.named_parameters()})


# ---- the reference classes --------------------------------------------------------------------
def ref_classes():
    init_log = []
    names = {}

    def kaiming_init(module, *a, **k):
        init_log.append([names.get(id(module), "?"), "kaiming_init", _plain(a), _plain(k)])

    def constant_init(module, val, *a, **k):
        init_log.append([names.get(id(module), "?"), "constant_init", _plain((val,) + a), _plain(k)])

    def normal_init(module, *a, **k):
        init_log.append([names.get(id(module), "?"), "normal_init", _plain(a), _plain(k)])

    base = dict(nn=nn, torch=torch, warnings=warnings, Sequence=Sequence, _BatchNorm=_BatchNorm,
                ABCMeta=ABCMeta, abstractmethod=abstractmethod, DynamicMixin=DynamicMixin,
                build_conv_layer=build_conv_layer, build_norm_layer=build_norm_layer,
                DynamicBottleneck=DynamicBottleneck, DynamicConvModule=DynamicConvModule,
                DynamicConv2d=DynamicConv2d, build_loss=build_loss,
                build_pixel_sampler=build_pixel_sampler, kaiming_init=kaiming_init,
                constant_init=constant_init, normal_init=normal_init)
    dh = "gaiaseg/models/decode_heads/"
    c = {}
    c["DynamicResLayer"] = _class("gaiaseg/models/utils/dynamic_res_layer.py", "DynamicResLayer", base)
    c["DynamicResNet"] = _class("gaiaseg/models/backbones/dynamic_resnet.py", "DynamicResNet",
                                dict(base, DynamicResLayer=c["DynamicResLayer"]))
    c["FCNHead"] = _class(dh + "fcn_head.py", "FCNHead", base)
    c["DynamicFCNHead"] = _class(dh + "dynamic_fcn_head.py", "DynamicFCNHead", dict(base, FCNHead=c["FCNHead"]))
    c["PSPHead"] = _class(dh + "psp_head.py", "PSPHead", base)
    c["DynamicPPM"] = _class(dh + "dynamic_psp_head.py", "DynamicPPM", base)
    c["DynamicPSPHead"] = _class(dh + "dynamic_psp_head.py", "DynamicPSPHead",
                                 dict(base, PSPHead=c["PSPHead"], DynamicPPM=c["DynamicPPM"]))
    c["DynamicBaseDecodeHead"] = _class(dh + "dynamic_decode_head.py", "DynamicBaseDecodeHead",
                                        dict(base, PSPHead=c["PSPHead"]))
    c["DynamicUPerHead"] = _class(dh + "dynamic_uper_head.py", "DynamicUPerHead",
                                  dict(base, BaseDecodeHead=c["DynamicBaseDecodeHead"],
                                       DynamicPPM=c["DynamicPPM"]))
    return c, init_log, names


CONV, NORM = dict(type="DynConv2d"), dict(type="DynSyncBN", requires_grad=True, group_size=1)
HEAD_NORM = dict(type="SyncBN", requires_grad=True)

BACKBONE_CASES = [
    dict(tag="os32_7x7", kwargs=dict(in_channels=3, stem_width=8, body_width=[4, 8, 12, 16],
                                     body_depth=[2, 1, 3, 2], conv_cfg=CONV, norm_cfg=NORM)),
    dict(tag="os8_deep_stem", kwargs=dict(in_channels=3, stem_width=[4, 4, 8], body_width=[4, 8, 12, 16],
                                          body_depth=[1, 2, 2, 3], strides=(1, 2, 1, 1),
                                          dilations=(1, 1, 2, 4), out_indices=(1, 3), deep_stem=True,
                                          contract_dilation=True, conv_cfg=CONV, norm_cfg=NORM)),
    dict(tag="avg_down_frozen", kwargs=dict(in_channels=3, stem_width=8, body_width=[4, 8, 12, 16],
                                            body_depth=[2, 2, 3, 1], avg_down=True, frozen_stages=1,
                                            frozen_layers=[0, 1, 2, 0], norm_eval=False,
                                            zero_init_residual=False, conv_cfg=CONV, norm_cfg=NORM)),
    dict(tag="three_stages_norm_eval", kwargs=dict(in_channels=3, stem_width=[4, 4, 8],
                                                   body_width=[4, 8, 12, 16], body_depth=[1, 1, 2, 9],
                                                   num_stages=3, strides=(1, 2, 2), dilations=(1, 1, 2),
                                                   out_indices=(0, 2), deep_stem=True, frozen_stages=0,
                                                   norm_eval=True, style="caffe", with_cp=False,
                                                   conv_cfg=CONV, norm_cfg=NORM)),
]

RESLAYER_CASES = [
    dict(tag="identity_width", kwargs=dict(inplanes=16, planes=4, depth=3, stride=1, dilation=1)),
    dict(tag="stride2", kwargs=dict(inplanes=16, planes=8, depth=2, stride=2, dilation=1)),
    dict(tag="avg_down_stride2", kwargs=dict(inplanes=16, planes=8, depth=2, stride=2, avg_down=True)),
    dict(tag="avg_down_stride1", kwargs=dict(inplanes=8, planes=8, depth=1, stride=1, avg_down=True)),
    dict(tag="dilated_contract", kwargs=dict(inplanes=32, planes=8, depth=3, stride=1, dilation=4,
                                             contract_dilation=True)),
    dict(tag="dilated_plain", kwargs=dict(inplanes=32, planes=8, depth=2, stride=1, dilation=2,
                                          contract_dilation=False, style="pytorch", with_cp=False)),
]

FCN_CASES = [
    dict(tag="decode_2convs_concat", kwargs=dict(in_channels=64, channels=16, num_classes=19, num_convs=2,
                                                 concat_input=True, dropout_ratio=0.1, conv_cfg=CONV,
                                                 norm_cfg=HEAD_NORM, in_index=3, align_corners=False)),
    dict(tag="aux_1conv", kwargs=dict(in_channels=48, channels=8, num_classes=19, num_convs=1,
                                      concat_input=False, dropout_ratio=0.1, conv_cfg=CONV,
                                      norm_cfg=HEAD_NORM, in_index=2,
                                      loss_decode=dict(type="CrossEntropyLoss", use_sigmoid=False,
                                                       loss_weight=0.4))),
    dict(tag="zero_convs_k1", kwargs=dict(in_channels=16, channels=16, num_classes=5, num_convs=0,
                                          concat_input=True, kernel_size=1, dropout_ratio=0.0,
                                          conv_cfg=CONV, norm_cfg=HEAD_NORM, in_index=-1,
                                          sampler=dict(type="OHEMPixelSampler", thresh=0.7, min_kept=100000))),
    dict(tag="resize_concat", kwargs=dict(in_channels=[8, 16, 32], channels=12, num_classes=7, num_convs=3,
                                          concat_input=False, conv_cfg=CONV, norm_cfg=HEAD_NORM,
                                          in_index=[0, 1, 3], input_transform="resize_concat",
                                          align_corners=True, ignore_index=254)),
]

PSP_CASES = [
    dict(tag="psp_1236", kwargs=dict(in_channels=64, channels=16, num_classes=19, pool_scales=(1, 2, 3, 6),
                                     dropout_ratio=0.1, conv_cfg=CONV, norm_cfg=HEAD_NORM, in_index=3)),
    dict(tag="psp_13_align", kwargs=dict(in_channels=24, channels=12, num_classes=4, pool_scales=(1, 3),
                                         dropout_ratio=0.0, conv_cfg=CONV, norm_cfg=HEAD_NORM, in_index=2,
                                         align_corners=True)),
]

UPER_CASES = [
    dict(tag="uper_4_levels", kwargs=dict(in_channels=[8, 16, 24, 32], in_index=[0, 1, 2, 3], channels=12,
                                          num_classes=19, pool_scales=(1, 2, 3, 6), dropout_ratio=0.1,
                                          conv_cfg=CONV, norm_cfg=HEAD_NORM, align_corners=False)),
    dict(tag="uper_3_levels", kwargs=dict(in_channels=[8, 12, 20], in_index=[1, 2, 3], channels=8,
                                          num_classes=4, pool_scales=(1, 2), dropout_ratio=0.0,
                                          conv_cfg=CONV, norm_cfg=HEAD_NORM, align_corners=True)),
]

BACKBONE_ATTRS = ["res_layers", "inplanes", "feat_dim", "active_feat_dim", "stem_state", "body_state",
                  "body_depth", "stem_width", "body_width", "num_stages", "strides", "dilations",
                  "out_indices", "style", "deep_stem", "avg_down", "frozen_stages", "frozen_layers",
                  "norm_eval", "zero_init_residual", "contract_dilation", "with_cp"]
HEAD_ATTRS = ["in_channels", "channels", "num_classes", "dropout_ratio", "in_index", "input_transform",
              "ignore_index", "align_corners", "num_convs", "concat_input", "kernel_size", "pool_scales",
              "fp16_enabled", "conv_cfg", "norm_cfg", "act_cfg"]


def main():
    torch.manual_seed(0)
    classes, init_log, names = ref_classes()
    out = {}

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rl = []
        for c in RESLAYER_CASES:
            kw = dict(c["kwargs"], block=DynamicBottleneck, conv_cfg=CONV, norm_cfg=NORM)
            lay = classes["DynamicResLayer"](**kw)
            rl.append(dict(tag=c["tag"], kwargs=_plain(c["kwargs"]), modules=dump(lay),
                           attrs=_attrs(lay, ["depth_state", "width_state", "avg_down"])))
        out["res_layer"] = rl

        bb = []
        for c in BACKBONE_CASES:
            net = classes["DynamicResNet"](**c["kwargs"])
            rec = dict(tag=c["tag"], kwargs=_plain(c["kwargs"]), modules=dump(net),
                       attrs=_attrs(net, BACKBONE_ATTRS),
                       layer_states={n: _attrs(getattr(net, n), ["depth_state", "width_state"])
                                     for n in net.res_layers},
                       flags_after_init=_flags(net))
            names.clear()
            names.update({id(m): p for p, m in net.named_modules()})
            del init_log[:]
            net.init_weights(None)
            rec["init_weights"] = list(init_log)
            net.train(True)
            rec["flags_train"] = _flags(net)
            net.train(False)
            rec["flags_eval"] = _flags(net)
            bb.append(rec)
        out["backbone"] = bb

        for key, cls, cases in [("fcn_head", "DynamicFCNHead", FCN_CASES),
                                ("psp_head", "DynamicPSPHead", PSP_CASES),
                                ("uper_head", "DynamicUPerHead", UPER_CASES)]:
            recs = []
            for c in cases:
                head = classes[cls](**c["kwargs"])
                names.clear()
                names.update({id(m): p for p, m in head.named_modules()})
                del init_log[:]
                head.init_weights()
                recs.append(dict(tag=c["tag"], kwargs=_plain(c["kwargs"]), modules=dump(head),
                                 attrs=_attrs(head, HEAD_ATTRS), init_weights=list(init_log)))
            out[key] = recs

    path = os.path.join(HERE, "ref_ctors.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote %s: %s" % (path, {k: len(v) for k, v in out.items()}))


if __name__ == "__main__":
    main()
