"""Generates tests/golden/ref_wiring.npz / ref_wiring.json by RUNNING the reference's own forward
methods -- the wiring of the hot path -- on stand-in children.  Run in the build container only:

    python tests/golden/make_ref_wiring_fixtures.py

The methods are cut out of the reference files by AST (the file text is never copied; the function
node is compiled in memory) and called with a fake `self` whose children are PLAIN TORCH modules
defined below (conv -> BatchNorm -> ReLU with seeded weights under the reference's attribute names,
mmseg's `resize` bound to F.interpolate behind a recorder).  What the children compute is this
build's reading of the absent gaiavision / mmcv bricks -- that part stays
unpinned -- but everything the reference's OWN code decides is pinned by these outputs: loop bounds
over depth_state, out_indices, which input each child sees, concat order and offsets, resize
sizes / mode / align_corners, the top-down add order of the FPN, dropout -> conv_seg.

  DynamicResLayer.forward / deploy_forward   gaiaseg/models/utils/dynamic_res_layer.py:159-172
  DynamicResNet.forward                      gaiaseg/models/backbones/dynamic_resnet.py:405-421
  DynamicPPM.forward                         gaiaseg/models/decode_heads/dynamic_psp_head.py:62-73
  PSPHead._transform_inputs/forward/cls_seg  gaiaseg/models/decode_heads/psp_head.py:203-241,277-282
  DynamicUPerHead.psp_forward / forward      gaiaseg/models/decode_heads/dynamic_uper_head.py:81-131
     (+ _transform_inputs / cls_seg of the in-tree base copy dynamic_decode_head.py)
  FCNHead._transform_inputs / cls_seg        gaiaseg/models/decode_heads/fcn_head.py:179-202,248-253
  DynamicFCNHead.forward                     gaiaseg/models/decode_heads/dynamic_fcn_head.py:128-135
  DynamicFCNHead.losses / DynamicPSPHead.losses   dynamic_fcn_head.py:137-159, dynamic_psp_head.py:149-173
     (with the reference's own cross_entropy / weight_reduce_loss / accuracy)
  DynamicDistiller.encode_decode / slide_inference / whole_inference / inference / simple_test /
     aug_test                                gaiaseg/models/segmentors/dynamic_distiller.py:252-262,416-540

Fixture: for every case the state_dict of the stand-ins (keys = the reference's module names), the
seeded inputs, the outputs, and the call trace [(child name, input shape, extra args)] incl. every
resize call.  tests/test_ref_wiring.py replays them through oracle/model.py (CPU) and
tests/test_ref_wiring_gpu.py through the product modules on the MI355X."""
import ast
import json
import os
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _extract(rel, func, cls, namespace):
    with open(os.path.join(REF, rel)) as f:
        tree = ast.parse(f.read())
    body = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls).body
    node = next(n for n in body if isinstance(n, ast.FunctionDef) and n.name == func)
    node.decorator_list = []          # @auto_fp16 / @force_fp32 are no-ops at fp16_enabled=False
    ns = dict(namespace)
    exec(compile(ast.Module(body=[node], type_ignores=[]), rel, "exec"), ns)
    return ns[func]


# ---- stand-in children (plain torch; names follow SURVEY.md Appendix C) ------------------------
class Trace(list):
    def note(self, name, x, *extra):
        shape = [list(t.shape) for t in x] if isinstance(x, (list, tuple)) else list(x.shape)
        self.append([name, shape, [e if isinstance(e, (int, float, str, bool, list)) else repr(e)
                                   for e in extra]])


class ConvModule(nn.Module):
    """conv (bias-free) -> BatchNorm (batch statistics) -> ReLU under the names .conv / .bn"""

    def __init__(self, trace, name, cin, cout, k, padding=0, stride=1, dilation=1, act=True):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride, padding, dilation, bias=False)
        self.bn = nn.BatchNorm2d(cout)
        self.trace, self.name, self.act = trace, name, act

    def forward(self, x, *extra):
        self.trace.note(self.name, x, *extra)
        x = self.bn(self.conv(x))
        return F.relu(x) if self.act else x


class Named(nn.Module):
    def __init__(self, trace, name, mod):
        super().__init__()
        self.mod, self.trace, self.name = mod, trace, name

    def forward(self, x):
        self.trace.note(self.name, x)
        return self.mod(x)


class Bottleneck(nn.Module):
    """torch restatement of the (absent) gaiavision DynamicBottleneck, style='pytorch'"""

    def __init__(self, trace, name, inplanes, planes, stride=1, dilation=1, downsample=False):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, dilation, dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride, bias=False),
                                            nn.BatchNorm2d(planes * 4))
        self.trace, self.name = trace, name

    def forward(self, x):
        self.trace.note(self.name, x)
        out = F.relu(self.bn1(self.conv1(x)))
        out = F.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        idt = self.downsample(x) if self.downsample is not None else x
        return F.relu(out + idt)


def _seed_params(mod, gen):
    with torch.no_grad():
        for name, p in mod.named_parameters():
            if p.dim() == 4:
                fan = p.shape[1] * p.shape[2] * p.shape[3]
                p.copy_(torch.randn(p.shape, generator=gen) * (2.0 / fan) ** 0.5)
            elif name.endswith("weight"):
                p.copy_(torch.rand(p.shape, generator=gen) + 0.5)
            else:
                p.copy_(torch.randn(p.shape, generator=gen) * 0.1)


def _resize_recorder(trace):
    def resize(input, size=None, scale_factor=None, mode="nearest", align_corners=None, warning=True):
        trace.note("resize", input, [int(s) for s in size], mode, bool(align_corners))
        return F.interpolate(input, size, scale_factor, mode, align_corners)
    return resize


def _state(mod, prefix=""):
    return {prefix + k: v.detach().clone() for k, v in mod.state_dict().items()
            if not k.endswith("num_batches_tracked")}


class FakeList(list):
    """a ModuleList look-alike: iterable, indexable, `del self[a:]`, attributes, callable"""
    forward = None

    def __call__(self, *a, **k):
        return type(self).forward(self, *a, **k)


def _feats(gen, n, chans, h, w):
    return [torch.randn(n, c, h >> i, w >> i, generator=gen) for i, c in enumerate(chans)]


# ---- cases --------------------------------------------------------------------------------------
def case_res_layer(out, meta, gen):
    ns = dict(torch=torch, getattr=getattr)
    fwd = _extract("gaiaseg/models/utils/dynamic_res_layer.py", "forward", "DynamicResLayer", ns)
    dep = _extract("gaiaseg/models/utils/dynamic_res_layer.py", "deploy_forward", "DynamicResLayer", ns)
    trace = Trace()
    blocks = [Bottleneck(trace, "0", 8, 4, stride=2, downsample=True)] + \
             [Bottleneck(trace, str(i), 16, 4) for i in range(1, 5)]
    holder = nn.ModuleList(blocks)
    _seed_params(holder, gen)
    holder.train()
    x = torch.randn(2, 8, 12, 10, generator=gen)

    class Layer(FakeList):
        forward = fwd
        deploy_forward = dep
    runs = []
    for depth, deploying in [(5, False), (2, False), (1, False), (3, True)]:
        lay = Layer(blocks)
        lay.depth_state = depth
        if deploying:
            lay._deploying = True
        del trace[:]
        y = lay(x)
        runs.append(dict(depth_state=depth, deploying=deploying, trace=list(trace),
                         blocks_left=len(lay)))
        out["reslayer_d%d%s_y" % (depth, "_deploy" if deploying else "")] = y.detach().numpy()
    out["reslayer_x"] = x.numpy()
    for k, v in _state(holder).items():
        out["reslayer_sd/" + k] = v.numpy()
    meta["res_layer"] = dict(inplanes=8, planes=4, depth_max=5, stride=2, runs=runs)


def case_resnet(out, meta, gen):
    fwd = _extract("gaiaseg/models/backbones/dynamic_resnet.py", "forward", "DynamicResNet",
                   dict(torch=torch, tuple=tuple, enumerate=enumerate, getattr=getattr))
    lfwd = _extract("gaiaseg/models/utils/dynamic_res_layer.py", "forward", "DynamicResLayer",
                    dict(getattr=getattr))
    cases = []
    for tag, deep, out_indices, depths in [("a", False, (0, 1, 2, 3), [2, 1, 2, 1]),
                                           ("b", True, (1, 3), [1, 2, 1, 2])]:
        trace = Trace()
        net = nn.Module()
        if deep:
            sw = [4, 4, 8]
            net.stem = nn.Sequential(
                nn.Conv2d(3, sw[0], 3, 2, 1, bias=False), nn.BatchNorm2d(sw[0]), nn.ReLU(inplace=True),
                nn.Conv2d(sw[0], sw[1], 3, 1, 1, bias=False), nn.BatchNorm2d(sw[1]), nn.ReLU(inplace=True),
                nn.Conv2d(sw[1], sw[2], 3, 1, 1, bias=False), nn.BatchNorm2d(sw[2]), nn.ReLU(inplace=True))
            inpl = sw[2]
        else:
            net.conv1 = nn.Conv2d(3, 8, 7, 2, 3, bias=False)
            net.bn1 = nn.BatchNorm2d(8)
            inpl = 8
        widths, depth_max, strides = [4, 8, 12, 16], [2, 2, 2, 2], [1, 2, 2, 2]

        class Layer(FakeList):
            forward = lfwd
        layers = []
        for i in range(4):
            blocks = [Bottleneck(trace, "layer%d.%d" % (i + 1, j), inpl if j == 0 else widths[i] * 4,
                                 widths[i], strides[i] if j == 0 else 1, downsample=(j == 0))
                      for j in range(depth_max[i])]
            setattr(net, "layer%d" % (i + 1), nn.ModuleList(blocks))
            lay = Layer(blocks)
            lay.depth_state = depths[i]
            layers.append(lay)
            inpl = widths[i] * 4
        _seed_params(net, gen)
        net.train()
        fk = types.SimpleNamespace(deep_stem=deep, out_indices=out_indices,
                                   res_layers=["layer1", "layer2", "layer3", "layer4"],
                                   maxpool=nn.MaxPool2d(3, 2, 1), relu=nn.ReLU(inplace=True))
        if deep:
            fk.stem = net.stem
        else:
            fk.conv1, fk.norm1 = net.conv1, net.bn1
        for i, lay in enumerate(layers):
            setattr(fk, "layer%d" % (i + 1), lay)
        x = torch.randn(2, 3, 64, 48, generator=gen)
        outs = fwd(fk, x)
        assert isinstance(outs, tuple)
        out["resnet_%s_x" % tag] = x.numpy()
        for i, o in enumerate(outs):
            out["resnet_%s_out%d" % (tag, i)] = o.detach().numpy()
        for k, v in _state(net).items():
            out["resnet_%s_sd/%s" % (tag, k)] = v.numpy()
        cases.append(dict(tag=tag, deep_stem=deep, out_indices=list(out_indices), depth=depths,
                          width=widths, depth_max=depth_max, strides=strides,
                          stem_width=[4, 4, 8] if deep else 8, n_outs=len(outs), trace=list(trace)))
    meta["resnet"] = cases


def _ppm(trace, gen, scales, cin, ch, align):
    pfwd = _extract("gaiaseg/models/decode_heads/dynamic_psp_head.py", "forward", "DynamicPPM",
                    dict(resize=_resize_recorder(trace)))

    class PPM(FakeList):
        forward = pfwd
    mods = nn.ModuleList()
    for i, s in enumerate(scales):
        mods.append(nn.Sequential(nn.AdaptiveAvgPool2d(s),
                                  ConvModule(trace, "psp_modules.%d.1" % i, cin, ch, 1)))
    ppm = PPM(mods)
    ppm.align_corners = align
    return ppm, mods


def _head_base(rel, cls, trace):
    ns = dict(torch=torch, resize=_resize_recorder(trace))
    return (_extract(rel, "_transform_inputs", cls, ns), _extract(rel, "cls_seg", cls, ns))


def case_psp(out, meta, gen):
    cases = []
    for tag, in_ch, in_index, ch, ncls, scales, align, hw, dropout in [
            ("a", 24, 3, 8, 5, (1, 2, 3, 6), False, (12, 16), True),
            ("b", 16, 2, 12, 3, (1, 3), True, (9, 7), False)]:
        trace = Trace()
        rel = "gaiaseg/models/decode_heads/psp_head.py"
        tin, cls_seg = _head_base(rel, "PSPHead", trace)
        fwd = _extract(rel, "forward", "PSPHead", dict(torch=torch))
        ppm, ppm_mods = _ppm(trace, gen, scales, in_ch, ch, align)
        head = nn.Module()
        head.psp_modules = ppm_mods
        head.bottleneck = ConvModule(trace, "bottleneck", in_ch + len(scales) * ch, ch, 3, padding=1)
        head.conv_seg = nn.Conv2d(ch, ncls, 1)
        _seed_params(head, gen)
        head.train()
        drop = nn.Dropout2d(0.0) if dropout else None     # p = 0: the call is made, nothing is drawn
        fk = types.SimpleNamespace(in_index=in_index, input_transform=None, align_corners=align,
                                   psp_modules=ppm, bottleneck=head.bottleneck,
                                   dropout=Named(trace, "dropout", drop) if drop else None,
                                   conv_seg=Named(trace, "conv_seg", head.conv_seg))
        fk._transform_inputs = types.MethodType(tin, fk)
        fk.cls_seg = types.MethodType(cls_seg, fk)
        feats = [torch.randn(2, c, hw[0], hw[1], generator=gen) for c in (4, 8, 16, 24)]
        y = fwd(fk, feats)
        for i, f_ in enumerate(feats):
            out["psp_%s_in%d" % (tag, i)] = f_.numpy()
        out["psp_%s_logits" % tag] = y.detach().numpy()
        for k, v in _state(head).items():
            out["psp_%s_sd/%s" % (tag, k)] = v.numpy()
        cases.append(dict(tag=tag, in_channels=in_ch, in_index=in_index, channels=ch,
                          num_classes=ncls, pool_scales=list(scales), align_corners=align,
                          dropout=dropout, trace=list(trace)))
    meta["psp"] = cases


def case_uper(out, meta, gen):
    cases = []
    for tag, in_chs, ch, ncls, scales, align, sizes in [
            ("a", [4, 8, 12, 16], 8, 5, (1, 2, 3, 6), False, [(25, 25), (13, 13), (7, 7), (4, 4)]),
            ("b", [8, 12, 20], 12, 4, (1, 2), True, [(16, 24), (8, 12), (4, 6)])]:
        trace = Trace()
        rel = "gaiaseg/models/decode_heads/dynamic_uper_head.py"
        tin, cls_seg = _head_base("gaiaseg/models/decode_heads/dynamic_decode_head.py",
                                  "DynamicBaseDecodeHead", trace)
        ns = dict(torch=torch, resize=_resize_recorder(trace))
        fwd = _extract(rel, "forward", "DynamicUPerHead", ns)
        pspf = _extract(rel, "psp_forward", "DynamicUPerHead", ns)
        ppm, ppm_mods = _ppm(trace, gen, scales, in_chs[-1], ch, align)
        head = nn.Module()
        head.psp_modules = ppm_mods
        head.bottleneck = ConvModule(trace, "bottleneck", in_chs[-1] + len(scales) * ch, ch, 3, padding=1)
        head.lateral_convs = nn.ModuleList([ConvModule(trace, "lateral_convs.%d" % i, c, ch, 1)
                                            for i, c in enumerate(in_chs[:-1])])
        head.fpn_convs = nn.ModuleList([ConvModule(trace, "fpn_convs.%d" % i, ch, ch, 3, padding=1)
                                        for i in range(len(in_chs) - 1)])
        head.fpn_bottleneck = ConvModule(trace, "fpn_bottleneck", len(in_chs) * ch, ch, 3, padding=1)
        head.conv_seg = nn.Conv2d(ch, ncls, 1)
        _seed_params(head, gen)
        head.train()
        fk = types.SimpleNamespace(in_index=list(range(len(in_chs))), input_transform="multiple_select",
                                   align_corners=align, psp_modules=ppm, bottleneck=head.bottleneck,
                                   lateral_convs=head.lateral_convs, fpn_convs=head.fpn_convs,
                                   fpn_bottleneck=head.fpn_bottleneck, dropout=None,
                                   conv_seg=Named(trace, "conv_seg", head.conv_seg))
        fk._transform_inputs = types.MethodType(tin, fk)
        fk.cls_seg = types.MethodType(cls_seg, fk)
        fk.psp_forward = types.MethodType(pspf, fk)
        feats = [torch.randn(2, c, s[0], s[1], generator=gen) for c, s in zip(in_chs, sizes)]
        y = fwd(fk, feats)
        for i, f_ in enumerate(feats):
            out["uper_%s_in%d" % (tag, i)] = f_.numpy()
        out["uper_%s_logits" % tag] = y.detach().numpy()
        for k, v in _state(head).items():
            out["uper_%s_sd/%s" % (tag, k)] = v.numpy()
        cases.append(dict(tag=tag, in_channels=in_chs, channels=ch, num_classes=ncls,
                          pool_scales=list(scales), align_corners=align, trace=list(trace)))
    meta["uper"] = cases


def case_fcn(out, meta, gen):
    cases = []
    for tag, in_ch, in_index, ch, ncls, num_convs, concat, k in [
            ("a", 16, 2, 8, 5, 2, True, 3),      # mmseg default decode head form (configs 1, 2)
            ("b", 16, 2, 8, 5, 1, False, 3),     # the aux head of the in-tree config
            ("c", 24, -1, 24, 3, 0, True, 1)]:   # num_convs = 0: convs is the identity (in_channels == channels)
        trace = Trace()
        tin, cls_seg = _head_base("gaiaseg/models/decode_heads/fcn_head.py", "FCNHead", trace)
        fwd = _extract("gaiaseg/models/decode_heads/dynamic_fcn_head.py", "forward", "DynamicFCNHead",
                       dict(torch=torch))
        head = nn.Module()
        convs = [ConvModule(trace, "convs.%d" % i, in_ch if i == 0 else ch, ch, k, padding=k // 2)
                 for i in range(num_convs)]
        head.convs = nn.Sequential(*convs) if num_convs else nn.Identity()
        if num_convs == 0:
            assert in_ch == ch              # fcn_head.py ctor: `assert self.in_channels == self.channels`
        if concat:
            head.conv_cat = ConvModule(trace, "conv_cat", in_ch + ch, ch, k, padding=k // 2)
        head.conv_seg = nn.Conv2d(ch, ncls, 1)
        _seed_params(head, gen)
        head.train()
        fk = types.SimpleNamespace(in_index=in_index, input_transform=None, align_corners=False,
                                   convs=head.convs, concat_input=concat, dropout=None,
                                   conv_seg=Named(trace, "conv_seg", head.conv_seg))
        if concat:
            fk.conv_cat = head.conv_cat
        fk._transform_inputs = types.MethodType(tin, fk)
        fk.cls_seg = types.MethodType(cls_seg, fk)
        feats = [torch.randn(2, c, 10, 14, generator=gen) for c in (4, 8, 16, 24)]
        y = fwd(fk, feats)
        for i, f_ in enumerate(feats):
            out["fcn_%s_in%d" % (tag, i)] = f_.numpy()
        out["fcn_%s_logits" % tag] = y.detach().numpy()
        for kk, v in _state(head).items():
            out["fcn_%s_sd/%s" % (tag, kk)] = v.numpy()
        cases.append(dict(tag=tag, in_channels=in_ch, in_index=in_index, channels=ch,
                          num_classes=ncls, num_convs=num_convs, concat_input=concat, kernel_size=k,
                          trace=list(trace)))
    meta["fcn"] = cases


def _load_file(path, name):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def case_losses(out, meta, gen):
    """`losses` of DynamicFCNHead (dynamic_fcn_head.py:137-159) and DynamicPSPHead
    (dynamic_psp_head.py:149-173): resize -> (sampler) -> squeeze -> loss_decode -> accuracy, run with
    the reference's OWN cross_entropy (losses/cross_entropy_loss.py:67-94), weight_reduce_loss and
    accuracy.  loss_decode is mmseg's CrossEntropyLoss [3P]: its forward is restated here as
    loss_weight * cross_entropy(score, label, weight, class_weight=..., reduction='mean',
    avg_factor=None, ignore_index=...)."""
    utils = _load_file(os.path.join(REF, "gaiaseg/models/losses/utils.py"), "ref_loss_utils3")
    acc = _load_file(os.path.join(REF, "gaiaseg/models/losses/accuracy.py"), "ref_accuracy3")
    with open(os.path.join(REF, "gaiaseg/models/losses/cross_entropy_loss.py")) as f:
        tree = ast.parse(f.read())
    node = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "cross_entropy")
    ns = dict(torch=torch, F=F, weight_reduce_loss=utils.weight_reduce_loss)
    exec(compile(ast.Module(body=[node], type_ignores=[]), "cross_entropy_loss.py", "exec"), ns)
    ref_ce = ns["cross_entropy"]
    cases = []
    for tag, head_kind, n, c, h, w, size, align, lw, use_sampler, cw in [
            ("a", "fcn", 2, 19, 8, 16, (64, 128), False, 1.0, False, False),
            ("b", "fcn", 2, 19, 5, 7, (33, 41), False, 0.4, True, False),     # aux-head weight, OHEM-style weights
            ("c", "psp", 1, 5, 6, 6, (24, 24), True, 1.0, False, True),       # align_corners, class weights
            ("d", "psp", 2, 19, 4, 8, (32, 64), False, 1.0, True, False)]:
        trace = Trace()
        rel = ("gaiaseg/models/decode_heads/dynamic_fcn_head.py", "DynamicFCNHead") if head_kind == "fcn" \
            else ("gaiaseg/models/decode_heads/dynamic_psp_head.py", "DynamicPSPHead")
        losses = _extract(rel[0], "losses", rel[1],
                          dict(dict=dict, resize=_resize_recorder(trace), accuracy=acc.accuracy))
        logits = torch.randn(n, c, h, w, generator=gen) * 2
        label = torch.randint(0, c, (n, 1) + size, generator=gen)
        label[:, :, :2] = 255
        label[0, 0, 5:9, 3:8] = 255
        pw = (torch.rand((n,) + size, generator=gen) > 0.35).float()
        class_w = (torch.rand(c, generator=gen) + 0.5) if cw else None

        class Sampler:
            def sample(self, seg_logit, seg_label):
                trace.note("sampler.sample", seg_logit, list(seg_label.shape))
                return pw

        def loss_decode(cls_score, lab, weight=None, ignore_index=-100):
            trace.note("loss_decode", cls_score, list(lab.shape), weight is not None, int(ignore_index))
            return lw * ref_ce(cls_score, lab, weight, class_weight=class_w, reduction="mean",
                               avg_factor=None, ignore_index=ignore_index)
        fk = types.SimpleNamespace(align_corners=align, sampler=Sampler() if use_sampler else None,
                                   loss_decode=loss_decode, ignore_index=255)
        res = losses(fk, logits, label)
        out["losses_%s_logits" % tag] = logits.numpy()
        out["losses_%s_label" % tag] = label.numpy()
        out["losses_%s_pixel_weight" % tag] = pw.numpy()
        if cw:
            out["losses_%s_class_weight" % tag] = class_w.numpy()
        out["losses_%s_loss_seg" % tag] = res["loss_seg"].numpy()
        out["losses_%s_acc_seg" % tag] = res["acc_seg"].numpy()
        cases.append(dict(tag=tag, head=head_kind, align_corners=align, loss_weight=lw,
                          sampler=use_sampler, class_weight=cw, keys=sorted(res.keys()),
                          resize_logit_shape=list(res["resize_logit"].shape) if "resize_logit" in res else None,
                          trace=list(trace)))
    meta["losses"] = cases


def standin_low_logits(img, w1x1):
    """The decode head of the inference fixtures: 8x average pooling of the image, then a fixed 1x1
    conv to the classes -- low-resolution logits [N, C, h/8, w/8] (ceil).  Shared with the tests."""
    return F.conv2d(F.avg_pool2d(img, 8, ceil_mode=True), w1x1)


def case_inference(out, meta, gen):
    """encode_decode / slide_inference / whole_inference / inference / simple_test / aug_test of the
    reference's segmentor (gaiaseg/models/segmentors/dynamic_distiller.py:252-262, 416-540) -- the
    test-time epilogue -- run from its own source.  extract_feat is the identity and the decode head's
    forward_test the stand-in above, so encode_decode's resize and everything behind it is the
    reference's code."""
    rel, cls = "gaiaseg/models/segmentors/dynamic_distiller.py", "DynamicDistiller"
    trace = Trace()
    ns = dict(torch=torch, F=F, resize=_resize_recorder(trace), all=all, list=list, len=len, range=range,
              max=max, min=min, int=int)
    fns = {n: _extract(rel, n, cls, ns) for n in ("encode_decode", "slide_inference", "whole_inference",
                                                  "inference", "simple_test", "aug_test")}
    ncls = 7
    w1x1 = torch.randn(ncls, 3, 1, 1, generator=gen) * 2
    out["inf_w1x1"] = w1x1.numpy()
    cases = []

    def segmentor(mode, align, crop=None, stride=None):
        fk = types.SimpleNamespace(align_corners=align, num_classes=ncls, with_neck=False,
                                   test_cfg=types.SimpleNamespace(mode=mode, crop_size=crop, stride=stride))
        fk.extract_feat = lambda img: img
        fk._decode_head_forward_test = lambda x, metas: standin_low_logits(x, w1x1)
        for n, f in fns.items():
            setattr(fk, n, types.MethodType(f, fk))
        return fk

    def meta_of(ori, flip=False, direction="horizontal"):
        return dict(ori_shape=(ori[0], ori[1], 3), flip=flip, flip_direction=direction)
    for tag, mode, align, hw, ori, flip, direction, crop, stride in [
            ("whole", "whole", False, (64, 96), (64, 96), False, "horizontal", None, None),
            ("whole_rescale_hflip", "whole", False, (64, 96), (50, 70), True, "horizontal", None, None),
            ("whole_align_vflip", "whole", True, (40, 56), (40, 56), True, "vertical", None, None),
            ("slide", "slide", False, (64, 96), (64, 96), False, "horizontal", (32, 48), (21, 32)),
            ("slide_rescale", "slide", False, (72, 104), (90, 130), False, "horizontal", (40, 56), (24, 40))]:
        fk = segmentor(mode, align, crop, stride)
        img = torch.randn(2, 3, hw[0], hw[1], generator=gen)
        metas = [meta_of(ori, flip, direction)] * 2
        del trace[:]
        prob = fk.inference(img, metas, True)
        seg = fk.simple_test(img, metas, True)
        out["inf_%s_img" % tag] = img.numpy()
        if tag in ("whole_rescale_hflip", "slide"):     # (probabilities of two cases: fixture size)
            out["inf_%s_prob" % tag] = prob.numpy()
        out["inf_%s_seg" % tag] = np.stack(seg)
        cases.append(dict(tag=tag, mode=mode, align_corners=align, ori_shape=list(ori), flip=flip,
                          flip_direction=direction, crop_size=crop and list(crop),
                          stride=stride and list(stride),
                          resizes=[t for t in trace if t[0] == "resize"][:4]))
    # aug_test: three views of one 60 x 84 original (scaled, flipped), averaged probabilities
    fk = segmentor("whole", False)
    views = [((60, 84), False, "horizontal"), ((48, 64), True, "horizontal"), ((72, 104), True, "vertical")]
    imgs = [torch.randn(1, 3, v[0][0], v[0][1], generator=gen) for v in views]
    metas = [[meta_of((60, 84), v[1], v[2])] for v in views]
    seg = fk.aug_test(imgs, metas, True)
    for i, im in enumerate(imgs):
        out["inf_aug_img%d" % i] = im.numpy()
    out["inf_aug_seg"] = np.stack(seg)
    meta["inference"] = dict(num_classes=ncls, cases=cases,
                             aug=dict(ori_shape=[60, 84], views=[dict(size=list(v[0]), flip=v[1],
                                                                      flip_direction=v[2]) for v in views]))


def case_forward_train(out, meta, gen):
    """EncoderDecoder.forward_train and its two head wrappers as restated in the reference tree
    ("dynamic_encoder_decoder-distill-backup (1).py":85-143; the class the in-tree
    DynamicEncoderDecoder inherits them from is mmseg's, absent): which head gets which arguments, the
    'decode.' / 'aux.' / 'aux_<i>.' key prefixes (mmseg.core.add_prefix [3P]: {prefix.key: value})."""
    rel, cls = "gaiaseg/models/segmentors/dynamic_encoder_decoder-distill-backup (1).py", "DynamicEncoderDecoder"

    def add_prefix(inputs, prefix):
        return {"%s.%s" % (prefix, k): v for k, v in inputs.items()}
    ns = dict(dict=dict, nn=nn, isinstance=isinstance, enumerate=enumerate, add_prefix=add_prefix)
    fns = {n: _extract(rel, n, cls, ns) for n in ("forward_train", "_decode_head_forward_train",
                                                  "_auxiliary_head_forward_train")}
    cases = []
    for tag, aux in [("single_aux", "one"), ("aux_list", "list"), ("no_aux", None)]:
        calls = []

        class Head(nn.Module):
            def __init__(self, name, base):
                super().__init__()
                self.name, self.base = name, base

            def forward_train(self, x, img_metas, gt, train_cfg):
                calls.append([self.name, [list(t.shape) for t in x], len(img_metas), list(gt.shape),
                              train_cfg])
                return {"loss_seg": torch.tensor(self.base + 0.25), "acc_seg": torch.tensor(self.base * 10)}
        fk = types.SimpleNamespace(train_cfg="TRAIN_CFG", decode_head=Head("decode_head", 1.0))
        fk.extract_feat = lambda img: (img[:, :1] * 2, img[:, 1:] * 3)
        if aux == "one":
            fk.auxiliary_head = Head("auxiliary_head", 2.0)
        elif aux == "list":
            fk.auxiliary_head = nn.ModuleList([Head("auxiliary_head.0", 2.0), Head("auxiliary_head.1", 3.0)])
        fk.with_auxiliary_head = aux is not None
        for n, f in fns.items():
            setattr(fk, n, types.MethodType(f, fk))
        losses = fk.forward_train(torch.zeros(2, 3, 8, 8), [{}, {}], torch.zeros(2, 1, 8, 8, dtype=torch.long))
        cases.append(dict(tag=tag, aux=aux, calls=calls, losses={k: float(v) for k, v in losses.items()},
                          order=list(losses.keys())))
    meta["forward_train"] = cases


def main():
    torch.manual_seed(0)
    gen = torch.Generator().manual_seed(20240)
    out, meta = {}, {}
    case_res_layer(out, meta, gen)
    case_resnet(out, meta, gen)
    case_psp(out, meta, gen)
    case_uper(out, meta, gen)
    case_fcn(out, meta, gen)
    case_losses(out, meta, gen)
    case_inference(out, meta, gen)
    case_forward_train(out, meta, gen)
    np.savez_compressed(os.path.join(HERE, "ref_wiring.npz"), **out)
    with open(os.path.join(HERE, "ref_wiring.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote ref_wiring.npz (%d arrays, %.1f KB) and ref_wiring.json"
          % (len(out), os.path.getsize(os.path.join(HERE, "ref_wiring.npz")) / 1024))


if __name__ == "__main__":
    main()
