"""Closed-form conv FLOPs / parameters of the active subnet (the analysis the
reference's tools/count_flops.py:118-179 obtains from gaiavision's get_model_complexity_info).

forward FLOPs of a conv = 2 * N * Ho * Wo * Cout_act * Cin_act * kh * kw,
Ho = floor((H + 2p - d(k-1) - 1)/s) + 1.  Only convolutions are counted (BN / ReLU / pooling are
O(activations) and excluded)."""
from ..hip.ops import conv_out_size
from .bricks import DynamicConv2d


def _conv(m, cin, h, w):
    co = m.width_state
    kh, kw = m.kernel_size
    ho = conv_out_size(h, kh, m.stride, m.padding, m.dilation)
    wo = conv_out_size(w, kw, m.stride, m.padding, m.dilation)
    flops = 2.0 * ho * wo * co * cin * kh * kw
    params = co * cin * kh * kw + (co if m.bias is not None else 0)
    return flops, params, co, ho, wo


def backbone_flops(backbone, h, w, in_channels=3):
    """(flops per image, params, flops of the 3x3 bottleneck convs, feature shapes)."""
    total = params = k3 = 0.0
    c = in_channels
    if backbone.deep_stem:
        for i in (0, 3, 6):
            f, p, c, h, w = _conv(backbone.stem[i], c, h, w)
            total += f
            params += p + 2 * c
    else:
        f, p, c, h, w = _conv(backbone.conv1, c, h, w)
        total += f
        params += p + 2 * c
    mp = backbone.maxpool
    h = (h + 2 * mp.padding - mp.kernel_size) // mp.stride + 1
    w = (w + 2 * mp.padding - mp.kernel_size) // mp.stride + 1
    feats = []
    for name in backbone.res_layers:
        layer = getattr(backbone, name)
        for blk in layer.active_blocks():
            cin = c
            f1, p1, c1, h1, w1 = _conv(blk.conv1, cin, h, w)
            f2, p2, c2, h2, w2 = _conv(blk.conv2, c1, h1, w1)
            f3, p3, c3, h3, w3 = _conv(blk.conv3, c2, h2, w2)
            total += f1 + f2 + f3
            k3 += f2
            params += p1 + p2 + p3 + 2 * (c1 + c2 + c3)
            if blk.downsample is not None:
                for m in blk.downsample:
                    if isinstance(m, DynamicConv2d):
                        fd, pd, cd, _, _ = _conv(m, cin, h, w)
                        total += fd
                        params += pd + 2 * cd
            c, h, w = c3, h3, w3
        feats.append((c, h, w))
    return total, params, k3, feats


def fcn_head_flops(head, c, h, w):
    total = 0.0
    x_c = c
    mods = [] if head.num_convs == 0 else list(head.convs)
    for m in mods:
        f, _, c, h, w = _conv(m.conv, c, h, w)
        total += f
    if head.concat_input:
        f, _, c, h, w = _conv(head.conv_cat.conv, x_c + c, h, w)
        total += f
    f, _, _, _, _ = _conv(head.conv_seg, c, h, w)
    return total + f


def psp_head_flops(head, c, h, w):
    total = 0.0
    for s, ppm in zip(head.pool_scales, head.psp_modules):
        f, _, _, _, _ = _conv(ppm[1].conv, c, s, s)
        total += f
    f, _, cb, _, _ = _conv(head.bottleneck.conv, c + len(head.pool_scales) * head.channels, h, w)
    total += f
    f, _, _, _, _ = _conv(head.conv_seg, cb, h, w)
    return total + f


def uper_head_flops(head, feats):
    """DynamicUPerHead (gaiaseg/models/decode_heads/dynamic_uper_head.py:81-131): PPM + bottleneck on
    the last level, one 1x1 lateral and one 3x3 fpn conv per other level at that level's size, the
    3x3 fpn_bottleneck over the concatenated levels and the classifier at the first level's size."""
    levels = [feats[i] for i in head.in_index]
    c5, h5, w5 = levels[-1]
    total = 0.0
    scales = head.psp_modules.pool_scales
    for s, ppm in zip(scales, head.psp_modules):
        total += _conv(ppm[1].conv, c5, s, s)[0]
    total += _conv(head.bottleneck.conv, c5 + len(scales) * head.channels, h5, w5)[0]
    for (c, h, w), lat, fpn in zip(levels[:-1], head.lateral_convs, head.fpn_convs):
        total += _conv(lat.conv, c, h, w)[0]
        total += _conv(fpn.conv, head.channels, h, w)[0]
    _, h0, w0 = levels[0]
    f, _, cb, _, _ = _conv(head.fpn_bottleneck.conv, len(levels) * head.channels, h0, w0)
    return total + f + _conv(head.conv_seg, cb, h0, w0)[0]


def model_flops(model, h, w):
    """dict(backbone, backbone_3x3, decode, aux, total) in FLOPs per image for the current arch."""
    b, params, k3, feats = backbone_flops(model.backbone, h, w)
    out = dict(backbone=b, backbone_3x3=k3, backbone_params=params)
    for key, head in (("decode", model.decode_head), ("aux", getattr(model, "auxiliary_head", None))):
        if head is None:
            continue
        if isinstance(head.in_index, int):
            c, fh, fw = feats[head.in_index % len(feats)]
        name = type(head).__name__
        if name == "DynamicFCNHead":
            out[key] = fcn_head_flops(head, c, fh, fw)
        elif name == "DynamicPSPHead":
            out[key] = psp_head_flops(head, c, fh, fw)
        elif name == "DynamicUPerHead":
            out[key] = uper_head_flops(head, feats)
        else:
            out[key] = float("nan")
    out["total"] = sum(v for k, v in out.items() if k in ("backbone", "decode", "aux") and v == v)
    return out
