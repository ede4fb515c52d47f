"""Helpers shared by the model-level tests: tiny supernet configs, paired product/oracle models."""
import copy

import torch

CONV = dict(type="DynConv2d")


def tiny_backbone(deep_stem=False, os8=False, norm="DynSyncBN", avg_down=False):
    cfg = dict(type="DynamicResNet", in_channels=3,
               stem_width=[16, 16, 32] if deep_stem else 32,
               body_depth=[2, 2, 3, 2], body_width=[32, 64, 96, 128], num_stages=4,
               out_indices=(0, 1, 2, 3), conv_cfg=CONV,
               norm_cfg=dict(type=norm, requires_grad=True, group_size=1) if norm == "DynSyncBN"
               else dict(type=norm, requires_grad=True), style="pytorch", deep_stem=deep_stem)
    if os8:
        cfg.update(strides=(1, 2, 1, 1), dilations=(1, 1, 2, 4), contract_dilation=True)
    if avg_down:
        cfg.update(avg_down=True)
    return cfg


def fcn_head(in_channels=512, in_index=3, channels=64, num_convs=2, concat_input=True,
             loss_weight=1.0, dropout=0.0):
    return dict(type="DynamicFCNHead", conv_cfg=CONV, in_channels=in_channels, in_index=in_index,
                channels=channels, num_convs=num_convs, concat_input=concat_input,
                dropout_ratio=dropout, num_classes=19, norm_cfg=dict(type="SyncBN", requires_grad=True),
                align_corners=False,
                loss_decode=dict(type="CrossEntropyLoss", use_sigmoid=False, loss_weight=loss_weight))


def psp_head(in_channels=512, in_index=3, channels=64, dropout=0.0):
    return dict(type="DynamicPSPHead", conv_cfg=CONV, in_channels=in_channels, in_index=in_index,
                channels=channels, pool_scales=(1, 2, 3, 6), dropout_ratio=dropout, num_classes=19,
                norm_cfg=dict(type="SyncBN", requires_grad=True), align_corners=False,
                loss_decode=dict(type="CrossEntropyLoss", use_sigmoid=False, loss_weight=1.0))


def uper_head(in_channels=(128, 256, 384, 512), channels=64, dropout=0.0):
    return dict(type="DynamicUPerHead", conv_cfg=CONV, in_channels=list(in_channels),
                in_index=[0, 1, 2, 3], channels=channels, pool_scales=(1, 2, 3, 6),
                dropout_ratio=dropout, num_classes=19,
                norm_cfg=dict(type="SyncBN", requires_grad=True), align_corners=False,
                loss_decode=dict(type="CrossEntropyLoss", use_sigmoid=False, loss_weight=1.0))


def model_cfg(head, aux=True, **bk):
    cfg = dict(type="DynamicEncoderDecoder", backbone=tiny_backbone(**bk), decode_head=head,
               train_cfg=dict(), test_cfg=dict(mode="whole"))
    if aux:
        cfg["auxiliary_head"] = fcn_head(in_channels=384, in_index=2, channels=32, num_convs=1,
                                         concat_input=False, loss_weight=0.4)
    return cfg


ARCHS = {
    "max": dict(stem=32, width=[32, 64, 96, 128], depth=[2, 2, 3, 2]),
    "sub": dict(stem=16, width=[16, 48, 64, 96], depth=[1, 2, 2, 1]),
    "min": dict(stem=16, width=[16, 32, 48, 64], depth=[1, 1, 1, 1]),
}


def arch_meta(name, deep_stem=False):
    a = ARCHS[name]
    stem = [a["stem"] // 2, a["stem"] // 2, a["stem"]] if deep_stem else a["stem"]
    return {"backbone": {"stem": {"width": stem},
                         "body": {"width": list(a["width"]), "depth": list(a["depth"])}}}


def randomize(model, seed=0):
    """Random conv weights and BN affine / running stats; norm3 NOT zeroed so that errors in
    the residual branches are not masked (SURVEY.md §7 'quirks')."""
    g = torch.Generator().manual_seed(seed)
    from torch.nn.modules.batchnorm import _BatchNorm
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, _BatchNorm):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
            elif hasattr(m, "weight") and getattr(m, "weight", None) is not None and m.weight.dim() == 4:
                fan_in = m.weight.shape[1] * m.weight.shape[2] * m.weight.shape[3]
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5)
                if m.bias is not None:
                    m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)


def make_pair(cfg, seed=0, ocfg=None):
    """(product model on CPU params — move with .cuda(), oracle on CPU) with equal weights."""
    from gaia_seg_amd.models import build_segmentor
    from oracle.model import OEncoderDecoder
    prod = build_segmentor(copy.deepcopy(cfg))
    randomize(prod, seed)
    orc = OEncoderDecoder(**{k: v for k, v in copy.deepcopy(ocfg or cfg).items() if k != "type"})
    sd = {k: v.detach().clone().contiguous() for k, v in prod.state_dict().items()}
    missing, unexpected = orc.load_state_dict(sd, strict=True)
    return prod, orc


def make_batch(n, h, w, seed=0, ncls=19):
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(n, 3, h, w, generator=g)
    gt = torch.randint(0, ncls, (n, 1, h, w), generator=g)
    gt[:, :, :2, :] = 255
    gt[:, :, :, -3:] = 255
    drop = torch.rand(n, 1, h, w, generator=g) < 0.05
    gt[drop] = 255
    return img, gt
