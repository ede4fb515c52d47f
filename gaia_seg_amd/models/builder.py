"""Registries and builders with mmseg's names (mmseg.models.builder), used by the reference at
tools/train_supernet.py:174 (build_segmentor) and throughout gaiaseg/models."""
import warnings

from ..core.registry import Registry, build_from_cfg

BACKBONES = Registry("backbone")
NECKS = Registry("neck")
HEADS = Registry("head")
LOSSES = Registry("loss")
SEGMENTORS = Registry("segmentor")
PIXEL_SAMPLERS = Registry("pixel sampler")


def build_backbone(cfg):
    return build_from_cfg(cfg, BACKBONES)


def build_neck(cfg):
    return build_from_cfg(cfg, NECKS)


def build_head(cfg):
    return build_from_cfg(cfg, HEADS)


def build_loss(cfg):
    return build_from_cfg(cfg, LOSSES)


def build_pixel_sampler(cfg, **default_args):
    return build_from_cfg(cfg, PIXEL_SAMPLERS, default_args)


def build_segmentor(cfg, train_cfg=None, test_cfg=None):
    if train_cfg is not None or test_cfg is not None:
        warnings.warn("train_cfg and test_cfg is deprecated, please specify them in model",
                      UserWarning)
    assert cfg.get("train_cfg") is None or train_cfg is None, \
        "train_cfg specified in both outer field and model field "
    assert cfg.get("test_cfg") is None or test_cfg is None, \
        "test_cfg specified in both outer field and model field "
    return build_from_cfg(cfg, SEGMENTORS, dict(train_cfg=train_cfg, test_cfg=test_cfg))
