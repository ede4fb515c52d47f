"""Minimal Registry / build_from_cfg with mmcv's contract (``type=`` string dispatch).

The reference resolves every model component through mmcv Registries
(``@BACKBONES.register_module()`` gaiaseg/models/backbones/dynamic_resnet.py:25,
``@HEADS.register_module()`` dynamic_fcn_head.py:23, ``@SEGMENTORS.register_module()``
dynamic_encoder_decoder.py:8); configs name components by ``type``.  This file provides that
surface without depending on mmcv (absent from the image).
"""
import inspect


class Registry:
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    def __len__(self):
        return len(self._module_dict)

    def __contains__(self, key):
        return key in self._module_dict

    def __repr__(self):
        return "Registry(name=%s, items=%s)" % (self._name, sorted(self._module_dict))

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key, None)

    def _register(self, cls, name=None, force=False):
        names = [name] if isinstance(name, str) else (name or [cls.__name__])
        for n in names:
            if not force and n in self._module_dict:
                raise KeyError("%s is already registered in %s" % (n, self._name))
            self._module_dict[n] = cls

    def register_module(self, name=None, force=False, module=None):
        if module is not None:
            self._register(module, name, force)
            return module

        def _deco(cls):
            self._register(cls, name, force)
            return cls
        return _deco


def build_from_cfg(cfg, registry, default_args=None):
    """Instantiate ``registry[cfg['type']](**rest_of_cfg, **default_args)``."""
    if not isinstance(cfg, dict):
        raise TypeError("cfg must be a dict, but got %s" % type(cfg))
    if "type" not in cfg:
        if default_args is None or "type" not in default_args:
            raise KeyError('`cfg` or `default_args` must contain the key "type", got %s' % (cfg,))
    args = dict(cfg)
    if default_args is not None:
        for k, v in default_args.items():
            args.setdefault(k, v)
    obj_type = args.pop("type")
    if isinstance(obj_type, str):
        obj_cls = registry.get(obj_type)
        if obj_cls is None:
            raise KeyError("%s is not in the %s registry" % (obj_type, registry.name))
    elif inspect.isclass(obj_type):
        obj_cls = obj_type
    else:
        raise TypeError("type must be a str or valid type, but got %s" % type(obj_type))
    try:
        return obj_cls(**args)
    except Exception as e:
        raise type(e)("%s: %s" % (obj_cls.__name__, e))
