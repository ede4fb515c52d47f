"""Python-file configs with ``_base_`` inheritance and ``--cfg-options`` overrides.

Same user-visible behaviour as the mmcv ``Config`` the reference CLI relies on
(tools/train_supernet.py:102-108): a config is a python file whose public names become keys,
``_base_`` lists files merged first (dicts merge recursively, ``_delete_=True`` replaces),
``merge_from_dict`` takes dotted keys, attribute access works on nested dicts.
"""
import ast
import copy
import os.path as osp

BASE_KEY = "_base_"
DELETE_KEY = "_delete_"


class ConfigDict(dict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError("'ConfigDict' object has no attribute '%s'" % name)

    def __setattr__(self, name, value):
        self[name] = value

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _to_config_dict(obj):
    if isinstance(obj, dict):
        return ConfigDict({k: _to_config_dict(v) for k, v in obj.items()})
    if isinstance(obj, list):
        return [_to_config_dict(v) for v in obj]
    if isinstance(obj, tuple):
        return tuple(_to_config_dict(v) for v in obj)
    return obj


def _merge_a_into_b(a, b):
    b = dict(b)
    for k, v in a.items():
        if isinstance(v, dict) and k in b and not v.get(DELETE_KEY, False):
            if not isinstance(b[k], dict):
                raise TypeError("%s=%s in child config cannot inherit from base because %s is a "
                                "dict in the child config but is of type %s in base config. You "
                                "may set `%s=True` to ignore the base config"
                                % (k, v, k, type(b[k]), DELETE_KEY))
            b[k] = _merge_a_into_b(v, b[k])
        else:
            if isinstance(v, dict):
                v = {kk: vv for kk, vv in v.items() if kk != DELETE_KEY}
            b[k] = v
    return b


def _load_py(filename):
    filename = osp.abspath(osp.expanduser(filename))
    if not osp.isfile(filename):
        raise FileNotFoundError("config file %s does not exist" % filename)
    if not filename.endswith(".py"):
        raise IOError("Only py type configs are supported")
    with open(filename, "r", encoding="utf-8") as f:
        text = f.read()
    ast.parse(text)  # syntax check with a clean error
    scope = {"__file__": filename}
    exec(compile(text, filename, "exec"), scope)
    cfg = {k: v for k, v in scope.items()
           if not k.startswith("__") and not callable(v) and not hasattr(v, "__loader__")}
    if BASE_KEY in cfg:
        base = cfg.pop(BASE_KEY)
        base = base if isinstance(base, list) else [base]
        base_cfg = {}
        for b in base:
            sub = _load_py(osp.join(osp.dirname(filename), b))
            dup = set(base_cfg) & set(sub)
            # mmcv forbids duplicate keys across bases; samplers/models legitimately share none
            if dup:
                raise KeyError("Duplicate key is not allowed among bases: %s" % sorted(dup))
            base_cfg.update(sub)
        cfg = _merge_a_into_b(cfg, base_cfg)
    return cfg


class Config:
    def __init__(self, cfg_dict=None, filename=None):
        cfg_dict = {} if cfg_dict is None else cfg_dict
        if not isinstance(cfg_dict, dict):
            raise TypeError("cfg_dict must be a dict, but got %s" % type(cfg_dict))
        object.__setattr__(self, "_cfg_dict", _to_config_dict(cfg_dict))
        object.__setattr__(self, "_filename", filename)

    @staticmethod
    def fromfile(filename):
        return Config(_load_py(filename), filename=filename)

    @property
    def filename(self):
        return self._filename

    def merge_from_dict(self, options):
        """options: {'a.b.c': v} dotted keys (the --cfg-options form)."""
        option_cfg = {}
        for full_key, v in options.items():
            d = option_cfg
            keys = full_key.split(".")
            for sub in keys[:-1]:
                d = d.setdefault(sub, {})
            d[keys[-1]] = v
        merged = _merge_a_into_b(option_cfg, self._cfg_dict)
        object.__setattr__(self, "_cfg_dict", _to_config_dict(merged))

    def get(self, key, default=None):
        return self._cfg_dict.get(key, default)

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __setattr__(self, name, value):
        self._cfg_dict[name] = _to_config_dict(value)

    def __setitem__(self, name, value):
        self._cfg_dict[name] = _to_config_dict(value)

    def __contains__(self, name):
        return name in self._cfg_dict

    def __iter__(self):
        return iter(self._cfg_dict)

    def __len__(self):
        return len(self._cfg_dict)

    def to_dict(self):
        def plain(o):
            if isinstance(o, dict):
                return {k: plain(v) for k, v in o.items()}
            if isinstance(o, (list, tuple)):
                return type(o)(plain(v) for v in o)
            return o
        return plain(self._cfg_dict)

    @property
    def pretty_text(self):
        lines = []
        for k, v in self._cfg_dict.items():
            lines.append("%s = %r" % (k, self.to_dict()[k]))
        return "\n".join(lines) + "\n"

    def dump(self, file=None):
        text = self.pretty_text
        if file is None:
            return text
        with open(file, "w", encoding="utf-8") as f:
            f.write(text)

    def __repr__(self):
        return "Config (path: %s): %r" % (self._filename, self.to_dict())


class DictAction:
    """argparse action for ``--cfg-options k=v k2=[a,b]`` (same value grammar as mmcv's)."""

    @staticmethod
    def parse_value(val):
        for cast in (int, float):
            try:
                return cast(val)
            except ValueError:
                pass
        if val.lower() in ("true", "false"):
            return val.lower() == "true"
        if val == "None":
            return None
        if (val.startswith("[") and val.endswith("]")) or (val.startswith("(") and val.endswith(")")):
            inner = val[1:-1]
            items = [DictAction.parse_value(v.strip()) for v in inner.split(",") if v.strip()]
            return tuple(items) if val.startswith("(") else items
        if "," in val:
            return [DictAction.parse_value(v) for v in val.split(",")]
        return val.strip("'\"")

    @staticmethod
    def parse(pairs):
        out = {}
        for kv in pairs or []:
            k, v = kv.split("=", 1)
            out[k] = DictAction.parse_value(v)
        return out
