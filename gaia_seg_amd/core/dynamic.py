"""DynamicMixin and the arch-meta wire format (gaiavision.core contract).

Evidence for the contract in the reference tree: every dynamic class declares ``search_space``
and one ``manipulate_<key>`` per key (gaiaseg/models/segmentors/dynamic_encoder_decoder.py:11,31-42;
gaiaseg/models/backbones/dynamic_resnet.py:67,381-403; gaiaseg/models/utils/dynamic_res_layer.py:34,149-157),
states are stored as ``<key>_state`` (dynamic_resnet.py:69-77) and ``deploy()`` flags modules with
``_deploying`` (tools/extract_subnet.py:94, dynamic_res_layer.py:167).
"""


class DynamicMixin:
    search_space = set()

    def init_state(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, "%s_state" % k, v)

    def manipulate_arch(self, arch_meta):
        """Dispatch each key of ``arch_meta`` to ``self.manipulate_<key>(value)``."""
        if not isinstance(arch_meta, dict):
            raise TypeError("arch_meta must be a dict, got %s" % type(arch_meta))
        for k, v in arch_meta.items():
            if k not in self.search_space:
                raise KeyError("%s: key '%s' is not in the search space %s"
                               % (type(self).__name__, k, sorted(self.search_space)))
            getattr(self, "manipulate_%s" % k)(v)

    def state_dict_of_arch(self):
        return {k: getattr(self, "%s_state" % k, None) for k in self.search_space}

    def deploy(self, mode=True):
        """Deploy mode: the next forward physically prunes to the current subnet."""
        self._deploying = mode
        modules = getattr(self, "modules", None)
        if modules is not None:
            for m in self.modules():
                if isinstance(m, DynamicMixin):
                    m._deploying = mode
        return self


def unzip_meta(meta):
    """Per-child view of a dict of equally long lists:
    {'width': [a, b], 'depth': [c, d]} -> [{'width': a, 'depth': c}, {'width': b, 'depth': d}]
    (what DynamicResNet hands to its stem convs / stages, dynamic_resnet.py:381-403)."""
    keys = list(meta)
    return [dict(zip(keys, vals)) for vals in zip(*(meta[k] for k in keys))]


def freeze(module):
    """Take a module out of training: eval mode (BatchNorm uses its running statistics) and no
    gradients for its parameters."""
    module.eval()
    for p in module.parameters():
        p.requires_grad = False


def fold_dict(flat, sep="."):
    """{'arch.backbone.body.width': [..]} -> {'arch': {'backbone': {'body': {'width': [..]}}}}."""
    out = {}
    for key, v in flat.items():
        parts = key.split(sep)
        d = out
        for p in parts[:-1]:
            nxt = d.get(p)
            if not isinstance(nxt, dict):
                nxt = {}
                d[p] = nxt
            d = nxt
        d[parts[-1]] = v
    return out


def unfold_dict(nested, sep=".", _prefix=""):
    """Inverse of fold_dict."""
    out = {}
    for k, v in nested.items():
        key = "%s%s%s" % (_prefix, sep, k) if _prefix else str(k)
        if isinstance(v, dict) and v:
            out.update(unfold_dict(v, sep, key))
        else:
            out[key] = v
    return out
