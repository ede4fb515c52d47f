"""Model samplers: the host-side mirror of ``gaiavision.model_space`` used by the supernet trainer.

The sampler classes are an absent dependency; their behaviour is reconstructed from the configs
that instantiate them (configs/_dynamic_/model_samplers/ar50to101v2.py:2-116) and the call sites
(tools/train_supernet.py:189-190 build; tools/extract_subnet.py:104-106 and
gaiaseg/core/evaluation/cross_arch_eval_hooks.py:53-59 ``traverse`` / ``anchor_name``) —
SURVEY.md Appendix A15.  A *meta* is a flat dict with dotted keys
(``'arch.backbone.body.width': [64,128,256,512]``); ``fold_dict`` turns it into the nested form
``manipulate_arch`` consumes.
"""
import copy
import random

from .registry import Registry, build_from_cfg

MODEL_SAMPLERS = Registry("model sampler")


def build_model_sampler(cfg):
    return build_from_cfg(cfg, MODEL_SAMPLERS)


class BaseModelSampler:
    def __init__(self, mode="sample"):
        self.mode = mode
        self._rng = random

    def set_mode(self, mode):
        assert mode in ("sample", "traverse")
        self.mode = mode

    def seed(self, seed):
        """Private RNG stream (default: the global ``random`` module, seeded by --seed)."""
        self._rng = random.Random(seed)
        for child in self.children():
            child.seed(self._rng.randrange(1 << 30))

    def children(self):
        return []

    def sample(self):
        raise NotImplementedError

    def traverse(self):
        raise NotImplementedError

    def __call__(self):
        return self.sample() if self.mode == "sample" else self.traverse()


@MODEL_SAMPLERS.register_module("anchor")
class AnchorSampler(BaseModelSampler):
    """anchors: list of metas. sample() = one uniformly; traverse() = all, in order."""

    def __init__(self, anchors, **kw):
        super().__init__(**kw)
        self.anchors = [dict(a) for a in anchors]

    def sample(self):
        return copy.deepcopy(self._rng.choice(self.anchors))

    def traverse(self):
        return copy.deepcopy(self.anchors)

    def anchor_name(self, idx):
        return self.anchors[idx].get("name", str(idx))

    def __len__(self):
        return len(self.anchors)


def _grid(start, end, step):
    vals, v = [], start
    while v <= end:
        vals.append(v)
        v += step
    return vals


@MODEL_SAMPLERS.register_module("range")
class RangeSampler(BaseModelSampler):
    """Uniform draw on {start, start+step, .., end}; per-element for list-valued ranges;
    ascending=True keeps the elements non-decreasing (redraw until satisfied)."""

    def __init__(self, key, start, end, step, ascending=False, **kw):
        super().__init__(**kw)
        self.key, self.start, self.end, self.step = key, start, end, step
        self.ascending = ascending
        self.is_list = isinstance(start, (list, tuple))
        if self.is_list:
            assert len(start) == len(end) == len(step)

    def _draw(self):
        if not self.is_list:
            return self._rng.choice(_grid(self.start, self.end, self.step))
        return [self._rng.choice(_grid(s, e, st)) for s, e, st in zip(self.start, self.end, self.step)]

    def sample(self):
        v = self._draw()
        if self.ascending and self.is_list:
            for _ in range(1000):
                if all(a <= b for a, b in zip(v, v[1:])):
                    break
                v = self._draw()
            else:
                v = sorted(v)
        return {self.key: v}

    def traverse(self):
        if not self.is_list:
            return [{self.key: v} for v in _grid(self.start, self.end, self.step)]
        out = [[]]
        for s, e, st in zip(self.start, self.end, self.step):
            out = [o + [v] for o in out for v in _grid(s, e, st)]
        if self.ascending:
            out = [o for o in out if all(a <= b for a, b in zip(o, o[1:]))]
        return [{self.key: o} for o in out]


@MODEL_SAMPLERS.register_module("candidate")
class CandidateSampler(BaseModelSampler):
    def __init__(self, key, candidates, **kw):
        super().__init__(**kw)
        self.key, self.candidates = key, list(candidates)

    def sample(self):
        return {self.key: copy.deepcopy(self._rng.choice(self.candidates))}

    def traverse(self):
        return [{self.key: copy.deepcopy(c)} for c in self.candidates]


class _Container(BaseModelSampler):
    def __init__(self, model_samplers, **kw):
        super().__init__(**kw)
        self.model_samplers = [build_model_sampler(c) if isinstance(c, dict) else c
                               for c in model_samplers]

    def children(self):
        return self.model_samplers

    def set_mode(self, mode):
        super().set_mode(mode)
        for c in self.model_samplers:
            c.set_mode(mode)


@MODEL_SAMPLERS.register_module("composite")
class CompositeSampler(_Container):
    """One draw of every child merged into one meta."""

    def sample(self):
        meta = {}
        for c in self.model_samplers:
            meta.update(c.sample())
        return meta

    def traverse(self):
        out = [{}]
        for c in self.model_samplers:
            out = [dict(o, **m) for o in out for m in c.traverse()]
        return out


@MODEL_SAMPLERS.register_module("concat")
class ConcatSampler(_Container):
    """Union of the children's candidates.  sample() returns the LIST of candidate metas of this
    step (every anchor + one draw of each random child), from which the hook picks one."""

    def candidates(self):
        out = []
        for c in self.model_samplers:
            if isinstance(c, AnchorSampler):
                out.extend(c.traverse())
            else:
                m = c.sample()
                out.extend(m if isinstance(m, list) else [m])
        return out

    def sample(self):
        return self._rng.choice(self.candidates())

    def traverse(self):
        out = []
        for c in self.model_samplers:
            out.extend(c.traverse())
        return out


@MODEL_SAMPLERS.register_module("repeat")
class RepeatSampler(BaseModelSampler):
    """``times`` independent draws of the wrapped sampler."""

    def __init__(self, times, model_sampler, **kw):
        super().__init__(**kw)
        self.times = times
        self.model_sampler = (build_model_sampler(model_sampler)
                              if isinstance(model_sampler, dict) else model_sampler)

    def children(self):
        return [self.model_sampler]

    def sample(self):
        return [self.model_sampler.sample() for _ in range(self.times)]

    def traverse(self):
        return self.model_sampler.traverse()


def arch_key(meta):
    """Hashable identity of the architecture part of a meta (cache key for plans / graphs)."""
    def freeze(v):
        if isinstance(v, (list, tuple)):
            return tuple(freeze(x) for x in v)
        if isinstance(v, dict):
            return tuple(sorted((k, freeze(x)) for k, x in v.items()))
        return v
    return tuple(sorted((k, freeze(v)) for k, v in meta.items() if k.startswith("arch")))
