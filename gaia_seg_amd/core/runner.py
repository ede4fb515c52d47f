"""Iteration-based training loop with hook points (mmcv IterBasedRunner contract) and the hooks
the supernet trainer registers (gaiaseg/apis/train.py:115-186):

  PolyLrUpdaterHook   lr_config = dict(policy='poly', power=0.9, min_lr=1e-4, by_epoch=False)
                      (configs/_dynamic_/models/pspnet_ar50to101v2_gsync.py:177)
  ArenaOptimizerHook  zero_grad -> loss.backward() -> (RCCL bucket all-reduce) -> fused SGD step
  ManipulateArchHook  gaivision hook (gaiaseg/apis/train.py:142-146): before every train iteration
                      sample a meta, make it identical on all ranks, manipulate_arch
  TextLoggerHook / CheckpointHook
"""
import os
import time
from collections import OrderedDict

import torch

from . import dist as gdist
from .dynamic import fold_dict
from .model_space import arch_key


class Hook:
    def before_run(self, runner):
        pass

    def after_run(self, runner):
        pass

    def before_train_iter(self, runner):
        pass

    def after_train_iter(self, runner):
        pass

    def every_n_iters(self, runner, n):
        return (runner.iter + 1) % n == 0 if n > 0 else False


class ManipulateArchHook(Hook):
    """One subnet per iteration: rank 0 draws a meta from the train sampler, the draw is broadcast
    (every rank must run the same subnet: the gradient buckets assume it), then
    ``model.manipulate_arch(fold_dict(meta)['arch'])`` (DECIDE)."""

    def __init__(self, sampler):
        self.sampler = sampler
        self.history = []

    def before_train_iter(self, runner):
        meta = self.sampler.sample() if gdist.rank() == 0 else None
        meta = gdist.broadcast_object(meta, src=0)
        runner.set_arch(meta)
        self.history.append(meta.get("name", "random"))


class PolyLrUpdaterHook(Hook):
    def __init__(self, power=1.0, min_lr=0.0, by_epoch=False, **unused):
        self.power, self.min_lr = power, min_lr
        self.base_lr = None

    def before_run(self, runner):
        self.base_lr = runner.base_lr

    def get_lr(self, runner):
        coeff = (1 - runner.iter / runner.max_iters) ** self.power
        return (self.base_lr - self.min_lr) * coeff + self.min_lr

    def before_train_iter(self, runner):
        runner.lr = self.get_lr(runner)


class FixedLrUpdaterHook(Hook):
    def before_train_iter(self, runner):
        runner.lr = runner.base_lr


class ArenaOptimizerHook(Hook):
    """OptimizerHook for the flat-arena SGD: the step touches only the active subnet's ranges."""

    def __init__(self, grad_clip=None):
        if grad_clip is not None:
            raise NotImplementedError("grad_clip is not configured by the in-tree configs")

    def after_train_iter(self, runner):
        from ..hip import ops
        prof = runner.host_prof
        t0 = time.perf_counter() if prof is not None else 0.0
        ops.SIDE_CHECKPOINT = None
        ops.DEFER_JOIN = True          # the tapes hand their weight gradients over but do not join
        try:
            runner.outputs["loss"].backward()
        finally:
            ops.DEFER_JOIN = False
        t1 = time.perf_counter() if prof is not None else 0.0
        runner.reducer.finish()
        scale = 1.0 / gdist.world_size()
        early, late = runner.split_ranges()
        ck = ops.SIDE_CHECKPOINT
        if ck is not None and early:
            # gradients of everything behind the checkpoint are final once the side stream has passed
            # it: update those parameters while the stem / stage-1 weight gradients still run
            torch.cuda.current_stream().wait_event(ck)
            runner.arena.sgd_step(early, runner.lr, runner.momentum, runner.weight_decay, scale, True)
            ops.join_side_streams()
            runner.arena.sgd_step(late, runner.lr, runner.momentum, runner.weight_decay, scale, True)
        else:
            ops.join_side_streams()
            runner.arena.sgd_step(runner.active_ranges, runner.lr, runner.momentum,
                                  runner.weight_decay, scale, True)
        # the step cleared exactly the ranges backward wrote: the next zero_grad has nothing to do
        runner.arena.grads_clean = True
        if prof is not None:
            prof["backward"] = prof.get("backward", 0.0) + (t1 - t0)
            prof["finish+sgd"] = prof.get("finish+sgd", 0.0) + (time.perf_counter() - t1)


class TextLoggerHook(Hook):
    def __init__(self, interval=50, by_epoch=False, logger=None, **unused):
        self.interval = interval
        self.logger = logger
        self._t0 = None

    def before_run(self, runner):
        self._t0 = time.time()

    def after_train_iter(self, runner):
        if not self.every_n_iters(runner, self.interval):
            return
        lv = runner.outputs["log_vars"]
        items = ", ".join("%s: %.4f" % (k, float(v)) for k, v in lv.items())
        dt = (time.time() - self._t0) / self.interval
        self._t0 = time.time()
        msg = "Iter [%d/%d]\tlr: %.3e, arch: %s, time: %.3f, %s" % (
            runner.iter + 1, runner.max_iters, runner.lr, runner.arch_name, dt, items)
        if gdist.rank() == 0:
            (self.logger.info if self.logger else print)(msg)


class CheckpointHook(Hook):
    def __init__(self, interval=-1, by_epoch=False, out_dir=None, **unused):
        self.interval, self.out_dir = interval, out_dir

    def after_train_iter(self, runner):
        if self.interval > 0 and self.every_n_iters(runner, self.interval) and gdist.rank() == 0:
            from .checkpoint import save_checkpoint
            out_dir = self.out_dir or runner.work_dir
            os.makedirs(out_dir, exist_ok=True)
            save_checkpoint(runner.model, os.path.join(out_dir, "iter_%d.pth" % (runner.iter + 1)),
                            optimizer=runner.arena, meta=dict(runner.meta or {}, iter=runner.iter + 1))


class IterBasedRunner:
    """``run(data_loaders, workflow)`` drives ``model.train_step`` for ``max_iters`` iterations."""

    def __init__(self, model, arena, reducer, base_lr=0.01, momentum=0.9, weight_decay=5e-4,
                 max_iters=80000, work_dir=None, logger=None, meta=None):
        self.model, self.arena, self.reducer = model, arena, reducer
        self.base_lr = self.lr = base_lr
        self.momentum, self.weight_decay = momentum, weight_decay
        self.max_iters = max_iters
        self.work_dir, self.logger, self.meta = work_dir, logger, meta
        self.iter = 0
        self.hooks = []
        self.outputs = None
        self.arch_name = "supernet"
        self.active_params = None      # parameters the current subnet uses
        self.trainable_params = None   # ... of those, the ones that receive gradients / updates
        self.active_ranges = None      # merged arena ranges of trainable_params
        self.arch_key = None
        self.arch_meta = None
        self._split_cache = {}
        # GS_HOST_PROF=1: accumulate host-side seconds per phase of train_iter (diagnostics)
        self.host_prof = {} if os.environ.get("GS_HOST_PROF") else None
        self.set_arch(None)

    def register_hook(self, hook):
        self.hooks.append(hook)

    def call_hook(self, name):
        for h in self.hooks:
            getattr(h, name)(self)

    @property
    def raw_model(self):
        return self.model

    def set_arch(self, meta):
        """Apply a sampled meta (None = keep the current, max, architecture)."""
        if meta is not None:
            self.model.manipulate_arch(fold_dict(meta)["arch"])
            self.arch_name = meta.get("name", "random")
            self.arch_key = arch_key(meta)
            self.arch_meta = meta
        else:
            self.arch_key = ("current",)
        self.refresh_active()

    def refresh_active(self):
        """Recompute the active parameter sets from the model's CURRENT arch state.  Frozen
        parameters (frozen_stages / frozen_layers / norm_cfg requires_grad=False) are left out of
        the zero / all-reduce / SGD ranges: torch.optim.SGD skips parameters without a gradient,
        so they must neither decay nor move (gaiaseg/models/backbones/dynamic_resnet.py:304-334)."""
        self.active_params = self.model.active_parameters()
        self.trainable_params = [p for p in self.active_params if p.requires_grad]
        key = self.arch_key if self.arch_key != ("current",) else None
        self.active_ranges = self.arena.ranges_for(self.trainable_params, key)

    def split_ranges(self):
        """(early, late) parts of active_ranges: `late` covers the parameters whose weight gradients
        are produced last in backward (backbone.late_gradient_parameters), `early` the rest."""
        key = self.arch_key if self.arch_key != ("current",) else None
        cached = self._split_cache.get(key) if key is not None else None
        if cached is not None:
            return cached
        late_fn = getattr(getattr(self.model, "backbone", None), "late_gradient_parameters", None)
        late_ids = {id(p) for p in late_fn()} if late_fn is not None else set()
        early_p = [p for p in self.trainable_params if id(p) not in late_ids]
        late_p = [p for p in self.trainable_params if id(p) in late_ids]
        out = (self.arena.ranges_for(early_p), self.arena.ranges_for(late_p))
        if key is not None:
            self._split_cache[key] = out
        return out

    def train_iter(self, data_batch):
        prof = self.host_prof
        t0 = time.perf_counter() if prof is not None else 0.0
        if not self.model.training:   # (a full module walk: ~1.5 ms of host time per call)
            self.model.train()
        self.call_hook("before_train_iter")
        t1 = time.perf_counter() if prof is not None else 0.0
        self.arena.zero_grad(self.active_ranges)   # (a no-op after a clearing optimizer step)
        self.arena.grads_clean = False             # backward is about to write gradients
        self.reducer.begin(self.trainable_params,
                           self.arch_key if self.arch_key != ("current",) else None)
        t2 = time.perf_counter() if prof is not None else 0.0
        self.outputs = self.model.train_step(data_batch, None)
        t3 = time.perf_counter() if prof is not None else 0.0
        self.call_hook("after_train_iter")
        if prof is not None:
            t4 = time.perf_counter()
            for k, v in (("hooks_before", t1 - t0), ("zero+begin", t2 - t1), ("forward", t3 - t2),
                         ("backward+opt", t4 - t3)):
                prof[k] = prof.get(k, 0.0) + v
            prof["iters"] = prof.get("iters", 0) + 1
        self.iter += 1
        return self.outputs

    def run(self, data_loaders, workflow=(("train", 1),), max_iters=None):
        if max_iters is not None:
            self.max_iters = max_iters
        self.call_hook("before_run")
        loader = iter(data_loaders[0])
        while self.iter < self.max_iters:
            try:
                batch = next(loader)
            except StopIteration:
                loader = iter(data_loaders[0])
                batch = next(loader)
            self.train_iter(batch)
        self.call_hook("after_run")

    def resume(self, checkpoint):
        from .checkpoint import load_checkpoint
        ck = load_checkpoint(self.model, checkpoint, strict=True)
        if "optimizer" in ck:
            self.arena.load_state_dict(ck["optimizer"], logger=self.logger)
        self.iter = ck.get("meta", {}).get("iter", 0)

    def load_checkpoint(self, checkpoint):
        from .checkpoint import load_checkpoint
        load_checkpoint(self.model, checkpoint, strict=False)
