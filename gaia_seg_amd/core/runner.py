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
    in_graph = False   # True: after_train_iter only enqueues device work and may be graph-captured

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


def _ranges_subtract(a, b):
    """a minus b for sorted lists of disjoint [begin, end) ranges."""
    out, j = [], 0
    for lo, hi in a:
        cur = lo
        while j < len(b) and b[j][1] <= cur:
            j += 1
        k = j
        while k < len(b) and b[k][0] < hi:
            if b[k][0] > cur:
                out.append((cur, b[k][0]))
            cur = max(cur, b[k][1])
            k += 1
        if cur < hi:
            out.append((cur, hi))
    return out


def _ranges_intersect(a, b):
    return _ranges_subtract(a, _ranges_subtract(a, b))


class ArenaOptimizerHook(Hook):
    """OptimizerHook for the flat-arena SGD: the step touches only the active subnet's ranges.

    Three instalments, each as early as its gradients are final:
      * stages 3.. of the backbone and the heads (~95 % of the parameters) on the optimizer stream, in
        the MIDDLE of backward — when the replay crosses the backbone's "stage2|stage3" mark.
        Opt-in (GS_EARLY_SGD=1): measured neutral — the end of the step is set by the total work of
        backward, not by where the optimizer's HBM traffic sits (profiles/r04_stream_experiments.md);
        without it these ranges are updated with stage 2's;
      * stage 2 once the weight-gradient stream has passed the checkpoint behind stage 1;
      * the stem and stage 1 after the weight-gradient stream has drained."""

    in_graph = True   # backward + SGD are part of a captured step graph (IterBasedRunner)
    EARLY = os.environ.get("GS_EARLY_SGD", "0") == "1"   # measured neutral (profiles/r04_stream_experiments.md): opt-in

    def __init__(self, grad_clip=None):
        if grad_clip is not None:
            raise NotImplementedError("grad_clip is not configured by the in-tree configs")

    def _early_step(self, runner, tag, done):
        """Backward has crossed ``tag``: update what is final, on the optimizer stream."""
        from ..hip import ops
        if tag != "stage2|stage3" or done:
            return
        ranges = runner.early_ranges()
        if not ranges:
            return
        dev = runner.arena.device
        opt = ops.opt_stream(dev)
        ops.fork_to(opt, dev)          # weight gradients (side stream), BN gradients (this stream, branches)
        with torch.cuda.stream(opt):
            covered = runner.reducer.wait_launched()   # multi-GPU: what has been all-reduced so far
            if covered is not None:
                ranges = _ranges_intersect(ranges, sorted(covered))
            if ranges:
                runner.arena.sgd_step(ranges, runner.lr, runner.momentum, runner.weight_decay,
                                      1.0 / gdist.world_size(), True, hyper=runner.hyper)
        done.extend(ranges)

    def after_train_iter(self, runner):
        from ..hip import ops
        prof = runner.host_prof
        t0 = time.perf_counter() if prof is not None else 0.0
        ops.SIDE_CHECKPOINT = None
        ops.DEFER_JOIN = True          # the tapes hand their weight gradients over but do not join
        runner.mark("fwd_end")
        done = []
        if self.EARLY and runner.arena.device.type == "cuda" and not torch.cuda.is_current_stream_capturing():
            ops.BACKWARD_MARK_CB = lambda tag: self._early_step(runner, tag, done)
        try:
            runner.outputs["loss"].backward()
        finally:
            ops.DEFER_JOIN = False
            ops.BACKWARD_MARK_CB = None
        t1 = time.perf_counter() if prof is not None else 0.0
        runner.mark("bwd_end_main")
        for i, sd in enumerate(ops._side_streams.get((runner.arena.device.type, runner.arena.device.index), ())):
            runner.mark("bwd_end_side%d" % i, sd)
        ops.join_branch_streams()      # (auxiliary head / shortcut work on the branch stream)
        runner.reducer.finish()
        scale = 1.0 / gdist.world_size()
        early, late = runner.split_ranges()
        if done:
            done.sort()
            early, late = _ranges_subtract(early, done), _ranges_subtract(late, done)
            runner.early_steps += 1
        ck = ops.SIDE_CHECKPOINT
        if ck is not None and early:
            # gradients of everything behind the checkpoint are final once the side stream has passed
            # it: update those parameters while the stem / stage-1 weight gradients still run
            for ev in ck:
                torch.cuda.current_stream().wait_event(ev)
            runner.arena.sgd_step(early, runner.lr, runner.momentum, runner.weight_decay, scale, True,
                                  hyper=runner.hyper)
            ops.join_side_streams()
            runner.arena.sgd_step(late, runner.lr, runner.momentum, runner.weight_decay, scale, True,
                                  hyper=runner.hyper)
        else:
            ops.join_side_streams()
            runner.arena.sgd_step(runner.active_ranges, runner.lr, runner.momentum,
                                  runner.weight_decay, scale, True, hyper=runner.hyper)
        if done:   # the next forward reads those parameters on this stream
            ops.join_from(ops.opt_stream(runner.arena.device), runner.arena.device)
        # the step cleared exactly the ranges backward wrote: the next zero_grad has nothing to do
        runner.arena.grads_clean = True
        runner.mark("step_end")
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


class _StepGraph:
    __slots__ = ("graph", "static", "outputs", "counters")

    def __init__(self, graph, static, outputs, counters):
        self.graph, self.static, self.outputs, self.counters = graph, static, outputs, counters


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
        self._early_cache = {}
        self.early_steps = 0           # optimizer steps whose first instalment ran inside backward
        self.step_events = {} if os.environ.get("GS_STEP_EVENTS") else None
        self._active_cache = {}
        # GS_HOST_PROF=1: accumulate host-side seconds per phase of train_iter (diagnostics)
        self.host_prof = {} if os.environ.get("GS_HOST_PROF") else None
        # ---- step graphs (see train_iter) ----
        # off by default: on ROCm 7.2 hipGraphLaunch spends as much host time per kernel node as the
        # eager path spends per launch (r02: 8.3 ms to launch the 560-node R50 step graph against
        # 8.0 ms of eager host work), so a replay neither frees the host nor closes launch gaps
        self.graphs_enabled = os.environ.get("GS_STEP_GRAPH", "0") == "1"
        self.graphs_paused = False     # e.g. while HIP-event timers are recorded inside the step
        self.max_graphs = int(os.environ.get("GS_STEP_GRAPH_MAX", "8"))
        self._graphs = OrderedDict()   # graph key -> _StepGraph (LRU)
        self._arch_seen = {}           # arch key -> eager steps run with it
        self.hyper = None              # device {lr, momentum, weight_decay, grad_scale} (graphs on)
        self.graph_stats = {"captured": 0, "replayed": 0, "eager": 0}
        self.set_arch(None)

    def register_hook(self, hook):
        self.hooks.append(hook)

    # ---- GS_STEP_EVENTS=1 (diagnostics): HIP events at the phase boundaries of every step, on the
    # stream the phase ends on; bench.py prints the mean offsets from the step's start.  Unlike a
    # profiler trace this does not slow the host down, so it shows the real critical path.
    def mark(self, name, stream=None):
        ev = self.step_events
        if ev is None:
            return
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream if stream is not None else torch.cuda.current_stream())
        ev.setdefault(name, []).append(e)

    def step_event_summary(self):
        """{phase: mean ms from 'step_begin'} over the recorded steps (call after a synchronize)."""
        ev = self.step_events
        if not ev or "step_begin" not in ev:
            return {}
        n = len(ev["step_begin"])
        out = {}
        for name, lst in ev.items():
            if name == "step_begin" or len(lst) != n:
                continue
            out[name] = sum(b.elapsed_time(e) for b, e in zip(ev["step_begin"], lst)) / n
        return out

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

    def refresh_active(self, force=False):
        """Recompute the active parameter sets from the model's CURRENT arch state.  Frozen
        parameters (frozen_stages / frozen_layers / norm_cfg requires_grad=False) are left out of
        the zero / all-reduce / SGD ranges: torch.optim.SGD skips parameters without a gradient,
        so they must neither decay nor move (gaiaseg/models/backbones/dynamic_resnet.py:304-334).
        The sets of a named / sampled subnet are cached by its arch key (the module walk costs
        ~0.4 ms per step otherwise); ``force`` drops the cache (requires_grad flags were edited)."""
        key = self.arch_key if self.arch_key != ("current",) else None
        if force:
            self._active_cache.clear()
        hit = self._active_cache.get(key) if key is not None else None
        if hit is None:
            active = self.model.active_parameters()
            hit = (active, [p for p in active if p.requires_grad])
            if key is not None:
                if len(self._active_cache) > 1024:
                    self._active_cache.clear()
                self._active_cache[key] = hit
        self.active_params, self.trainable_params = hit
        self.active_ranges = self.arena.ranges_for(self.trainable_params, key)

    def early_ranges(self):
        """Arena ranges of the trainable parameters whose gradients are final when backward crosses
        the backbone's "stage2|stage3" mark: stages 3.. and everything outside the backbone."""
        key = self.arch_key if self.arch_key != ("current",) else None
        cached = self._early_cache.get(key) if key is not None else None
        if cached is not None:
            return cached
        bb = getattr(self.model, "backbone", None)
        fn = getattr(bb, "early_gradient_parameters", None)
        if fn is None:
            out = []
        else:
            bb_ids = {id(p) for p in bb.parameters()}
            early_ids = {id(p) for p in fn()}
            out = self.arena.ranges_for([p for p in self.trainable_params
                                         if id(p) in early_ids or id(p) not in bb_ids])
        if key is not None:
            if len(self._early_cache) > 1024:
                self._early_cache.clear()
            self._early_cache[key] = out
        return out

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

    # ------------------------------------------------------------------------------------------
    # Step graphs.  A training step of a given subnet on a given batch shape is a fixed sequence of
    # ~600-1100 kernel launches (forward, backward, SGD) whose only step-dependent inputs are the
    # batch and the learning rate.  For subnets that come back (the named anchors of the train
    # sampler, or any subnet seen before) the whole step is captured ONCE into a HIP graph
    # (torch.cuda.CUDAGraph: stream capture of the launches the C-ABI makes on torch's current
    # stream and on the weight-gradient side stream) and replayed afterwards: the batch is copied
    # into the graph's static input tensors, lr / momentum / weight decay sit in a 16-byte device
    # buffer the SGD kernel reads (gs_sgd_step_hyper), and one graph launch replaces the host-side
    # walk over the modules.  Same kernels, same order, same results as the eager step (bit-identical:
    # tests/test_runner_gpu.py).  Never-repeating random subnets, multi-rank runs (RCCL launches are
    # left out of capture), SyncBN groups and diagnostic traces keep the eager path.
    # MEASURED (r02, MI355X, ROCm 7.2): the graph launch itself costs the host ~15 us per kernel node,
    # i.e. as much as the eager path; R50 192.9 img/s replayed vs 198.5 eager, the sampled mix loses
    # 40 % (switching between large graphs).  The feature is therefore OPT-IN (GS_STEP_GRAPH=1).
    # ------------------------------------------------------------------------------------------
    def _graph_key(self, data_batch):
        from ..hip import ops
        if (not self.graphs_enabled or self.graphs_paused or self.arch_key is None
                or self.arch_key == ("current",) or gdist.world_size() != 1
                or ops.RELU_TRACE is not None or ops.POOL_TRACE is not None or ops.TIMER is not None):
            return None
        sig = []
        for k in sorted(data_batch):
            v = data_batch[k]
            if torch.is_tensor(v):
                if not v.is_cuda:
                    return None
                sig.append((k, tuple(v.shape), v.dtype))
        return (self.arch_key, tuple(sig), self.model.training)

    def _write_hyper(self):
        """lr / momentum / weight decay / gradient scale of THIS step -> the device buffer."""
        from ..hip import lib as _lib
        from ..hip.runtime import current_stream_ptr
        if self.hyper is None:
            self.hyper = torch.zeros(4, dtype=torch.float32, device=self.arena.flat_param.device)
        _lib.check(_lib.load().gs_sgd_set_hyper(self.hyper.data_ptr(), self.lr, self.momentum,
                                                self.weight_decay, 1.0 / gdist.world_size(),
                                                current_stream_ptr()), "gs_sgd_set_hyper")

    def _after_hooks(self, in_graph):
        for h in self.hooks:
            if bool(getattr(h, "in_graph", False)) == in_graph:
                h.after_train_iter(self)

    def _capture_step(self, key, data_batch):
        """Capture forward + backward + SGD of the current subnet into a graph, then run it once."""
        from ..hip import ops, runtime
        dev = self.arena.flat_param.device
        static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in data_batch.items()}
        ops.reserve_workspaces(dev)   # no workspace may be (re)allocated inside the capture
        graph = torch.cuda.CUDAGraph()
        runtime.CAPTURE_LOG = []
        try:
            with torch.cuda.graph(graph):
                self.outputs = self.model.train_step(static, None)
                self._after_hooks(True)
            counters = runtime.CAPTURE_LOG
        finally:
            runtime.CAPTURE_LOG = None
        entry = _StepGraph(graph, static, self.outputs, counters)
        self._graphs[key] = entry
        while len(self._graphs) > self.max_graphs:
            self._graphs.popitem(last=False)
        self.graph_stats["captured"] += 1
        graph.replay()                 # the capture only recorded the step: this performs it
        return entry

    def _replay_step(self, entry, data_batch):
        for k, v in entry.static.items():
            if torch.is_tensor(v):
                v.copy_(data_batch[k], non_blocking=True)
        entry.graph.replay()
        for m in entry.counters:       # host-side effects of the step that the graph cannot carry
            m._nbt_pending = getattr(m, "_nbt_pending", 0) + 1
        self.outputs = entry.outputs
        self.arena.grads_clean = True  # (the captured SGD step clears the gradients it consumed)
        self.graph_stats["replayed"] += 1

    def prepare_graphs(self, metas, data_batch):
        """Capture the step graphs of the given subnets ahead of time (e.g. the sampler's anchors at
        start-up) instead of at their first occurrence.  Every capture performs one real training
        step on ``data_batch``; the architecture in place before the call is restored."""
        keep = self.arch_meta
        done = 0
        for meta in metas:
            self.set_arch(meta)
            gkey = self._graph_key(data_batch)
            if gkey is None or gkey in self._graphs:
                continue
            for eager in ((True, False) if self.graph_stats["eager"] == 0 else (False,)):
                self._write_hyper()
                self.arena.zero_grad(self.active_ranges, trust_clean=True)
                self.arena.grads_clean = False
                self.reducer.begin(self.trainable_params, self.arch_key)
                if eager:
                    self.outputs = self.model.train_step(data_batch, None)
                    self._after_hooks(True)
                    self.graph_stats["eager"] += 1
                else:
                    self._capture_step(gkey, data_batch)
                    done += 1
        if keep is not None:
            self.set_arch(keep)
        return done

    # The training step runs on a HIGH-PRIORITY stream (GS_TRAIN_PRIORITY=0: on the caller's stream):
    # the weight-gradient stream, the optimizer stream and the gradient exchange keep normal priority,
    # so when both have workgroups ready the dispatcher serves the chain every later kernel waits for
    # — data gradients, BatchNorm passes — first and the weight gradients fill in behind.  Same
    # kernels, same order on every stream, same results (tests/test_runner_gpu.py); r04 A/B on the
    # sampled mix: 166.4 -> 169.9 images/s (five interleaved runs each, the two sets do not overlap),
    # R50 unchanged (profiles/r04_stream_experiments.md).
    # The stream becomes the CURRENT stream of the calling thread at the first train_iter and stays it
    # (after waiting for what the caller had queued on its own stream): going back and forth between
    # the legacy default stream and this one every step made every launch 3x as expensive on the host
    # (45 us per conv + BN call instead of 15: measured, 166 -> 83 images/s).
    TRAIN_PRIORITY = os.environ.get("GS_TRAIN_PRIORITY", "1") != "0"

    def _enter_priority_stream(self):
        dev = self.arena.device
        if not self.TRAIN_PRIORITY or dev.type != "cuda" or torch.cuda.is_current_stream_capturing():
            return
        hp = self.__dict__.get("_hp_stream")
        if hp is None:
            hp = self._hp_stream = torch.cuda.Stream(device=dev, priority=-1)
        cur = torch.cuda.current_stream(dev)
        if cur != hp:
            hp.wait_stream(cur)
            torch.cuda.set_stream(hp)

    def train_iter(self, data_batch):
        self._enter_priority_stream()
        return self._train_iter(data_batch)

    def _train_iter(self, data_batch):
        prof = self.host_prof
        t0 = time.perf_counter() if prof is not None else 0.0
        self.mark("step_begin")
        if not self.model.training:   # (a full module walk: ~1.5 ms of host time per call)
            self.model.train()
        self.call_hook("before_train_iter")
        t1 = time.perf_counter() if prof is not None else 0.0
        gkey = self._graph_key(data_batch)
        if self.graphs_enabled and gdist.world_size() == 1:
            self._write_hyper()
        entry = self._graphs.get(gkey) if gkey is not None else None
        t2 = t1
        if entry is not None:
            self._graphs.move_to_end(gkey)
            self._replay_step(entry, data_batch)
            t3 = time.perf_counter() if prof is not None else 0.0
        else:
            self.arena.zero_grad(self.active_ranges, trust_clean=True)   # (a no-op after a clearing optimizer step)
            self.arena.grads_clean = False             # backward is about to write gradients
            self.reducer.begin(self.trainable_params,
                               self.arch_key if self.arch_key != ("current",) else None)
            t2 = time.perf_counter() if prof is not None else 0.0
            # capture at first sight what is known to come back (named anchors), anything else the
            # second time it shows up; the very first step of a process always runs eagerly (module
            # loading and lazily created handles must not happen inside a capture)
            known = self.arch_name != "random" or self._arch_seen.get(self.arch_key, 0) > 0
            if gkey is not None and known and self.graph_stats["eager"] > 0:
                self._capture_step(gkey, data_batch)
                t3 = time.perf_counter() if prof is not None else 0.0
            else:
                self.outputs = self.model.train_step(data_batch, None)
                t3 = time.perf_counter() if prof is not None else 0.0
                self._after_hooks(True)
                self.graph_stats["eager"] += 1
                if gkey is not None:
                    self._arch_seen[self.arch_key] = self._arch_seen.get(self.arch_key, 0) + 1
                    if len(self._arch_seen) > 4096:
                        self._arch_seen.clear()
        self._after_hooks(False)
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
