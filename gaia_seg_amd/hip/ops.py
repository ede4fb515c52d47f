"""Tape-aware operators: each function launches the forward HIP kernel(s) through the C-ABI and
records the closure that launches the backward kernel(s).

Nothing in this file computes with torch ops on the data path; torch is used for allocation,
stream handles and (SyncBN only) the tiny statistics exchange over RCCL.
"""
import ctypes

import torch

import os

from . import lib as _lib
from .runtime import WORKSPACE, Act, _Workspace, current_stream_ptr, round_up

_ws = WORKSPACE

# ---- weight gradients on a side stream -------------------------------------------------------
# In backward the weight gradient of a conv is needed only by the optimizer (and the gradient
# all-reduce), while the data gradient is on the critical path.  The wgrad kernels (MFMA-bound) are
# therefore launched on a second HIP stream, fenced by events, and run concurrently with the
# BN-backward / dgrad kernels of the following layers (HBM-bound) on the main stream.  The main
# stream joins the side stream at the end of every tape backward and before a gradient bucket is
# handed to RCCL.  GS_SIDE_WGRAD=0 keeps everything on one stream.
SIDE_WGRAD = os.environ.get("GS_SIDE_WGRAD", "1") != "0"
# Weight-gradient streams per device.  r04: the events of the un-profiled step (GS_STEP_EVENTS) put the
# end of the weight-gradient stream 0.45-0.52 ms BEHIND the end of backward on the training stream —
# it is the critical path of the step's last half millisecond, where the stage-1 / stem weight
# gradients (K = 65 536 pixels, a few output tiles, each followed by its slab reduce) run one after the
# other.  GS_SIDE_STREAMS = N deals the jobs round-robin to N streams, so that one job's slab reduce
# (HBM-bound) runs beside the next job's contraction (MFMA-bound).  MEASURED (r04, A/B/A/B on one box,
# profiles/r04_stream_experiments.md): R50 9.54 / 9.63 / 9.68 ms and the sampled mix 12.49 / 12.59 /
# 12.73 ms per step for N = 1 / 2 / 3 — the end of the step does not move with the queue layout: with
# two or three kernels resident the chip is throughput-bound during backward, more streams only add
# contention for the training stream.  Default 1.
N_SIDE = max(1, int(os.environ.get("GS_SIDE_STREAMS", "1")))
_side_streams = {}     # (device type, index) -> [torch.cuda.Stream] * N_SIDE
_side_dirty = {}
_side_next = {}        # round-robin position per device
_ws_side = _Workspace()


SIDE_PRIORITY = int(os.environ.get("GS_SIDE_PRIORITY", "0"))   # A/B knob: priority of the weight-gradient stream(s)


def _side_list(dev):
    key = (dev.type, dev.index)
    lst = _side_streams.get(key)
    if lst is None:
        # (stream priorities made no difference, r01 A/B)
        lst = _side_streams[key] = [torch.cuda.Stream(device=dev, priority=SIDE_PRIORITY) for _ in range(N_SIDE)]
    return lst


def _side_stream(dev):
    """The primary weight-gradient stream (gradient buckets are issued in its context)."""
    return _side_list(dev)[0]


# Weight gradients are handed to the side stream in BATCHES: every hand-over costs one event on the
# main stream (hipEventRecord puts a marker packet into the queue; the r02 kernel trace shows a
# ~7 us bubble behind each of them, 57 per R50 step).  A conv's backward therefore only queues its
# weight-gradient job; after WGRAD_BATCH jobs (or at the end of a tape, or before a gradient bucket
# is all-reduced) one event covers them all and the jobs are launched on the side stream.  The
# "gradient ready" notifications of the reducer fire when the job is actually enqueued.
WGRAD_BATCH = max(1, int(os.environ.get("GS_WGRAD_BATCH", "4")))
_wgrad_jobs = []
_wgrad_stream = None   # raw handle of the stream the queued jobs' operands were produced on


_side_keep = []   # (dy, x) of weight-gradient kernels launched on the side stream, until the join


def queue_wgrad(d, x, dy, weight, gw, need, dev):
    """x: Act (kept alive until the launch), dy: gradient of the conv output, gw: weight.grad."""
    global _wgrad_stream
    st = current_stream_ptr()
    if _wgrad_jobs and st != _wgrad_stream:
        flush_wgrads()     # one event covers one batch: a batch never mixes producer streams
    _wgrad_stream = st
    _wgrad_jobs.append((d, x, dy, weight, gw, need, dev))
    if len(_wgrad_jobs) >= WGRAD_BATCH:
        flush_wgrads()


def flush_wgrads():
    """One event for all queued weight-gradient jobs, then launch them on the side stream."""
    global _wgrad_jobs
    if not _wgrad_jobs:
        return
    jobs, _wgrad_jobs = _wgrad_jobs, []
    L = _L()
    dev = jobs[0][6]
    sides = _side_list(dev)
    key = (dev.type, dev.index)
    pos = _side_next.get(key, 0)
    with torch.cuda.device(dev):
        for i in range(min(len(jobs), len(sides))):     # every stream that gets a job waits for the batch
            _lib.check(L.gs_stream_fork(_wgrad_stream, sides[(pos + i) % len(sides)].cuda_stream),
                       "gs_stream_fork")
        for d, x, dy, weight, gw, need, _ in jobs:
            slot = pos % len(sides)
            side = sides[slot]
            pos += 1
            ws_s = _side_workspace(need, dev, side, slot)
            # dy and x must outlive the side-stream kernel that reads them: they are kept referenced
            # until the main stream has joined the side stream (join_side_streams) instead of being
            # handed to the caching allocator with record_stream (two calls and one pending event
            # per convolution and step)
            _side_keep.append((dy, x))
            # (descriptors are shared per layer: the operand-loader coefficients are per call)
            d.in_affine = x.affine.data_ptr() if x.affine is not None else None
            _lib.check(L.gs_conv2d_wgrad(ctypes.byref(d), x.ptr, dy.data_ptr(), gw.data_ptr(),
                                         ws_s.data_ptr(), ws_s.numel(), side.cuda_stream),
                       "gs_conv2d_wgrad")
    _side_next[key] = pos % len(sides)
    _side_dirty[key] = True
    for job in jobs:
        _notify(job[3])


DEFER_JOIN = False   # the runner sets it to overlap the optimizer step with the last weight gradients
SIDE_CHECKPOINT = None   # events on the side streams: every weight gradient queued before them is done


def side_checkpoint(tape):
    """Mark a point of the backward replay (recorded in forward order, so it fires when backward
    crosses it): all weight gradients of the layers AFTER this point have been handed to the side
    stream; an event on the side stream lets the optimizer update those parameters while the side
    stream still works on the layers before the point (the runner waits on SIDE_CHECKPOINT)."""
    def backward():
        global SIDE_CHECKPOINT
        if not DEFER_JOIN:
            return
        flush_wgrads()
        events = [s.record_event() for key, lst in _side_streams.items() if _side_dirty.get(key)
                  for s in lst]
        SIDE_CHECKPOINT = events or None
    tape.record(backward)


# ---- the optimizer stream ------------------------------------------------------------------------
# At the end of backward the weight-gradient stream is the critical path (it lags the training stream
# by the stem / stage-1 weight gradients, r04 trace: ~0.4 ms), and the optimizer step of everything
# else — ~95 % of the parameters, HBM-bound — used to run right there, beside those kernels.  The
# runner now updates the parameters of stages 3-4 and the heads on a THIRD stream as soon as backward
# has crossed into stage 2 (their gradients are final then), i.e. in the middle of backward.
BACKWARD_MARK_CB = None     # set by the runner for the duration of backward: called with the mark's tag
_opt_streams = {}


def opt_stream(dev):
    key = (dev.type, dev.index)
    s = _opt_streams.get(key)
    if s is None:
        s = _opt_streams[key] = torch.cuda.Stream(device=dev)
    return s


def backward_mark(tape, tag):
    """Record a point of the tape; when the backward replay crosses it, the runner's callback (if any)
    is told.  Everything recorded AFTER the mark has run its backward by then."""
    def backward():
        cb = BACKWARD_MARK_CB
        if cb is not None:
            cb(tag)
    tape.record(backward)


def fork_to(stream, dev):
    """``stream`` waits for everything queued so far on the current stream, on the weight-gradient
    stream (queued jobs are handed over first) and on the branch streams of ``dev``."""
    flush_wgrads()
    key = (dev.type, dev.index)
    L = _L()
    with torch.cuda.device(dev):
        cur = current_stream_ptr()
        srcs = {cur}
        if _side_dirty.get(key):
            srcs.update(sd.cuda_stream for sd in _side_streams.get(key, ()))
        for bkey, br in _branch_streams.items():
            if bkey[:2] == key and _branch_dirty.get(bkey):
                srcs.add(br.cuda_stream)
        for src in srcs:
            if src != stream.cuda_stream:
                _lib.check(L.gs_stream_fork(src, stream.cuda_stream), "gs_stream_fork")


def join_from(stream, dev):
    """The current stream waits for everything queued on ``stream``."""
    with torch.cuda.device(dev):
        _lib.check(_L().gs_stream_fork(stream.cuda_stream, current_stream_ptr()), "gs_stream_fork")


def join_side_streams(dev=None):
    """Make the current stream wait for the weight-gradient kernels queued on the side stream."""
    flush_wgrads()
    for key, lst in _side_streams.items():
        if _side_dirty.get(key) and (dev is None or (dev.type, dev.index) == key):
            with torch.cuda.device(lst[0].device):
                for s in lst:
                    _lib.check(_L().gs_stream_fork(s.cuda_stream, current_stream_ptr()), "gs_stream_fork")
            _side_dirty[key] = False
    if dev is None or not any(_side_dirty.values()):
        # everything the side stream read is now ordered before whatever the current stream does
        # next: the operands may go back to the allocator
        _side_keep.clear()


def side_stream_after_main(dev):
    """The side stream, made to wait for everything queued so far on the current stream and on the
    other compute stream of this device (a gradient bucket holds BatchNorm gradients written on the
    training stream and, where a branch ran, on the branch stream)."""
    s = _side_stream(dev)
    key = (dev.type, dev.index)
    with torch.cuda.device(dev):
        cur = current_stream_ptr()
        _lib.check(_L().gs_stream_fork(cur, s.cuda_stream), "gs_stream_fork")
        others = {sd.cuda_stream for sd in _side_list(dev)[1:]} if _side_dirty.get(key) else set()
        for bkey, br in _branch_streams.items():
            if bkey[:2] == key and _branch_dirty.get(bkey):
                others.add(br.cuda_stream)
                others.add(_branch_origin.get(bkey))
        for other in others:
            if other is not None and other != cur:
                _lib.check(_L().gs_stream_fork(other, s.cuda_stream), "gs_stream_fork")
    _side_dirty[key] = True   # the main stream joins it at the end of backward
    return s


def _side_workspace(need, dev, side, slot=0):
    """Split-K scratch of a side stream (allocated under that stream, so the caching allocator
    orders its reuse against that stream's kernels; one buffer per stream)."""
    buf = _ws_side._buf.get((dev.type, dev.index, slot))
    if buf is not None and buf.numel() >= need:
        return buf
    keep = _ws_side.slot
    _ws_side.slot = slot
    try:
        with torch.cuda.stream(side):
            return _ws_side.get(need, dev)
    finally:
        _ws_side.slot = keep


def reserve_workspaces(dev, nbytes=192 << 20):
    """Size the main and the side-stream scratch buffers for the largest request a convolution of
    the supernet can make (96 MiB of split-K slabs + reduction partials), so that a step graph
    capture never sees a workspace being (re)allocated."""
    _ws.get(nbytes, dev)
    for slot, side in enumerate(_side_list(dev)):
        _side_workspace(nbytes, dev, side, slot)



# ---- the branch stream -------------------------------------------------------------------------
# A bs-2 step is ~450 launches of 5-100 us, each a single round of 1-4 workgroups per CU with its
# ramp and its tail; the only thing that fills those is ANOTHER launch running beside
# it.  Two sub-chains of the network are independent of the main chain and run on a second in-order
# queue, fenced by events:
#   * the projection shortcut (conv + BN) of a stage's first block, beside conv1 -> conv2, forward
#     and backward (gaiaseg/models/utils/dynamic_res_layer.py:70-125);
#   * the auxiliary head with its loss, beside stage 4 / the decode head
#     ("dynamic_encoder_decoder-distill-backup (1).py":85-143: two independent consumers of x).
# Same kernels, same operands, same results; only the queue differs.  GS_BRANCH=0 keeps one queue;
# GS_BRANCH_SHORTCUT / GS_BRANCH_AUX switch the two uses separately.
BRANCH = os.environ.get("GS_BRANCH", "1") != "0"
BRANCH_SHORTCUT = BRANCH and os.environ.get("GS_BRANCH_SHORTCUT", "1") != "0"
BRANCH_AUX = BRANCH and os.environ.get("GS_BRANCH_AUX", "1") != "0"
BRANCH_SHORTCUT_MAX_GFLOP = float(os.environ.get("GS_BRANCH_SHORTCUT_MAX_GFLOP", "8"))
# GS_AUX_PREFORK=1: the auxiliary heads' stream forks behind the stage they read (they then run beside
# the later stages too) instead of behind the whole backbone (beside the decode head only).  Measured
# (profiles/r04_stream_experiments.md): no faster, and the stage-4 K3 launches share the chip with the
# auxiliary head's 3x3 (K3 0.574 vs 0.582 of peak on R50) -- default off.
AUX_PREFORK = os.environ.get("GS_AUX_PREFORK", "0") == "1"
SLOT_SHORTCUT, SLOT_AUX = 1, 2   # scratch slot / branch stream index (0 = the training stream)
_branch_streams = {}     # (device type, index, slot) -> torch.cuda.Stream
_branch_slot_of = {}     # raw stream handle -> slot (adopt_current_stream)
_branch_origin = {}      # branch key -> raw handle of the stream it was last forked from
_branch_dirty = {}
_branch_cb_armed = False
BRANCH_PRIORITY = os.environ.get("GS_BRANCH_PRIORITY", "0") == "1"


def _branch_stream(dev, slot):
    key = (dev.type, dev.index, slot)
    s = _branch_streams.get(key)
    if s is None:
        # (GS_BRANCH_PRIORITY=1: the branch streams at the training stream's high priority — A/B knob)
        s = torch.cuda.Stream(device=dev, priority=-1 if BRANCH_PRIORITY else 0)
        _branch_streams[key] = s
        _branch_slot_of[s.cuda_stream] = slot
    return s


def on_branch():
    return _ws.slot != 0


def prefork_branch(dev, slot):
    """Make branch stream ``slot`` wait for what is queued on the current stream NOW; a later
    ``branch_scope(..., forked=True)`` then starts from this point of the current stream instead of
    from its tail at that time (the auxiliary head forks behind stage 3 while the host goes on to
    queue stage 4)."""
    if not BRANCH or dev.type != "cuda" or on_branch():
        return
    br = _branch_stream(dev, slot)
    key = (dev.type, dev.index, slot)
    cur = current_stream_ptr()
    with torch.cuda.device(dev):
        _lib.check(_L().gs_stream_fork(cur, br.cuda_stream), "gs_stream_fork")
    _branch_origin[key] = cur
    _branch_dirty[key] = True


def _enter_branch(dev, slot, fork=True):
    """Fork: branch stream ``slot`` waits for everything queued on the current stream, then becomes
    the current stream (with its own scratch buffers).  Returns the stream to go back to."""
    flush_wgrads()            # (backward) queued weight gradients belong to the stream we leave
    br = _branch_stream(dev, slot)
    prev = torch.cuda.current_stream(dev)
    key = (dev.type, dev.index, slot)
    if fork or _branch_origin.get(key) != prev.cuda_stream:
        with torch.cuda.device(dev):
            _lib.check(_L().gs_stream_fork(prev.cuda_stream, br.cuda_stream), "gs_stream_fork")
        _branch_origin[key] = prev.cuda_stream
    _branch_dirty[key] = True
    torch.cuda.set_stream(br)
    _ws.slot = slot
    return prev


def _leave_branch(prev):
    flush_wgrads()
    torch.cuda.set_stream(prev)
    _ws.slot = 0


def join_branch(dev, slot):
    """The current stream waits for everything queued on branch stream ``slot`` so far."""
    key = (dev.type, dev.index, slot)
    br = _branch_streams.get(key)
    if br is None or not _branch_dirty.get(key):
        return
    with torch.cuda.device(dev):
        _lib.check(_L().gs_stream_fork(br.cuda_stream, current_stream_ptr()), "gs_stream_fork")


def join_branch_streams():
    """End of a step's backward: the stream the branches were forked from waits for them."""
    global _branch_cb_armed
    _branch_cb_armed = False
    for key, br in _branch_streams.items():
        if _branch_dirty.get(key) and _branch_origin.get(key) is not None:
            with torch.cuda.device(br.device):
                _lib.check(_L().gs_stream_fork(br.cuda_stream, _branch_origin[key]), "gs_stream_fork")
            _branch_dirty[key] = False


def adopt_current_stream():
    """Entry of an autograd node's backward: autograd has made the node's forward stream current;
    pick the matching scratch slot.  On a branch stream also arrange for the final join (nothing
    else would order a later reader on the training stream behind, e.g., the auxiliary head's
    BatchNorm gradients when no upstream node consumes a gradient of this one)."""
    global _branch_cb_armed
    prev = _ws.slot
    if _branch_slot_of:
        slot = _branch_slot_of.get(current_stream_ptr(), 0)
        if slot != prev:
            flush_wgrads()
        _ws.slot = slot
        if slot and not _branch_cb_armed:
            _branch_cb_armed = True
            torch.autograd.Variable._execution_engine.queue_callback(join_branch_streams)
    return prev


def restore_stream_slot(prev):
    if _ws.slot != prev:
        flush_wgrads()
        _ws.slot = prev


class branch_scope:
    """``with ops.branch_scope(dev):`` evaluate a whole sub-model (the auxiliary head and its loss)
    on a branch stream.  Its autograd nodes run their backward there too (autograd's stream
    semantics); ``ops.join_branch(dev, slot)`` afterwards orders the current stream behind it.
    ``forked``: the branch already waits for the right point of the current stream
    (``prefork_branch``)."""

    def __init__(self, dev, enabled=True, slot=SLOT_AUX, forked=False):
        self.dev, self.slot, self.forked = dev, slot, forked
        self.on = bool(enabled and BRANCH and dev.type == "cuda" and not on_branch())
        self.prev = None

    def __enter__(self):
        if self.on:
            self.prev = _enter_branch(self.dev, self.slot, fork=not self.forked)
        return self

    def __exit__(self, *exc):
        if self.on:
            _leave_branch(self.prev)
        return False


class Branch:
    """A sub-chain of ONE tape evaluated on a branch stream.

        br = Branch(tape, dev)
        with br:                          # forward: fork; ops run (and record) on the branch
            identity = shortcut(x)
        a = conv1(x)
        br.record_backward_join(tape)     # backward: everything recorded BEFORE this point runs
                                          #           after the branch's backward has finished
        b = conv2(a)
        br.record_backward_body(tape)     # backward: fork here, replay the branch's closures
        br.join()                         # forward: the current stream waits for the branch
        out = conv3(b) + identity

    Backward order on the host: conv3, [fork, branch body], conv2, [join], conv1 — the branch's data
    gradient is the FIRST writer of x.g, conv1's accumulates onto it after the join.  With the
    branch switched off the same closures run in the same order on one stream."""

    __slots__ = ("tape", "dev", "on", "ops", "slot", "_saved", "_prev")

    def __init__(self, tape, dev, enabled=True, slot=SLOT_SHORTCUT):
        self.tape, self.dev, self.slot = tape, dev, slot
        self.on = bool(enabled and BRANCH and dev.type == "cuda" and not on_branch())
        self.ops = []
        self._saved = self._prev = None

    def __enter__(self):
        self._saved = self.tape.ops
        self.tape.ops = self.ops
        if self.on:
            self._prev = _enter_branch(self.dev, self.slot)
        return self

    def __exit__(self, *exc):
        if self.on:
            _leave_branch(self._prev)
        self.tape.ops = self._saved
        self._saved = self._prev = None
        return False

    def join(self):
        if self.on:
            join_branch(self.dev, self.slot)

    def record_backward_join(self, tape):
        if not self.on:
            return
        dev, slot = self.dev, self.slot
        tape.record(lambda: join_branch(dev, slot))

    def record_backward_body(self, tape):
        body, dev, on, slot = self.ops, self.dev, self.on, self.slot

        def backward():
            prev = _enter_branch(dev, slot) if on else None
            try:
                for fn in reversed(body):
                    fn()
            finally:
                if on:
                    _leave_branch(prev)
        if body:
            tape.record(backward)


def _L():
    return _lib.load()


_INT32_ARRAYS = {}


def _int32_array(n):
    """ctypes array type of n int32 (cached: creating the type per call leaves a cycle of type
    objects for the garbage collector at every step)."""
    t = _INT32_ARRAYS.get(n)
    if t is None:
        t = _INT32_ARRAYS[n] = ctypes.c_int32 * n
    return t


def ensure_grad(param):
    """``param.grad`` with the physical layout of ``param`` (zeros on first use).

    Parameters with a padded / permuted physical layout (HWIO conv weights, the padded conv_seg
    bias) carry a ``_gs_grad_factory`` that allocates matching storage; with a ParamArena installed
    the gradients already exist as views of the flat gradient buffer."""
    g = param.grad
    if g is not None and g.stride() == param.stride():   # the common case, checked first
        return g
    if param.grad is None:
        factory = getattr(param, "_gs_grad_factory", None)
        if factory is not None:
            param.grad = factory()
        else:
            param.grad = torch.zeros_like(param, memory_format=torch.preserve_format)
    if param.grad.stride() != param.stride():
        raise RuntimeError("gradient layout %s differs from parameter layout %s"
                           % (param.grad.stride(), param.stride()))
    return param.grad


def _notify(param):
    hook = param.__dict__.get("_gs_grad_ready")
    if hook is not None:
        hook(param)


def conv_out_size(h, k, s, p, d):
    return (h + 2 * p - d * (k - 1) - 1) // s + 1


def _conv_desc(x, weight, co, stride, pad, dil, ldy, ld_add=0, role=0):
    """gs_conv_desc for activation ``x`` and a max-size weight Parameter (logical OIHW, physical
    HWIO with row pitch ``Co_ld``)."""
    co_max, ci_max, kh, kw = weight.shape
    co_ld = weight.stride(1)
    d = _lib.ConvDesc()
    if x.nchw_image:
        n, c, h, w = x.t.shape
        d.x_sn, d.x_sc, d.x_sh, d.x_sw = x.t.stride()
    else:
        n, h, w, c = x.t.shape
        d.x_sn, d.x_sh, d.x_sw, d.x_sc = x.t.stride()
    d.N, d.H, d.W, d.Ci, d.Co = n, h, w, c, co
    d.Ci_max, d.Co_ld, d.KH, d.KW = ci_max, co_ld, kh, kw
    d.stride, d.pad, d.dil = stride, pad, dil
    d.Ho = conv_out_size(h, kh, stride, pad, dil)
    d.Wo = conv_out_size(w, kw, stride, pad, dil)
    d.ldy, d.ld_add = ldy, ld_add
    d.role = role
    if c > ci_max:
        raise ValueError("input has %d channels, conv supports at most %d" % (c, ci_max))
    return d


class KernelTimer:
    """Optional HIP-event timing of tagged kernel launches on the current stream (bench.py uses it
    to time the dynamic 3x3 bottleneck conv inside real training steps)."""

    def __init__(self):
        self.records = {}   # tag -> list of (start_event, end_event, flops)

    def begin(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def end(self, tag, start, flops):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.records.setdefault(tag, []).append((start, e, flops))

    def summary(self):
        """tag -> (launches, total_ms, total_flops); call after a device synchronize."""
        out = {}
        for tag, recs in self.records.items():
            ms = sum(s.elapsed_time(e) for s, e, _ in recs)
            out[tag] = (len(recs), ms, float(sum(f for _, _, f in recs)))
        return out


TIMER = None  # set to a KernelTimer to enable

# Diagnostics: when set to a list, every BatchNorm(+residual)+ReLU records (gamma Parameter,
# bool NHWC tensor "output > 0", taken at once: UPer's top-down add later updates outputs in place).  The gradient-parity tests use it to evaluate the CPU
# oracle on the same ReLU branch pattern as this path (tests/conftest.py).  None = off (no cost).
RELU_TRACE = None
POOL_TRACE = None   # same, for MaxPool2d: list of uint8 [N, Ho, Wo, C] argmax tap tensors (kh * k + kw)


def _trace_relu(bn, out):
    if RELU_TRACE is not None and bn.weight is not None:
        RELU_TRACE.append((bn.weight, out.t > 0))


def _trace_relu_deferred(bn, y, coeffs):
    """Same for a deferred BN+ReLU: the mask comes from the real apply kernel (same expression as the
    fused loaders and the backward mask), run into a scratch buffer only while tracing."""
    if RELU_TRACE is not None and bn.weight is not None:
        tmp = Act.empty(y.N, y.H, y.W, y.C, y.t.device)
        _lib.check(_L().gs_bn_apply(y.ptr, y.rows, y.C, y.ld, coeffs.data_ptr(), None, 0, 1, tmp.ptr,
                                    tmp.ld, current_stream_ptr()), "gs_bn_apply")
        RELU_TRACE.append((bn.weight, tmp.t > 0))


# Loader fusion of BN + ReLU into the consumer convolution (gs_conv_desc.in_affine).  Which edges of
# a bottleneck use it: "conv3" = bn2 -> conv3 (1x1), "conv2" = bn1 -> conv2 (the 3x3, K3).
# GS_DEFER_BN = comma list, "all" or "none".  r02 A/B on the sampled mix (bench.py, 30 steps):
# none 145.5 img/s, K3 at 0.555 of peak; all 145.4 img/s, K3 at 0.500 (the affine + select in the
# store slot of the K loop costs the 3x3 what the two removed bn_apply launches save); the default
# keeps the headline 3x3 kernel free of it.
_defer = os.environ.get("GS_DEFER_BN", "conv3").lower()
DEFER_EDGES = {"all": {"conv2", "conv3"}, "none": set()}.get(_defer, set(_defer.split(",")))
DEFER_BN = True   # master switch (tests flip it to compare the two forms bit for bit)
# the projection shortcut's BatchNorm inside the block's last apply pass (gs_bn_args.residual_coeffs)
DEFER_SHORTCUT_BN = os.environ.get("GS_DEFER_SHORTCUT_BN", "1") != "0"


def materialize(tape, x):
    """Write out relu(bn(t)) of a deferred activation (consumers without the loader fusion)."""
    if x.affine is None:
        return x
    z = Act.empty(x.N, x.H, x.W, x.C, x.t.device)
    _lib.check(_L().gs_bn_apply(x.ptr, x.rows, x.C, x.ld, x.affine.data_ptr(), None, 0, 1, z.ptr, z.ld,
                                current_stream_ptr()), "gs_bn_apply")
    z.requires_grad = x.requires_grad
    add_grad_passthrough(tape, x, z)
    return z


def materialize_residual(tape, r):
    """Write out bn(t) of a residual whose BatchNorm was left to its consumer (Act.res_affine), for
    consumers that take a plain addend."""
    if r is None or r.res_affine is None:
        return r
    z = Act.empty(r.N, r.H, r.W, r.C, r.t.device)
    _lib.check(_L().gs_bn_apply(r.ptr, r.rows, r.C, r.ld, r.res_affine.data_ptr(), None, 0, 0, z.ptr,
                                z.ld, current_stream_ptr()), "gs_bn_apply")
    z.requires_grad = r.requires_grad
    add_grad_passthrough(tape, r, z)
    return z


def conv2d(tape, x, weight, bias, co, stride=1, pad=0, dil=1, out=None, tag=None):
    """DynConv2d forward: y = conv(x, weight[:co, :x.C]) (+ bias[:co]).

    ``co`` is the active output width; the active input width is x.C."""
    L = _L()
    x = materialize(tape, x)
    co_eff = round_up(co, 4)
    dev = x.t.device
    kh, kw = weight.shape[2], weight.shape[3]
    if x.nchw_image:
        n, _, h, w = x.t.shape
    else:
        n, h, w, _ = x.t.shape
    ho, wo = conv_out_size(h, kh, stride, pad, dil), conv_out_size(w, kw, stride, pad, dil)
    if out is None:
        out = Act.empty(n, ho, wo, co, dev)
    d = _conv_desc(x, weight, co_eff, stride, pad, dil, out.ld, role=1 if tag == "k3" else 0)
    need = L.gs_conv2d_workspace_bytes(ctypes.byref(d))
    ws = _ws.get(need, dev)
    st = current_stream_ptr()
    timed = TIMER is not None and tag is not None
    if timed:
        t0 = TIMER.begin()
    _lib.check(L.gs_conv2d_forward(ctypes.byref(d), x.ptr, weight.data_ptr(),
                                   bias.data_ptr() if bias is not None else None, None, out.ptr,
                                   ws.data_ptr(), ws.numel(), st), "gs_conv2d_forward")
    if timed:
        ci = x.t.shape[1] if x.nchw_image else x.C
        TIMER.end(tag + ".fwd", t0, 2.0 * n * ho * wo * co * ci * kh * kw)

    def backward():
        dy = out.g
        if dy is None:
            return
        ws_b = _ws.get(need, dev)
        s = current_stream_ptr()
        if weight.requires_grad:
            gw = ensure_grad(weight)
            if SIDE_WGRAD:
                queue_wgrad(d, x, dy, weight, gw, need, dev)   # notifies when launched
            else:
                _lib.check(L.gs_conv2d_wgrad(ctypes.byref(d), x.ptr, dy.data_ptr(), gw.data_ptr(),
                                             ws_b.data_ptr(), ws_b.numel(), s), "gs_conv2d_wgrad")
                _notify(weight)
        if bias is not None and bias.requires_grad:
            gb = ensure_grad(bias)
            rows = out.rows
            nb = L.gs_colsum_workspace_bytes(rows, co_eff)
            wsb = _ws.get(nb, dev)
            # the padded bias storage holds co_eff floats; the sum of the pad column is zero
            _lib.check(L.gs_colsum(dy.data_ptr(), rows, co_eff, out.ld, gb.data_ptr(),
                                   wsb.data_ptr(), wsb.numel(), s), "gs_colsum")
            _notify(bias)
        if x.requires_grad:
            acc = x.g is not None
            if not acc:
                x.new_grad() if x.parent is None else _alloc_parent_grad(x)
            _lib.check(L.gs_conv2d_dgrad(ctypes.byref(d), dy.data_ptr(), weight.data_ptr(),
                                         x.g.data_ptr(), 1 if acc else 0, ws_b.data_ptr(),
                                         ws_b.numel(), s), "gs_conv2d_dgrad")

    tape.record(backward)
    return out


def _alloc_parent_grad(a):
    """Gradient storage for a channel slice: allocate (zeroed) on the owning buffer."""
    root = a
    while root.parent is not None:
        root = root.parent
    if root.g is None:
        root.new_grad().zero_()


class BNParams:
    """What the BN kernels need from a DynBN / SyncBN module (leading-slice semantics, A2)."""
    __slots__ = ("weight", "bias", "running_mean", "running_var", "eps", "momentum", "training",
                 "process_group", "num_batches_tracked")

    def __init__(self, weight, bias, running_mean, running_var, eps, momentum, training,
                 process_group=None, num_batches_tracked=None):
        self.weight, self.bias = weight, bias
        self.running_mean, self.running_var = running_mean, running_var
        self.eps, self.momentum, self.training = eps, momentum, training
        self.process_group = process_group
        self.num_batches_tracked = num_batches_tracked


def _sync_stats(sums, count, C, pg, equal_counts=False):
    """SyncBN: merge per-rank (shifted sums, count) into global statistics (Chan et al.).

    Mirrors torch.nn.SyncBatchNorm's all_gather of [mean, var, count]; the
    payload is 2C+1 floats per rank."""
    import torch.distributed as dist
    world = dist.get_world_size(pg)
    if sums.is_cuda and equal_counts:
        # device path of the training step: two launches around one all_gather
        L = _L()
        st = current_stream_ptr()
        local = torch.empty(2 * C + 1, dtype=torch.float64, device=sums.device)
        _lib.check(L.gs_bn_sync_local(sums.data_ptr(), float(count), C, local.data_ptr(), st),
                   "gs_bn_sync_local")
        gathered = torch.empty((world, 2 * C + 1), dtype=torch.float64, device=sums.device)
        if dist.get_backend(pg) == "nccl":
            dist.all_gather_into_tensor(gathered, local, group=pg)
        else:   # gloo (tests, rehearsals): list form, rows of `gathered` as outputs
            dist.all_gather([gathered[r] for r in range(world)], local, group=pg)
        merged = torch.empty(3 * C, dtype=torch.float32, device=sums.device)
        _lib.check(L.gs_bn_sync_merge(gathered.data_ptr(), world, C, merged.data_ptr(), st),
                   "gs_bn_sync_merge")
        return merged, float(count) * world
    d1 = sums[:C].double() / count
    mean = sums[2 * C:3 * C].double() + d1
    var = (sums[C:2 * C].double() / count - d1 * d1).clamp_(min=0)
    # (torch.full, not torch.tensor([...], device=...): a host-to-device copy from pageable memory
    # synchronises the stream and would stall the host once per SyncBN layer and step)
    local = torch.cat([mean, var, torch.full((1,), float(count), dtype=torch.float64,
                                             device=sums.device)])
    gathered = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(gathered, local, group=pg)
    g = torch.stack(gathered)                     # [world, 2C+1]
    cnt = g[:, 2 * C:]                            # [world, 1]
    total = cnt.sum()
    gmean = (g[:, :C] * cnt).sum(0) / total
    gvar = ((g[:, C:2 * C] + (g[:, :C] - gmean) ** 2) * cnt).sum(0) / total
    merged = torch.empty_like(sums)
    merged[:C] = 0
    merged[C:2 * C] = (gvar * total).float()
    merged[2 * C:3 * C] = gmean.float()
    if not equal_counts:
        return merged, float(total.item())   # exact for any per-rank counts; synchronises the host
    # equal_counts: every rank holds the same batch and crop size (the training path: fixed
    # samples_per_gpu and crop), so the total is count * world and is NOT read back from the device
    # -- a .item() here drains the GPU queue once per SyncBN layer and step, which serialises host
    # and GPU in multi-GPU runs.  (The weighting of the per-rank statistics above uses the
    # gathered counts either way.)
    return merged, float(count) * world


def batchnorm(tape, x, bn, relu=False, residual=None, out=None, inplace=False):
    """y = act(BN(x) (+ residual)).  Training: batch statistics over N*H*W of the active slice,
    running statistics updated on the slice; eval: running statistics.  ``inplace`` reuses x's
    storage for y (only when x is not needed by backward, i.e. never in training)."""
    L = _L()
    x = materialize(tape, x)
    residual = materialize_residual(tape, residual)
    dev = x.t.device
    C, rows = x.C, x.rows
    st = current_stream_ptr()
    buf = torch.empty(7 * C, dtype=torch.float32, device=dev)   # {sums[3C], coeffs[4C]}
    sums_ptr = buf.data_ptr()
    coeffs_ptr = sums_ptr + 12 * C
    gamma = bn.weight.data_ptr() if bn.weight is not None else None
    beta = bn.bias.data_ptr() if bn.bias is not None else None
    count = float(rows)
    use_batch = bn.training or bn.running_mean is None
    if use_batch:
        if rows <= 1 and bn.process_group is None:
            raise ValueError("Expected more than 1 value per channel when training, got input "
                             "size %s" % (tuple(x.t.shape),))
        nb = L.gs_bn_stats_workspace_bytes(rows, C)
        ws = _ws.get(nb, dev)
        rm = bn.running_mean.data_ptr() if (bn.running_mean is not None and bn.training) else None
        rv = bn.running_var.data_ptr() if (bn.running_var is not None and bn.training) else None
        mom = bn.momentum if bn.momentum is not None else 0.1
        if bn.process_group is None:
            # rank-local statistics: partial sums + (sum, finalize) in two launches
            _lib.check(L.gs_bn_stats_finalize(x.ptr, rows, C, x.ld, gamma, beta, bn.eps, mom, rm, rv,
                                              coeffs_ptr, ws.data_ptr(), ws.numel(), st),
                       "gs_bn_stats_finalize")
        else:
            _lib.check(L.gs_bn_stats(x.ptr, rows, C, x.ld, sums_ptr, ws.data_ptr(),
                                     ws.numel(), st), "gs_bn_stats")
            merged, count = _sync_stats(buf[:3 * C], count, C, bn.process_group, equal_counts=True)
            _lib.check(L.gs_bn_finalize(merged.data_ptr(), count, C, gamma, beta, bn.eps, mom, rm,
                                        rv, coeffs_ptr, st), "gs_bn_finalize")
        if bn.training and bn.num_batches_tracked is not None:
            bn.num_batches_tracked()  # host-side counter: no device op per BN per step
    else:
        _lib.check(L.gs_bn_eval_coeffs(bn.running_mean.data_ptr(), bn.running_var.data_ptr(), C,
                                       gamma, beta, bn.eps, coeffs_ptr, st),
                   "gs_bn_eval_coeffs")
    if out is None:
        out = x if (inplace and not tape.enabled) else Act.empty(x.N, x.H, x.W, C, dev)
    _lib.check(L.gs_bn_apply(x.ptr, rows, C, x.ld, coeffs_ptr,
                             residual.ptr if residual is not None else None,
                             residual.ld if residual is not None else 0, 1 if relu else 0,
                             out.ptr, out.ld, st), "gs_bn_apply")
    if relu:
        _trace_relu(bn, out)

    def backward():
        dy = out.g
        if dy is None:
            return
        s = current_stream_ptr()
        keep = buf  # noqa: F841 -- the closure owns the coefficient storage behind coeffs_ptr
        mask = 0 if not relu else (2 if residual is not None else 1)
        nbw = L.gs_bn_bwd_workspace_bytes(rows, C)
        wsb = _ws.get(nbw, dev)
        bsums = torch.empty(2 * C, dtype=torch.float32, device=dev)
        # the masked gradient g is also the gradient of the identity branch: write it in place
        want_g = residual is not None and residual.requires_grad and relu
        _lib.check(L.gs_bn_bwd_reduce(dy.data_ptr(), dy.stride(2),
                                      x.ptr, x.ld, out.ptr, out.ld, rows, C, coeffs_ptr,
                                      mask, dy.data_ptr() if want_g else None, dy.stride(2),
                                      bsums.data_ptr(), wsb.data_ptr(), wsb.numel(), s),
                   "gs_bn_bwd_reduce")
        bcount = count
        wgrad = bn.weight is not None and bn.weight.requires_grad
        bgrad = bn.bias is not None and bn.bias.requires_grad
        gw = ensure_grad(bn.weight) if wgrad else None
        gb = ensure_grad(bn.bias) if bgrad else None
        synced = use_batch and bn.process_group is not None
        if synced:
            # dx needs the GROUP sums (torch.nn.SyncBatchNorm backward); dgamma / dbeta are this
            # rank's own sums -- the data-parallel gradient all-reduce adds the ranks up afterwards.
            # Writing the group sums there would count every rank world_size times.
            import torch.distributed as dist
            if wgrad:
                gw[:C].copy_(bsums[C:2 * C])
            if bgrad:
                gb[:C].copy_(bsums[:C])
            dist.all_reduce(bsums, group=bn.process_group)
        mask_apply = 0 if want_g else mask  # dy already masked in place
        if x.g is not None:
            raise RuntimeError("BN input has more than one consumer; unsupported")
        x.new_grad() if x.parent is None else _alloc_parent_grad(x)
        _lib.check(L.gs_bn_bwd_apply(dy.data_ptr(), dy.stride(2), x.ptr, x.ld, out.ptr, out.ld,
                                     rows, C, coeffs_ptr, bsums.data_ptr(), bcount,
                                     mask_apply, 1 if use_batch else 0, x.g.data_ptr(),
                                     x.g.stride(2), gw.data_ptr() if (wgrad and not synced) else None,
                                     gb.data_ptr() if (bgrad and not synced) else None, s),
                   "gs_bn_bwd_apply")
        if wgrad:
            _notify(bn.weight)
        if bgrad:
            _notify(bn.bias)
        if residual is not None and residual.requires_grad:
            src = dy  # masked (relu) or plain (no relu) upstream gradient
            if residual.g is None and residual.parent is None and src.stride() == residual.t.stride():
                residual.g = src
            else:
                if residual.g is None:
                    residual.new_grad() if residual.parent is None else _alloc_parent_grad(residual)
                    acc = 0 if residual.parent is None else 1
                else:
                    acc = 1
                _lib.check(L.gs_copy2d(src.data_ptr(), src.stride(2), residual.g.data_ptr(),
                                       residual.g.stride(2), rows, C, 1.0, acc, s), "gs_copy2d")

    tape.record(backward)
    return out


BNBWD_FUSE = os.environ.get("GS_NO_BNBWD_FUSE") is None
RELU_MASK_BYTES = os.environ.get("GS_RELU_MASK_BYTES", "1") != "0"   # gs_bn_bwd_fuse mode 3
BNBWD_FUSED_COUNT = 0   # diagnostics: how many BN-backward reductions ran inside a dgrad epilogue
BNBWD_FUSED_MASK_COUNT = 0   # ... of which took their ReLU mask from the mask bytes (mode 3)


class _ConvBnPlan:
    """Everything about one conv + BN layer call that depends only on the layer and on the geometry
    of its input: the descriptor, the BN argument block, sizes.  Cached on the weight Parameter per
    (input geometry, active widths, flags): a supernet layer sees a handful of them, and building
    the two ctypes structures was a third of the host time of a layer call.  Pointers (the input's
    deferred-BN coefficients, the BN parameters — an arena rebuild moves them) are NOT part of the
    plan: they are written into the cached structures at every use, forward and backward."""
    __slots__ = ("d", "args", "need", "n", "ho", "wo", "rows", "C", "use_batch", "affine_ok")


def _conv_bn_plan(L, x, weight, co_eff, bn, stride, pad, dil, relu, tag, use_batch):
    kh, kw = weight.shape[2], weight.shape[3]
    if x.nchw_image:
        n, _, h, w = x.t.shape
    else:
        n, h, w = x.N, x.H, x.W
    pl = _ConvBnPlan()
    pl.n = n
    pl.ho, pl.wo = conv_out_size(h, kh, stride, pad, dil), conv_out_size(w, kw, stride, pad, dil)
    pl.rows = n * pl.ho * pl.wo
    pl.C = co_eff
    pl.use_batch = use_batch
    pl.d = d = _conv_desc(x, weight, co_eff, stride, pad, dil, round_up(co_eff, 4),
                          role=1 if tag == "k3" else 0)
    pl.affine_ok = bool(L.gs_conv2d_in_affine_supported(ctypes.byref(d)))
    pl.need = L.gs_conv_bn_workspace_bytes(ctypes.byref(d))
    pl.args = args = _lib.BnArgs()
    args.eps = bn.eps
    args.momentum = bn.momentum if bn.momentum is not None else 0.1
    args.use_batch_stats = 1 if use_batch else 0
    args.update_running = 1 if (bn.training and bn.running_mean is not None) else 0
    args.relu = 1 if relu else 0
    return pl


def _bn_pointers(args, bn):
    w, b, rm, rv = bn.weight, bn.bias, bn.running_mean, bn.running_var
    args.gamma = w.data_ptr() if w is not None else None
    args.beta = b.data_ptr() if b is not None else None
    args.running_mean = rm.data_ptr() if rm is not None else None
    args.running_var = rv.data_ptr() if rv is not None else None


def conv_bn(tape, x, weight, co, bn, stride=1, pad=0, dil=1, relu=False, residual=None, out=None,
            tag=None, defer=False, owns_input_grad=False, defer_residual=False):
    """z = act(BN(conv(x, weight[:co, :x.C])) (+ residual)) through ONE library call per direction
    (gs_conv_bn_forward / gs_conv_bn_backward): same kernels, same results as conv2d() followed by
    batchnorm(), a third of the host work.  Rank-local BatchNorm only (``bn.process_group`` None);
    the conv has no bias.

    ``defer`` (with relu, no residual, no ``out``): the BN + ReLU is NOT applied here; the returned
    activation carries the coefficients (Act.affine) and its consumer — the next conv_bn — evaluates
    relu(bn(y)) in its operand loaders, forward and weight gradient (gs_conv_desc.in_affine).  A
    deferred INPUT ``x`` is consumed that way when the library supports the shape, else written out.

    ``defer_residual`` (no relu, no residual, no ``out``): likewise the BN is not applied; the returned
    activation (the raw conv output) carries the coefficients in ``Act.res_affine`` and must be used
    as the ``residual`` of a conv_bn call, whose apply pass adds bn(residual)
    (gs_bn_args.residual_coeffs).  A ``residual`` that carries ``res_affine`` is consumed that way.

    ``owns_input_grad``: this conv's data gradient is the LAST contribution to ``x.g`` (accumulation
    included).  If x came out of a training-mode BN + ReLU (``x.bnb``), the dgrad epilogue then also
    applies that ReLU's mask and reduces that BN's backward sums (gs_bn_bwd_fuse): the producer's
    backward finds them in ``x.bnb_sums`` and skips its reduction pass over dz and y."""
    L = _L()
    co_eff = round_up(co, 4)
    if co_eff != co:   # channel counts that are not multiples of 4 keep the two-step path
        raise ValueError("conv_bn needs an output width that is a multiple of 4, got %d" % co)
    dev = x.t.device
    defer = bool(defer and DEFER_BN and relu and residual is None and out is None)
    defer_residual = bool(defer_residual and DEFER_BN and not relu and residual is None and out is None)
    use_batch = bn.training or bn.running_mean is None
    plans = weight.__dict__.get("_gs_plans")
    if plans is None:
        plans = weight._gs_plans = {}
    pl = None
    for _ in range(2):   # (second round: the deferred input had to be written out)
        key = None if x.nchw_image else (x.N, x.H, x.W, x.C, x.ld, stride, pad, dil, co, relu, tag,
                                         bn.training, use_batch, bn.momentum, bn.eps)
        pl = plans.get(key) if key is not None else None
        if pl is None:
            pl = _conv_bn_plan(L, x, weight, co_eff, bn, stride, pad, dil, relu, tag, use_batch)
            if key is not None:
                if len(plans) > 64:
                    plans.clear()
                plans[key] = pl
        if x.affine is None or pl.affine_ok:
            break
        x = materialize(tape, x)
    if use_batch and pl.rows <= 1:
        raise ValueError("Expected more than 1 value per channel when training, got input size %s"
                         % ((pl.n, co, pl.ho, pl.wo),))
    d, args, need, C, rows = pl.d, pl.args, pl.need, pl.C, pl.rows
    y = Act.empty(pl.n, pl.ho, pl.wo, co, dev)
    in_affine = x.affine   # kept alive by the backward closure
    d.in_affine = in_affine.data_ptr() if in_affine is not None else None
    if defer or defer_residual:
        out = y
    elif out is None:
        out = Act.empty(pl.n, pl.ho, pl.wo, co, dev)
    _bn_pointers(args, bn)
    res_affine = residual.res_affine if residual is not None else None   # (kept alive by the closure)
    args.residual_coeffs = res_affine.data_ptr() if res_affine is not None else None
    # A block output relu(bn(y) + identity) whose consumer may fold this BatchNorm's backward
    # reduction into its data-gradient epilogue: the apply pass also writes the ReLU mask as one byte
    # per channel quad, so that epilogue reads rows * C / 4 bytes instead of the activation again
    relu_mask = None
    if (RELU_MASK_BYTES and relu and residual is not None and not defer and use_batch and BNBWD_FUSE
            and tape.enabled and out.parent is None):
        relu_mask = torch.empty((rows, C // 4), dtype=torch.uint8, device=dev)
    args.relu_mask = relu_mask.data_ptr() if relu_mask is not None else None
    # [scale | beta | mean | invstd][C] + a 2C slot for this BatchNorm's backward sums (filled by
    # this layer's backward, or by its consumer's dgrad epilogue): one small allocation per layer
    coeffs = torch.empty(6 * C, dtype=torch.float32, device=dev)
    ws = _ws.get(need, dev)
    _lib.check(L.gs_conv_bn_forward(ctypes.byref(d), x.ptr, weight.data_ptr(), ctypes.byref(args),
                                    residual.ptr if residual is not None else None,
                                    residual.ld if residual is not None else 0, y.ptr,
                                    coeffs.data_ptr(), None if (defer or defer_residual) else out.ptr,
                                    out.ld, ws.data_ptr(), ws.numel(), current_stream_ptr()),
               "gs_conv_bn_forward")
    if use_batch and bn.training and bn.num_batches_tracked is not None:
        bn.num_batches_tracked()
    if defer:
        out.affine = coeffs
        _trace_relu_deferred(bn, y, coeffs)
    elif defer_residual:
        out.res_affine = coeffs
    elif relu:
        _trace_relu(bn, out)
    if relu and use_batch and BNBWD_FUSE and tape.enabled and out.parent is None:
        # (y, coefficients, mask mode).  Mode 2 masks with the post-activation output, i.e. with the
        # activation that carries this tuple: it is NOT stored in the tuple — a self-reference would
        # make every residual output an uncollectable-by-refcount cycle, its device memory would
        # live until Python's cyclic GC happens to run, and the caching allocator would answer with
        # ~15 hipMalloc calls per step (measured: reserved memory 7.7 -> 25 GB over 60 steps)
        # Likewise a deferred output IS its BN input y: stored as None, the consumer substitutes x.
        # (mode 3: the mask bytes written above stand in for the activation of mode 2)
        out.bnb = (None if out is y else y, coeffs,
                   3 if relu_mask is not None else (2 if residual is not None else 1), relu_mask)
    x_bnb = x.bnb if (owns_input_grad and BNBWD_FUSE and x.requires_grad) else None
    if not tape.enabled:
        return out

    def backward():
        dz = out.g
        if dz is None:
            return
        s = current_stream_ptr()
        mask = 0 if not relu else (2 if residual is not None else 1)
        sums_ready = out.bnb_sums is not None    # a consumer's dgrad epilogue did the reduction
        want_g = residual is not None and residual.requires_grad and relu
        wgrad_bn = bn.weight is not None and bn.weight.requires_grad
        bgrad_bn = bn.bias is not None and bn.bias.requires_grad
        ggamma = ensure_grad(bn.weight) if wgrad_bn else None
        gbeta = ensure_grad(bn.bias) if bgrad_bn else None
        gw = ensure_grad(weight) if weight.requires_grad else None
        dy = torch.empty_like(y.t)
        bsums_ptr = coeffs.data_ptr() + 16 * C
        acc = 0
        dx_ptr = None
        if x.requires_grad:
            acc = 1 if x.g is not None else 0
            if not acc:
                x.new_grad() if x.parent is None else _alloc_parent_grad(x)
            dx_ptr = x.g.data_ptr()
        fuse, fused_flag = None, None
        if x_bnb is not None and dx_ptr is not None and x.parent is None:
            py, pcoeffs, pmode, pmask = x_bnb
            if py is None:
                py = x
            pact = x if pmode == 2 else None
            fused_flag = ctypes.c_int32(0)
            fuse = _lib.BnBwdFuse()
            fuse.y, fuse.ldy = py.ptr, py.ld
            fuse.act, fuse.ldact = (pact.ptr, pact.ld) if pact is not None else (None, 0)
            pc = pcoeffs.data_ptr()
            fuse.coeffs, fuse.sums = pc, pc + 16 * (pcoeffs.numel() // 6)   # the producer's sums slot
            fuse.fused = ctypes.pointer(fused_flag)
            fuse.mode, fuse.reserved = pmode, 0
            fuse.mask, fuse.ldmask, fuse.reserved2 = (
                (pmask.data_ptr(), pmask.shape[1], 0) if pmode == 3 else (None, 0, 0))
        ws_b = _ws.get(need, dev)
        queued = SIDE_WGRAD and gw is not None
        d.in_affine = in_affine.data_ptr() if in_affine is not None else None
        _bn_pointers(args, bn)
        _lib.check(L.gs_conv_bn_backward(
            ctypes.byref(d), x.ptr, weight.data_ptr(), y.ptr, out.ptr, out.ld, coeffs.data_ptr(),
            ctypes.byref(args), dz.data_ptr(), dz.stride(2), mask, 1 if want_g else 0,
            dy.data_ptr(), bsums_ptr, ggamma.data_ptr() if wgrad_bn else None,
            gbeta.data_ptr() if bgrad_bn else None,
            gw.data_ptr() if (gw is not None and not queued) else None,
            dx_ptr, acc, ws_b.data_ptr(), ws_b.numel(), None, 0, s, None,
            ctypes.byref(fuse) if fuse is not None else None, 1 if sums_ready else 0),
            "gs_conv_bn_backward")
        if fuse is not None and fused_flag.value:
            global BNBWD_FUSED_COUNT, BNBWD_FUSED_MASK_COUNT
            BNBWD_FUSED_COUNT += 1
            if fuse.mode == 3:
                BNBWD_FUSED_MASK_COUNT += 1
            x.bnb_sums = True        # x.g now holds the MASKED gradient of the producer's ReLU
        if wgrad_bn:
            _notify(bn.weight)
        if bgrad_bn:
            _notify(bn.bias)
        if queued:
            queue_wgrad(d, x, dy, weight, gw, need, dev)     # the side stream gets it in a batch
        elif gw is not None:
            _notify(weight)
        if residual is not None and residual.requires_grad:
            src = dz  # masked (relu) or plain (no relu) upstream gradient
            if residual.g is None and residual.parent is None and src.stride() == residual.t.stride():
                residual.g = src
            else:
                if residual.g is None:
                    residual.new_grad() if residual.parent is None else _alloc_parent_grad(residual)
                    racc = 0 if residual.parent is None else 1
                else:
                    racc = 1
                _lib.check(L.gs_copy2d(src.data_ptr(), src.stride(2), residual.g.data_ptr(),
                                       residual.g.stride(2), rows, C, 1.0, racc, s), "gs_copy2d")

    tape.record(backward)
    return out


def maxpool(tape, x, k=3, s=2, p=1):
    L = _L()
    dev = x.t.device
    ho, wo = (x.H + 2 * p - k) // s + 1, (x.W + 2 * p - k) // s + 1
    out = Act.empty(x.N, ho, wo, x.C, dev)
    idx = torch.empty((x.N, ho, wo, x.C), dtype=torch.uint8, device=dev)
    _lib.check(L.gs_maxpool_forward(x.ptr, x.N, x.H, x.W, x.C, x.ld, k, s, p, ho, wo, out.ptr,
                                    out.ld, idx.data_ptr(), current_stream_ptr()),
               "gs_maxpool_forward")
    if POOL_TRACE is not None:
        POOL_TRACE.append(idx)

    def backward():
        dy = out.g
        if dy is None or not x.requires_grad:
            return
        acc = x.g is not None
        if not acc:
            x.new_grad() if x.parent is None else _alloc_parent_grad(x)
        _lib.check(L.gs_maxpool_backward(dy.data_ptr(), dy.stride(2), idx.data_ptr(), x.N, x.H, x.W,
                                         x.C, k, s, p, ho, wo, x.g.data_ptr(), x.g.stride(2),
                                         1 if acc else 0, current_stream_ptr()),
                   "gs_maxpool_backward")

    tape.record(backward)
    return out


def avgpool_ceil(tape, x, s):
    """nn.AvgPool2d(s, stride=s, ceil_mode=True, count_include_pad=False) (avg_down shortcut)."""
    L = _L()
    x = materialize(tape, x)
    dev = x.t.device
    ho, wo = (x.H + s - 1) // s, (x.W + s - 1) // s
    out = Act.empty(x.N, ho, wo, x.C, dev)
    _lib.check(L.gs_avgpool_ceil_forward(x.ptr, x.N, x.H, x.W, x.C, x.ld, s, out.ptr, out.ld,
                                         current_stream_ptr()), "gs_avgpool_ceil_forward")

    def backward():
        dy = out.g
        if dy is None or not x.requires_grad:
            return
        acc = x.g is not None
        if not acc:
            x.new_grad() if x.parent is None else _alloc_parent_grad(x)
        _lib.check(L.gs_avgpool_ceil_backward(dy.data_ptr(), dy.stride(2), x.N, x.H, x.W, x.C, s,
                                              x.g.data_ptr(), x.g.stride(2), 1 if acc else 0,
                                              current_stream_ptr()), "gs_avgpool_ceil_backward")

    tape.record(backward)
    return out


def copy_into(tape, x, dst):
    """dst[..., :x.C] = x  (first member of a fused concat)."""
    L = _L()
    _lib.check(L.gs_copy2d(x.ptr, x.ld, dst.ptr, dst.ld, x.rows, x.C, 1.0, 0, current_stream_ptr()),
               "gs_copy2d")

    def backward():
        dg = dst.g
        if dg is None or not x.requires_grad:
            return
        acc = x.g is not None
        if not acc:
            x.new_grad() if x.parent is None else _alloc_parent_grad(x)
        _lib.check(L.gs_copy2d(dg.data_ptr(), dg.stride(2), x.g.data_ptr(), x.g.stride(2), x.rows,
                               x.C, 1.0, 1 if acc else 0, current_stream_ptr()), "gs_copy2d")

    tape.record(backward)
    return dst


def adaptive_avgpool(tape, x, scales):
    """All PPM pool scales in one read of x. Returns one Act [N,s,s,C] per scale."""
    L = _L()
    dev = x.t.device
    ns = len(scales)
    arr = _int32_array(ns)(*scales)
    total_bins = sum(s * s for s in scales)
    ybuf = torch.empty(x.N * total_bins * x.C, dtype=torch.float32, device=dev)
    nb = L.gs_adaptive_avgpool_workspace_bytes(x.N, x.H, x.W, x.C, arr, ns)
    ws = _ws.get(nb, dev)
    _lib.check(L.gs_adaptive_avgpool_forward(x.ptr, x.N, x.H, x.W, x.C, x.ld, arr, ns,
                                             ybuf.data_ptr(), ws.data_ptr(), ws.numel(),
                                             current_stream_ptr()), "gs_adaptive_avgpool_forward")
    outs, off = [], 0
    for s in scales:
        n_el = x.N * s * s * x.C
        outs.append(Act(ybuf[off:off + n_el].view(x.N, s, s, x.C)))
        off += n_el

    def backward():
        if not x.requires_grad:
            return
        if all(o.g is None for o in outs):
            return
        gbuf = torch.zeros_like(ybuf)
        off2 = 0
        for s, o in zip(scales, outs):
            n_el = x.N * s * s * x.C
            if o.g is not None:
                gbuf[off2:off2 + n_el].view(x.N, s, s, x.C).copy_(o.g)
            off2 += n_el
        acc = x.g is not None
        if not acc:
            x.new_grad() if x.parent is None else _alloc_parent_grad(x)
        _lib.check(L.gs_adaptive_avgpool_backward(gbuf.data_ptr(), x.N, x.H, x.W, x.C, arr, ns,
                                                  x.g.data_ptr(), x.g.stride(2), 1 if acc else 0,
                                                  current_stream_ptr()),
                   "gs_adaptive_avgpool_backward")

    tape.record(backward)
    return outs


def bilinear(tape, x, size, align_corners=False, out=None, accumulate=False):
    """mmseg.ops.resize(mode='bilinear').  ``out`` may be a channel slice of a concat buffer;
    ``accumulate`` adds into ``out`` (UPer top-down path)."""
    L = _L()
    dev = x.t.device
    ho, wo = int(size[0]), int(size[1])
    if out is None:
        out = Act.empty(x.N, ho, wo, x.C, dev)
    al = 1 if align_corners else 0
    _lib.check(L.gs_bilinear_forward(x.ptr, x.N, x.H, x.W, x.C, x.ld, ho, wo, al, out.ptr, out.ld,
                                     1 if accumulate else 0, current_stream_ptr()),
               "gs_bilinear_forward")

    def backward():
        dy = out.g
        if dy is None or not x.requires_grad:
            return
        acc = x.g is not None
        if not acc:
            x.new_grad() if x.parent is None else _alloc_parent_grad(x)
        nb = L.gs_bilinear_backward_workspace_bytes(x.N, x.H, x.W, x.C, ho, wo)
        ws = _ws.get(nb, dev)
        _lib.check(L.gs_bilinear_backward(dy.data_ptr(), dy.stride(2), x.N, x.H, x.W, x.C, ho, wo,
                                          al, x.g.data_ptr(), x.g.stride(2), 1 if acc else 0,
                                          ws.data_ptr(), ws.numel(), current_stream_ptr()),
                   "gs_bilinear_backward")

    tape.record(backward)
    return out


def dropout2d(tape, x, p, training, generator=None):
    """nn.Dropout2d: one Bernoulli(1-p) draw per (n, channel), scaled by 1/(1-p)."""
    if not training or p <= 0.0:
        return x
    L = _L()
    dev = x.t.device
    keep = 1.0 - p
    mask = torch.empty((x.N, x.C), dtype=torch.float32, device=dev)
    mask.bernoulli_(keep, generator=generator).div_(keep)
    out = Act.empty(x.N, x.H, x.W, x.C, dev)
    ppi = x.H * x.W
    _lib.check(L.gs_scale_nc(x.ptr, x.ld, mask.data_ptr(), x.N, ppi, x.C, out.ptr, out.ld,
                             current_stream_ptr()), "gs_scale_nc")

    def backward():
        dy = out.g
        if dy is None or not x.requires_grad:
            return
        if x.g is not None:
            raise RuntimeError("dropout input has more than one consumer; unsupported")
        x.new_grad() if x.parent is None else _alloc_parent_grad(x)
        _lib.check(L.gs_scale_nc(dy.data_ptr(), dy.stride(2), mask.data_ptr(), x.N, ppi, x.C,
                                 x.g.data_ptr(), x.g.stride(2), current_stream_ptr()),
                   "gs_scale_nc")

    tape.record(backward)
    return out


def add_grad_passthrough(tape, src, dst):
    """Record that ``dst`` is the same values as ``src`` (identity edge): route dst.g into src.g."""
    L = _L()

    def backward():
        dg = dst.g
        if dg is None or not src.requires_grad:
            return
        if src.g is None and src.parent is None and dg.stride() == src.t.stride():
            src.g = dg
            return
        acc = src.g is not None
        if not acc:
            src.new_grad() if src.parent is None else _alloc_parent_grad(src)
            acc = src.parent is not None
        _lib.check(L.gs_copy2d(dg.data_ptr(), dg.stride(2), src.g.data_ptr(), src.g.stride(2),
                               src.rows, src.C, 1.0, 1 if acc else 0, current_stream_ptr()),
                   "gs_copy2d")

    tape.record(backward)
