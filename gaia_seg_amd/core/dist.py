"""Data-parallel gradient exchange for supernet training: one process per GPU, RCCL over xGMI.

Replaces ``MMDistributedDataParallel(find_unused_parameters=True, broadcast_buffers=False)``
(gaiaseg/apis/train.py:88-96) for this path.  Differences by design (MI355X-first):

* zero-copy buckets: gradients already live in the flat gradient arena, so a bucket is just a
  slice of it — no pack / unpack kernels, no bucket views to rebuild;
* only the parameters of the sampled subnet are reduced.  Every rank runs the same subnet (the arch
  meta is broadcast from rank 0, see ManipulateArchHook), so depth-skipped blocks need no traffic
  and no ``find_unused_parameters`` graph walk: the R50 anchor of the in-tree PSP supernet moves
  ~1/2 of the 457 MB the reference all-reduces every step;
* overlap with backward: the backward tape reports each parameter as soon as its gradient is
  final; when a bucket is complete its all-reduce is enqueued immediately.  torch.distributed's
  RCCL backend runs collectives on its own HIP stream, fenced by events against the compute
  stream, so the exchange overlaps the rest of backward; ``finish()`` makes the compute stream
  wait before the optimizer step;
* gradients are summed; the 1/world_size factor is folded into the fused SGD kernel
  (``grad_scale``);
* BN running statistics are NOT broadcast (broadcast_buffers=False in the reference).

The same code runs over gloo on CPU tensors, which is how the multi-process tests exercise it.
"""
import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def world_size():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def broadcast_object(obj, src=0):
    """gaiavision ``broadcast_object`` (call sites cross_arch_eval_hooks.py:59,140): pickled
    python object from ``src`` to every rank."""
    if not is_dist():
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src, group=_host_group())
    return box[0]


def sync_module_states(model, arena=None, src=0):
    """The wrap-time broadcast of MMDistributedDataParallel (gaiaseg/apis/train.py:88-96): every
    rank starts from rank ``src``'s parameters AND buffers.  (``broadcast_buffers=False`` there only
    disables the per-forward re-broadcast; the reference launch passes no --seed, so without this
    every rank would train its own random initialisation.)  One broadcast for the parameter arena,
    one per buffer dtype."""
    if not is_dist():
        return
    with torch.no_grad():
        if arena is not None:
            dist.broadcast(arena.flat_param, src=src)
        else:
            for p in model.parameters():
                t = p.data.contiguous()
                dist.broadcast(t, src=src)
                p.data.copy_(t)
        by_dtype = {}
        for b in model.buffers():
            by_dtype.setdefault(b.dtype, []).append(b)
        for bufs in by_dtype.values():
            flat = torch.cat([b.reshape(-1) for b in bufs])
            dist.broadcast(flat, src=src)
            off = 0
            for b in bufs:
                b.copy_(flat[off:off + b.numel()].view(b.shape))
                off += b.numel()


def gather_objects(obj):
    """Every rank's python object, in rank order, on every rank (host group: no device sync)."""
    if not is_dist():
        return [obj]
    out = [None] * world_size()
    dist.all_gather_object(out, obj, group=_host_group())
    return out


_HOST_GROUP = None


def _host_group():
    """A gloo group for small host objects (the per-step subnet meta).  Broadcasting a pickled
    object over the RCCL group moves its size to the GPU and reads it back, i.e. one full
    host<->GPU synchronisation per training step: the host could never run ahead of the GPU.
    Every rank reaches the first call at the same point (collective creation)."""
    global _HOST_GROUP
    if _HOST_GROUP is None:
        if dist.get_backend() == "gloo":
            _HOST_GROUP = dist.group.WORLD
        else:
            try:
                _HOST_GROUP = dist.new_group(backend="gloo")
            except Exception as e:  # no gloo transport on this box: correct but synchronising
                import warnings
                warnings.warn("gloo side group unavailable (%s): host objects travel over the "
                              "default group, which synchronises host and GPU once per step" % (e,))
                _HOST_GROUP = dist.group.WORLD
    return _HOST_GROUP


class GradReducer:
    """Bucketed all-reduce over slices of a flat gradient buffer, driven by per-parameter
    'gradient ready' notifications from the backward tape."""

    def __init__(self, flat_grad, segments, bucket_bytes=64 << 20, group=None, hole_frac=0.05):
        """flat_grad: 1-D tensor; segments: {id(param): (offset, numel)} (ParamArena.segments).
        hole_frac: holes inside a bucket (parameters of depth-skipped blocks: their gradient stays
        zero on every rank) are reduced along with it while they add up to less than this share of
        the bucket -- one collective for the bucket instead of one per contiguous run."""
        self.flat_grad = flat_grad
        self.segments = segments
        self.bucket_elems = max(1, bucket_bytes // flat_grad.element_size())
        self.group = group
        self.hole_frac = hole_frac
        self.collectives = 0          # launches handed to the backend (a coalesced group counts once)
        self._no_coalesce = False
        # GS_CHECK_HOLES=1 (debug): before a bucket goes out, assert that the padded holes inside its
        # runs (gradients of parameters the sampled subnet does not use) are exactly zero on this rank
        # -- a stale value there would be summed over the ranks every step and consumed when the block
        # becomes active (synchronises the host: diagnostics only)
        import os
        self.check_holes = os.environ.get("GS_CHECK_HOLES") == "1"
        self._plans = {}
        self._active = None
        self._works = []
        self.bytes_reduced = 0

    # ---- planning (cached per arch) ----
    def _plan(self, params, key):
        plan = self._plans.get(key) if key is not None else None
        if plan is not None:
            return plan
        # backward produces gradients roughly in reverse forward (= reverse arena) order:
        # walk the active segments from the end and cut buckets of ~bucket_elems
        segs = sorted(((self.segments[id(p)], p) for p in params), key=lambda t: -t[0][0])
        buckets, cur, cur_elems = [], [], 0
        for (o, n), p in segs:
            cur.append((o, n, p))
            cur_elems += n
            if cur_elems >= self.bucket_elems:
                buckets.append(cur)
                cur, cur_elems = [], 0
        if cur:
            buckets.append(cur)
        plan = []
        for b in buckets:
            # contiguous runs inside the bucket (skipped blocks leave holes)
            runs = []
            for o, n, _ in sorted((o, n, 0) for o, n, _ in b):
                if runs and runs[-1][1] == o:
                    runs[-1][1] = o + n
                else:
                    runs.append([o, o + n])
            plan.append(dict(param_ids=[id(p) for _, _, p in b],
                             runs=self.pad_holes([tuple(r) for r in runs], self.hole_frac)))
        if key is not None:
            self._plans[key] = plan
        return plan

    @staticmethod
    def pad_holes(runs, hole_frac):
        """Merge neighbouring runs of one bucket across small holes: smallest hole first, while the
        holes swallowed so far stay below hole_frac of the bucket's payload.  (The holes hold the
        gradients of parameters the sampled subnet does not use: zero on every rank, and the optimizer
        never touches them, so summing them changes nothing.)"""
        runs = sorted(runs)
        payload = sum(b - a for a, b in runs)
        budget = int(hole_frac * payload)
        holes = sorted((runs[i + 1][0] - runs[i][1], i) for i in range(len(runs) - 1))
        take, used = set(), 0
        for size, i in holes:
            if used + size > budget:
                break
            take.add(i)
            used += size
        out = []
        for i, (a, b) in enumerate(runs):
            if out and (i - 1) in take:
                out[-1] = (out[-1][0], b)
            else:
                out.append((a, b))
        return out

    def begin(self, params, key=None):
        """Arm the reducer for one backward pass over ``params`` (the active subnet)."""
        if world_size() == 1:      # nothing to exchange: no plan, no per-parameter hooks
            self._active = None
            self._works = []
            return
        plan = self._plan(params, key)
        pending, owner = [], {}
        for bi, b in enumerate(plan):
            pending.append(len(b["param_ids"]))
            for pid in b["param_ids"]:
                owner[pid] = bi
        self._active = dict(plan=plan, pending=pending, owner=owner, launched=[False] * len(plan))
        self._works = []
        for p in params:
            if p.requires_grad:
                p._gs_grad_ready = self._on_ready
            else:  # frozen: never reported, count it as done
                self._count(id(p))

    def _count(self, pid):
        st = self._active
        bi = st["owner"].get(pid)
        if bi is None:
            return
        st["pending"][bi] -= 1
        if st["pending"][bi] == 0 and not st["launched"][bi]:
            self._launch(bi)

    def _on_ready(self, param):
        if self._active is not None:
            self._count(id(param))

    def _launch(self, bi):
        st = self._active
        st["launched"][bi] = True
        if world_size() == 1:
            return
        runs = st["plan"][bi]["runs"]
        if self.flat_grad.is_cuda:
            # The bucket's weight gradients were queued on the side stream, its BN / bias gradients
            # on the main stream.  The collective must wait for both, the MAIN stream for neither:
            # the side stream takes a dependency on the main stream's work so far (one event) and
            # the all-reduce is issued in the side stream's context — torch's RCCL backend makes its
            # communication stream wait for the stream that is current at the call.  (r01 joined the
            # side stream into the main stream here, which serialised backward behind every bucket.)
            from ..hip import ops as _ops
            _ops.flush_wgrads()   # (no-op when this call comes from a flush: the queue is already empty)
            side = _ops.side_stream_after_main(self.flat_grad.device)
            with torch.cuda.stream(side):
                self._issue(runs)
        else:
            self._issue(runs)

    def _coalescing(self):
        """RCCL can take a bucket's runs as ONE grouped launch (ncclGroupStart / End behind
        ProcessGroupNCCL.allreduce_coalesced, reached through torch's coalescing manager); gloo cannot
        coalesce device tensors, so every other backend keeps one call per run."""
        if not self.flat_grad.is_cuda:
            return False
        try:
            return dist.get_backend(self.group) == "nccl"
        except Exception:
            return False

    def _issue(self, runs):
        """All-reduce one bucket.  A bucket is a single arena range unless depth-skipped blocks left
        holes too large to pad (pad_holes); several runs go out as one grouped RCCL launch where the
        backend supports it (tests/test_ddp_gpu.py exercises that branch on a one-rank RCCL group),
        else as one call per run."""
        tensors = [self.flat_grad[a:b] for a, b in runs]
        self.bytes_reduced += sum(t.numel() * t.element_size() for t in tensors)
        if self.check_holes:
            self._assert_holes_zero(runs)
        if len(tensors) > 1 and self._coalescing() and not self._no_coalesce:
            # torch's coalescing manager is a private API (and has only ever run on a one-rank RCCL
            # group in this pipeline): if it is missing or its signature has changed, fall back to one
            # call per run for the rest of the process instead of failing every multi-GPU step
            try:
                from torch.distributed.distributed_c10d import _coalescing_manager
                with _coalescing_manager(group=self.group, async_ops=True) as cm:
                    for t in tensors:
                        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            except (ImportError, TypeError, AttributeError) as e:
                import warnings
                warnings.warn("grouped RCCL launch unavailable (%s): one all-reduce per run" % (e,))
                self._no_coalesce = True
            else:
                self._works.append(cm)
                self.collectives += 1
                return
        for t in tensors:
            self._works.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group,
                                               async_op=True))
            self.collectives += 1

    def wait_launched(self):
        """Make the CURRENT stream wait for every bucket launched so far and return their arena runs
        (None on one rank: nothing is exchanged, every gradient is final as written).  Used by the
        early optimizer instalment, which may only touch what has been reduced."""
        st = self._active
        if st is None:
            return None
        for w in self._works:
            w.wait()
        self._works = []
        runs = []
        for bi, launched in enumerate(st["launched"]):
            if launched:
                runs.extend(st["plan"][bi]["runs"])
        return runs

    def _assert_holes_zero(self, runs):
        st = self._active
        if st is None:
            return
        active = sorted(self.segments[pid] for b in st["plan"] for pid in b["param_ids"])
        for a, b in runs:
            cur = a
            for o, n in active:
                if o + n <= a or o >= b:
                    continue
                if o > cur and float(self.flat_grad[cur:o].abs().max()) != 0.0:
                    raise AssertionError("non-zero gradient in the padded hole [%d, %d) of a bucket" % (cur, o))
                cur = max(cur, o + n)
            if cur < b and float(self.flat_grad[cur:b].abs().max()) != 0.0:
                raise AssertionError("non-zero gradient in the padded hole [%d, %d) of a bucket" % (cur, b))

    def finish(self):
        """Flush buckets whose parameters never reported (no gradient this step) and wait."""
        st = self._active
        if st is None:
            return
        for bi in range(len(st["plan"])):
            if not st["launched"][bi]:
                self._launch(bi)
        for w in self._works:
            w.wait()
        self._works = []
        self._active = None
