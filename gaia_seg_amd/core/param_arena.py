"""Flat fp32 arenas for parameters, gradients and SGD momentum (laid out for 288 GB of HBM3E).

Every parameter of the supernet becomes a view into ONE contiguous buffer (and its gradient a view
into a second one with identical offsets), in forward order.  This gives
  * a fused SGD step over a few merged ranges instead of ~300 small tensors (K18: torch.optim.SGD of
    configs/_dynamic_/models/pspnet_ar50to101v2_gsync.py:175 — momentum 0.9, weight decay 5e-4),
  * zero-copy gradient buckets for the data-parallel all-reduce (the reducer all-reduces slices of
    the flat gradient buffer; nothing is packed or unpacked),
  * one memset to clear the gradients of a step.
Only the parameters that take part in the sampled subnet are touched by a step: depth-skipped
blocks get no gradient and no update at all (DECIDE); inactive width
slices of used tensors have zero gradient but do receive weight decay / momentum, like the
reference (they belong to a tensor that has a gradient).
"""
import torch

from ..hip import lib as _lib
from ..hip.runtime import current_stream_ptr, round_up
from .bricks import hwio_logical_view

_ALIGN = 64  # floats: 256-byte aligned segments (float4 loads need 16 B; keep cache lines whole)


def arena_layout(model):
    """[(name, parameter, physical shape or None, offset, physical numel)], total elements: where every
    parameter sits in the flat arenas (forward order, 256-byte aligned segments).  Host arithmetic
    only — `bench.py --plan-only` sizes the gradient buckets from it without a GPU."""
    off, layout = 0, []
    for name, p in model.named_parameters():
        phys = getattr(p, "_gs_phys_shape", None)
        n_phys = 1
        for s in (phys if phys is not None else p.shape):
            n_phys *= s
        layout.append((name, p, phys, off, n_phys))
        off += round_up(max(n_phys, 1), _ALIGN)
    return layout, off


class ParamArena:
    def __init__(self, model):
        from .bricks import DynamicConv2d, relayout_conv_params
        for m in model.modules():  # (re)attach the physical-layout descriptors (lost by deepcopy)
            if isinstance(m, DynamicConv2d):
                relayout_conv_params(m)
        params = [(n, p) for n, p in model.named_parameters()]
        if not params:
            raise ValueError("model has no parameters")
        dev = params[0][1].device
        if dev.type != "cuda":
            raise _lib.HipLibraryError("ParamArena needs the model on the MI355X (got %s)" % dev)
        self.device = dev
        self.segments = {}   # id(param) -> (offset, numel_phys)
        self.names = {}
        layout, off = arena_layout(model)
        self.numel = off
        self.flat_param = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_mom = torch.zeros(off, dtype=torch.float32, device=dev)
        self._layout = [(name, p, phys, o, n) for name, p, phys, o, n in layout]
        for name, p, phys, o, n in layout:
            pv = self._view(self.flat_param, p, phys, o, n)
            pv.copy_(p.data)
            p.data = pv
            p.grad = self._view(self.flat_grad, p, phys, o, n)
            # gradients now always exist: drop the lazy factories
            if hasattr(p, "_gs_grad_factory"):
                del p._gs_grad_factory
            self.segments[id(p)] = (o, round_up(max(n, 1), _ALIGN))
            self.names[id(p)] = name
        self._range_cache = {}
        # True while every gradient element is zero outside a running step (see zero_grad)
        self.grads_clean = False

    @staticmethod
    def _view(flat, p, phys, off, n):
        seg = flat[off:off + n]
        if phys is None:
            return seg.view(p.shape)
        if len(phys) == 4:   # conv weight, physical HWIO
            return hwio_logical_view(seg.view(*phys), p.shape[0])
        if len(phys) == 1:   # padded conv bias
            return seg[:p.shape[0]]
        raise ValueError("unknown physical layout %s" % (phys,))

    # ---- ranges ----
    def ranges_for(self, params, key=None):
        """Merged [begin, end) element ranges covering ``params`` (cached by ``key``)."""
        if key is not None and key in self._range_cache:
            return self._range_cache[key]
        segs = sorted(self.segments[id(p)] for p in params)
        merged = []
        for o, n in segs:
            if merged and merged[-1][1] == o:
                merged[-1][1] = o + n
            else:
                merged.append([o, o + n])
        merged = [tuple(r) for r in merged]
        if key is not None:
            self._range_cache[key] = merged
        return merged

    def zero_grad(self, ranges=None, trust_clean=False):
        """Clear gradients.  ``trust_clean`` (passed only by IterBasedRunner, which owns the
        invariant): a no-op while ``grads_clean`` holds -- the buffer starts at zero and every
        sgd_step(zero_grad=True) clears what the step wrote, so a training loop that always follows
        backward with such a step over the same ranges never needs the fill kernels.  Any other caller
        (a manual train_step + backward in a test, a tool, a val workflow) gets a real clear: the flag
        says nothing about gradients produced outside the runner."""
        if trust_clean and self.grads_clean:
            return
        if ranges is None:
            self.flat_grad.zero_()
        else:
            for a, b in ranges:
                self.flat_grad[a:b].zero_()

    def sgd_step(self, ranges, lr, momentum=0.9, weight_decay=5e-4, grad_scale=1.0, zero_grad=False,
                 hyper=None):
        """torch.optim.SGD(momentum, weight_decay, dampening=0, nesterov=False) on the ranges.
        ``hyper``: device tensor {lr, momentum, weight_decay, grad_scale} read by the kernel at run
        time instead of the by-value arguments (step graphs, core/runner.py)."""
        L = _lib.load()
        st = current_stream_ptr()
        pb, gb, mb = self.flat_param.data_ptr(), self.flat_grad.data_ptr(), self.flat_mom.data_ptr()
        if hyper is not None:
            hp = hyper.data_ptr()
            for a, b in ranges:
                _lib.check(L.gs_sgd_step_hyper(pb + 4 * a, gb + 4 * a, mb + 4 * a, b - a, hp,
                                               1 if zero_grad else 0, st), "gs_sgd_step_hyper")
            return
        for a, b in ranges:
            _lib.check(L.gs_sgd_step(pb + 4 * a, gb + 4 * a, mb + 4 * a, b - a, lr, momentum,
                                     weight_decay, grad_scale, 1 if zero_grad else 0, st),
                       "gs_sgd_step")

    def momentum_views(self):
        """{parameter name: momentum buffer as a LOGICAL (OIHW / plain) view of the flat arena}."""
        return {name: self._view(self.flat_mom, p, phys, o, n) for name, p, phys, o, n in self._layout}

    def state_dict(self):
        """Optimizer state keyed by parameter name in the logical layout (contiguous OIHW for conv
        weights), so that neither the arena offsets nor the physical HWIO layout leak into a
        checkpoint (tools/train_supernet.py:197-202 stores `optimizer` next to `state_dict`)."""
        return {"format": "gaia_seg_amd.sgd_momentum.v1",
                "state": {k: v.detach().clone().contiguous() for k, v in self.momentum_views().items()}}

    def load_state_dict(self, sd, logger=None):
        """Accepts this build's format; any other optimizer state (e.g. a torch.optim.SGD state_dict
        of an mmcv checkpoint, whose integer keys carry no parameter names) is skipped with a warning
        and the momentum restarts from zero."""
        import warnings
        state = sd.get("state") if isinstance(sd, dict) else None
        if not isinstance(sd, dict) or sd.get("format") != "gaia_seg_amd.sgd_momentum.v1" \
                or not isinstance(state, dict):
            msg = "optimizer state of a foreign format: momentum buffers restart from zero"
            (logger.warning if logger is not None else warnings.warn)(msg)
            return False
        views = self.momentum_views()
        missing = [k for k in views if k not in state]
        with torch.no_grad():
            for k, v in state.items():
                if k in views:
                    if tuple(views[k].shape) != tuple(v.shape):
                        raise RuntimeError("momentum of %s: checkpoint %s vs model %s"
                                           % (k, tuple(v.shape), tuple(views[k].shape)))
                    views[k].copy_(v)
        if missing:
            msg = "optimizer state lacks %d parameters (e.g. %s): their momentum is zero" % (
                len(missing), missing[0])
            (logger.warning if logger is not None else warnings.warn)(msg)
        return True
