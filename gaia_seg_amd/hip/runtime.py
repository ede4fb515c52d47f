"""Execution runtime of the MI355X path: NHWC activations, a hand-run backward tape, workspaces.

Design (MI355X-first, not a translation of the reference's per-op autograd graph):

* Activations live in HBM as NHWC fp32 (`Act`), possibly as a channel slice of a wider buffer so
  that ``torch.cat`` never materialises (PPM / FCN / FPN concat fusion).
* A module-level forward records closures on a `Tape`; the whole backward of a backbone or head is
  ONE autograd node (`tape_function`) that replays the tape in reverse and writes parameter
  gradients straight into ``param.grad`` (views of the flat gradient arena when one is installed).
  Torch autograd only stitches the 4-5 coarse nodes of a step together.
* Every kernel is launched on torch's current HIP stream through the C-ABI; nothing here computes
  on the CPU and nothing falls back to eager PyTorch.
"""
import torch

from . import lib as _lib


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def current_stream_ptr():
    """hipStream_t of torch's current stream on the current device (raw handle: this is called
    once per kernel launch, torch.cuda.current_stream() costs ~9 us of host time each)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def require_gpu_tensor(t, what):
    if not t.is_cuda:
        raise _lib.HipLibraryError(
            "%s is on %s: the gaia_seg_amd operators run only as HIP kernels on an MI355X "
            "(there is no CPU fallback)" % (what, t.device))
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32, got %s" % (what, t.dtype))


class _Workspace:
    """Grow-only scratch buffer per device and stream slot (split-K slabs, reduction partials).

    All kernels of one in-order queue share a buffer.  ``slot`` selects the queue: 0 = the training
    stream, 1 = the branch stream (hip/ops.py: projection shortcuts, the auxiliary head); whoever
    switches the current stream switches the slot with it."""

    def __init__(self):
        self._buf = {}
        self._retired = []   # outgrown buffers: captured step graphs may still point into them
        self.slot = 0

    def get(self, nbytes, device):
        nbytes = max(int(nbytes), 256)
        key = (device.type, device.index, self.slot)
        buf = self._buf.get(key)
        if buf is None or buf.numel() < nbytes:
            # round up generously so later, larger requests rarely reallocate
            size = max(nbytes, 1 << 20)
            size = 1 << (size - 1).bit_length()
            if buf is not None:
                self._retired.append(buf)
            buf = torch.empty(size, dtype=torch.uint8, device=device)
            self._buf[key] = buf
        return buf

    def reserve(self, nbytes, device):
        self.get(nbytes, device)


WORKSPACE = _Workspace()


def round_up(x, m):
    return (x + m - 1) // m * m


class Act:
    """An NHWC activation: ``t`` has shape [N, H, W, C] with strides (H*W*ld, W*ld, ld, 1).

    ``g`` is its gradient (same shape / layout), filled during the backward replay.
    ``nchw_image`` marks the network input, which is read in place through NCHW strides."""

    __slots__ = ("t", "_g", "requires_grad", "parent", "c0", "nchw_image",
                 "N", "H", "W", "C", "ld", "rows", "ptr", "affine", "bnb", "bnb_sums", "res_affine")

    def __init__(self, t, requires_grad=True, parent=None, c0=0, nchw_image=False):
        self.t = t
        self._g = None
        self.requires_grad = requires_grad
        self.parent = parent
        self.c0 = c0
        self.nchw_image = nchw_image
        # Deferred BatchNorm + ReLU: when set (a coefficient tensor [scale | beta | mean | invstd][C]),
        # the LOGICAL value of this activation is relu(bn(t)) and ``t`` holds the BN input; the
        # consumer convolution applies it in its operand loader (gs_conv_desc.in_affine) or
        # ops.materialize() writes it out.  ``g`` is always the gradient of the logical value.
        self.affine = None
        # Deferred BatchNorm WITHOUT activation (the projection shortcut): when set (a coefficient
        # tensor), the logical value is bn(t); its only consumer, the residual add of the block's last
        # BatchNorm, applies it (gs_bn_args.residual_coeffs).
        self.res_affine = None
        # Cross-layer fusion of the BatchNorm backward reduction (ops.conv_bn): ``bnb`` = (y Act,
        # coefficient tensor, mask mode) describes the BN + ReLU that
        # produced this activation; a consumer whose data gradient is the last contribution to ``g``
        # may fold that BN's reduction into its dgrad epilogue and leaves the sums in ``bnb_sums``.
        self.bnb = None
        self.bnb_sums = None
        # geometry, fixed for the life of the object (t is never rebound): plain attributes, these
        # are read several times per kernel launch
        shp = t.shape
        self.N, self.H, self.W, self.C = shp[0], shp[1], shp[2], shp[3]
        self.ld = t.stride(2)
        self.rows = shp[0] * shp[1] * shp[2]
        self.ptr = t.data_ptr()

    # ---- gradient ----
    @property
    def g(self):
        if self.parent is not None:
            pg = self.parent.g
            return None if pg is None else pg[..., self.c0:self.c0 + self.C]
        return self._g

    @g.setter
    def g(self, value):
        if self.parent is not None:
            raise RuntimeError("gradient of a channel slice is owned by its parent buffer")
        self._g = value

    def new_grad(self):
        """Allocate (uninitialised) gradient storage with the layout of ``t``."""
        if self.parent is not None:
            raise RuntimeError("allocate the gradient on the parent buffer")
        n, h, w, c = self.t.shape
        ld = self.t.stride(2)
        if ld == c:
            self._g = torch.empty((n, h, w, c), dtype=self.t.dtype, device=self.t.device)
        else:
            # padded pixel stride (e.g. 19 classes in 20 floats): the pad columns must stay zero
            # because the kernels move whole float4s
            self._g = torch.zeros((n, h, w, ld), dtype=self.t.dtype, device=self.t.device)[..., :c]
        return self._g

    def slice(self, c0, c1):
        """Channel slice [c0, c1) as an Act whose gradient is a view of this one's."""
        return Act(self.t[..., c0:c1], self.requires_grad, parent=self, c0=c0)

    # ---- boundary conversions (logical NCHW tensors at module boundaries) ----
    @staticmethod
    def empty(N, H, W, C, device, ld=None, requires_grad=True):
        ld = ld or round_up(C, 4)
        buf = torch.empty((N, H, W, ld), dtype=torch.float32, device=device)
        return Act(buf if ld == C else buf[..., :C], requires_grad)

    @staticmethod
    def from_nchw(x, requires_grad=None):
        """Wrap a logical NCHW tensor. Channels-last storage is used in place (no copy); the
        3-channel image stays NCHW and is read through strides by the stem kernel."""
        require_gpu_tensor(x, "input tensor")
        rg = x.requires_grad if requires_grad is None else requires_grad
        n, c, h, w = x.shape
        if c % 4 != 0 and x.stride(3) == 1 and not rg:
            return Act(x.detach(), False, nchw_image=True)
        xd = x.detach()
        nhwc = xd.permute(0, 2, 3, 1)
        ok = (nhwc.stride(3) == 1 and nhwc.stride(2) % 4 == 0 and nhwc.stride(2) >= c
              and nhwc.stride(1) == w * nhwc.stride(2) and nhwc.stride(0) == h * nhwc.stride(1)
              and nhwc.data_ptr() % 16 == 0)
        if not ok:
            a = Act.empty(n, h, w, c, x.device)
            a.t.copy_(nhwc)
            a.requires_grad = rg
            return a
        return Act(nhwc, rg)

    def as_nchw(self):
        return self.t.permute(0, 3, 1, 2)

    def set_grad_from_nchw(self, grad):
        """Install an upstream gradient given as a logical NCHW tensor (alias when layouts match)."""
        gn = grad.permute(0, 2, 3, 1)
        same = (gn.stride() == self.t.stride() and gn.data_ptr() % 16 == 0
                and gn.dtype == torch.float32)
        cur = self.g
        if cur is None:
            if self.parent is not None:
                self.parent.new_grad().zero_()
                self.g.copy_(gn)
            elif same:
                self._g = gn
            else:
                self.new_grad().copy_(gn)
        else:
            cur.add_(gn)


class Tape:
    """Backward closures in forward order; ``backward`` replays them in reverse."""

    __slots__ = ("ops", "enabled")

    def __init__(self, enabled=True):
        self.ops = []
        self.enabled = enabled

    def record(self, fn):
        if self.enabled:
            self.ops.append(fn)

    def backward(self):
        ops = self.ops
        self.ops = []
        for fn in reversed(ops):
            fn()
        from . import ops as _ops
        if _ops.DEFER_JOIN:
            _ops.flush_wgrads()        # the caller joins (runner: after the early part of the SGD step)
        else:
            _ops.join_side_streams()   # weight gradients queued on the side stream


# While a training step is being captured into a HIP graph (core/runner.py) host-side effects that a
# replay must repeat are logged here (BatchNorm layers whose num_batches_tracked counts the step).
CAPTURE_LOG = None

BACKWARD_PROFILE = None
if __import__("os").environ.get("GS_CPROFILE"):
    import cProfile
    BACKWARD_PROFILE = cProfile.Profile()


class _TapeFunction(torch.autograd.Function):
    """One autograd node for a whole module forward (backbone, head).

    ``runner(tape, acts_in) -> list[Act]``.  Parameter gradients are written by the tape straight
    into ``param.grad``; the ``anchor`` input only makes autograd schedule this node."""

    @staticmethod
    def forward(ctx, runner, need_tape, anchor, *inputs):
        # (grad mode is always off inside Function.forward: the caller decides need_tape)
        # outputs nobody consumes (e.g. the stage-1/2 feature maps under an FCN head) must arrive as
        # None in backward, not as materialised zero tensors that would be filled, copied into NHWC
        # and added to the real gradients (150 us per step in the r01 trace)
        ctx.set_materialize_grads(False)
        tape = Tape(enabled=need_tape)
        acts_in = [Act.from_nchw(t) for t in inputs]
        outs = runner(tape, acts_in)
        ctx.tape = tape
        ctx.acts_in = acts_in
        ctx.outs = outs
        results = tuple(o.as_nchw() for o in outs)
        return results

    @staticmethod
    def backward(ctx, *grads):
        if BACKWARD_PROFILE is not None:   # diagnostics (GS_CPROFILE): autograd runs this thread
            BACKWARD_PROFILE.enable()
            try:
                return _TapeFunction._backward(ctx, *grads)
            finally:
                BACKWARD_PROFILE.disable()
        return _TapeFunction._backward(ctx, *grads)

    @staticmethod
    def _backward(ctx, *grads):
        # autograd runs a node's backward on the stream its forward ran on: a head evaluated on the
        # branch stream (ops.branch_scope) comes back here with that stream current
        from . import ops as _ops
        prev_slot = _ops.adopt_current_stream()
        try:
            return _TapeFunction._backward_on_stream(ctx, *grads)
        finally:
            _ops.restore_stream_slot(prev_slot)

    @staticmethod
    def _backward_on_stream(ctx, *grads):
        for o, g in zip(ctx.outs, grads):
            if g is not None:
                o.set_grad_from_nchw(g)
        ctx.tape.backward()
        in_grads = []
        for a in ctx.acts_in:
            if a.requires_grad and a.g is not None:
                in_grads.append(a.g.permute(0, 3, 1, 2))
            else:
                in_grads.append(None)
        ctx.outs = None
        ctx.acts_in = None
        return (None, None, None) + tuple(in_grads)


_ANCHORS = {}


def grad_anchor(device):
    """A scalar that requires grad: lets a node whose tensor inputs need no gradient (the image)
    still take part in backward so that its parameters receive gradients."""
    key = (device.type, device.index)
    a = _ANCHORS.get(key)
    if a is None:
        a = torch.zeros((), device=device, requires_grad=True)
        _ANCHORS[key] = a
    return a


def tape_function(runner, inputs, needs_param_grad):
    """Run ``runner`` as one autograd node. ``inputs`` are logical NCHW tensors."""
    inputs = list(inputs)
    for t in inputs:
        require_gpu_tensor(t, "input tensor")
    grad_on = torch.is_grad_enabled()
    anchor = grad_anchor(inputs[0].device) if (needs_param_grad and grad_on) else None
    need_tape = grad_on and (anchor is not None or any(t.requires_grad for t in inputs))
    return _TapeFunction.apply(runner, need_tape, anchor, *inputs)
