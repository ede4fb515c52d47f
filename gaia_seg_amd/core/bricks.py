"""Dynamic (slimmable) bricks: the host-side mirror of ``gaiavision.core`` ops/bricks.

None of these modules is in the reference tree (gaiavision is an absent dependency); their
contract is reconstructed from the reference call sites and restated in SURVEY.md Appendix A:

* ``DynamicConv2d`` ('DynConv2d')      — A1; call sites gaiaseg/models/decode_heads/dynamic_fcn_head.py:76,
  gaiaseg/models/backbones/dynamic_resnet.py:259-297 (via build_conv_layer).
* ``DynamicBatchNorm2d`` ('DynBN', 'DynSyncBN', 'BN', 'SyncBN') — A2; cfg
  configs/_dynamic_/models/pspnet_ar50to101v2_gsync.py:20-23,34,48.
* ``DynamicBottleneck``                — A3; gaiaseg/models/utils/dynamic_res_layer.py:105-125.
* ``DynamicConvModule``                — A4; dynamic_fcn_head.py:94-126, dynamic_psp_head.py:53-59.

Every forward runs hand-written HIP kernels (gaia_seg_amd/csrc) on NHWC activations; parameters
stay addressable as max-size OIHW tensors under the reference's state_dict names while their
physical layout is HWIO so the kernels read the active leading slice in place.
"""
import math
import warnings

import os

import torch
import torch.nn as nn
from torch.nn.modules.batchnorm import _BatchNorm

from ..hip import ops
from ..hip import runtime as _runtime
from ..hip.runtime import round_up, tape_function
from .dynamic import DynamicMixin
from .registry import Registry, build_from_cfg

CONV_LAYERS = Registry("conv layer")
NORM_LAYERS = Registry("norm layer")
ACTIVATION_LAYERS = Registry("activation layer")
ACTIVATION_LAYERS.register_module("ReLU", module=nn.ReLU)


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


# ------------------------------------------------------------------------------------------
# physical layouts
# ------------------------------------------------------------------------------------------
def hwio_logical_view(phys, co):
    """phys [KH, KW, Ci, Co_ld] -> logical OIHW view [co, Ci, KH, KW] (no copy)."""
    return phys[..., :co].permute(3, 2, 0, 1)


def is_hwio(weight):
    """True when a logical OIHW tensor is stored [KH][KW][Ci][Co_ld] with Co_ld % 4 == 0."""
    co, ci, kh, kw = weight.shape
    co_ld = weight.stride(1)
    return (weight.stride(0) == 1 and co_ld % 4 == 0 and co_ld >= co
            and weight.stride(3) == ci * co_ld and weight.stride(2) == kw * ci * co_ld
            and weight.data_ptr() % 16 == 0)


def relayout_conv_params(conv):
    """(Re)create HWIO / padded storage for a conv's parameters on their current device, keeping
    the Parameter objects (and therefore optimizer / state_dict identity)."""
    w = conv.weight
    co, ci, kh, kw = w.shape
    co_ld = round_up(co, 4)
    if not is_hwio(w):
        phys = torch.zeros((kh, kw, ci, co_ld), dtype=w.dtype, device=w.device)
        view = hwio_logical_view(phys, co)
        view.copy_(w.data)
        w.data = view
        w.grad = None
    w.__dict__.pop("_gs_plans", None)   # cached launch plans (hip/ops.py) describe the old storage
    conv.__dict__.pop("_layout_ptr", None)
    w._gs_phys_shape = (kh, kw, ci, co_ld)
    w._gs_grad_factory = lambda w=w, co=co, shape=(kh, kw, ci, co_ld): hwio_logical_view(
        torch.zeros(shape, dtype=w.dtype, device=w.device), co)
    b = conv.bias
    if b is not None:
        ok = b.data_ptr() % 16 == 0 and (co == co_ld or _storage_room(b) >= co_ld)
        if not ok:
            phys = torch.zeros(co_ld, dtype=b.dtype, device=b.device)
            phys[:co].copy_(b.data)
            b.data = phys[:co]
            b.grad = None
        b._gs_phys_shape = (co_ld,)
        b._gs_grad_factory = lambda b=b, co=co, co_ld=co_ld: torch.zeros(
            co_ld, dtype=b.dtype, device=b.device)[:co]


def _storage_room(t):
    """number of elements available from t's first element to the end of its storage"""
    return t.untyped_storage().nbytes() // t.element_size() - t.storage_offset()


# ------------------------------------------------------------------------------------------
# DynConv2d
# ------------------------------------------------------------------------------------------
@CONV_LAYERS.register_module(["DynConv2d", "Conv2d", "Conv"])
class DynamicConv2d(nn.Module, DynamicMixin):
    """Conv2d over the leading slice of a max-size weight: ``F.conv2d(x, W[:width, :x.size(1)])``.

    ``width_state`` is the active number of output channels; the active number of input channels
    is taken from the input tensor."""
    search_space = {"width"}

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, padding_mode="zeros"):
        super().__init__()
        if groups != 1:
            raise NotImplementedError("DynConv2d: groups != 1 is not used on the supernet hot path")
        if padding_mode != "zeros":
            raise NotImplementedError("DynConv2d: only zero padding is supported")
        kh, kw = _pair(kernel_size)
        sh, sw = _pair(stride)
        ph, pw = _pair(padding)
        dh, dw = _pair(dilation)
        if sh != sw or ph != pw or dh != dw:
            raise NotImplementedError("DynConv2d: anisotropic stride/padding/dilation unsupported")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.dilation = (kh, kw), sh, ph, dh
        self.groups = 1
        co_ld = round_up(out_channels, 4)
        phys = torch.zeros(kh, kw, in_channels, co_ld)
        self.weight = nn.Parameter(hwio_logical_view(phys, out_channels))
        if bias:
            self.bias = nn.Parameter(torch.zeros(co_ld)[:out_channels])
        else:
            self.register_parameter("bias", None)
        self.init_state(width=out_channels)
        self.reset_parameters()
        relayout_conv_params(self)

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.in_channels * self.kernel_size[0] * self.kernel_size[1]
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            nn.init.uniform_(self.bias, -bound, bound)

    def extra_repr(self):
        return "%d, %d, kernel_size=%s, stride=%d, padding=%d, dilation=%d, bias=%s, width=%s" % (
            self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding,
            self.dilation, self.bias is not None, self.width_state)

    # ---- dynamic interface ----
    def manipulate_width(self, width):
        if not 0 < width <= self.out_channels:
            raise ValueError("width %s out of range (1..%d)" % (width, self.out_channels))
        self.width_state = width

    def _check_layout(self):
        if not is_hwio(self.weight) or (self.bias is not None and self.bias.data_ptr() % 16):
            # e.g. after .to(device) of a padded view or load of a foreign tensor
            relayout_conv_params(self)

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        relayout_conv_params(self)

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        relayout_conv_params(self)
        return out

    # ---- execution ----
    def forward_act(self, tape, x, out=None, tag=None):
        self._check_layout()
        if getattr(self, "_deploying", False):
            self._deploy_slice(x.t.shape[1] if x.nchw_image else x.C)
        return ops.conv2d(tape, x, self.weight, self.bias, self.width_state, self.stride,
                          self.padding, self.dilation, out=out, tag=tag)

    def forward(self, x):
        needs = any(p.requires_grad for p in self.parameters())
        return tape_function(lambda tape, acts: [self.forward_act(tape, acts[0])], [x], needs)[0]

    def _deploy_slice(self, ci):
        """Physically prune to [:width, :ci] (tools/extract_subnet.py semantics)."""
        co = self.width_state
        if self.weight.shape[0] == co and self.weight.shape[1] == ci:
            return
        w = self.weight.data[:co, :ci].clone()
        b = self.bias.data[:co].clone() if self.bias is not None else None
        self.in_channels, self.out_channels = ci, co
        kh, kw = self.kernel_size
        phys = torch.zeros((kh, kw, ci, round_up(co, 4)), dtype=w.dtype, device=w.device)
        view = hwio_logical_view(phys, co)
        view.copy_(w)
        self.weight = nn.Parameter(view, requires_grad=self.weight.requires_grad)
        if b is not None:
            pb = torch.zeros(round_up(co, 4), dtype=b.dtype, device=b.device)
            pb[:co].copy_(b)
            self.bias = nn.Parameter(pb[:co], requires_grad=self.bias.requires_grad)
        relayout_conv_params(self)


def build_conv_layer(cfg, *args, **kwargs):
    """mmcv.cnn.build_conv_layer: cfg None -> a plain (fixed-width) conv, which here is the same
    kernel-backed class."""
    cfg = dict(type="Conv2d") if cfg is None else dict(cfg)
    layer_type = cfg.pop("type")
    cls = CONV_LAYERS.get(layer_type)
    if cls is None:
        raise KeyError("Unrecognized conv type %s" % layer_type)
    return cls(*args, **kwargs, **cfg)


# ------------------------------------------------------------------------------------------
# DynBN / DynSyncBN / BN / SyncBN
# ------------------------------------------------------------------------------------------
class DynamicBatchNorm2d(_BatchNorm, DynamicMixin):
    """BatchNorm2d over the leading ``x.size(1)`` channels of max-size parameters / buffers.

    ``sync`` selects the statistics scope: None = this rank only; 'world' = all ranks
    (torch.nn.SyncBatchNorm semantics); int g = groups of g consecutive ranks (DynSyncBN
    ``group_size``; g == 1 is local, the config of record)."""
    search_space = set()
    _abbr_ = "bn"

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True,
                 track_running_stats=True, sync=None):
        super().__init__(num_features, eps, momentum, affine, track_running_stats)
        self.sync = sync
        self._group = None
        self._group_ready = False

    def _check_input_dim(self, input):
        if input.dim() != 4:
            raise ValueError("expected 4D input (got %dD input)" % input.dim())

    def _process_group(self):
        import torch.distributed as dist
        if self.sync is None or not (dist.is_available() and dist.is_initialized()):
            return None
        world = dist.get_world_size()
        if world == 1:
            return None
        if self.sync == "world":
            return dist.group.WORLD
        g = int(self.sync)
        if g <= 1:
            return None
        if g >= world:
            return dist.group.WORLD
        if not self._group_ready:
            rank = dist.get_rank()
            mine = None
            for start in range(0, world, g):  # every rank creates every group (collective)
                grp = dist.new_group(list(range(start, min(start + g, world))))
                if start <= rank < start + g:
                    mine = grp
            self._group, self._group_ready = mine, True
        return self._group

    def bn_params(self, c):
        if c > self.num_features:
            raise ValueError("input has %d channels, norm supports at most %d" % (c, self.num_features))
        training = self.training
        # (one BNParams per mode, reused: building it walks nn.Module.__getattr__ five times.  The
        # cache is dropped whenever parameters / buffers may have been replaced.)
        cache = self.__dict__.get("_bnp_cache")
        if cache is None:
            cache = self.__dict__["_bnp_cache"] = {}
        bnp = cache.get(training)
        if bnp is None:
            bnp = cache[training] = ops.BNParams(
                self.weight, self.bias, self.running_mean, self.running_var, self.eps, self.momentum,
                training, num_batches_tracked=self._count_batch
                if self.num_batches_tracked is not None else None)
        bnp.process_group = self._process_group() if (training and self.sync is not None) else None
        return bnp

    def _apply(self, fn, recurse=True):
        self.__dict__.pop("_bnp_cache", None)
        return super()._apply(fn, recurse)

    def _load_from_state_dict(self, *args, **kwargs):
        self.__dict__.pop("_bnp_cache", None)
        super()._load_from_state_dict(*args, **kwargs)

    def _count_batch(self):
        """num_batches_tracked += 1 without a device op per BN per step: counted on the host and
        folded into the buffer whenever it is read through state_dict()."""
        d = self.__dict__   # (nn.Module.__setattr__ costs 3 us; this runs once per BN layer and step)
        d["_nbt_pending"] = d.get("_nbt_pending", 0) + 1
        log = _runtime.CAPTURE_LOG
        if log is not None:   # a captured step graph repeats this count at every replay
            log.append(self)

    def flush_counters(self):
        pending = getattr(self, "_nbt_pending", 0)
        if pending and self.num_batches_tracked is not None:
            self.num_batches_tracked += pending
        self._nbt_pending = 0

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self.flush_counters()
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def forward_act(self, tape, x, relu=False, residual=None, out=None):
        if getattr(self, "_deploying", False):
            self._deploy_slice(x.C)
        return ops.batchnorm(tape, x, self.bn_params(x.C), relu=relu, residual=residual, out=out)

    def forward(self, x):
        self._check_input_dim(x)
        needs = any(p.requires_grad for p in self.parameters())
        return tape_function(lambda tape, acts: [self.forward_act(tape, acts[0])], [x], needs)[0]

    def _deploy_slice(self, c):
        if self.num_features == c:
            return
        self.num_features = c
        self.__dict__.pop("_bnp_cache", None)
        if self.affine:
            self.weight = nn.Parameter(self.weight.data[:c].clone(), self.weight.requires_grad)
            self.bias = nn.Parameter(self.bias.data[:c].clone(), self.bias.requires_grad)
        if self.track_running_stats:
            self.running_mean = self.running_mean[:c].clone()
            self.running_var = self.running_var[:c].clone()


def _register_norm(names, sync_default):
    def factory(num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True,
                group_size=None, **unused):
        sync = sync_default
        if sync_default == "group":
            # DECIDE: omitted group_size => all ranks
            sync = "world" if group_size in (None, 0) else int(group_size)
            if sync == 1:
                sync = None
        return DynamicBatchNorm2d(num_features, eps, momentum, affine, track_running_stats, sync)
    for n in names:
        NORM_LAYERS.register_module(n, module=factory)


_register_norm(["BN", "BN2d", "DynBN"], None)
_register_norm(["SyncBN"], "world")
_register_norm(["DynSyncBN"], "group")


def build_norm_layer(cfg, num_features, postfix=""):
    """mmcv.cnn.build_norm_layer: returns (name, layer); BN-family abbreviation is 'bn'."""
    if not isinstance(cfg, dict) or "type" not in cfg:
        raise KeyError('the cfg dict must contain the key "type"')
    cfg_ = dict(cfg)
    layer_type = cfg_.pop("type")
    factory = NORM_LAYERS.get(layer_type)
    if factory is None:
        raise KeyError("Unrecognized norm type %s" % layer_type)
    requires_grad = cfg_.pop("requires_grad", True)
    cfg_.setdefault("eps", 1e-5)
    layer = factory(num_features, **cfg_)
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return "bn%s" % postfix, layer


def build_activation_layer(cfg):
    return build_from_cfg(cfg, ACTIVATION_LAYERS)


_NO_FUSED_CALLS = os.environ.get("GS_NO_FUSED_CALLS") is not None


def fused_call_ok(conv, norm):
    """conv -> norm can go through the one-call-per-direction entry (ops.conv_bn): a bias-free conv of a
    width that is a multiple of 4 followed by a rank-local BatchNorm, not while extracting a subnet."""
    return (conv._parameters["bias"] is None and conv.width_state % 4 == 0
            and not conv.__dict__.get("_deploying", False)
            and not norm.__dict__.get("_deploying", False) and not _NO_FUSED_CALLS
            and not (norm.training and norm.sync is not None and norm._process_group() is not None))


def conv_bn_act(tape, conv, norm, x, relu=False, residual=None, out=None, tag=None, defer=False,
                owns_input_grad=False, defer_residual=False):
    """conv -> norm (+ residual) (+ ReLU).  ``defer``: leave the BN + ReLU to the consumer's operand
    loaders where the fused path allows it (ops.conv_bn).  Rank-local BatchNorm after a bias-free conv goes through
    the one-call-per-direction library entry (ops.conv_bn); SyncBN with a process group, a conv
    bias, widths that are not multiples of 4 and subnet extraction take the module-by-module path.
    Both paths launch the same kernels."""
    c = conv.width_state
    cd = conv.__dict__
    weight = conv._parameters["weight"]
    fused = (conv._parameters["bias"] is None and c % 4 == 0 and not cd.get("_deploying", False)
             and not norm.__dict__.get("_deploying", False) and not _NO_FUSED_CALLS)
    if fused:
        bnp = norm.bn_params(c)
        if bnp.process_group is None:
            # the HWIO layout check touches five strides: redo it only when the storage moved
            if cd.get("_layout_ptr") != weight.data_ptr():
                conv._check_layout()
                weight = conv._parameters["weight"]
                cd["_layout_ptr"] = weight.data_ptr()
            return ops.conv_bn(tape, x, weight, c, bnp, conv.stride, conv.padding,
                               conv.dilation, relu=relu, residual=residual, out=out, tag=tag,
                               defer=defer, owns_input_grad=owns_input_grad,
                               defer_residual=defer_residual)
    y = conv.forward_act(tape, x, tag=tag)
    return norm.forward_act(tape, y, relu=relu, residual=residual, out=out)


# ------------------------------------------------------------------------------------------
# DynamicConvModule  (mmcv ConvModule with dynamic conv / norm)
# ------------------------------------------------------------------------------------------
class DynamicConvModule(nn.Module, DynamicMixin):
    """conv(bias = norm is None) -> norm -> ReLU, fused as conv kernel + BN(+ReLU) kernels."""
    search_space = {"width"}

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias="auto", conv_cfg=None, norm_cfg=None, act_cfg=dict(type="ReLU"),
                 inplace=True, order=("conv", "norm", "act")):
        super().__init__()
        if tuple(order) != ("conv", "norm", "act"):
            raise NotImplementedError("only the conv-norm-act order is used by the heads")
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if self.with_activation and act_cfg.get("type") != "ReLU":
            raise NotImplementedError("only ReLU activations are fused")
        if bias == "auto":
            bias = not self.with_norm
        self.with_bias = bias
        if self.with_norm and self.with_bias:
            warnings.warn("ConvModule has norm and bias at the same time")
        self.conv = build_conv_layer(conv_cfg, in_channels, out_channels, kernel_size,
                                     stride=stride, padding=padding, dilation=dilation,
                                     groups=groups, bias=bias)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.inplace = inplace
        if self.with_norm:
            self.norm_name, norm = build_norm_layer(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        else:
            self.norm_name = None
        if self.with_activation:
            self.activate = nn.ReLU(inplace=inplace)
        self.init_weights()

    @property
    def norm(self):
        return getattr(self, self.norm_name) if self.norm_name else None

    def init_weights(self):
        # mmcv ConvModule default: kaiming_init(conv, a=0, nonlinearity='relu') (fan_out, normal)
        kaiming_init(self.conv, a=0, nonlinearity="relu")
        if self.with_norm:
            constant_init(self.norm, 1, bias=0)

    def manipulate_width(self, width):
        self.conv.manipulate_width(width)

    def forward_act(self, tape, x, activate=True, norm=True, out=None):
        relu = bool(activate) and self.with_activation
        if norm and self.with_norm:
            return conv_bn_act(tape, self.conv, self.norm, x, relu=relu, out=out)
        if relu:
            raise NotImplementedError("ReLU without norm is not used on the hot path")
        return self.conv.forward_act(tape, x, out=out)

    def forward(self, x, activate=True, norm=True):
        needs = any(p.requires_grad for p in self.parameters())
        return tape_function(
            lambda tape, acts: [self.forward_act(tape, acts[0], activate, norm)], [x], needs)[0]


# ------------------------------------------------------------------------------------------
# DynamicBottleneck
# ------------------------------------------------------------------------------------------
class DynamicBottleneck(nn.Module, DynamicMixin):
    """ResNet bottleneck (expansion 4, stride on the 3x3 for style='pytorch') with dynamic width.

    out = relu( bn3(conv3( relu(bn2(conv2( relu(bn1(conv1(x))) ))) )) + shortcut(x) )"""
    expansion = 4
    search_space = {"width"}

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, style="pytorch",
                 with_cp=False, conv_cfg=None, norm_cfg=dict(type="BN"), dcn=None, plugins=None):
        super().__init__()
        if style not in ("pytorch", "caffe"):
            raise ValueError("style must be pytorch or caffe")
        if dcn is not None or plugins:
            raise NotImplementedError("dcn / plugins are not used by the in-tree configs")
        self.inplanes, self.planes = inplanes, planes
        self.stride, self.dilation, self.style, self.with_cp = stride, dilation, style, with_cp
        self.conv_cfg, self.norm_cfg = conv_cfg, norm_cfg
        if style == "pytorch":
            self.conv1_stride, self.conv2_stride = 1, stride
        else:
            self.conv1_stride, self.conv2_stride = stride, 1
        self.norm1_name, norm1 = build_norm_layer(norm_cfg, planes, postfix=1)
        self.norm2_name, norm2 = build_norm_layer(norm_cfg, planes, postfix=2)
        self.norm3_name, norm3 = build_norm_layer(norm_cfg, planes * self.expansion, postfix=3)
        self.conv1 = build_conv_layer(conv_cfg, inplanes, planes, kernel_size=1,
                                      stride=self.conv1_stride, bias=False)
        self.add_module(self.norm1_name, norm1)
        self.conv2 = build_conv_layer(conv_cfg, planes, planes, kernel_size=3,
                                      stride=self.conv2_stride, padding=dilation,
                                      dilation=dilation, bias=False)
        self.add_module(self.norm2_name, norm2)
        self.conv3 = build_conv_layer(conv_cfg, planes, planes * self.expansion, kernel_size=1,
                                      bias=False)
        self.add_module(self.norm3_name, norm3)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.init_state(width=planes)

    @property
    def norm1(self):
        return getattr(self, self.norm1_name)

    @property
    def norm2(self):
        return getattr(self, self.norm2_name)

    @property
    def norm3(self):
        return getattr(self, self.norm3_name)

    def manipulate_width(self, width):
        self.width_state = width
        self.conv1.manipulate_width(width)
        self.conv2.manipulate_width(width)
        self.conv3.manipulate_width(width * self.expansion)
        if self.downsample is not None:
            for m in self.downsample:
                if isinstance(m, DynamicConv2d):
                    m.manipulate_width(width * self.expansion)

    def _shortcut_gflop(self, x):
        """2 * rows * Ci * Co of the projection shortcut's 1x1 conv for input activation x, in GFLOP."""
        conv = next(m for m in self.downsample if isinstance(m, DynamicConv2d))
        s = max(self.stride, 1)
        rows = x.N * ((x.H + s - 1) // s) * ((x.W + s - 1) // s)
        return 2e-9 * rows * x.C * conv.width_state

    def _shortcut_act(self, tape, x, defer_residual=False):
        """The projection shortcut: [AvgPool2d,] 1x1 conv, norm (dynamic_res_layer.py:70-94).
        ``defer_residual``: leave the norm to the residual add of norm3's apply pass."""
        identity = x
        members = list(self.downsample)
        if isinstance(members[0], nn.AvgPool2d):
            # avg_down (dynamic_res_layer.py:75-82): AvgPool2d(stride, ceil_mode=True,
            # count_include_pad=False) in front of a stride-1 1x1 conv
            pool = members.pop(0)
            k = pool.kernel_size if isinstance(pool.kernel_size, int) else pool.kernel_size[0]
            st = pool.stride if isinstance(pool.stride, int) else pool.stride[0]
            if k != st or not pool.ceil_mode or pool.count_include_pad or pool.padding not in (0, (0, 0)):
                raise NotImplementedError("downsample AvgPool2d other than (k = stride, "
                                          "ceil_mode=True, count_include_pad=False)")
            identity = ops.avgpool_ceil(tape, identity, st) if st > 1 else identity
        if (len(members) == 2 and isinstance(members[0], DynamicConv2d)
                and isinstance(members[1], DynamicBatchNorm2d)):
            return conv_bn_act(tape, members[0], members[1], identity, relu=False,
                               defer_residual=defer_residual)
        raise NotImplementedError("downsample branch %s: expected [AvgPool2d,] conv, norm"
                                  % [type(m).__name__ for m in self.downsample])

    def forward_act(self, tape, x):
        # bn1 -> conv2 / bn2 -> conv3 (ops.DEFER_EDGES): the normalised activation is never stored;
        # the consumer (forward and weight gradient) evaluates relu(bn(.)) in its operand loaders
        # owns_input_grad: the conv's data gradient is the last contribution to its input's gradient,
        # so its epilogue may also do the BatchNorm-backward reduction of the layer that produced
        # that input (conv2 / conv3 are the only consumers of theirs; conv1's input also feeds the
        # identity branch or the projection shortcut, whose gradient is in place before conv1's
        # backward runs, see below)
        hot = self.__dict__.get("_hot")   # (six nn.Module.__getattr__ walks per block otherwise)
        if hot is None:
            hot = self.__dict__["_hot"] = (self.conv1, self.norm1, self.conv2, self.norm2,
                                           self.conv3, self.norm3)
        conv1, norm1, conv2, norm2, conv3, norm3 = hot
        edges = ops.DEFER_EDGES
        identity, br = x, None
        if self.downsample is not None:
            # The projection shortcut is independent of conv1 -> conv2: it runs on the branch stream
            # beside them, forward and backward (ops.Branch).  In backward its data gradient is the
            # FIRST writer of x.g (a strided one clears the pixel classes it has no tap for) and
            # conv1's accumulates onto it after the join, so conv1 owns the input gradient here too.
            # The shortcut's BatchNorm is applied inside norm3's apply pass (relu(bn3(y3) + bn_s(y_s))):
            # its normalised output is never stored.
            # (only where the shortcut is a small launch: at OS8 the stage-3/4 shortcuts are 17-69 GF
            # convolutions that fill the chip alone and only take it away from the block's K3 launch:
            # K3 0.77 -> 0.64 of peak on the v1c supernet with no gain in images/s)
            small = self._shortcut_gflop(x) <= ops.BRANCH_SHORTCUT_MAX_GFLOP
            br = ops.Branch(tape, x.t.device, ops.BRANCH_SHORTCUT and small)
            with br:
                identity = self._shortcut_act(tape, x, ops.DEFER_SHORTCUT_BN
                                              and fused_call_ok(conv3, norm3))
        out = conv_bn_act(tape, conv1, norm1, x, relu=True,
                          defer="conv2" in edges, owns_input_grad=True)
        if br is not None:
            br.record_backward_join(tape)
        out = conv_bn_act(tape, conv2, norm2, out, relu=True, tag="k3",
                          defer="conv3" in edges, owns_input_grad=True)
        if br is not None:
            br.record_backward_body(tape)
            br.join()
        return conv_bn_act(tape, conv3, norm3, out, relu=True, residual=identity,
                           owns_input_grad=True)

    def forward(self, x):
        needs = any(p.requires_grad for p in self.parameters())
        return tape_function(lambda tape, acts: [self.forward_act(tape, acts[0])], [x], needs)[0]


# ------------------------------------------------------------------------------------------
# weight initialisers (mmcv.cnn semantics)
# ------------------------------------------------------------------------------------------
def kaiming_init(module, a=0, mode="fan_out", nonlinearity="relu", bias=0, distribution="normal"):
    if distribution == "uniform":
        nn.init.kaiming_uniform_(module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
    else:
        nn.init.kaiming_normal_(module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
    if getattr(module, "bias", None) is not None:
        nn.init.constant_(module.bias, bias)


def constant_init(module, val, bias=0):
    if getattr(module, "weight", None) is not None:
        nn.init.constant_(module.weight, val)
    if getattr(module, "bias", None) is not None:
        nn.init.constant_(module.bias, bias)


def normal_init(module, mean=0, std=1, bias=0):
    if getattr(module, "weight", None) is not None:
        nn.init.normal_(module.weight, mean, std)
    if getattr(module, "bias", None) is not None:
        nn.init.constant_(module.bias, bias)
