"""ctypes binding of ``libgaiaseg_hip.so`` (C-ABI declared in ``include/gaiaseg_hip.h``).

The library is the product: there is no CPU or eager-PyTorch fallback anywhere in this package.
``load()`` raises ``HipLibraryError`` when the shared object is missing or does not export every
symbol of the header, and every wrapper in :mod:`gaia_seg_amd.hip.functional` raises on a non-zero
return code.
"""
import ctypes
import os
import subprocess
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int32, c_int64, c_size_t,
                    c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(_HERE)
REPO_ROOT = os.path.dirname(PKG_ROOT)
LIB_PATH = os.path.join(PKG_ROOT, "lib", "libgaiaseg_hip.so")
CSRC_DIR = os.path.join(PKG_ROOT, "csrc")
ABI_VERSION = 9


class HipLibraryError(RuntimeError):
    pass


class ConvDesc(Structure):
    """Mirror of ``gs_conv_desc``."""
    _fields_ = [("N", c_int32), ("H", c_int32), ("W", c_int32), ("Ci", c_int32), ("Co", c_int32),
                ("Ci_max", c_int32), ("Co_ld", c_int32), ("KH", c_int32), ("KW", c_int32),
                ("stride", c_int32), ("pad", c_int32), ("dil", c_int32), ("Ho", c_int32),
                ("Wo", c_int32), ("x_sn", c_int64), ("x_sh", c_int64), ("x_sw", c_int64),
                ("x_sc", c_int64), ("ldy", c_int32), ("ld_add", c_int32),
                ("role", c_int32), ("reserved", c_int32), ("in_affine", c_void_p)]


class BnArgs(Structure):
    """Mirror of ``gs_bn_args``."""
    _fields_ = [("gamma", c_void_p), ("beta", c_void_p), ("running_mean", c_void_p),
                ("running_var", c_void_p), ("eps", c_float), ("momentum", c_float),
                ("use_batch_stats", c_int32), ("update_running", c_int32), ("relu", c_int32),
                ("reserved", c_int32), ("relu_mask", c_void_p), ("residual_coeffs", c_void_p)]


class BnBwdFuse(Structure):
    """Mirror of ``gs_bn_bwd_fuse``."""
    _fields_ = [("y", c_void_p), ("act", c_void_p), ("coeffs", c_void_p), ("sums", c_void_p),
                ("fused", POINTER(c_int32)), ("ldy", c_int32), ("ldact", c_int32), ("mode", c_int32),
                ("reserved", c_int32), ("mask", c_void_p), ("ldmask", c_int32), ("reserved2", c_int32)]


class DebugLaunch(Structure):
    """Mirror of ``gs_debug_launch``."""
    _fields_ = [(k, c_int32) for k in ("op", "kloop", "bm", "bn", "splits", "ksteps_per_split",
                                       "in_affine", "bn_bwd_mode")]


OP_FORWARD, OP_DGRAD, OP_WGRAD = 0, 1, 2
KLOOP_GENERIC, KLOOP_FP32, KLOOP_FP32_PAIRS, KLOOP_BF16X3, KLOOP_STREAM = 0, 1, 2, 3, 4
KLOOP_COUNT = 5


class CeDesc(Structure):
    """Mirror of ``gs_ce_desc``."""
    _fields_ = [("N", c_int32), ("h", c_int32), ("w", c_int32), ("Cls", c_int32), ("H", c_int32),
                ("W", c_int32), ("l_sn", c_int64), ("l_sh", c_int64), ("l_sw", c_int64),
                ("l_sc", c_int64), ("ignore_index", c_int32), ("align_corners", c_int32)]


class SlideDesc(Structure):
    """Mirror of ``gs_slide_desc``."""
    _fields_ = [(k, c_int32) for k in ("N", "C", "ld", "hl", "wl", "hc", "wc", "H", "W", "Ho", "Wo",
                                       "ny", "nx", "align_corners", "flip", "reserved")]


class AugmentDesc(Structure):
    """Mirror of ``gs_augment_desc``."""
    _fields_ = ([(k, c_int32) for k in ("src_h", "src_w", "src_is_rgb", "res_h", "res_w", "crop_y",
                                        "crop_x", "crop_h", "crop_w", "out_h", "out_w", "flip",
                                        "pm_enable", "pm_brightness", "pm_contrast",
                                        "pm_contrast_first", "pm_saturation", "pm_hue")] +
                [("pm_delta", c_float), ("pm_alpha", c_float), ("pm_sat_alpha", c_float),
                 ("pm_hue_delta", c_int32), ("to_rgb", c_int32), ("mean", c_float * 3),
                 ("std", c_float * 3), ("pad_val", c_float), ("seg_pad_val", c_int32)])


_P = c_void_p  # device pointers and the stream travel as plain addresses
_i32, _i64, _f32, _f64, _sz = c_int32, c_int64, c_float, c_double, c_size_t
_CD, _CE, _BN = POINTER(ConvDesc), POINTER(CeDesc), POINTER(BnArgs)

# name -> (restype, argtypes): one entry per declaration in include/gaiaseg_hip.h
PROTOTYPES = {
    "gs_abi_version": (_i32, []),
    "gs_error_string": (c_char_p, [_i32]),
    "gs_target_arch": (c_char_p, []),
    "gs_conv2d_workspace_bytes": (_sz, [_CD]),
    "gs_conv2d_in_affine_supported": (_i32, [_CD]),
    "gs_conv2d_forward": (_i32, [_CD, _P, _P, _P, _P, _P, _P, _sz, _P]),
    "gs_conv2d_dgrad": (_i32, [_CD, _P, _P, _P, _i32, _P, _sz, _P]),
    "gs_conv2d_wgrad": (_i32, [_CD, _P, _P, _P, _P, _sz, _P]),
    "gs_colsum_workspace_bytes": (_sz, [_i64, _i32]),
    "gs_colsum": (_i32, [_P, _i64, _i32, _i32, _P, _P, _sz, _P]),
    "gs_bn_stats_workspace_bytes": (_sz, [_i64, _i32]),
    "gs_bn_stats": (_i32, [_P, _i64, _i32, _i32, _P, _P, _sz, _P]),
    "gs_bn_finalize": (_i32, [_P, _f64, _i32, _P, _P, _f32, _f32, _P, _P, _P, _P]),
    "gs_bn_sync_local": (_i32, [_P, _f64, _i32, _P, _P]),
    "gs_bn_sync_merge": (_i32, [_P, _i32, _i32, _P, _P]),
    "gs_bn_stats_finalize": (_i32, [_P, _i64, _i32, _i32, _P, _P, _f32, _f32, _P, _P, _P, _P, _sz,
                                    _P]),
    "gs_bn_eval_coeffs": (_i32, [_P, _P, _i32, _P, _P, _f32, _P, _P]),
    "gs_bn_apply": (_i32, [_P, _i64, _i32, _i32, _P, _P, _i32, _i32, _P, _i32, _P]),
    "gs_bn_apply_mask": (_i32, [_P, _i64, _i32, _i32, _P, _P, _i32, _P, _i32, _P, _P]),
    "gs_bn_bwd_workspace_bytes": (_sz, [_i64, _i32]),
    "gs_bn_bwd_reduce": (_i32, [_P, _i32, _P, _i32, _P, _i32, _i64, _i32, _P, _i32, _P, _i32, _P,
                                _P, _sz, _P]),
    "gs_bn_bwd_apply": (_i32, [_P, _i32, _P, _i32, _P, _i32, _i64, _i32, _P, _P, _f64, _i32, _i32,
                               _P, _i32, _P, _P, _P]),
    "gs_avgpool_ceil_forward": (_i32, [_P, _i32, _i32, _i32, _i32, _i32, _i32, _P, _i32, _P]),
    "gs_avgpool_ceil_backward": (_i32, [_P, _i32, _i32, _i32, _i32, _i32, _i32, _P, _i32, _i32, _P]),
    "gs_maxpool_forward": (_i32, [_P, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                  _P, _i32, _P, _P]),
    "gs_maxpool_backward": (_i32, [_P, _i32, _P, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                   _i32, _P, _i32, _i32, _P]),
    "gs_adaptive_avgpool_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32, POINTER(_i32), _i32]),
    "gs_adaptive_avgpool_forward": (_i32, [_P, _i32, _i32, _i32, _i32, _i32, POINTER(_i32), _i32,
                                           _P, _P, _sz, _P]),
    "gs_adaptive_avgpool_backward": (_i32, [_P, _i32, _i32, _i32, _i32, POINTER(_i32), _i32, _P,
                                            _i32, _i32, _P]),
    "gs_bilinear_forward": (_i32, [_P, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _P, _i32,
                                   _i32, _P]),
    "gs_bilinear_backward_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32, _i32, _i32]),
    "gs_bilinear_backward": (_i32, [_P, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _P, _i32,
                                    _i32, _P, _sz, _P]),
    "gs_copy2d": (_i32, [_P, _i32, _P, _i32, _i64, _i32, _f32, _i32, _P]),
    "gs_scale_nc": (_i32, [_P, _i32, _P, _i32, _i64, _i32, _P, _i32, _P]),
    "gs_ce_workspace_bytes": (_sz, [_CE]),
    "gs_ce_forward": (_i32, [_CE, _P, _P, _P, _P, _P, _P, _P, _sz, _P]),
    "gs_ce_forward_scaled": (_i32, [_CE, _P, _P, _P, _P, _P, _f32, _f32, _P, _P, _sz, _P]),
    "gs_ce_backward": (_i32, [_CE, _P, _P, _P, _P, _P, _f32, _P, _i32, _P]),
    "gs_ce_backward_workspace_bytes": (_sz, [_CE, _i32]),
    "gs_ce_backward_ws": (_i32, [_CE, _P, _P, _P, _P, _P, _f32, _P, _i32, _P, _sz, _P]),
    "gs_ce_label_prob": (_i32, [_CE, _P, _P, _P, _P]),
    "gs_resize_argmax": (_i32, [_CE, _P, _P, _P, _P]),
    "gs_slide_fuse": (_i32, [POINTER(SlideDesc), POINTER(_i32), POINTER(_i32), _P, _P, _P, _P, _P]),
    "gs_debug_set_slide_strip": (_i32, [_i32]),
    "gs_seg_augment": (_i32, [POINTER(AugmentDesc), _P, _P, _P, _P, _P]),
    "gs_ohem_workspace_bytes": (_sz, []),
    "gs_ohem_weights": (_i32, [_P, _i64, _i64, _f32, _i32, _P, _P, _sz, _P]),
    "gs_confusion_matrix": (_i32, [_P, _P, _i64, _i32, _i32, _P, _P]),
    "gs_sgd_step": (_i32, [_P, _P, _P, _i64, _f32, _f32, _f32, _f32, _i32, _P]),
    "gs_sgd_step_hyper": (_i32, [_P, _P, _P, _i64, _P, _i32, _P]),
    "gs_sgd_set_hyper": (_i32, [_P, _f32, _f32, _f32, _f32, _P]),
    "gs_debug_force_plan": (_i32, [_i32, _i32, _i32]),
    "gs_debug_query_plan": (_i32, [_i32, _i32, _i32, _i32, POINTER(_i32), POINTER(_i32), POINTER(_i32),
                                   POINTER(_i32)]),
    "gs_debug_last_conv_launch": (_i32, [POINTER(DebugLaunch)]),
    "gs_debug_conv_launch_counts": (_i32, [POINTER(_i64), _i32]),
    "gs_debug_k3_flops": (_i32, [POINTER(_f64), _i32]),
    "gs_debug_set_x3_fwd": (_i32, [_i32]),
    "gs_debug_set_splitk_inkernel": (_i32, [_i32]),
    "gs_debug_splitk_combined": (_i64, [_i32]),
    "gs_debug_set_col_finalize": (_i32, [_i32]),
    "gs_debug_col_finalized": (_i64, [_i32]),
    "gs_debug_num_cu": (_i32, []),
    "gs_debug_set_stream_mode": (_i32, [_i32]),
    "gs_debug_conv_launch_flops": (_i32, [POINTER(_f64), _i32]),
    "gs_debug_query_conv_launch": (_i32, [_CD, _i32, POINTER(DebugLaunch)]),
    "gs_stream_fork": (_i32, [_P, _P]),
    "gs_conv_bn_workspace_bytes": (_sz, [_CD]),
    "gs_conv_bn_forward": (_i32, [_CD, _P, _P, _BN, _P, _i32, _P, _P, _P, _i32, _P, _sz, _P]),
    "gs_conv_bn_backward": (_i32, [_CD, _P, _P, _P, _P, _i32, _P, _BN, _P, _i32, _i32, _i32, _P, _P,
                                   _P, _P, _P, _P, _i32, _P, _sz, _P, _sz, _P, _P, POINTER(BnBwdFuse),
                                   _i32]),
    "gs_k3_timer_enable": (_i32, [_i32]),
    "gs_k3_timer_read": (_i32, [POINTER(_i64), POINTER(_f64), POINTER(_f64)]),
}

_lib = None


def build(verbose=False, jobs=None):
    """Compile every HIP source for gfx950 into ``gaia_seg_amd/lib/libgaiaseg_hip.so``.

    hipcc cross-compiles without a GPU, so this runs in the build container as well.
    """
    jobs = jobs or min(8, os.cpu_count() or 1)
    cmd = ["make", "-C", CSRC_DIR, "-j%d" % jobs]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise HipLibraryError("building libgaiaseg_hip.so failed (exit %d)" % res.returncode)
    global _lib
    _lib = None
    return LIB_PATH


def load():
    """Load the shared object, bind every prototype, check the ABI version. Raises on any gap."""
    global _lib, LIB_PATH
    if _lib is not None:
        return _lib
    if os.environ.get("GS_HIP_LIB"):   # A/B of two builds of the same library (tuning only)
        LIB_PATH = os.environ["GS_HIP_LIB"]
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            "%s not found: the HIP extension is required (no CPU fallback). Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C gaia_seg_amd/csrc`."
            % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # missing ROCm runtime etc.
        raise HipLibraryError("cannot load %s: %s" % (LIB_PATH, e)) from e
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError("%s does not export %s" % (LIB_PATH, name)) from e
        fn.restype = res
        fn.argtypes = args
    if lib.gs_abi_version() != ABI_VERSION:
        raise HipLibraryError("ABI mismatch: library %d, binding %d"
                              % (lib.gs_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


CALL_PROFILE = None


def enable_call_profile():
    """Diagnostics (GS_HOST_PROF): time every call into the library on the host.  Returns the dict
    {name: [calls, seconds]} that the wrappers fill; the seconds are host time inside the C-ABI call
    (argument conversion + the HIP launches it makes), not device time."""
    global CALL_PROFILE
    import time
    lib = load()
    if CALL_PROFILE is not None:
        return CALL_PROFILE
    CALL_PROFILE = {}
    for name in PROTOTYPES:
        fn = getattr(lib, name)
        rec = CALL_PROFILE.setdefault(name, [0, 0.0])

        def wrapped(*a, _fn=fn, _rec=rec, _t=time.perf_counter):
            t0 = _t()
            r = _fn(*a)
            _rec[1] += _t() - t0
            _rec[0] += 1
            return r
        setattr(lib, name, wrapped)
    return CALL_PROFILE


def error_string(code):
    return load().gs_error_string(code).decode()


def check(code, what):
    if code != 0:
        raise HipLibraryError("%s failed: %s (code %d)" % (what, error_string(code), code))
