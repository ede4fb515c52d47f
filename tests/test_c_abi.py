"""The C-ABI shared library loads and exports every symbol declared in include/gaiaseg_hip.h, and
the ctypes binding agrees with the header (no compute calls: this runs without a GPU)."""
import ctypes
import os
import re

import pytest

from gaia_seg_amd.hip import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gaiaseg_hip.h")


def _header_decls():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"typedef struct .*?\} \w+;", "", text, flags=re.S)
    decls = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(gs_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        name, args = m.group(2), m.group(3).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        decls[name] = n
    return decls


def test_header_and_binding_agree():
    decls = _header_decls()
    assert len(decls) >= 30
    assert set(decls) == set(lib.PROTOTYPES), (set(decls) ^ set(lib.PROTOTYPES))
    for name, nargs in decls.items():
        assert len(lib.PROTOTYPES[name][1]) == nargs, name


def test_library_exports_every_symbol():
    if not os.path.exists(lib.LIB_PATH):
        lib.build()
    cdll = ctypes.CDLL(lib.LIB_PATH)
    for name in _header_decls():
        assert hasattr(cdll, name), name


def test_load_binds_and_reports_identity():
    L = lib.load()
    assert L.gs_abi_version() == lib.ABI_VERSION
    assert L.gs_target_arch() == b"gfx950"
    assert lib.error_string(0) == "success"
    assert "NULL" in lib.error_string(-4)


def test_descriptor_struct_sizes_match_header():
    # gs_conv_desc: 14 int32 + 4 int64 + 4 int32 + 1 pointer ; gs_ce_desc: 6 int32 + 4 int64 + 2 int32
    assert ctypes.sizeof(lib.ConvDesc) == 14 * 4 + 4 * 8 + 4 * 4 + 8
    assert ctypes.sizeof(lib.CeDesc) == 6 * 4 + 4 * 8 + 2 * 4
    assert ctypes.sizeof(lib.SlideDesc) == 16 * 4


def test_argument_validation_needs_no_gpu():
    """Bad descriptors are rejected before any launch (return codes, not exceptions)."""
    L = lib.load()
    d = lib.ConvDesc()
    assert L.gs_conv2d_workspace_bytes(ctypes.byref(d)) == 0
    assert L.gs_conv2d_forward(ctypes.byref(d), None, None, None, None, None, None, 0, None) == -1
    assert L.gs_sgd_step(None, None, None, 16, 0.1, 0.9, 0.0, 1.0, 0, None) == -4
    with pytest.raises(lib.HipLibraryError):
        lib.check(-3, "probe")


def test_product_path_fails_loudly_without_gpu_tensor():
    """There is no CPU fallback: CPU tensors are refused by every module forward."""
    import torch
    from gaia_seg_amd.core.bricks import DynamicBatchNorm2d, DynamicConv2d
    with pytest.raises(lib.HipLibraryError):
        DynamicConv2d(8, 8, 3, padding=1)(torch.randn(1, 8, 4, 4))
    with pytest.raises(lib.HipLibraryError):
        DynamicBatchNorm2d(8)(torch.randn(2, 8, 4, 4))


def _plan(L, M, N, K, max_splits=64):
    bm, bn, sp, ks = (ctypes.c_int32() for _ in range(4))
    assert L.gs_debug_query_plan(M, N, K, max_splits, ctypes.byref(bm), ctypes.byref(bn),
                                 ctypes.byref(sp), ctypes.byref(ks)) == 0
    return bm.value, bn.value, sp.value, ks.value


def test_conv_planner_invariants_need_no_gpu():
    """The tile / split-K cost model (csrc/igemm_core.h make_plan) is host arithmetic: check its
    invariants and a few decisions the r01 sweeps pinned down."""
    L = lib.load()
    shapes = [(65536, 64, 576), (16384, 128, 1152), (4096, 256, 2304), (1024, 512, 4608),
              (1024, 512, 18432), (4096, 192, 1728), (65536, 256, 64), (1024, 2048, 512), (99, 48, 1152)]
    for M, N, K in shapes:
        for ms in (64, 512):
            bm, bn, sp, ks = _plan(L, M, N, K, ms)
            nk = -(-K // 16)
            assert bm == 64 and bn in (80, 64, 48, 32)
            assert 1 <= sp <= ms and sp * ks >= nk and (sp - 1) * ks < nk     # K range covered exactly
            assert sp == 1 or ks >= 4                                          # no degenerate splits
            assert sp * M * N * 4 <= 96 << 20 or sp == 1                        # slab bound
            assert _plan(L, M, N, K, ms) == (bm, bn, sp, ks)                    # deterministic (cached)
    # a grid that already fills the chip several times over is never split
    assert _plan(L, 65536, 64, 576)[2] == 1
    assert _plan(L, 65536, 256, 64)[2] == 1
    # few tiles and a long K range: split until the workgroups make whole rounds of the 256 CUs
    bm, bn, sp, ks = _plan(L, 1024, 512, 18432)
    tiles = (1024 // 64) * -(-512 // bn)
    assert sp > 1 and (tiles * sp) % 256 == 0
    # the least-padded width wins when the tile count is comparable
    assert _plan(L, 16384, 96, 864)[1] == 48


def test_x3_dgrad_cases_pass_the_dispatch_gate_without_gpu():
    """The operator cases of tests/test_dgrad_x3_gpu.py are sized to reach the bf16x3 data-gradient
    loop.  gs_debug_query_conv_launch is host arithmetic, so the claim is checked here already (the
    GPU test re-checks it against the launch that really happened)."""
    if os.environ.get("GS_X3", "4") == "0":
        pytest.skip("bf16x3 switched off")
    from test_dgrad_x3_gpu import X3_CASES, _desc
    L = lib.load()
    seen_bn, seen_split, seen_odd = set(), False, False
    for case in X3_CASES:
        n, h, w, ci, co, k, dil, ci_max, co_ld, ldx, ldy, acc, force = case
        d = _desc(lib, n, h, w, ci, co, k, dil, ci_max, co_ld, ldx, ldy)
        if force:
            assert L.gs_debug_force_plan(*force) == 0
        q = lib.DebugLaunch()
        try:
            assert L.gs_debug_query_conv_launch(ctypes.byref(d), lib.OP_DGRAD, ctypes.byref(q)) == 0
        finally:
            L.gs_debug_force_plan(0, 0, 0)
        if case is X3_CASES[-1]:
            assert q.kloop != lib.KLOOP_BF16X3 and q.splits == 4 and q.ksteps_per_split < 48
            continue
        assert q.kloop == lib.KLOOP_BF16X3, case
        tiles = -(-n * h * w // 64) * -(-ci // q.bn) * q.splits
        assert q.bm == 64 and q.bn in (64, 48) and tiles >= 512 and q.ksteps_per_split >= 4
        seen_bn.add(q.bn)
        seen_split |= q.splits > 1 and q.ksteps_per_split >= 48
        seen_odd |= q.splits == 1 and q.ksteps_per_split % 2 == 1
    assert seen_bn == {64, 48} and seen_split and seen_odd
    # forward and weight gradient of a bf16x3 data-gradient shape stay on the fp32 loops
    d = _desc(lib, 2, 64, 64, 1024, 256, 1, 1, 1024, 256, 1024, 256)
    assert L.gs_debug_query_conv_launch(ctypes.byref(d), lib.OP_DGRAD, ctypes.byref(q)) == 0
    assert q.kloop == lib.KLOOP_BF16X3
    for op in (lib.OP_FORWARD, lib.OP_WGRAD):
        assert L.gs_debug_query_conv_launch(ctypes.byref(d), op, ctypes.byref(q)) == 0
        assert q.kloop in (lib.KLOOP_FP32, lib.KLOOP_FP32_PAIRS) and q.op == op
    assert L.gs_debug_query_conv_launch(ctypes.byref(d), 7, ctypes.byref(q)) == -1
    assert ctypes.sizeof(lib.DebugLaunch) == 8 * 4


def test_stream_1x1_cases_dispatch_without_gpu():
    """tests/test_stream_1x1_gpu.py's shapes reach the streaming 1x1 kernel (host arithmetic), the
    stage-3/4 shapes and the 3x3s do not."""
    if os.environ.get("GS_STREAM", "1") == "0":
        pytest.skip("streaming kernel switched off")
    from test_stream_1x1_gpu import STREAM_CASES, _sdesc
    L = lib.load()
    q = lib.DebugLaunch()
    widths = set()
    # production dispatch (mode 1): one column block, short data-gradient contractions
    d = _sdesc(lib, STREAM_CASES[0])      # 64 -> 256 at 2 x 128 x 256
    assert L.gs_debug_query_conv_launch(ctypes.byref(d), lib.OP_FORWARD, ctypes.byref(q)) == 0
    assert q.kloop == lib.KLOOP_STREAM and q.bn == 256
    assert L.gs_debug_query_conv_launch(ctypes.byref(d), lib.OP_DGRAD, ctypes.byref(q)) == 0
    assert q.kloop == lib.KLOOP_BF16X3    # K = 256: stays on the tile kernel's bf16x3 loop
    assert L.gs_debug_set_stream_mode(2) == 0 and L.gs_debug_set_stream_mode(5) == -1
    try:
        _stream_all_shapes(L, STREAM_CASES, _sdesc, q, widths)
    finally:
        L.gs_debug_set_stream_mode(-1)


def _stream_all_shapes(L, STREAM_CASES, _sdesc, q, widths):
    for case in STREAM_CASES:
        d = _sdesc(lib, case)
        for op in case[-1]:
            assert L.gs_debug_query_conv_launch(ctypes.byref(d), op, ctypes.byref(q)) == 0
            assert q.kloop == lib.KLOOP_STREAM and q.bm == 128 and q.splits == 1, (case, op)
            widths.add(q.bn)
    assert widths == {64, 128, 256}
    for n, h, w, ci, co, k in [(2, 32, 64, 256, 1024, 1), (2, 16, 32, 2048, 512, 1), (2, 128, 256, 64, 64, 3),
                               (2, 64, 128, 512, 128, 1)]:
        d = lib.ConvDesc(N=n, H=h, W=w, Ci=ci, Co=co, Ci_max=ci, Co_ld=co, KH=k, KW=k, stride=1, pad=k // 2,
                         dil=1, Ho=h, Wo=w, x_sn=h * w * ci, x_sh=w * ci, x_sw=ci, x_sc=1, ldy=co,
                         ld_add=0, role=0, reserved=0, in_affine=None)
        assert L.gs_debug_query_conv_launch(ctypes.byref(d), lib.OP_FORWARD, ctypes.byref(q)) == 0
        assert q.kloop != lib.KLOOP_STREAM
