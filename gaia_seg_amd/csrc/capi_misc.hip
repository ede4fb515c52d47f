// Library identification and error text of the gaiaseg_hip C-ABI.
#include "common.h"

extern "C" int gs_abi_version(void) { return 9; }

extern "C" const char* gs_target_arch(void) { return "gfx950"; }

extern "C" const char* gs_error_string(int code) {
  switch (code) {
    case GS_OK: return "success";
    case GS_E_BADARG: return "gaiaseg_hip: inconsistent or unsupported descriptor";
    case GS_E_ALIGN: return "gaiaseg_hip: pointer, channel count or stride not aligned as documented";
    case GS_E_WORKSPACE: return "gaiaseg_hip: workspace missing or too small";
    case GS_E_NULL: return "gaiaseg_hip: required pointer is NULL";
    default: break;
  }
  if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
  return "gaiaseg_hip: unknown error code";
}

// Tuning hook (not part of the operator ABI): force the tile / split-K plan of the next conv calls.
// bm in {64,128}, bn in {32,48,64,80,96,128}, splits >= 1; bm = 0 restores the planner.
namespace gs { int g_force_plan[3] = {0, 0, 0}; }
extern "C" int gs_debug_force_plan(int bm, int bn, int splits) {
  if (bm != 0) {
    const bool bn_ok = bn == 32 || bn == 48 || bn == 64 || bn == 80 || bn == 96 || bn == 128;
    if ((bm != 64 && bm != 128) || !bn_ok || splits < 1) return GS_E_BADARG;
  }
  gs::g_force_plan[0] = bm; gs::g_force_plan[1] = bn; gs::g_force_plan[2] = splits;
  return GS_OK;
}

// Dispatch observability for the parity tests (see gs_debug_launch in the header).
namespace gs {
thread_local gs_debug_launch g_last_launch = {-1, 0, 0, 0, 0, 0, 0, 0};
long long g_launch_counts[3][GS_KLOOP_COUNT][3] = {};
double g_launch_flops[3][GS_KLOOP_COUNT] = {};
double g_k3_flops[GS_KLOOP_COUNT] = {};
int g_stream_mode = -1;
int g_x3_fwd = -1;
int g_splitk_inkernel = -1;
long long g_splitk_combined = 0;
int g_col_finalize = -1;
long long g_col_finalized = 0;
}
extern "C" int gs_debug_last_conv_launch(gs_debug_launch* out) {
  if (!out) return GS_E_NULL;
  if (gs::g_last_launch.op < 0) return GS_E_BADARG;
  *out = gs::g_last_launch;
  return GS_OK;
}
extern "C" int gs_debug_conv_launch_counts(int64_t* counts, int32_t reset) {
  for (int o = 0; o < 3; ++o)
    for (int k = 0; k < GS_KLOOP_COUNT; ++k)
      for (int m = 0; m < 3; ++m) {
        if (counts) counts[(o * GS_KLOOP_COUNT + k) * 3 + m] = __atomic_load_n(&gs::g_launch_counts[o][k][m], __ATOMIC_RELAXED);
        if (reset) __atomic_store_n(&gs::g_launch_counts[o][k][m], 0LL, __ATOMIC_RELAXED);
      }
  return GS_OK;
}

extern "C" int gs_debug_k3_flops(double* flops, int32_t reset) {
  for (int k = 0; k < GS_KLOOP_COUNT; ++k) {
    if (flops) flops[k] = gs::flops_load(&gs::g_k3_flops[k]);
    if (reset) gs::flops_store(&gs::g_k3_flops[k], 0.0);
  }
  return GS_OK;
}

extern "C" int gs_debug_set_x3_fwd(int32_t mode) {
  if (mode < -1 || mode > 3) return GS_E_BADARG;
  gs::g_x3_fwd = mode;           // -1: back to the environment's GS_X3_FWD (default 3)
  return GS_OK;
}

// Split-K of the forward / data-gradient row kernels: 1 = the slabs are combined inside the launch by
// each tile's last-arriving workgroup (default), 0 = a separate reduce launch, -1 = GS_SPLITK_INKERNEL.
extern "C" int gs_debug_set_splitk_inkernel(int32_t mode) {
  if (mode < -1 || mode > 1) return GS_E_BADARG;
  gs::g_splitk_inkernel = mode;
  return GS_OK;
}

extern "C" int64_t gs_debug_splitk_combined(int32_t reset) {
  const long long v = __atomic_load_n(&gs::g_splitk_combined, __ATOMIC_RELAXED);
  if (reset) __atomic_store_n(&gs::g_splitk_combined, 0LL, __ATOMIC_RELAXED);
  return v;
}

// Per-tile partials (BatchNorm statistics of a forward conv, BatchNorm-backward sums of a data
// gradient) merged inside the launch by each column tile's last workgroup: 1 = on for launches with
// few row tiles, 0 = always the separate bn_tile_finalize / sum_partials launch (default), -1 =
// GS_COL_FINALIZE.
extern "C" int gs_debug_set_col_finalize(int32_t mode) {
  if (mode < -1 || mode > 3) return GS_E_BADARG;   // mask: 1 = forward statistics, 2 = dgrad sums
  gs::g_col_finalize = mode;
  return GS_OK;
}
extern "C" int64_t gs_debug_col_finalized(int32_t reset) {
  const long long v = __atomic_load_n(&gs::g_col_finalized, __ATOMIC_RELAXED);
  if (reset) __atomic_store_n(&gs::g_col_finalized, 0LL, __ATOMIC_RELAXED);
  return v;
}

extern "C" int gs_debug_num_cu(void) { return gs::num_cu(); }

extern "C" int gs_debug_set_stream_mode(int32_t mode) {
  if (mode < -1 || mode > 2) return GS_E_BADARG;
  gs::g_stream_mode = mode;      // -1: back to the environment's GS_STREAM (default 1)
  return GS_OK;
}

extern "C" int gs_debug_conv_launch_flops(double* flops, int32_t reset) {
  for (int o = 0; o < 3; ++o)
    for (int k = 0; k < GS_KLOOP_COUNT; ++k) {
      if (flops) flops[o * GS_KLOOP_COUNT + k] = gs::flops_load(&gs::g_launch_flops[o][k]);
      if (reset) gs::flops_store(&gs::g_launch_flops[o][k], 0.0);
    }
  return GS_OK;
}

// Stream fork / join without per-call event objects on the host side: work enqueued on `to` after
// this call starts only once everything enqueued on `from` before it has finished.  (hip events
// are re-recordable; hipStreamWaitEvent snapshots the record it sees, so a small ring suffices.)
extern "C" int gs_stream_fork(void* from, void* to) {
  constexpr int kRing = 256;
  static thread_local hipEvent_t ring[kRing];
  static thread_local int next = 0, made = 0;
  if (made < kRing && next == made) {
    hipError_t e = hipEventCreateWithFlags(&ring[made], hipEventDisableTiming);
    if (e != hipSuccess) return static_cast<int>(e);
    ++made;
  }
  hipEvent_t ev = ring[next];
  next = (next + 1) % kRing;
  hipError_t e = hipEventRecord(ev, gs::as_stream(from));
  if (e != hipSuccess) return static_cast<int>(e);
  e = hipStreamWaitEvent(gs::as_stream(to), ev, 0);
  return e == hipSuccess ? GS_OK : static_cast<int>(e);
}
