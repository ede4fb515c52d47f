// DynConv2d forward (implicit GEMM, fp32 MFMA) — see igemm_core.h
#include <map>
#include <mutex>
#include <vector>
#include "igemm_core.h"
#include "igemm_stream.h"

using namespace gs;

size_t gs_dgrad_strided_slab_bytes(const gs_conv_desc* d);  // igemm_dgrad.hip
size_t gs_wgrad_slab_bytes(const gs_conv_desc* d);          // igemm_wgrad.hip
#include "fused_internal.h"

extern "C" size_t gs_conv2d_workspace_bytes(const gs_conv_desc* d) {
  if (check_desc(d) != GS_OK) return 0;
  size_t b = 0;
  {
    const Plan pl = plan_fwd(d);
    b = std::max(b, slab_bytes(pl, (long)d->N * d->Ho * d->Wo, d->Co));
  }
  if (d->x_sc == 1 && (d->Ci & 3) == 0) {
    const Plan pl = plan_dgrad(d);
    b = std::max(b, slab_bytes(pl, (long)d->N * d->H * d->W, d->Ci));
    if (d->stride > 1) b = std::max(b, gs_dgrad_strided_slab_bytes(d));
  }
  b = std::max(b, gs_wgrad_slab_bytes(d));
  return b;
}

namespace gs {
extern int g_splitk_inkernel;   // capi_misc.hip: -1 = GS_SPLITK_INKERNEL (default on), 0 / 1 = forced
constexpr long kMaxTickets = 16384;

extern int g_col_finalize;      // capi_misc.hip: -1 = GS_COL_FINALIZE (default OFF: measured neutral), 0 / 1 = forced
constexpr long kMaxColTickets = 1024;

// One zero-at-rest counter buffer per (device, stream): [kMaxTickets tile counters | kMaxColTickets
// column counters].  Buffers are carved out of chunks of kChunk that are allocated and zeroed OUTSIDE
// stream capture (hipMalloc is not permitted while a stream captures, and a step graph is captured on
// streams the eager warm-up never saw): a stream met for the first time during capture takes a spare
// buffer of an earlier chunk; without one it gets NULL and that launch keeps the separate reduce.
static unsigned* ticket_buffer(hipStream_t st) {
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, unsigned*> bufs;
  static std::map<int, std::vector<unsigned*>> spares;   // per device
  constexpr int kChunk = 16;
  constexpr size_t kWords = kMaxTickets + kMaxColTickets;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  std::lock_guard<std::mutex> lock(mu);
  const auto key = std::make_pair(dev, st);
  const auto it = bufs.find(key);
  if (it != bufs.end()) return it->second;
  std::vector<unsigned*>& sp = spares[dev];
  if (sp.empty()) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (cs != hipStreamCaptureStatusNone) return nullptr;   // (not cached: an eager call may allocate later)
    unsigned* p = nullptr;
    const size_t bytes = kChunk * kWords * sizeof(unsigned);
    // zeroed on this stream and waited for: the spares go to other streams later
    if (hipMalloc(&p, bytes) != hipSuccess || hipMemsetAsync(p, 0, bytes, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
      (void)hipGetLastError();
      if (p) (void)hipFree(p);
      return nullptr;
    }
    for (int i = kChunk - 1; i >= 0; --i) sp.push_back(p + (size_t)i * kWords);
  }
  unsigned* buf = sp.back();
  sp.pop_back();
  bufs[key] = buf;
  return buf;
}

unsigned* splitk_tickets(hipStream_t st, long ntiles) {
  static const int env_on = env_int("GS_SPLITK_INKERNEL", 1);
  const int on = g_splitk_inkernel >= 0 ? g_splitk_inkernel : env_on;
  if (!on || ntiles > kMaxTickets) return nullptr;
  return ticket_buffer(st);
}

// kind: 1 = forward (BatchNorm statistics), 2 = data gradient (BatchNorm-backward sums); the mode is a
// mask of the kinds that merge inside the launch (GS_COL_FINALIZE = 0 ... 3)
unsigned* column_tickets(hipStream_t st, long tiles_m, long tiles_n, int kind) {
  static const int env_on = env_int("GS_COL_FINALIZE", 0);
  static const int max_tiles = env_int("GS_COL_FINALIZE_MAX", 160);
  const int on = g_col_finalize >= 0 ? g_col_finalize : env_on;
  if (!(on & kind) || tiles_m > max_tiles || tiles_n > kMaxColTickets) return nullptr;
  unsigned* p = ticket_buffer(st);
  return p ? p + kMaxTickets : nullptr;
}

// the streaming 1x1 kernel takes a forward when: 1x1, stride 1, no padding, NHWC x with contiguous
// pixel rows, no bias / addend, and stream_plan() finds a column-block width whose weights fit in LDS
static StreamPlan stream_fwd_plan(const gs_conv_desc* d, bool fast, const float* bias,
                                  const float* addend) {
  StreamPlan none{0, 0, 0, 0, 0};
  if (!fast || bias || addend || d->KH != 1 || d->KW != 1 || d->stride != 1 || d->pad != 0)
    return none;
  if (d->x_sh != (int64_t)d->W * d->x_sw || d->x_sn != (int64_t)d->H * d->x_sh) return none;
  if (d->in_affine && d->Ci > 256) return none;
  return stream_plan((long)d->N * d->Ho * d->Wo, d->Co, d->Ci, false, d->ldy);
}

// Forward with the options of the fused conv+BN entry point (fused_layers.hip):
//   want_stats: the caller wants BatchNorm batch statistics of y.  Without split-K the epilogue
//     writes per-tile partials to the start of `workspace` (info->mode = 1); with split-K the
//     reduce launch is left to the caller, who fuses it with the statistics pass (mode = 2).
//     mode = 0: y is complete, no statistics were produced.
int conv2d_forward_impl(const gs_conv_desc* d, const float* x, const float* w, const float* bias,
                        const float* addend, float* y, void* workspace, size_t workspace_bytes,
                        void* stream, bool want_stats, ConvFwdInfo* info) {
  int rc = check_desc(d);
  if (rc != GS_OK) return rc;
  if (!x || !w || !y) return GS_E_NULL;
  if (!aligned16(w) || !aligned16(y) || (bias && !aligned16(bias)) || (addend && !aligned16(addend)))
    return GS_E_ALIGN;
  if (addend && ((d->ld_add & 3) || d->ld_add < d->Co)) return GS_E_ALIGN;
  const bool vec = x_is_vector(d);
  if (vec && !aligned16(x)) return GS_E_ALIGN;
  const Plan pl = plan_fwd(d);
  const long M = (long)d->N * d->Ho * d->Wo;
  const size_t need = slab_bytes(pl, M, d->Co);
  if (need > workspace_bytes || (need && !workspace)) return GS_E_WORKSPACE;

  IgemmArgs a{};
  a.src = x; a.dense = w; a.out = y; a.slab = need ? static_cast<float*>(workspace) : nullptr;
  a.bias = bias; a.addend = addend;
  a.s_n = d->x_sn; a.s_h = d->x_sh; a.s_w = d->x_sw; a.s_c = d->x_sc;
  a.Hs = d->H; a.Ws = d->W; a.Cs = d->Ci;
  a.Hp = d->Ho; a.Wp = d->Wo; a.npix = (int)M;
  a.KW = d->KW; a.taps = d->KH * d->KW;
  a.mul_h = a.mul_w = d->stride; a.base_h = a.base_w = -d->pad;
  a.step_h = a.step_w = d->dil; a.div_h = a.div_w = 1;
  a.d_tap = (long)d->Ci_max * d->Co_ld; a.d_row = d->Co_ld; a.n_lim = d->Co;
  a.M = (int)M; a.Nn = d->Co; a.Ktot = a.taps * d->Ci;
  a.ld_out = d->ldy; a.ld_add = d->ld_add;
  a.nk_total = pl.nk_total; a.nk_per_split = pl.nk_per_split;
  a.accumulate = 0; a.tiles_m = pl.tiles_m; a.tiles_n = pl.tiles_n;
  hipStream_t st = as_stream(stream);
  const int ks = ksize_tag(d);
  a.kh_n = d->KH; a.kw_n = d->KW;
  a.d_tap_h = (long)d->KW * a.d_tap; a.d_tap_w = a.d_tap;
  const size_t src_b = (size_t)d->N * d->x_sn * sizeof(float);
  const size_t dense_b = (size_t)a.taps * a.d_tap * sizeof(float);
  a.src_bytes = (unsigned)src_b;
  a.dense_bytes = (unsigned)dense_b;
  const bool fast = vec && fast_rows_ok(d->Ci, ks, src_b, dense_b) && getenv("GS_NO_FAST") == nullptr;
  if (d->in_affine) {
    if (!conv_in_affine_ok(d) || !fast || pl.bm != 64) return GS_E_BADARG;
    if (!aligned16(d->in_affine)) return GS_E_ALIGN;
    a.a_coeffs = d->in_affine;
  }
  // the stem: 7x7 stride-2 conv of the 3-channel NCHW image (stem.hip)
  if (!vec && !bias && !addend && stem_conv_ok(d)) {
    const int tiles = d->N * d->Ho * (d->Wo / 128);
    float* ts = nullptr;
    int mode = 0;
    if (want_stats && info && workspace && aligned16(workspace) &&
        (size_t)3 * d->Co * tiles * sizeof(float) <= workspace_bytes) {
      ts = static_cast<float*>(workspace);
      mode = 1;
    }
    if (info) {
      info->mode = mode; info->splits = 1; info->tiles_m = tiles; info->bm = 128;
      info->slab = nullptr; info->slab_bytes = 0; info->timed = false; info->tile_part = ts;
      info->flops = 2.0 * (double)M * d->Co * d->Ci * d->KH * d->KW;
    }
    return stem_forward(d, x, w, y, ts, nullptr, st);
  }
  // short-K 1x1 convs over many rows: the streaming kernel (igemm_stream.h)
  const StreamPlan sp = stream_fwd_plan(d, fast, bias, addend);
  if (sp.ok) {
    int mode = 0;
    if (want_stats && info) {
      const size_t part_b = (size_t)3 * d->Co * sp.row_groups * sizeof(float);
      if (workspace && part_b <= workspace_bytes && aligned16(workspace)) {
        a.tile_stats = static_cast<float*>(workspace);
        mode = 1;
      }
    }
    if (info) {
      info->mode = mode; info->splits = 1; info->tiles_m = sp.row_groups;
      info->bm = sp.tiles_per_wg * kStreamBM;
      info->slab = nullptr; info->slab_bytes = 0; info->timed = false; info->tile_part = a.tile_stats;
      info->flops = 2.0 * (double)M * d->Co * d->Ci;
    }
    a.slab = nullptr;
    launch_stream<false>(sp, a, st);
    return launch_status();
  }
  const bool timed = d->role == GS_CONV_ROLE_BOTTLENECK3X3 && k3_prof_on();
  if (timed) k3_prof_begin(st);
  const double flops = 2.0 * (double)M * d->Co * d->Ci * d->KH * d->KW;
  // split-K on the fast row kernels: the slabs are combined inside the launch (splitk_publish), the
  // epilogue of the tile's last workgroup is then the unsplit one (statistics included)
  const bool stats_ok = want_stats && info && fast && !bias && !addend;
  const size_t part_b = (size_t)3 * d->Co * pl.tiles_m * sizeof(float);
  const size_t part_off = pl.splits > 1 ? ((need + 255) & ~(size_t)255) : 0;
  const bool part_fits = workspace && part_off + part_b <= workspace_bytes && aligned16(workspace);
  if (splitk_combine_ok(pl) && vec && fast && need < (1ull << 32) && (!stats_ok || part_fits))
    a.tickets = splitk_tickets(st, (long)pl.tiles_m * pl.tiles_n);
  a.slab_bytes = (unsigned)need;
  int mode = 0;
  if (stats_ok) {
    if (pl.splits == 1 || a.tickets) {
      if (part_fits) {
        a.tile_stats = reinterpret_cast<float*>(static_cast<char*>(workspace) + part_off);
        mode = 1;
      }
    } else if (d->ldy >= d->Co && !timed) {
      // (while the K3 timer runs, a split bottleneck conv2 keeps its own reduce launch, so that
      // the timed interval is exactly the conv and its slab reduction)
      mode = 2;
    }
  }
  // few row tiles: the column's last workgroup merges the tile partials and writes the BatchNorm
  // coefficients itself (column_finalize_stats) — no bn_tile_finalize launch
  if (mode == 1 && info->fin_bn && info->fin_coeffs && (pl.splits == 1 || a.tickets) &&
      splitk_combine_tile(pl.bm, pl.bn)) {
    a.col_tickets = column_tickets(st, pl.tiles_m, pl.tiles_n, 1);
    if (a.col_tickets) {
      const gs_bn_args* bn = info->fin_bn;
      a.fin_gamma = bn->gamma; a.fin_beta = bn->beta;
      a.fin_running_mean = bn->update_running ? bn->running_mean : nullptr;
      a.fin_running_var = bn->update_running ? bn->running_var : nullptr;
      a.fin_coeffs = info->fin_coeffs;
      a.fin_eps = bn->eps; a.fin_momentum = bn->momentum;
      a.fin_bm = pl.bm; a.fin_n_last = (int)(M - (long)(pl.tiles_m - 1) * pl.bm);
      a.fin_inv_M = 1.0 / (double)M;
      info->finalized = true;
      __atomic_fetch_add(&g_col_finalized, 1LL, __ATOMIC_RELAXED);
    }
  }
  if (info) {
    info->mode = mode; info->splits = pl.splits; info->tiles_m = pl.tiles_m; info->bm = pl.bm;
    info->slab = a.slab; info->slab_bytes = need; info->timed = timed; info->flops = flops;
    info->tile_part = a.tile_stats;
  }
  if (!vec) launch_rows<false, false, true, 0>(pl, a, st);
  else if (fast && ks == 1) launch_rows_fast<false, 1>(pl, a, st);
  else if (fast && ks == 3 && d->role == GS_CONV_ROLE_BOTTLENECK3X3) launch_rows_fast<false, 3, 1>(pl, a, st);
  else if (fast && ks == 3) launch_rows_fast<false, 3>(pl, a, st);
  else if (ks == 1) launch_rows<false, false, false, 1>(pl, a, st);
  else if (ks == 3) launch_rows<false, false, false, 3>(pl, a, st);
  else launch_rows<false, false, false, 0>(pl, a, st);
  rc = launch_status();
  if (rc != GS_OK) return rc;
  if (mode == 2) return rc;   // the caller reduces the slabs (and closes the K3 timer)
  if (pl.splits > 1 && !a.tickets) {
    launch_reduce(a, pl.splits, 0, st, d->role == GS_CONV_ROLE_BOTTLENECK3X3 ? 1 : 0);
    rc = launch_status();
  }
  if (timed) k3_prof_end(st, flops);
  return rc;
}
}  // namespace gs

extern "C" int gs_conv2d_in_affine_supported(const gs_conv_desc* d) {
  if (check_desc(d) != GS_OK) return 0;
  return conv_in_affine_ok(d) ? 1 : 0;
}

extern "C" int gs_conv2d_forward(const gs_conv_desc* d, const float* x, const float* w,
                                 const float* bias, const float* addend, float* y, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  return conv2d_forward_impl(d, x, w, bias, addend, y, workspace, workspace_bytes, stream, false,
                             nullptr);
}


// Tuning / test hook: the plan the library would use for a GEMM of M x N x K (rows kernels:
// max_splits = 64; wgrad: 512).  Pure host arithmetic.
extern "C" int gs_debug_query_plan(int32_t M, int32_t N, int32_t K, int32_t max_splits, int32_t* bm,
                                   int32_t* bn, int32_t* splits, int32_t* ksteps_per_split) {
  if (!bm || !bn || !splits || !ksteps_per_split) return GS_E_NULL;
  if (M <= 0 || N <= 0 || K <= 0 || max_splits <= 0) return GS_E_BADARG;
  const Plan pl = make_plan(M, N, K, true, max_splits);
  *bm = pl.bm; *bn = pl.bn; *splits = pl.splits; *ksteps_per_split = pl.nk_per_split;
  return GS_OK;
}

// Test hook: what the three conv entry points would launch for this descriptor (mirrors their
// dispatch; tests/test_hip_ops_gpu.py asserts it equals gs_debug_last_conv_launch after real calls).
extern "C" int gs_debug_query_conv_launch(const gs_conv_desc* d, int32_t op, gs_debug_launch* out) {
  if (!out) return GS_E_NULL;
  int rc = check_desc(d);
  if (rc != GS_OK) return rc;
  if (op < GS_OP_FORWARD || op > GS_OP_WGRAD) return GS_E_BADARG;
  const int ks = ksize_tag(d);
  const bool vec = x_is_vector(d);
  const bool no_fast = getenv("GS_NO_FAST") != nullptr;
  const size_t w_bytes = (size_t)d->KH * d->KW * d->Ci_max * d->Co_ld * sizeof(float);
  Plan pl{};
  int kloop = GS_KLOOP_GENERIC;
  const bool aff = d->in_affine != nullptr && op != GS_OP_DGRAD;
  if (!vec && (op == GS_OP_FORWARD || (op == GS_OP_WGRAD && stem_wgrad_on())) && stem_conv_ok(d)) {
    // stem.hip: 128-row tiles, its own loops
    const int tiles = d->N * d->Ho * (d->Wo / 128);
    if (op == GS_OP_FORWARD) pl = Plan{128, d->Co, 1, 37, 37, tiles, 1};
    else {
      const int groups = (int)(stem_wgrad_slab_bytes(d) / ((size_t)147 * d->Co * sizeof(float)));
      pl = Plan{128, d->Co, groups, tiles, (int)ceil_div(tiles, groups), 1, 1};
    }
    *out = gs_debug_launch{op, GS_KLOOP_GENERIC, pl.bm, pl.bn, pl.splits, pl.nk_per_split, 0, 0};
    return GS_OK;
  }
  if (op == GS_OP_FORWARD) {
    pl = plan_fwd(d);
    const size_t src_b = (size_t)d->N * d->x_sn * sizeof(float);
    if (vec && fast_rows_ok(d->Ci, ks, src_b, w_bytes) && !no_fast) {
      kloop = rows_fast_kloop<false>(pl, aff, ks);
      const StreamPlan sp = stream_fwd_plan(d, true, nullptr, nullptr);
      if (sp.ok) {
        kloop = GS_KLOOP_STREAM;
        pl = Plan{kStreamBM, sp.bnw, 1, (int)ceil_div(d->Ci, BK), (int)ceil_div(d->Ci, BK), sp.row_groups, sp.ncb};
      }
    }
  } else if (op == GS_OP_DGRAD) {
    const size_t dy_b = (size_t)d->N * d->Ho * d->Wo * d->ldy * sizeof(float);
    const bool fast = fast_rows_ok(d->Co, ks, dy_b, w_bytes) && !no_fast;
    if (d->stride > 1 && fast && (long)d->N * d->H * d->W * d->x_sw < (1L << 31)) {
      const int s = d->stride;
      const TapAxis th = tap_axis(0, d->pad, d->dil, s, d->KH), tw = tap_axis(0, d->pad, d->dil, s, d->KW);
      const long Mc = (long)d->N * class_len(d->H, s, 0) * class_len(d->W, s, 0);
      pl = make_plan((int)Mc, d->Ci, std::max(1, th.n * tw.n) * d->Co, true);
      kloop = rows_fast_kloop<true>(pl, false);
    } else {
      pl = plan_dgrad(d);
      if (d->stride == 1 && fast) {
        kloop = rows_fast_kloop<true>(pl, false);
        const bool same_rows = d->pad == 0 && d->H == d->Ho && d->W == d->Wo;
        const StreamPlan sp = (ks == 1 && same_rows)
                                  ? stream_plan((long)d->N * d->H * d->W, d->Ci, d->Co, true, d->x_sw)
                                  : StreamPlan{0, 0, 0, 0, 0};
        if (sp.ok && d->x_sc == 1) {
          kloop = GS_KLOOP_STREAM;
          pl = Plan{kStreamBM, sp.bnw, 1, (int)ceil_div(d->Co, BK), (int)ceil_div(d->Co, BK), sp.row_groups, sp.ncb};
        }
      }
    }
  } else {
    pl = plan_wgrad(d);
    const size_t src_b = (size_t)d->N * d->x_sn * sizeof(float);
    const size_t dy_b = (size_t)d->N * d->Ho * d->Wo * d->ldy * sizeof(float);
    if (vec && src_b < (1ull << 31) && dy_b < (1ull << 31) && !no_fast)
      kloop = pair_loop_ok(pl) ? GS_KLOOP_FP32_PAIRS : GS_KLOOP_FP32;
  }
  *out = gs_debug_launch{op, kloop, pl.bm, pl.bn, pl.splits, pl.nk_per_split, aff ? 1 : 0, 0};
  return GS_OK;
}
