// conv -> BatchNorm (+ residual) (+ ReLU) as ONE host call per direction, and the live K3 timer.
//
// The reference runs DynConv2d, DynBN and ReLU as three Python-level modules
// (gaiavision DynamicConvModule; DynamicBottleneck.forward at the call site
// gaiaseg/models/utils/dynamic_res_layer.py:84-125).  Every launch of this path is a few tens of
// microseconds of work on an MI355X, so the per-module host cost of an eager framework (~35 us)
// is of the same order as the kernels themselves: with one host call per module the host, not
// the GPU, sets the step time for the deeper subnets.  These entry points issue the launches of a
// conv+BN pair back to back from C (forward: conv, [split-K reduce], BN statistics, finalize,
// apply; backward: BN reduce, BN apply, wgrad on the side stream, dgrad), and let the BatchNorm
// batch statistics come out of the conv (epilogue partials / fused slab reduce, see norm.hip).
#include <vector>
#include "common.h"
#include "fused_internal.h"

using namespace gs;

static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" size_t gs_conv_bn_workspace_bytes(const gs_conv_desc* d) {
  if (!d) return 0;
  const size_t a = gs_conv2d_workspace_bytes(d);
  const int64_t rows = (int64_t)d->N * d->Ho * d->Wo;
  const size_t b = gs_bn_stats_workspace_bytes(rows, d->Co);
  // split-K slabs followed by the partials of the fused reduce + statistics pass; or the per-tile
  // partials of the conv epilogue (3 floats per channel and 64-row tile)
  const size_t fused = align256(a) + bn_fused_reduce_bytes(rows, d->Co);
  // (behind the slabs when a split-K launch combines them itself and writes the tile partials)
  const size_t tiles = align256(a) + (size_t)3 * d->Co * ((rows + 63) / 64) * sizeof(float);
  size_t m = a > b ? a : b;
  if (fused > m) m = fused;
  if (tiles > m) m = tiles;
  const size_t bnb = dgrad_bnbwd_part_bytes(d);   // backward: fused BN-backward epilogue partials
  if (bnb > m) m = bnb;
  return m;
}

extern "C" int gs_conv_bn_forward(const gs_conv_desc* d, const float* x, const float* w,
                                  const gs_bn_args* bn, const float* residual, int32_t ld_res,
                                  float* y, float* coeffs, float* z, int32_t ldz, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  if (!d || !bn || !coeffs) return GS_E_NULL;
  static const bool no_fuse = getenv("GS_NO_STATS_FUSION") != nullptr;
  ConvFwdInfo info{};
  if (bn->use_batch_stats && !no_fuse) { info.fin_bn = bn; info.fin_coeffs = coeffs; }
  int rc = conv2d_forward_impl(d, x, w, nullptr, nullptr, y, workspace, workspace_bytes, stream,
                               bn->use_batch_stats && !no_fuse, &info);
  if (rc != GS_OK) return rc;
  const int64_t rows = (int64_t)d->N * d->Ho * d->Wo;
  const int32_t C = d->Co;
  float* rm = bn->update_running ? bn->running_mean : nullptr;
  float* rv = bn->update_running ? bn->running_var : nullptr;
  if (info.mode == 1 && info.finalized) {
    // the conv's last workgroups merged the tile partials and wrote the coefficients
  } else if (info.mode == 1) {
    rc = bn_tile_finalize(info.tile_part, info.tiles_m, info.bm, rows, C,
                          bn->gamma, bn->beta, bn->eps, bn->momentum, rm, rv, coeffs,
                          as_stream(stream));
  } else if (info.mode == 2) {
    const size_t off = align256(info.slab_bytes);
    if (off > workspace_bytes) return GS_E_WORKSPACE;
    rc = bn_reduce_stats_finalize(info.slab, info.splits, rows, C, y, d->ldy, bn->gamma, bn->beta,
                                  bn->eps, bn->momentum, rm, rv, coeffs,
                                  reinterpret_cast<float*>(static_cast<char*>(workspace) + off),
                                  workspace_bytes - off, as_stream(stream), d->role, info.timed,
                                  info.flops);
  } else if (bn->use_batch_stats) {
    rc = gs_bn_stats_finalize(y, rows, C, d->ldy, bn->gamma, bn->beta, bn->eps, bn->momentum,
                              bn->update_running ? bn->running_mean : nullptr,
                              bn->update_running ? bn->running_var : nullptr, coeffs, workspace,
                              workspace_bytes, stream);
  } else {
    if (!bn->running_mean || !bn->running_var) return GS_E_NULL;
    rc = gs_bn_eval_coeffs(bn->running_mean, bn->running_var, C, bn->gamma, bn->beta, bn->eps,
                           coeffs, stream);
  }
  if (rc != GS_OK) return rc;
  // z == NULL: the consumer applies relu(bn(y)) in its operand loader (gs_conv_desc::in_affine)
  if (!z) return residual ? GS_E_BADARG : GS_OK;
  if (bn->residual_coeffs)
    return bn_apply_resaff(y, rows, C, d->ldy, coeffs, residual, ld_res, bn->residual_coeffs, bn->relu,
                           z, ldz, bn->relu_mask, stream);
  if (bn->relu_mask)
    return bn->relu ? gs_bn_apply_mask(y, rows, C, d->ldy, coeffs, residual, ld_res, z, ldz,
                                       bn->relu_mask, stream)
                    : GS_E_BADARG;
  return gs_bn_apply(y, rows, C, d->ldy, coeffs, residual, ld_res, bn->relu, z, ldz, stream);
}

extern "C" int gs_conv_bn_backward(const gs_conv_desc* d, const float* x, const float* w,
                                   const float* y, const float* z, int32_t ldz,
                                   const float* coeffs, const gs_bn_args* bn, float* dz,
                                   int32_t ld_dz, int32_t mask_mode, int32_t write_g, float* dy,
                                   float* bsums, float* dgamma, float* dbeta, float* dw, float* dx,
                                   int32_t accumulate_dx, void* workspace, size_t workspace_bytes,
                                   void* side_workspace, size_t side_workspace_bytes, void* stream,
                                   void* side_stream, const gs_bn_bwd_fuse* input_bn,
                                   int32_t sums_ready) {
  if (!d || !bn || !coeffs || !dz || !dy || !bsums) return GS_E_NULL;
  if (input_bn && input_bn->reserved != 0) return GS_E_BADARG;
  const int64_t rows = (int64_t)d->N * d->Ho * d->Wo;
  const int32_t C = d->Co;
  int rc = GS_OK;
  if (!sums_ready) {
    rc = gs_bn_bwd_reduce(dz, ld_dz, y, d->ldy, z, ldz, rows, C, coeffs, mask_mode,
                          write_g ? dz : nullptr, ld_dz, bsums, workspace, workspace_bytes, stream);
    if (rc != GS_OK) return rc;
  }
  // dz already holds the masked gradient when write_g (or when a consumer's dgrad epilogue produced
  // it together with the sums): the apply pass must not mask again
  rc = gs_bn_bwd_apply(dz, ld_dz, y, d->ldy, z, ldz, rows, C, coeffs, bsums, (double)rows,
                       (write_g || sums_ready) ? 0 : mask_mode, bn->use_batch_stats, dy, d->ldy,
                       dgamma, dbeta, stream);
  if (rc != GS_OK) return rc;
  if (dw) {
    if (side_stream) {
      rc = gs_stream_fork(stream, side_stream);   // dy ready
      if (rc != GS_OK) return rc;
      rc = gs_conv2d_wgrad(d, x, dy, dw, side_workspace, side_workspace_bytes, side_stream);
    } else {
      rc = gs_conv2d_wgrad(d, x, dy, dw, workspace, workspace_bytes, stream);
    }
    if (rc != GS_OK) return rc;
  }
  if (input_bn && input_bn->fused) *input_bn->fused = 0;
  if (dx) {
    int fused = 0;
    rc = conv2d_dgrad_impl(d, dy, w, dx, accumulate_dx, workspace, workspace_bytes, stream, input_bn,
                           &fused);
    if (input_bn && input_bn->fused) *input_bn->fused = fused;
  }
  return rc;
}

// ------------------------------------------------------------------------------------------
// Live timer of the role-1 (bottleneck conv2, K3) forward launches: HIP events recorded on the
// launch stream around gs_conv2d_forward's kernels (conv + its split-K reduce).  bench.py turns it
// on for the timed steps and reads the sums after a device synchronize.
// ------------------------------------------------------------------------------------------
namespace gs {
struct K3Prof {
  bool on = false;
  std::vector<hipEvent_t> pool;   // pairs: start, stop
  size_t used = 0;
  double flops = 0.0;
};
static K3Prof g_k3;
bool k3_prof_on() { return g_k3.on; }
static hipEvent_t k3_event() {
  if (g_k3.used == g_k3.pool.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    g_k3.pool.push_back(e);
  }
  return g_k3.pool[g_k3.used++];
}
// An interval = two events.  Where the role-1 op is ONE kernel (unsplit, or split-K combined inside the
// launch) the events are attached to that kernel's dispatch (k3_launch_events -> hipExtLaunchKernelGGL)
// and their difference is the kernel's own duration, the figure a rocprofv3 kernel trace reports; where
// a reduce launch follows (GS_SPLITK_INKERNEL=0) or GS_K3_TIMER_MARKERS is set, they are marker events
// recorded on the stream in front of the conv and behind the last launch (the r01-r03 method: it adds
// the 2-3 us between the markers and the dispatch).
struct K3Pending {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  bool open = false, start_set = false, stop_set = false;
};
static thread_local K3Pending t_k3;
void k3_prof_begin(hipStream_t st) {
  (void)st;
  t_k3 = K3Pending{};
  t_k3.e0 = k3_event();
  t_k3.e1 = k3_event();
  t_k3.open = t_k3.e0 && t_k3.e1;
}
bool k3_launch_events(hipStream_t st, bool single_kernel, hipEvent_t* e0, hipEvent_t* e1) {
  if (!t_k3.open || t_k3.start_set) return false;
  static const bool markers = getenv("GS_K3_TIMER_MARKERS") != nullptr;
  t_k3.start_set = true;
  if (single_kernel && !markers) {
    *e0 = t_k3.e0; *e1 = t_k3.e1;
    t_k3.stop_set = true;
    return true;
  }
  (void)hipEventRecord(t_k3.e0, st);
  return false;
}
void k3_prof_end(hipStream_t st, double flops) {
  if (!t_k3.open) return;
  if (!t_k3.start_set) {
    // the op did not go through the fast row kernels: no interval (give the two events back)
    if (g_k3.used >= 2) g_k3.used -= 2;
  } else {
    if (!t_k3.stop_set) (void)hipEventRecord(t_k3.e1, st);
    g_k3.flops += flops;
  }
  t_k3 = K3Pending{};
}
}  // namespace gs

extern "C" int gs_k3_timer_enable(int32_t on) {
  g_k3.on = on != 0;
  if (on) { g_k3.used = 0; g_k3.flops = 0.0; }
  return GS_OK;
}

extern "C" int gs_k3_timer_read(int64_t* launches, double* total_ms, double* total_flops) {
  if (!launches || !total_ms || !total_flops) return GS_E_NULL;
  double ms = 0.0;
  const size_t pairs = g_k3.used / 2;
  for (size_t i = 0; i < pairs; ++i) {
    float t = 0.f;
    hipError_t e = hipEventElapsedTime(&t, g_k3.pool[2 * i], g_k3.pool[2 * i + 1]);
    if (e != hipSuccess) return static_cast<int>(e);
    ms += t;
  }
  *launches = (int64_t)pairs;
  *total_ms = ms;
  *total_flops = g_k3.flops;
  return GS_OK;
}
