// DynConv2d data gradient (implicit GEMM, fp32 MFMA) — see igemm_core.h
#include "igemm_core.h"
#include "igemm_stream.h"
#include "fused_internal.h"

using namespace gs;

// workspace needed by the strided (parity-class) path: max over the classes
size_t gs_dgrad_strided_slab_bytes(const gs_conv_desc* d) {
  size_t need = 0;
  const int s = d->stride;
  for (int ph = 0; ph < s; ++ph)
    for (int pw = 0; pw < s; ++pw) {
      const TapAxis th = tap_axis(ph, d->pad, d->dil, s, d->KH), tw = tap_axis(pw, d->pad, d->dil, s, d->KW);
      const int Hq = class_len(d->H, s, ph), Wq = class_len(d->W, s, pw);
      if (!th.n || !tw.n || !Hq || !Wq) continue;
      const long Mc = (long)d->N * Hq * Wq;
      const Plan pl = make_plan((int)Mc, d->Ci, th.n * tw.n * d->Co, true);
      need = std::max(need, slab_bytes(pl, Mc, d->Ci));
    }
  return need;
}

// dx[r][0..C) = 0 for a channel slice of a wider buffer (pixel stride ld > C)
static __global__ __launch_bounds__(256) void zero_rows_kernel(float* __restrict__ dx, long rows,
                                                               int c4, long ld) {
  const long total = rows * c4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const long r = i / c4;
    const int q = (int)(i - r * c4);
    *reinterpret_cast<f32x4*>(dx + r * ld + q * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

static int dgrad_strided_fast(const gs_conv_desc* d, const float* dy, const float* w, float* dx,
                              int accumulate, void* workspace, size_t workspace_bytes,
                              hipStream_t st) {
  const int s = d->stride, ks = ksize_tag(d);
  const long d_tap = (long)d->Ci_max * d->Co_ld;
  bool any_empty = false;
  for (int ph = 0; ph < s && !any_empty; ++ph)
    for (int pw = 0; pw < s; ++pw)
      if (!tap_axis(ph, d->pad, d->dil, s, d->KH).n || !tap_axis(pw, d->pad, d->dil, s, d->KW).n)
        any_empty = true;
  if (!accumulate && any_empty) {
    // classes without taps (e.g. 3 of the 4 classes of a 1x1 stride-2 conv) stay zero.  A dense dx is
    // cleared with the 1-D memset (a fill kernel); hipMemset2DAsync is staged through ~30 buffer
    // copies per call on ROCm 7.2 (seen in the r01 trace), so a sliced dx gets its own fill kernel.
    const size_t rows = (size_t)d->N * d->H * d->W;
    if (d->x_sw == d->Ci) {
      hipError_t e = hipMemsetAsync(dx, 0, rows * d->Ci * sizeof(float), st);
      if (e != hipSuccess) return static_cast<int>(e);
    } else {
      const long quads = (long)rows * (d->Ci / 4);
      hipLaunchKernelGGL(zero_rows_kernel, dim3(stream_grid(quads, 256)), dim3(256), 0, st, dx,
                         (long)rows, d->Ci / 4, (long)d->x_sw);
    }
  }
  for (int ph = 0; ph < s; ++ph)
    for (int pw = 0; pw < s; ++pw) {
      const TapAxis th = tap_axis(ph, d->pad, d->dil, s, d->KH), tw = tap_axis(pw, d->pad, d->dil, s, d->KW);
      const int Hq = class_len(d->H, s, ph), Wq = class_len(d->W, s, pw);
      if (!th.n || !tw.n || !Hq || !Wq) continue;
      const long Mc = (long)d->N * Hq * Wq;
      const int ktot = th.n * tw.n * d->Co;
      const Plan pl = make_plan((int)Mc, d->Ci, ktot, true);
      const size_t need = slab_bytes(pl, Mc, d->Ci);
      if (need > workspace_bytes || (need && !workspace)) return GS_E_WORKSPACE;
      IgemmArgs a{};
      a.src = dy; a.out = dx; a.slab = need ? static_cast<float*>(workspace) : nullptr;
      const long tap0 = (long)th.k0 * d->KW + tw.k0;
      a.dense = w + tap0 * d_tap;
      a.s_c = 1; a.s_w = d->ldy; a.s_h = (long)d->Wo * d->ldy; a.s_n = (long)d->Ho * a.s_h;
      a.Hs = d->Ho; a.Ws = d->Wo; a.Cs = d->Co;
      a.Hp = Hq; a.Wp = Wq; a.npix = (int)Mc;
      a.KW = tw.n; a.taps = th.n * tw.n;
      a.mul_h = a.mul_w = 1; a.base_h = th.off0; a.base_w = tw.off0;
      a.step_h = th.step; a.step_w = tw.step; a.div_h = a.div_w = 1;
      a.d_tap = d_tap; a.d_row = d->Co_ld; a.n_lim = d->Ci;
      a.kh_n = th.n; a.kw_n = tw.n;
      a.d_tap_h = (long)th.dk * d->KW * d_tap; a.d_tap_w = (long)tw.dk * d_tap;
      a.M = (int)Mc; a.Nn = d->Ci; a.Ktot = ktot;
      a.ld_out = (int)d->x_sw;
      a.nk_total = pl.nk_total; a.nk_per_split = pl.nk_per_split;
      a.accumulate = accumulate ? 1 : 0; a.tiles_m = pl.tiles_m; a.tiles_n = pl.tiles_n;
      a.src_bytes = (unsigned)((size_t)d->N * a.s_n * sizeof(float));
      a.dense_bytes = (unsigned)(((size_t)d->KH * d->KW - tap0) * d_tap * sizeof(float));
      a.o_s = s; a.o_ph = ph; a.o_pw = pw; a.o_Hq = Hq; a.o_Wq = Wq; a.o_H = d->H; a.o_W = d->W;
      if (splitk_combine_ok(pl)) {   // slabs combined inside the launch (igemm_core.h splitk_publish)
        a.tickets = splitk_tickets(st, (long)pl.tiles_m * pl.tiles_n);
        a.slab_bytes = (unsigned)need;
      }
      if (ks == 1) launch_rows_fast<true, 1>(pl, a, st);
      else launch_rows_fast<true, 3>(pl, a, st);
      int rc = launch_status();
      if (rc != GS_OK) return rc;
      if (pl.splits > 1 && !a.tickets) {
        launch_reduce(a, pl.splits, 0, st);
        rc = launch_status();
        if (rc != GS_OK) return rc;
      }
    }
  return GS_OK;
}

extern "C" int gs_conv2d_dgrad(const gs_conv_desc* d, const float* dy, const float* w, float* dx,
                               int accumulate, void* workspace, size_t workspace_bytes,
                               void* stream) {
  return conv2d_dgrad_impl(d, dy, w, dx, accumulate, workspace, workspace_bytes, stream, nullptr,
                           nullptr);
}

namespace gs {
// per-tile partial sums of the fused BatchNorm-backward epilogue: [2][Ci/4][tiles_m] float4
static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
// workspace of the fused path: per-tile partials (no split-K), or the slabs followed by the row-block
// partials of the fused slab reduce
size_t dgrad_bnbwd_part_bytes(const gs_conv_desc* d) {
  if (check_desc(d) != GS_OK || d->stride != 1 || (d->Ci & 3)) return 0;
  const Plan pl = plan_dgrad(d);
  const long M = (long)d->N * d->H * d->W;
  if (pl.splits != 1)   // slabs, then the partials of the slab reduce or (in-launch combine) of the tiles
    return align256(slab_bytes(pl, M, d->Ci)) +
           std::max(bn_reduce_bnbwd_bytes(M, d->Ci), (size_t)2 * d->Ci * pl.tiles_m * sizeof(float));
  return (size_t)2 * d->Ci * pl.tiles_m * sizeof(float);
}

// the fused BatchNorm-backward epilogue can take this request: mode 1 (mask recomputed), 2 (mask from
// the activation) or 3 (mask bytes), operands present and 16-byte aligned where they are read as quads
static inline bool bnbwd_fuse_ok(const gs_bn_bwd_fuse* bw, const gs_conv_desc* d, const void* workspace) {
  if (!bw || !bw->y || !bw->coeffs || !bw->sums || !workspace || !aligned16(workspace)) return false;
  if (!aligned16(bw->y) || !aligned16(bw->coeffs) || (bw->ldy & 3) != 0 || bw->ldy < d->Ci) return false;
  if (bw->mode == 1) return true;
  if (bw->mode == 2)
    return bw->act && aligned16(bw->act) && (bw->ldact & 3) == 0 && bw->ldact >= d->Ci;
  if (bw->mode == 3) return bw->mask && bw->ldmask >= d->Ci / 4 && bw->reserved2 == 0;
  return false;
}

int conv2d_dgrad_impl(const gs_conv_desc* d, const float* dy, const float* w, float* dx,
                      int accumulate, void* workspace, size_t workspace_bytes, void* stream,
                      const gs_bn_bwd_fuse* bw, int* fused) {
  if (fused) *fused = 0;
  int rc = check_desc(d);
  if (rc != GS_OK) return rc;
  if (!dy || !w || !dx) return GS_E_NULL;
  if (d->x_sc != 1 || (d->Ci & 3) || (d->x_sw & 3)) return GS_E_ALIGN;
  if (d->x_sh != (int64_t)d->W * d->x_sw || d->x_sn != (int64_t)d->H * d->x_sh) return GS_E_BADARG;
  if (!aligned16(dy) || !aligned16(w) || !aligned16(dx)) return GS_E_ALIGN;
  hipStream_t st = as_stream(stream);
  {
    const int ks0 = ksize_tag(d);
    const size_t sb = (size_t)d->N * d->Ho * d->Wo * d->ldy * sizeof(float);
    const size_t db = (size_t)d->KH * d->KW * d->Ci_max * d->Co_ld * sizeof(float);
    if (d->stride > 1 && fast_rows_ok(d->Co, ks0, sb, db) && getenv("GS_NO_FAST") == nullptr &&
        (long)d->N * d->H * d->W * d->x_sw < (1L << 31))
      return dgrad_strided_fast(d, dy, w, dx, accumulate, workspace, workspace_bytes, st);
  }
  const Plan pl = plan_dgrad(d);
  const long M = (long)d->N * d->H * d->W;
  const size_t need = slab_bytes(pl, M, d->Ci);
  if (need > workspace_bytes || (need && !workspace)) return GS_E_WORKSPACE;

  IgemmArgs a{};
  a.src = dy; a.dense = w; a.out = dx; a.slab = need ? static_cast<float*>(workspace) : nullptr;
  a.s_c = 1; a.s_w = d->ldy; a.s_h = (long)d->Wo * d->ldy; a.s_n = (long)d->Ho * a.s_h;
  a.Hs = d->Ho; a.Ws = d->Wo; a.Cs = d->Co;
  a.Hp = d->H; a.Wp = d->W; a.npix = (int)M;
  a.KW = d->KW; a.taps = d->KH * d->KW;
  a.mul_h = a.mul_w = 1; a.base_h = a.base_w = d->pad;
  a.step_h = a.step_w = -d->dil; a.div_h = a.div_w = d->stride;
  a.d_tap = (long)d->Ci_max * d->Co_ld; a.d_row = d->Co_ld; a.n_lim = d->Ci;
  a.M = (int)M; a.Nn = d->Ci; a.Ktot = a.taps * d->Co;
  a.ld_out = (int)d->x_sw; a.ld_add = 0;
  a.nk_total = pl.nk_total; a.nk_per_split = pl.nk_per_split;
  a.accumulate = accumulate ? 1 : 0; a.tiles_m = pl.tiles_m; a.tiles_n = pl.tiles_n;
  a.kh_n = d->KH; a.kw_n = d->KW;
  a.d_tap_h = (long)d->KW * a.d_tap; a.d_tap_w = a.d_tap;
  const int ks = ksize_tag(d);
  const size_t src_b = (size_t)d->N * a.s_n * sizeof(float);
  const size_t dense_b = (size_t)a.taps * a.d_tap * sizeof(float);
  a.src_bytes = (unsigned)src_b;
  a.dense_bytes = (unsigned)dense_b;
  const bool fast = fast_rows_ok(d->Co, ks, src_b, dense_b) && getenv("GS_NO_FAST") == nullptr;
  static const bool no_bnb = getenv("GS_NO_BNBWD_FUSE") != nullptr;
  bool bnb = false, bnb_split = false;
  // short-K 1x1 data gradients over many rows: the streaming kernel (igemm_stream.h); the fused
  // BatchNorm-backward sums come out as one partial per workgroup row range
  // (a padded 1x1 has Ho = H + 2 pad: the streaming kernel maps dy row m to dx row m, so it takes
  // only the unpadded form; the tile kernels handle padding through base_h / base_w)
  const bool same_rows = d->pad == 0 && d->H == d->Ho && d->W == d->Wo;
  const StreamPlan sp = (fast && ks == 1 && d->stride == 1 && same_rows)
                            ? stream_plan(M, d->Ci, d->Co, true, std::max<long>(d->x_sw, bw ? std::max(bw->ldy, bw->ldact) : 0))
                            : StreamPlan{0, 0, 0, 0, 0};
  if (sp.ok) {
    if (bnbwd_fuse_ok(bw, d, workspace) && !no_bnb &&
        (size_t)2 * d->Ci * sp.row_groups * sizeof(float) <= workspace_bytes) {
      a.bw_y = bw->y; a.bw_ldy = bw->ldy; a.bw_act = bw->act; a.bw_ldact = bw->ldact;
      a.bw_coeffs = bw->coeffs; a.bw_mode = bw->mode;
      a.bw_mask = bw->mask; a.bw_ldmask = bw->ldmask;
      a.bw_part = static_cast<float*>(workspace);
      bnb = true;
    }
    a.slab = nullptr;
    launch_stream<true>(sp, a, st);
    rc = launch_status();
    if (bnb && rc == GS_OK) {
      rc = bn_sum_partials(a.bw_part, sp.row_groups, 2 * d->Ci, bw->sums, st);
      if (fused) *fused = 1;
    }
    return rc;
  }
  // split-K on the fast row kernels: combined inside the launch (igemm_core.h splitk_publish); the
  // tile's last workgroup then runs the unsplit epilogue, the fused BatchNorm-backward one included
  if (splitk_combine_ok(pl) && d->stride == 1 && fast && need < (1ull << 32)) {
    a.tickets = splitk_tickets(st, (long)pl.tiles_m * pl.tiles_n);
    a.slab_bytes = (unsigned)need;
  }
  if (!no_bnb && d->stride == 1 && fast && bnbwd_fuse_ok(bw, d, workspace)) {
    const size_t part_b = (size_t)2 * d->Ci * pl.tiles_m * sizeof(float);
    const size_t part_off = pl.splits > 1 ? align256(need) : 0;
    if ((pl.splits == 1 || a.tickets) && part_off + part_b <= workspace_bytes) {
      a.bw_y = bw->y; a.bw_ldy = bw->ldy; a.bw_act = bw->act; a.bw_ldact = bw->ldact;
      a.bw_coeffs = bw->coeffs; a.bw_mode = bw->mode;
      a.bw_mask = bw->mask; a.bw_ldmask = bw->ldmask;
      a.bw_part = reinterpret_cast<float*>(static_cast<char*>(workspace) + part_off);
      bnb = true;
      // few row tiles: the column's last workgroup sums the partials itself (column_finalize_bwsums)
      if (splitk_combine_tile(pl.bm, pl.bn)) a.col_tickets = column_tickets(st, pl.tiles_m, pl.tiles_n, 2);
      if (a.col_tickets) {
        a.fin_bw_sums = bw->sums;
        __atomic_fetch_add(&g_col_finalized, 1LL, __ATOMIC_RELAXED);
      }
    } else if (pl.splits == 1) {
      // (no room for the tile partials: plain dgrad, the caller runs the BatchNorm reduction itself)
    } else if (align256(need) + bn_reduce_bnbwd_bytes(M, d->Ci) <= workspace_bytes) {
      a.tickets = nullptr;
      bnb_split = true;   // the slab reduce does the masking and the sums
    }
  }
  if (d->stride == 1) {
    if (fast && ks == 1) launch_rows_fast<true, 1>(pl, a, st);
    else if (fast && ks == 3) launch_rows_fast<true, 3>(pl, a, st);
    else if (ks == 1) launch_rows<true, false, false, 1>(pl, a, st);
    else if (ks == 3) launch_rows<true, false, false, 3>(pl, a, st);
    else launch_rows<true, false, false, 0>(pl, a, st);
  } else {
    if (ks == 1) launch_rows<true, true, false, 1>(pl, a, st);
    else if (ks == 3) launch_rows<true, true, false, 3>(pl, a, st);
    else launch_rows<true, true, false, 0>(pl, a, st);
  }
  rc = launch_status();
  if (rc != GS_OK) return rc;
  if (bnb_split) {
    const size_t off = align256(need);
    rc = bn_reduce_bnbwd(a.slab, pl.splits, M, d->Ci, dx, (int)d->x_sw, accumulate ? 1 : 0, bw,
                         reinterpret_cast<float*>(static_cast<char*>(workspace) + off),
                         workspace_bytes - off, st);
    if (rc == GS_OK && fused) *fused = 1;
    return rc;
  }
  if (pl.splits > 1 && !a.tickets) {
    launch_reduce(a, pl.splits, 0, st);
    rc = launch_status();
  }
  if (bnb && rc == GS_OK) {
    if (!a.col_tickets) rc = bn_sum_partials(a.bw_part, pl.tiles_m, 2 * d->Ci, bw->sums, st);
    if (fused) *fused = 1;
  }
  return rc;
}
}  // namespace gs

