// DynConv2d weight gradient (implicit GEMM over pixels, split-K) — see igemm_core.h
#include "igemm_core.h"
#include "fused_internal.h"

using namespace gs;

// slab bytes of the weight-gradient path gs_conv2d_wgrad takes for this descriptor
size_t gs_wgrad_slab_bytes(const gs_conv_desc* d) {
  const Plan pl = plan_wgrad(d);
  return std::max(slab_bytes(pl, (long)d->KH * d->KW * d->Ci, d->Co), stem_wgrad_slab_bytes(d));
}

extern "C" int gs_conv2d_wgrad(const gs_conv_desc* d, const float* x, const float* dy, float* dw,
                               void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_desc(d);
  if (rc != GS_OK) return rc;
  if (!x || !dy || !dw) return GS_E_NULL;
  if (!aligned16(dy) || !aligned16(dw)) return GS_E_ALIGN;
  const bool vec = x_is_vector(d);
  if (vec && !aligned16(x)) return GS_E_ALIGN;
  if (!vec && stem_wgrad_on() && stem_conv_ok(d))   // the stem: stem.hip
    return stem_wgrad(d, x, dy, dw, workspace, workspace_bytes, as_stream(stream));
  const Plan pl = plan_wgrad(d);
  const long M = (long)d->KH * d->KW * d->Ci;
  const size_t need = slab_bytes(pl, M, d->Co);
  if (need > workspace_bytes || (need && !workspace)) return GS_E_WORKSPACE;

  IgemmArgs a{};
  a.src = x; a.dense = dy; a.out = dw; a.slab = need ? static_cast<float*>(workspace) : nullptr;
  a.s_n = d->x_sn; a.s_h = d->x_sh; a.s_w = d->x_sw; a.s_c = d->x_sc;
  a.Hs = d->H; a.Ws = d->W; a.Cs = d->Ci;
  a.Hp = d->Ho; a.Wp = d->Wo; a.npix = d->N * d->Ho * d->Wo;
  a.KW = d->KW; a.taps = d->KH * d->KW;
  a.mul_h = a.mul_w = d->stride; a.base_h = a.base_w = -d->pad;
  a.step_h = a.step_w = d->dil; a.div_h = a.div_w = 1;
  a.d_tap = 0; a.d_row = d->ldy; a.n_lim = d->Co;
  a.M = (int)M; a.Nn = d->Co; a.Ktot = a.npix;
  a.o_tap = (long)d->Ci_max * d->Co_ld; a.o_row = d->Co_ld;
  a.nk_total = pl.nk_total; a.nk_per_split = pl.nk_per_split;
  a.tiles_m = pl.tiles_m; a.tiles_n = pl.tiles_n;
  hipStream_t st = as_stream(stream);
  const int ks = ksize_tag(d);
  const size_t src_b = (size_t)d->N * d->x_sn * sizeof(float);
  const size_t dense_b = (size_t)a.npix * d->ldy * sizeof(float);
  a.src_bytes = (unsigned)src_b;
  a.dense_bytes = (unsigned)dense_b;
  const bool fast = vec && src_b < (1ull << 31) && dense_b < (1ull << 31) &&
                    getenv("GS_NO_FAST") == nullptr;
  if (d->in_affine) {
    if (!conv_in_affine_ok(d) || !fast || pl.bm != 64 || (ks != 1 && ks != 3)) return GS_E_BADARG;
    if (!aligned16(d->in_affine)) return GS_E_ALIGN;
    a.a_coeffs = d->in_affine;
  }
  if (fast && ks == 1) launch_wgrad_fast<1>(pl, a, st);
  else if (fast && ks == 3) launch_wgrad_fast<3>(pl, a, st);
  else if (fast) launch_wgrad_fast<0>(pl, a, st);
  else if (!vec) launch_wgrad<true, 0>(pl, a, st);
  else if (ks == 1) launch_wgrad<false, 1>(pl, a, st);
  else if (ks == 3) launch_wgrad<false, 3>(pl, a, st);
  else launch_wgrad<false, 0>(pl, a, st);
  rc = launch_status();
  if (rc != GS_OK) return rc;
  if (pl.splits > 1) {
    launch_reduce(a, pl.splits, 1, st);
    rc = launch_status();
  }
  return rc;
}
