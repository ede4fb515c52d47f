// DynConv2d data gradient (implicit GEMM, fp32 MFMA) — see igemm_core.h
#include "igemm_core.h"

using namespace gs;

extern "C" int gs_conv2d_dgrad(const gs_conv_desc* d, const float* dy, const float* w, float* dx,
                               int accumulate, void* workspace, size_t workspace_bytes,
                               void* stream) {
  int rc = check_desc(d);
  if (rc != GS_OK) return rc;
  if (!dy || !w || !dx) return GS_E_NULL;
  if (d->x_sc != 1 || (d->Ci & 3) || (d->x_sw & 3)) return GS_E_ALIGN;
  if (d->x_sh != (int64_t)d->W * d->x_sw || d->x_sn != (int64_t)d->H * d->x_sh) return GS_E_BADARG;
  if (!aligned16(dy) || !aligned16(w) || !aligned16(dx)) return GS_E_ALIGN;
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
  hipStream_t st = as_stream(stream);
  const int ks = ksize_tag(d);
  const size_t src_b = (size_t)d->N * a.s_n * sizeof(float);
  const size_t dense_b = (size_t)a.taps * a.d_tap * sizeof(float);
  a.src_bytes = (unsigned)src_b;
  a.dense_bytes = (unsigned)dense_b;
  const bool fast = fast_rows_ok(d->Co, ks, src_b, dense_b) && getenv("GS_NO_FAST") == nullptr;
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
  if (pl.splits > 1) {
    launch_reduce(a, pl.splits, 0, st);
    rc = launch_status();
  }
  return rc;
}

