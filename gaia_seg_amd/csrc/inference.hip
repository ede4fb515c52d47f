// Inference epilogue for whole-image and sliding-window test mode as ONE gather
// kernel.
//
// Replaces gaiaseg/models/segmentors/dynamic_distiller.py:416-459 (slide_inference: per window
// `preds += F.pad(resize(logit))`, `count_mat += 1`, `preds / count_mat`, optional resize to
// ori_shape), :461-473 (whole_inference), :475-508 (softmax, flip back), :510-540 (argmax; aug_test
// sums the probabilities of the augmented views).
//
// The reference materialises a [N,19,H,W] fp32 tensor (159 MB at 1024x2048) and makes a
// read-modify-write pass over a crop of it per window, then a divide pass, a softmax pass, an argmax
// pass.  Here the low-resolution logits of ALL windows stay resident (9 x 16x32x20 floats = 369 KB
// for BASELINE config 5: L2 / L1 hits) and one kernel computes, per output pixel, the sum over the
// windows that cover it of their bilinearly up-sampled logits, the division by the cover count (the
// window list is a product of row and column intervals, so the count is (#rows) x (#columns) and no
// count_mat exists), the second resize to ori_shape when the image was rescaled, softmax / argmax and
// the flip.  HBM traffic: the label map (8 B / pixel) and/or the probabilities (4*C B / pixel),
// written once.
#include <algorithm>
#include "common.h"
#include "resize.h"

namespace gs {

constexpr int kMaxAxis = 64;   // windows per axis
constexpr int kCQ = 8;         // class quads held in registers per pass (32 classes)

struct SlideArgs {
  gs_slide_desc d;
  float sh, sw;     // window-local scale low-res -> crop
  float rh, rw;     // rescale image -> output (only used when the sizes differ)
  int ys[kMaxAxis], xs[kMaxAxis];
};

// value quads [q0, q0 + kCQ) of the normalised prediction at image pixel (y, x)
__device__ __forceinline__ void pixel_pred(const SlideArgs& a, const float* __restrict__ logits,
                                           int n, int y, int x, int q0, f32x4 (&v)[kCQ]) {
  const gs_slide_desc& d = a.d;
#pragma unroll
  for (int q = 0; q < kCQ; ++q) v[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int C4 = d.ld >> 2;
  int cy = 0, cx = 0;
  for (int ix = 0; ix < d.nx; ++ix) cx += (x >= a.xs[ix] && x < a.xs[ix] + d.wc) ? 1 : 0;
  for (int iy = 0; iy < d.ny; ++iy) {
    const int ly_ = y - a.ys[iy];
    if (ly_ < 0 || ly_ >= d.hc) continue;
    ++cy;
    const Lerp ly = lerp_coord(ly_, a.sh, d.hl, d.align_corners);
    for (int ix = 0; ix < d.nx; ++ix) {
      const int lx_ = x - a.xs[ix];
      if (lx_ < 0 || lx_ >= d.wc) continue;
      const Lerp lx = lerp_coord(lx_, a.sw, d.wl, d.align_corners);
      const float* base = logits + (((long)(iy * d.nx + ix) * d.N + n) * d.hl) * d.wl * d.ld;
      const f32x4* r0 = reinterpret_cast<const f32x4*>(base + (long)ly.i0 * d.wl * d.ld);
      const f32x4* r1 = reinterpret_cast<const f32x4*>(base + (long)ly.i1 * d.wl * d.ld);
      const int o0 = lx.i0 * C4, o1 = lx.i1 * C4;
#pragma unroll
      for (int q = 0; q < kCQ; ++q) {
        if (q0 + q < C4) {
          const f32x4 t00 = r0[o0 + q0 + q], t01 = r0[o1 + q0 + q];
          const f32x4 t10 = r1[o0 + q0 + q], t11 = r1[o1 + q0 + q];
          // ATen upsample_bilinear2d order: h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11)
          v[q] += ly.l0 * (lx.l0 * t00 + lx.l1 * t01) + ly.l1 * (lx.l0 * t10 + lx.l1 * t11);
        }
      }
    }
  }
  const float cnt = (float)(cy * cx);
  if (cnt > 1.f) {
#pragma unroll
    for (int q = 0; q < kCQ; ++q) v[q] = v[q] / cnt;
  }
}

// value quads at OUTPUT pixel (oy, ox): identity, or the bilinear resize image -> ori_shape
template <bool RESCALE>
__device__ __forceinline__ void output_pred(const SlideArgs& a, const float* __restrict__ logits,
                                            int n, int oy, int ox, int q0, f32x4 (&v)[kCQ]) {
  if (!RESCALE) {
    pixel_pred(a, logits, n, oy, ox, q0, v);
    return;
  }
  const gs_slide_desc& d = a.d;
  const Lerp ly = lerp_coord(oy, a.rh, d.H, d.align_corners);
  const Lerp lx = lerp_coord(ox, a.rw, d.W, d.align_corners);
  f32x4 p00[kCQ], p01[kCQ], p10[kCQ], p11[kCQ];
  pixel_pred(a, logits, n, ly.i0, lx.i0, q0, p00);
  pixel_pred(a, logits, n, ly.i0, lx.i1, q0, p01);
  pixel_pred(a, logits, n, ly.i1, lx.i0, q0, p10);
  pixel_pred(a, logits, n, ly.i1, lx.i1, q0, p11);
#pragma unroll
  for (int q = 0; q < kCQ; ++q)
    v[q] = ly.l0 * (lx.l0 * p00[q] + lx.l1 * p01[q]) + ly.l1 * (lx.l0 * p10[q] + lx.l1 * p11[q]);
}

// One thread per output pixel.  labels[n, oy, ox] = argmax_c (probs_in + softmax(pred))[c];
// probs_out[n, c, oy, ox] = probs_in + softmax(pred)  (NCHW, like the reference's `inference`).
template <bool RESCALE>
__global__ __launch_bounds__(256) void slide_fuse_kernel(const SlideArgs a,
                                                         const float* __restrict__ logits,
                                                         const float* probs_in, float* probs_out,
                                                         int64_t* __restrict__ labels) {
  const gs_slide_desc& d = a.d;
  const long plane = (long)d.Ho * d.Wo;
  const long total = (long)d.N * plane;
  const int C4 = d.ld >> 2;
  const bool want_probs = probs_in != nullptr || probs_out != nullptr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % d.Wo);
    const long r = i / d.Wo;
    const int oy = (int)(r % d.Ho);
    const int n = (int)(r / d.Ho);
    // the view's pixel that lands at (oy, ox) after the flip back (dynamic_distiller.py:497-505)
    const int sy = d.flip == 2 ? d.Ho - 1 - oy : oy;
    const int sx = d.flip == 1 ? d.Wo - 1 - ox : ox;
    f32x4 v[kCQ];
    float m = -__builtin_huge_valf(), s = 0.f;
    int amax = 0;
    if (!want_probs) {
      // label map only: softmax is monotone, the arg max of the normalised logits is the label --
      // no exponentials at all (r03 evaluated the running softmax sum here too: 19 expf per pixel,
      // the larger half of this path's VALU work)
      for (int q0 = 0; q0 < C4; q0 += kCQ) {
        output_pred<RESCALE>(a, logits, n, sy, sx, q0, v);
#pragma unroll
        for (int q = 0; q < kCQ; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int c = (q0 + q) * 4 + e;
            if (c < d.C && v[q][e] > m) { m = v[q][e]; amax = c; }
          }
      }
      labels[i] = amax;
      continue;
    }
    for (int q0 = 0; q0 < C4; q0 += kCQ) {
      output_pred<RESCALE>(a, logits, n, sy, sx, q0, v);
#pragma unroll
      for (int q = 0; q < kCQ; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = (q0 + q) * 4 + e;
          if (c < d.C) {
            const float z = v[q][e];
            if (z > m) { s = s * expf(m - z) + 1.f; m = z; amax = c; }
            else s += expf(z - m);
          }
        }
    }
    const float inv = 1.f / s;
    const long pbase = (long)n * d.C * plane + (long)oy * d.Wo + ox;
    float best = -__builtin_huge_valf();
    int bestc = 0;
    for (int q0 = 0; q0 < C4; q0 += kCQ) {
      if (C4 > kCQ) output_pred<RESCALE>(a, logits, n, sy, sx, q0, v);   // (one chunk: still in v)
#pragma unroll
      for (int q = 0; q < kCQ; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = (q0 + q) * 4 + e;
          if (c < d.C) {
            float p = expf(v[q][e] - m) * inv;
            if (probs_in) p += probs_in[pbase + (long)c * plane];
            if (probs_out) probs_out[pbase + (long)c * plane] = p;
            if (p > best) { best = p; bestc = c; }
          }
        }
    }
    if (labels) labels[i] = bestc;
  }
}

}  // namespace gs

using namespace gs;

extern "C" int gs_slide_fuse(const gs_slide_desc* d, const int32_t* win_y, const int32_t* win_x,
                             const float* logits, const float* probs_in, float* probs_out,
                             int64_t* labels, void* stream) {
  if (!d || !win_y || !win_x || !logits) return GS_E_NULL;
  if (!labels && !probs_out) return GS_E_NULL;
  if (d->N <= 0 || d->C <= 0 || d->hl <= 0 || d->wl <= 0 || d->hc <= 0 || d->wc <= 0 ||
      d->H <= 0 || d->W <= 0 || d->Ho <= 0 || d->Wo <= 0)
    return GS_E_BADARG;
  if (d->ny <= 0 || d->nx <= 0 || d->ny > kMaxAxis || d->nx > kMaxAxis) return GS_E_BADARG;
  if (d->flip < 0 || d->flip > 2 || d->reserved != 0) return GS_E_BADARG;
  if ((d->ld & 3) || d->ld < d->C || !aligned16(logits)) return GS_E_ALIGN;
  SlideArgs a{};
  a.d = *d;
  // every image pixel must be covered, every window must lie inside the image
  // (dynamic_distiller.py:446 asserts count_mat != 0)
  for (int axis = 0; axis < 2; ++axis) {
    const int n = axis ? d->nx : d->ny, len = axis ? d->W : d->H, ext = axis ? d->wc : d->hc;
    const int32_t* src = axis ? win_x : win_y;
    int* dst = axis ? a.xs : a.ys;
    int covered_to = 0;
    for (int i = 0; i < n; ++i) {
      if (src[i] < 0 || src[i] + ext > len) return GS_E_BADARG;
      if (src[i] > covered_to) return GS_E_BADARG;
      covered_to = std::max(covered_to, src[i] + ext);
      dst[i] = src[i];
    }
    if (covered_to < len) return GS_E_BADARG;
  }
  a.sh = resize_scale(d->hl, d->hc, d->align_corners);
  a.sw = resize_scale(d->wl, d->wc, d->align_corners);
  a.rh = resize_scale(d->H, d->Ho, d->align_corners);
  a.rw = resize_scale(d->W, d->Wo, d->align_corners);
  const long total = (long)d->N * d->Ho * d->Wo;
  // one wave owns 64 consecutive pixels of an output row; plenty of blocks to fill 256 CUs
  const int grid = (int)std::min<long>(ceil_div(total, 256), (long)num_cu() * 16);
  hipStream_t st = as_stream(stream);
  if (d->Ho == d->H && d->Wo == d->W)
    hipLaunchKernelGGL(slide_fuse_kernel<false>, dim3(grid), dim3(256), 0, st, a, logits, probs_in,
                       probs_out, labels);
  else
    hipLaunchKernelGGL(slide_fuse_kernel<true>, dim3(grid), dim3(256), 0, st, a, logits, probs_in,
                       probs_out, labels);
  return launch_status();
}
