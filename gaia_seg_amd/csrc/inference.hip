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

// ---- label map only, no rescale: one thread per strip of kStrip output pixels of a row -------------
// The per-pixel kernel above reads the four corner vectors of every covering window for every pixel:
// 80 16-byte L1 loads per pixel under four windows (r03 / r04 profiles: config 5 slide 98 / 91 us, whole
// 50 / 39 us -- bound by L1 load issue, not by the 17 MB label map).  Neighbouring pixels of a row share
// their low-resolution cell: a thread that walks kStrip consecutive pixels keeps the cell's four
// corner vectors in registers, shifts them when the walk enters the next cell (new left column = old
// right column) and loads only the new right column -- a quarter of the loads.  The arithmetic is the
// per-pixel kernel's, expression by expression and window by window (ATen's upsample_bilinear2d
// order, then the division by the cover count), so the label maps are bit-identical
// (tests/test_inference_ohem_gpu.py).
constexpr int kStrip = 4;
constexpr int kStripCQ = 5;     // class quads in registers: ld <= 20 (19 Cityscapes classes)
int g_slide_strip = -1;         // -1: GS_SLIDE_STRIP (default 1); gs_debug_set_slide_strip

// state of the walk through one window: the four corner vectors (NQ class quads) of the current cell
template <int NQ>
struct StripCell {
  f32x4 p00[NQ], p01[NQ], p10[NQ], p11[NQ];
  int i0, i1;
};

template <int Q0, int NQ>
__device__ __forceinline__ void strip_pixel(StripCell<NQ>& cell, f32x4 (&acc)[NQ], bool live, int lx_,
                                            const SlideArgs& a, const Lerp& ly,
                                            const f32x4* __restrict__ r0, const f32x4* __restrict__ r1) {
  if (!live) return;
  const Lerp lx = lerp_coord(lx_, a.sw, a.d.wl, a.d.align_corners);
  if (lx.i0 != cell.i0 || lx.i1 != cell.i1) {
    if (lx.i0 == cell.i1 && cell.i0 >= 0) {   // next cell: the old right column is the new left one
#pragma unroll
      for (int q = 0; q < NQ; ++q) { cell.p00[q] = cell.p01[q]; cell.p10[q] = cell.p11[q]; }
    } else {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        cell.p00[q] = r0[lx.i0 * kStripCQ + Q0 + q];
        cell.p10[q] = r1[lx.i0 * kStripCQ + Q0 + q];
      }
    }
    if (lx.i1 == lx.i0) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) { cell.p01[q] = cell.p00[q]; cell.p11[q] = cell.p10[q]; }
    } else {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        cell.p01[q] = r0[lx.i1 * kStripCQ + Q0 + q];
        cell.p11[q] = r1[lx.i1 * kStripCQ + Q0 + q];
      }
    }
    cell.i0 = lx.i0; cell.i1 = lx.i1;
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q)
    acc[q] += ly.l0 * (lx.l0 * cell.p00[q] + lx.l1 * cell.p01[q]) +
              ly.l1 * (lx.l0 * cell.p10[q] + lx.l1 * cell.p11[q]);
}

// One pass over the covering windows for class quads [Q0, Q0 + NQ): the class range is split in two
// passes (12 + 8 classes) so that the corner vectors and the four pixels' sums of a pass fit in ~120
// registers -- four waves per SIMD instead of two; the kernel waits on dependent L1 / L2 loads, not on
// arithmetic.  Updates the running (max, arg max) of the strip's four pixels.
template <int Q0, int NQ>
__device__ __forceinline__ void strip_pass(const SlideArgs& a, const float* __restrict__ logits, int n,
                                           int sy, int sx0, const int (&cnt)[kStrip],
                                           float (&best)[kStrip], int (&amax)[kStrip]) {
  const gs_slide_desc& d = a.d;
  f32x4 acc0[NQ], acc1[NQ], acc2[NQ], acc3[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    acc0[q] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[q] = acc0[q]; acc2[q] = acc0[q]; acc3[q] = acc0[q];
  }
#pragma unroll 1
  for (int iy = 0; iy < d.ny; ++iy) {
    const int ly_ = sy - a.ys[iy];
    if (ly_ < 0 || ly_ >= d.hc) continue;
    const Lerp ly = lerp_coord(ly_, a.sh, d.hl, d.align_corners);
#pragma unroll 1
    for (int ix = 0; ix < d.nx; ++ix) {
      const int wx = a.xs[ix];
      if (sx0 + kStrip <= wx || sx0 >= wx + d.wc) continue;
      const float* base = logits + (((long)(iy * d.nx + ix) * d.N + n) * d.hl) * d.wl * d.ld;
      const f32x4* r0 = reinterpret_cast<const f32x4*>(base + (long)ly.i0 * d.wl * d.ld);
      const f32x4* r1 = reinterpret_cast<const f32x4*>(base + (long)ly.i1 * d.wl * d.ld);
      StripCell<NQ> cell;
      cell.i0 = cell.i1 = -1;
      const int l0 = sx0 - wx;
      auto live = [&](int p) {
        const int X = sx0 + p, lx_ = l0 + p;
        return X >= 0 && X < d.Wo && lx_ >= 0 && lx_ < d.wc;
      };
      strip_pixel<Q0, NQ>(cell, acc0, live(0), l0 + 0, a, ly, r0, r1);
      strip_pixel<Q0, NQ>(cell, acc1, live(1), l0 + 1, a, ly, r0, r1);
      strip_pixel<Q0, NQ>(cell, acc2, live(2), l0 + 2, a, ly, r0, r1);
      strip_pixel<Q0, NQ>(cell, acc3, live(3), l0 + 3, a, ly, r0, r1);
    }
  }
  auto upd = [&](const f32x4 (&acc)[NQ], int p) {
    const float c_ = (float)cnt[p];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      f32x4 v = acc[q];
      if (c_ > 1.f) v = v / c_;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = (Q0 + q) * 4 + e;
        if (c < d.C && v[e] > best[p]) { best[p] = v[e]; amax[p] = c; }
      }
    }
  };
  upd(acc0, 0); upd(acc1, 1); upd(acc2, 2); upd(acc3, 3);
}

// ld == 20 exactly (kStripCQ class quads): the Cityscapes heads of this path
__global__ __launch_bounds__(256) void slide_label_strip_kernel(const SlideArgs a,
                                                                const float* __restrict__ logits,
                                                                int64_t* __restrict__ labels) {
  const gs_slide_desc& d = a.d;
  const int spr = (d.Wo + kStrip - 1) / kStrip;           // strips per row
  // grid = (strips of a row / 256, output rows, images): no index divisions (the per-pixel kernel
  // spends four 64-bit divisions per pixel on them -- more instructions than its arithmetic)
  const int oy = blockIdx.y, n = blockIdx.z;
  for (int js = blockIdx.x * blockDim.x + threadIdx.x; js < spr; js += gridDim.x * blockDim.x) {
    const int ox0 = js * kStrip;
    // the view's pixels that land on this strip after the flip back: a horizontal flip mirrors the
    // strip, so it is walked through source columns sx0 .. sx0 + kStrip - 1 and written mirrored
    const int sy = d.flip == 2 ? d.Ho - 1 - oy : oy;
    const int sx0 = d.flip == 1 ? d.Wo - kStrip - ox0 : ox0;   // may be < 0 for the ragged last strip
    // cover counts: the window list is a product of row and column intervals
    int cy = 0;
    for (int iy = 0; iy < d.ny; ++iy) cy += (sy >= a.ys[iy] && sy < a.ys[iy] + d.hc) ? 1 : 0;
    int cnt[kStrip];
    float best[kStrip];
    int amax[kStrip];
#pragma unroll
    for (int p = 0; p < kStrip; ++p) {
      const int X = sx0 + p;
      int cx = 0;
      for (int ix = 0; ix < d.nx; ++ix) cx += (X >= a.xs[ix] && X < a.xs[ix] + d.wc) ? 1 : 0;
      cnt[p] = cy * cx;
      best[p] = -__builtin_huge_valf();
      amax[p] = 0;
    }
    strip_pass<0, 3>(a, logits, n, sy, sx0, cnt, best, amax);   // classes 0..11 first: the arg max
    strip_pass<3, 2>(a, logits, n, sy, sx0, cnt, best, amax);   // keeps the FIRST maximum, like the
#pragma unroll                                                   // per-pixel kernel's class loop
    for (int p = 0; p < kStrip; ++p) {
      const int X = sx0 + p;
      if (X < 0 || X >= d.Wo) continue;
      const int ox = d.flip == 1 ? d.Wo - 1 - X : X;
      labels[((long)n * d.Ho + oy) * d.Wo + ox] = amax[p];
    }
  }
}

}  // namespace gs

using namespace gs;

// Test / A-B hook: 1 = the strip kernel for label-only, un-rescaled calls (default), 0 = always the
// per-pixel kernel, -1 = back to GS_SLIDE_STRIP.
extern "C" int gs_debug_set_slide_strip(int32_t mode) {
  if (mode < -1 || mode > 1) return GS_E_BADARG;
  g_slide_strip = mode;
  return GS_OK;
}

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
  if (g_slide_strip < 0) {
    const char* e = getenv("GS_SLIDE_STRIP");
    g_slide_strip = (e && e[0] == '0') ? 0 : 1;
  }
  if (g_slide_strip && labels && !probs_in && !probs_out && d->Ho == d->H && d->Wo == d->W &&
      d->ld == 4 * kStripCQ && d->Ho <= 65535 && d->N <= 65535) {
    const int spr = (int)ceil_div(d->Wo, kStrip);
    const dim3 sgrid((unsigned)ceil_div(spr, 256), (unsigned)d->Ho, (unsigned)d->N);
    hipLaunchKernelGGL(slide_label_strip_kernel, sgrid, dim3(256), 0, st, a, logits, labels);
    return launch_status();
  }
  if (d->Ho == d->H && d->Wo == d->W)
    hipLaunchKernelGGL(slide_fuse_kernel<false>, dim3(grid), dim3(256), 0, st, a, logits, probs_in,
                       probs_out, labels);
  else
    hipLaunchKernelGGL(slide_fuse_kernel<true>, dim3(grid), dim3(256), 0, st, a, logits, probs_in,
                       probs_out, labels);
  return launch_status();
}
