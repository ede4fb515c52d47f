// Pixel-wise cross entropy with on-the-fly bilinear upsampling of the logits — gfx950, HBM-bound.
//
// Replaces, in one pass over the labels, the chain of
//   resize(seg_logit, size=label.shape[2:], 'bilinear')           (dynamic_fcn_head.py:141-145)
//   F.cross_entropy(..., reduction='none', ignore_index=255)      (losses/cross_entropy_loss.py:81-86)
//   weight_reduce_loss(mean over ALL pixels)                      (losses/utils.py:26-55)
//   accuracy(top-1)                                               (losses/accuracy.py:38-49)
// The [N,Cls,H,W] tensor (79.7 MB at 2x19x512x1024) is never written: each full-resolution pixel
// interpolates its Cls logits from the 4 neighbouring low-resolution pixels (L1/L2 resident).
// Backward is a gather: one workgroup per low-resolution logit pixel walks its bilinear footprint
// and reduces in a fixed order (bit-reproducible, no float atomics).
#include <algorithm>
#include "common.h"
#include "resize.h"

namespace gs {

struct CeArgs {
  gs_ce_desc d;
  float sh, sw;
};

// the four neighbour pointers + weights of full-resolution pixel (n, Y, X)
struct Taps {
  const float* p00; const float* p01; const float* p10; const float* p11;
  float w00, w01, w10, w11;
};
__device__ __forceinline__ Taps make_taps(const CeArgs& a, const float* logits, int n, int Y, int X) {
  const Lerp ly = lerp_coord(Y, a.sh, a.d.h, a.d.align_corners);
  const Lerp lx = lerp_coord(X, a.sw, a.d.w, a.d.align_corners);
  const float* b = logits + (long)n * a.d.l_sn;
  Taps t;
  t.p00 = b + ly.i0 * a.d.l_sh + lx.i0 * a.d.l_sw;
  t.p01 = b + ly.i0 * a.d.l_sh + lx.i1 * a.d.l_sw;
  t.p10 = b + ly.i1 * a.d.l_sh + lx.i0 * a.d.l_sw;
  t.p11 = b + ly.i1 * a.d.l_sh + lx.i1 * a.d.l_sw;
  t.w00 = lx.l0; t.w01 = lx.l1; t.w10 = ly.l0; t.w11 = ly.l1;  // (x weights, y weights)
  return t;
}
// same association order as ATen: h0*(w0*p00 + w1*p01) + h1*(w0*p10 + w1*p11)
__device__ __forceinline__ float tap_value(const Taps& t, long coff) {
  return t.w10 * (t.w00 * t.p00[coff] + t.w01 * t.p01[coff]) +
         t.w11 * (t.w00 * t.p10[coff] + t.w01 * t.p11[coff]);
}

// mode 0: loss/acc partial sums (+ optional lse)   mode 1: prob of the label (OHEM)
template <int MODE>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const CeArgs a, const float* __restrict__ logits,
                                                     const int64_t* __restrict__ labels,
                                                     const float* __restrict__ pw,
                                                     const float* __restrict__ cw,
                                                     float* __restrict__ lse_out,
                                                     double* __restrict__ part,
                                                     float* __restrict__ prob_out) {
  const long total = (long)a.d.N * a.d.H * a.d.W;
  double loss = 0.0, correct = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int X = (int)(i % a.d.W);
    const long r = i / a.d.W;
    const int Y = (int)(r % a.d.H);
    const int n = (int)(r / a.d.H);
    const Taps t = make_taps(a, logits, n, Y, X);
    const long lab = labels[i];
    const bool valid = lab != a.d.ignore_index && lab >= 0 && lab < a.d.Cls;
    float m = -__builtin_huge_valf(), s = 0.f, zl = 0.f;
    int amax = 0;
    for (int c = 0; c < a.d.Cls; ++c) {
      const float z = tap_value(t, (long)c * a.d.l_sc);
      if (c == lab) zl = z;
      if (z > m) {
        s = s * expf(m - z) + 1.f;
        m = z;
        amax = c;
      } else {
        s += expf(z - m);
      }
    }
    const float lse = m + logf(s);
    if (MODE == 0) {
      if (lse_out) lse_out[i] = lse;
      if (valid) {
        float l = lse - zl;
        if (cw) l *= cw[lab];
        if (pw) l *= pw[i];
        loss += (double)l;
      }
      if ((long)amax == lab) correct += 1.0;
    } else {
      prob_out[i] = valid ? expf(zl - lse) : 2.0f;
    }
  }
  if (MODE == 0) {
    __shared__ double sh[8];
    loss = wave_sum_d(loss);
    correct = wave_sum_d(correct);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sh[wave] = loss; sh[4 + wave] = correct; }
    __syncthreads();
    if (threadIdx.x == 0) {
      part[2 * blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
      part[2 * blockIdx.x + 1] = sh[4] + sh[5] + sh[6] + sh[7];
    }
  }
}

// one block: lanes stride over the partials, fixed-order wave + LDS combine (bit-reproducible)
// SCALED: out2 = {float(sum loss) * loss_scale, float(#correct) * acc_scale} as fp32 — the rounding
// order of `out.float() * scale` on the host side, which it replaces (five tiny launches per head)
template <bool SCALED>
__global__ __launch_bounds__(256) void ce_final_kernel(const double* __restrict__ part, int nparts,
                                                       double* out, float loss_scale,
                                                       float acc_scale, float* out2) {
  __shared__ double sh[8];
  double l = 0.0, c = 0.0;
  for (int p = threadIdx.x; p < nparts; p += 256) { l += part[2 * p]; c += part[2 * p + 1]; }
  l = wave_sum_d(l);
  c = wave_sum_d(c);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sh[wave] = l; sh[4 + wave] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double ls = sh[0] + sh[1] + sh[2] + sh[3], cs = sh[4] + sh[5] + sh[6] + sh[7];
    if (SCALED) {
      out2[0] = (float)ls * loss_scale;
      out2[1] = (float)cs * acc_scale;
    } else {
      out[0] = ls;
      out[1] = cs;
    }
  }
}

// One workgroup per low-resolution pixel (n, y, x); classes in chunks of CCH kept in registers.
// blockDim.x = 64, 128 or 256, about one full-resolution pixel of the support per thread
// (gs_ce_backward): at config 4's 193 -> 769 the support is ~8 x 8 pixels, and 256 threads per
// low-resolution pixel left three of four waves with nothing but the reductions (r03: 1988 us).
constexpr int CCH = 32;
__global__ __launch_bounds__(256) void ce_bwd_kernel(const CeArgs a, const float* __restrict__ logits,
                                                     const int64_t* __restrict__ labels,
                                                     const float* __restrict__ pw,
                                                     const float* __restrict__ cw,
                                                     const float* __restrict__ lse, float gscale,
                                                     float* __restrict__ dlogits, int ld_d) {
  __shared__ float sh[4][CCH];
  const int x = blockIdx.x % a.d.w;
  const int r = blockIdx.x / a.d.w;
  const int y = r % a.d.h;
  const int n = r / a.d.h;
  int ylo, yhi, xlo, xhi;
  dst_range(y, a.sh, a.d.H, ylo, yhi);
  dst_range(x, a.sw, a.d.W, xlo, xhi);
  const int nx = xhi - xlo + 1, npx = (yhi - ylo + 1) * nx;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nthr = blockDim.x, nwave = nthr >> 6;
  float* orow = dlogits + ((long)(n * a.d.h + y) * a.d.w + x) * ld_d;

  for (int c0 = 0; c0 < a.d.Cls; c0 += CCH) {
    const int nc = min(CCH, a.d.Cls - c0);
    float acc[CCH];
#pragma unroll
    for (int c = 0; c < CCH; ++c) acc[c] = 0.f;
    for (int q = threadIdx.x; q < npx; q += nthr) {
      const int Y = ylo + q / nx, X = xlo + q % nx;
      const Lerp ly = lerp_coord(Y, a.sh, a.d.h, a.d.align_corners);
      const Lerp lx = lerp_coord(X, a.sw, a.d.w, a.d.align_corners);
      const float wy = (ly.i0 == y ? ly.l0 : 0.f) + (ly.i1 == y ? ly.l1 : 0.f);
      const float wx = (lx.i0 == x ? lx.l0 : 0.f) + (lx.i1 == x ? lx.l1 : 0.f);
      const float wgt = wy * wx;
      if (wgt == 0.f) continue;
      const long pi = ((long)n * a.d.H + Y) * a.d.W + X;
      const long lab = labels[pi];
      const bool valid = lab != a.d.ignore_index && lab >= 0 && lab < a.d.Cls;
      if (!valid) continue;
      float coef = wgt * gscale;
      if (cw) coef *= cw[lab];
      if (pw) coef *= pw[pi];
      if (coef == 0.f) continue;
      const float l = lse[pi];
      const float* b = logits + (long)n * a.d.l_sn;
      Taps t;
      t.p00 = b + ly.i0 * a.d.l_sh + lx.i0 * a.d.l_sw;
      t.p01 = b + ly.i0 * a.d.l_sh + lx.i1 * a.d.l_sw;
      t.p10 = b + ly.i1 * a.d.l_sh + lx.i0 * a.d.l_sw;
      t.p11 = b + ly.i1 * a.d.l_sh + lx.i1 * a.d.l_sw;
      t.w00 = lx.l0; t.w01 = lx.l1; t.w10 = ly.l0; t.w11 = ly.l1;
#pragma unroll
      for (int c = 0; c < CCH; ++c) {
        if (c < nc) {
          const float z = tap_value(t, (long)(c0 + c) * a.d.l_sc);
          float p = expf(z - l);
          if (c0 + c == lab) p -= 1.f;
          acc[c] += coef * p;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CCH; ++c) {
      if (c < nc) {   // (block-uniform: 19 of the 32 slots carry a class at Cityscapes)
        const float v = wave_sum(acc[c]);
        if (lane == 0) sh[wave][c] = v;
      }
    }
    __syncthreads();
    if (threadIdx.x < CCH && threadIdx.x < nc) {
      float v = sh[0][threadIdx.x];               // wave order: fixed for a given block size
      for (int wv = 1; wv < nwave; ++wv) v += sh[wv][threadIdx.x];
      orow[c0 + threadIdx.x] = v;
    }
    __syncthreads();
  }
  // zero the padding columns Cls..ld_d-1
  for (int c = a.d.Cls + threadIdx.x; c < ld_d; c += nthr) orow[c] = 0.f;
}


// ---- tile form of the backward for power-of-two integer up-scaling (align_corners = false) ----
// The gather form above evaluates every (full-resolution pixel, class) softmax term once per
// neighbouring low-resolution pixel, i.e. four times.  For an integer scale s the full-resolution
// pixels Y with floor((2Y+1+s)/(2s)) == ty all interpolate between low-resolution rows ty-1 and ty
// (clamped at the borders), so one workgroup per (ty, tx) tile evaluates each term ONCE and splits
// it over its four corners; a second kernel adds, for every low-resolution pixel, the four corner
// sums of its four adjacent tiles in a fixed order (bit-reproducible, no atomics).  Power-of-two
// scales keep the fp32 source coordinates exact, so the tile membership is exact too.
constexpr int TCH = 20;   // classes per pass (19 Cityscapes classes + pad)
// blockDim.x = 64, 128 or 256: about four pixels of the tile per thread (gs_ce_backward_tiled), so
// that the 4 * TCH wave reductions at the end are amortised over enough softmax terms.
__global__ __launch_bounds__(256) void ce_bwd_tile_kernel(
    const CeArgs a, const float* __restrict__ logits, const int64_t* __restrict__ labels,
    const float* __restrict__ pw, const float* __restrict__ cw, const float* __restrict__ lse,
    float gscale, int sy, int sx, float* __restrict__ part, int cp) {
  __shared__ float sh[4][4 * TCH];
  __shared__ float4 corner[TCH];   // per class: logits of the tile's four low-resolution pixels
  const int nthr = blockDim.x, nwave = nthr >> 6;
  const int tw = a.d.w + 1, th = a.d.h + 1;
  const int tx = blockIdx.x % tw;
  const int r = blockIdx.x / tw;
  const int ty = r % th;
  const int n = r / th;
  const int y0 = max(0, ty * sy - (sy + 1) / 2), y1 = min(a.d.H, ty * sy - (sy + 1) / 2 + sy);
  const int x0 = max(0, tx * sx - (sx + 1) / 2), x1 = min(a.d.W, tx * sx - (sx + 1) / 2 + sx);
  const int nx = x1 - x0, npx = (y1 - y0) * nx;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* prow = part + (long)blockIdx.x * 4 * cp;
  // every pixel of the tile interpolates between the same (clamped) low-resolution rows / columns
  const Lerp uy = lerp_coord(min(y0, a.d.H - 1), a.sh, a.d.h, 0);
  const Lerp ux = lerp_coord(min(x0, a.d.W - 1), a.sw, a.d.w, 0);
  const float* ub = logits + (long)n * a.d.l_sn;
  const float* q00 = ub + uy.i0 * a.d.l_sh + ux.i0 * a.d.l_sw;
  const float* q01 = ub + uy.i0 * a.d.l_sh + ux.i1 * a.d.l_sw;
  const float* q10 = ub + uy.i1 * a.d.l_sh + ux.i0 * a.d.l_sw;
  const float* q11 = ub + uy.i1 * a.d.l_sh + ux.i1 * a.d.l_sw;
  for (int c0 = 0; c0 < a.d.Cls; c0 += TCH) {
    const int nc = min(TCH, a.d.Cls - c0);
    if (threadIdx.x < TCH) {   // classes past nc repeat the last one; their sums are not written
      const long off = (long)(c0 + min((int)threadIdx.x, nc - 1)) * a.d.l_sc;
      corner[threadIdx.x] = make_float4(q00[off], q01[off], q10[off], q11[off]);
    }
    __syncthreads();
    float a00[TCH], a01[TCH], a10[TCH], a11[TCH];   // corner (row slot, column slot) sums
#pragma unroll
    for (int c = 0; c < TCH; ++c) { a00[c] = 0.f; a01[c] = 0.f; a10[c] = 0.f; a11[c] = 0.f; }
    for (int q = threadIdx.x; q < npx; q += nthr) {
      const int Y = y0 + q / nx, X = x0 + q % nx;
      const long pi = ((long)n * a.d.H + Y) * a.d.W + X;
      const long lab = labels[pi];
      if (lab == a.d.ignore_index || lab < 0 || lab >= a.d.Cls) continue;
      float coef = gscale;
      if (cw) coef *= cw[lab];
      if (pw) coef *= pw[pi];
      if (coef == 0.f) continue;
      const Lerp ly = lerp_coord(Y, a.sh, a.d.h, 0);
      const Lerp lx = lerp_coord(X, a.sw, a.d.w, 0);
      // slot 1 = low-resolution index ty (tx), slot 0 = the one before it
      const float wy1 = (ly.i0 == ty ? ly.l0 : 0.f) + (ly.i1 == ty ? ly.l1 : 0.f);
      const float wy0 = (ly.i0 != ty ? ly.l0 : 0.f) + (ly.i1 != ty ? ly.l1 : 0.f);
      const float wx1 = (lx.i0 == tx ? lx.l0 : 0.f) + (lx.i1 == tx ? lx.l1 : 0.f);
      const float wx0 = (lx.i0 != tx ? lx.l0 : 0.f) + (lx.i1 != tx ? lx.l1 : 0.f);
      const float w00 = wy0 * wx0 * coef, w01 = wy0 * wx1 * coef, w10 = wy1 * wx0 * coef,
                  w11 = wy1 * wx1 * coef;
      const float l = lse[pi];
      const int labc = (int)lab - c0;
#pragma unroll
      for (int c = 0; c < TCH; ++c) {
        const float4 L = corner[c];
        // tap_value()'s association order, so that exp(z - lse) sums to one as in the forward
        const float z = ly.l0 * (lx.l0 * L.x + lx.l1 * L.y) + ly.l1 * (lx.l0 * L.z + lx.l1 * L.w);
        float p = expf(z - l);
        if (c == labc) p -= 1.f;
        a00[c] += w00 * p; a01[c] += w01 * p; a10[c] += w10 * p; a11[c] += w11 * p;
      }
    }
#pragma unroll
    for (int c = 0; c < TCH; ++c) {
      const float v0 = wave_sum_dpp(a00[c]), v1 = wave_sum_dpp(a01[c]), v2 = wave_sum_dpp(a10[c]),
                  v3 = wave_sum_dpp(a11[c]);
      if (lane == 0) {
        sh[wave][c] = v0; sh[wave][TCH + c] = v1; sh[wave][2 * TCH + c] = v2;
        sh[wave][3 * TCH + c] = v3;
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * TCH; i += nthr) {
      const int slot = i / TCH, c = i - slot * TCH;
      if (c < nc) {
        float v = sh[0][i];
        for (int w = 1; w < nwave; ++w) v += sh[w][i];
        prow[slot * cp + c0 + c] = v;
      }
    }
    __syncthreads();
  }
}

// ---- tile form for ANY scale (config 4: 193 -> 769, ratio 3.98; align_corners either way) ----
// Tile (ty, tx), ty in 0..h, tx in 0..w as above: the full-resolution pixels whose FIRST source row is
// ty - 1 and whose first source column is tx - 1 (lerp_coord's i0: a monotone function of the
// destination index, so a tile is a contiguous rectangle; its bounds are found with lerp_coord
// itself, so membership is exact for every scale).  All its pixels interpolate between rows
// (ty - 1, min(ty, h - 1)) and columns (tx - 1, min(tx, w - 1)); row / column 0 tiles are empty and
// write zeros (ce_bwd_gather_kernel reads them).  At ratio ~4 a tile is ~16 pixels, so ONE 16-lane
// DPP row owns a tile (one pixel per lane and pass), four tiles per wave, sixteen per workgroup;
// the corner sums are reduced inside the row with four row_shr steps (fixed order) and written by
// the row's last lane.  Every softmax term is evaluated once: 47 M terms at config 4 instead of the
// gather form's 4 x 47 M (r03: 854 us per head and step).
__device__ __forceinline__ int first_dst_with_i0_ge(int k, float scale, int in, int out, int align) {
  if (k <= 0) return 0;
  if (k > in - 1) return out;          // i0 is clamped to in - 1
  float est = align ? (scale > 0.f ? (float)k / scale : (float)out)
                    : ((float)k + 0.5f) / scale - 0.5f;
  int y = (int)floorf(est) - 1;
  y = y < 0 ? 0 : (y > out ? out : y);
  while (y < out && lerp_coord(y, scale, in, align).i0 < k) ++y;
  while (y > 0 && lerp_coord(y - 1, scale, in, align).i0 >= k) --y;
  return y;
}

// inclusive sum over the 16 lanes of a DPP row; lane 15 of the row holds the row's total
__device__ __forceinline__ float row16_sum_dpp(float v) {
  auto shr = [](float x, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x),
                                                                 decltype(ctrl)::value, 0xf, 0xf, true));
  };
  using std::integral_constant;
  v += shr(v, integral_constant<int, 0x111>{});   // row_shr:1
  v += shr(v, integral_constant<int, 0x112>{});   // row_shr:2
  v += shr(v, integral_constant<int, 0x114>{});   // row_shr:4
  v += shr(v, integral_constant<int, 0x118>{});   // row_shr:8
  return v;
}

// Registers decide this kernel's speed: it is a chain of two global round trips (corner logits, then the
// pixels' label / lse / weight) in front of ~2 us of arithmetic per wave, so what hides the latency is
// the number of resident waves.  With all 20 classes' accumulators (80) and the hoisted corner vectors
// (80) a wave held 195 registers = two waves per SIMD: 193 us at config 4.  The classes are therefore
// walked in two halves of kRowCls = 10 (40 + 40 registers, four waves per SIMD), all corner vectors of
// the tile staged in LDS once, and the first pixel's label / lse / weight loads are issued BEFORE the
// barrier that publishes the corners, beside the corner loads instead of behind them.
// (kRowCls: classes per register pass; GS_CE_ROWCLS = 10 / 5 / 4 selects 155 / 105 / 95 registers)
// Four values per lane summed over the 16 lanes of a DPP row at once: two reduce-scatter steps inside
// the quads (xor 1, xor 2: a lane keeps half of its values and adds its partner's copy of them), then the
// four quads are added with two row_shr steps.  Lanes 12..15 of the row end up holding one total each:
// lane 12 -> v0, 13 -> v2, 14 -> v1, 15 -> v3 (slot = ((l16 & 1) << 1) | ((l16 >> 1) & 1)).  11 VALU
// operations and one store per class instead of 16 and four; fixed order.
__device__ __forceinline__ float row16_sum4_dpp(float v0, float v1, float v2, float v3, int l16) {
  auto dpp = [](float x, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x),
                                                                 decltype(ctrl)::value, 0xf, 0xf, true));
  };
  using std::integral_constant;
  const bool b0 = (l16 & 1) != 0, b1 = (l16 & 2) != 0;
  float k0 = b0 ? v2 : v0, g0 = b0 ? v0 : v2;
  float k1 = b0 ? v3 : v1, g1 = b0 ? v1 : v3;
  k0 += dpp(g0, integral_constant<int, 0xB1>{});   // quad_perm:[1,0,3,2]
  k1 += dpp(g1, integral_constant<int, 0xB1>{});
  float k = b1 ? k1 : k0;
  const float g = b1 ? k0 : k1;
  k += dpp(g, integral_constant<int, 0x4E>{});     // quad_perm:[2,3,0,1]
  k += dpp(k, integral_constant<int, 0x114>{});    // row_shr:4
  k += dpp(k, integral_constant<int, 0x118>{});    // row_shr:8
  return k;
}

template <int kRowCls>
__global__ __launch_bounds__(256) void ce_bwd_rowtile_kernel(
    const CeArgs a, const float* __restrict__ logits, const int64_t* __restrict__ labels,
    const float* __restrict__ pw, const float* __restrict__ cw, const float* __restrict__ lse,
    float gscale, long ntiles, float* __restrict__ part, int cp) {
  __shared__ float4 corner[16][TCH];   // per row group and class: the tile's four corner logits
  const int grp = threadIdx.x >> 4, l16 = threadIdx.x & 15;
  const long tile = (long)blockIdx.x * 16 + grp;
  const bool live = tile < ntiles;
  const int tw = a.d.w + 1, th = a.d.h + 1;
  const int tx = live ? (int)(tile % tw) : 0;
  const long r = live ? tile / tw : 0;
  const int ty = (int)(r % th);
  const int n = (int)(r / th);
  const int al = a.d.align_corners;
  int y0 = 0, y1 = 0, x0 = 0, x1 = 0;
  if (live && ty > 0 && tx > 0) {
    y0 = first_dst_with_i0_ge(ty - 1, a.sh, a.d.h, a.d.H, al);
    y1 = first_dst_with_i0_ge(ty, a.sh, a.d.h, a.d.H, al);
    x0 = first_dst_with_i0_ge(tx - 1, a.sw, a.d.w, a.d.W, al);
    x1 = first_dst_with_i0_ge(tx, a.sw, a.d.w, a.d.W, al);
  }
  const int nx = x1 - x0, npx = (y1 - y0) * nx;
  const int r0 = max(ty - 1, 0), r1 = min(ty, a.d.h - 1), c0i = max(tx - 1, 0), c1i = min(tx, a.d.w - 1);
  const float* ub = logits + (long)n * a.d.l_sn;
  const float* q00 = ub + (long)r0 * a.d.l_sh + (long)c0i * a.d.l_sw;
  const float* q01 = ub + (long)r0 * a.d.l_sh + (long)c1i * a.d.l_sw;
  const float* q10 = ub + (long)r1 * a.d.l_sh + (long)c0i * a.d.l_sw;
  const float* q11 = ub + (long)r1 * a.d.l_sh + (long)c1i * a.d.l_sw;
  float* prow = part + tile * 4 * cp;
  // A pixel's state: interpolation weights, the four corner weights (x its loss coefficient), lse, label.
  // Set up ONCE per pixel, outside the class passes (the lane's first pixel is the only one at ratios up
  // to 4; its label / lse / weight loads are issued here, beside the corner logits).
  struct Px {
    bool ok;
    float yl0, yl1, xl0, xl1, w00, w01, w10, w11, l;
    int lab;
  };
  auto setup = [&](int q) -> Px {
    Px p{};
    p.ok = false;
    if (q >= npx) return p;
    const int Y = y0 + q / nx, X = x0 + q % nx;
    const long pi = ((long)n * a.d.H + Y) * a.d.W + X;
    const long lab = labels[pi];
    if (lab == a.d.ignore_index || lab < 0 || lab >= a.d.Cls) return p;
    float coef = gscale;
    if (cw) coef *= cw[lab];
    if (pw) coef *= pw[pi];
    if (coef == 0.f) return p;
    const Lerp ly = lerp_coord(Y, a.sh, a.d.h, al);
    const Lerp lx = lerp_coord(X, a.sw, a.d.w, al);
    // slot 1 = low-resolution index ty (tx), slot 0 = the one before it (both weights go there at
    // the far border, where i1 == i0 == ty - 1)
    const float wy1 = (ly.i0 == ty ? ly.l0 : 0.f) + (ly.i1 == ty ? ly.l1 : 0.f);
    const float wy0 = (ly.i0 != ty ? ly.l0 : 0.f) + (ly.i1 != ty ? ly.l1 : 0.f);
    const float wx1 = (lx.i0 == tx ? lx.l0 : 0.f) + (lx.i1 == tx ? lx.l1 : 0.f);
    const float wx0 = (lx.i0 != tx ? lx.l0 : 0.f) + (lx.i1 != tx ? lx.l1 : 0.f);
    p.yl0 = ly.l0; p.yl1 = ly.l1; p.xl0 = lx.l0; p.xl1 = lx.l1;
    p.w00 = wy0 * wx0 * coef; p.w01 = wy0 * wx1 * coef; p.w10 = wy1 * wx0 * coef; p.w11 = wy1 * wx1 * coef;
    p.l = lse[pi];
    p.lab = (int)lab;
    p.ok = true;
    return p;
  };
  const Px p0 = setup(l16);
  for (int c0 = 0; c0 < a.d.Cls; c0 += TCH) {
    const int nc = min(TCH, a.d.Cls - c0);
    if (c0 > 0) __syncthreads();
    for (int c = l16; c < TCH; c += 16) {   // classes past nc repeat the last one; not written
      const long off = (long)(c0 + min(c, nc - 1)) * a.d.l_sc;
      corner[grp][c] = npx > 0 ? make_float4(q00[off], q01[off], q10[off], q11[off])
                               : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
#pragma unroll 1
    for (int hb = 0; hb < TCH; hb += kRowCls) {
      float a00[kRowCls], a01[kRowCls], a10[kRowCls], a11[kRowCls];
#pragma unroll
      for (int c = 0; c < kRowCls; ++c) { a00[c] = 0.f; a01[c] = 0.f; a10[c] = 0.f; a11[c] = 0.f; }
      auto accumulate = [&](const Px& p) {
        const int labc = p.lab - c0 - hb;
#pragma unroll
        for (int c = 0; c < kRowCls; ++c) {
          const float4 L = corner[grp][hb + c];
          // tap_value()'s association order, so that exp(z - lse) sums to one as in the forward
          const float z = p.yl0 * (p.xl0 * L.x + p.xl1 * L.y) + p.yl1 * (p.xl0 * L.z + p.xl1 * L.w);
          float e = expf(z - p.l);
          if (c == labc) e -= 1.f;
          a00[c] += p.w00 * e; a01[c] += p.w01 * e; a10[c] += p.w10 * e; a11[c] += p.w11 * e;
        }
      };
      if (p0.ok) accumulate(p0);
      for (int q = l16 + 16; q < npx; q += 16) {
        const Px p = setup(q);
        if (p.ok) accumulate(p);
      }
      const int slot = ((l16 & 1) << 1) | ((l16 >> 1) & 1);   // which corner sum lanes 12..15 end up with
#pragma unroll
      for (int c = 0; c < kRowCls; ++c) {
        const float v = row16_sum4_dpp(a00[c], a01[c], a10[c], a11[c], l16);
        if (live && l16 >= 12 && hb + c < nc) prow[slot * cp + c0 + hb + c] = v;
      }
    }
  }
}

// dlogits[n, y, x, c] = corner sums of the four tiles around low-resolution pixel (y, x)
__global__ __launch_bounds__(256) void ce_bwd_gather_kernel(const float* __restrict__ part, int N,
                                                            int h, int w, int Cls, int cp,
                                                            float* __restrict__ dlogits, int ld_d) {
  const long total = (long)N * h * w * ld_d;
  const int tw = w + 1, th = h + 1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % ld_d);
    const long px = i / ld_d;
    float v = 0.f;
    if (c < Cls) {
      const int x = (int)(px % w);
      const long r = px / w;
      const int y = (int)(r % h), n = (int)(r / h);
      const long t11 = ((long)n * th + y) * tw + x;   // tile (y, x): this pixel is its slot (1, 1)
      v = part[((t11 + tw + 1) * 4 + 0) * cp + c];    // tile (y+1, x+1), slot (0, 0)
      v += part[((t11 + tw) * 4 + 1) * cp + c];       // tile (y+1, x  ), slot (0, 1)
      v += part[((t11 + 1) * 4 + 2) * cp + c];        // tile (y,   x+1), slot (1, 0)
      v += part[(t11 * 4 + 3) * cp + c];              // tile (y,   x  ), slot (1, 1)
    }
    dlogits[i] = v;
  }
}

// argmax (and optional softmax probabilities) of the resized logits
__global__ __launch_bounds__(256) void resize_argmax_kernel(const CeArgs a,
                                                            const float* __restrict__ logits,
                                                            int64_t* __restrict__ seg,
                                                            float* __restrict__ probs) {
  const long total = (long)a.d.N * a.d.H * a.d.W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int X = (int)(i % a.d.W);
    const long r = i / a.d.W;
    const int Y = (int)(r % a.d.H);
    const int n = (int)(r / a.d.H);
    const Taps t = make_taps(a, logits, n, Y, X);
    float m = -__builtin_huge_valf(), s = 0.f;
    int amax = 0;
    for (int c = 0; c < a.d.Cls; ++c) {
      const float z = tap_value(t, (long)c * a.d.l_sc);
      if (z > m) { s = s * expf(m - z) + 1.f; m = z; amax = c; }
      else s += expf(z - m);
    }
    if (seg) seg[i] = amax;
    if (probs) {
      const float inv = 1.f / s;
      for (int c = 0; c < a.d.Cls; ++c)
        probs[i * a.d.Cls + c] = expf(tap_value(t, (long)c * a.d.l_sc) - m) * inv;
    }
  }
}

static int check_ce(const gs_ce_desc* d, CeArgs& a) {
  if (!d) return GS_E_NULL;
  if (d->N <= 0 || d->h <= 0 || d->w <= 0 || d->Cls <= 0 || d->H <= 0 || d->W <= 0)
    return GS_E_BADARG;
  a.d = *d;
  a.sh = resize_scale(d->h, d->H, d->align_corners);
  a.sw = resize_scale(d->w, d->W, d->align_corners);
  return GS_OK;
}
static int ce_grid(const gs_ce_desc* d) {
  return stream_grid((long)d->N * d->H * d->W, 256);
}

}  // namespace gs

using namespace gs;

extern "C" size_t gs_ce_workspace_bytes(const gs_ce_desc* d) {
  CeArgs a;
  if (check_ce(d, a)) return 0;
  return (size_t)ce_grid(d) * 2 * sizeof(double);
}

extern "C" int gs_ce_forward(const gs_ce_desc* d, const float* logits, const int64_t* labels,
                             const float* pixel_weight, const float* class_weight, float* lse,
                             double* out, void* workspace, size_t workspace_bytes, void* stream) {
  CeArgs a;
  int rc = check_ce(d, a);
  if (rc) return rc;
  if (!logits || !labels || !out || !workspace) return GS_E_NULL;
  const int grid = ce_grid(d);
  if ((size_t)grid * 2 * sizeof(double) > workspace_bytes) return GS_E_WORKSPACE;
  if (reinterpret_cast<uintptr_t>(workspace) & 7) return GS_E_ALIGN;
  hipStream_t st = as_stream(stream);
  double* part = static_cast<double*>(workspace);
  hipLaunchKernelGGL(ce_fwd_kernel<0>, dim3(grid), dim3(256), 0, st, a, logits, labels,
                     pixel_weight, class_weight, lse, part, (float*)nullptr);
  hipLaunchKernelGGL(ce_final_kernel<false>, dim3(1), dim3(256), 0, st, part, grid, out, 0.f, 0.f,
                     (float*)nullptr);
  return launch_status();
}

extern "C" int gs_ce_forward_scaled(const gs_ce_desc* d, const float* logits, const int64_t* labels,
                                    const float* pixel_weight, const float* class_weight,
                                    float* lse, float loss_scale, float acc_scale, float* out2,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  CeArgs a;
  int rc = check_ce(d, a);
  if (rc) return rc;
  if (!logits || !labels || !out2 || !workspace) return GS_E_NULL;
  const int grid = ce_grid(d);
  if ((size_t)grid * 2 * sizeof(double) > workspace_bytes) return GS_E_WORKSPACE;
  if (reinterpret_cast<uintptr_t>(workspace) & 7) return GS_E_ALIGN;
  hipStream_t st = as_stream(stream);
  double* part = static_cast<double*>(workspace);
  hipLaunchKernelGGL(ce_fwd_kernel<0>, dim3(grid), dim3(256), 0, st, a, logits, labels,
                     pixel_weight, class_weight, lse, part, (float*)nullptr);
  hipLaunchKernelGGL(ce_final_kernel<true>, dim3(1), dim3(256), 0, st, part, grid,
                     (double*)nullptr, loss_scale, acc_scale, out2);
  return launch_status();
}

extern "C" int gs_ce_backward(const gs_ce_desc* d, const float* logits, const int64_t* labels,
                              const float* pixel_weight, const float* class_weight,
                              const float* lse, float grad_scale, float* dlogits, int32_t ld_d,
                              void* stream) {
  CeArgs a;
  int rc = check_ce(d, a);
  if (rc) return rc;
  if (!logits || !labels || !lse || !dlogits) return GS_E_NULL;
  if (ld_d < d->Cls) return GS_E_BADARG;
  // support of a low-resolution pixel: ~(2 H/h) x (2 W/w) full-resolution pixels
  const long sup = (2L * d->H / d->h + 1) * (2L * d->W / d->w + 1);
  const int threads = sup > 128 ? 256 : sup > 64 ? 128 : 64;
  hipLaunchKernelGGL(ce_bwd_kernel, dim3(d->N * d->h * d->w), dim3(threads), 0, as_stream(stream), a,
                     logits, labels, pixel_weight, class_weight, lse, grad_scale, dlogits, ld_d);
  return launch_status();
}

static bool ce_tile_scales(const gs_ce_desc* d, int& sy, int& sx) {
  if (d->align_corners || d->H % d->h || d->W % d->w) return false;
  sy = d->H / d->h; sx = d->W / d->w;
  auto pow2 = [](int v) { return v >= 2 && (v & (v - 1)) == 0; };
  return pow2(sy) && pow2(sx);
}

// the row-tile form takes any up-scaling whose tiles are small enough for one 16-lane row each
// (<= 64 pixels on average: four passes); larger non-power-of-two ratios keep the gather form
static bool ce_rowtile_ok(const gs_ce_desc* d) {
  if (d->H < d->h || d->W < d->w || d->h < 1 || d->w < 1) return false;
  return (double)d->H * d->W <= 64.0 * (double)d->h * d->w;
}

extern "C" size_t gs_ce_backward_workspace_bytes(const gs_ce_desc* d, int32_t ld_d) {
  int sy, sx;
  if (!d || !(ce_tile_scales(d, sy, sx) || ce_rowtile_ok(d))) return 0;
  return (size_t)d->N * (d->h + 1) * (d->w + 1) * 4 * ld_d * sizeof(float);
}

// As gs_ce_backward; with a workspace of gs_ce_backward_workspace_bytes() and a power-of-two
// integer up-scaling the tile form is used (each softmax term evaluated once instead of four times).
extern "C" int gs_ce_backward_ws(const gs_ce_desc* d, const float* logits, const int64_t* labels,
                                 const float* pixel_weight, const float* class_weight,
                                 const float* lse, float grad_scale, float* dlogits, int32_t ld_d,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  CeArgs a;
  int rc = check_ce(d, a);
  if (rc) return rc;
  if (!logits || !labels || !lse || !dlogits) return GS_E_NULL;
  if (ld_d < d->Cls) return GS_E_BADARG;
  int sy = 0, sx = 0;
  const size_t need = gs_ce_backward_workspace_bytes(d, ld_d);
  static const bool no_tile = getenv("GS_CE_NO_TILE") != nullptr;
  static const bool no_rowtile = getenv("GS_CE_NO_ROWTILE") != nullptr;
  const bool pow2 = ce_tile_scales(d, sy, sx);
  const bool rowtile = !pow2 && !no_rowtile && ce_rowtile_ok(d);
  if (no_tile || !(pow2 || rowtile) || !workspace || workspace_bytes < need || need == 0)
    return gs_ce_backward(d, logits, labels, pixel_weight, class_weight, lse, grad_scale, dlogits,
                          ld_d, stream);
  hipStream_t st = as_stream(stream);
  float* part = static_cast<float*>(workspace);
  const int tiles = d->N * (d->h + 1) * (d->w + 1);
  if (rowtile) {
    const long ntiles = tiles;
    static const int rowcls = [] { const char* v = getenv("GS_CE_ROWCLS"); return v && *v ? atoi(v) : 5; }();
    const dim3 rgrid((unsigned)((ntiles + 15) / 16));
    if (rowcls == 10)
      hipLaunchKernelGGL(ce_bwd_rowtile_kernel<10>, rgrid, dim3(256), 0, st, a, logits, labels,
                         pixel_weight, class_weight, lse, grad_scale, ntiles, part, ld_d);
    else if (rowcls == 4)
      hipLaunchKernelGGL(ce_bwd_rowtile_kernel<4>, rgrid, dim3(256), 0, st, a, logits, labels,
                         pixel_weight, class_weight, lse, grad_scale, ntiles, part, ld_d);
    else
      hipLaunchKernelGGL(ce_bwd_rowtile_kernel<5>, rgrid, dim3(256), 0, st, a, logits, labels,
                         pixel_weight, class_weight, lse, grad_scale, ntiles, part, ld_d);
    const long total_r = (long)d->N * d->h * d->w * ld_d;
    hipLaunchKernelGGL(ce_bwd_gather_kernel, dim3(stream_grid(total_r, 256)), dim3(256), 0, st, part,
                       d->N, d->h, d->w, d->Cls, ld_d, dlogits, ld_d);
    return launch_status();
  }
  const int tile_px = sy * sx;
  const int threads = tile_px >= 1024 ? 256 : tile_px >= 512 ? 128 : 64;
  hipLaunchKernelGGL(ce_bwd_tile_kernel, dim3(tiles), dim3(threads), 0, st, a, logits, labels,
                     pixel_weight, class_weight, lse, grad_scale, sy, sx, part, ld_d);
  const long total = (long)d->N * d->h * d->w * ld_d;
  hipLaunchKernelGGL(ce_bwd_gather_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, st, part,
                     d->N, d->h, d->w, d->Cls, ld_d, dlogits, ld_d);
  return launch_status();
}

extern "C" int gs_ce_label_prob(const gs_ce_desc* d, const float* logits, const int64_t* labels,
                                float* prob, void* stream) {
  CeArgs a;
  int rc = check_ce(d, a);
  if (rc) return rc;
  if (!logits || !labels || !prob) return GS_E_NULL;
  hipLaunchKernelGGL(ce_fwd_kernel<1>, dim3(ce_grid(d)), dim3(256), 0, as_stream(stream), a, logits,
                     labels, (const float*)nullptr, (const float*)nullptr, (float*)nullptr,
                     (double*)nullptr, prob);
  return launch_status();
}

extern "C" int gs_resize_argmax(const gs_ce_desc* d, const float* logits, int64_t* seg,
                                float* probs, void* stream) {
  CeArgs a;
  int rc = check_ce(d, a);
  if (rc) return rc;
  if (!logits || (!seg && !probs)) return GS_E_NULL;
  hipLaunchKernelGGL(resize_argmax_kernel, dim3(ce_grid(d)), dim3(256), 0, as_stream(stream), a,
                     logits, seg, probs);
  return launch_status();
}
