// HBM-bound NHWC pixel operators of the supernet hot path on gfx950: max / adaptive-average
// pooling, bilinear resize (+accumulate = FPN top-down add, +channel-slice output = concat
// fusion), strided copies, Dropout2d scaling and the fused SGD step.
// Every kernel moves 16 B per lane along the channel dimension (coalesced float4) and the
// backward kernels are written in gather form, so results are bit-reproducible (no atomics).
#include <algorithm>
#include "common.h"
#include "resize.h"

namespace gs {

// ------------------------------------------------------------------------------------------
// nn.MaxPool2d  (gaiaseg/models/backbones/dynamic_resnet.py:302,413)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, int N, int H,
                                                          int W, int C4, int ldx, int k, int s,
                                                          int p, int Ho, int Wo,
                                                          float* __restrict__ y, int ldy,
                                                          uint8_t* __restrict__ idx) {
  const long total = (long)N * Ho * Wo * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int cq = (int)(i % C4);
    long pix = i / C4;
    const int wo = (int)(pix % Wo);
    pix /= Wo;
    const int ho = (int)(pix % Ho);
    const int n = (int)(pix / Ho);
    const float ninf = -__builtin_huge_valf();
    f32x4 best{ninf, ninf, ninf, ninf};
    int bi[4] = {-1, -1, -1, -1};
    for (int kh = 0; kh < k; ++kh) {
      const int h = ho * s - p + kh;
      if ((unsigned)h >= (unsigned)H) continue;
      for (int kw = 0; kw < k; ++kw) {
        const int w = wo * s - p + kw;
        if ((unsigned)w >= (unsigned)W) continue;
        const f32x4 v =
            *reinterpret_cast<const f32x4*>(x + ((long)(n * H + h) * W + w) * ldx + cq * 4);
        const int tap = kh * k + kw;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // ATen scan order: first maximum wins, NaN propagates; the first in-bounds tap seeds idx
          if (bi[e] < 0 || v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; bi[e] = tap; }
        }
      }
    }
    const long o = ((long)(n * Ho + ho) * Wo + wo);
    *reinterpret_cast<f32x4*>(y + o * ldy + cq * 4) = best;
    uchar4 b;
    b.x = (uint8_t)bi[0]; b.y = (uint8_t)bi[1]; b.z = (uint8_t)bi[2]; b.w = (uint8_t)bi[3];
    *reinterpret_cast<uchar4*>(idx + (o * C4 + cq) * 4) = b;
  }
}

template <bool ACC>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, int ld_dy,
                                                          const uint8_t* __restrict__ idx, int N,
                                                          int H, int W, int C4, int k, int s, int p,
                                                          int Ho, int Wo, float* dx, int ld_dx) {
  const long total = (long)N * H * W * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int cq = (int)(i % C4);
    long pix = i / C4;
    const int w = (int)(pix % W);
    pix /= W;
    const int h = (int)(pix % H);
    const int n = (int)(pix / H);
    f32x4 g{0.f, 0.f, 0.f, 0.f};
    // windows containing (h, w): ho*s - p <= h <= ho*s - p + k - 1
    int ho_lo = (h + p - k + 1 + s - 1);
    ho_lo = ho_lo <= 0 ? 0 : ho_lo / s;
    int wo_lo = (w + p - k + 1 + s - 1);
    wo_lo = wo_lo <= 0 ? 0 : wo_lo / s;
    const int ho_hi = min((h + p) / s, Ho - 1), wo_hi = min((w + p) / s, Wo - 1);
    for (int ho = ho_lo; ho <= ho_hi; ++ho) {
      const int kh = h - (ho * s - p);
      for (int wo = wo_lo; wo <= wo_hi; ++wo) {
        const int kw = w - (wo * s - p);
        const int tap = kh * k + kw;
        const long o = ((long)(n * Ho + ho) * Wo + wo);
        const uchar4 b = *reinterpret_cast<const uchar4*>(idx + (o * C4 + cq) * 4);
        const f32x4 d = *reinterpret_cast<const f32x4*>(dy + o * ld_dy + cq * 4);
        if (b.x == tap) g[0] += d[0];
        if (b.y == tap) g[1] += d[1];
        if (b.z == tap) g[2] += d[2];
        if (b.w == tap) g[3] += d[3];
      }
    }
    float* o = dx + ((long)(n * H + h) * W + w) * ld_dx + cq * 4;
    if (ACC) g += *reinterpret_cast<const f32x4*>(o);
    *reinterpret_cast<f32x4*>(o) = g;
  }
}

// ------------------------------------------------------------------------------------------
// nn.AdaptiveAvgPool2d for all PPM scales (gaiaseg/models/decode_heads/dynamic_psp_head.py:48-51)
// ------------------------------------------------------------------------------------------
constexpr int kMaxScales = 8;
struct Scales {
  int n;
  int s[kMaxScales];
  int off[kMaxScales + 1];  // bin offsets: off[i] = sum_{j<i} s[j]^2
};

__device__ __forceinline__ int bin_lo(int i, int L, int s) { return (i * L) / s; }
__device__ __forceinline__ int bin_hi(int i, int L, int s) { return ((i + 1) * L + s - 1) / s; }

// grid: (N * total_bins, column blocks, pixel splits); part[z][n][bin][C] = partial SUM
__global__ __launch_bounds__(256) void avgpool_partial_kernel(const float* __restrict__ x, int H,
                                                              int W, int C, int ldx, Scales sc,
                                                              int nsplit, float* __restrict__ part,
                                                              long part_stride) {
  const int C4 = C >> 2;
  const int tb = sc.off[sc.n];
  const int n = blockIdx.x / tb, bin = blockIdx.x - n * tb;
  int si = 0;
  while (bin >= sc.off[si + 1]) ++si;
  const int s = sc.s[si], b = bin - sc.off[si];
  const int bi = b / s, bj = b - bi * s;
  const int h0 = bin_lo(bi, H, s), h1 = bin_hi(bi, H, s);
  const int w0 = bin_lo(bj, W, s), w1 = bin_hi(bj, W, s);
  const int bw = w1 - w0, npx = (h1 - h0) * bw;
  // this thread: channel quad cq, pixel lane rr of rpi
  int rpi, rr, cq;
  bool active;
  if (C4 <= 256) {
    rpi = 256 / C4; rr = threadIdx.x / C4; cq = threadIdx.x - rr * C4; active = rr < rpi;
  } else {
    rpi = 1; rr = 0; cq = blockIdx.y * 256 + threadIdx.x; active = cq < C4;
  }
  __shared__ f32x4 sh[256];
  f32x4 acc{0.f, 0.f, 0.f, 0.f};
  if (active) {
    for (int q = blockIdx.z * rpi + rr; q < npx; q += nsplit * rpi) {
      const int h = h0 + q / bw, w = w0 + q % bw;
      acc += *reinterpret_cast<const f32x4*>(x + ((long)(n * H + h) * W + w) * ldx + cq * 4);
    }
  }
  if (rpi > 1) {
    sh[threadIdx.x] = acc;
    __syncthreads();
    if (active && rr == 0)
      for (int r = 1; r < rpi; ++r) acc += sh[r * C4 + cq];
  }
  if (active && rr == 0)
    *reinterpret_cast<f32x4*>(part + blockIdx.z * part_stride + ((long)n * tb + bin) * C + cq * 4) =
        acc;
}

// y[n][bin][C] = sum_z part / area
__global__ __launch_bounds__(256) void avgpool_final_kernel(const float* __restrict__ part,
                                                            long part_stride, int nsplit, int N,
                                                            int H, int W, int C, Scales sc,
                                                            float* __restrict__ y) {
  const int C4 = C >> 2, tb = sc.off[sc.n];
  const long total = (long)N * tb * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int cq = (int)(i % C4);
    const long nb = i / C4;
    const int bin = (int)(nb % tb);
    int si = 0;
    while (bin >= sc.off[si + 1]) ++si;
    const int s = sc.s[si], b = bin - sc.off[si];
    const int bi = b / s, bj = b - bi * s;
    const float area = (float)((bin_hi(bi, H, s) - bin_lo(bi, H, s)) *
                               (bin_hi(bj, W, s) - bin_lo(bj, W, s)));
    f32x4 v = *reinterpret_cast<const f32x4*>(part + nb * C + cq * 4);
    for (int z = 1; z < nsplit; ++z)
      v += *reinterpret_cast<const f32x4*>(part + z * part_stride + nb * C + cq * 4);
    // layout of y: scale blocks one after another, each [N][s][s][C]
    const int n = (int)(nb / tb);
    const long yo = (long)N * sc.off[si] * C + ((long)n * s * s + b) * C + cq * 4;
    *reinterpret_cast<f32x4*>(y + yo) = v / area;
  }
}

template <bool ACC>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dy, int N, int H,
                                                          int W, int C, Scales sc, float* dx,
                                                          int ld_dx) {
  const int C4 = C >> 2;
  const long total = (long)N * H * W * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int cq = (int)(i % C4);
    long pix = i / C4;
    const int w = (int)(pix % W);
    pix /= W;
    const int h = (int)(pix % H);
    const int n = (int)(pix / H);
    f32x4 g{0.f, 0.f, 0.f, 0.f};
    for (int si = 0; si < sc.n; ++si) {
      const int s = sc.s[si];
      const float* base = dy + (long)N * sc.off[si] * C + (long)n * s * s * C + cq * 4;
      // bins that can contain h: floor(h*s/H)-1 .. ceil((h+1)*s/H)  (several when s > H)
      const int ilo = max((h * s) / H - 1, 0), ihi = min(((h + 1) * s + H - 1) / H, s - 1);
      const int jlo = max((w * s) / W - 1, 0), jhi = min(((w + 1) * s + W - 1) / W, s - 1);
      for (int bi = ilo; bi <= ihi; ++bi) {
        const int h0 = bin_lo(bi, H, s), h1 = bin_hi(bi, H, s);
        if (h < h0 || h >= h1) continue;
        for (int bj = jlo; bj <= jhi; ++bj) {
          const int w0 = bin_lo(bj, W, s), w1 = bin_hi(bj, W, s);
          if (w < w0 || w >= w1) continue;
          const float inv = 1.f / (float)((h1 - h0) * (w1 - w0));
          g += *reinterpret_cast<const f32x4*>(base + (long)(bi * s + bj) * C) * inv;
        }
      }
    }
    float* o = dx + ((long)(n * H + h) * W + w) * ld_dx + cq * 4;
    if (ACC) g += *reinterpret_cast<const f32x4*>(o);
    *reinterpret_cast<f32x4*>(o) = g;
  }
}

// ------------------------------------------------------------------------------------------
// bilinear resize == F.interpolate(mode='bilinear')  (mmseg.ops.resize; dynamic_psp_head.py:67-71,
// dynamic_uper_head.py:108-112,123-127).  Index arithmetic follows ATen's
// area_pixel_compute_source_index in fp32 so that weights match the CPU path bit for bit.
// ------------------------------------------------------------------------------------------
template <bool ACC>
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const float* __restrict__ x, int N, int Hi,
                                                           int Wi, int C4, int ldx, int Ho, int Wo,
                                                           int align, float sh, float sw, float* y,
                                                           int ldy) {
  const long total = (long)N * Ho * Wo * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int cq = (int)(i % C4);
    long pix = i / C4;
    const int X = (int)(pix % Wo);
    pix /= Wo;
    const int Y = (int)(pix % Ho);
    const int n = (int)(pix / Ho);
    const Lerp ly = lerp_coord(Y, sh, Hi, align), lx = lerp_coord(X, sw, Wi, align);
    const float* b = x + (long)n * Hi * Wi * ldx + cq * 4;
    const f32x4 p00 = *reinterpret_cast<const f32x4*>(b + ((long)ly.i0 * Wi + lx.i0) * ldx);
    const f32x4 p01 = *reinterpret_cast<const f32x4*>(b + ((long)ly.i0 * Wi + lx.i1) * ldx);
    const f32x4 p10 = *reinterpret_cast<const f32x4*>(b + ((long)ly.i1 * Wi + lx.i0) * ldx);
    const f32x4 p11 = *reinterpret_cast<const f32x4*>(b + ((long)ly.i1 * Wi + lx.i1) * ldx);
    f32x4 v = ly.l0 * (lx.l0 * p00 + lx.l1 * p01) + ly.l1 * (lx.l0 * p10 + lx.l1 * p11);
    float* o = y + ((long)(n * Ho + Y) * Wo + X) * ldy + cq * 4;
    if (ACC) v += *reinterpret_cast<const f32x4*>(o);
    *reinterpret_cast<f32x4*>(o) = v;
  }
}

// Adjoint in gather form.  grid.y = nsplit destination-row slices (partials) for big footprints.
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const float* __restrict__ dy, int ld_dy,
                                                           int N, int Hi, int Wi, int C4, int Ho,
                                                           int Wo, int align, float sh, float sw,
                                                           float* out, int ld_out, int accumulate,
                                                           int nsplit, long part_stride) {
  const long total = (long)N * Hi * Wi * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int cq = (int)(i % C4);
    long pix = i / C4;
    const int x = (int)(pix % Wi);
    pix /= Wi;
    const int y = (int)(pix % Hi);
    const int n = (int)(pix / Hi);
    int ylo, yhi, xlo, xhi;
    dst_range(y, sh, Ho, ylo, yhi);
    dst_range(x, sw, Wo, xlo, xhi);
    f32x4 g{0.f, 0.f, 0.f, 0.f};
    for (int Y = ylo + (int)blockIdx.y; Y <= yhi; Y += nsplit) {
      const Lerp ly = lerp_coord(Y, sh, Hi, align);
      const float wy = adj_weight(ly, y);
      if (wy == 0.f) continue;
      f32x4 row{0.f, 0.f, 0.f, 0.f};
      for (int X = xlo; X <= xhi; ++X) {
        const Lerp lx = lerp_coord(X, sw, Wi, align);
        const float wx = adj_weight(lx, x);
        if (wx == 0.f) continue;
        row += wx * *reinterpret_cast<const f32x4*>(dy + ((long)(n * Ho + Y) * Wo + X) * ld_dy +
                                                    cq * 4);
      }
      g += wy * row;
    }
    if (nsplit > 1) {
      // dense partial [z][N*Hi*Wi][C4*4]
      *reinterpret_cast<f32x4*>(out + blockIdx.y * part_stride + i * 4) = g;
    } else {
      float* o = out + ((long)(n * Hi + y) * Wi + x) * ld_out + cq * 4;
      if (accumulate) g += *reinterpret_cast<const f32x4*>(o);
      *reinterpret_cast<f32x4*>(o) = g;
    }
  }
}

__global__ __launch_bounds__(256) void sum_slices_kernel(const float* __restrict__ part,
                                                         long part_stride, int nsplit, long rows,
                                                         int C4, float* out, int ld_out,
                                                         int accumulate) {
  const long total = rows * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    f32x4 v = *reinterpret_cast<const f32x4*>(part + i * 4);
    for (int z = 1; z < nsplit; ++z)
      v += *reinterpret_cast<const f32x4*>(part + z * part_stride + i * 4);
    const long r = i / C4;
    const int cq = (int)(i - r * C4);
    float* o = out + r * ld_out + cq * 4;
    if (accumulate) v += *reinterpret_cast<const f32x4*>(o);
    *reinterpret_cast<f32x4*>(o) = v;
  }
}

// ------------------------------------------------------------------------------------------
// elementwise
// ------------------------------------------------------------------------------------------
template <bool ACC>
__global__ __launch_bounds__(256) void copy2d_kernel(const float* src, int ld_src, float* dst,
                                                     int ld_dst, long rows, int C4, float alpha) {
  const long total = rows * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const long r = i / C4;
    const int cq = (int)(i - r * C4);
    f32x4 v = *reinterpret_cast<const f32x4*>(src + r * ld_src + cq * 4) * alpha;
    float* o = dst + r * ld_dst + cq * 4;
    if (ACC) v += *reinterpret_cast<const f32x4*>(o);
    *reinterpret_cast<f32x4*>(o) = v;
  }
}

__global__ __launch_bounds__(256) void scale_nc_kernel(const float* x, int ldx,
                                                       const float* __restrict__ mask, int N,
                                                       long ppi, int C, float* y, int ldy) {
  const int C4 = C >> 2;
  const long total = (long)N * ppi * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const long r = i / C4;
    const int cq = (int)(i - r * C4);
    const int n = (int)(r / ppi);
    const f32x4 m = *reinterpret_cast<const f32x4*>(mask + (long)n * C + cq * 4);
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + cq * 4) * m;
    *reinterpret_cast<f32x4*>(y + r * ldy + cq * 4) = v;
  }
}

// torch.optim.SGD step over a flat range (cfg: pspnet_ar50to101v2_gsync.py:175)
// ZERO: the gradient is cleared once it has been consumed (the next step then needs no zero_grad
// fill kernels in front of its forward pass)
// HYPER: lr / momentum / weight decay / gradient scale are read from device memory at run time
// (`hyper` = {lr, momentum, wd, gscale}), so that a captured hipGraph of the whole training step can
// be replayed under a learning-rate schedule without re-capturing.
template <bool ZERO, bool HYPER>
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, float* __restrict__ g,
                                                  float* __restrict__ m, long n4, float lr,
                                                  float momentum, float wd, float gscale,
                                                  const float* __restrict__ hyper) {
  if (HYPER) { lr = hyper[0]; momentum = hyper[1]; wd = hyper[2]; gscale = hyper[3]; }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (long)gridDim.x * blockDim.x) {
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
    f32x4 gv = reinterpret_cast<const f32x4*>(g)[i] * gscale + pv * wd;
    f32x4 mv = reinterpret_cast<f32x4*>(m)[i] * momentum + gv;
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(p)[i] = pv - mv * lr;
    if (ZERO) reinterpret_cast<f32x4*>(g)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

// nn.AvgPool2d(kernel_size=s, stride=s, ceil_mode=True, count_include_pad=False): the pooling in
// front of the 1x1 shortcut conv when avg_down=True (gaiaseg/models/utils/dynamic_res_layer.py:75-82).
// Windows never overlap; a border window that ceil_mode lets hang over the image is averaged over
// its in-bounds pixels only.  BWD = false: y = mean(window); BWD = true: dx = dy / count(window).
template <bool BWD, bool ACC>
__global__ __launch_bounds__(256) void avgpool_ceil_kernel(const float* __restrict__ src, int ld_src,
                                                           int N, int H, int W, int C4, int s, int Ho,
                                                           int Wo, float* dst, int ld_dst) {
  const long total = (long)N * (BWD ? (long)H * W : (long)Ho * Wo) * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int cq = (int)(i % C4);
    long pix = i / C4;
    if (BWD) {
      const int w = (int)(pix % W);
      pix /= W;
      const int h = (int)(pix % H);
      const int n = (int)(pix / H);
      const int ho = h / s, wo = w / s;
      const int cnt = (min(ho * s + s, H) - ho * s) * (min(wo * s + s, W) - wo * s);
      f32x4 v = *reinterpret_cast<const f32x4*>(src + ((long)(n * Ho + ho) * Wo + wo) * ld_src + cq * 4);
      v = v / (float)cnt;
      float* o = dst + ((long)(n * H + h) * W + w) * ld_dst + cq * 4;
      if (ACC) v += *reinterpret_cast<const f32x4*>(o);
      *reinterpret_cast<f32x4*>(o) = v;
    } else {
      const int wo = (int)(pix % Wo);
      pix /= Wo;
      const int ho = (int)(pix % Ho);
      const int n = (int)(pix / Ho);
      const int h1 = min(ho * s + s, H), w1 = min(wo * s + s, W);
      f32x4 acc{0.f, 0.f, 0.f, 0.f};
      for (int h = ho * s; h < h1; ++h)
        for (int w = wo * s; w < w1; ++w)
          acc += *reinterpret_cast<const f32x4*>(src + ((long)(n * H + h) * W + w) * ld_src + cq * 4);
      acc = acc / (float)((h1 - ho * s) * (w1 - wo * s));
      *reinterpret_cast<f32x4*>(dst + ((long)(n * Ho + ho) * Wo + wo) * ld_dst + cq * 4) = acc;
    }
  }
}

static int check_nhwc(const void* p, int C, int ld) {
  if (!p) return GS_E_NULL;
  if (C <= 0) return GS_E_BADARG;
  if ((C & 3) || (ld & 3) || ld < C || !aligned16(p)) return GS_E_ALIGN;
  return GS_OK;
}
static int fill_scales(const int32_t* scales, int nscales, Scales& sc) {
  if (!scales) return GS_E_NULL;
  if (nscales <= 0 || nscales > kMaxScales) return GS_E_BADARG;
  sc.n = nscales;
  sc.off[0] = 0;
  for (int i = 0; i < nscales; ++i) {
    if (scales[i] <= 0) return GS_E_BADARG;
    sc.s[i] = scales[i];
    sc.off[i + 1] = sc.off[i] + scales[i] * scales[i];
  }
  return GS_OK;
}

}  // namespace gs

using namespace gs;

extern "C" int gs_maxpool_forward(const float* x, int32_t N, int32_t H, int32_t W, int32_t C,
                                  int32_t ldx, int32_t k, int32_t s, int32_t p, int32_t Ho,
                                  int32_t Wo, float* y, int32_t ldy, uint8_t* idx, void* stream) {
  int rc = check_nhwc(x, C, ldx);
  if (rc) return rc;
  if ((rc = check_nhwc(y, C, ldy))) return rc;
  if (!idx) return GS_E_NULL;
  if (k <= 0 || k > 15 || s <= 0 || p < 0 || 2 * p > k) return GS_E_BADARG;
  if (Ho != (H + 2 * p - k) / s + 1 || Wo != (W + 2 * p - k) / s + 1 || Ho <= 0 || Wo <= 0)
    return GS_E_BADARG;
  if (reinterpret_cast<uintptr_t>(idx) & 3) return GS_E_ALIGN;
  const long total = (long)N * Ho * Wo * (C >> 2);
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(stream_grid(total, 256)), dim3(256), 0,
                     as_stream(stream), x, N, H, W, C >> 2, ldx, k, s, p, Ho, Wo, y, ldy, idx);
  return launch_status();
}

extern "C" int gs_maxpool_backward(const float* dy, int32_t ld_dy, const uint8_t* idx, int32_t N,
                                   int32_t H, int32_t W, int32_t C, int32_t k, int32_t s, int32_t p,
                                   int32_t Ho, int32_t Wo, float* dx, int32_t ld_dx,
                                   int32_t accumulate, void* stream) {
  int rc = check_nhwc(dy, C, ld_dy);
  if (rc) return rc;
  if ((rc = check_nhwc(dx, C, ld_dx))) return rc;
  if (!idx) return GS_E_NULL;
  if (k <= 0 || k > 15 || s <= 0 || p < 0) return GS_E_BADARG;
  const long total = (long)N * H * W * (C >> 2);
  const dim3 grid(stream_grid(total, 256));
  if (accumulate)
    hipLaunchKernelGGL(maxpool_bwd_kernel<true>, grid, dim3(256), 0, as_stream(stream), dy, ld_dy,
                       idx, N, H, W, C >> 2, k, s, p, Ho, Wo, dx, ld_dx);
  else
    hipLaunchKernelGGL(maxpool_bwd_kernel<false>, grid, dim3(256), 0, as_stream(stream), dy, ld_dy,
                       idx, N, H, W, C >> 2, k, s, p, Ho, Wo, dx, ld_dx);
  return launch_status();
}

// pixel splits so that big bins (OS8 feature maps) still fill the chip
static int avgpool_splits(int N, int H, int W, int C, int total_bins) {
  const int C4 = C >> 2;
  const int rpi = C4 <= 256 ? 256 / C4 : 1;
  const long blocks = (long)N * total_bins * (C4 <= 256 ? 1 : ceil_div(C4, 256));
  long want = ceil_div(2 * num_cu(), blocks);
  const long max_by_px = std::max<long>(1, ((long)H * W) / (8L * rpi));
  if (want > max_by_px) want = max_by_px;
  if (want > 32) want = 32;
  return (int)std::max<long>(want, 1);
}

extern "C" int gs_avgpool_ceil_forward(const float* x, int32_t N, int32_t H, int32_t W, int32_t C,
                                       int32_t ldx, int32_t s, float* y, int32_t ldy, void* stream) {
  int rc = check_nhwc(x, C, ldx);
  if (rc) return rc;
  if ((rc = check_nhwc(y, C, ldy))) return rc;
  if (N <= 0 || H <= 0 || W <= 0 || s <= 0) return GS_E_BADARG;
  const int Ho = (H + s - 1) / s, Wo = (W + s - 1) / s;
  const long total = (long)N * Ho * Wo * (C >> 2);
  hipLaunchKernelGGL((avgpool_ceil_kernel<false, false>), dim3(stream_grid(total, 256)), dim3(256), 0,
                     as_stream(stream), x, ldx, N, H, W, C >> 2, s, Ho, Wo, y, ldy);
  return launch_status();
}

extern "C" int gs_avgpool_ceil_backward(const float* dy, int32_t ld_dy, int32_t N, int32_t H, int32_t W,
                                        int32_t C, int32_t s, float* dx, int32_t ld_dx,
                                        int32_t accumulate, void* stream) {
  int rc = check_nhwc(dy, C, ld_dy);
  if (rc) return rc;
  if ((rc = check_nhwc(dx, C, ld_dx))) return rc;
  if (N <= 0 || H <= 0 || W <= 0 || s <= 0) return GS_E_BADARG;
  const int Ho = (H + s - 1) / s, Wo = (W + s - 1) / s;
  const long total = (long)N * H * W * (C >> 2);
  const dim3 grid(stream_grid(total, 256));
  if (accumulate)
    hipLaunchKernelGGL((avgpool_ceil_kernel<true, true>), grid, dim3(256), 0, as_stream(stream), dy,
                       ld_dy, N, H, W, C >> 2, s, Ho, Wo, dx, ld_dx);
  else
    hipLaunchKernelGGL((avgpool_ceil_kernel<true, false>), grid, dim3(256), 0, as_stream(stream), dy,
                       ld_dy, N, H, W, C >> 2, s, Ho, Wo, dx, ld_dx);
  return launch_status();
}

extern "C" size_t gs_adaptive_avgpool_workspace_bytes(int32_t N, int32_t H, int32_t W, int32_t C,
                                                      const int32_t* scales, int32_t nscales) {
  Scales sc;
  if (fill_scales(scales, nscales, sc)) return 0;
  const int ns = avgpool_splits(N, H, W, C, sc.off[sc.n]);
  return (size_t)ns * N * sc.off[sc.n] * C * sizeof(float);
}

extern "C" int gs_adaptive_avgpool_forward(const float* x, int32_t N, int32_t H, int32_t W,
                                           int32_t C, int32_t ldx, const int32_t* scales,
                                           int32_t nscales, float* y, void* workspace,
                                           size_t workspace_bytes, void* stream) {
  int rc = check_nhwc(x, C, ldx);
  if (rc) return rc;
  if (!y || !workspace) return GS_E_NULL;
  if (!aligned16(y) || !aligned16(workspace)) return GS_E_ALIGN;
  Scales sc;
  if ((rc = fill_scales(scales, nscales, sc))) return rc;
  const int tb = sc.off[sc.n];
  const int ns = avgpool_splits(N, H, W, C, tb);
  const long pstride = (long)N * tb * C;
  if ((size_t)ns * pstride * sizeof(float) > workspace_bytes) return GS_E_WORKSPACE;
  const int C4 = C >> 2;
  hipStream_t st = as_stream(stream);
  float* part = static_cast<float*>(workspace);
  hipLaunchKernelGGL(avgpool_partial_kernel,
                     dim3(N * tb, C4 <= 256 ? 1 : (unsigned)ceil_div(C4, 256), ns), dim3(256), 0, st,
                     x, H, W, C, ldx, sc, ns, part, pstride);
  hipLaunchKernelGGL(avgpool_final_kernel, dim3(stream_grid((long)N * tb * C4, 256)), dim3(256), 0,
                     st, part, pstride, ns, N, H, W, C, sc, y);
  return launch_status();
}

extern "C" int gs_adaptive_avgpool_backward(const float* dy, int32_t N, int32_t H, int32_t W,
                                            int32_t C, const int32_t* scales, int32_t nscales,
                                            float* dx, int32_t ld_dx, int32_t accumulate,
                                            void* stream) {
  int rc = check_nhwc(dx, C, ld_dx);
  if (rc) return rc;
  if (!dy) return GS_E_NULL;
  if (!aligned16(dy)) return GS_E_ALIGN;
  Scales sc;
  if ((rc = fill_scales(scales, nscales, sc))) return rc;
  const long total = (long)N * H * W * (C >> 2);
  const dim3 grid(stream_grid(total, 256));
  if (accumulate)
    hipLaunchKernelGGL(avgpool_bwd_kernel<true>, grid, dim3(256), 0, as_stream(stream), dy, N, H, W,
                       C, sc, dx, ld_dx);
  else
    hipLaunchKernelGGL(avgpool_bwd_kernel<false>, grid, dim3(256), 0, as_stream(stream), dy, N, H, W,
                       C, sc, dx, ld_dx);
  return launch_status();
}

extern "C" int gs_bilinear_forward(const float* x, int32_t N, int32_t Hi, int32_t Wi, int32_t C,
                                   int32_t ldx, int32_t Ho, int32_t Wo, int32_t align_corners,
                                   float* y, int32_t ldy, int32_t accumulate, void* stream) {
  int rc = check_nhwc(x, C, ldx);
  if (rc) return rc;
  if ((rc = check_nhwc(y, C, ldy))) return rc;
  if (N <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return GS_E_BADARG;
  const float sh = resize_scale(Hi, Ho, align_corners), sw = resize_scale(Wi, Wo, align_corners);
  const long total = (long)N * Ho * Wo * (C >> 2);
  const dim3 grid(stream_grid(total, 256));
  if (accumulate)
    hipLaunchKernelGGL(bilinear_fwd_kernel<true>, grid, dim3(256), 0, as_stream(stream), x, N, Hi,
                       Wi, C >> 2, ldx, Ho, Wo, align_corners, sh, sw, y, ldy);
  else
    hipLaunchKernelGGL(bilinear_fwd_kernel<false>, grid, dim3(256), 0, as_stream(stream), x, N, Hi,
                       Wi, C >> 2, ldx, Ho, Wo, align_corners, sh, sw, y, ldy);
  return launch_status();
}

static int bilinear_bwd_splits(int N, int Hi, int Wi, int C, int Ho) {
  const long threads = (long)N * Hi * Wi * (C >> 2);
  const long rows_per_src = std::max<long>(1, (2L * Ho) / std::max(Hi, 1));  // footprint rows
  long want = ceil_div((long)num_cu() * 256 * 2, threads);
  if (want > rows_per_src / 2) want = rows_per_src / 2;
  if (want > 32) want = 32;
  return (int)std::max<long>(want, 1);
}

extern "C" size_t gs_bilinear_backward_workspace_bytes(int32_t N, int32_t Hi, int32_t Wi, int32_t C,
                                                       int32_t Ho, int32_t Wo) {
  (void)Wo;
  const int ns = bilinear_bwd_splits(N, Hi, Wi, C, Ho);
  return ns > 1 ? (size_t)ns * N * Hi * Wi * C * sizeof(float) : 0;
}

extern "C" int gs_bilinear_backward(const float* dy, int32_t ld_dy, int32_t N, int32_t Hi,
                                    int32_t Wi, int32_t C, int32_t Ho, int32_t Wo,
                                    int32_t align_corners, float* dx, int32_t ld_dx,
                                    int32_t accumulate, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  int rc = check_nhwc(dy, C, ld_dy);
  if (rc) return rc;
  if ((rc = check_nhwc(dx, C, ld_dx))) return rc;
  if (N <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return GS_E_BADARG;
  const float sh = resize_scale(Hi, Ho, align_corners), sw = resize_scale(Wi, Wo, align_corners);
  const int C4 = C >> 2;
  const long total = (long)N * Hi * Wi * C4;
  const int ns = bilinear_bwd_splits(N, Hi, Wi, C, Ho);
  hipStream_t st = as_stream(stream);
  if (ns > 1) {
    const long pstride = total * 4;
    if ((size_t)ns * pstride * sizeof(float) > workspace_bytes || !workspace) return GS_E_WORKSPACE;
    if (!aligned16(workspace)) return GS_E_ALIGN;
    float* part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(stream_grid(total, 256), ns), dim3(256), 0, st, dy,
                       ld_dy, N, Hi, Wi, C4, Ho, Wo, align_corners, sh, sw, part, 0, 0, ns, pstride);
    hipLaunchKernelGGL(sum_slices_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, st, part,
                       pstride, ns, (long)N * Hi * Wi, C4, dx, ld_dx, accumulate);
  } else {
    hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(stream_grid(total, 256), 1), dim3(256), 0, st, dy,
                       ld_dy, N, Hi, Wi, C4, Ho, Wo, align_corners, sh, sw, dx, ld_dx, accumulate, 1,
                       0L);
  }
  return launch_status();
}

extern "C" int gs_copy2d(const float* src, int32_t ld_src, float* dst, int32_t ld_dst, int64_t rows,
                         int32_t C, float alpha, int32_t accumulate, void* stream) {
  int rc = check_nhwc(src, C, ld_src);
  if (rc) return rc;
  if ((rc = check_nhwc(dst, C, ld_dst))) return rc;
  if (rows <= 0) return GS_E_BADARG;
  const long total = rows * (C >> 2);
  const dim3 grid(stream_grid(total, 256));
  if (accumulate)
    hipLaunchKernelGGL(copy2d_kernel<true>, grid, dim3(256), 0, as_stream(stream), src, ld_src, dst,
                       ld_dst, (long)rows, C >> 2, alpha);
  else
    hipLaunchKernelGGL(copy2d_kernel<false>, grid, dim3(256), 0, as_stream(stream), src, ld_src,
                       dst, ld_dst, (long)rows, C >> 2, alpha);
  return launch_status();
}

extern "C" int gs_scale_nc(const float* x, int32_t ldx, const float* mask, int32_t N,
                           int64_t pixels_per_image, int32_t C, float* y, int32_t ldy,
                           void* stream) {
  int rc = check_nhwc(x, C, ldx);
  if (rc) return rc;
  if ((rc = check_nhwc(y, C, ldy))) return rc;
  if (!mask) return GS_E_NULL;
  if (!aligned16(mask)) return GS_E_ALIGN;
  if (N <= 0 || pixels_per_image <= 0) return GS_E_BADARG;
  const long total = (long)N * pixels_per_image * (C >> 2);
  hipLaunchKernelGGL(scale_nc_kernel, dim3(stream_grid(total, 256)), dim3(256), 0,
                     as_stream(stream), x, ldx, mask, N, (long)pixels_per_image, C, y, ldy);
  return launch_status();
}

extern "C" int gs_sgd_step(float* param, float* grad, float* momentum_buf, int64_t n, float lr,
                           float momentum, float weight_decay, float grad_scale, int32_t zero_grad,
                           void* stream) {
  if (!param || !grad || !momentum_buf) return GS_E_NULL;
  if (n <= 0) return GS_E_BADARG;
  if ((n & 3) || !aligned16(param) || !aligned16(grad) || !aligned16(momentum_buf))
    return GS_E_ALIGN;
  if (zero_grad)
    hipLaunchKernelGGL((sgd_kernel<true, false>), dim3(stream_grid(n >> 2, 256)), dim3(256), 0,
                       as_stream(stream), param, grad, momentum_buf, (long)(n >> 2), lr, momentum,
                       weight_decay, grad_scale, (const float*)nullptr);
  else
    hipLaunchKernelGGL((sgd_kernel<false, false>), dim3(stream_grid(n >> 2, 256)), dim3(256), 0,
                       as_stream(stream), param, grad, momentum_buf, (long)(n >> 2), lr, momentum,
                       weight_decay, grad_scale, (const float*)nullptr);
  return launch_status();
}

__global__ void set_hyper_kernel(float* hyper, float lr, float momentum, float wd, float gscale) {
  hyper[0] = lr; hyper[1] = momentum; hyper[2] = wd; hyper[3] = gscale;
}

extern "C" int gs_sgd_set_hyper(float* hyper, float lr, float momentum, float weight_decay,
                                float grad_scale, void* stream) {
  if (!hyper) return GS_E_NULL;
  if (!aligned16(hyper)) return GS_E_ALIGN;
  hipLaunchKernelGGL(set_hyper_kernel, dim3(1), dim3(1), 0, as_stream(stream), hyper, lr, momentum,
                     weight_decay, grad_scale);
  return launch_status();
}

extern "C" int gs_sgd_step_hyper(float* param, float* grad, float* momentum_buf, int64_t n,
                                 const float* hyper, int32_t zero_grad, void* stream) {
  if (!param || !grad || !momentum_buf || !hyper) return GS_E_NULL;
  if (n <= 0) return GS_E_BADARG;
  if ((n & 3) || !aligned16(param) || !aligned16(grad) || !aligned16(momentum_buf) ||
      !aligned16(hyper))
    return GS_E_ALIGN;
  if (zero_grad)
    hipLaunchKernelGGL((sgd_kernel<true, true>), dim3(stream_grid(n >> 2, 256)), dim3(256), 0,
                       as_stream(stream), param, grad, momentum_buf, (long)(n >> 2), 0.f, 0.f, 0.f,
                       0.f, hyper);
  else
    hipLaunchKernelGGL((sgd_kernel<false, true>), dim3(stream_grid(n >> 2, 256)), dim3(256), 0,
                       as_stream(stream), param, grad, momentum_buf, (long)(n >> 2), 0.f, 0.f, 0.f,
                       0.f, hyper);
  return launch_status();
}
