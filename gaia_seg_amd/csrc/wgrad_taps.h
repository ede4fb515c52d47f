// 3x3 weight gradient, all nine taps from one staged halo (stride 1, dilation 1, W % 16 == 0).
//
//   dW[kh][kw][ci][co] = sum_{n,h,w} x~[n, h+kh-1, w+kw-1, ci] * dy[n, h, w, co]
//
// The tap-major kernel (igemm_wgrad_fast_kernel) treats the nine taps as nine independent GEMM row
// blocks: every tap re-gathers x and re-reads dy, so the stage-1/2 weight gradients of the
// bottleneck 3x3 (65536 / 16384 pixels, 48..160 channels) move 9x the data and run at the L2 / HBM
// rate: 158 us for 4.8 GFLOP at stage 1 of R50 (0.19 of the fp32 MFMA peak, r01 profile).
// Here a workgroup owns (<= 64 input channels) x (64 output channels) x ALL taps — 36 accumulator
// tiles (144 VGPRs) per wave — and walks down a 16-pixel-wide column strip of the image: a K step is
// one image row of the strip (16 output pixels); it needs x rows h-1, h, h+1 (18 pixels wide), of
// which two are already in LDS from the previous steps (4-slot ring), and one dy row.  Per K step a
// workgroup loads 18x64 + 16x64 floats and issues 4 x 36 MFMAs per wave: 7x the arithmetic per
// byte of the tap-major form, 13 LDS fragment reads per 36 MFMAs instead of 5 per 4, one barrier per
// 4608 MFMA cycles.
//
// Split-K runs over (image, column strip, row range); the partial slabs use the tap-major kernel's
// layout [split][tap * Ci + ci][Co] and its fixed-order reduce (splitk_reduce_kernel).
// AFF: x~ = relu(bn(x)) evaluated in the loader (IgemmArgs::a_coeffs), padding exactly zero.
#pragma once
#include "igemm_core.h"

namespace gs {

struct WgTapsArgs {
  const float* x;
  const float* dy;
  float* out;        // dW (no split)
  float* slab;       // [nsplit][9 * Ci][Co] or NULL
  const float* a_coeffs;
  long xs_n, xs_h, xs_w;   // element strides of x (channel stride 1)
  int N, H, W, Ci, Co, ld_dy;
  int cb;                  // input channels per workgroup (multiple of 16, <= 64)
  int tiles_m, tiles_n;    // ci blocks, co blocks (64 wide)
  int wseg;                // W / 16
  int rows_per_split, hsplits;
  long o_tap;
  int o_row;
  unsigned x_bytes, dy_bytes;
};

constexpr int WT_PX = 80;                 // LDS pitch of a pixel (64 channels + 16: pitch % 32 == 16)
constexpr int WT_XS = 18 * WT_PX;         // one x row slot: 18 pixels
constexpr int WT_DS = 16 * WT_PX;         // one dy row stage: 16 pixels
constexpr int WT_LDS = 4 * WT_XS + 2 * WT_DS;

template <bool AFF>
__global__ __launch_bounds__(NT) void wgrad_taps_kernel(const WgTapsArgs p) {
  __shared__ __attribute__((aligned(16))) float lds[WT_LDS];
  float* xs = lds;
  float* ds = lds + 4 * WT_XS;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int kk = lane >> 4, li = lane & 15;

  // block -> (split, tile); split -> (image, column strip, row range)
  const int ntiles = p.tiles_m * p.tiles_n;
  const int nsplit = p.N * p.wseg * p.hsplits;
  const int lin = xcd_remap(blockIdx.x, ntiles * nsplit);
  const int split = lin / ntiles;
  const int tile = lin - split * ntiles;
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int ci0 = tm * p.cb, n0 = tn * 64;
  const int hs = split % p.hsplits;
  const int strip = split / p.hsplits;
  const int n = strip / p.wseg;
  const int w0 = (strip - n * p.wseg) * 16;
  const int h0 = hs * p.rows_per_split;
  const int h1 = min(h0 + p.rows_per_split, p.H);
  const int cb = min(p.cb, p.Ci - ci0);          // active input channels of this block

  const __amdgpu_buffer_rsrc_t rs_x =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dy =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, p.dy_bytes, 0x00020000);
  constexpr unsigned kOOB = 0xFFFFFFFFu;

  // loader roles.  x row: 18 pixels x 16 channel quads = 288 quads -> slots t and t + 256;
  // dy row: 16 pixels x 16 column quads = 256 quads -> one per thread.
  const int xq = t & 15;                  // channel quad (fixed per thread, both slots)
  const int xpx0 = t >> 4;                // pixel 0..15 (slot 0), 16..17 for t < 32 (slot 1)
  const bool x_cv = xq * 4 < cb;          // channel quad inside the block
  const bool slot1 = t < 32;
  const int xcol0 = w0 - 1 + xpx0, xcol1 = w0 - 1 + 16 + xpx0;
  const bool c0ok = x_cv && (unsigned)xcol0 < (unsigned)p.W;
  const bool c1ok = x_cv && slot1 && (unsigned)xcol1 < (unsigned)p.W;
  const long xbase = (long)n * p.xs_n + ci0 + xq * 4;
  const int dq = t & 15, dpx = t >> 4;
  const bool d_ok = n0 + dq * 4 < p.Co;
  const long dybase = ((long)n * p.H * p.W + w0 + dpx) * p.ld_dy + n0 + dq * 4;

  f32x4 a_mean{0.f, 0.f, 0.f, 0.f}, a_scale{0.f, 0.f, 0.f, 0.f}, a_beta{0.f, 0.f, 0.f, 0.f};
  if constexpr (AFF) {
    if (x_cv) {   // coefficient order in memory: scale, beta, mean, invstd
      a_scale = *reinterpret_cast<const f32x4*>(p.a_coeffs + ci0 + xq * 4);
      a_beta = *reinterpret_cast<const f32x4*>(p.a_coeffs + p.Ci + ci0 + xq * 4);
      a_mean = *reinterpret_cast<const f32x4*>(p.a_coeffs + 2 * p.Ci + ci0 + xq * 4);
    }
  }

  auto load_x = [&](int row, f32x4& r0, f32x4& r1, unsigned& ok) __attribute__((always_inline)) {
    const bool rv = (unsigned)row < (unsigned)p.H;
    const bool ok0 = rv && c0ok, ok1 = rv && c1ok;
    const long rb = xbase + (long)row * p.xs_h;
    const unsigned o0 = ok0 ? (unsigned)(4 * (rb + (long)xcol0 * p.xs_w)) : kOOB;
    const unsigned o1 = ok1 ? (unsigned)(4 * (rb + (long)xcol1 * p.xs_w)) : kOOB;
    r0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, o0, 0, 0));
    r1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, o1, 0, 0));
    ok = (ok0 ? 1u : 0u) | (ok1 ? 2u : 0u);
  };
  auto store_x = [&](int row, f32x4 r0, f32x4 r1, unsigned ok) __attribute__((always_inline)) {
    float* slot = xs + (row & 3) * WT_XS;
    if constexpr (AFF) {
      r0 = bn_relu_affine(r0, a_mean, a_scale, a_beta);
      r1 = bn_relu_affine(r1, a_mean, a_scale, a_beta);
      if (!(ok & 1u)) r0 = f32x4{0.f, 0.f, 0.f, 0.f};   // padding / inactive channels stay zero
      if (!(ok & 2u)) r1 = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    *reinterpret_cast<f32x4*>(slot + xpx0 * WT_PX + xq * 4) = r0;
    if (slot1) *reinterpret_cast<f32x4*>(slot + (16 + xpx0) * WT_PX + xq * 4) = r1;
  };
  auto load_dy = [&](int row, f32x4& r) __attribute__((always_inline)) {
    const bool ok = d_ok && row < h1;
    const unsigned o = ok ? (unsigned)(4 * (dybase + (long)row * p.W * p.ld_dy)) : kOOB;
    r = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, o, 0, 0));
  };
  auto store_dy = [&](int row, f32x4 r) __attribute__((always_inline)) {
    *reinterpret_cast<f32x4*>(ds + (row & 1) * WT_DS + dpx * WT_PX + dq * 4) = r;
  };

  f32x4 acc[9][4];
#pragma unroll
  for (int a = 0; a < 9; ++a)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[a][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const bool wave_on = wave * 16 < cb;    // wave-uniform: this wave's 16 channels exist
  const int a_off = kk * WT_PX + wave * 16 + li;
  const int b_off = kk * WT_PX + li;

  if (h0 < h1) {
    // prologue: rows h0-1, h0, h0+1 and dy row h0
    f32x4 r0, r1, rd;
    unsigned ok;
#pragma unroll
    for (int dr = -1; dr <= 1; ++dr) {
      load_x(h0 + dr, r0, r1, ok);
      store_x(h0 + dr, r0, r1, ok);
    }
    load_dy(h0, rd);
    store_dy(h0, rd);
    __syncthreads();
    for (int h = h0; h < h1; ++h) {
      // next step's operands: x row h+2 into the free ring slot, dy row h+1 into the other stage
      load_x(h + 2, r0, r1, ok);
      load_dy(h + 1, rd);
      if (wave_on) {
        const float* dsb = ds + (h & 1) * WT_DS + b_off;
        const float* x0 = xs + ((h - 1) & 3) * WT_XS + a_off;
        const float* x1 = xs + (h & 3) * WT_XS + a_off;
        const float* x2 = xs + ((h + 1) & 3) * WT_XS + a_off;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float b[4], a[9];
#pragma unroll
          for (int j = 0; j < 4; ++j) b[j] = dsb[g * 4 * WT_PX + j * 16];
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            a[kw] = x0[(g * 4 + kw) * WT_PX];
            a[3 + kw] = x1[(g * 4 + kw) * WT_PX];
            a[6 + kw] = x2[(g * 4 + kw) * WT_PX];
          }
#pragma unroll
          for (int tp = 0; tp < 9; ++tp)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[tp][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tp], b[j], acc[tp][j], 0, 0, 0);
        }
      }
      store_x(h + 2, r0, r1, ok);
      store_dy(h + 1, rd);
      __syncthreads();
    }
  }

  // epilogue: C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
  if (wave_on) {
    const int M = 9 * p.Ci;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = n0 + j * 16 + li;
        if (col < p.Co) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int ci = wave * 16 + kk * 4 + r;
            if (ci < cb) {
              if (p.slab)
                p.slab[((long)split * M + (long)tp * p.Ci + ci0 + ci) * p.Co + col] = acc[tp][j][r];
              else
                p.out[(long)tp * p.o_tap + (long)(ci0 + ci) * p.o_row + col] = acc[tp][j][r];
            }
          }
        }
      }
  }
}

// Shapes the all-taps kernel takes: 3x3, stride 1, dilation 1 (pad 1), NHWC x, W % 16 == 0.
static inline bool wgrad_taps_ok(const gs_conv_desc* d) {
  static const int off = env_int("GS_NO_WGRAD_TAPS", 0);
  if (off) return false;
  if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->dil != 1 || d->pad != 1) return false;
  if (!x_is_vector(d) || (d->Ci & 15) || (d->Co & 3) || (d->W & 15)) return false;
  if (d->Ho != d->H || d->Wo != d->W) return false;
  const size_t xb = (size_t)d->N * d->x_sn * sizeof(float);
  const size_t db = (size_t)d->N * d->H * d->W * d->ldy * sizeof(float);
  return xb < (1ull << 31) && db < (1ull << 31);
}

struct WgTapsPlan {
  int cb, tiles_m, tiles_n, rows_per_split, hsplits, nsplit;
};

static inline WgTapsPlan wgrad_taps_plan(const gs_conv_desc* d) {
  WgTapsPlan pl{};
  const int nb = (int)ceil_div(d->Ci, 64);
  pl.cb = (int)ceil_div(ceil_div(d->Ci, nb), 16) * 16;   // balanced ci blocks (80 -> 48 + 32)
  pl.tiles_m = (int)ceil_div(d->Ci, pl.cb);
  pl.tiles_n = (int)ceil_div(d->Co, 64);
  const int tiles = pl.tiles_m * pl.tiles_n;
  const int strips = d->N * (d->W / 16);
  // workgroups: about two per CU; a row range keeps >= 4 rows (3 halo rows of prologue each)
  static const int target = env_int("GS_WGT_TARGET", 2 * kNumCU);
  long want = ceil_div(target, (long)tiles * strips);
  const long max_by_rows = std::max<long>(1, d->H / 4);
  if (want > max_by_rows) want = max_by_rows;
  if (want < 1) want = 1;
  const size_t per_split = (size_t)9 * d->Ci * d->Co * sizeof(float);
  while (want > 1 && (size_t)want * strips * per_split > kMaxSlabBytes) --want;
  pl.rows_per_split = (int)ceil_div(d->H, want);
  pl.hsplits = (int)ceil_div(d->H, pl.rows_per_split);
  pl.nsplit = strips * pl.hsplits;
  return pl;
}

static inline size_t wgrad_taps_slab_bytes(const gs_conv_desc* d) {
  const WgTapsPlan pl = wgrad_taps_plan(d);
  return pl.nsplit > 1 ? (size_t)pl.nsplit * 9 * d->Ci * d->Co * sizeof(float) : 0;
}

}  // namespace gs
