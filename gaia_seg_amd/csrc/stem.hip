// The stem of the dynamic ResNet: 7x7 stride-2 pad-3 convolution of the 3-channel NCHW image
// (gaiaseg/models/backbones/dynamic_resnet.py:290-297, `conv1`), forward and weight gradient.
//
// The generic implicit-GEMM kernels gather every operand element of this shape with scalar index
// arithmetic (K = 7 * 7 * 3 = 147 is not a multiple of anything, the source is NCHW with three
// channels): r04 trace 121 us forward (2.5 GF: 20 TF) + a separate 17 us statistics pass, 114 us weight
// gradient (4.9 GF: 43 TF) -- in EVERY step whatever the sampled subnet, the forward at the head of the
// dependent chain, the weight gradient at the very end of backward.  Here: 67 us and 82-91 us.
//
// Here a workgroup owns 128 consecutive output pixels of ONE output row: their receptive field is a
// 3 x 7 x 261 patch of the image (22 KB) that is loaded once, coalesced along W, zero-filled outside
// the image, and both contractions read their A operand straight out of that patch in MFMA layout
// (v_mfma_f32_16x16x4_f32: exact fp32 products, like every other forward / weight-gradient kernel):
//   forward   y[p][co]  = sum_k patch[k @ p] * W[k][co]     the weights [147][Co] sit beside the patch
//   wgrad     dW[k][co] = sum_p patch[k @ p] * dy[p][co]    accumulated over a workgroup's tiles in
//             registers (10 x Co/16 accumulator blocks per wave, the four waves split the pixels),
//             one [147][Co] slab per workgroup, summed by the slab reduce in a fixed order.
// k = (kh * 7 + kw) * 3 + c is the row index of the physical HWIO weight layout, so W is read and dW
// written in place.  The forward epilogue also leaves the per-tile BatchNorm partial sums
// {sum (v - shift), sum (v - shift)^2, shift} that bn_tile_finalize merges (see rows_epilogue).
// Taken when: 7x7 / stride 2 / pad 3 / dilation 1, Ci = Ci_max = 3, unit W stride, Co in {32, 48, 64},
// Wo % 128 == 0 (1024 x 512, 2048 x 1024, 512 x 512 crops; config 4's 769 x 769 keeps the generic kernel).
#include "igemm_core.h"
#include "fused_internal.h"

namespace gs {

constexpr int kStemTW = 128;                 // output pixels per tile
constexpr int kStemPW = 2 * kStemTW + 5;     // input columns under a tile
constexpr int kStemPWp = 264;                // patch row pitch
constexpr int kStemK = 147;                  // 7 * 7 * 3
constexpr int kStemPatch = 3 * 7 * kStemPWp; // floats

// patch[(c * 7 + kh) * PWp + j] = x[n][c][2 ho - 3 + kh][2 wo0 - 3 + j] (0 outside the image)
__device__ __forceinline__ void stem_load_patch(float* __restrict__ patch, const float* __restrict__ x,
                                                long x_sn, long x_sc, long x_sh, int H, int W, int n,
                                                int ho, int wo0, int t) {
  const float* xb = x + (long)n * x_sn;
  for (int i = t; i < 3 * 7 * kStemPW; i += 256) {
    const int ck = i / kStemPW;              // c * 7 + kh
    const int j = i - ck * kStemPW;
    const int c = ck / 7, kh = ck - c * 7;
    const int hi = 2 * ho - 3 + kh, wi = 2 * wo0 - 3 + j;
    float v = 0.f;
    if (hi >= 0 && hi < H && wi >= 0 && wi < W) v = xb[(long)c * x_sc + (long)hi * x_sh + wi];
    patch[ck * kStemPWp + j] = v;
  }
}

// offset of tap row k inside the patch, relative to the pixel's column 2 * p
__device__ __forceinline__ int stem_koff(int k) {
  const int kc = k < kStemK ? k : kStemK - 1;
  const int tap = kc / 3, c = kc - 3 * tap;
  const int kh = tap / 7, kw = tap - 7 * kh;
  return (c * 7 + kh) * kStemPWp + kw;
}

// One element of the patch under tile (n, ho, wo0): flat index i in [0, 3 * 7 * 261)
__device__ __forceinline__ float stem_patch_elem(const float* __restrict__ xb, long x_sc, long x_sh, int H,
                                                 int W, int ho, int wo0, int i) {
  const int ck = i / kStemPW;
  const int j = i - ck * kStemPW;
  const int c = ck / 7, kh = ck - c * 7;
  const int hi = 2 * ho - 3 + kh, wi = 2 * wo0 - 3 + j;
  return (hi >= 0 && hi < H && wi >= 0 && wi < W) ? xb[(long)c * x_sc + (long)hi * x_sh + wi] : 0.f;
}
constexpr int kStemPre = (3 * 7 * kStemPW + 255) / 256;   // patch elements per thread (22)

// Forward.  Persistent workgroups (two per CU): the weights [147][Co] go to LDS once; the patch of the
// NEXT tile is fetched into registers while the MFMA loop of the current one runs, so the loop never
// waits for global memory (the first version loaded, computed and stored one tile per workgroup and ran
// at 107 us: r04 trace).  The C tile goes out through the patch's LDS in two 64-row halves.
template <int NB>   // Co = 16 * NB
__global__ __launch_bounds__(256, 2) void stem7x7_fwd_kernel(
    const float* __restrict__ x, long x_sn, long x_sc, long x_sh, int H, int W,
    const float* __restrict__ w, int co_ld, float* __restrict__ y, int ldy, int Ho, int Wo,
    int tiles_per_row, int tiles_total, float* __restrict__ tile_stats, int np) {
  constexpr int CO = 16 * NB, WP = CO + 8, KP = 148, PC = CO + 4;
  constexpr int LDSF = kStemPatch + KP * WP;
  static_assert(64 * PC + 2 * 256 <= kStemPatch, "half a C tile + statistics scratch fit in the patch area");
  __shared__ __attribute__((aligned(16))) float lds[LDSF];
  float* patch = lds;
  float* Wl = lds + kStemPatch;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, li = lane & 15, kq = lane >> 4;
  for (int i = t; i < KP * (CO / 4); i += 256) {
    const int k = i / (CO / 4), q = i - k * (CO / 4);
    f32x4 v{0.f, 0.f, 0.f, 0.f};
    if (k < kStemK) v = *reinterpret_cast<const f32x4*>(w + (long)k * co_ld + q * 4);
    *reinterpret_cast<f32x4*>(Wl + k * WP + q * 4) = v;
  }
  float pre[kStemPre];
  auto fetch = [&](int tile) {
    const int seg = tile % tiles_per_row;
    const int r = tile / tiles_per_row;
    const int ho = r % Ho, n = r / Ho;
    const float* xb = x + (long)n * x_sn;
#pragma unroll
    for (int u = 0; u < kStemPre; ++u) {
      const int i = t + u * 256;
      pre[u] = i < 3 * 7 * kStemPW ? stem_patch_elem(xb, x_sc, x_sh, H, W, ho, seg * kStemTW, i) : 0.f;
    }
  };
  int tile = blockIdx.x;
  if (tile < tiles_total) fetch(tile);
  const int p0 = 2 * (wave * 32 + li);       // patch column of this lane's pixel in row block 0
  constexpr int G = 256 / CO;                // statistics: G row groups x CO columns
  const int sc = t % CO, srg = t / CO;
  for (; tile < tiles_total; tile += gridDim.x) {
    const int seg = tile % tiles_per_row;
    const int r = tile / tiles_per_row;
    const int ho = r % Ho, n = r / Ho;
    const int wo0 = seg * kStemTW;
    __syncthreads();                          // the previous tile's C halves have left the patch area
#pragma unroll
    for (int u = 0; u < kStemPre; ++u) {
      const int i = t + u * 256;
      if (i < 3 * 7 * kStemPW) {
        const int ck = i / kStemPW;
        patch[ck * kStemPWp + (i - ck * kStemPW)] = pre[u];
      }
    }
    __syncthreads();
    if (tile + (int)gridDim.x < tiles_total) fetch(tile + gridDim.x);   // in flight during the MFMA loop
    f32x4 acc[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int kk = 0; kk < KP / 4; ++kk) {
      const int k = kk * 4 + kq;
      const int koff = stem_koff(k);
      float a[2], b[NB];
      a[0] = patch[koff + p0];
      a[1] = patch[koff + p0 + 32];
#pragma unroll
      for (int j = 0; j < NB; ++j) b[j] = Wl[k * WP + j * 16 + li];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    // C tile -> LDS (the patch area) -> coalesced rows, in two halves of 64 rows (waves 0-1, waves 2-3)
    float* Cs = lds;                          // [64][PC]
    float* red = lds + 64 * PC;               // [2][G][CO]
    const long m0 = ((long)n * Ho + ho) * Wo + wo0;
    float s1 = 0.f, s2 = 0.f, shift = 0.f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      __syncthreads();                        // operands / the previous half are no longer read
      if ((wave >> 1) == half) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              Cs[((wave & 1) * 32 + i * 16 + kq * 4 + e) * PC + j * 16 + li] = acc[i][j][e];
      }
      __syncthreads();
      for (int idx = t; idx < 64 * (CO / 4); idx += 256) {
        const int row = idx / (CO / 4), q = idx - row * (CO / 4);
        if (wo0 + half * 64 + row < Wo)
          *reinterpret_cast<f32x4*>(y + (m0 + half * 64 + row) * ldy + q * 4) =
              *reinterpret_cast<const f32x4*>(&Cs[row * PC + q * 4]);
      }
      if (tile_stats && srg < G) {            // (a statistics buffer only when every tile is full)
        if (half == 0) shift = Cs[sc];        // the tile's first row
        for (int rr = srg; rr < 64; rr += G) {
          const float v = Cs[rr * PC + sc] - shift;
          s1 += v;
          s2 += v * v;
        }
      }
    }
    if (tile_stats) {
      if (srg < G) {
        red[srg * CO + sc] = s1;
        red[(G + srg) * CO + sc] = s2;
      }
      __syncthreads();
      if (srg == 0) {
        for (int g = 1; g < G; ++g) {
          s1 += red[g * CO + sc];
          s2 += red[(G + g) * CO + sc];
        }
        const long C4 = CO >> 2;
        const int cq = sc >> 2, e = sc & 3;
        tile_stats[((0 * C4 + cq) * np + tile) * 4 + e] = s1;
        tile_stats[((1 * C4 + cq) * np + tile) * 4 + e] = s2;
        tile_stats[((2 * C4 + cq) * np + tile) * 4 + e] = shift;
      }
    }
  }
}

// Weight gradient.  Persistent workgroups; waves (0, 1) own tap rows 0..79, waves (2, 3) rows 80..159,
// and inside a pair the two waves split the tile's 128 pixels — 5 x Co/16 accumulator blocks per wave,
// which leaves registers for the NEXT tile's patch and dy rows, fetched while the MFMA loop of the
// current tile runs (the first version kept 10 x Co/16 blocks per wave and loaded, computed, loaded:
// 107 us alone, 289 us beside the optimizer's HBM traffic).
template <int NB>
__global__ __launch_bounds__(256, 2) void stem7x7_wgrad_kernel(
    const float* __restrict__ x, long x_sn, long x_sc, long x_sh, int H, int W,
    const float* __restrict__ dy, int ldy, float* __restrict__ slab, int Ho, int Wo, int tiles_per_row,
    int tiles_total) {
  constexpr int CO = 16 * NB, DP = CO + 8;
  constexpr int LDSF = kStemPatch + kStemTW * DP;
  constexpr int DYV = (kStemTW * (CO / 4) + 255) / 256;   // dy float4 per thread and tile (8 at Co = 64)
  static_assert(2 * 16 * CO <= LDSF, "cross-wave reduction scratch fits");
  __shared__ __attribute__((aligned(16))) float lds[LDSF];
  float* patch = lds;
  float* dyl = lds + kStemPatch;              // [128][DP]
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, li = lane & 15, kq = lane >> 4;
  const int rhalf = wave >> 1, phalf = wave & 1;           // tap-row half, pixel half
  f32x4 acc[5][NB];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int koff[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) koff[i] = stem_koff((rhalf * 5 + i) * 16 + li);   // (rows >= 147: never stored)
  float pre[kStemPre];
  f32x4 pdy[DYV];
  auto fetch = [&](int tile) {
    const int seg = tile % tiles_per_row;
    const int r = tile / tiles_per_row;
    const int ho = r % Ho, n = r / Ho;
    const int wo0 = seg * kStemTW;
    const float* xb = x + (long)n * x_sn;
#pragma unroll
    for (int u = 0; u < kStemPre; ++u) {
      const int i = t + u * 256;
      pre[u] = i < 3 * 7 * kStemPW ? stem_patch_elem(xb, x_sc, x_sh, H, W, ho, wo0, i) : 0.f;
    }
    const long m0 = ((long)n * Ho + ho) * Wo + wo0;
#pragma unroll
    for (int u = 0; u < DYV; ++u) {
      const int idx = t + u * 256;
      const int row = idx / (CO / 4), q = idx - row * (CO / 4);
      pdy[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (idx < kStemTW * (CO / 4) && wo0 + row < Wo)
        pdy[u] = *reinterpret_cast<const f32x4*>(dy + (m0 + row) * ldy + q * 4);
    }
  };
  int tile = blockIdx.x;
  if (tile < tiles_total) fetch(tile);
  for (; tile < tiles_total; tile += gridDim.x) {
    __syncthreads();                          // the previous tile's operands are no longer read
#pragma unroll
    for (int u = 0; u < kStemPre; ++u) {
      const int i = t + u * 256;
      if (i < 3 * 7 * kStemPW) {
        const int ck = i / kStemPW;
        patch[ck * kStemPWp + (i - ck * kStemPW)] = pre[u];
      }
    }
#pragma unroll
    for (int u = 0; u < DYV; ++u) {
      const int idx = t + u * 256;
      if (idx < kStemTW * (CO / 4)) {
        const int row = idx / (CO / 4), q = idx - row * (CO / 4);
        *reinterpret_cast<f32x4*>(dyl + row * DP + q * 4) = pdy[u];
      }
    }
    __syncthreads();
    if (tile + (int)gridDim.x < tiles_total) fetch(tile + gridDim.x);   // in flight during the MFMA loop
#pragma unroll 2
    for (int pg = 0; pg < 16; ++pg) {
      const int p = phalf * 64 + pg * 4 + kq;   // this lane's pixel of the contraction group
      float b[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) b[j] = dyl[p * DP + j * 16 + li];
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const float a = patch[koff[i] + 2 * p];
#pragma unroll
        for (int j = 0; j < NB; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[j], acc[i][j], 0, 0, 0);
      }
    }
  }
  // the two waves of a pair hold partial sums over different pixels: add them (fixed order) and store
  float* red = lds;                           // [2 pixel halves][16][CO], one tap-row half at a time
  float* out = slab + (long)blockIdx.x * kStemK * CO;
#pragma unroll
  for (int rh = 0; rh < 2; ++rh)
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      __syncthreads();
      if (rhalf == rh) {
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            red[(phalf * 16 + kq * 4 + e) * CO + j * 16 + li] = acc[i][j][e];
      }
      __syncthreads();
      for (int idx = t; idx < 16 * CO; idx += 256) {
        const int row = idx / CO, k = (rh * 5 + i) * 16 + row;
        if (k < kStemK) out[(long)k * CO + (idx - row * CO)] = red[idx] + red[16 * CO + idx];
      }
    }
}

// MEASURED (r04, R50 step under rocprofv3, 1024 x 512 bs 2).  First version (one tile per workgroup,
// load -> compute -> store): forward 107 us, weight gradient 107 us alone / up to 289 us beside the
// optimizer's HBM traffic -- no better than the generic kernels (121 + 17 us statistics pass; 114 + 7 us
// reduce).  This version (persistent workgroups, next tile's operands prefetched into registers
// during the MFMA loop): forward 67 us, weight gradient 82-91 us.  GS_STEM_WGRAD=0 / GS_NO_STEM_KERNEL=1
// put the weight gradient / both back on the generic kernels.
bool stem_wgrad_on() {
  static const bool on = !(getenv("GS_STEM_WGRAD") != nullptr && getenv("GS_STEM_WGRAD")[0] == '0');
  return on;
}

bool stem_conv_ok(const gs_conv_desc* d) {
  static const bool off = getenv("GS_NO_STEM_KERNEL") != nullptr;
  if (off || !d) return false;
  if (d->KH != 7 || d->KW != 7 || d->stride != 2 || d->pad != 3 || d->dil != 1) return false;
  if (d->Ci != 3 || d->Ci_max != 3 || d->x_sw != 1 || d->in_affine) return false;
  if (d->Co != 32 && d->Co != 48 && d->Co != 64) return false;
  if (d->Co_ld < d->Co || (d->Co_ld & 3) || (d->ldy & 3) || d->ldy < d->Co) return false;
  if (d->Wo <= 0 || d->Wo % kStemTW) return false;
  return (long)d->N * d->Ho * d->Wo < (1L << 31) / 64;
}

static inline int stem_wgrad_groups(const gs_conv_desc* d, int* tiles_per_wg) {
  const int tiles = d->N * d->Ho * (d->Wo / kStemTW);
  const int groups = std::min(tiles, 2 * num_cu());      // persistent: tiles dealt round-robin
  *tiles_per_wg = (int)ceil_div(tiles, groups);
  return groups;
}

size_t stem_wgrad_slab_bytes(const gs_conv_desc* d) {
  if (!stem_conv_ok(d) || !stem_wgrad_on()) return 0;
  int tpw;
  return (size_t)stem_wgrad_groups(d, &tpw) * kStemK * d->Co * sizeof(float);
}

// y = conv(x, w); optional per-tile BatchNorm partials (np = tiles, 128 rows each)
int stem_forward(const gs_conv_desc* d, const float* x, const float* w, float* y, float* tile_stats,
                 int* np, hipStream_t st) {
  const int tpr = d->Wo / kStemTW;
  const int tiles = d->N * d->Ho * tpr;
  if (np) *np = tiles;
  const dim3 grid(std::min(tiles, 2 * num_cu())), block(256);
  Plan pl{};
  pl.bm = kStemTW; pl.bn = d->Co; pl.splits = 1; pl.nk_total = pl.nk_per_split = 37;
  pl.tiles_m = tiles; pl.tiles_n = 1;
  note_launch(GS_OP_FORWARD, GS_KLOOP_GENERIC, pl, false, 0,
              2.0 * d->N * d->Ho * (double)d->Wo * d->Co * kStemK);
#define GS_STEM_FWD(NB)                                                                              \
  hipLaunchKernelGGL((stem7x7_fwd_kernel<NB>), grid, block, 0, st, x, (long)d->x_sn, (long)d->x_sc,  \
                     (long)d->x_sh, d->H, d->W, w, d->Co_ld, y, d->ldy, d->Ho, d->Wo, tpr, tiles,    \
                     tile_stats, tiles)
  if (d->Co == 64) GS_STEM_FWD(4);
  else if (d->Co == 48) GS_STEM_FWD(3);
  else GS_STEM_FWD(2);
#undef GS_STEM_FWD
  return launch_status();
}

// dw[k][co] (physical HWIO rows, pitch Co_ld) = sum over pixels; workspace >= stem_wgrad_slab_bytes
int stem_wgrad(const gs_conv_desc* d, const float* x, const float* dy, float* dw, void* workspace,
               size_t workspace_bytes, hipStream_t st) {
  int tpw = 0;
  const int groups = stem_wgrad_groups(d, &tpw);
  const size_t need = (size_t)groups * kStemK * d->Co * sizeof(float);
  if (!workspace || workspace_bytes < need) return GS_E_WORKSPACE;
  const int tpr = d->Wo / kStemTW;
  const int tiles = d->N * d->Ho * tpr;
  float* slab = static_cast<float*>(workspace);
  Plan pl{};
  pl.bm = kStemTW; pl.bn = d->Co; pl.splits = groups; pl.nk_total = tiles; pl.nk_per_split = tpw;
  pl.tiles_m = 1; pl.tiles_n = 1;
  note_launch(GS_OP_WGRAD, GS_KLOOP_GENERIC, pl, false, 0,
              2.0 * d->N * d->Ho * (double)d->Wo * d->Co * kStemK);
#define GS_STEM_WG(NB)                                                                               \
  hipLaunchKernelGGL((stem7x7_wgrad_kernel<NB>), dim3(groups), dim3(256), 0, st, x, (long)d->x_sn,   \
                     (long)d->x_sc, (long)d->x_sh, d->H, d->W, dy, d->ldy, slab, d->Ho, d->Wo, tpr,  \
                     tiles)
  if (d->Co == 64) GS_STEM_WG(4);
  else if (d->Co == 48) GS_STEM_WG(3);
  else GS_STEM_WG(2);
#undef GS_STEM_WG
  int rc = launch_status();
  if (rc != GS_OK) return rc;
  IgemmArgs a{};
  a.slab = slab; a.out = dw;
  a.M = kStemK; a.Nn = d->Co; a.Cs = 3;
  a.o_tap = (long)d->Ci_max * d->Co_ld; a.o_row = d->Co_ld;
  launch_reduce(a, groups, 1, st);
  return launch_status();
}

}  // namespace gs
