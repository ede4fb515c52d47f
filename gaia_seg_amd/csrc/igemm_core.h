// Dynamic (slimmable) convolution for gfx950 as implicit GEMM on the fp32 MFMA
// (v_mfma_f32_16x16x4_f32: exact fp32, bit-for-bit an fmaf chain — the reference path is fp32
// throughout, gaiaseg/models/decode_heads/dynamic_fcn_head.py:81).
//
// Replaces F.conv2d(x, weight[:Co,:Ci], ...) of gaiavision DynConv2d and its autograd backward at
// the call sites of gaiaseg/models/utils/dynamic_res_layer.py:84-125,
// gaiaseg/models/backbones/dynamic_resnet.py:255-302 and the decode heads
// (dynamic_fcn_head.py:76-126, dynamic_psp_head.py:53-59,123,140-147, dynamic_uper_head.py:40-79).
//
// Layout: activations NHWC (pixel stride ld), weights physical [KH][KW][Ci_max][Co_ld]; the active
// leading slice [:Ci,:Co] is read in place (runtime Ci/Co — one binary for all 85k subnets).
//
// Three GEMM views share one tile engine (256 threads = 4 waves stacked along M, BK = 16):
//   forward : Y[m, co]  = sum_{tap,ci} Xg[m,(tap,ci)] * W[tap][ci][co]      m = output pixel
//   dgrad   : dX[m, ci] = sum_{tap,co} dYg[m,(tap,co)] * W[tap][ci][co]     m = input pixel
//   wgrad   : dW[(tap,ci), co] = sum_m Xg[m,(tap,ci)] * dY[m, co]           K = pixels (split-K)
// A and B tiles are staged global -> registers -> LDS (issue-early / write-late double buffer, one
// barrier per K step); LDS images are k-major with a per-4-row skew so that both the transposed
// stores and the MFMA fragment reads (lane l reads [k = l>>4][i = l&15]) are bank-conflict free.
#pragma once
#include <algorithm>
#include <cstdio>
#include <unordered_map>
#include <hip/hip_ext.h>
#include "common.h"

namespace gs {

constexpr int BK = 16;
constexpr int NT = 256;

struct IgemmArgs {
  const float* src;    // gathered operand (x for forward / wgrad, dy for dgrad)
  const float* dense;  // dense operand (w for forward / dgrad, dy for wgrad)
  float* out;
  float* slab;         // split-K partials [splits][M][Nn] (NULL when splits == 1)
  const float* bias;
  const float* addend;
  long s_n, s_h, s_w, s_c;  // element strides of src
  int Hs, Ws, Cs;           // src spatial dims, channels gathered per tap
  int Hp, Wp;               // pixel-row space (forward/wgrad: output pixels, dgrad: input pixels)
  int npix;                 // Nb * Hp * Wp
  int KW, taps;
  int mul_h, mul_w, base_h, base_w, step_h, step_w, div_h, div_w;
  long d_tap;               // dense tap stride (weights)
  int d_row;                // forward: stride of ci rows; dgrad: stride of ci rows (n index);
                            // wgrad: pixel stride of dy
  int n_lim;                // forward/wgrad: number of dense columns that may be read (mult. of 4)
  int M, Nn, Ktot;          // GEMM sizes; wgrad: M = taps*Cs, Ktot = npix
  int ld_out, ld_add;
  long o_tap;               // wgrad: dw tap stride
  int o_row;                // wgrad: dw ci-row stride
  int nk_total, nk_per_split;
  int accumulate;
  int tiles_m, tiles_n;
  unsigned src_bytes, dense_bytes;  // extents for the bounds-checked buffer loads (0 = unknown)
  // fast kernel: runtime tap grid (sub-sampled for the parity classes of a strided dgrad)
  int kh_n, kw_n;           // taps of this launch: kh_n x kw_n (<= 3 x 3)
  long d_tap_h, d_tap_w;    // dense offset per tap row / tap column
  // output row lattice: row m = (n, hq, wq) -> pixel (n, hq*o_s + o_ph, wq*o_s + o_pw) of an
  // o_H x o_W image (o_s == 0: rows are dense pixels)
  int o_s, o_ph, o_pw, o_Hq, o_Wq, o_H, o_W;
  // fast kernels: 1-D grid of tiles x splits, split-major; tile_order 1 = tm fastest
  int nsplits, tile_order;
  // forward only: per-tile BatchNorm partial sums written by the epilogue (NULL = off), quad-major
  // {s1, s2, shift}[C/4][tiles_m] float4 with shift = the tile's first row (see rows_epilogue)
  float* tile_stats;
  // fast forward / wgrad kernels: when set, the gathered operand is relu(bn(src)) evaluated on the
  // fly from the producer BatchNorm's coefficients [scale | beta | mean | invstd][Cs] (the
  // normalised activation is never stored; padding stays exactly zero)
  const float* a_coeffs;
  // dgrad only: the output dX is the gradient of z = relu(bn(y) [+ residual]) of the PRODUCER layer.
  // bw_mode != 0 folds that BatchNorm's backward reduction into this epilogue: the tile (after the
  // optional accumulate) is masked with the ReLU mask (mode 1: (y - mean) * scale + beta > 0,
  // mode 2: bw_act > 0, mode 3: the same mask as bytes, one per channel quad, written by the
  // producer's bn_apply), the masked gradient g is what gets stored, and per-tile sums
  // {sum g, sum g * xhat} go to bw_part ([2][Nn/4][tiles_m] float4, quad-major) for
  // sum_partials_kernel — the separate bn_bwd_partial pass over dz and y disappears.
  const float* bw_y;       // the producer BN's input, pixel stride bw_ldy
  const float* bw_act;     // mode 2: its post-activation output, pixel stride bw_ldact
  const float* bw_coeffs;  // [scale | beta | mean | invstd][Nn]
  float* bw_part;
  int bw_ldy, bw_ldact, bw_mode;
  const unsigned char* bw_mask;   // mode 3: [pixels][bw_ldmask] bytes, bit e of byte q <=> channel 4q+e
  int bw_ldmask;
  // fast row kernels, split-K combined INSIDE the launch (see splitk_publish): one arrival counter
  // per output tile, zero at rest (the last arriver puts it back to zero); NULL = the slabs are summed
  // by a separate reduce launch
  unsigned* tickets;
  unsigned slab_bytes;            // extent of `slab` for the write-through buffer stores / loads
  // fast row kernels with per-tile partials (tile_stats / bw_part) over at most a few hundred row
  // tiles: the partials are merged INSIDE the launch by the last workgroup of every column tile
  // (column_finalize_*), which writes the BatchNorm coefficients (forward) or the BatchNorm-backward
  // sums (dgrad) itself — bn_tile_finalize / sum_partials launches disappear.  col_tickets: one
  // arrival counter per column tile, zero at rest; NULL = the separate launch.
  unsigned* col_tickets;
  const float* fin_gamma;         // forward: the BatchNorm whose statistics the tiles carry
  const float* fin_beta;
  float* fin_running_mean;        // NULL: no running-statistics update
  float* fin_running_var;
  float* fin_coeffs;              // [scale | beta | mean | invstd][Nn]
  float fin_eps, fin_momentum;
  int fin_bm, fin_n_last;         // rows of a full tile / of the last tile
  double fin_inv_M;               // 1 / rows
  float* fin_bw_sums;             // dgrad: {sum g, sum g * xhat}[2][Nn]
};

constexpr int kAffMaxC = 640;   // widest gathered operand of the supernet (stage-4 planes)

// relu((v - mean) * scale + beta): the expression of bn_apply_kernel / masked_grad (norm.hip), so
// that the forward value, the backward mask and the fused loaders agree bit for bit
__device__ __forceinline__ f32x4 bn_relu_affine(f32x4 v, f32x4 mean, f32x4 scale, f32x4 beta) {
  v = (v - mean) * scale + beta;
  v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f);
  v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
  return v;
}

// pixel index of GEMM row m in the output tensor
__device__ __forceinline__ long out_pixel(const IgemmArgs& p, int m) {
  if (p.o_s == 0) return m;
  const int hwq = p.o_Hq * p.o_Wq;
  const int n = m / hwq;
  const int rem = m - n * hwq;
  const int hq = rem / p.o_Wq;
  const int wq = rem - hq * p.o_Wq;
  return ((long)n * p.o_H + (hq * p.o_s + p.o_ph)) * p.o_W + (wq * p.o_s + p.o_pw);
}

// k-major LDS image with skew: element (k, i) at k*P + 8*(k>>2) + i, P % 32 == 16.
//  * MFMA fragment read (ds_read_b32, lanes 0-31 hold k = 4ks + {0,1}, i = 0..15): rows k and k+1
//    start 16 banks apart -> 32 distinct banks.
//  * transposed store (lane -> (i = t>>2, kq = t&3), element j): rows kq*4+j start 8*kq (+16*(j&1))
//    banks apart -> the 8 consecutive i of 4 kq cover 32 distinct banks.
template <int BM, int BN>
struct Tile {
  static constexpr int PA = BM + 16;
  static constexpr int PB = ((BN + 31) / 32) * 32 + 16;
  static constexpr int A_SZ = BK * PA + 32;
  static constexpr int B_SZ = BK * PB + 32;
  static constexpr int STAGE = A_SZ + B_SZ;
  static constexpr int CCH = BN < 64 ? BN : 64;  // epilogue column chunk
  static constexpr int PC = CCH + 4;
  static constexpr int C_SZ = BM * PC;
  // (+512 floats behind the C image: scratch of the epilogue's per-tile BatchNorm partial sums)
  static constexpr int LDSF = (2 * STAGE > C_SZ + 512) ? 2 * STAGE : C_SZ + 512;
  // four stages for the paired K loop (two K steps per barrier)
  static constexpr int LDSF2 = (4 * STAGE > C_SZ + 512) ? 4 * STAGE : C_SZ + 512;
  static constexpr int WM = BM / 4;   // rows per wave
  static constexpr int TM = WM / 16;  // 16x16 tiles per wave along M
  static constexpr int TN = BN / 16;
  static constexpr int BV = (BK * BN / 4 + NT - 1) / NT;  // dense float4 per thread
};

// Which output columns the wave's TN 16-column MFMA blocks own.  The B fragment of block j is one
// float per lane (lane l = its column inside the block); were block j the columns [16j, 16j+16)
// every k-group would cost TN ds_read_b32.  Instead the blocks are grouped (fours, then a pair, then
// a single) and a group of V blocks owns V*16 consecutive columns INTERLEAVED: block j of the group
// takes columns base + V*l + (j - first), so one ds_read_b128 / b64 per lane fetches the fragments
// of all V blocks, and the epilogue writes V consecutive accumulator columns with one vector store.
template <int TN>
struct ColGroups {
  static constexpr int kFull = (TN / 4) * 4;
  __host__ __device__ static constexpr int width(int j) {
    return j < kFull ? 4 : ((TN - kFull >= 2 && j - kFull < 2) ? 2 : 1);
  }
  __host__ __device__ static constexpr int first(int j) {
    return j < kFull ? (j / 4) * 4 : ((TN - kFull >= 2 && j - kFull < 2) ? kFull : j);
  }
  // first column of block j's group (groups never straddle a 64-column epilogue chunk)
  __host__ __device__ static constexpr int base(int j) { return 16 * first(j); }
};
typedef float f32x2 __attribute__((ext_vector_type(2)));

// B fragments of one k-group: `row` points at column 0 of the lane's k row in the k-major image
template <int TN>
__device__ __forceinline__ void read_b_fragments(const float* __restrict__ row, int li,
                                                 float (&b)[TN]) {
  using G = ColGroups<TN>;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    if (G::first(j) != j) continue;
    if (G::width(j) == 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(row + G::base(j) + 4 * li);
      b[j] = v[0]; b[j + 1 < TN ? j + 1 : j] = v[1]; b[j + 2 < TN ? j + 2 : j] = v[2];
      b[j + 3 < TN ? j + 3 : j] = v[3];
    } else if (G::width(j) == 2) {
      const f32x2 v = *reinterpret_cast<const f32x2*>(row + G::base(j) + 2 * li);
      b[j] = v[0]; b[j + 1 < TN ? j + 1 : j] = v[1];
    } else {
      b[j] = row[G::base(j) + li];
    }
  }
}

template <int BM, int BN, bool NOREAD = false>
__device__ __forceinline__ void mfma_stage(const float* __restrict__ As, const float* __restrict__ Bs,
                                           f32x4 (&acc)[Tile<BM, BN>::TM][Tile<BM, BN>::TN],
                                           int wave, int lane) {
  using T = Tile<BM, BN>;
  const int kk = lane >> 4, li = lane & 15;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int ko = ks * 4 + kk;
    float a[T::TM], b[T::TN];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
      a[i] = NOREAD ? (float)(lane + i) : As[ko * T::PA + 8 * ks + wave * T::WM + i * 16 + li];
    if constexpr (NOREAD) {
#pragma unroll
      for (int j = 0; j < T::TN; ++j) b[j] = (float)(lane - j);
    } else {
      read_b_fragments<T::TN>(Bs + ko * T::PB + 8 * ks, li, b);
    }
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
      for (int j = 0; j < T::TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
  }
}

// Accumulators -> LDS (column chunk `ch`) in [row][col] order.  C/D map of the 16x16 MFMA:
// row = (lane >> 4) * 4 + reg, column inside block j = lane & 15, i.e. (ColGroups) global column
// base(j) + width(j) * (lane & 15) + (j - first(j)): one vector store per group and register.
// QUAD (bf16x3 loop, 64x64 tiles): the four accumulator blocks of a wave are a 2 x 2 arrangement
// inside the wave's 32 x 32 quadrant of the tile (wave = 2 * row half + column half; block q = 2 *
// row block + column block) instead of a 16 x 64 stripe.
template <int BM, int BN, bool QUAD = false>
__device__ __forceinline__ void acc_to_lds(float* __restrict__ Cs,
                                           const f32x4 (&acc)[Tile<BM, BN>::TM][Tile<BM, BN>::TN],
                                           int ch, int wave, int lane) {
  using T = Tile<BM, BN>;
  using G = ColGroups<T::TN>;
  if constexpr (QUAD) {
    static_assert(BM == 64 && BN == 64, "quadrant layout: 64x64 tiles");
    const int r0 = (wave >> 1) * 32 + (lane >> 4) * 4, c0 = (wave & 1) * 32 + (lane & 15);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        Cs[(r0 + (q >> 1) * 16 + r) * T::PC + c0 + (q & 1) * 16] = acc[0][q][r];
    return;
  }
  constexpr int TPC = T::CCH / 16;  // 16-col blocks per chunk
  const int li = lane & 15;
#pragma unroll
  for (int i = 0; i < T::TM; ++i)
#pragma unroll
    for (int jj = 0; jj < TPC; ++jj) {
      const int j = ch * TPC + jj;
      if (j < T::TN && G::first(j) == j) {
        const int c0 = G::base(j) - ch * T::CCH;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* dst = Cs + (wave * T::WM + i * 16 + (lane >> 4) * 4 + r) * T::PC + c0;
          if (G::width(j) == 4) {
            *reinterpret_cast<f32x4*>(dst + 4 * li) =
                f32x4{acc[i][j][r], acc[i][j + 1 < T::TN ? j + 1 : j][r],
                      acc[i][j + 2 < T::TN ? j + 2 : j][r], acc[i][j + 3 < T::TN ? j + 3 : j][r]};
          } else if (G::width(j) == 2) {
            *reinterpret_cast<f32x2*>(dst + 2 * li) =
                f32x2{acc[i][j][r], acc[i][j + 1 < T::TN ? j + 1 : j][r]};
          } else {
            dst[li] = acc[i][j][r];
          }
        }
      }
    }
}

// ------------------------------------------------------------------------------------------
// forward / dgrad: GEMM rows are pixels, the gathered operand is A.
// ------------------------------------------------------------------------------------------
// KS: compile-time kernel size (1 or 3; 0 = runtime KW) — also tags the kernel name in profiles
template <int BM, int BN, bool BTRANS, bool DIVS, bool SCALAR, int KS>
__global__ __launch_bounds__(NT) void igemm_rows_kernel(const IgemmArgs p) {
  using T = Tile<BM, BN>;
  __shared__ __attribute__((aligned(16))) float lds[T::LDSF];
  constexpr int AS = BM / 64;  // A float4 slots per thread

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int ntiles = p.tiles_m * p.tiles_n;
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int split = blockIdx.y;
  const int kt0 = split * p.nk_per_split;
  const int kt1 = min(kt0 + p.nk_per_split, p.nk_total);

  // per-thread row geometry (constant over the K loop)
  const int kq = t & 3;
  int hb[AS], wb[AS];
  long nb[AS];
  bool rv[AS];
  const int hw = p.Hp * p.Wp;
#pragma unroll
  for (int s = 0; s < AS; ++s) {
    const int m = m0 + (t >> 2) + 64 * s;
    rv[s] = m < p.M;
    const int mm = rv[s] ? m : 0;
    const int n = mm / hw;
    const int rem = mm - n * hw;
    const int hp = rem / p.Wp;
    const int wp = rem - hp * p.Wp;
    hb[s] = hp * p.mul_h + p.base_h;
    wb[s] = wp * p.mul_w + p.base_w;
    nb[s] = (long)n * p.s_n;
  }

  // block-uniform (tap, channel) of the first k of the current K step
  int tap_u = (kt0 * BK) / p.Cs;
  int c_u = kt0 * BK - tap_u * p.Cs;

  f32x4 ra[AS];
  f32x4 rb[T::BV];
  f32x4 acc[T::TM][T::TN];
#pragma unroll
  for (int i = 0; i < T::TM; ++i)
#pragma unroll
    for (int j = 0; j < T::TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto gather = [&](int s, int tap, int c) -> const float* {
    // returns the address of src element for row slot s / (tap, c), or nullptr when padded
    const int kwid = KS ? KS : p.KW;
    const int kh = KS == 1 ? 0 : tap / kwid, kw = KS == 1 ? 0 : tap - kh * kwid;
    const int hn = hb[s] + kh * p.step_h, wn = wb[s] + kw * p.step_w;
    int hi = hn, wi = wn;
    bool ok = rv[s] && tap < p.taps;
    if constexpr (DIVS) {
      ok = ok && hn >= 0 && wn >= 0;
      hi = hn / p.div_h;
      wi = wn / p.div_w;
      ok = ok && hi * p.div_h == hn && wi * p.div_w == wn;
    }
    ok = ok && (unsigned)hi < (unsigned)p.Hs && (unsigned)wi < (unsigned)p.Ws;
    return ok ? p.src + nb[s] + (long)hi * p.s_h + (long)wi * p.s_w + (long)c * p.s_c : nullptr;
  };

  auto load_tiles = [&](int kt) {
    // ---- A (gathered) ----
    if constexpr (!SCALAR) {
      int c = c_u + kq * 4, tap = tap_u;
      while (c >= p.Cs) { c -= p.Cs; ++tap; }
#pragma unroll
      for (int s = 0; s < AS; ++s) {
        const float* q = gather(s, tap, c);
        ra[s] = q ? *reinterpret_cast<const f32x4*>(q) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    } else {
#pragma unroll
      for (int s = 0; s < AS; ++s) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = kt * BK + kq * 4 + e;
          const int tap = k / p.Cs, c = k - tap * p.Cs;
          const float* q = gather(s, tap, c);
          ra[s][e] = q ? *q : 0.f;
        }
      }
    }
    // ---- B (dense) ----
#pragma unroll
    for (int r = 0; r < T::BV; ++r) {
      const int idx = t + NT * r;
      rb[r] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (idx < BK * BN / 4) {
        if constexpr (!BTRANS) {
          const int kr = idx / (BN / 4), nq = idx - kr * (BN / 4);
          int c, tap;
          if constexpr (SCALAR) {
            const int k = kt * BK + kr;
            tap = k / p.Cs;
            c = k - tap * p.Cs;
          } else {
            c = c_u + kr;
            tap = tap_u;
            while (c >= p.Cs) { c -= p.Cs; ++tap; }
          }
          const int col = n0 + nq * 4;
          if (tap < p.taps && col < p.n_lim)
            rb[r] = *reinterpret_cast<const f32x4*>(p.dense + (long)tap * p.d_tap +
                                                    (long)c * p.d_row + col);
        } else {
          const int nrow = idx >> 2, kq2 = idx & 3;
          int c = c_u + kq2 * 4, tap = tap_u;
          while (c >= p.Cs) { c -= p.Cs; ++tap; }
          const int ng = n0 + nrow;
          if (tap < p.taps && ng < p.Nn)
            rb[r] = *reinterpret_cast<const f32x4*>(p.dense + (long)tap * p.d_tap +
                                                    (long)ng * p.d_row + c);
        }
      }
    }
    // advance the uniform (tap, c) to the next K step
    c_u += BK;
    while (c_u >= p.Cs) { c_u -= p.Cs; ++tap_u; }
  };

  auto store_tiles = [&](int buf) {
    float* As = lds + buf * T::STAGE;
    float* Bs = As + T::A_SZ;
#pragma unroll
    for (int s = 0; s < AS; ++s) {
      const int row = (t >> 2) + 64 * s;
#pragma unroll
      for (int j = 0; j < 4; ++j) As[(kq * 4 + j) * T::PA + 8 * kq + row] = ra[s][j];
    }
#pragma unroll
    for (int r = 0; r < T::BV; ++r) {
      const int idx = t + NT * r;
      if (idx < BK * BN / 4) {
        if constexpr (!BTRANS) {
          const int kr = idx / (BN / 4), nq = idx - kr * (BN / 4);
          *reinterpret_cast<f32x4*>(&Bs[kr * T::PB + 8 * (kr >> 2) + nq * 4]) = rb[r];
        } else {
          const int nrow = idx >> 2, kq2 = idx & 3;
#pragma unroll
          for (int j = 0; j < 4; ++j) Bs[(kq2 * 4 + j) * T::PB + 8 * kq2 + nrow] = rb[r][j];
        }
      }
    }
  };

  if (kt0 < kt1) {
    load_tiles(kt0);
    store_tiles(0);
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
      const int buf = (kt - kt0) & 1;
      const bool more = kt + 1 < kt1;
      if (more) load_tiles(kt + 1);  // global loads in flight under the MFMAs
      mfma_stage<BM, BN>(lds + buf * T::STAGE, lds + buf * T::STAGE + T::A_SZ, acc, wave, lane);
      if (more) store_tiles(buf ^ 1);
      __syncthreads();
    }
  }

  // ---- epilogue: accumulators -> LDS -> coalesced float4 rows ----
  float* Cs = lds;
  constexpr int NCH = (BN + T::CCH - 1) / T::CCH;
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    if (ch > 0) __syncthreads();
    acc_to_lds<BM, BN>(Cs, acc, ch, wave, lane);
    __syncthreads();
    constexpr int QPR = T::CCH / 4;  // float4 per row of a full chunk
    for (int idx = t; idx < BM * QPR; idx += NT) {
      const int row = idx / QPR, q = idx - row * QPR;
      const int m = m0 + row;
      const int col = n0 + ch * T::CCH + q * 4;
      if (ch * T::CCH + q * 4 < BN && m < p.M && col < p.Nn) {
        f32x4 v = *reinterpret_cast<const f32x4*>(&Cs[row * T::PC + q * 4]);
        if (p.slab) {
          *reinterpret_cast<f32x4*>(p.slab + ((long)blockIdx.y * p.M + m) * p.Nn + col) = v;
        } else {
          if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + col);
          if (p.addend) v += *reinterpret_cast<const f32x4*>(p.addend + (long)m * p.ld_add + col);
          float* o = p.out + (long)m * p.ld_out + col;
          if (p.accumulate) v += *reinterpret_cast<const f32x4*>(o);
          *reinterpret_cast<f32x4*>(o) = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Split-K combined inside the launch (forward / dgrad row kernels).  The `nsplits` workgroups of an
// output tile each publish their partial tile to the slab; the one that draws the last arrival
// ticket sums the slabs in split order (the order of splitk_reduce_kernel: bit-identical) and runs
// the ordinary epilogue on the sum — output store, bias / addend / accumulate, BatchNorm tile
// statistics, BatchNorm-backward partials — so the separate reduce launch (6-14 us plus a kernel
// boundary, 60-110 of them per training step) disappears.
// Visibility does not depend on which XCD / CU the workgroups land on:
//  * every slab store is a write-through (sc1) buffer store; EVERY storing wave drains its stores
//    (s_waitcnt vmcnt(0)) before the workgroup barrier, and only then lane 0 adds to the counter
//    (relaxed, agent scope: performed at the L2 all XCDs share through the fabric);
//  * the last arriver reads the slabs ONLY with sc1 buffer loads (they bypass the CU's L1, which is
//    never refreshed by other CUs' stores), its own partial included, so no acquire is needed;
//  * nobody waits on anybody: a workgroup that is not last simply exits, so residency is irrelevant.
// The counter is back at zero when the launch ends (the last arriver resets it: all `nsplits`
// arrivals are in by then and nothing else touches it until the next launch on this stream).
// ------------------------------------------------------------------------------------------
constexpr int kAuxSc1 = 16;   // aux bit of the raw buffer intrinsics: sc1 (write-through / L1 bypass)

template <int BM, int BN, bool QUAD>
__device__ __forceinline__ bool splitk_publish(
    const IgemmArgs& p, float* lds, const f32x4 (&acc)[Tile<BM, BN>::TM][Tile<BM, BN>::TN], int m0,
    int n0, int t, int wave, int lane, int split, int tile) {
  using T = Tile<BM, BN>;
  float* Cs = lds;
  constexpr int NCH = (BN + T::CCH - 1) / T::CCH;
  constexpr int QPR = T::CCH / 4;
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(p.slab, 0, p.slab_bytes, 0x00020000);
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    if (ch > 0) __syncthreads();
    acc_to_lds<BM, BN, QUAD>(Cs, acc, ch, wave, lane);
    __syncthreads();
    // (branch-free: an element outside the tile goes to an out-of-range offset, which a buffer store
    // drops)
    constexpr int NI = (BM * QPR + NT - 1) / NT;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int idx = t + i * NT;
      const int row = idx / QPR, q = idx - row * QPR;
      const int m = m0 + row;
      const int col = n0 + ch * T::CCH + q * 4;
      const bool ok = idx < BM * QPR && ch * T::CCH + q * 4 < BN && m < p.M && col < p.Nn;
      const int rr = idx < BM * QPR ? row : 0;
      const f32x4 v = *reinterpret_cast<const f32x4*>(&Cs[rr * T::PC + q * 4]);
      const unsigned off = ok ? (unsigned)((((long)split * p.M + m) * p.Nn + col) * 4) : 0xFFFFFFF0u;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, off, 0, kAuxSc1);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave, before the barrier
  __syncthreads();
  unsigned* flag = reinterpret_cast<unsigned*>(lds + T::C_SZ);
  if (t == 0) {
    const unsigned tk = __hip_atomic_fetch_add(p.tickets + tile, 1u, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
    if (tk == (unsigned)(p.nsplits - 1))
      __hip_atomic_store(p.tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = tk;
  }
  __syncthreads();
  const bool last = *flag == (unsigned)(p.nsplits - 1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // no instruction: keeps the slab loads below
  return last;
}

// column chunk `ch` of the tile = fixed-order sum of the slabs, into the LDS C image (the layout
// acc_to_lds leaves).  Latency, not bandwidth, is what this costs (one workgroup, 16 KB per slab, every
// load a trip past the L1 to the far side of the fabric), so the loop is branch-free — an element
// outside the tile or a slab index past nsplits reads from an out-of-range offset, which a buffer load
// answers with zero without touching memory — and a thread has its NI quads x kZB slabs in flight at
// once (a branch around a load would make hipcc drain vmcnt per element).
template <int BM, int BN>
__device__ __forceinline__ void slabs_to_lds(const IgemmArgs& p, float* __restrict__ Cs, int ch, int m0,
                                             int n0, int t) {
  using T = Tile<BM, BN>;
  constexpr int QPR = T::CCH / 4;
  constexpr int NI = (BM * QPR + NT - 1) / NT;
  // slabs in flight per quad: 4 (16 loads, 64 registers) where the tile's registers leave room for it
  // at three workgroups per CU (a split launch is planned at 2.5-3 per CU); the 80-wide tiles are at
  // 156 registers already and 168 is the limit for three waves per SIMD
  constexpr int kZB = BN > 64 ? 2 : 4;
  constexpr unsigned kOOB = 0xFFFFFFF0u;
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(p.slab, 0, p.slab_bytes, 0x00020000);
  const unsigned zstride = (unsigned)((long)p.M * p.Nn * 4);
  unsigned off[NI];
  f32x4 v[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int idx = t + i * NT;
    const int row = idx / QPR, q = idx - row * QPR;
    const int m = m0 + row;
    const int col = n0 + ch * T::CCH + q * 4;
    const bool ok = idx < BM * QPR && ch * T::CCH + q * 4 < BN && m < p.M && col < p.Nn;
    off[i] = ok ? (unsigned)(((long)m * p.Nn + col) * 4) : kOOB;
    v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int z0 = 0; z0 < p.nsplits; z0 += kZB) {
    u32x4 r[kZB][NI];
#pragma unroll
    for (int zz = 0; zz < kZB; ++zz)
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const bool ok = z0 + zz < p.nsplits && off[i] != kOOB;
        const unsigned o = ok ? off[i] + (unsigned)(z0 + zz) * zstride : kOOB;
        r[zz][i] = __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, kAuxSc1);
      }
#pragma unroll
    for (int zz = 0; zz < kZB; ++zz)
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        // (the first slab is taken as it is, like splitk_reduce_kernel: 0 + x would lose a -0)
        const f32x4 x = __builtin_bit_cast(f32x4, r[zz][i]);
        const f32x4 sum = v[i] + x;
        const bool first = z0 + zz == 0, live = z0 + zz < p.nsplits;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[i][e] = first ? x[e] : (live ? sum[e] : v[i][e]);
      }
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int idx = t + i * NT;
    const int row = idx / QPR, q = idx - row * QPR;
    if (idx < BM * QPR) *reinterpret_cast<f32x4*>(&Cs[row * T::PC + q * 4]) = v[i];
  }
}

// ------------------------------------------------------------------------------------------
// Per-tile partials merged inside the launch (IgemmArgs::col_tickets).  Every workgroup that has
// written its tile's partials (write-through stores, drained, then the workgroup barrier) adds to the
// arrival counter of its COLUMN tile; the one that draws tiles_m - 1 reads the partials of all row
// tiles of its columns back (sc1 loads: past the L1, wherever the writers ran) and finishes them.
// Same hand-off as splitk_publish; nobody waits.  One workgroup reads tiles_m * BN * 12 (forward) or
// * 8 (dgrad) bytes: the host enables this up to a few hundred row tiles (stages 2-4 of the
// backbone, the heads), where the separate launch is nothing but latency (5-7 us + a kernel boundary).
// MEASURED (r04, profiles/r04_splitk_inkernel.md): neutral on the training step — every workgroup,
// not only the last, pays the drain + counter round trip (~3 us) before it can leave its CU slot, the
// last one another ~3 us of dependent loads and double arithmetic, and the conv2 launches (the
// headline kernel) run 4-6 % longer — so it is OFF by default (GS_COL_FINALIZE=1 turns it on).
// ------------------------------------------------------------------------------------------
template <int BM, int BN>
__device__ __forceinline__ bool column_arrive(const IgemmArgs& p, float* lds, int n0, int t) {
  using T = Tile<BM, BN>;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave: its partial stores are out
  __syncthreads();
  unsigned* flag = reinterpret_cast<unsigned*>(lds + T::C_SZ);
  if (t == 0) {
    const int tn = n0 / BN;
    const unsigned tk = __hip_atomic_fetch_add(p.col_tickets + tn, 1u, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
    if (tk == (unsigned)(p.tiles_m - 1))
      __hip_atomic_store(p.col_tickets + tn, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = tk;
  }
  __syncthreads();
  const bool last = *flag == (unsigned)(p.tiles_m - 1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  return last;
}

__device__ __forceinline__ double shfl_xor_d(double v, int mask) {
  return __shfl_xor(v, mask, 64);
}

// forward: the arithmetic of bn_tile_finalize_kernel (norm.hip) for the channels [n0, n0 + BN) —
// one pass over the tiles' {s1, s2, shift}, re-centred on tile 0's shift, in double; sixteen threads
// share a channel quad (tiles p = sub, sub + 16, ...) and are combined with a fixed xor tree; sixteen
// quads at a time.  Kept small in registers (two tiles = six loads in flight per thread): this code
// sits behind every forward launch's epilogue, also the many-round ones whose occupancy matters.
// (No private array is indexed with a run-time value here: see the NOTE in bn_tile_finalize_kernel.)
template <int BN>
__device__ __forceinline__ void column_finalize_stats(const IgemmArgs& p, int n0, int t) {
  const int C = p.Nn, C4 = C >> 2, np = p.tiles_m;
  const int nq = (min(BN, C - n0) + 3) >> 2;
  const int sub = t & 15;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      p.tile_stats, 0, (unsigned)((long)3 * C * np * 4), 0x00020000);
#pragma unroll 1
  for (int qb = 0; qb < nq; qb += 16) {
    const int qi = qb + (t >> 4);
    const bool act = qi < nq;
    const int q = (n0 >> 2) + (act ? qi : 0);
    const unsigned o1 = (unsigned)(((0L * C4 + q) * np) * 16);
    const unsigned o2 = (unsigned)(((1L * C4 + q) * np) * 16);
    const unsigned o3 = (unsigned)(((2L * C4 + q) * np) * 16);
    const f32x4 ref4 =
        __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, o3, 0, kAuxSc1));
    const int fc = q * 4 + (sub & 3);
    float g_pre = 1.f, be_pre = 0.f, rm_pre = 0.f, rv_pre = 0.f;
    if (sub < 4) {
      if (p.fin_gamma) g_pre = p.fin_gamma[fc];
      if (p.fin_beta) be_pre = p.fin_beta[fc];
      if (p.fin_running_mean) rm_pre = p.fin_running_mean[fc];
      if (p.fin_running_var) rv_pre = p.fin_running_var[fc];
    }
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    // (the 80-wide pair-loop tiles are at 156 registers: 168 is the limit for three waves per SIMD)
    constexpr int kUnroll = BN > 64 ? 1 : 2;
#pragma unroll kUnroll
    for (int tp = sub; tp < np; tp += 16) {
      const double n = (double)(tp == np - 1 ? p.fin_n_last : p.fin_bm);
      const unsigned po = (unsigned)tp * 16u;
      const f32x4 s1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, o1 + po, 0, kAuxSc1));
      const f32x4 s2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, o2 + po, 0, kAuxSc1));
      const f32x4 shv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, o3 + po, 0, kAuxSc1));
#define GS_ACC(E, A, B)                                              \
      {                                                              \
        const double d = (double)shv[E] - (double)ref4[E];           \
        A += (double)s1[E] + n * d;                                  \
        B += (double)s2[E] + d * (2.0 * (double)s1[E] + n * d);      \
      }
      GS_ACC(0, a0, b0) GS_ACC(1, a1, b1) GS_ACC(2, a2, b2) GS_ACC(3, a3, b3)
#undef GS_ACC
    }
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
      a0 += shfl_xor_d(a0, m); a1 += shfl_xor_d(a1, m); a2 += shfl_xor_d(a2, m); a3 += shfl_xor_d(a3, m);
      b0 += shfl_xor_d(b0, m); b1 += shfl_xor_d(b1, m); b2 += shfl_xor_d(b2, m); b3 += shfl_xor_d(b3, m);
    }
    if (act && sub < 4) {
      const double S1 = sub == 0 ? a0 : sub == 1 ? a1 : sub == 2 ? a2 : a3;
      const double S2 = sub == 0 ? b0 : sub == 1 ? b1 : sub == 2 ? b2 : b3;
      const float rf = sub == 0 ? ref4[0] : sub == 1 ? ref4[1] : sub == 2 ? ref4[2] : ref4[3];
      const double MU = (double)rf + S1 * p.fin_inv_M;
      double var = (S2 - S1 * S1 * p.fin_inv_M) * p.fin_inv_M;
      if (var < 0.0) var = 0.0;
      const double invstd = 1.0 / sqrt(var + (double)p.fin_eps);
      float* co = p.fin_coeffs;
      co[fc] = (float)((double)g_pre * invstd);
      co[C + fc] = be_pre;
      co[2 * C + fc] = (float)MU;
      co[3 * C + fc] = (float)invstd;
      if (p.fin_running_mean)
        p.fin_running_mean[fc] = (1.f - p.fin_momentum) * rm_pre + p.fin_momentum * (float)MU;
      if (p.fin_running_var) {
        const double M = (double)p.M;
        const double unbiased = p.M > 1 ? var * M / (M - 1.0) : var;
        p.fin_running_var[fc] = (1.f - p.fin_momentum) * rv_pre + p.fin_momentum * (float)unbiased;
      }
    }
  }
}

// dgrad: sums[k][c] = sum over the row tiles of bw_part[k][c / 4][tile] (k = 0: sum g, 1: sum g * xhat)
// for the channels [n0, n0 + BN) — sum_partials_kernel's job; eight threads per (k, quad), double,
// 32 (k, quad) items at a time.
template <int BN>
__device__ __forceinline__ void column_finalize_bwsums(const IgemmArgs& p, int n0, int t) {
  const int C = p.Nn, C4 = C >> 2, np = p.tiles_m;
  const int nq = (min(BN, C - n0) + 3) >> 2;
  const int sub = t & 7;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      p.bw_part, 0, (unsigned)((long)2 * C * np * 4), 0x00020000);
#pragma unroll 1
  for (int ib = 0; ib < 2 * nq; ib += 32) {
    const int item = ib + (t >> 3);              // item = k * nq + quad
    const bool act = item < 2 * nq;
    const int k = act ? item / nq : 0;
    const int q = (n0 >> 2) + (act ? item - k * nq : 0);
    const unsigned o = (unsigned)((((long)k * C4 + q) * np) * 16);
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll 4
    for (int tp = sub; tp < np; tp += 8) {
      const f32x4 v = __builtin_bit_cast(
          f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, o + (unsigned)tp * 16u, 0, kAuxSc1));
      a0 += (double)v[0]; a1 += (double)v[1]; a2 += (double)v[2]; a3 += (double)v[3];
    }
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
      a0 += shfl_xor_d(a0, m); a1 += shfl_xor_d(a1, m); a2 += shfl_xor_d(a2, m); a3 += shfl_xor_d(a3, m);
    }
    if (act && sub == 0)
      *reinterpret_cast<f32x4*>(p.fin_bw_sums + (long)k * C + q * 4) =
          f32x4{(float)a0, (float)a1, (float)a2, (float)a3};
  }
}

// Shared epilogue of the row kernels: accumulators -> LDS -> coalesced float4 rows.
// (`tile`: linear tile index of the launch, the workgroup's slot in IgemmArgs::tickets)
// XE ("extended epilogue"): the instantiation for launches with few workgroups per CU that combine
// their split-K slabs (p.tickets) and / or merge their tile partials (p.col_tickets) themselves; its
// loads in flight and double arithmetic cost registers, so the many-round launches keep the plain one.
template <int BM, int BN, bool QUAD = false, bool XE = false>
__device__ __forceinline__ void rows_epilogue(
    const IgemmArgs& p, float* lds, const f32x4 (&acc)[Tile<BM, BN>::TM][Tile<BM, BN>::TN], int m0,
    int n0, int t, int wave, int lane, int split, int tile) {
  using T = Tile<BM, BN>;
  float* Cs = lds;
  constexpr int NCH = (BN + T::CCH - 1) / T::CCH;
  // split-K combined in the launch: everyone publishes, the tile's last arriver goes on with the sum
  bool combine = false;
  if constexpr (XE) {
    if (p.tickets) {
      if (!splitk_publish<BM, BN, QUAD>(p, lds, acc, m0, n0, t, wave, lane, split, tile)) return;
      combine = true;
    }
  }
  const bool to_slab = p.slab && !combine;
  if (p.bw_mode != 0 && !to_slab) {
    // ---- dgrad + BatchNorm-backward reduction of the producer layer (see IgemmArgs::bw_*) ----
    // fixed thread -> column-quad map (q = t & 15, rows t >> 4, +16, ...), so a thread keeps its
    // coefficients and its two partial sums in registers; the 16 threads of a quad are then summed
    // in a fixed order through LDS (the C image is free by then).
    static_assert(T::CCH <= 64 && NT == 256, "16 quads x 16 row groups");
    const int q = t & 15, rg = t >> 4;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      if (ch > 0) __syncthreads();
      if (XE && combine) slabs_to_lds<BM, BN>(p, Cs, ch, m0, n0, t);
      else acc_to_lds<BM, BN, QUAD>(Cs, acc, ch, wave, lane);
      __syncthreads();
      const int col = n0 + ch * T::CCH + q * 4;
      const bool cv = q * 4 < T::CCH && ch * T::CCH + q * 4 < BN && col < p.Nn;
      f32x4 s1{0.f, 0.f, 0.f, 0.f}, s2{0.f, 0.f, 0.f, 0.f};
      if (cv) {
        const f32x4 scale = *reinterpret_cast<const f32x4*>(p.bw_coeffs + col);
        const f32x4 beta = *reinterpret_cast<const f32x4*>(p.bw_coeffs + p.Nn + col);
        const f32x4 mean = *reinterpret_cast<const f32x4*>(p.bw_coeffs + 2 * p.Nn + col);
        const f32x4 invstd = *reinterpret_cast<const f32x4*>(p.bw_coeffs + 3 * p.Nn + col);
        for (int row = rg; row < BM; row += 16) {
          const int m = m0 + row;
          if (m >= p.M) break;
          f32x4 v = *reinterpret_cast<const f32x4*>(&Cs[row * T::PC + q * 4]);
          const long pix = out_pixel(p, m);
          float* o = p.out + pix * p.ld_out + col;
          if (p.accumulate) v += *reinterpret_cast<const f32x4*>(o);
          const f32x4 yv = *reinterpret_cast<const f32x4*>(p.bw_y + pix * p.bw_ldy + col);
          if (p.bw_mode == 3) {
            const unsigned bits = p.bw_mask[pix * p.bw_ldmask + (col >> 2)];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ((bits >> e) & 1u) ? v[e] : 0.f;
          } else {
            f32x4 key;
            if (p.bw_mode == 2) key = *reinterpret_cast<const f32x4*>(p.bw_act + pix * p.bw_ldact + col);
            else key = (yv - mean) * scale + beta;          // the expression of bn_apply / masked_grad
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = key[e] > 0.f ? v[e] : 0.f;
          }
          *reinterpret_cast<f32x4*>(o) = v;
          s1 += v;
          s2 += v * ((yv - mean) * invstd);
        }
      }
      __syncthreads();                       // every thread is done with the C image
      f32x4* red = reinterpret_cast<f32x4*>(Cs);   // [2][16 row groups][16 quads]
      red[rg * 16 + q] = s1;
      red[256 + rg * 16 + q] = s2;
      __syncthreads();
      if (rg == 0 && cv) {
        for (int g = 1; g < 16; ++g) {
          s1 += red[g * 16 + q];
          s2 += red[256 + g * 16 + q];
        }
        const long C4 = p.Nn >> 2, np = p.tiles_m, tm = m0 / BM;
        if constexpr (XE) {
          // (write-through: the column's last workgroup may read them back inside this launch)
          const __amdgpu_buffer_rsrc_t rs_part = __builtin_amdgcn_make_buffer_rsrc(
              p.bw_part, 0, (unsigned)((long)2 * p.Nn * np * 4), 0x00020000);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, s1), rs_part,
                                                 (unsigned)(((0 * C4 + (col >> 2)) * np + tm) * 16), 0, kAuxSc1);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, s2), rs_part,
                                                 (unsigned)(((1 * C4 + (col >> 2)) * np + tm) * 16), 0, kAuxSc1);
        } else {
          f32x4* part4 = reinterpret_cast<f32x4*>(p.bw_part);
          part4[(0 * C4 + (col >> 2)) * np + tm] = s1;
          part4[(1 * C4 + (col >> 2)) * np + tm] = s2;
        }
      }
    }
    if constexpr (XE) {
      if (p.col_tickets && column_arrive<BM, BN>(p, lds, n0, t)) column_finalize_bwsums<BN>(p, n0, t);
    }
    return;
  }
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    if (ch > 0) __syncthreads();
    if (XE && combine) slabs_to_lds<BM, BN>(p, Cs, ch, m0, n0, t);
    else acc_to_lds<BM, BN, QUAD>(Cs, acc, ch, wave, lane);
    __syncthreads();
    constexpr int QPR = T::CCH / 4;
    for (int idx = t; idx < BM * QPR; idx += NT) {
      const int row = idx / QPR, q = idx - row * QPR;
      const int m = m0 + row;
      const int col = n0 + ch * T::CCH + q * 4;
      if (ch * T::CCH + q * 4 < BN && m < p.M && col < p.Nn) {
        f32x4 v = *reinterpret_cast<const f32x4*>(&Cs[row * T::PC + q * 4]);
        if (to_slab) {
          *reinterpret_cast<f32x4*>(p.slab + ((long)split * p.M + m) * p.Nn + col) = v;
        } else {
          if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + col);
          if (p.addend) v += *reinterpret_cast<const f32x4*>(p.addend + (long)m * p.ld_add + col);
          float* o = p.out + out_pixel(p, m) * p.ld_out + col;
          if (p.accumulate) v += *reinterpret_cast<const f32x4*>(o);
          *reinterpret_cast<f32x4*>(o) = v;
        }
      }
    }
    if (p.tile_stats && !to_slab) {
      // BatchNorm statistics of the tile while it is still in LDS (the BN that follows every conv
      // of this path would otherwise re-read the whole output from HBM): per column
      // s1 = sum(v - shift), s2 = sum (v - shift)^2 over the tile's rows, shift = its first row
      // (well conditioned per tile; bn_tile_finalize merges the tiles with Chan's formula).
      constexpr int G = NT / T::CCH;
      const int c = t % T::CCH, rg = t / T::CCH;
      float* red = lds + T::C_SZ;   // [2][G][CCH]
      const int colg = n0 + ch * T::CCH + c;
      const bool cv = rg < G && (ch * T::CCH + c) < BN && colg < p.Nn;
      const int nrows = min(BM, p.M - m0);
      float s1 = 0.f, s2 = 0.f;
      const float shift = Cs[c];
      if (cv)
        for (int r = rg; r < nrows; r += G) {
          const float v = Cs[r * T::PC + c] - shift;
          s1 += v;
          s2 += v * v;
        }
      if (rg < G) {
        red[rg * T::CCH + c] = s1;
        red[(G + rg) * T::CCH + c] = s2;
      }
      __syncthreads();
      if (cv && rg == 0) {
        for (int g = 1; g < G; ++g) {
          s1 += red[g * T::CCH + c];
          s2 += red[(G + g) * T::CCH + c];
        }
        const long C4 = p.Nn >> 2, np = p.tiles_m, tm = m0 / BM;
        const int cq = colg >> 2, e = colg & 3;
        float* ts1 = p.tile_stats + ((0 * C4 + cq) * np + tm) * 4 + e;
        float* ts2 = p.tile_stats + ((1 * C4 + cq) * np + tm) * 4 + e;
        float* ts3 = p.tile_stats + ((2 * C4 + cq) * np + tm) * 4 + e;
        if constexpr (XE) {
          // (write-through stores: the column's last workgroup may read them back inside this launch)
          __hip_atomic_store(ts1, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(ts2, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(ts3, shift, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          *ts1 = s1; *ts2 = s2; *ts3 = shift;
        }
      }
    }
  }
  if constexpr (XE) {
    if (p.tile_stats && !to_slab && p.col_tickets && column_arrive<BM, BN>(p, lds, n0, t))
      column_finalize_stats<BN>(p, n0, t);
  }
}


// Progress priority (experiment, GS_PRIO > 0): a wave lowers its issue priority as it advances through
// its K range, so that the co-resident workgroups of a CU -- which the arbiter otherwise serves
// oldest-first, finishing them one after the other and leaving the CU under-occupied for the last
// third of a single-round launch -- advance together.  `done` of `total` loop units.
#ifndef GS_PRIO
#define GS_PRIO 0
#endif
__device__ __forceinline__ void progress_prio(int done, int total) {
#if GS_PRIO == 1
  const int q = total >> 2;
  if (done >= 3 * q) __builtin_amdgcn_s_setprio(0);
  else if (done >= 2 * q) __builtin_amdgcn_s_setprio(1);
  else if (done >= q) __builtin_amdgcn_s_setprio(2);
  else __builtin_amdgcn_s_setprio(3);
#elif GS_PRIO == 2   // finer towards the end: 50 %, 75 %, 90 %
  if (10 * done >= 9 * total) __builtin_amdgcn_s_setprio(0);
  else if (4 * done >= 3 * total) __builtin_amdgcn_s_setprio(1);
  else if (2 * done >= total) __builtin_amdgcn_s_setprio(2);
  else __builtin_amdgcn_s_setprio(3);
#elif GS_PRIO == 3   // inverted (control): priority rises with progress
  const int q = total >> 2;
  if (done >= 3 * q) __builtin_amdgcn_s_setprio(3);
  else if (done >= 2 * q) __builtin_amdgcn_s_setprio(2);
  else if (done >= q) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
#else
  (void)done; (void)total;
#endif
}
__device__ __forceinline__ void progress_prio_end() {
#if GS_PRIO
  __builtin_amdgcn_s_setprio(0);
#endif
}

// ------------------------------------------------------------------------------------------
// Software-pipelined K loop shared by the fast row and wgrad kernels.
// One wave can run only ONE MFMA ahead of its instruction stream, so everything else a K step
// needs (fragment reads of the next k-group, the gather's address math + global loads for step
// i+3, the LDS stores of step i+1) is issued in the ~24 free issue cycles behind each 32-cycle MFMA
// instead of in serial sections between MFMA bursts (r01 stamps of the sectioned loop: 2465 cycles
// per K step for 1024 cycles of MFMA work per wave, MFMA pipe 75 % busy at 2 waves/SIMD).
// sched_barrier(0) after every slot pins the hand-placed order.
//   load_a(ra) / load_b(rb): issue the global loads of the NEXT unloaded K step (out-of-range
//                            steps must load zeros: bounds-checked buffer loads), load_b advances;
//   store_a(ra, As) / store_b(rb, Bs): registers -> k-major skewed LDS image.
// Step i computes from buf[i&1]; two register sets hold steps i+1 and i+2 (in flight); the set
// freed at step i is refilled with step i+3.  Unrolled by 6: buffer parity and set index static.
// ------------------------------------------------------------------------------------------
template <int BM, int BN, int AS, class LA, class LB, class SA, class SB>
__device__ __forceinline__ void pipelined_k_loop(int nk, float* lds,
                                                 f32x4 (&acc)[Tile<BM, BN>::TM][Tile<BM, BN>::TN],
                                                 int wave, int lane, LA&& load_a, LB&& load_b,
                                                 SA&& store_a, SB&& store_b) {
  using T = Tile<BM, BN>;
  constexpr int NQ = T::TM * T::TN;            // MFMAs per k-group
  const int kk = lane >> 4, li = lane & 15;
  const int a_base = kk * T::PA + wave * T::WM + li;
  const int b_base = T::A_SZ + kk * T::PB;
  float fa[2][T::TM], fb[2][T::TN];
  auto read_a = [&](const float* buf, int g, float (&a)[T::TM]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < T::TM; ++i) a[i] = buf[a_base + g * (4 * T::PA + 8) + i * 16];
  };
  auto read_b = [&](const float* buf, int g, float (&b)[T::TN]) __attribute__((always_inline)) {
    read_b_fragments<T::TN>(buf + b_base + g * (4 * T::PB + 8), li, b);
  };
  // slots (index of the MFMA within its k-group) behind which the side work is issued
  constexpr int Q_RB = NQ > 1 ? 1 : 0, Q_LA = NQ > 2 ? 2 : NQ - 1, Q_LB = NQ > 4 ? 4 : NQ - 1;
  f32x4 ra0[AS], ra1[AS], ra2[AS];
  f32x4 rb0[T::BV], rb1[T::BV], rb2[T::BV];
  float* buf0 = lds;
  float* buf1 = lds + T::STAGE;
  auto phase = [&](f32x4 (&rla)[AS], f32x4 (&rlb)[T::BV], const f32x4 (&rsa)[AS],
                   const f32x4 (&rsb)[T::BV], const float* bc, float* bn)
                   __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int cur = g & 1, nxt = cur ^ 1;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int i = q / T::TN, j = q - i * T::TN;
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
        if (g < 3 && q == 0) read_a(bc, g + 1, fa[nxt]);
        if (g < 3 && q == Q_RB) read_b(bc, g + 1, fb[nxt]);
        if (g == 0 && q == Q_LA) load_a(rla);
        if (g == 0 && q == Q_LB) load_b(rlb);
        if (g == 1 && q == Q_LA) store_a(rsa, bn);
        if (g == 1 && q == Q_LB) store_b(rsb, bn + T::A_SZ);
        // The step's only barrier sits at the end of k-group 2: by then every wave has read all of
        // the current stage (group 3's fragments were fetched at the start of group 2) and has
        // stored its share of the next one (group 1), so the barrier both frees the current buffer
        // for the stores of the step after next and publishes the next stage -- whose first
        // fragments are then fetched behind the MFMAs of group 3 instead of after the step.
        if (g == 2 && q == NQ - 1) __syncthreads();
        if (g == 3 && q == 0) read_a(bn, 0, fa[0]);
        if (g == 3 && q == Q_RB) read_b(bn, 0, fb[0]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  if (nk <= 0) return;
  load_a(ra0); load_b(rb0);
  load_a(ra1); load_b(rb1);
  load_a(ra2); load_b(rb2);
  store_a(ra0, buf0); store_b(rb0, buf0 + T::A_SZ);
  __syncthreads();
  read_a(buf0, 0, fa[0]);
  read_b(buf0, 0, fb[0]);
  int ib = 0;
  for (; ib + 6 <= nk; ib += 6) {   // unconditional body (see pipelined_k_loop_pairs)
    progress_prio(ib, nk);
    phase(ra0, rb0, ra1, rb1, buf0, buf1);
    phase(ra1, rb1, ra2, rb2, buf1, buf0);
    phase(ra2, rb2, ra0, rb0, buf0, buf1);
    phase(ra0, rb0, ra1, rb1, buf1, buf0);
    phase(ra1, rb1, ra2, rb2, buf0, buf1);
    phase(ra2, rb2, ra0, rb0, buf1, buf0);
  }
  if (ib + 0 < nk) phase(ra0, rb0, ra1, rb1, buf0, buf1);
  if (ib + 1 < nk) phase(ra1, rb1, ra2, rb2, buf1, buf0);
  if (ib + 2 < nk) phase(ra2, rb2, ra0, rb0, buf0, buf1);
  if (ib + 3 < nk) phase(ra0, rb0, ra1, rb1, buf1, buf0);
  if (ib + 4 < nk) phase(ra1, rb1, ra2, rb2, buf0, buf1);
  progress_prio_end();
}


// Paired variant: TWO K steps per barrier (four LDS stages, four register sets).  With 64-row tiles
// a K step is only 512 MFMA cycles per wave, and the per-step barrier (+ what cannot be hidden
// around it) costs a third of that: the MFMA pipe was 62-64 % busy over whole launches of the
// 64x64 kernels against ~82 % for the 128x128 ones (r01 SQ counters).  Phase p computes the pair
// (2p, 2p+1) from buffers C0/C1, stores pair p+1 (loaded during phase p-1) into N0/N1 and issues
// the global loads of pair p+2.  The barrier sits at the end of k-group 6 of 8: all reads of the
// current pair are complete by then (group 7's fragments are fetched at the start of group 6) and
// the next pair is stored; group 7 runs behind it while the next pair's first fragments arrive.
// An odd K-step count pads the last pair with a zero step (bounds-checked loads return zeros).
template <int BM, int BN, int AS, class LA, class LB, class SA, class SB>
__device__ __forceinline__ void pipelined_k_loop_pairs(
    int nk, float* lds, f32x4 (&acc)[Tile<BM, BN>::TM][Tile<BM, BN>::TN], int wave, int lane,
    LA&& load_a, LB&& load_b, SA&& store_a, SB&& store_b) {
  using T = Tile<BM, BN>;
  constexpr int NQ = T::TM * T::TN;
  const int kk = lane >> 4, li = lane & 15;
  const int a_base = kk * T::PA + wave * T::WM + li;
  const int b_base = T::A_SZ + kk * T::PB;
  float fa[2][T::TM], fb[2][T::TN];
  auto read_a = [&](const float* buf, int g, float (&a)[T::TM]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < T::TM; ++i) a[i] = buf[a_base + g * (4 * T::PA + 8) + i * 16];
  };
  auto read_b = [&](const float* buf, int g, float (&b)[T::TN]) __attribute__((always_inline)) {
    read_b_fragments<T::TN>(buf + b_base + g * (4 * T::PB + 8), li, b);
  };
  constexpr int Q_RB = NQ > 1 ? 1 : 0, Q_LA = NQ > 2 ? 2 : NQ - 1, Q_LB = NQ > 4 ? 4 : NQ - 1;
  f32x4 ra0[AS], ra1[AS], ra2[AS], ra3[AS];
  f32x4 rb0[T::BV], rb1[T::BV], rb2[T::BV], rb3[T::BV];
  float* A0 = lds;
  float* A1 = lds + T::STAGE;
  float* B0 = lds + 2 * T::STAGE;
  float* B1 = lds + 3 * T::STAGE;
  // rl*: sets refilled with pair p+2; rs*: sets holding pair p+1, stored this phase
  auto phase = [&](f32x4 (&rla0)[AS], f32x4 (&rlb0)[T::BV], f32x4 (&rla1)[AS], f32x4 (&rlb1)[T::BV],
                   const f32x4 (&rsa0)[AS], const f32x4 (&rsb0)[T::BV], const f32x4 (&rsa1)[AS],
                   const f32x4 (&rsb1)[T::BV], const float* c0, const float* c1, float* n0,
                   float* n1) __attribute__((always_inline)) {
#pragma unroll
    for (int G = 0; G < 8; ++G) {
      const int cur = G & 1, nxt = cur ^ 1;
      const float* cb = G < 4 ? c0 : c1;                 // buffer of this k-group
      const float* nb = (G + 1) < 4 ? c0 : c1;           // buffer of the next k-group (G < 7)
      const int gn = (G + 1) & 3;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int i = q / T::TN, j = q - i * T::TN;
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
        if (G < 7 && q == 0) read_a(nb, gn, fa[nxt]);
        if (G < 7 && q == Q_RB) read_b(nb, gn, fb[nxt]);
        if (G == 0 && q == Q_LA) load_a(rla0);
        if (G == 0 && q == Q_LB) load_b(rlb0);
        if (G == 1 && q == Q_LA) load_a(rla1);
        if (G == 1 && q == Q_LB) load_b(rlb1);
        if (G == 2 && q == Q_LA) store_a(rsa0, n0);
        if (G == 2 && q == Q_LB) store_b(rsb0, n0 + T::A_SZ);
        if (G == 4 && q == Q_LA) store_a(rsa1, n1);
        if (G == 4 && q == Q_LB) store_b(rsb1, n1 + T::A_SZ);
        if (G == 6 && q == NQ - 1) __syncthreads();
        if (G == 7 && q == 0) read_a(n0, 0, fa[0]);
        if (G == 7 && q == Q_RB) read_b(n0, 0, fb[0]);
        (void)cb;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  if (nk <= 0) return;
  const int npairs = (nk + 1) >> 1;
  load_a(ra0); load_b(rb0);
  load_a(ra1); load_b(rb1);
  load_a(ra2); load_b(rb2);
  load_a(ra3); load_b(rb3);
  store_a(ra0, A0); store_b(rb0, A0 + T::A_SZ);
  store_a(ra1, A1); store_b(rb1, A1 + T::A_SZ);
  __syncthreads();
  read_a(A0, 0, fa[0]);
  read_b(A0, 0, fb[0]);
  // (the loop body is unconditional: a conditional second phase made the compiler rotate the
  // accumulators through VGPRs at every iteration -- accvgpr reads that wait for the MFMA pipe to
  // drain, r01 ISA; the odd pair is peeled off instead)
  int pp = 0;
  for (; pp + 1 < npairs; pp += 2) {
    progress_prio(pp, npairs);
    // pair in A, next pair (held in sets 2,3) -> B, refill sets 0,1
    phase(ra0, rb0, ra1, rb1, ra2, rb2, ra3, rb3, A0, A1, B0, B1);
    // pair in B, next pair (sets 0,1) -> A, refill sets 2,3
    phase(ra2, rb2, ra3, rb3, ra0, rb0, ra1, rb1, B0, B1, A0, A1);
  }
  if (pp < npairs) phase(ra0, rb0, ra1, rb1, ra2, rb2, ra3, rb3, A0, A1, B0, B1);
  progress_prio_end();
}

// ------------------------------------------------------------------------------------------
// bf16x3 K loop (stride-1 data gradient by default, GS_X3=<min K steps>, 0 = off): the fp32 contraction as SIX bf16 MFMAs over an
// exact three-way bf16 split of both operands (x = x0 + x1 + x2 with 8 mantissa bits each; products
// a_i * b_j for i + j <= 2, smallest first, fp32 accumulation in v_mfma_f32_16x16x32_bf16): as
// accurate as the fp32 MFMA against fp64 (profiles/r02_bf16x3_probe.md) at 6 x 16 instead of
// 8 x 32 MFMA cycles per 16x16x32 of work.  One step = TWO 16-channel K steps of the loaders (the
// thread that stages (row, kq) holds channels 4kq..4kq+3 of both: its 8 values are one 16-byte bf16
// chunk per piece; any k order works as long as both operands use it).  Operands are split ONCE,
// at the stage store; LDS holds [piece][row][32 bf16 + 16 B pad] (80-byte rows: conflict-free
// 16-byte fragment reads).  Both operands must be k-contiguous per row: dgrad (BTRANS) only.
// Global loads run kX3Sets = 2 steps ahead of the MFMAs (two register sets); the split + stage store
// of step i+1 follows the MFMAs of step i into the SAME single LDS stage (kX3Stages = 1: one barrier
// before the store frees the stage, one after publishes it) -- 112 VGPRs and 31 KB of LDS, i.e. four
// workgroups per CU, which is where the loop's speed comes from.
// Non-finite operands: x3_split turns +-Inf into (Inf, NaN, NaN) pieces (Inf - Inf), so an output
// that contracts an overflowed element becomes NaN where the fp32 MFMA loop would give +-Inf; NaN
// stays NaN.  Finite inputs (the only case the parity bar covers) are split exactly.
// ------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ u32x4 x3_pack(const u32x4 a, const u32x4 b) {
  return u32x4{(a[0] >> 16) | (a[1] & 0xFFFF0000u), (a[2] >> 16) | (a[3] & 0xFFFF0000u),
               (b[0] >> 16) | (b[1] & 0xFFFF0000u), (b[2] >> 16) | (b[3] & 0xFFFF0000u)};
}
__device__ __forceinline__ void x3_split(const f32x4 lo4, const f32x4 hi4, u32x4& p0, u32x4& p1,
                                         u32x4& p2) {
  const u32x4 mask{0xFFFF0000u, 0xFFFF0000u, 0xFFFF0000u, 0xFFFF0000u};
  const u32x4 hl = __builtin_bit_cast(u32x4, lo4) & mask, hh = __builtin_bit_cast(u32x4, hi4) & mask;
  const f32x4 r1l = lo4 - __builtin_bit_cast(f32x4, hl), r1h = hi4 - __builtin_bit_cast(f32x4, hh);
  const u32x4 ml = __builtin_bit_cast(u32x4, r1l) & mask, mh = __builtin_bit_cast(u32x4, r1h) & mask;
  const f32x4 r2l = r1l - __builtin_bit_cast(f32x4, ml), r2h = r1h - __builtin_bit_cast(f32x4, mh);
  p0 = x3_pack(hl, hh);
  p1 = x3_pack(ml, mh);
  p2 = x3_pack(__builtin_bit_cast(u32x4, r2l), __builtin_bit_cast(u32x4, r2h));
}

// Interleaved form for the forward: the 16-byte chunk holds [lo0, hi0, lo1, hi1, lo2, hi2, lo3, hi3]
// (element j of the first 16-channel step next to element j of the second), so that a thread which
// holds ONE k row of the [k][n] weights per step owns an adjacent bf16 pair per output column.
__device__ __forceinline__ u32x4 x3_pack_il(const u32x4 a, const u32x4 b) {
  return u32x4{(a[0] >> 16) | (b[0] & 0xFFFF0000u), (a[1] >> 16) | (b[1] & 0xFFFF0000u),
               (a[2] >> 16) | (b[2] & 0xFFFF0000u), (a[3] >> 16) | (b[3] & 0xFFFF0000u)};
}
__device__ __forceinline__ void x3_split_il(const f32x4 lo4, const f32x4 hi4, u32x4& p0, u32x4& p1,
                                            u32x4& p2) {
  const u32x4 mask{0xFFFF0000u, 0xFFFF0000u, 0xFFFF0000u, 0xFFFF0000u};
  const u32x4 hl = __builtin_bit_cast(u32x4, lo4) & mask, hh = __builtin_bit_cast(u32x4, hi4) & mask;
  const f32x4 r1l = lo4 - __builtin_bit_cast(f32x4, hl), r1h = hi4 - __builtin_bit_cast(f32x4, hh);
  const u32x4 ml = __builtin_bit_cast(u32x4, r1l) & mask, mh = __builtin_bit_cast(u32x4, r1h) & mask;
  const f32x4 r2l = r1l - __builtin_bit_cast(f32x4, ml), r2h = r1h - __builtin_bit_cast(f32x4, mh);
  p0 = x3_pack_il(hl, hh);
  p1 = x3_pack_il(ml, mh);
  p2 = x3_pack_il(__builtin_bit_cast(u32x4, r2l), __builtin_bit_cast(u32x4, r2h));
}

// one LDS stage (two barriers per step, three workgroups per CU by registers) or two (one barrier,
// two workgroups per CU by LDS)
constexpr int kX3Stages = 1;
// Product terms a_i * b_j kept per element pair: 6 = all with i + j <= 2 (drops a1 b2 + a2 b1, each
// 2^-24 of |a||b| -- the size of one fp32 rounding -- and a2 b2); 8 = those two as well, leaving only
// a2 b2 (2^-32).  Measured (r03): 8 terms cost 9 % (stage-1 3x3 dgrad 47.6 -> 52.4 us) and do NOT
// lower the loop's noise -- with the forward on x3 the conditioned-gradient error ratios of the
// full-size parity tests were 1.53 / 3 outliers with 6 terms and 1.59 / 1.61 with 8 -- so the extra
// noise over the fp32 MFMA's exact fmaf chain is the bf16 MFMA's internal 32-term accumulation, not
// the dropped cross terms.  6 it is.
#ifndef GS_X3_TERMS
#define GS_X3_TERMS 6
#endif
constexpr int kX3Terms = GS_X3_TERMS;
constexpr int kX3Sets = 2;     // register sets = how many steps the global loads run ahead

template <int BN>
struct X3Tile {
  static constexpr int ROWB = 80;
  static constexpr int PA = 64 * ROWB, PB = BN * ROWB;
  static constexpr int STAGE = 3 * PA + 3 * PB;            // bytes
  static constexpr int LDS_FLOATS = kX3Stages * STAGE / 4;
};

// BFWD (forward): B arrives as one k ROW of the [k][n] weights per thread and step -- four columns
// (b_row .. b_row + 3) of k row b_kq (0..15) -- and both operands use the interleaved chunk order;
// the pair (step 1 value, step 2 value) of a column is one 32-bit LDS store per piece.
template <int BM, int BN, int AS, bool BFWD = false, class LA, class LB>
__device__ __forceinline__ void x3_k_loop(int nk16, float* ldsf,
                                          f32x4 (&acc)[Tile<BM, BN>::TM][Tile<BM, BN>::TN],
                                          int wave, int lane, int t, int b_row, int b_kq,
                                          LA&& load_a, LB&& load_b) {
  using T = Tile<BM, BN>;
  using X = X3Tile<BN>;
  using G = ColGroups<T::TN>;
  static_assert(BM == 64 && AS == 1 && T::BV == 1, "bf16x3 loop: 64-row tiles, BN <= 64");
  unsigned char* lds = reinterpret_cast<unsigned char*>(ldsf);
  const int li = lane & 15, fk = lane >> 4;
  const int row = t >> 2, kq = t & 3;
  // the thread's B staging slot: column b_row of the tile, k chunk b_kq (dgrad: (t >> 2, t & 3)
  // like A; forward: (t & 63, t >> 6), see the kernel's load_b)
  const bool b_on = b_row < BN;
  // register sets: step k lives in set k % kX3Sets
  f32x4 a0[kX3Sets][AS], a1[kX3Sets][AS], b0[kX3Sets][T::BV], b1[kX3Sets][T::BV];
  auto gload = [&](int set) __attribute__((always_inline)) {
    load_a(a0[set]); load_b(b0[set]);     // (load_b advances the K state)
    load_a(a1[set]); load_b(b1[set]);
  };
  auto sstore = [&](int set, unsigned char* st) __attribute__((always_inline)) {
    u32x4 p0, p1, p2;
    if constexpr (BFWD) x3_split_il(a0[set][0], a1[set][0], p0, p1, p2);
    else x3_split(a0[set][0], a1[set][0], p0, p1, p2);
    unsigned char* pa = st + row * X::ROWB + kq * 16;
    *reinterpret_cast<u32x4*>(pa) = p0;
    *reinterpret_cast<u32x4*>(pa + X::PA) = p1;
    *reinterpret_cast<u32x4*>(pa + 2 * X::PA) = p2;
    if constexpr (BFWD) {
      if (b_on) {
        // element e of the thread's row pair = column b_row + e: (step 1, step 2) as one dword at
        // chunk b_kq >> 2, pair slot b_kq & 3 of that column's row
        x3_split_il(b0[set][0], b1[set][0], p0, p1, p2);
        unsigned char* pb = st + 3 * X::PA + b_row * X::ROWB + (b_kq >> 2) * 16 + (b_kq & 3) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          *reinterpret_cast<unsigned*>(pb + e * X::ROWB) = p0[e];
          *reinterpret_cast<unsigned*>(pb + e * X::ROWB + X::PB) = p1[e];
          *reinterpret_cast<unsigned*>(pb + e * X::ROWB + 2 * X::PB) = p2[e];
        }
      }
    } else if (b_on) {
      x3_split(b0[set][0], b1[set][0], p0, p1, p2);
      unsigned char* pb = st + 3 * X::PA + b_row * X::ROWB + b_kq * 16;
      *reinterpret_cast<u32x4*>(pb) = p0;
      *reinterpret_cast<u32x4*>(pb + X::PB) = p1;
      *reinterpret_cast<u32x4*>(pb + 2 * X::PB) = p2;
    }
  };
  constexpr bool QUAD = BN == 64;   // 2 x 2 waves of 32 x 32: each wave re-reads half of B, not all
  // B fragment row of MFMA block j for this lane: the column the epilogue expects there
  int brow[T::TN];
#pragma unroll
  for (int j = 0; j < T::TN; ++j) brow[j] = G::base(j) + G::width(j) * li + (j - G::first(j));
  auto compute = [&](const unsigned char* cb) __attribute__((always_inline)) {
    if constexpr (QUAD) {
      bf16x8 qa[2][3], qb[2][3];
      const int ar = (wave >> 1) * 32 + li, bc = (wave & 1) * 32 + li;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          qa[h][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(
              cb + p * X::PA + (ar + h * 16) * X::ROWB + fk * 16));
          qb[h][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(
              cb + 3 * X::PA + p * X::PB + (bc + h * 16) * X::ROWB + fk * 16));
        }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i2 = q >> 1, j2 = q & 1;
        if constexpr (kX3Terms == 8) {   // the 2^-24-level cross terms (see kX3Terms)
          acc[0][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[i2][2], qb[j2][1], acc[0][q], 0, 0, 0);
          acc[0][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[i2][1], qb[j2][2], acc[0][q], 0, 0, 0);
        }
        acc[0][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[i2][2], qb[j2][0], acc[0][q], 0, 0, 0);
        acc[0][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[i2][1], qb[j2][1], acc[0][q], 0, 0, 0);
        acc[0][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[i2][0], qb[j2][2], acc[0][q], 0, 0, 0);
        acc[0][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[i2][1], qb[j2][0], acc[0][q], 0, 0, 0);
        acc[0][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[i2][0], qb[j2][1], acc[0][q], 0, 0, 0);
        acc[0][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[i2][0], qb[j2][0], acc[0][q], 0, 0, 0);
      }
      return;
    }
    bf16x8 fa[3], fb[3];
#pragma unroll
    for (int p = 0; p < 3; ++p)
      fa[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(
          cb + p * X::PA + (wave * 16 + li) * X::ROWB + fk * 16));
#pragma unroll
    for (int j = 0; j < T::TN; ++j) {
#pragma unroll
      for (int p = 0; p < 3; ++p)
        fb[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(
            cb + 3 * X::PA + p * X::PB + brow[j] * X::ROWB + fk * 16));
      if constexpr (kX3Terms == 8) {
        acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[2], fb[1], acc[0][j], 0, 0, 0);
        acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1], fb[2], acc[0][j], 0, 0, 0);
      }
      acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[2], fb[0], acc[0][j], 0, 0, 0);
      acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1], fb[1], acc[0][j], 0, 0, 0);
      acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[2], acc[0][j], 0, 0, 0);
      acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1], fb[0], acc[0][j], 0, 0, 0);
      acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[1], acc[0][j], 0, 0, 0);
      acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[0], acc[0][j], 0, 0, 0);
    }
  };
  const int nst = (nk16 + 1) >> 1;
  if (nst <= 0) return;
  unsigned char* buf0 = lds;
  unsigned char* buf1 = kX3Stages == 2 ? lds + X::STAGE : lds;
  // (loads past the K range return zeros: the loaders' kvalid test, so the prologue needs no guard)
#pragma unroll
  for (int u = 0; u < kX3Sets; ++u) gload(u);
  sstore(0, buf0);
  __syncthreads();
  // phase i (set = i % kX3Sets): compute step i from cur, refill its set with step i + kX3Sets,
  // store step i + 1 (the next set, loaded kX3Sets - 1 phases ago) into the (other) stage
#define GS_X3_PHASE(I, CUR, NXT)                                  \
  if (s + (I) < nst) {                                             \
    gload((I) % kX3Sets);                                          \
    compute(CUR);                                                  \
    if (kX3Stages == 1) __syncthreads();                           \
    sstore(((I) + 1) % kX3Sets, NXT);                              \
    __syncthreads();                                               \
  }
  static_assert(12 % kX3Sets == 0 && kX3Sets >= 2, "the phase loop is unrolled by 12");
  for (int s = 0; s < nst; s += 12) {
    GS_X3_PHASE(0, buf0, buf1) GS_X3_PHASE(1, buf1, buf0) GS_X3_PHASE(2, buf0, buf1)
    GS_X3_PHASE(3, buf1, buf0) GS_X3_PHASE(4, buf0, buf1) GS_X3_PHASE(5, buf1, buf0)
    GS_X3_PHASE(6, buf0, buf1) GS_X3_PHASE(7, buf1, buf0) GS_X3_PHASE(8, buf0, buf1)
    GS_X3_PHASE(9, buf1, buf0) GS_X3_PHASE(10, buf0, buf1) GS_X3_PHASE(11, buf1, buf0)
  }
#undef GS_X3_PHASE
}

// ------------------------------------------------------------------------------------------
// Fast path of forward / stride-1 dgrad for the shapes that carry the FLOPs: NHWC source,
// channels per tap a multiple of BK (every width of the search space is a multiple of 16), 1x1 or
// 3x3.  Versus the general kernel above:
//  * a K step never straddles a tap, so (tap, channel) is block-uniform: scalar registers;
//  * the per-row gather is hoisted out of the K loop: one base pointer and one tap-validity
//    bit mask per row slot, so a K step costs a mask test and one add per load (the general
//    kernel spent ~100 VALU + ~60 SALU instructions per K step on index math, r01 ISA census);
//  * global loads run THREE K steps ahead of the MFMAs (three register sets), because one K step of
//    fp32 MFMA work (~1k cycles per wave) is shorter than the gather's memory latency.
// ------------------------------------------------------------------------------------------
// ABL > 0 are timing-only ablation builds used by scratch/kbench.hip (1: no global loads,
// 2: + no LDS stores, 3: + no LDS reads); the library only instantiates ABL = 0.
// ABL == 9 is a diagnostic build (scratch/kbench.hip): s_memtime stamps around the sections of
// every phase, written per wave to p.slab as 8 x uint64.
__device__ __forceinline__ unsigned long long gs_stamp() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
// ROLE only names the instantiation (gs_conv_desc::role): role 1 = the bottleneck conv2 (K3), so
// that rocprofv3 attributes the headline kernel separately from the other 3x3 convolutions.
// AFF: the gathered operand is relu(bn(src)) (IgemmArgs::a_coeffs): the producer BatchNorm's
// coefficients sit in LDS behind the tile stages; each register set of the pipelined loop carries,
// besides its AS data quads, the three coefficient quads of its K step (fetched from LDS when the
// global loads are issued, two K steps before they are needed) and the tap-validity bits, so the
// affine + ReLU + zero-padding select run in the store slot on values that are already there.
// SK: the instantiation with the extended epilogue (rows_epilogue XE): launches that combine their
// split-K slabs (IgemmArgs::tickets) and / or merge their tile partials (col_tickets) themselves.
// A variant of its own because that code's loads in flight cost registers the many-round launches
// would pay for in occupancy.
template <int BM, int BN, bool BTRANS, int KS, int ABL = 0, int ROLE = 0, bool PIPE = true,
          bool PAIR = false, bool AFF = false, bool X3 = false, bool SK = false>
__global__ __launch_bounds__(NT) void igemm_rows_fast_kernel(const IgemmArgs p) {
  using T = Tile<BM, BN>;
  static_assert(!X3 || (PIPE && !PAIR && !AFF && ABL == 0 && BN <= 64), "bf16x3 loop: no loader fusion");
  constexpr int LDS_X3 = X3Tile<BN>::LDS_FLOATS > T::C_SZ + 512 ? X3Tile<BN>::LDS_FLOATS : T::C_SZ + 512;
  constexpr int LDS_TILES = X3 ? LDS_X3 : (PAIR ? T::LDSF2 : T::LDSF);
  __shared__ __attribute__((aligned(16))) float lds[LDS_TILES + (AFF ? 3 * kAffMaxC : 0)];
  constexpr int AS = BM / 64;
  constexpr int AX = AFF ? AS + 4 : AS;   // register-set size handed to the pipelined loop
  static_assert(!AFF || (PIPE && !BTRANS && ABL == 0), "AFF: forward, pipelined loop only");
  constexpr int MAXTAPS = KS * KS;  // KS only bounds the tap loop and tags the kernel name
  const int ntaps = p.kh_n * p.kw_n;
  unsigned long long st_entry = 0;
  if constexpr (ABL == 9) st_entry = __builtin_amdgcn_s_memrealtime();
  (void)st_entry;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // 1-D grid over (split, tile), split-major, remapped so that every XCD owns one contiguous run:
  // the tiles that share an operand sit behind the same L2.  tile_order 1 walks tm fastest (the
  // dense operand is the larger one: each XCD then needs only a few of its column blocks).
  const int ntiles = p.tiles_m * p.tiles_n;
  const int lin = xcd_remap(blockIdx.x, ntiles * p.nsplits);
  const int split = lin / ntiles;
  const int tile = lin - split * ntiles;
  int tm, tn;
  if (p.tile_order) { tn = tile / p.tiles_m; tm = tile - tn * p.tiles_m; }
  else { tm = tile / p.tiles_n; tn = tile - tm * p.tiles_n; }
  const int m0 = tm * BM, n0 = tn * BN;
  const int kt0 = split * p.nk_per_split;
  const int kt1 = min(kt0 + p.nk_per_split, p.nk_total);
  const int nk = kt1 - kt0;

  // ---- per-thread constants of the gather ----
  const int kq = t & 3;
  // bounds-checked buffer loads: an offset >= num_records returns 0, so padding / ragged edges
  // need no branch (the K loop stays one basic block the compiler can software-pipeline)
  const __amdgpu_buffer_rsrc_t rs_src =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src), 0, p.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dense =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dense), 0, p.dense_bytes, 0x00020000);
  constexpr unsigned kOOB = 0xFFFFFFFFu;
  int rowoff[AS];  // byte offset of (n, hb, wb, kq*4); wraps for the (masked) negative halo
  unsigned vmask[AS];
  const int hw = p.Hp * p.Wp;
#pragma unroll
  for (int s = 0; s < AS; ++s) {
    const int m = m0 + (t >> 2) + 64 * s;
    const bool rv = m < p.M;
    const int mm = rv ? m : 0;
    const int n = mm / hw;
    const int rem = mm - n * hw;
    const int hp = rem / p.Wp;
    const int wp = rem - hp * p.Wp;
    const int hb = hp * p.mul_h + p.base_h, wb = wp * p.mul_w + p.base_w;
    rowoff[s] = (int)(4 * ((long)n * p.s_n + (long)hb * p.s_h + (long)wb * p.s_w + kq * 4));
    unsigned mk = 0;
#pragma unroll
    for (int tp = 0; tp < MAXTAPS; ++tp) {
      const int jh = tp / p.kw_n, jw = tp - jh * p.kw_n;
      const int hi = hb + jh * p.step_h, wi = wb + jw * p.step_w;
      const bool ok = rv && tp < ntaps && (unsigned)hi < (unsigned)p.Hs &&
                      (unsigned)wi < (unsigned)p.Ws;
      mk |= (ok ? 1u : 0u) << tp;
    }
    vmask[s] = mk;
  }
  // ---- per-thread constants of the dense operand ----
  int boff[T::BV];  // bytes
  bool bok[T::BV];
#pragma unroll
  for (int r = 0; r < T::BV; ++r) {
    const int idx = t + NT * r;
    const bool iv = idx < BK * BN / 4;
    if constexpr (!BTRANS) {
      const int kr = idx / (BN / 4), nq = idx - kr * (BN / 4);
      const int col = n0 + nq * 4;
      bok[r] = iv && col < p.n_lim;
      boff[r] = 4 * (kr * p.d_row + col);
    } else {
      const int nrow = idx >> 2, kq2 = idx & 3;
      const int ng = n0 + nrow;
      bok[r] = iv && ng < p.Nn;
      boff[r] = 4 * (ng * p.d_row + kq2 * 4);
    }
  }

  // block-uniform position of the next K step to load
  int tap = (kt0 * BK) / p.Cs;
  int c0 = kt0 * BK - tap * p.Cs;

  float* coefL = lds + LDS_TILES;   // AFF: [mean | scale | beta][Cs]
  if constexpr (AFF) {
    const int cq = p.Cs >> 2;
    for (int i = t; i < 3 * cq; i += NT) {
      const int which = i / cq, q = i - which * cq;
      // coefficient order in memory: scale, beta, mean, invstd
      const int srcrow = which == 0 ? 2 : (which == 1 ? 0 : 1);
      *reinterpret_cast<f32x4*>(coefL + which * p.Cs + q * 4) =
          *reinterpret_cast<const f32x4*>(p.a_coeffs + (long)srcrow * p.Cs + q * 4);
    }
    __syncthreads();
  }

  f32x4 acc[T::TM][T::TN];
#pragma unroll
  for (int i = 0; i < T::TM; ++i)
#pragma unroll
    for (int j = 0; j < T::TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto load = [&](f32x4 (&ra)[AS], f32x4 (&rb)[T::BV]) {
    const bool kvalid = tap < ntaps;
    const int kh = tap / p.kw_n, kw = tap - kh * p.kw_n;
    const int aoff = 4 * (kh * p.step_h * (int)p.s_h + kw * p.step_w * (int)p.s_w + c0);
#pragma unroll
    for (int s = 0; s < AS; ++s) {
      const bool ok = kvalid && ((vmask[s] >> tap) & 1u);
      const unsigned off = ok ? (unsigned)(rowoff[s] + aoff) : kOOB;
      if constexpr (ABL >= 1 && ABL != 9) ra[s] = f32x4{(float)off, 1.f, 2.f, 3.f};
      else
        ra[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_src, off, 0, 0));
    }
    const int bbase = 4 * (kh * (int)p.d_tap_h + kw * (int)p.d_tap_w + (BTRANS ? c0 : c0 * p.d_row));
#pragma unroll
    for (int r = 0; r < T::BV; ++r) {
      const unsigned off = (bok[r] && kvalid) ? (unsigned)(boff[r] + bbase) : kOOB;
      if constexpr (ABL >= 1 && ABL != 9) rb[r] = f32x4{(float)off, 1.f, 2.f, 3.f};
      else
        rb[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dense, off, 0, 0));
    }
    c0 += BK;
    if (c0 >= p.Cs) { c0 = 0; ++tap; }
  };

  auto store = [&](const f32x4 (&ra)[AS], const f32x4 (&rb)[T::BV], int buf) {
    if constexpr (ABL >= 2 && ABL != 9) {
      asm volatile("" ::"v"(ra[0][0]), "v"(rb[0][0]));
      return;
    }
    float* As = lds + buf * T::STAGE;
    float* Bs = As + T::A_SZ;
#pragma unroll
    for (int s = 0; s < AS; ++s) {
      const int row = (t >> 2) + 64 * s;
#pragma unroll
      for (int j = 0; j < 4; ++j) As[(kq * 4 + j) * T::PA + 8 * kq + row] = ra[s][j];
    }
#pragma unroll
    for (int r = 0; r < T::BV; ++r) {
      const int idx = t + NT * r;
      if (idx < BK * BN / 4) {
        if constexpr (!BTRANS) {
          const int kr = idx / (BN / 4), nq = idx - kr * (BN / 4);
          *reinterpret_cast<f32x4*>(&Bs[kr * T::PB + 8 * (kr >> 2) + nq * 4]) = rb[r];
        } else {
          const int nrow = idx >> 2, kq2 = idx & 3;
#pragma unroll
          for (int j = 0; j < 4; ++j) Bs[(kq2 * 4 + j) * T::PB + 8 * kq2 + nrow] = rb[r][j];
        }
      }
    }
  };

  f32x4 ra0[AS], ra1[AS], ra2[AS];
  f32x4 rb0[T::BV], rb1[T::BV], rb2[T::BV];
  float* buf0 = lds;
  float* buf1 = lds + T::STAGE;
  unsigned long long st_k0 = 0, st_l0 = 0, st_l1 = 0, d_load = 0, d_mfma = 0, d_store = 0, d_bar = 0;
  unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
  (void)s0; (void)s1; (void)s2; (void)s3; (void)s4; (void)d_load; (void)d_mfma; (void)d_store; (void)d_bar;
#define GS_STAMP(x) if constexpr (ABL == 9) { x = gs_stamp(); }
  if constexpr (ABL == 9) st_k0 = __builtin_amdgcn_s_memrealtime();
  if constexpr (PIPE) {
    // scalar K-step state of the NEXT stage to load (no divisions, no branches in the loop).
    // (An incremental form with precomputed deltas saves ~10 SALU per K step but pushed the kernel
    // past its SGPR budget: the buffer descriptors were spilled to VGPRs and every load became a
    // waterfall loop.  The kernel sits at 103 SGPRs; keep the live scalar set small.)
    const int a_step_h = 4 * p.step_h * (int)p.s_h, a_step_w = 4 * p.step_w * (int)p.s_w;
    const int b_step_h = 4 * (int)p.d_tap_h, b_step_w = 4 * (int)p.d_tap_w;
    const int b_cmul = BTRANS ? 4 : 4 * p.d_row;
    int kh = tap / p.kw_n, kw = tap - kh * p.kw_n;
    int aoff = kh * a_step_h + kw * a_step_w + 4 * c0;
    int bbase = kh * b_step_h + kw * b_step_w + c0 * b_cmul;
    unsigned tapbit = tap < 32 ? (1u << tap) : 0u;
    int k_left = nk;
    auto load_a = [&](f32x4 (&ra)[AX]) __attribute__((always_inline)) {
      const bool kvalid = k_left > 0;
      unsigned okbits = 0;
#pragma unroll
      for (int s = 0; s < AS; ++s) {
        const bool ok = kvalid && (vmask[s] & tapbit) != 0;
        const unsigned off = ok ? (unsigned)(rowoff[s] + aoff) : kOOB;
        okbits |= (ok ? 1u : 0u) << s;
        if constexpr (ABL >= 1 && ABL != 9) ra[s] = f32x4{(float)off, 1.f, 2.f, 3.f};
        else
          ra[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_src, off, 0, 0));
      }
      if constexpr (AFF) {
        const float* cp = coefL + c0 + kq * 4;   // channels of this K step handled by this thread
        ra[AS] = *reinterpret_cast<const f32x4*>(cp);
        ra[AS + 1] = *reinterpret_cast<const f32x4*>(cp + p.Cs);
        ra[AS + 2] = *reinterpret_cast<const f32x4*>(cp + 2 * p.Cs);
        ra[AS + 3][0] = __builtin_bit_cast(float, okbits);
      }
    };
    // bf16x3 forward: thread (k row kr = lane & 15 of the 16-channel step, column quad
    // nq = (lane >> 4) + 4 * wave) fetches ONE 16-byte quad of the [k][n] weights per step, like the
    // fp32 loop (a wave reads 16 rows x 64 contiguous bytes).  The k rows run fastest over the lanes so
    // that the 32-bit pair stores of x3_k_loop<BFWD> walk consecutive LDS banks.  (r02 fetched four
    // consecutive k of one column as four dword loads per step: 4x the vector-memory instructions,
    // level with the fp32 loop.)
    const int x3_kr = lane & 15, x3_nq = (lane >> 4) + 4 * wave;
    const int x3_col = n0 + 4 * x3_nq;
    const bool x3_bok = 4 * x3_nq < BN && x3_col < p.n_lim;
    const int x3_boff = 4 * (x3_kr * p.d_row + x3_col);
    auto load_b = [&](f32x4 (&rb)[T::BV]) __attribute__((always_inline)) {
      const bool kvalid = k_left > 0;
      if constexpr (X3 && !BTRANS) {
        const unsigned off = (x3_bok && kvalid) ? (unsigned)(x3_boff + bbase) : kOOB;
        rb[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dense, off, 0, 0));
      } else
#pragma unroll
      for (int r = 0; r < T::BV; ++r) {
        const unsigned off = (bok[r] && kvalid) ? (unsigned)(boff[r] + bbase) : kOOB;
        if constexpr (ABL >= 1 && ABL != 9) rb[r] = f32x4{(float)off, 1.f, 2.f, 3.f};
        else
          rb[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dense, off, 0, 0));
      }
      // advance to the following K step: pure integer arithmetic, the offsets are recomputed from
      // (kh, kw, c0) every time -- written as selects this became four scalar branches per K step
      // inside the MFMA stream (r01 ISA)
      --k_left;
      const int nc0 = c0 + BK;
      const int wrap = nc0 >= p.Cs ? 1 : 0;          // next K step starts a new tap
      const int nkw = kw + wrap;
      const int wrapw = nkw >= p.kw_n ? 1 : 0;
      kh += wrapw;
      kw = nkw * (1 - wrapw);
      c0 = nc0 * (1 - wrap);
      tapbit <<= wrap;
      aoff = kh * a_step_h + kw * a_step_w + 4 * c0;
      bbase = kh * b_step_h + kw * b_step_w + c0 * b_cmul;
    };
    auto store_a = [&](const f32x4 (&ra)[AX], float* As) __attribute__((always_inline)) {
      if constexpr (ABL >= 2 && ABL != 9) { asm volatile("" ::"v"(ra[0][0])); return; }
#pragma unroll
      for (int s = 0; s < AS; ++s) {
        const int row = (t >> 2) + 64 * s;
        f32x4 v = ra[s];
        if constexpr (AFF) {
          const unsigned okbits = __builtin_bit_cast(unsigned, ra[AS + 3][0]);
          v = bn_relu_affine(v, ra[AS], ra[AS + 1], ra[AS + 2]);
          if (!((okbits >> s) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};   // zero padding of the ACTIVATION
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) As[(kq * 4 + j) * T::PA + 8 * kq + row] = v[j];
      }
    };
    auto store_b = [&](const f32x4 (&rb)[T::BV], float* Bs) __attribute__((always_inline)) {
      if constexpr (ABL >= 2 && ABL != 9) { asm volatile("" ::"v"(rb[0][0])); return; }
#pragma unroll
      for (int r = 0; r < T::BV; ++r) {
        const int idx = t + NT * r;
        if ((r + 1) * NT <= BK * BN / 4 || idx < BK * BN / 4) {
          if constexpr (!BTRANS) {
            const int kr = idx / (BN / 4), nq = idx - kr * (BN / 4);
            *reinterpret_cast<f32x4*>(&Bs[kr * T::PB + 8 * (kr >> 2) + nq * 4]) = rb[r];
          } else {
            const int nrow = idx >> 2, kq2 = idx & 3;
#pragma unroll
            for (int j = 0; j < 4; ++j) Bs[(kq2 * 4 + j) * T::PB + 8 * kq2 + nrow] = rb[r][j];
          }
        }
      }
    };
    GS_STAMP(st_l0)
    if constexpr (X3)
      x3_k_loop<BM, BN, AX, !BTRANS>(nk, lds, acc, wave, lane, t, BTRANS ? (t >> 2) : 4 * x3_nq,
                                     BTRANS ? (t & 3) : x3_kr, load_a, load_b);
    else if constexpr (PAIR)
      pipelined_k_loop_pairs<BM, BN, AX>(nk, lds, acc, wave, lane, load_a, load_b, store_a, store_b);
    else
      pipelined_k_loop<BM, BN, AX>(nk, lds, acc, wave, lane, load_a, load_b, store_a, store_b);
  } else {
  // step i computes from buf[i&1]; two register sets hold steps i+1 and i+2 (in flight); the set
  // freed at step i is refilled with step i+3.  Unrolled by 6: buffer parity and set index static.
#define GS_ROW_PHASE(I, RL_A, RL_B, RS_A, RS_B, BC, BI)                                   \
  if ((I) < nk) {                                                                        \
    GS_STAMP(s0)                                                                         \
    if ((I) + 3 < nk) load(RL_A, RL_B);                                                  \
    GS_STAMP(s1)                                                                         \
    mfma_stage<BM, BN, (ABL >= 3 && ABL != 9)>(BC, BC + T::A_SZ, acc, wave, lane);       \
    GS_STAMP(s2)                                                                         \
    if ((I) + 1 < nk) store(RS_A, RS_B, BI);                                             \
    GS_STAMP(s3)                                                                         \
    __syncthreads();                                                                     \
    GS_STAMP(s4)                                                                         \
    if constexpr (ABL == 9) {                                                            \
      d_load += s1 - s0; d_mfma += s2 - s1; d_store += s3 - s2; d_bar += s4 - s3;        \
    }                                                                                    \
  }
  if (nk > 0) {
    load(ra0, rb0);
    if (nk > 1) load(ra1, rb1);
    if (nk > 2) load(ra2, rb2);
    store(ra0, rb0, 0);
    __syncthreads();
    GS_STAMP(st_l0)
    for (int ib = 0; ib < nk; ib += 6) {
      GS_ROW_PHASE(ib + 0, ra0, rb0, ra1, rb1, buf0, 1)
      GS_ROW_PHASE(ib + 1, ra1, rb1, ra2, rb2, buf1, 0)
      GS_ROW_PHASE(ib + 2, ra2, rb2, ra0, rb0, buf0, 1)
      GS_ROW_PHASE(ib + 3, ra0, rb0, ra1, rb1, buf1, 0)
      GS_ROW_PHASE(ib + 4, ra1, rb1, ra2, rb2, buf0, 1)
      GS_ROW_PHASE(ib + 5, ra2, rb2, ra0, rb0, buf1, 0)
    }
  }
  }
#undef GS_ROW_PHASE
  GS_STAMP(st_l1)
  if constexpr (ABL == 9) {
    float* keep = p.slab;
    IgemmArgs q = p;
    q.slab = nullptr;
    q.tickets = nullptr;
    q.col_tickets = nullptr;
    rows_epilogue<BM, BN>(q, lds, acc, m0, n0, t, wave, lane, split, tile);
    const unsigned long long r_end = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(keep) + ((long)blockIdx.x * 4 + wave) * 8;
      o[0] = st_k0; o[1] = st_l1 - st_l0; o[2] = st_entry; o[3] = r_end;
      o[4] = d_load; o[5] = d_mfma; o[6] = d_store;
      // HW_REG_HW_ID (4) and HW_REG_XCC_ID (20), all 32 bits: which CU / SIMD ran this wave
      o[7] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
             ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
      (void)d_bar;
    }
  } else {
    rows_epilogue<BM, BN, (X3 && BN == 64), SK>(p, lds, acc, m0, n0, t, wave, lane, split, tile);
  }
#undef GS_STAMP
}

// ------------------------------------------------------------------------------------------
// wgrad: GEMM rows are (tap, ci), K runs over pixels; both operands are k-major in memory.
// ------------------------------------------------------------------------------------------
template <int BM, int BN, bool SCALAR, int KS>
__global__ __launch_bounds__(NT) void igemm_wgrad_kernel(const IgemmArgs p) {
  using T = Tile<BM, BN>;
  __shared__ __attribute__((aligned(16))) float lds[T::LDSF];
  constexpr int QA = BM / 4;        // float4 per k-row of A
  constexpr int AS = BK * QA / NT;  // A float4 slots per thread (1 or 2)
  constexpr int KSTR = NT / QA;     // k rows covered per slot

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int ntiles = p.tiles_m * p.tiles_n;
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int i0 = tm * BM, n0 = tn * BN;
  const int kt0 = blockIdx.y * p.nk_per_split;
  const int kt1 = min(kt0 + p.nk_per_split, p.nk_total);

  // fixed (tap, c) of this thread's 4 GEMM rows
  const int iq = t % QA, krA = t / QA;
  int offh[SCALAR ? 4 : 1], offw[SCALAR ? 4 : 1];
  long offc[SCALAR ? 4 : 1];
  bool iv[SCALAR ? 4 : 1];
#pragma unroll
  for (int e = 0; e < (SCALAR ? 4 : 1); ++e) {
    const int i = i0 + iq * 4 + e;
    iv[e] = i < p.M;
    const int ii = iv[e] ? i : 0;
    const int tap = ii / p.Cs, c = ii - tap * p.Cs;
    const int kwid = KS ? KS : p.KW;
    const int kh = tap / kwid, kw = tap - kh * kwid;
    offh[e] = p.base_h + kh * p.step_h;
    offw[e] = p.base_w + kw * p.step_w;
    offc[e] = (long)c * p.s_c;
  }
  const int hw = p.Hp * p.Wp;

  f32x4 ra[AS];
  f32x4 rb[T::BV];
  f32x4 acc[T::TM][T::TN];
#pragma unroll
  for (int i = 0; i < T::TM; ++i)
#pragma unroll
    for (int j = 0; j < T::TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto load_tiles = [&](int kt) {
#pragma unroll
    for (int s = 0; s < AS; ++s) {
      const int m = kt * BK + krA + s * KSTR;
      const bool mv = m < p.npix;
      const int mm = mv ? m : 0;
      const int n = mm / hw;
      const int rem = mm - n * hw;
      const int hp = rem / p.Wp;
      const int wp = rem - hp * p.Wp;
      const long nbase = (long)n * p.s_n;
      const int hbase = hp * p.mul_h, wbase = wp * p.mul_w;
      if constexpr (!SCALAR) {
        const int hi = hbase + offh[0], wi = wbase + offw[0];
        const bool ok = mv && iv[0] && (unsigned)hi < (unsigned)p.Hs && (unsigned)wi < (unsigned)p.Ws;
        ra[s] = ok ? *reinterpret_cast<const f32x4*>(p.src + nbase + (long)hi * p.s_h +
                                                     (long)wi * p.s_w + offc[0])
                   : f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int hi = hbase + offh[e], wi = wbase + offw[e];
          const bool ok =
              mv && iv[e] && (unsigned)hi < (unsigned)p.Hs && (unsigned)wi < (unsigned)p.Ws;
          ra[s][e] = ok ? p.src[nbase + (long)hi * p.s_h + (long)wi * p.s_w + offc[e]] : 0.f;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < T::BV; ++r) {
      const int idx = t + NT * r;
      rb[r] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (idx < BK * BN / 4) {
        const int kr = idx / (BN / 4), nq = idx - kr * (BN / 4);
        const int m = kt * BK + kr;
        const int col = n0 + nq * 4;
        if (m < p.npix && col < p.n_lim)
          rb[r] = *reinterpret_cast<const f32x4*>(p.dense + (long)m * p.d_row + col);
      }
    }
  };

  auto store_tiles = [&](int buf) {
    float* As = lds + buf * T::STAGE;
    float* Bs = As + T::A_SZ;
#pragma unroll
    for (int s = 0; s < AS; ++s) {
      const int kr = krA + s * KSTR;
      *reinterpret_cast<f32x4*>(&As[kr * T::PA + 8 * (kr >> 2) + iq * 4]) = ra[s];
    }
#pragma unroll
    for (int r = 0; r < T::BV; ++r) {
      const int idx = t + NT * r;
      if (idx < BK * BN / 4) {
        const int kr = idx / (BN / 4), nq = idx - kr * (BN / 4);
        *reinterpret_cast<f32x4*>(&Bs[kr * T::PB + 8 * (kr >> 2) + nq * 4]) = rb[r];
      }
    }
  };

  if (kt0 < kt1) {
    load_tiles(kt0);
    store_tiles(0);
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
      const int buf = (kt - kt0) & 1;
      const bool more = kt + 1 < kt1;
      if (more) load_tiles(kt + 1);
      mfma_stage<BM, BN>(lds + buf * T::STAGE, lds + buf * T::STAGE + T::A_SZ, acc, wave, lane);
      if (more) store_tiles(buf ^ 1);
      __syncthreads();
    }
  }

  float* Cs = lds;
  constexpr int NCH = (BN + T::CCH - 1) / T::CCH;
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    if (ch > 0) __syncthreads();
    acc_to_lds<BM, BN>(Cs, acc, ch, wave, lane);
    __syncthreads();
    constexpr int QPR = T::CCH / 4;
    for (int idx = t; idx < BM * QPR; idx += NT) {
      const int row = idx / QPR, q = idx - row * QPR;
      const int i = i0 + row;
      const int col = n0 + ch * T::CCH + q * 4;
      if (ch * T::CCH + q * 4 < BN && i < p.M && col < p.Nn) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(&Cs[row * T::PC + q * 4]);
        if (p.slab) {
          *reinterpret_cast<f32x4*>(p.slab + ((long)blockIdx.y * p.M + i) * p.Nn + col) = v;
        } else {
          const int tap = i / p.Cs, c = i - tap * p.Cs;
          *reinterpret_cast<f32x4*>(p.out + (long)tap * p.o_tap + (long)c * p.o_row + col) = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Fast wgrad (NHWC source, Ci % 4 == 0): a streaming kernel — at stage 1 it moves 12 KB per K step
// for 1k cycles of MFMA work, so it lives on bytes in flight.  Same recipe as the fast row kernel:
// index math hoisted / incremental, bounds-checked buffer loads (no branches), and THREE K steps
// of global loads in flight (three register sets, statically rotated).
// ------------------------------------------------------------------------------------------
// WALIGN: Wp % BK == 0, so the BK pixels of a K step lie in one image row: (n, h, w0) of the step are
// scalars and the per-thread gather state (three counters with carries, ~20 VALU per K step inside
// the MFMA stream) reduces to two adds.
// AFF: the gathered operand is relu(bn(src)) (IgemmArgs::a_coeffs).  A thread's four GEMM rows are
// four fixed channels, so their coefficients live in registers for the whole kernel; each register
// set carries one extra quad with the validity bits of its loads (padding must stay zero).
template <int BM, int BN, int KS, bool PAIR = false, bool WALIGN = false, bool AFF = false>
__global__ __launch_bounds__(NT) void igemm_wgrad_fast_kernel(const IgemmArgs p) {
  using T = Tile<BM, BN>;
  __shared__ __attribute__((aligned(16))) float lds[PAIR ? T::LDSF2 : T::LDSF];
  constexpr int QA = BM / 4;
  constexpr int AS = BK * QA / NT;  // 1 (BM = 64) or 2 (BM = 128)
  constexpr int AX = AFF ? AS + 1 : AS;
  constexpr int KSTR = NT / QA;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // 1-D grid over (split, tile), split-major + XCD remap: all tiles of one pixel range (they gather
  // the same x region and read the same dy rows) sit behind the same L2
  const int ntiles = p.tiles_m * p.tiles_n;
  const int lin = xcd_remap(blockIdx.x, ntiles * p.nsplits);
  const int split = lin / ntiles;
  const int tile = lin - split * ntiles;
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int i0 = tm * BM, n0 = tn * BN;
  const int kt0 = split * p.nk_per_split;
  const int kt1 = min(kt0 + p.nk_per_split, p.nk_total);
  const int nk = kt1 - kt0;

  const __amdgpu_buffer_rsrc_t rs_src =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src), 0, p.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dense =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dense), 0, p.dense_bytes, 0x00020000);
  constexpr unsigned kOOB = 0xFFFFFFFFu;

  // this thread's four GEMM rows share one tap (Cs % 4 == 0): constant source offsets
  const int iq = t % QA, krA = t / QA;
  const int i = i0 + iq * 4;
  const bool iv = i < p.M;
  const int ii = iv ? i : 0;
  const int tap = ii / p.Cs, c = ii - tap * p.Cs;
  const int kwid = KS ? KS : p.KW;
  const int kh = tap / kwid, kw = tap - kh * kwid;
  const int offh = p.base_h + kh * p.step_h, offw = p.base_w + kw * p.step_w;
  f32x4 a_mean{0.f, 0.f, 0.f, 0.f}, a_scale{0.f, 0.f, 0.f, 0.f}, a_beta{0.f, 0.f, 0.f, 0.f};
  if constexpr (AFF) {   // coefficient order in memory: scale, beta, mean, invstd
    a_scale = *reinterpret_cast<const f32x4*>(p.a_coeffs + c);
    a_beta = *reinterpret_cast<const f32x4*>(p.a_coeffs + p.Cs + c);
    a_mean = *reinterpret_cast<const f32x4*>(p.a_coeffs + 2 * p.Cs + c);
  }
  // pixel state per slot, advanced by BK pixels per K step
  int pn[AS], ph[AS], pw[AS];
  const int hw = p.Hp * p.Wp;
  const int Nb = p.npix / hw;
#pragma unroll
  for (int s = 0; s < AS; ++s) {
    const int m = kt0 * BK + krA + s * KSTR;
    pn[s] = m / hw;
    const int rem = m - pn[s] * hw;
    ph[s] = rem / p.Wp;
    pw[s] = rem - ph[s] * p.Wp;
  }
  // dense operand (dy rows): per-thread constants
  int brow[T::BV], bcol[T::BV];
  bool bok[T::BV];
#pragma unroll
  for (int r = 0; r < T::BV; ++r) {
    const int idx = t + NT * r;
    const int kr = idx / (BN / 4), nq = idx - kr * (BN / 4);
    brow[r] = kr;
    bcol[r] = n0 + nq * 4;
    bok[r] = idx < BK * BN / 4 && bcol[r] < p.n_lim;
  }
  int kt_load = kt0;

  f32x4 acc[T::TM][T::TN];
#pragma unroll
  for (int a = 0; a < T::TM; ++a)
#pragma unroll
    for (int b = 0; b < T::TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // advancing a slot by BK pixels in the (n, h, w) mixed radix, branch-free (one carry per digit)
  const int adv_w = BK % p.Wp, adv_q = BK / p.Wp;
  const int adv_h = adv_q % p.Hp, adv_n = adv_q / p.Hp;
  int k_left = nk;
  // WALIGN scalar pixel state of the next K step
  int s_n = 0, s_h = 0, s_w0 = 0;
  if constexpr (WALIGN) {
    const int m0 = kt0 * BK;
    s_n = m0 / hw;
    const int rem = m0 - s_n * hw;
    s_h = rem / p.Wp;
    s_w0 = rem - s_h * p.Wp;
  }
  const int cw_off = krA * p.mul_w + offw;   // + s * KSTR * mul_w per slot
  auto load_a = [&](f32x4 (&ra)[AX]) __attribute__((always_inline)) {
    const bool kvalid = k_left > 0;
    unsigned okbits = 0;
    if constexpr (WALIGN) {
      const bool rowv = kvalid && s_n < Nb;
      const int hb = s_h * p.mul_h, wb = s_w0 * p.mul_w;
      const int nbase = s_n * (int)p.s_n;
      const int hi = hb + offh;
#pragma unroll
      for (int s = 0; s < AS; ++s) {
        const int wi = wb + cw_off + s * KSTR * p.mul_w;
        const bool ok = rowv && iv && (unsigned)hi < (unsigned)p.Hs && (unsigned)wi < (unsigned)p.Ws;
        const unsigned off =
            ok ? 4u * (unsigned)(nbase + hi * (int)p.s_h + wi * (int)p.s_w + c) : kOOB;
        okbits |= (ok ? 1u : 0u) << s;
        ra[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_src, off, 0, 0));
      }
      if constexpr (AFF) ra[AS][0] = __builtin_bit_cast(float, okbits);
      const int w2 = s_w0 + BK;
      const int cw = w2 >= p.Wp ? 1 : 0;
      s_w0 = w2 * (1 - cw);
      const int h2 = s_h + cw;
      const int ch = h2 >= p.Hp ? 1 : 0;
      s_h = h2 * (1 - ch);
      s_n += ch;
      return;
    }
#pragma unroll
    for (int s = 0; s < AS; ++s) {
      const int hi = ph[s] * p.mul_h + offh, wi = pw[s] * p.mul_w + offw;
      const bool ok = kvalid && iv && pn[s] < Nb && (unsigned)hi < (unsigned)p.Hs &&
                      (unsigned)wi < (unsigned)p.Ws;
      const unsigned off =
          ok ? 4u * (unsigned)(pn[s] * (int)p.s_n + hi * (int)p.s_h + wi * (int)p.s_w + c) : kOOB;
      okbits |= (ok ? 1u : 0u) << s;
      ra[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_src, off, 0, 0));
      const int w2 = pw[s] + adv_w;
      const int cw = w2 >= p.Wp ? 1 : 0;
      pw[s] = w2 - (cw ? p.Wp : 0);
      const int h2 = ph[s] + adv_h + cw;
      const int ch = h2 >= p.Hp ? 1 : 0;
      ph[s] = h2 - (ch ? p.Hp : 0);
      pn[s] += adv_n + ch;
    }
    if constexpr (AFF) ra[AS][0] = __builtin_bit_cast(float, okbits);
  };
  auto load_b = [&](f32x4 (&rb)[T::BV]) __attribute__((always_inline)) {
    const bool kvalid = k_left > 0;
#pragma unroll
    for (int r = 0; r < T::BV; ++r) {
      const int m = kt_load * BK + brow[r];
      const unsigned off =
          (kvalid && bok[r] && m < p.npix) ? 4u * (unsigned)(m * p.d_row + bcol[r]) : kOOB;
      rb[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dense, off, 0, 0));
    }
    ++kt_load;
    --k_left;
  };
  auto store_a = [&](const f32x4 (&ra)[AX], float* As) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < AS; ++s) {
      const int kr = krA + s * KSTR;
      f32x4 v = ra[s];
      if constexpr (AFF) {
        const unsigned okbits = __builtin_bit_cast(unsigned, ra[AS][0]);
        v = bn_relu_affine(v, a_mean, a_scale, a_beta);
        if (!((okbits >> s) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      *reinterpret_cast<f32x4*>(&As[kr * T::PA + 8 * (kr >> 2) + iq * 4]) = v;
    }
  };
  auto store_b = [&](const f32x4 (&rb)[T::BV], float* Bs) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < T::BV; ++r) {
      const int idx = t + NT * r;
      if ((r + 1) * NT <= BK * BN / 4 || idx < BK * BN / 4) {
        const int kr = idx / (BN / 4), nq = idx - kr * (BN / 4);
        *reinterpret_cast<f32x4*>(&Bs[kr * T::PB + 8 * (kr >> 2) + nq * 4]) = rb[r];
      }
    }
  };
  if constexpr (PAIR)
    pipelined_k_loop_pairs<BM, BN, AX>(nk, lds, acc, wave, lane, load_a, load_b, store_a, store_b);
  else
    pipelined_k_loop<BM, BN, AX>(nk, lds, acc, wave, lane, load_a, load_b, store_a, store_b);

  float* Cs = lds;
  constexpr int NCH = (BN + T::CCH - 1) / T::CCH;
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    if (ch > 0) __syncthreads();
    acc_to_lds<BM, BN>(Cs, acc, ch, wave, lane);
    __syncthreads();
    constexpr int QPR = T::CCH / 4;
    for (int idx = t; idx < BM * QPR; idx += NT) {
      const int row = idx / QPR, q = idx - row * QPR;
      const int ir = i0 + row;
      const int col = n0 + ch * T::CCH + q * 4;
      if (ch * T::CCH + q * 4 < BN && ir < p.M && col < p.Nn) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(&Cs[row * T::PC + q * 4]);
        if (p.slab) {
          *reinterpret_cast<f32x4*>(p.slab + ((long)split * p.M + ir) * p.Nn + col) = v;
        } else {
          const int tp = ir / p.Cs, cc = ir - tp * p.Cs;
          *reinterpret_cast<f32x4*>(p.out + (long)tp * p.o_tap + (long)cc * p.o_row + col) = v;
        }
      }
    }
  }
}

// Fixed-order sum of the split-K partial slabs + epilogue.  rows_are_taps selects the wgrad
// output addressing.
// WIDE = false: one thread per output float4 walks the splits (few splits, many outputs).
// WIDE = true : many splits, few outputs (stage-1 / stem weight gradients: 256 splits of a 64 x 256
//               tile).  A block is 16 adjacent output quads x 16 split groups: thread (q, g) sums the
//               slabs g, g + 16, ... of its quad — the 16 quads of a slab row are 256 contiguous
//               bytes, so every load instruction reads whole lines, and a thread's splits / 16 loads
//               are independent — and the 16 group sums are combined through LDS in group order.
//               (r03 form: one wave per quad with the lanes striding over the slabs, 16 B out of every
//               line it touched: 37-117 us per launch at the end of backward, where the side stream
//               is the critical path.)  Fixed order: bit-reproducible.
template <bool WIDE, int ROLE = 0>
static __global__ __launch_bounds__(256) void splitk_reduce_kernel(const IgemmArgs p, int splits,
                                                                   int rows_are_taps) {
  const int qpr = p.Nn / 4;
  const long total = (long)p.M * qpr;
  __shared__ f32x4 red[WIDE ? 256 : 1];
  const int qi = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const long first = WIDE ? ((long)blockIdx.x * 16 + qi)
                          : ((long)blockIdx.x * blockDim.x + threadIdx.x);
  const long stride = WIDE ? (long)gridDim.x * 16 : (long)gridDim.x * blockDim.x;
  const long bound = WIDE ? ((total + 15) / 16) * 16 : total;   // WIDE: whole blocks reach the barriers
  for (long idx = first; idx < bound; idx += stride) {
    const bool live = idx < total;
    const int row = live ? (int)(idx / qpr) : 0;
    const int col = live ? (int)(idx - (long)row * qpr) * 4 : 0;
    f32x4 v{0.f, 0.f, 0.f, 0.f};
    if (WIDE) {
      if (live)
        for (int z = grp; z < splits; z += 16)
          v += *reinterpret_cast<const f32x4*>(p.slab + ((long)z * p.M + row) * p.Nn + col);
      red[grp * 16 + qi] = v;
      __syncthreads();
      if (grp == 0) {
        for (int g = 1; g < 16; ++g) v += red[g * 16 + qi];
      }
      __syncthreads();
      if (grp != 0 || !live) continue;
    } else {
      v = *reinterpret_cast<const f32x4*>(p.slab + (long)row * p.Nn + col);
      for (int z = 1; z < splits; ++z)
        v += *reinterpret_cast<const f32x4*>(p.slab + ((long)z * p.M + row) * p.Nn + col);
    }
    if (rows_are_taps) {
      const int tap = row / p.Cs, c = row - tap * p.Cs;
      *reinterpret_cast<f32x4*>(p.out + (long)tap * p.o_tap + (long)c * p.o_row + col) = v;
    } else {
      if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + col);
      if (p.addend) v += *reinterpret_cast<const f32x4*>(p.addend + (long)row * p.ld_add + col);
      float* o = p.out + out_pixel(p, row) * p.ld_out + col;
      if (p.accumulate) v += *reinterpret_cast<const f32x4*>(o);
      *reinterpret_cast<f32x4*>(o) = v;
    }
  }
}

// ------------------------------------------------------------------------------------------
// host side: tile selection and launch
// ------------------------------------------------------------------------------------------
struct Plan {
  int bm, bn, splits, nk_total, nk_per_split, tiles_m, tiles_n;
};

// tiles whose fast row kernels have a split-K-combining instantiation: the planner's (64-row tiles,
// columns <= 80); a forced wider plan keeps the separate reduce launch
constexpr bool splitk_combine_tile(int bm, int bn) { return bm == 64 && bn <= 80; }
static inline bool splitk_combine_ok(const Plan& pl) {
  return pl.splits > 1 && splitk_combine_tile(pl.bm, pl.bn);
}

static const int kBN[6] = {128, 96, 80, 64, 48, 32};
constexpr size_t kMaxSlabBytes = 96u << 20;

// What the last implicit-GEMM launch of this thread was (gs_debug_last_conv_launch) and how many
// launches each (op, K loop) pair has seen in this process (gs_debug_conv_launch_counts): lets the
// parity tests assert WHICH K loop produced the numbers they compare (capi_misc.hip owns the storage).
extern thread_local gs_debug_launch g_last_launch;
extern long long g_launch_counts[3][GS_KLOOP_COUNT][3];
extern double g_launch_flops[3][GS_KLOOP_COUNT];   // algorithmic 2*M*N*K per (op, K loop)
extern double g_k3_flops[GS_KLOOP_COUNT];          // the same for the role-1 (bottleneck conv2) forward launches
// (the forward thread, the autograd thread and their side-stream launches all add to these:
// flops_add in common.h)
static inline void note_launch(int op, int kloop, const Plan& pl, bool aff, int bw_mode,
                               double flops = 0.0) {
  flops_add(&g_launch_flops[op][kloop], flops);
  g_last_launch = gs_debug_launch{op, kloop, pl.bm, pl.bn, pl.splits, pl.nk_per_split, aff ? 1 : 0,
                                  bw_mode};
  // (mode 3 = mode 2's mask read from bytes: counted with mode 2, the record keeps the 3)
  const int bwi = bw_mode == 3 ? 2 : (bw_mode < 0 || bw_mode > 2 ? 0 : bw_mode);
  __atomic_fetch_add(&g_launch_counts[op][kloop][bwi], 1LL, __ATOMIC_RELAXED);
}

// Tuning knobs (read once): GS_WG_TARGET = workgroups a launch should reach before we stop
// shrinking tiles / splitting K (default 2 per CU); GS_MIN_KSTEPS = K steps per split at least.
static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}
static bool wg_target_forced() { static const bool v = getenv("GS_WG_TARGET") != nullptr; return v; }
static int wg_target() { static const int v = env_int("GS_WG_TARGET", 2 * num_cu()); return v; }
static int dyn_lds() { static const int v = env_int("GS_DYN_LDS", 0); return v; }
static int pair_min_ksteps() { static const int v = env_int("GS_PAIR_MIN", 16); return v; }
static int min_ksteps() { static const int v = env_int("GS_MIN_KSTEPS", 4); return v; }

extern long long g_col_finalized;    // capi_misc.hip: launches that merged their tile partials themselves
extern long long g_splitk_combined;   // capi_misc.hip: split-K launches that combined their slabs themselves
extern int g_force_plan[3];  // capi_misc.hip: {bm, bn, splits} set by gs_debug_force_plan (0 = off)

static Plan make_plan(int M, int Nn, int Ktot, bool allow_split, int max_splits = 64,
                      bool pipelined = true, double wg_per_cu = 2.5) {
  Plan pl{};
  if (g_force_plan[0] > 0) {  // tuning sweeps (tools/sweep_conv_plans.py)
    pl.bm = g_force_plan[0]; pl.bn = g_force_plan[1];
    pl.tiles_m = (int)ceil_div(M, pl.bm); pl.tiles_n = (int)ceil_div(Nn, pl.bn);
    pl.nk_total = (int)ceil_div(Ktot, BK);
    int splits = std::max(1, std::min(g_force_plan[2], pl.nk_total));
    if (!allow_split) splits = 1;
    pl.nk_per_split = (int)ceil_div(pl.nk_total, splits);
    pl.splits = (int)ceil_div(pl.nk_total, pl.nk_per_split);
    return pl;
  }
  if (pipelined && !wg_target_forced() && env_int("GS_FORCE_BM", 0) == 0) {
    // Pipelined kernels (rows and wgrad): 64-row tiles and a small cost model over the column
    // width {80,64,48,32} and the split factor, fitted to the r01 plan sweeps
    // (tools/sweep_conv_plans.py over the supernet's GEMM shapes at 1024x512 bs 2,
    // profiles/r01_plan_sweep*.json).  In units of one K step of a 64x64 tile:
    //   cost = rounds * (ksteps_per_wg + 4) * (bn/64) * width_eff * occupancy_eff  +  slab
    // with rounds = ceil(workgroups / 256 CUs) -- the quantisation matters: 640 workgroups take as
    // long as 768 --, occupancy_eff = 1.2 / 1.08 / 1 at 1 / 2 / >=3 rounds (fill and drain
    // overlap only across co-resident workgroups), and slab = (2s+1) * M * N * 1.4e-6 for the
    // split-K partials that are written and re-read.  Against the nearest-to-2.5-WGs/CU rule this
    // model is 10 % (forward), 4 % (dgrad) and 5 % (wgrad) faster over the sweep and within 1.5 %
    // of the per-shape optimum.  Plans are cached per shape (the search is ~2k evaluations).
    struct Key { int M, N, K, ms; bool operator==(const Key& o) const { return M == o.M && N == o.N && K == o.K && ms == o.ms; } };
    struct KeyHash { size_t operator()(const Key& k) const { return ((size_t)k.M * 1000003u) ^ ((size_t)k.N << 20) ^ ((size_t)k.K * 7919u) ^ (size_t)k.ms; } };
    static thread_local std::unordered_map<Key, Plan, KeyHash> cache;
    const Key key{M, Nn, Ktot, allow_split ? max_splits : 1};
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    static const int kBNp[4] = {80, 64, 48, 32};
    static const double kEff[4] = {1.03, 1.0, 1.05, 1.2};
    pl.bm = 64;
    pl.tiles_m = (int)ceil_div(M, 64);
    pl.nk_total = (int)ceil_div(Ktot, BK);
    const int s_max = allow_split ? std::max(1, std::min(max_splits, pl.nk_total / min_ksteps())) : 1;
    double best_cost = 1e300;
    int best_bn = 64, best_s = 1;
    for (int i = 0; i < 4; ++i) {
      const int bn = kBNp[i];
      const long tiles = (long)pl.tiles_m * ceil_div(Nn, bn);
      for (int sp = 1; sp <= s_max; ++sp) {
        if (sp > 1 && (size_t)sp * M * Nn * sizeof(float) > kMaxSlabBytes) break;
        const long rounds = ceil_div(tiles * sp, num_cu());
        const double occ = rounds == 1 ? 1.2 : (rounds == 2 ? 1.08 : 1.0);
        double cost = (double)rounds * ((double)ceil_div(pl.nk_total, sp) + 4.0) * (bn / 64.0) *
                      kEff[i] * occ;
        static const double slab_cost = 1.4e-6 * env_int("GS_SLAB_COST_PCT", 100) / 100.0;
        if (sp > 1) cost += (2.0 * sp + 1.0) * (double)M * (double)Nn * slab_cost;
        if (cost < best_cost - 1e-9) { best_cost = cost; best_bn = bn; best_s = sp; }
      }
    }
    pl.bn = best_bn;
    pl.tiles_n = (int)ceil_div(Nn, pl.bn);
    pl.nk_per_split = (int)ceil_div(pl.nk_total, best_s);
    pl.splits = (int)ceil_div(pl.nk_total, pl.nk_per_split);
    static const int dbg = env_int("GS_PLAN_DEBUG", 0);
    if (dbg) fprintf(stderr, "[gs plan] M=%d N=%d K=%d max_splits=%d -> 64x%d splits %d (ksteps %d)\n", M, Nn, Ktot, key.ms, pl.bn, pl.splits, pl.nk_per_split);
    cache.emplace(key, pl);
    return pl;
  }
  // BN: least padded width, larger tile on ties
  int best = 32, best_pad = 1 << 30;
  static const int bn_cap = env_int("GS_BN_CAP", 128);
  for (int i = 0; i < 6; ++i) {
    const int bn = kBN[i];
    if (bn > bn_cap) continue;
    const int pad = (int)ceil_div(Nn, bn) * bn;
    if (pad < best_pad) { best_pad = pad; best = bn; }
  }
  pl.bn = best;
  pl.tiles_n = (int)ceil_div(Nn, pl.bn);
  pl.nk_total = (int)ceil_div(Ktot, BK);
  // BM: 128-row tiles run ~15 % faster per FLOP than 64-row ones (each wave owns two 16-row
  // MFMA tiles per B fragment), so prefer them whenever tiles x attainable K-splits still
  // fills the chip; parallelism then comes from split-K (r01 sweep: s3/s4 3x3 +12 %, FCN head
  // conv 82 -> 100 TF).
  const long t128 = ceil_div(M, 128) * pl.tiles_n;
  // Workgroups a launch should reach.  The software-pipelined row kernels keep the MFMA pipe fed
  // with ONE wave per SIMD, so one workgroup per CU is enough; two per CU still run ~8 % faster
  // per FLOP, which only pays while the K range left to each workgroup stays long compared with the
  // split-K slab it must then write and the reduce must re-read (r01 sweep: s3 1x1 dgrad 41.6 ->
  // 28.0 us unsplit, FCN-head 3x3 188 vs 200 us with 16 instead of 8 splits).
  long target = wg_target();
  if (pipelined && !wg_target_forced()) {
    const long s_hi = std::max<long>(1, ceil_div(2L * num_cu(), t128));
    target = (pl.nk_total / s_hi >= 48) ? 2 * num_cu() : num_cu();
  }
  const long can_split = allow_split ? std::min<long>(max_splits, std::max(1, pl.nk_total / min_ksteps())) : 1;
  static const int force_bm = env_int("GS_FORCE_BM", 0);
  const long pad128 = ceil_div(M, 128) * 128, pad64 = ceil_div(M, 64) * 64;
  const bool wasteful = pad128 * 100 > pad64 * 115;  // e.g. M = 64 rows of a stage-1 1x1 wgrad
  pl.bm = force_bm ? force_bm : ((t128 * can_split >= target && !wasteful) ? 128 : 64);
  pl.tiles_m = (int)ceil_div(M, pl.bm);
  const long tiles = (long)pl.tiles_m * pl.tiles_n;
  int splits = 1;
  if (allow_split && tiles < target && pl.nk_total >= 2 * min_ksteps()) {
    splits = (int)ceil_div(target, tiles);
    const int max_by_k = pl.nk_total / min_ksteps();
    if (splits > max_by_k) splits = max_by_k;
    if (splits > max_splits) splits = max_splits;
    while (splits > 1 && (size_t)splits * M * Nn * sizeof(float) > kMaxSlabBytes) --splits;
    if (splits < 1) splits = 1;
  }
  pl.nk_per_split = (int)ceil_div(pl.nk_total, splits);
  pl.splits = (int)ceil_div(pl.nk_total, pl.nk_per_split);
  return pl;
}

template <bool BTRANS, bool DIVS, bool SCALAR, int KS>
static void launch_rows(const Plan& pl, const IgemmArgs& a, hipStream_t st) {
  const dim3 grid(pl.tiles_m * pl.tiles_n, pl.splits), block(NT);
  note_launch(BTRANS ? GS_OP_DGRAD : GS_OP_FORWARD, GS_KLOOP_GENERIC, pl, false, 0,
              2.0 * a.M * (double)a.Nn * a.Ktot);
#define GS_ROWS(BM_, BN_)                                                                     \
  if (pl.bm == BM_ && pl.bn == BN_) {                                                         \
    hipLaunchKernelGGL((igemm_rows_kernel<BM_, BN_, BTRANS, DIVS, SCALAR, KS>), grid, block, 0, st, a); \
    return;                                                                                   \
  }
  GS_ROWS(128, 128) GS_ROWS(128, 96) GS_ROWS(128, 80) GS_ROWS(128, 64) GS_ROWS(128, 48) GS_ROWS(128, 32)
  GS_ROWS(64, 128) GS_ROWS(64, 96) GS_ROWS(64, 80) GS_ROWS(64, 64) GS_ROWS(64, 48) GS_ROWS(64, 32)
#undef GS_ROWS
}

// K loop of a fast row launch (GS_KLOOP_*).  bf16x3 contraction (see x3_k_loop): stride-1 dgrad,
// 64-row tiles, BN 64 / 48.  r02 sweep over the supernet's data-gradient shapes
// (profiles/r02_bf16x3_probe.md): +7.5 % in sum against the fp32 loop, ahead everywhere except short
// split-K ranges (a split's 16 K steps are 8 bf16 steps: the fill does not amortise) and
// one-workgroup-per-CU launches (its single LDS stage wants co-resident workgroups to hide the two
// barriers per step: s3 1x1 256->1024, 30.5 vs 25.6 us), which keep the fp32 loop.  GS_X3=0 switches
// it off, GS_X3=n (n > 1) raises the minimum K steps per workgroup (3 K steps = 2 bf16 steps, a
// quarter wasted: -10 %).  The forward runs on it only behind GS_X3_FWD=<min K steps> (the [k][n]
// weights are staged with eight dword loads per thread and step: level with the fp32 loop).
// Long K ranges otherwise run two K steps per barrier (pipelined_k_loop_pairs) when the launch has at
// most three workgroups per CU anyway (its four LDS stages allow no more); big grids and short K
// ranges keep the two-stage loop, whose smaller footprint lets five workgroups per CU overlap their
// fill / drain (r01 A/B: s2..s4 3x3 and the head convs +3..7 %, s1 3x3 -3 % if paired).
// forward on the bf16x3 loop: 0 = never, 1 = every 3x3 where it measured ahead of the fp32 loops, 2 =
// wherever the loop's gate admits it (tests, sweeps), 3 = the split-K 3x3s only (DEFAULT); GS_X3_FWD
// sets the initial value, gs_debug_set_x3_fwd changes it at run time.
// Why not everywhere it is faster (K3 +9 % at stage 1): the bf16x3 contraction is ~1.3-1.5x noisier
// than the exact fmaf chain of the fp32 MFMA (the bf16 MFMA's internal accumulation: see kX3Terms) and
// forward noise is amplified by every layer behind it.  With mode 1 the median error ratio of the
// ill-conditioned parameter gradients against the fp32 oracle rose from 1.15 to 1.53 on config 4
// (bound 1.5, tests/parity.py) and three parameters of config 3 left the 3x bound.  Mode 3 keeps the
// early layers on the fp32 MFMA and takes the loop only where a 3x3 is split along K -- stages 3-4 at
// bs 2 and the heads' big-K convs, +3..5 % per launch: the margins of the full-size tests do not move
// (config 4 median 1.13 vs 1.15, config 3 p90 1.72 vs 1.70, largest ratio 2.06 both), K3 on the
// sampled mix 0.541 -> 0.551 of the fp32 peak, the step +0.65 % (A/B/A/B on one box).
extern int g_x3_fwd;   // capi_misc.hip (-1 = not yet read from the environment)
static inline int x3_fwd_mode() {
  if (g_x3_fwd < 0) g_x3_fwd = env_int("GS_X3_FWD", 3);
  return g_x3_fwd;
}
static inline bool pair_loop_ok(const Plan& pl) {
  return pl.nk_per_split >= pair_min_ksteps() &&
         (long)pl.tiles_m * pl.tiles_n * pl.splits <= 3L * num_cu();
}
static inline bool x3_grid_ok(const Plan& pl, int min_ksteps_) {
  // (r03 A/B: 32 instead of 48 puts the MIN anchor's split-K launches on the loop as well -- MIN +1 %,
  // sampled mix +-0; kept at 48)
  static const int split_min = env_int("GS_X3_SPLIT_MIN", 48);
  return min_ksteps_ > 0 && pl.bm == 64 && pl.nk_per_split >= min_ksteps_ &&
         (pl.splits == 1 || pl.nk_per_split >= split_min) &&
         (long)pl.tiles_m * pl.tiles_n * pl.splits >= 2L * num_cu();
}
template <bool BTRANS>
static inline int rows_fast_kloop(const Plan& pl, bool in_affine, int ks = 3) {
  if constexpr (BTRANS) {
    static const int x3_min = env_int("GS_X3", 4);
    if (x3_grid_ok(pl, x3_min) && (pl.bn == 64 || pl.bn == 48)) return GS_KLOOP_BF16X3;
  } else {
    // Forward (r03: the [k][n] weights staged one k row per thread and step, x3_k_loop<BFWD>).
    // Per shape against the fp32 loops, kernels alone (profiles/r03_fwd_x3_per_shape.md): 3x3 at
    // stage 1 +9 % (one K step per barrier there), the split-K 3x3s of stages 3-4 +3..5 %, the
    // unsplit 3x3 of stage 2 -4 % (the two-steps-per-barrier fp32 loop wins), 1x1s -10..+8 %
    // without a pattern.  Production: 3x3 only, and not where the paired fp32 loop runs unsplit.
    const int mode = x3_fwd_mode();
    if (mode > 0 && !in_affine && x3_grid_ok(pl, 4) && (pl.bn == 64 || pl.bn == 48)) {
      if (mode == 2) return GS_KLOOP_BF16X3;
      if (mode == 3) {   // only the split-K 3x3s (stages 3-4: late layers, the least amplification)
        if (ks == 3 && pl.splits > 1) return GS_KLOOP_BF16X3;
      } else if (ks == 3 && !(pair_loop_ok(pl) && pl.splits == 1)) {
        return GS_KLOOP_BF16X3;
      }
    }
  }
  return pair_loop_ok(pl) ? GS_KLOOP_FP32_PAIRS : GS_KLOOP_FP32;
}

// fused_layers.hip: the live timer of the role-1 (K3) launches.  If a timer interval is open for the
// launch about to be issued on `st`: single_kernel (the launch is the whole op: unsplit, or split-K
// combined inside the launch) -> true and the two events to attach to the kernel itself
// (hipExtLaunchKernelGGL: they carry the dispatch's own begin / end timestamps, what rocprofv3's kernel
// trace reports); otherwise a marker event is recorded in front of the launch and the caller's
// k3_prof_end records the closing one behind the reduce launch.
bool k3_launch_events(hipStream_t st, bool single_kernel, hipEvent_t* e0, hipEvent_t* e1);

template <class K>
static inline void launch_rows_kernel(K kernel, const dim3& grid, const dim3& block, int lds_dyn,
                                      hipStream_t st, const IgemmArgs& a, hipEvent_t e0, hipEvent_t e1) {
  if (e0) hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)lds_dyn, st, e0, e1, 0u, a);
  else hipLaunchKernelGGL(kernel, grid, block, lds_dyn, st, a);
}

template <bool BTRANS, int KS, int ROLE = 0>
static void launch_rows_fast(const Plan& pl, const IgemmArgs& a_in, hipStream_t st) {
  IgemmArgs a = a_in;
  a.nsplits = pl.splits;
  {  // the larger operand should cross the fabric once: see the kernel's tile decode
    const double a_bytes = 4.0 * a.npix * a.Cs, b_bytes = 4.0 * a.taps * a.Cs * a.Nn;
    static const int force = env_int("GS_TILE_ORDER", -1);
    a.tile_order = force >= 0 ? force : (b_bytes > a_bytes ? 1 : 0);
  }
  const dim3 grid(pl.tiles_m * pl.tiles_n * pl.splits), block(NT);
  const int kloop = rows_fast_kloop<BTRANS>(pl, a.a_coeffs != nullptr, KS);
  const bool pair = kloop == GS_KLOOP_FP32_PAIRS;
  note_launch(BTRANS ? GS_OP_DGRAD : GS_OP_FORWARD, kloop, pl, a.a_coeffs != nullptr, a.bw_mode,
              2.0 * a.M * (double)a.Nn * a.Ktot);
  if (ROLE == 1 && !BTRANS) flops_add(&g_k3_flops[kloop], 2.0 * a.M * (double)a.Nn * a.Ktot);
  if (a.tickets && splitk_combine_ok(pl)) __atomic_fetch_add(&g_splitk_combined, 1LL, __ATOMIC_RELAXED);
  else a.tickets = nullptr;
  // (GS_SKL: the extended-epilogue instantiation when the launch carries arrival counters)
  if (!splitk_combine_tile(pl.bm, pl.bn)) a.col_tickets = nullptr;   // (callers check; see column_tickets)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (ROLE == 1 && !BTRANS && !k3_launch_events(st, pl.splits == 1 || a.tickets != nullptr, &ev0, &ev1))
    ev0 = ev1 = nullptr;
#define GS_SKL(...)                                                                            \
  do {                                                                                         \
    if (a.tickets || a.col_tickets)                                                            \
      launch_rows_kernel((igemm_rows_fast_kernel<__VA_ARGS__, true>), grid, block, lds_dyn, st, a, ev0, ev1);   \
    else                                                                                       \
      launch_rows_kernel((igemm_rows_fast_kernel<__VA_ARGS__, false>), grid, block, lds_dyn, st, a, ev0, ev1);  \
  } while (0)
#define GS_PLAIN(...) \
  launch_rows_kernel((igemm_rows_fast_kernel<__VA_ARGS__, false>), grid, block, lds_dyn, st, a, ev0, ev1)
  int lds_dyn = 0;
  if (kloop == GS_KLOOP_BF16X3) {
    if constexpr (BTRANS) {
      if (pl.bn == 64) GS_SKL(64, 64, true, KS, 0, ROLE, true, false, false, true);
      else GS_SKL(64, 48, true, KS, 0, ROLE, true, false, false, true);
    } else if (pl.bn == 64) {
      GS_SKL(64, 64, false, KS, 0, ROLE, true, false, false, true);
    } else {
      GS_SKL(64, 48, false, KS, 0, ROLE, true, false, false, true);
    }
    return;
  }
  if constexpr (!BTRANS) {
    if (a.a_coeffs) {   // relu(bn(x)) evaluated in the loader: 64-row tiles (the planner's choice)
#define GS_FAST_AFF(BN_)                                                                   \
  if (pl.bm == 64 && pl.bn == BN_) {                                                       \
    if constexpr (splitk_combine_tile(64, BN_)) {                                          \
      if (pair) GS_SKL(64, BN_, false, KS, 0, ROLE, true, true, true, false);              \
      else GS_SKL(64, BN_, false, KS, 0, ROLE, true, false, true, false);                  \
    } else {                                                                               \
      if (pair) GS_PLAIN(64, BN_, false, KS, 0, ROLE, true, true, true, false);            \
      else GS_PLAIN(64, BN_, false, KS, 0, ROLE, true, false, true, false);                \
    }                                                                                      \
    return;                                                                                \
  }
      GS_FAST_AFF(128) GS_FAST_AFF(96) GS_FAST_AFF(80) GS_FAST_AFF(64) GS_FAST_AFF(48) GS_FAST_AFF(32)
#undef GS_FAST_AFF
      return;   // (unreachable: conv_in_affine_ok() admits only plans with 64-row tiles)
    }
  }
  lds_dyn = dyn_lds();
#define GS_FAST(BM_, BN_)                                                                  \
  if (pl.bm == BM_ && pl.bn == BN_) {                                                      \
    if constexpr (splitk_combine_tile(BM_, BN_)) {                                         \
      if (pair) GS_SKL(BM_, BN_, BTRANS, KS, 0, ROLE, true, true, false, false);           \
      else GS_SKL(BM_, BN_, BTRANS, KS, 0, ROLE, true, false, false, false);               \
    } else {                                                                               \
      if (pair) GS_PLAIN(BM_, BN_, BTRANS, KS, 0, ROLE, true, true, false, false);         \
      else GS_PLAIN(BM_, BN_, BTRANS, KS, 0, ROLE, true, false, false, false);             \
    }                                                                                      \
    return;                                                                                \
  }
  GS_FAST(128, 128) GS_FAST(128, 96) GS_FAST(128, 80) GS_FAST(128, 64) GS_FAST(128, 48) GS_FAST(128, 32)
  GS_FAST(64, 128) GS_FAST(64, 96) GS_FAST(64, 80) GS_FAST(64, 64) GS_FAST(64, 48) GS_FAST(64, 32)
#undef GS_FAST
#undef GS_SKL
#undef GS_PLAIN
}

// the fast row kernel needs: NHWC vector source, channels per tap % BK == 0, 1x1 or 3x3
static inline bool fast_rows_ok(int cs, int ks, size_t src_bytes, size_t dense_bytes) {
  return (ks == 1 || ks == 3) && (cs % BK) == 0 && src_bytes < (1ull << 31) &&
         dense_bytes < (1ull << 31);
}

template <int KS>
static void launch_wgrad_fast(const Plan& pl, const IgemmArgs& a_in, hipStream_t st) {
  IgemmArgs a = a_in;
  a.nsplits = pl.splits;
  const dim3 grid(pl.tiles_m * pl.tiles_n * pl.splits), block(NT);
  const bool pair = pair_loop_ok(pl);
  note_launch(GS_OP_WGRAD, pair ? GS_KLOOP_FP32_PAIRS : GS_KLOOP_FP32, pl, a.a_coeffs != nullptr, 0,
              2.0 * a.M * (double)a.Nn * a.Ktot);
  static const int no_walign = env_int("GS_NO_WALIGN", 0);
  const bool walign = !no_walign && a.Wp % BK == 0;
  if (a.a_coeffs) {
#define GS_WGF_AFF(BN_)                                                                   \
  if (pl.bm == 64 && pl.bn == BN_) {                                                      \
    if (pair && walign)                                                                   \
      hipLaunchKernelGGL((igemm_wgrad_fast_kernel<64, BN_, KS, true, true, true>), grid, block, 0, st, a); \
    else if (pair)                                                                        \
      hipLaunchKernelGGL((igemm_wgrad_fast_kernel<64, BN_, KS, true, false, true>), grid, block, 0, st, a); \
    else if (walign)                                                                      \
      hipLaunchKernelGGL((igemm_wgrad_fast_kernel<64, BN_, KS, false, true, true>), grid, block, 0, st, a); \
    else                                                                                  \
      hipLaunchKernelGGL((igemm_wgrad_fast_kernel<64, BN_, KS, false, false, true>), grid, block, 0, st, a); \
    return;                                                                               \
  }
    GS_WGF_AFF(128) GS_WGF_AFF(96) GS_WGF_AFF(80) GS_WGF_AFF(64) GS_WGF_AFF(48) GS_WGF_AFF(32)
#undef GS_WGF_AFF
    return;
  }
#define GS_WGF(BM_, BN_)                                                                  \
  if (pl.bm == BM_ && pl.bn == BN_) {                                                     \
    if (pair && walign)                                                                   \
      hipLaunchKernelGGL((igemm_wgrad_fast_kernel<BM_, BN_, KS, true, true>), grid, block, dyn_lds(), st, a); \
    else if (pair)                                                                        \
      hipLaunchKernelGGL((igemm_wgrad_fast_kernel<BM_, BN_, KS, true, false>), grid, block, dyn_lds(), st, a); \
    else if (walign)                                                                      \
      hipLaunchKernelGGL((igemm_wgrad_fast_kernel<BM_, BN_, KS, false, true>), grid, block, dyn_lds(), st, a); \
    else                                                                                  \
      hipLaunchKernelGGL((igemm_wgrad_fast_kernel<BM_, BN_, KS, false, false>), grid, block, dyn_lds(), st, a); \
    return;                                                                               \
  }
  GS_WGF(128, 128) GS_WGF(128, 96) GS_WGF(128, 80) GS_WGF(128, 64) GS_WGF(128, 48) GS_WGF(128, 32)
  GS_WGF(64, 128) GS_WGF(64, 96) GS_WGF(64, 80) GS_WGF(64, 64) GS_WGF(64, 48) GS_WGF(64, 32)
#undef GS_WGF
}

template <bool SCALAR, int KS>
static void launch_wgrad(const Plan& pl, const IgemmArgs& a, hipStream_t st) {
  const dim3 grid(pl.tiles_m * pl.tiles_n, pl.splits), block(NT);
  note_launch(GS_OP_WGRAD, GS_KLOOP_GENERIC, pl, false, 0, 2.0 * a.M * (double)a.Nn * a.Ktot);
#define GS_WG(BM_, BN_)                                                                \
  if (pl.bm == BM_ && pl.bn == BN_) {                                                  \
    hipLaunchKernelGGL((igemm_wgrad_kernel<BM_, BN_, SCALAR, KS>), grid, block, 0, st, a); \
    return;                                                                            \
  }
  GS_WG(128, 128) GS_WG(128, 96) GS_WG(128, 80) GS_WG(128, 64) GS_WG(128, 48) GS_WG(128, 32)
  GS_WG(64, 128) GS_WG(64, 96) GS_WG(64, 80) GS_WG(64, 64) GS_WG(64, 48) GS_WG(64, 32)
#undef GS_WG
}

static int check_desc(const gs_conv_desc* d) {
  if (!d) return GS_E_NULL;
  if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->Ci <= 0 || d->Co <= 0) return GS_E_BADARG;
  if (d->KH <= 0 || d->KW <= 0 || d->stride <= 0 || d->dil <= 0 || d->pad < 0) return GS_E_BADARG;
  if (d->role < 0 || d->role > GS_CONV_ROLE_BOTTLENECK3X3 || d->reserved != 0) return GS_E_BADARG;
  if (d->Ci > d->Ci_max || d->Co > d->Co_ld) return GS_E_BADARG;
  if ((d->Co & 3) || (d->Co_ld & 3) || (d->ldy & 3) || d->ldy < d->Co) return GS_E_ALIGN;
  const int ho = (d->H + 2 * d->pad - d->dil * (d->KH - 1) - 1) / d->stride + 1;
  const int wo = (d->W + 2 * d->pad - d->dil * (d->KW - 1) - 1) / d->stride + 1;
  if (ho != d->Ho || wo != d->Wo || ho <= 0 || wo <= 0) return GS_E_BADARG;
  if ((long)d->N * d->H * d->W >= (1L << 31) || (long)d->N * ho * wo >= (1L << 31)) return GS_E_BADARG;
  return GS_OK;
}

static bool x_is_vector(const gs_conv_desc* d) {
  return d->x_sc == 1 && (d->Ci & 3) == 0 && (d->x_sw & 3) == 0 && (d->x_sh & 3) == 0 &&
         (d->x_sn & 3) == 0;
}

static Plan plan_fwd(const gs_conv_desc* d) {
  return make_plan(d->N * d->Ho * d->Wo, d->Co, d->KH * d->KW * d->Ci, true);
}
static Plan plan_dgrad(const gs_conv_desc* d) {
  return make_plan(d->N * d->H * d->W, d->Ci, d->KH * d->KW * d->Co, true);
}
static Plan plan_wgrad(const gs_conv_desc* d) {
  // wgrad: K runs over pixels (up to 131072 at stage 1) while M x N is tiny: allow deep split-K
  static const int old_plan = env_int("GS_WGRAD_OLD_PLAN", 0);
  if (old_plan) return make_plan(d->KH * d->KW * d->Ci, d->Co, d->N * d->Ho * d->Wo, true, 512, false);
  return make_plan(d->KH * d->KW * d->Ci, d->Co, d->N * d->Ho * d->Wo, true, 512, true, 4.0);
}
static size_t slab_bytes(const Plan& pl, long M, int Nn) {
  return pl.splits > 1 ? (size_t)pl.splits * M * Nn * sizeof(float) : 0;
}

static inline void launch_reduce(const IgemmArgs& a, int splits, int rows_are_taps, hipStream_t st,
                                 int role = 0) {
  const long total = (long)a.M * (a.Nn / 4);
  if (role == 1 && splits < 48) {
    hipLaunchKernelGGL((splitk_reduce_kernel<false, 1>), dim3(stream_grid(total, 256)), dim3(256), 0,
                       st, a, splits, rows_are_taps);
  } else if (splits >= 48 || (splits >= 16 && total < 65536)) {
    // few outputs per slab: spread the slabs over 16 thread groups (one thread per output quad would
    // leave most of the chip idle behind `splits` loads each)
    const int grid = (int)std::min<long>(ceil_div(total, 16), (long)num_cu() * 16);
    hipLaunchKernelGGL(splitk_reduce_kernel<true>, dim3(grid), dim3(256), 0, st, a, splits,
                       rows_are_taps);
  } else {
    hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3(stream_grid(total, 256)), dim3(256), 0, st,
                       a, splits, rows_are_taps);
  }
}

// ---- strided dgrad as s*s stride-1 sub-problems (one per input-pixel parity class) ----
// For input row h = hq*s + ph the taps kh with (ph + pad - kh*dil) % s == 0 contribute, reading
// dy row hq + (ph + pad - kh*dil)/s.  Those taps form an arithmetic progression, so each class is
// an ordinary gather-GEMM over a sub-sampled tap grid: no MFMA work is spent on structural zeros
// (the single-launch form wastes 1 - 1/s^2 of it).
struct TapAxis {
  int n;      // number of valid taps
  int k0;     // first valid tap
  int dk;     // tap step
  int off0;   // source offset of the first valid tap
  int step;   // source offset step per valid tap
};
static inline TapAxis tap_axis(int ph, int pad, int dil, int s, int K) {
  TapAxis a{0, 0, 1, 0, 0};
  int first = -1, second = -1;
  for (int k = 0; k < K; ++k) {
    const int num = ph + pad - k * dil;
    if (((num % s) + s) % s == 0) {
      if (first < 0) first = k;
      else if (second < 0) second = k;
      ++a.n;
    }
  }
  if (a.n == 0) return a;
  a.k0 = first;
  a.dk = second > 0 ? second - first : 1;
  a.off0 = (ph + pad - first * dil) / s;          // exact division
  a.step = -(a.dk * dil) / s;
  return a;
}
static inline int class_len(int L, int s, int ph) { return L > ph ? (L - ph + s - 1) / s : 0; }

// gs_conv_desc::in_affine (relu(bn(x)) in the operand loaders) needs the fast forward and wgrad
// kernels with 64-row tiles and the coefficient image in LDS
static inline bool conv_in_affine_ok(const gs_conv_desc* d) {
  if (!x_is_vector(d) || d->Ci > kAffMaxC || (d->Ci % BK) != 0) return false;
  const int ks = (d->KH == 1 && d->KW == 1) ? 1 : ((d->KH == 3 && d->KW == 3) ? 3 : 0);
  if (!ks) return false;
  const size_t src_b = (size_t)d->N * d->x_sn * sizeof(float);
  const size_t dense_b = (size_t)d->KH * d->KW * d->Ci_max * d->Co_ld * sizeof(float);
  if (!fast_rows_ok(d->Ci, ks, src_b, dense_b) || getenv("GS_NO_FAST")) return false;
  if ((d->Ci & 3) || (d->Co & 3)) return false;
  return plan_fwd(d).bm == 64 && plan_wgrad(d).bm == 64;
}

static inline int ksize_tag(const gs_conv_desc* d) {
  if (d->KH == 1 && d->KW == 1) return 1;
  if (d->KH == 3 && d->KW == 3) return 3;
  return 0;
}

}  // namespace gs
