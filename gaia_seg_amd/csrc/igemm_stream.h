// 1x1 convolutions with a short contraction (the bottleneck's conv1 / conv3 / projection shortcut at
// stages 1-2, gaiaseg/models/utils/dynamic_res_layer.py:84-125, and their stride-1 data gradients) as a
// STREAMING GEMM: Y[M, N] = X[M, K] * W[K, N] with M = 16k..64k pixels and K * BNW <= 32k weights.
//
// The tile kernel of igemm_core.h gives every 64 x 64 output tile its own workgroup: at K = 64 that is
// four K steps of MFMA work behind a full pipeline fill (global -> register -> LDS) and in front of
// an epilogue, 4096 times per launch, with the co-resident workgroups of a CU in lockstep (they start
// together, so they fill, compute and drain together; profiles/r02_wg_timeline.md: K loop 7.8 us of a
// 32.5 us launch).  Here instead
//   * ONE persistent 512-thread workgroup per CU (two waves per SIMD) owns a column block of BNW
//     outputs and a contiguous range of rows;
//   * the weight slice W[:, n0 : n0 + BNW] is staged into LDS ONCE, as a plain [k][BNW] image (the
//     four lane groups of a fragment read are whole rows apart, so the 16-byte reads are conflict
//     free without padding), and stays there;
//   * the activations never touch LDS: a wave owns 16-row strips, and a lane (row i = lane & 15,
//     quad q = lane >> 4) loads the EIGHT consecutive channels 8q..8q+7 of each 32-channel chunk of its
//     row straight into registers (two 16-byte loads; the four lanes of a row fetch 128 contiguous
//     bytes).  The MFMA sums over k in any order as long as both operands agree, so "k-group" j of a
//     chunk is channel 8q + j for lane group q: the A operand is register element j, the B fragment
//     comes from weight row 32c + 8q + j.  No transposing LDS stores, no fragment reads for A, and
//     -- since nothing is shared between waves after the weight fill -- NO BARRIER in the main loop:
//     the waves drift apart and share the SIMD's MFMA pipe instead of meeting at a barrier per step;
//   * global loads run kStreamRing - 1 chunks ahead of the MFMAs in a register ring, across strip
//     boundaries (the next strip's activations are in flight while a strip is written out);
//   * the epilogue stores straight from the accumulators (ColGroups: a lane holds four consecutive
//     columns, 16 lanes cover 256 contiguous bytes of a row) with bounds-checked buffer stores, and the
//     BatchNorm statistics of the output (forward) or the producer BatchNorm's backward sums (dgrad,
//     gs_bn_bwd_fuse) are accumulated in registers over ALL rows of the workgroup: one partial
//     record per workgroup instead of one per 64-row tile.
// Each wave owns 16 rows x BNW columns: BNW / 16 accumulator blocks, BNW / 16 MFMAs
// (v_mfma_f32_16x16x4_f32, exact fp32) per k-group, eight k-groups per chunk.
#pragma once
#include "igemm_core.h"

namespace gs {

constexpr int kStreamThreads = 512;
constexpr int kStreamWaves = kStreamThreads / 64;
constexpr int kStreamBM = 16 * kStreamWaves;       // rows a workgroup covers per round of strips
constexpr int kStreamRing = 4;                     // register ring: loads run 3 chunks ahead
constexpr int kStreamBFloats = 32768;              // resident weights: K * BNW <= 32k floats (128 KB)
constexpr int kStreamMaxK = 512;

template <int BNW>
struct StreamTile {
  static constexpr int TN = BNW / 16;
  static constexpr int COEF_FLOATS = 3 * kStreamMaxK;          // AFF: [mean | scale | beta][K]
  static constexpr int RED_FLOATS = kStreamWaves * 2 * BNW;    // end-of-kernel reduction over the waves
  static constexpr int LDS_FLOATS = kStreamBFloats + COEF_FLOATS + RED_FLOATS + BNW;
};

struct StreamPlan {
  int ok, bnw, ncb, row_groups, tiles_per_wg;    // a "tile" = kStreamBM rows (one strip per wave)
};

// 0 = off, 1 = where it measured ahead (default), 2 = wherever the weights fit (tests, sweeps);
// GS_STREAM sets the initial value, gs_debug_set_stream_mode changes it at run time.
extern int g_stream_mode;   // capi_misc.hip (-1 = not yet read from the environment)
static inline int stream_mode() {
  if (g_stream_mode < 0) g_stream_mode = env_int("GS_STREAM", 1);
  return g_stream_mode;
}

// Host: which column-block width, how the rows are dealt to the workgroups.
// ld_wide: the widest pixel stride among the buffers the epilogue addresses with 32-bit byte offsets
// (the output, and the y / activation operands of the fused BatchNorm-backward epilogue).
static inline StreamPlan stream_plan(long M, int Nn, int Cs, bool dgrad, long ld_wide) {
  StreamPlan sp{0, 0, 0, 0, 0};
  const int on = stream_mode();
  static const long min_rows = env_int("GS_STREAM_MIN_ROWS", 16384);
  if (!on || (Cs % BK) != 0 || (Nn & 3) || Cs > kStreamMaxK || M < min_rows || M >= (1L << 31) / 4)
    return sp;
  // epilogue offsets 4u * (m * ld + col) and the buffer descriptors' record counts are 32-bit: the
  // tile kernels (64-bit pointer arithmetic) take anything wider
  if (M * std::max<long>(ld_wide, Nn) * 4 >= (1L << 31)) return sp;
  const int kpad = (int)ceil_div(Cs, 32) * 32;   // whole 32-channel chunks
  int best = 0, best_pad = 1 << 30;
  for (int bnw : {256, 128, 64}) {
    if ((long)kpad * bnw > kStreamBFloats) continue;
    const int pad = (int)ceil_div(Nn, bnw) * bnw;
    if (pad < best_pad) { best_pad = pad; best = bnw; }
  }
  if (!best) return sp;
  sp.bnw = best;
  sp.ncb = (int)ceil_div(Nn, best);
  const int tiles = (int)ceil_div(M, kStreamBM);
  int rg = std::max(1, num_cu() / sp.ncb);
  if (rg > tiles) rg = tiles;
  sp.tiles_per_wg = (int)ceil_div(tiles, rg);
  sp.row_groups = (int)ceil_div(tiles, sp.tiles_per_wg);
  // worth it only when every wave streams at least two strips through the resident weights
  sp.ok = sp.tiles_per_wg >= 2 && (long)sp.row_groups * sp.ncb >= num_cu() / 2;
  // ... and where it measured ahead of the tile kernels (profiles/r03_conv1x1_stream.md, kernels
  // alone, 1024x512 bs 2): ONE column block, so that the activations are read exactly once (several
  // blocks re-read them per block: 0.75-0.9x), little column padding, and for data gradients only the
  // short contractions (the tile kernel runs those on the bf16x3 loop, and the transposed weight
  // fill costs ~3 us here).  GS_STREAM=2 lifts these restrictions (tests, sweeps).
  if (on < 2) {
    const bool padded = (long)sp.ncb * sp.bnw * 10 > (long)Nn * 11;
    if (sp.ncb != 1 || padded || (dgrad && Cs > 64)) sp.ok = 0;
  }
  return sp;
}

template <int BNW, bool BTRANS, bool AFF>
__global__ __launch_bounds__(kStreamThreads) void conv1x1_stream_kernel(const IgemmArgs p,
                                                                        const int row_groups,
                                                                        const int tiles_per_wg,
                                                                        const int ncb) {
  using S = StreamTile<BNW>;
  constexpr int TN = S::TN, NG = TN / 4, RING = kStreamRing;
  static_assert(TN % 4 == 0, "whole groups of four 16-column blocks");
  __shared__ __attribute__((aligned(16))) float lds[S::LDS_FLOATS];
  float* Bres = lds;                               // [kpad][BNW]
  float* coefL = lds + kStreamBFloats;             // AFF: [mean | scale | beta][Cs]
  float* red = coefL + S::COEF_FLOATS;             // [waves][2][BNW]
  float* shiftL = red + S::RED_FLOATS;             // forward statistics: the workgroup's first row

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int kk = lane >> 4, li = lane & 15;
  const int lin = xcd_remap(blockIdx.x, row_groups * ncb);
  const int rg = lin / ncb, cb = lin - rg * ncb;
  const int n0 = cb * BNW;
  const int tile0 = rg * tiles_per_wg;
  const int ntiles_all = (p.M + kStreamBM - 1) / kStreamBM;
  const int ntiles = min(tiles_per_wg, ntiles_all - tile0);     // strips of THIS wave
  const int nchunk = (p.Cs + 31) >> 5;             // 32-channel chunks per strip
  const int kpad = nchunk * 32;

  const __amdgpu_buffer_rsrc_t rs_src =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src), 0, p.src_bytes, 0x00020000);
  constexpr unsigned kOOB = 0xFFFFFFFFu;

  // ---- activations: lane (row li, quad kk) loads channels 32c + 8kk .. + 7 of its row ----
  int l_strip = 0, l_chunk = 0;                    // the next chunk to load: (strip, chunk in strip)
  auto gload = [&](f32x4 (&r)[2]) __attribute__((always_inline)) {
    const int m = (tile0 + l_strip) * kStreamBM + wave * 16 + li;
    const bool rv = l_strip < ntiles && m < p.M;
    const int k = l_chunk * 32 + kk * 8;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const unsigned off = (rv && k + 4 * h < p.Cs) ? 4u * ((unsigned)m * (unsigned)p.s_w + (unsigned)(k + 4 * h)) : kOOB;
      r[h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_src, off, 0, 0));
    }
    if (++l_chunk == nchunk) { l_chunk = 0; ++l_strip; }
  };
  // the first chunks are in flight while the weights are staged
  f32x4 ar[RING][2];
#pragma unroll
  for (int u = 0; u < RING - 1; ++u) gload(ar[u]);

  // ---- resident weights: plain [k][BNW] image (zeros past K / N) ----
  // A thread's loads are issued in batches of 8 quads before their LDS stores (a load -> wait -> store
  // loop is one serial round trip to L2 / HBM per quad).
  {
    const int count = kpad * (BNW / 4);            // quads of the image
    for (int base = 0; base < count; base += 8 * kStreamThreads) {
      f32x4 wv[8];
      if constexpr (!BTRANS) {     // W[k][n], row stride d_row: a straight copy
        constexpr int QN = BNW / 4;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int idx = base + t + u * kStreamThreads;
          const int k = idx / QN, nq = idx - k * QN;
          const int col = n0 + nq * 4;
          wv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (idx < count && k < p.Cs && col < p.n_lim)
            wv[u] = *reinterpret_cast<const f32x4*>(p.dense + (long)k * p.d_row + col);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int idx = base + t + u * kStreamThreads;
          if (idx < count) *reinterpret_cast<f32x4*>(&Bres[idx * 4]) = wv[u];
        }
      } else {                     // W[n][k] (k contiguous), row stride d_row: transposed into the image
        // consecutive lanes take consecutive n (LDS banks) for one k quad: the transposing stores
        // are conflict free (n fastest; the other order put all 64 lanes on one bank)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int idx = base + t + u * kStreamThreads;
          const int kq4 = idx / BNW, n = idx - kq4 * BNW;
          wv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (idx < count && n0 + n < p.Nn && kq4 * 4 < p.Cs)
            wv[u] = *reinterpret_cast<const f32x4*>(p.dense + (long)(n0 + n) * p.d_row + kq4 * 4);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int idx = base + t + u * kStreamThreads;
          const int kq4 = idx / BNW, n = idx - kq4 * BNW;
          if (idx < count) {
#pragma unroll
            for (int j = 0; j < 4; ++j) Bres[(kq4 * 4 + j) * BNW + n] = wv[u][j];
          }
        }
      }
    }
    if constexpr (AFF) {           // coefficient order in memory: scale, beta, mean, invstd
      const int cq = p.Cs >> 2;
      for (int i = t; i < 3 * cq; i += kStreamThreads) {
        const int which = i / cq, q = i - which * cq;
        const int srcrow = which == 0 ? 2 : (which == 1 ? 0 : 1);
        *reinterpret_cast<f32x4*>(coefL + which * p.Cs + q * 4) =
            *reinterpret_cast<const f32x4*>(p.a_coeffs + (long)srcrow * p.Cs + q * 4);
      }
    }
  }

  f32x4 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-workgroup sums of the epilogue (forward: BatchNorm statistics; dgrad: BatchNorm backward)
  const bool want_stats = !BTRANS && p.tile_stats != nullptr;
  const bool want_bnb = BTRANS && p.bw_mode != 0;
  f32x4 s1[NG], s2[NG], shift[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    s1[g] = f32x4{0.f, 0.f, 0.f, 0.f}; s2[g] = s1[g]; shift[g] = s1[g];
  }

  // Branch-free epilogue: bounds-checked buffer accesses (an offset of kOOB reads zeros / drops the
  // store) instead of a conditional block per row -- the first version spent 16 branches, each with
  // its own wait, per tile and wave.
  const unsigned out_bytes = (unsigned)(4u * ((unsigned)(p.M - 1) * (unsigned)p.ld_out + (unsigned)p.Nn));
  const __amdgpu_buffer_rsrc_t rs_out =
      __builtin_amdgcn_make_buffer_rsrc(p.out, 0, out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_bwy = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.bw_y), 0,
      want_bnb ? (unsigned)(4u * ((unsigned)(p.M - 1) * (unsigned)p.bw_ldy + (unsigned)p.Nn)) : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_bwa = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.bw_act), 0,
      (want_bnb && p.bw_mode == 2) ? (unsigned)(4u * ((unsigned)(p.M - 1) * (unsigned)p.bw_ldact + (unsigned)p.Nn)) : 0u,
      0x00020000);
  auto epilogue = [&](int tile_i) __attribute__((always_inline)) {
    const int mbase = (tile0 + tile_i) * kStreamBM + wave * 16 + kk * 4;
    if (want_stats && tile_i == 0) {
      // the statistics are accumulated around the workgroup's first row (well conditioned sums);
      // every wave reaches its first epilogue, so this is the main loop's only barrier
      if (wave == 0 && kk == 0) {
#pragma unroll
        for (int g = 0; g < NG; ++g)
          *reinterpret_cast<f32x4*>(shiftL + 64 * g + 4 * li) =
              f32x4{acc[4 * g][0], acc[4 * g + 1][0], acc[4 * g + 2][0], acc[4 * g + 3][0]};
      }
      __syncthreads();
#pragma unroll
      for (int g = 0; g < NG; ++g) shift[g] = *reinterpret_cast<const f32x4*>(shiftL + 64 * g + 4 * li);
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int col = n0 + 64 * g + 4 * li;
      const bool cv = col < p.Nn;
      f32x4 scale{0.f, 0.f, 0.f, 0.f}, beta = scale, mean = scale, invstd = scale;
      if (want_bnb && cv) {
        scale = *reinterpret_cast<const f32x4*>(p.bw_coeffs + col);
        beta = *reinterpret_cast<const f32x4*>(p.bw_coeffs + p.Nn + col);
        mean = *reinterpret_cast<const f32x4*>(p.bw_coeffs + 2 * p.Nn + col);
        invstd = *reinterpret_cast<const f32x4*>(p.bw_coeffs + 3 * p.Nn + col);
      }
      // issue every load of the group first, then combine and store
      f32x4 prev[4], yv[4], av[4];
      unsigned off[4], mbits[4];
      bool ok[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = mbase + r;
        ok[r] = cv && m < p.M;
        off[r] = ok[r] ? 4u * ((unsigned)m * (unsigned)p.ld_out + (unsigned)col) : kOOB;
        if constexpr (BTRANS) {
          if (p.accumulate)
            prev[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_out, off[r], 0, 0));
          if (want_bnb) {
            const unsigned oy = ok[r] ? 4u * ((unsigned)m * (unsigned)p.bw_ldy + (unsigned)col) : kOOB;
            yv[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_bwy, oy, 0, 0));
            if (p.bw_mode == 2) {
              const unsigned oa = ok[r] ? 4u * ((unsigned)m * (unsigned)p.bw_ldact + (unsigned)col) : kOOB;
              av[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_bwa, oa, 0, 0));
            } else if (p.bw_mode == 3) {
              mbits[r] = ok[r] ? (unsigned)p.bw_mask[(long)m * p.bw_ldmask + (col >> 2)] : 0u;
            }
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        f32x4 v{acc[4 * g][r], acc[4 * g + 1][r], acc[4 * g + 2][r], acc[4 * g + 3][r]};
        if constexpr (BTRANS) {
          if (p.accumulate) v += prev[r];
          if (want_bnb) {
            if (p.bw_mode == 3) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = ((mbits[r] >> e) & 1u) ? v[e] : 0.f;
            } else {
              const f32x4 key = p.bw_mode == 2 ? av[r] : (yv[r] - mean) * scale + beta;   // as bn_apply / masked_grad
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = (ok[r] && key[e] > 0.f) ? v[e] : 0.f;
            }
            s1[g] += v;
            s2[g] += v * ((yv[r] - mean) * invstd);
          }
        } else if (want_stats) {
          f32x4 dlt = v - shift[g];
          if (!ok[r]) dlt = f32x4{0.f, 0.f, 0.f, 0.f};
          s1[g] += dlt;
          s2[g] += dlt * dlt;
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_out, off[r], 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  // ---- one chunk: 8 k-groups; lane group kk contracts channel 32c + 8kk + j in k-group j ----
  // B fragments are fetched one k-group ahead (two fragment sets), so the LDS latency hides behind
  // the previous k-group's MFMAs.
  auto bfrag = [&](int row, float (&b)[TN]) __attribute__((always_inline)) {
    read_b_fragments<TN>(Bres + row * BNW, li, b);
  };
  auto compute = [&](const f32x4 (&r)[2], int chunk) __attribute__((always_inline)) {
    f32x4 v0 = r[0], v1 = r[1];
    if constexpr (AFF) {
      const int k = chunk * 32 + kk * 8;
      if (k < p.Cs) {
        v0 = bn_relu_affine(v0, *reinterpret_cast<const f32x4*>(coefL + k),
                            *reinterpret_cast<const f32x4*>(coefL + p.Cs + k),
                            *reinterpret_cast<const f32x4*>(coefL + 2 * p.Cs + k));
      }
      if (k + 4 < p.Cs) {
        v1 = bn_relu_affine(v1, *reinterpret_cast<const f32x4*>(coefL + k + 4),
                            *reinterpret_cast<const f32x4*>(coefL + p.Cs + k + 4),
                            *reinterpret_cast<const f32x4*>(coefL + 2 * p.Cs + k + 4));
      } else {
        v1 = f32x4{0.f, 0.f, 0.f, 0.f};      // (channels past K: the weights there are zero anyway)
      }
      if (k >= p.Cs) v0 = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int row0 = chunk * 32 + kk * 8;
    float b[2][TN];
    bfrag(row0, b[0]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float a = j < 4 ? v0[j & 3] : v1[j & 3];
#pragma unroll
      for (int n = 0; n < TN; ++n) {
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[j & 1][n], acc[n], 0, 0, 0);
        // the next k-group's fragments go out behind the first MFMA of this one and have the other
        // TN - 1 MFMAs (>= 96 cycles, 480 at BNW 256) to arrive; pinned, or the scheduler sinks the
        // reads next to their uses and every group of four MFMAs waits for LDS
        if (n == 0 && j < 7) bfrag(row0 + j + 1, b[(j + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  __syncthreads();                 // the resident weights (and the coefficient image) are published

  // ---- the stream: chunk i lives in ring slot i % RING; slot (i + RING - 1) % RING is refilled ----
  const int total = ntiles * nchunk;
  int c_strip = 0, c_chunk = 0;
  for (int i = 0; i < total; i += RING) {
#pragma unroll
    for (int u = 0; u < RING; ++u) {
      if (i + u < total) {
        gload(ar[(u + RING - 1) % RING]);
        compute(ar[u], c_chunk);
        if (++c_chunk == nchunk) { c_chunk = 0; epilogue(c_strip); ++c_strip; }
      }
    }
  }

  // ---- one partial record per workgroup: fixed-order sum over the lane groups and the waves ----
  if (want_stats || want_bnb) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = s1[g][e], b = s2[g][e];
        a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
        a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
        s1[g][e] = a; s2[g][e] = b;
      }
      if (kk == 0) {
        *reinterpret_cast<f32x4*>(red + (wave * 2 + 0) * BNW + 64 * g + 4 * li) = s1[g];
        *reinterpret_cast<f32x4*>(red + (wave * 2 + 1) * BNW + 64 * g + 4 * li) = s2[g];
      }
    }
    __syncthreads();
    if (t < BNW / 4) {
      const int col = n0 + 4 * t;
      if (col < p.Nn) {
        f32x4 a{0.f, 0.f, 0.f, 0.f}, b = a;
        for (int w8 = 0; w8 < kStreamWaves; ++w8) {
          a += *reinterpret_cast<const f32x4*>(red + (w8 * 2 + 0) * BNW + 4 * t);
          b += *reinterpret_cast<const f32x4*>(red + (w8 * 2 + 1) * BNW + 4 * t);
        }
        const long C4 = p.Nn >> 2, np = row_groups;
        if (want_stats) {
          f32x4* part4 = reinterpret_cast<f32x4*>(p.tile_stats);
          part4[(0 * C4 + (col >> 2)) * np + rg] = a;
          part4[(1 * C4 + (col >> 2)) * np + rg] = b;
          part4[(2 * C4 + (col >> 2)) * np + rg] = *reinterpret_cast<const f32x4*>(shiftL + 4 * t);
        } else {
          f32x4* part4 = reinterpret_cast<f32x4*>(p.bw_part);
          part4[(0 * C4 + (col >> 2)) * np + rg] = a;
          part4[(1 * C4 + (col >> 2)) * np + rg] = b;
        }
      }
    }
  }
}

template <bool BTRANS>
static void launch_stream(const StreamPlan& sp, const IgemmArgs& a, hipStream_t st) {
  const dim3 grid(sp.row_groups * sp.ncb), block(kStreamThreads);
  Plan pl{kStreamBM, sp.bnw, 1, (int)ceil_div(a.Cs, BK), (int)ceil_div(a.Cs, BK), sp.row_groups, sp.ncb};
  note_launch(BTRANS ? GS_OP_DGRAD : GS_OP_FORWARD, GS_KLOOP_STREAM, pl, a.a_coeffs != nullptr,
              a.bw_mode, 2.0 * a.M * (double)a.Nn * a.Cs);
#define GS_STREAM_LAUNCH(BNW_)                                                                       \
  if (sp.bnw == BNW_) {                                                                              \
    if (!BTRANS && a.a_coeffs)                                                                       \
      hipLaunchKernelGGL((conv1x1_stream_kernel<BNW_, BTRANS, !BTRANS>), grid, block, 0, st, a,      \
                         sp.row_groups, sp.tiles_per_wg, sp.ncb);                                    \
    else                                                                                             \
      hipLaunchKernelGGL((conv1x1_stream_kernel<BNW_, BTRANS, false>), grid, block, 0, st, a,        \
                         sp.row_groups, sp.tiles_per_wg, sp.ncb);                                    \
    return;                                                                                          \
  }
  GS_STREAM_LAUNCH(256) GS_STREAM_LAUNCH(128) GS_STREAM_LAUNCH(64)
#undef GS_STREAM_LAUNCH
}

}  // namespace gs
