// Dynamic BatchNorm (+residual add, +ReLU) for NHWC fp32 activations on gfx950 — HBM-bound.
//
// Replaces F.batch_norm over the leading C-slice of the max-size DynBN / DynSyncBN parameters
// (call sites gaiaseg/models/backbones/dynamic_resnet.py:267-300,411,
// gaiaseg/models/utils/dynamic_res_layer.py:92 and the norm1-3 of gaiavision DynamicBottleneck),
// the following ReLU(inplace) and the bottleneck's `out += identity; relu`.
//
// All kernels use one thread->(row, channel-quad) map: a 256-thread block covers `rpi` whole
// rows of C/4 float4 per iteration, so a thread keeps its channel quad for the whole loop
// (per-channel accumulators live in registers) and every wave reads contiguous 16 B/lane.
// Reductions are two-level with a fixed summation order (block partials -> one thread per
// channel), hence bit-reproducible run to run.
#include <algorithm>
#include "common.h"
#include "fused_internal.h"

namespace gs {

struct ColMap {
  int cq;        // channel quad handled by this thread (float4 index within a row)
  int rr;        // row offset within an iteration
  int rpi;       // rows per iteration
  bool active;
};

// C4 = C/4.  When C4 <= 256 one block covers all channels (blockIdx.y == 0) and several rows
// per iteration; otherwise blockIdx.y walks 256-quad column blocks and rpi == 1.
__device__ __forceinline__ ColMap col_map(int C4) {
  ColMap m;
  const int t = threadIdx.x;
  if (C4 <= 256) {
    m.rpi = 256 / C4;
    m.rr = t / C4;
    m.cq = t - m.rr * C4;
    m.active = m.rr < m.rpi;
  } else {
    m.rpi = 1;
    m.rr = 0;
    m.cq = blockIdx.y * 256 + t;
    m.active = m.cq < C4;
  }
  return m;
}

// Block reduction of two float4 accumulators over the rr dimension; result valid for rr == 0.
__device__ __forceinline__ void block_reduce_rows(f32x4& a, f32x4& b, const ColMap& m, int C4,
                                                  f32x4* sh /* [2*256] */) {
  if (m.rpi == 1) return;
  const int t = threadIdx.x;
  sh[t] = a;
  sh[256 + t] = b;
  __syncthreads();
  if (m.active && m.rr == 0) {
    for (int r = 1; r < m.rpi; ++r) {
      a += sh[r * C4 + m.cq];
      b += sh[256 + r * C4 + m.cq];
    }
  }
}

// ---------------- forward statistics ----------------
// part[blockIdx.x][2][C] : shifted sums over this block's row range.
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ x,
                                                               long rows, int C, int ldx,
                                                               long rows_per_block,
                                                               float* __restrict__ part) {
  __shared__ f32x4 sh[512];
  const int C4 = C >> 2;
  const ColMap m = col_map(C4);
  f32x4 s1{0.f, 0.f, 0.f, 0.f}, s2{0.f, 0.f, 0.f, 0.f};
  if (m.active) {
    const f32x4 shift = *reinterpret_cast<const f32x4*>(x + m.cq * 4);  // row 0
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(r0 + rows_per_block, rows);
    for (long r = r0 + m.rr; r < r1; r += m.rpi) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + m.cq * 4) - shift;
      s1 += v;
      s2 += v * v;
    }
  }
  block_reduce_rows(s1, s2, m, C4, sh);
  if (m.active && m.rr == 0) {
    // quad-major partial layout (element (p, Q) at part4[Q * nparts + p]): the summing kernels
    // then read every quad's partials as one contiguous run
    f32x4* part4 = reinterpret_cast<f32x4*>(part);
    part4[(long)m.cq * gridDim.x + blockIdx.x] = s1;
    part4[((long)C4 + m.cq) * gridDim.x + blockIdx.x] = s2;
  }
}

// Fixed-order block sum of NV column quads over the partial rows: thread t takes rows t, t+256, ...
// (four independent 16-B loads in flight per quad), accumulates in double, the wave combines with
// an xor butterfly and the four wave totals are added in wave order, so the result is
// bit-reproducible.  After the call every thread holds the totals.  (History: one thread per
// column walking 1024 partials serially cost 150 us per call; one WAVE per quad still chained
// 16 dependent-latency iterations, 12 us; this form is launch-latency bound.)
template <int NV>
__device__ __forceinline__ void block_sum_quads(const float* __restrict__ part, int nparts,
                                                const int (&col)[NV] /* quad indices */,
                                                double (&a)[4 * NV], double* sh /* [4][4*NV] */) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
#pragma unroll
  for (int e = 0; e < 4 * NV; ++e) a[e] = 0.0;
  for (int base = 0; base < nparts; base += 1024) {
    f32x4 v[4][NV];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = base + u * 256 + t;
#pragma unroll
      for (int j = 0; j < NV; ++j)
        v[u][j] = p < nparts ? reinterpret_cast<const f32x4*>(part)[(long)col[j] * nparts + p]
                             : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) a[4 * j + e] += v[u][j][e];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
#pragma unroll
    for (int e = 0; e < 4 * NV; ++e) a[e] += __shfl_xor(a[e], off, 64);
  if (lane == 0)
#pragma unroll
    for (int e = 0; e < 4 * NV; ++e) sh[wave * 4 * NV + e] = a[e];
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 4 * NV; ++e)
    a[e] = ((sh[e] + sh[4 * NV + e]) + sh[8 * NV + e]) + sh[12 * NV + e];
}

// out[0..width) = sum over partial rows of part[p][width]; one block per column quad.
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ part,
                                                           int nparts, int width,
                                                           float* __restrict__ out,
                                                           const float* __restrict__ copy_src,
                                                           int copy_n) {
  __shared__ double sh[16];
  // optional tail copy out[width + i] = copy_src[i] (the BN conditioning shift = row 0 of x)
  if (copy_src) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < copy_n) out[width + i] = copy_src[i];
  }
  const int q = blockIdx.x;
  if (q * 4 >= width) return;   // block-uniform
  const int col[1] = {q};
  double a[4];
  block_sum_quads<1>(part, nparts, col, a, sh);
  if (threadIdx.x == 0)
    *reinterpret_cast<f32x4*>(out + q * 4) = f32x4{(float)a[0], (float)a[1], (float)a[2], (float)a[3]};
}

static inline void launch_sum_partials(const float* part, int nparts, int width, float* out,
                                       const float* copy_src, int copy_n, hipStream_t st) {
  int grid = width / 4;
  if (copy_src) grid = std::max(grid, (copy_n + 255) / 256);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(grid), dim3(256), 0, st, part, nparts, width, out,
                     copy_src, copy_n);
}

// Local-BN fast path: fixed-order sum of the partials + finalize in one launch (one block per
// channel quad; threads 0..3 finish one channel each).  Saves one launch per BN layer per step
// versus sum_partials + bn_finalize.
__global__ __launch_bounds__(256) void bn_sum_finalize_kernel(
    const float* __restrict__ part, int nparts, int C, const float* __restrict__ x, double count,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
    float* running_mean, float* running_var, float* __restrict__ coeffs) {
  __shared__ double sh[32];
  const int q = blockIdx.x;
  const int col[2] = {q, C / 4 + q};
  const int lane = threadIdx.x;
  // (fetched beside the partials, not behind the reduction: see bn_tile_finalize_kernel)
  const int pc = q * 4 + (lane & 3);
  float x_pre = 0.f, g_pre = 1.f, be_pre = 0.f, rm_pre = 0.f, rv_pre = 0.f;
  if (lane < 4) {
    x_pre = x[pc];
    if (gamma) g_pre = gamma[pc];
    if (beta) be_pre = beta[pc];
    if (running_mean) rm_pre = running_mean[pc];
    if (running_var) rv_pre = running_var[pc];
  }
  double a[8];
  block_sum_quads<2>(part, nparts, col, a, sh);
  if (lane < 4) {
    const int c = q * 4 + lane;
    double s1 = a[0], s2 = a[4];
    if (lane == 1) { s1 = a[1]; s2 = a[5]; }
    if (lane == 2) { s1 = a[2]; s2 = a[6]; }
    if (lane == 3) { s1 = a[3]; s2 = a[7]; }
    // the partial sums were rounded to float when stored, exactly as in the two-launch path
    const double d1 = (double)(float)s1 / count;
    const double mean = (double)x_pre + d1;
    double var = (double)(float)s2 / count - d1 * d1;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    coeffs[c] = (float)((double)g_pre * invstd);
    coeffs[C + c] = be_pre;
    coeffs[2 * C + c] = (float)mean;
    coeffs[3 * C + c] = (float)invstd;
    if (running_mean) running_mean[c] = (1.f - momentum) * rm_pre + momentum * (float)mean;
    if (running_var) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      running_var[c] = (1.f - momentum) * rv_pre + momentum * (float)unbiased;
    }
  }
}

// coeffs: [0,C) scale = gamma*invstd ; [C,2C) beta ; [2C,3C) mean ; [3C,4C) invstd
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ sums,
                                                          double count, int C,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps,
                                                          float momentum, float* running_mean,
                                                          float* running_var,
                                                          float* __restrict__ coeffs) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double d1 = (double)sums[c] / count;
  const double mean = (double)sums[2 * C + c] + d1;
  double var = (double)sums[C + c] / count - d1 * d1;
  if (var < 0.0) var = 0.0;
  const double invstd = 1.0 / sqrt(var + (double)eps);
  const float g = gamma ? gamma[c] : 1.f;
  coeffs[c] = (float)((double)g * invstd);
  coeffs[C + c] = beta ? beta[c] : 0.f;
  coeffs[2 * C + c] = (float)mean;
  coeffs[3 * C + c] = (float)invstd;
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
  if (running_var) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

__global__ __launch_bounds__(256) void bn_eval_coeffs_kernel(const float* __restrict__ rm,
                                                             const float* __restrict__ rv, int C,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             float eps, float* __restrict__ coeffs) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double invstd = 1.0 / sqrt((double)rv[c] + (double)eps);
  const float g = gamma ? gamma[c] : 1.f;
  coeffs[c] = (float)((double)g * invstd);
  coeffs[C + c] = beta ? beta[c] : 0.f;
  coeffs[2 * C + c] = rm[c];
  coeffs[3 * C + c] = (float)invstd;
}

// y = act((x - mean) * scale + beta (+ residual))
// MASKOUT (with RELU): also mask[r][cq] = one byte per channel quad, bit e set <=> y[r][4cq+e] > 0 --
// what the fused BatchNorm-backward epilogue of the consumer's data gradient needs of y (mode 3 of
// gs_bn_bwd_fuse): 1/16 of the bytes of reading y again.
// RESAFF (with RES): res is the raw output of the projection shortcut's conv and rcoeffs that conv's
// BatchNorm coefficients; the addend is (res - rmean) * rscale + rbeta — the same expression, in the
// same order, as a separate apply pass of the shortcut would have stored.
template <bool RES, bool RELU, bool MASKOUT = false, bool RESAFF = false>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* x, long rows, int C, int ldx,
                                                       const float* __restrict__ coeffs,
                                                       const float* res, int ld_res, float* y,
                                                       int ldy, unsigned char* __restrict__ mask,
                                                       const float* __restrict__ rcoeffs = nullptr) {  // x / res / y may alias
  const int C4 = C >> 2;
  const ColMap m = col_map(C4);
  if (!m.active) return;
  const f32x4 scale = *reinterpret_cast<const f32x4*>(coeffs + m.cq * 4);
  const f32x4 beta = *reinterpret_cast<const f32x4*>(coeffs + C + m.cq * 4);
  const f32x4 mean = *reinterpret_cast<const f32x4*>(coeffs + 2 * C + m.cq * 4);
  f32x4 rscale{1.f, 1.f, 1.f, 1.f}, rbeta{0.f, 0.f, 0.f, 0.f}, rmean{0.f, 0.f, 0.f, 0.f};
  if (RESAFF) {
    rscale = *reinterpret_cast<const f32x4*>(rcoeffs + m.cq * 4);
    rbeta = *reinterpret_cast<const f32x4*>(rcoeffs + C + m.cq * 4);
    rmean = *reinterpret_cast<const f32x4*>(rcoeffs + 2 * C + m.cq * 4);
  }
  const long step = (long)gridDim.x * m.rpi;
  for (long r = (long)blockIdx.x * m.rpi + m.rr; r < rows; r += step) {
    f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + m.cq * 4);
    v = (v - mean) * scale + beta;
    if (RES) {
      f32x4 a = *reinterpret_cast<const f32x4*>(res + r * ld_res + m.cq * 4);
      if (RESAFF) a = (a - rmean) * rscale + rbeta;
      v += a;
    }
    if (RELU) {
      if (MASKOUT)
        mask[r * C4 + m.cq] = (unsigned char)((v[0] > 0.f ? 1 : 0) | (v[1] > 0.f ? 2 : 0) |
                                              (v[2] > 0.f ? 4 : 0) | (v[3] > 0.f ? 8 : 0));
      v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f);
      v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
    }
    *reinterpret_cast<f32x4*>(y + r * ldy + m.cq * 4) = v;
  }
}

// ---------------- backward ----------------
template <int MASK>
__device__ __forceinline__ f32x4 masked_grad(f32x4 dy, f32x4 x, f32x4 act, f32x4 mean, f32x4 scale,
                                             f32x4 beta) {
  if (MASK == 1) {
    const f32x4 yv = (x - mean) * scale + beta;
#pragma unroll
    for (int e = 0; e < 4; ++e) dy[e] = yv[e] > 0.f ? dy[e] : 0.f;
  } else if (MASK == 2) {
#pragma unroll
    for (int e = 0; e < 4; ++e) dy[e] = act[e] > 0.f ? dy[e] : 0.f;
  }
  return dy;
}

// part[blockIdx.x][2][C] = { sum g, sum g*xhat } ; optionally writes g (masked dy)
template <int MASK>
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(
    const float* __restrict__ dy, int ld_dy, const float* __restrict__ x, int ldx,
    const float* __restrict__ act, int ld_act, long rows, int C, const float* __restrict__ coeffs,
    long rows_per_block, float* g_out, int ld_g, float* __restrict__ part) {
  __shared__ f32x4 sh[512];
  const int C4 = C >> 2;
  const ColMap m = col_map(C4);
  f32x4 s1{0.f, 0.f, 0.f, 0.f}, s2{0.f, 0.f, 0.f, 0.f};
  if (m.active) {
    const f32x4 scale = *reinterpret_cast<const f32x4*>(coeffs + m.cq * 4);
    const f32x4 beta = *reinterpret_cast<const f32x4*>(coeffs + C + m.cq * 4);
    const f32x4 mean = *reinterpret_cast<const f32x4*>(coeffs + 2 * C + m.cq * 4);
    const f32x4 invstd = *reinterpret_cast<const f32x4*>(coeffs + 3 * C + m.cq * 4);
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(r0 + rows_per_block, rows);
    for (long r = r0 + m.rr; r < r1; r += m.rpi) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + r * ldx + m.cq * 4);
      f32x4 g = *reinterpret_cast<const f32x4*>(dy + r * ld_dy + m.cq * 4);
      f32x4 av{0.f, 0.f, 0.f, 0.f};
      if (MASK == 2) av = *reinterpret_cast<const f32x4*>(act + r * ld_act + m.cq * 4);
      g = masked_grad<MASK>(g, xv, av, mean, scale, beta);
      if (g_out) *reinterpret_cast<f32x4*>(g_out + r * ld_g + m.cq * 4) = g;
      s1 += g;
      s2 += g * ((xv - mean) * invstd);
    }
  }
  block_reduce_rows(s1, s2, m, C4, sh);
  if (m.active && m.rr == 0) {
    // quad-major partial layout (element (p, Q) at part4[Q * nparts + p]): the summing kernels
    // then read every quad's partials as one contiguous run
    f32x4* part4 = reinterpret_cast<f32x4*>(part);
    part4[(long)m.cq * gridDim.x + blockIdx.x] = s1;
    part4[((long)C4 + m.cq) * gridDim.x + blockIdx.x] = s2;
  }
}

// dx = scale * (g - sum_g/n - xhat*sum_gx/n)  or  dx = scale*g (eval-mode statistics)
template <int MASK, bool BATCH>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const float* __restrict__ dy, int ld_dy, const float* __restrict__ x, int ldx,
    const float* __restrict__ act, int ld_act, long rows, int C, const float* __restrict__ coeffs,
    const float* __restrict__ sums, float inv_count, float* __restrict__ dx, int ld_dx,
    float* dgamma, float* dbeta) {
  const int C4 = C >> 2;
  const ColMap m = col_map(C4);
  if (!m.active) return;
  const f32x4 scale = *reinterpret_cast<const f32x4*>(coeffs + m.cq * 4);
  const f32x4 beta = *reinterpret_cast<const f32x4*>(coeffs + C + m.cq * 4);
  const f32x4 mean = *reinterpret_cast<const f32x4*>(coeffs + 2 * C + m.cq * 4);
  const f32x4 invstd = *reinterpret_cast<const f32x4*>(coeffs + 3 * C + m.cq * 4);
  const f32x4 sg = *reinterpret_cast<const f32x4*>(sums + m.cq * 4);
  const f32x4 sgx = *reinterpret_cast<const f32x4*>(sums + C + m.cq * 4);
  if (blockIdx.x == 0 && m.rr == 0) {
    if (dgamma) *reinterpret_cast<f32x4*>(dgamma + m.cq * 4) = sgx;
    if (dbeta) *reinterpret_cast<f32x4*>(dbeta + m.cq * 4) = sg;
  }
  const f32x4 c1 = sg * inv_count, c2 = sgx * inv_count;
  const long step = (long)gridDim.x * m.rpi;
  for (long r = (long)blockIdx.x * m.rpi + m.rr; r < rows; r += step) {
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + r * ldx + m.cq * 4);
    f32x4 g = *reinterpret_cast<const f32x4*>(dy + r * ld_dy + m.cq * 4);
    f32x4 av{0.f, 0.f, 0.f, 0.f};
    if (MASK == 2) av = *reinterpret_cast<const f32x4*>(act + r * ld_act + m.cq * 4);
    g = masked_grad<MASK>(g, xv, av, mean, scale, beta);
    f32x4 o;
    if (BATCH) o = scale * (g - c1 - ((xv - mean) * invstd) * c2);
    else o = scale * g;
    *reinterpret_cast<f32x4*>(dx + r * ld_dx + m.cq * 4) = o;
  }
}

// plain per-column sums (conv_seg bias gradient)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, long rows,
                                                             int C, int ldx, long rows_per_block,
                                                             float* __restrict__ part) {
  __shared__ f32x4 sh[512];
  const int C4 = C >> 2;
  const ColMap m = col_map(C4);
  f32x4 s1{0.f, 0.f, 0.f, 0.f}, s2{0.f, 0.f, 0.f, 0.f};
  if (m.active) {
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(r0 + rows_per_block, rows);
    for (long r = r0 + m.rr; r < r1; r += m.rpi)
      s1 += *reinterpret_cast<const f32x4*>(x + r * ldx + m.cq * 4);
  }
  block_reduce_rows(s1, s2, m, C4, sh);
  if (m.active && m.rr == 0)
    reinterpret_cast<f32x4*>(part)[(long)m.cq * gridDim.x + blockIdx.x] = s1;
}


// ---------------- statistics fused with the producing convolution ----------------
// (a) conv without split-K: the conv epilogue wrote per-tile partials {s1, s2, shift} (quad-major,
//     igemm_core.h rows_epilogue).  Merge the tiles exactly (Chan et al.): tile mean
//     m_t = shift_t + s1_t/n_t, tile M2_t = s2_t - s1_t^2/n_t; mean = sum n_t m_t / N,
//     M2 = sum [M2_t + n_t (m_t - mean)^2] -- evaluated in one pass around a common reference (see
//     the kernel).  Fixed-order block sums in double.
__device__ __forceinline__ void block_sum_d4(double (&a)[4], double* sh /* [16] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] += __shfl_xor(a[e], off, 64);
  __syncthreads();   // sh may still be read from a previous call
  if (lane == 0)
#pragma unroll
    for (int e = 0; e < 4; ++e) sh[wave * 4 + e] = a[e];
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 4; ++e) a[e] = ((sh[e] + sh[4 + e]) + sh[8 + e]) + sh[12 + e];
}

__global__ __launch_bounds__(256) void bn_tile_finalize_kernel(
    const float* __restrict__ part, int np, int bm, long M, int C, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, float momentum, float* running_mean,
    float* running_var, float* __restrict__ coeffs, int n_last, double inv_full, double inv_last,
    double inv_M) {
  __shared__ double sh[16];
  const int q = blockIdx.x, C4 = C >> 2, t = threadIdx.x;
  const f32x4* p4 = reinterpret_cast<const f32x4*>(part);
  const f32x4* s1p = p4 + (0L * C4 + q) * np;
  const f32x4* s2p = p4 + (1L * C4 + q) * np;
  const f32x4* shp = p4 + (2L * C4 + q) * np;
  // every tile has bm rows except possibly the last one (n_last); the reciprocals come from the
  // host (no fp64 division per thread).  NOTE: no private array may be indexed by a run-time value
  // in this kernel: the compiler would move it to LDS, address it with the flat work-item id, and
  // fetch the workgroup size from the dispatch packet in HOST memory in every wave (measured:
  // 30 us instead of 5 for 512 workgroups).
  // ONE pass over the partials (r03; two dependent passes -- means first, then M2 around the means --
  // cost two L2 / fabric round trips in a kernel that is nothing but latency): every tile's sums are
  // re-centred on a common reference r = tile 0's shift (a sample of the channel, so |r - mean| is a
  // few standard deviations at most):  sum (v - r) = s1 + n d,  sum (v - r)^2 = s2 + 2 d s1 + n d^2
  // with d = shift_t - r, accumulated in double; mean = r + S1 / M, M2 = S2 - S1^2 / M.
  const f32x4 ref4 = shp[0];
  // lanes 0..3 finish one channel each; what they need besides the sums is fetched NOW, beside the
  // partials, not behind the reductions (one more dependent L2 round trip in a latency-only kernel)
  const int fc = q * 4 + (t & 3);
  float g_pre = 1.f, be_pre = 0.f, rm_pre = 0.f, rv_pre = 0.f;
  if (t < 4) {
    if (gamma) g_pre = gamma[fc];
    if (beta) be_pre = beta[fc];
    if (running_mean) rm_pre = running_mean[fc];
    if (running_var) rv_pre = running_var[fc];
  }
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;
  for (int p = t; p < np; p += 256) {
    const double n = (double)(p == np - 1 ? n_last : bm);
    const f32x4 s1 = s1p[p], s2 = s2p[p], shv = shp[p];
#define GS_ACC(E, A, B)                                              \
    {                                                                \
      const double d = (double)shv[E] - (double)ref4[E];             \
      A += (double)s1[E] + n * d;                                    \
      B += (double)s2[E] + d * (2.0 * (double)s1[E] + n * d);        \
    }
    GS_ACC(0, a0, b0) GS_ACC(1, a1, b1) GS_ACC(2, a2, b2) GS_ACC(3, a3, b3)
#undef GS_ACC
  }
  {
    double a[4] = {a0, a1, a2, a3};
    block_sum_d4(a, sh);
    double b[4] = {b0, b1, b2, b3};
    block_sum_d4(b, sh);
    // b <- M2, a <- mean
    b0 = b[0] - a[0] * a[0] * inv_M; b1 = b[1] - a[1] * a[1] * inv_M;
    b2 = b[2] - a[2] * a[2] * inv_M; b3 = b[3] - a[3] * a[3] * inv_M;
    a0 = (double)ref4[0] + a[0] * inv_M; a1 = (double)ref4[1] + a[1] * inv_M;
    a2 = (double)ref4[2] + a[2] * inv_M; a3 = (double)ref4[3] + a[3] * inv_M;
  }
  (void)inv_full; (void)inv_last;
  if (t < 4) {
    // (selects, not a run-time index into a private array: see the NOTE above)
    const double MU = t == 0 ? a0 : t == 1 ? a1 : t == 2 ? a2 : a3;
    const double M2 = t == 0 ? b0 : t == 1 ? b1 : t == 2 ? b2 : b3;
    const int c = fc;
    double var = M2 * inv_M;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    coeffs[c] = (float)((double)g_pre * invstd);
    coeffs[C + c] = be_pre;
    coeffs[2 * C + c] = (float)MU;
    coeffs[3 * C + c] = (float)invstd;
    if (running_mean) running_mean[c] = (1.f - momentum) * rm_pre + momentum * (float)MU;
    if (running_var) {
      const double unbiased = M > 1 ? var * (double)M / ((double)M - 1.0) : var;
      running_var[c] = (1.f - momentum) * rv_pre + momentum * (float)unbiased;
    }
  }
}

// (b) conv with split-K: the fixed-order sum of the partial slabs (igemm_core.h
//     splitk_reduce_kernel, same order: split 0 first), the store of y AND the shifted sums of
//     bn_stats_partial_kernel in one pass, with that kernel's geometry and shift (= y[0, :],
//     recomputed from the slabs): bit-identical to reduce followed by bn_stats.
template <int ROLE>   // ROLE only names the instantiation (see gs_conv_desc::role)
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(
    const float* __restrict__ slab, int splits, long rows, int C, float* __restrict__ y, int ldy,
    long rows_per_block, float* __restrict__ part) {
  __shared__ f32x4 sh[512];
  const int C4 = C >> 2;
  const ColMap m = col_map(C4);
  f32x4 s1{0.f, 0.f, 0.f, 0.f}, s2{0.f, 0.f, 0.f, 0.f};
  if (m.active) {
    const long zs = rows * C;   // slab stride
    f32x4 shift = *reinterpret_cast<const f32x4*>(slab + m.cq * 4);
    for (int z = 1; z < splits; ++z) shift += *reinterpret_cast<const f32x4*>(slab + z * zs + m.cq * 4);
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(r0 + rows_per_block, rows);
    for (long r = r0 + m.rr; r < r1; r += m.rpi) {
      const float* sp = slab + r * C + m.cq * 4;
      f32x4 v = *reinterpret_cast<const f32x4*>(sp);
      for (int z = 1; z < splits; ++z) v += *reinterpret_cast<const f32x4*>(sp + z * zs);
      *reinterpret_cast<f32x4*>(y + r * ldy + m.cq * 4) = v;
      v -= shift;
      s1 += v;
      s2 += v * v;
    }
  }
  block_reduce_rows(s1, s2, m, C4, sh);
  if (m.active && m.rr == 0) {
    f32x4* part4 = reinterpret_cast<f32x4*>(part);
    part4[(long)m.cq * gridDim.x + blockIdx.x] = s1;
    part4[((long)C4 + m.cq) * gridDim.x + blockIdx.x] = s2;
  }
}


// (c) data gradient with split-K whose output is the gradient of z = relu(bn_prev(y) [+ residual]):
//     fixed-order sum of the slabs (+ the gradient already accumulated in dx), the producer's ReLU
//     mask, the store of the masked gradient g and that BatchNorm's backward sums {sum g, sum g xhat}
//     in one pass — the split-K form of the fused dgrad epilogue (igemm_core.h rows_epilogue, bw_*).
template <int MODE>
__global__ __launch_bounds__(256) void splitk_reduce_bnbwd_kernel(
    const float* __restrict__ slab, int splits, long rows, int C, float* dx, int ld_dx, int accumulate,
    const float* __restrict__ yprev, int ldy, const float* __restrict__ act, int ldact,
    const float* __restrict__ coeffs, long rows_per_block, float* __restrict__ part,
    const unsigned char* __restrict__ mask, int ldmask) {
  __shared__ f32x4 sh[512];
  const int C4 = C >> 2;
  const ColMap m = col_map(C4);
  f32x4 s1{0.f, 0.f, 0.f, 0.f}, s2{0.f, 0.f, 0.f, 0.f};
  if (m.active) {
    const f32x4 scale = *reinterpret_cast<const f32x4*>(coeffs + m.cq * 4);
    const f32x4 beta = *reinterpret_cast<const f32x4*>(coeffs + C + m.cq * 4);
    const f32x4 mean = *reinterpret_cast<const f32x4*>(coeffs + 2 * C + m.cq * 4);
    const f32x4 invstd = *reinterpret_cast<const f32x4*>(coeffs + 3 * C + m.cq * 4);
    const long zs = rows * C;   // slab stride
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(r0 + rows_per_block, rows);
    for (long r = r0 + m.rr; r < r1; r += m.rpi) {
      const float* sp = slab + r * C + m.cq * 4;
      f32x4 v = *reinterpret_cast<const f32x4*>(sp);
      for (int z = 1; z < splits; ++z) v += *reinterpret_cast<const f32x4*>(sp + z * zs);
      float* o = dx + r * ld_dx + m.cq * 4;
      if (accumulate) v += *reinterpret_cast<const f32x4*>(o);
      const f32x4 yv = *reinterpret_cast<const f32x4*>(yprev + r * ldy + m.cq * 4);
      f32x4 av{0.f, 0.f, 0.f, 0.f};
      if (MODE == 2) av = *reinterpret_cast<const f32x4*>(act + r * ldact + m.cq * 4);
      if (MODE == 3) {
        const unsigned bits = mask[r * ldmask + m.cq];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ((bits >> e) & 1u) ? v[e] : 0.f;
      } else {
        v = masked_grad<MODE>(v, yv, av, mean, scale, beta);
      }
      *reinterpret_cast<f32x4*>(o) = v;
      s1 += v;
      s2 += v * ((yv - mean) * invstd);
    }
  }
  block_reduce_rows(s1, s2, m, C4, sh);
  if (m.active && m.rr == 0) {
    f32x4* part4 = reinterpret_cast<f32x4*>(part);
    part4[(long)m.cq * gridDim.x + blockIdx.x] = s1;
    part4[((long)C4 + m.cq) * gridDim.x + blockIdx.x] = s2;
  }
}

// ---------------- SyncBN exchange helpers (one launch each instead of ~25 tiny tensor ops) -------
// local[0..C) = mean, local[C..2C) = biased variance, local[2C] = count  (double), from the shifted
// sums {S1, S2, shift} of gs_bn_stats: the payload a rank contributes to the all_gather.
__global__ __launch_bounds__(256) void bn_sync_local_kernel(const float* __restrict__ sums,
                                                            double count, int C,
                                                            double* __restrict__ local) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0) local[2 * C] = count;
  if (c >= C) return;
  const double d1 = (double)sums[c] / count;
  double var = (double)sums[C + c] / count - d1 * d1;
  if (var < 0.0) var = 0.0;
  local[c] = (double)sums[2 * C + c] + d1;
  local[C + c] = var;
}
// gathered[world][2C+1] -> merged sums {0, gvar * total, gmean} (float) for gs_bn_finalize:
// gmean = sum n_r m_r / N, gvar = sum n_r (v_r + (m_r - gmean)^2) / N (Chan et al.), rank order.
__global__ __launch_bounds__(256) void bn_sync_merge_kernel(const double* __restrict__ g, int world,
                                                            int C, float* __restrict__ merged) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int ld = 2 * C + 1;
  double total = 0.0, msum = 0.0;
  for (int r = 0; r < world; ++r) {
    const double n = g[(long)r * ld + 2 * C];
    total += n;
    msum += n * g[(long)r * ld + c];
  }
  const double gmean = msum / total;
  double vsum = 0.0;
  for (int r = 0; r < world; ++r) {
    const double n = g[(long)r * ld + 2 * C], dm = g[(long)r * ld + c] - gmean;
    vsum += n * (g[(long)r * ld + C + c] + dm * dm);
  }
  merged[c] = 0.f;
  merged[C + c] = (float)vsum;          // = gvar * total
  merged[2 * C + c] = (float)gmean;
}

// ---- host helpers ----
struct RedGeom {
  int gx, gy;
  long rows_per_block;
};
static RedGeom red_geom(long rows, int C) {
  const int C4 = C >> 2;
  RedGeom g;
  g.gy = C4 <= 256 ? 1 : (int)ceil_div(C4, 256);
  const int rpi = C4 <= 256 ? 256 / C4 : 1;
  // ~4 iterations per block at least, at most 1024 row blocks (partials stay small)
  static const int it_min = getenv("GS_BNORM_ITERS") ? atoi(getenv("GS_BNORM_ITERS")) : 4;
  static const int gx_cap = getenv("GS_BNORM_PARTS") ? atoi(getenv("GS_BNORM_PARTS")) : 1024;
  long gx = ceil_div(rows, (long)rpi * it_min);
  const long cap = std::max<long>(1, gx_cap / g.gy);
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  g.rows_per_block = ceil_div(rows, gx);
  // keep whole iterations inside a block so the rr lanes stay balanced
  g.rows_per_block = ceil_div(g.rows_per_block, rpi) * rpi;
  g.gx = (int)ceil_div(rows, g.rows_per_block);
  return g;
}
static int apply_grid(long rows, int C) {
  const int C4 = C >> 2;
  const int rpi = C4 <= 256 ? 256 / C4 : 1;
  long gx = ceil_div(rows, (long)rpi * 2);
  const long cap = (long)num_cu() * 8;
  if (gx > cap) gx = cap;
  return (int)std::max<long>(gx, 1);
}
static int check_rows(const void* p, long rows, int C, int ld) {
  if (!p) return GS_E_NULL;
  if (rows <= 0 || C <= 0) return GS_E_BADARG;
  if ((C & 3) || (ld & 3) || ld < C || !aligned16(p)) return GS_E_ALIGN;
  return GS_OK;
}

}  // namespace gs

using namespace gs;

// ---- internal entry points used by fused_layers.hip ----
namespace gs {
size_t bn_fused_reduce_bytes(long rows, int C) {
  const RedGeom g = red_geom(rows, C);
  return (size_t)g.gx * 2 * C * sizeof(float);
}
// y = sum of `splits` slabs [rows][C]; coeffs from the batch statistics of y (rank-local BN)
int bn_reduce_stats_finalize(const float* slab, int splits, long rows, int C, float* y, int ldy,
                             const float* gamma, const float* beta, float eps, float momentum,
                             float* running_mean, float* running_var, float* coeffs, float* part,
                             size_t part_bytes, hipStream_t st, int role, bool timed, double flops) {
  const RedGeom g = red_geom(rows, C);
  if ((size_t)g.gx * 2 * C * sizeof(float) > part_bytes) return GS_E_WORKSPACE;
  if (role == 1)
    hipLaunchKernelGGL(splitk_reduce_stats_kernel<1>, dim3(g.gx, g.gy), dim3(256), 0, st, slab,
                       splits, rows, C, y, ldy, g.rows_per_block, part);
  else
    hipLaunchKernelGGL(splitk_reduce_stats_kernel<0>, dim3(g.gx, g.gy), dim3(256), 0, st, slab,
                       splits, rows, C, y, ldy, g.rows_per_block, part);
  if (timed) k3_prof_end(st, flops);   // the K3 interval covers the conv and its slab reduction
  hipLaunchKernelGGL(bn_sum_finalize_kernel, dim3(C / 4), dim3(256), 0, st, part, g.gx, C, y,
                     (double)rows, gamma, beta, eps, momentum, running_mean, running_var, coeffs);
  return launch_status();
}
size_t bn_reduce_bnbwd_bytes(long rows, int C) {
  const RedGeom g = red_geom(rows, C);
  return (size_t)g.gx * 2 * C * sizeof(float);
}
// dx = mask(sum of slabs [+ dx]); sums = {sum g, sum g * xhat_prev}  (see splitk_reduce_bnbwd_kernel)
int bn_reduce_bnbwd(const float* slab, int splits, long rows, int C, float* dx, int ld_dx,
                    int accumulate, const gs_bn_bwd_fuse* bw, float* part, size_t part_bytes,
                    hipStream_t st) {
  const RedGeom g = red_geom(rows, C);
  if ((size_t)g.gx * 2 * C * sizeof(float) > part_bytes) return GS_E_WORKSPACE;
  if (bw->mode == 3)
    hipLaunchKernelGGL(splitk_reduce_bnbwd_kernel<3>, dim3(g.gx, g.gy), dim3(256), 0, st, slab, splits,
                       rows, C, dx, ld_dx, accumulate, bw->y, bw->ldy, (const float*)nullptr, 0,
                       bw->coeffs, g.rows_per_block, part, bw->mask, bw->ldmask);
  else if (bw->mode == 2)
    hipLaunchKernelGGL(splitk_reduce_bnbwd_kernel<2>, dim3(g.gx, g.gy), dim3(256), 0, st, slab, splits,
                       rows, C, dx, ld_dx, accumulate, bw->y, bw->ldy, bw->act, bw->ldact, bw->coeffs,
                       g.rows_per_block, part, (const unsigned char*)nullptr, 0);
  else
    hipLaunchKernelGGL(splitk_reduce_bnbwd_kernel<1>, dim3(g.gx, g.gy), dim3(256), 0, st, slab, splits,
                       rows, C, dx, ld_dx, accumulate, bw->y, bw->ldy, bw->act, bw->ldact, bw->coeffs,
                       g.rows_per_block, part, (const unsigned char*)nullptr, 0);
  launch_sum_partials(part, g.gx, 2 * C, bw->sums, nullptr, 0, st);
  return launch_status();
}
int bn_sum_partials(const float* part, int nparts, int width, float* out, hipStream_t st) {
  launch_sum_partials(part, nparts, width, out, nullptr, 0, st);
  return launch_status();
}
// coeffs from the per-tile partials the conv epilogue wrote
int bn_tile_finalize(const float* part, int np, int bm, long rows, int C, const float* gamma,
                     const float* beta, float eps, float momentum, float* running_mean,
                     float* running_var, float* coeffs, hipStream_t st) {
  const int n_last = (int)(rows - (long)(np - 1) * bm);
  hipLaunchKernelGGL(bn_tile_finalize_kernel, dim3(C / 4), dim3(256), 0, st, part, np, bm, rows, C,
                     gamma, beta, eps, momentum, running_mean, running_var, coeffs, n_last,
                     1.0 / (double)bm, 1.0 / (double)n_last, 1.0 / (double)rows);
  return launch_status();
}
}  // namespace gs

extern "C" size_t gs_bn_stats_workspace_bytes(int64_t rows, int32_t C) {
  if (rows <= 0 || C <= 0 || (C & 3)) return 0;
  const RedGeom g = red_geom(rows, C);
  return (size_t)g.gx * 2 * C * sizeof(float);
}
extern "C" size_t gs_bn_bwd_workspace_bytes(int64_t rows, int32_t C) {
  return gs_bn_stats_workspace_bytes(rows, C);
}
extern "C" size_t gs_colsum_workspace_bytes(int64_t rows, int32_t C) {
  return gs_bn_stats_workspace_bytes(rows, C);
}

extern "C" int gs_bn_stats(const float* x, int64_t rows, int32_t C, int32_t ldx, float* sums,
                           void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_rows(x, rows, C, ldx);
  if (rc) return rc;
  if (!sums || !workspace) return GS_E_NULL;
  const RedGeom g = red_geom(rows, C);
  if ((size_t)g.gx * 2 * C * sizeof(float) > workspace_bytes) return GS_E_WORKSPACE;
  if (!aligned16(workspace)) return GS_E_ALIGN;
  hipStream_t st = as_stream(stream);
  float* part = static_cast<float*>(workspace);
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(g.gx, g.gy), dim3(256), 0, st, x, (long)rows, C,
                     ldx, g.rows_per_block, part);
  launch_sum_partials(part, g.gx, 2 * C, sums, x, C, st);  // sums = {S1, S2, shift = x[0, :]}
  return launch_status();
}

extern "C" int gs_bn_stats_finalize(const float* x, int64_t rows, int32_t C, int32_t ldx,
                                    const float* gamma, const float* beta, float eps,
                                    float momentum, float* running_mean, float* running_var,
                                    float* coeffs, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  int rc = check_rows(x, rows, C, ldx);
  if (rc) return rc;
  if (!coeffs || !workspace) return GS_E_NULL;
  const RedGeom g = red_geom(rows, C);
  if ((size_t)g.gx * 2 * C * sizeof(float) > workspace_bytes) return GS_E_WORKSPACE;
  if (!aligned16(workspace)) return GS_E_ALIGN;
  hipStream_t st = as_stream(stream);
  float* part = static_cast<float*>(workspace);
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(g.gx, g.gy), dim3(256), 0, st, x, (long)rows, C,
                     ldx, g.rows_per_block, part);
  hipLaunchKernelGGL(bn_sum_finalize_kernel, dim3(C / 4), dim3(256), 0, st, part, g.gx, C,
                     x, (double)rows, gamma, beta, eps, momentum, running_mean, running_var, coeffs);
  return launch_status();
}

extern "C" int gs_bn_finalize(const float* sums, double count, int32_t C, const float* gamma,
                              const float* beta, float eps, float momentum, float* running_mean,
                              float* running_var, float* coeffs, void* stream) {
  if (!sums || !coeffs) return GS_E_NULL;
  if (C <= 0 || count <= 0.0) return GS_E_BADARG;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, as_stream(stream),
                     sums, count, C, gamma, beta, eps, momentum, running_mean, running_var, coeffs);
  return launch_status();
}

extern "C" int gs_bn_eval_coeffs(const float* running_mean, const float* running_var, int32_t C,
                                 const float* gamma, const float* beta, float eps, float* coeffs,
                                 void* stream) {
  if (!running_mean || !running_var || !coeffs) return GS_E_NULL;
  if (C <= 0) return GS_E_BADARG;
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((C + 255) / 256), dim3(256), 0, as_stream(stream),
                     running_mean, running_var, C, gamma, beta, eps, coeffs);
  return launch_status();
}

static int bn_apply_impl(const float* x, int64_t rows, int32_t C, int32_t ldx, const float* coeffs,
                         const float* residual, int32_t ld_res, int32_t relu, float* y, int32_t ldy,
                         unsigned char* mask, void* stream, const float* rcoeffs = nullptr) {
  int rc = check_rows(x, rows, C, ldx);
  if (rc) return rc;
  rc = check_rows(y, rows, C, ldy);
  if (rc) return rc;
  if (!coeffs) return GS_E_NULL;
  if (!aligned16(coeffs)) return GS_E_ALIGN;
  if (residual && (rc = check_rows(residual, rows, C, ld_res))) return rc;
  const dim3 grid(apply_grid(rows, C), (C >> 2) <= 256 ? 1 : (unsigned)ceil_div(C >> 2, 256));
  hipStream_t st = as_stream(stream);
  if (rcoeffs) {   // the residual is a raw conv output with its own BatchNorm coefficients
    if (!residual) return GS_E_BADARG;
    if (!aligned16(rcoeffs)) return GS_E_ALIGN;
    if (mask && !relu) return GS_E_BADARG;
    if (mask)
      hipLaunchKernelGGL((bn_apply_kernel<true, true, true, true>), grid, dim3(256), 0, st, x, (long)rows,
                         C, ldx, coeffs, residual, ld_res, y, ldy, mask, rcoeffs);
    else if (relu)
      hipLaunchKernelGGL((bn_apply_kernel<true, true, false, true>), grid, dim3(256), 0, st, x, (long)rows,
                         C, ldx, coeffs, residual, ld_res, y, ldy, (unsigned char*)nullptr, rcoeffs);
    else
      hipLaunchKernelGGL((bn_apply_kernel<true, false, false, true>), grid, dim3(256), 0, st, x, (long)rows,
                         C, ldx, coeffs, residual, ld_res, y, ldy, (unsigned char*)nullptr, rcoeffs);
    return launch_status();
  }
#define GS_APPLY(R, A)                                                                          \
  hipLaunchKernelGGL((bn_apply_kernel<R, A>), grid, dim3(256), 0, st, x, (long)rows, C, ldx,    \
                     coeffs, residual, ld_res, y, ldy, (unsigned char*)nullptr)
  if (mask) {
    if (!relu) return GS_E_BADARG;
    if (residual)
      hipLaunchKernelGGL((bn_apply_kernel<true, true, true>), grid, dim3(256), 0, st, x, (long)rows, C,
                         ldx, coeffs, residual, ld_res, y, ldy, mask);
    else
      hipLaunchKernelGGL((bn_apply_kernel<false, true, true>), grid, dim3(256), 0, st, x, (long)rows, C,
                         ldx, coeffs, residual, ld_res, y, ldy, mask);
    return launch_status();
  }
  if (residual) { if (relu) GS_APPLY(true, true); else GS_APPLY(true, false); }
  else { if (relu) GS_APPLY(false, true); else GS_APPLY(false, false); }
#undef GS_APPLY
  return launch_status();
}

extern "C" int gs_bn_apply(const float* x, int64_t rows, int32_t C, int32_t ldx,
                           const float* coeffs, const float* residual, int32_t ld_res,
                           int32_t relu, float* y, int32_t ldy, void* stream) {
  return bn_apply_impl(x, rows, C, ldx, coeffs, residual, ld_res, relu, y, ldy, nullptr, stream);
}

extern "C" int gs_bn_apply_mask(const float* x, int64_t rows, int32_t C, int32_t ldx,
                                const float* coeffs, const float* residual, int32_t ld_res, float* y,
                                int32_t ldy, uint8_t* mask, void* stream) {
  if (!mask) return GS_E_NULL;
  return bn_apply_impl(x, rows, C, ldx, coeffs, residual, ld_res, 1, y, ldy, mask, stream);
}

namespace gs {
// gs_conv_bn_forward's apply pass (fused_layers.hip): residual with its own coefficients, optional mask
int bn_apply_resaff(const float* x, int64_t rows, int32_t C, int32_t ldx, const float* coeffs,
                    const float* residual, int32_t ld_res, const float* rcoeffs, int32_t relu, float* y,
                    int32_t ldy, uint8_t* mask, void* stream) {
  if (!rcoeffs) return GS_E_NULL;
  return bn_apply_impl(x, rows, C, ldx, coeffs, residual, ld_res, relu, y, ldy, mask, stream, rcoeffs);
}
}  // namespace gs

extern "C" int gs_bn_bwd_reduce(const float* dy, int32_t ld_dy, const float* x, int32_t ldx,
                                const float* act, int32_t ld_act, int64_t rows, int32_t C,
                                const float* coeffs, int32_t mask_mode, float* g_out, int32_t ld_g,
                                float* sums, void* workspace, size_t workspace_bytes,
                                void* stream) {
  int rc = check_rows(dy, rows, C, ld_dy);
  if (rc) return rc;
  if ((rc = check_rows(x, rows, C, ldx))) return rc;
  if (mask_mode < 0 || mask_mode > 2) return GS_E_BADARG;
  if (mask_mode == 2 && (rc = check_rows(act, rows, C, ld_act))) return rc;
  if (g_out && (rc = check_rows(g_out, rows, C, ld_g))) return rc;
  if (!coeffs || !sums || !workspace) return GS_E_NULL;
  if (!aligned16(coeffs) || !aligned16(workspace)) return GS_E_ALIGN;
  const RedGeom g = red_geom(rows, C);
  if ((size_t)g.gx * 2 * C * sizeof(float) > workspace_bytes) return GS_E_WORKSPACE;
  hipStream_t st = as_stream(stream);
  float* part = static_cast<float*>(workspace);
#define GS_BWDP(MK)                                                                             \
  hipLaunchKernelGGL((bn_bwd_partial_kernel<MK>), dim3(g.gx, g.gy), dim3(256), 0, st, dy, ld_dy, \
                     x, ldx, act, ld_act, (long)rows, C, coeffs, g.rows_per_block, g_out, ld_g, part)
  if (mask_mode == 0) GS_BWDP(0); else if (mask_mode == 1) GS_BWDP(1); else GS_BWDP(2);
#undef GS_BWDP
  launch_sum_partials(part, g.gx, 2 * C, sums, nullptr, 0, st);
  return launch_status();
}

extern "C" int gs_bn_bwd_apply(const float* dy, int32_t ld_dy, const float* x, int32_t ldx,
                               const float* act, int32_t ld_act, int64_t rows, int32_t C,
                               const float* coeffs, const float* sums, double count,
                               int32_t mask_mode, int32_t use_batch_stats, float* dx,
                               int32_t ld_dx, float* dgamma, float* dbeta, void* stream) {
  int rc = check_rows(dy, rows, C, ld_dy);
  if (rc) return rc;
  if ((rc = check_rows(x, rows, C, ldx))) return rc;
  if ((rc = check_rows(dx, rows, C, ld_dx))) return rc;
  if (mask_mode < 0 || mask_mode > 2 || count <= 0.0) return GS_E_BADARG;
  if (mask_mode == 2 && (rc = check_rows(act, rows, C, ld_act))) return rc;
  if (!coeffs || !sums) return GS_E_NULL;
  if (!aligned16(coeffs) || !aligned16(sums) || (dgamma && !aligned16(dgamma)) ||
      (dbeta && !aligned16(dbeta)))
    return GS_E_ALIGN;
  const dim3 grid(apply_grid(rows, C), (C >> 2) <= 256 ? 1 : (unsigned)ceil_div(C >> 2, 256));
  hipStream_t st = as_stream(stream);
  const float inv = (float)(1.0 / count);
#define GS_BWDA(MK, B)                                                                           \
  hipLaunchKernelGGL((bn_bwd_apply_kernel<MK, B>), grid, dim3(256), 0, st, dy, ld_dy, x, ldx, act, \
                     ld_act, (long)rows, C, coeffs, sums, inv, dx, ld_dx, dgamma, dbeta)
  if (use_batch_stats) {
    if (mask_mode == 0) GS_BWDA(0, true); else if (mask_mode == 1) GS_BWDA(1, true); else GS_BWDA(2, true);
  } else {
    if (mask_mode == 0) GS_BWDA(0, false); else if (mask_mode == 1) GS_BWDA(1, false); else GS_BWDA(2, false);
  }
#undef GS_BWDA
  return launch_status();
}

extern "C" int gs_colsum(const float* src, int64_t rows, int32_t C, int32_t ld, float* out,
                         void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_rows(src, rows, C, ld);
  if (rc) return rc;
  if (!out || !workspace) return GS_E_NULL;
  const RedGeom g = red_geom(rows, C);
  if ((size_t)g.gx * C * sizeof(float) > workspace_bytes) return GS_E_WORKSPACE;
  if (!aligned16(workspace)) return GS_E_ALIGN;
  hipStream_t st = as_stream(stream);
  float* part = static_cast<float*>(workspace);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(g.gx, g.gy), dim3(256), 0, st, src, (long)rows, C,
                     ld, g.rows_per_block, part);
  launch_sum_partials(part, g.gx, C, out, nullptr, 0, st);
  return launch_status();
}

extern "C" int gs_bn_sync_local(const float* sums, double count, int32_t C, double* local,
                                void* stream) {
  if (!sums || !local) return GS_E_NULL;
  if (C <= 0 || count <= 0.0) return GS_E_BADARG;
  hipLaunchKernelGGL(bn_sync_local_kernel, dim3((C + 255) / 256), dim3(256), 0, as_stream(stream),
                     sums, count, C, local);
  return launch_status();
}

extern "C" int gs_bn_sync_merge(const double* gathered, int32_t world, int32_t C, float* merged,
                                void* stream) {
  if (!gathered || !merged) return GS_E_NULL;
  if (C <= 0 || world <= 0) return GS_E_BADARG;
  hipLaunchKernelGGL(bn_sync_merge_kernel, dim3((C + 255) / 256), dim3(256), 0, as_stream(stream),
                     gathered, world, C, merged);
  return launch_status();
}
