// Probe for the r03 plan: fp32 contraction as SIX bf16 MFMAs over an exact three-way bf16 split of
// both operands (x = x0 + x1 + x2, each piece 8 mantissa bits; products a_i * b_j with i + j <= 2,
// fp32 accumulation inside v_mfma_f32_16x16x32_bf16).  Two questions:
//   1. accuracy against fp64, next to the fp32 MFMA (v_mfma_f32_16x16x4_f32) on the same data;
//   2. throughput of a 64x64 tile K loop that stages fp32 from global memory, splits at the stage
//      store (once per element and workgroup), keeps bf16 triples in LDS (6 B / element, rows padded
//      to 80 B so that 16-byte fragment reads are conflict-free) and reads 16-byte fragments.
// Both operands are taken k-contiguous (the dgrad case; the forward's B operand needs an 8k x 4n
// register micro-tile per thread instead, same LDS traffic).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ROWB = 80;                 // bytes per (row, piece): 32 bf16 + 16 B pad
constexpr int PIECE = 64 * ROWB;         // one piece of a 64-row operand
constexpr int OPER = 3 * PIECE;          // three pieces
constexpr int STAGEB = 2 * OPER;         // A + B

// three bf16 pieces of 8 consecutive fp32 values -> 3 x 16 bytes (truncating split: exact, since
// 8 + 8 + 8 mantissa bits cover fp32's 24)
__device__ __forceinline__ u32x4 pack_hi16(const u32x4 a, const u32x4 b) {
  // {bf16(a0), bf16(a1)}, {a2, a3}, {b0, b1}, {b2, b3}: element 2q in the low half of dword q
  return u32x4{(a[0] >> 16) | (a[1] & 0xFFFF0000u), (a[2] >> 16) | (a[3] & 0xFFFF0000u),
               (b[0] >> 16) | (b[1] & 0xFFFF0000u), (b[2] >> 16) | (b[3] & 0xFFFF0000u)};
}
__device__ __forceinline__ void split8(const f32x4 lo4, const f32x4 hi4, u32x4& p0, u32x4& p1, u32x4& p2) {
  const u32x4 mask{0xFFFF0000u, 0xFFFF0000u, 0xFFFF0000u, 0xFFFF0000u};
  const u32x4 hl = __builtin_bit_cast(u32x4, lo4) & mask, hh = __builtin_bit_cast(u32x4, hi4) & mask;
  const f32x4 r1l = lo4 - __builtin_bit_cast(f32x4, hl), r1h = hi4 - __builtin_bit_cast(f32x4, hh);
  const u32x4 ml = __builtin_bit_cast(u32x4, r1l) & mask, mh = __builtin_bit_cast(u32x4, r1h) & mask;
  const f32x4 r2l = r1l - __builtin_bit_cast(f32x4, ml), r2h = r1h - __builtin_bit_cast(f32x4, mh);
  p0 = pack_hi16(hl, hh);
  p1 = pack_hi16(ml, mh);
  p2 = pack_hi16(__builtin_bit_cast(u32x4, r2l), __builtin_bit_cast(u32x4, r2h));
}

// C[64][64] (+)= A[64][K] * Bt[64][K]^T; grid = tiles; every workgroup streams its own K range of
// `src` (L2 resident) so that the loop is fed like the conv kernels' gather.
template <int MODE /*0: bf16x3, 1: fp32 MFMA*/>
__global__ __launch_bounds__(256) void gemm_tile(const float* __restrict__ A, const float* __restrict__ Bt,
                                                 float* __restrict__ C, int K, int lda, long tile_stride) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGEB];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* a = A + (long)blockIdx.x * tile_stride;
  const float* b = Bt + (long)blockIdx.x * tile_stride;
  f32x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = f32x4{0, 0, 0, 0};
  const int row = t >> 2, kq = t & 3;   // staging: 8 consecutive k of one row per thread and operand
  const int li = lane & 15, fk = lane >> 4;
  if (MODE == 0) {
    f32x4 ra0, ra1, rb0, rb1;
    auto gload = [&](int k0) {
      ra0 = *reinterpret_cast<const f32x4*>(a + (long)row * lda + k0 + kq * 8);
      ra1 = *reinterpret_cast<const f32x4*>(a + (long)row * lda + k0 + kq * 8 + 4);
      rb0 = *reinterpret_cast<const f32x4*>(b + (long)row * lda + k0 + kq * 8);
      rb1 = *reinterpret_cast<const f32x4*>(b + (long)row * lda + k0 + kq * 8 + 4);
    };
    auto sstore = [&](unsigned char* st) {
      u32x4 p0, p1, p2;
      split8(ra0, ra1, p0, p1, p2);
      unsigned char* pa = st + row * ROWB + kq * 16;
      *reinterpret_cast<u32x4*>(pa) = p0; *reinterpret_cast<u32x4*>(pa + PIECE) = p1; *reinterpret_cast<u32x4*>(pa + 2 * PIECE) = p2;
      split8(rb0, rb1, p0, p1, p2);
      unsigned char* pb = st + OPER + row * ROWB + kq * 16;
      *reinterpret_cast<u32x4*>(pb) = p0; *reinterpret_cast<u32x4*>(pb + PIECE) = p1; *reinterpret_cast<u32x4*>(pb + 2 * PIECE) = p2;
    };
    gload(0);
    sstore(lds);
    __syncthreads();
    const int nst = K / 32;
    for (int s = 0; s < nst; ++s) {
      const unsigned char* cb = lds + (s & 1) * STAGEB;
      if (s + 1 < nst) gload((s + 1) * 32);
      bf16x8 fa[3], fb[3];
#pragma unroll
      for (int p = 0; p < 3; ++p)
        fa[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(cb + p * PIECE + (wave * 16 + li) * ROWB + fk * 16));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
          fb[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(cb + OPER + p * PIECE + (j * 16 + li) * ROWB + fk * 16));
        // smallest terms first
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[2], fb[0], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1], fb[1], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[2], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1], fb[0], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[1], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[0], acc[j], 0, 0, 0);
      }
      if (s + 1 < nst) sstore(lds + ((s + 1) & 1) * STAGEB);
      __syncthreads();
    }
  } else {
    // fp32 MFMA reference loop with the same staging granularity (k-contiguous rows in LDS, fp32)
    float* lf = reinterpret_cast<float*>(lds);   // [2][A 64 x 36 | B 64 x 36]
    constexpr int P = 36, OP = 64 * P, ST = 2 * OP;
    f32x4 ra0, ra1, rb0, rb1;
    auto gload = [&](int k0) {
      ra0 = *reinterpret_cast<const f32x4*>(a + (long)row * lda + k0 + kq * 8);
      ra1 = *reinterpret_cast<const f32x4*>(a + (long)row * lda + k0 + kq * 8 + 4);
      rb0 = *reinterpret_cast<const f32x4*>(b + (long)row * lda + k0 + kq * 8);
      rb1 = *reinterpret_cast<const f32x4*>(b + (long)row * lda + k0 + kq * 8 + 4);
    };
    auto sstore = [&](float* st) {
      *reinterpret_cast<f32x4*>(st + row * P + kq * 8) = ra0; *reinterpret_cast<f32x4*>(st + row * P + kq * 8 + 4) = ra1;
      *reinterpret_cast<f32x4*>(st + OP + row * P + kq * 8) = rb0; *reinterpret_cast<f32x4*>(st + OP + row * P + kq * 8 + 4) = rb1;
    };
    gload(0); sstore(lf); __syncthreads();
    const int nst = K / 32;
    for (int s = 0; s < nst; ++s) {
      const float* cb = lf + (s & 1) * ST;
      if (s + 1 < nst) gload((s + 1) * 32);
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const float av = cb[(wave * 16 + li) * P + g * 4 + fk];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float bv = cb[OP + (j * 16 + li) * P + g * 4 + fk];
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[j], 0, 0, 0);
        }
      }
      if (s + 1 < nst) sstore(lf + ((s + 1) & 1) * ST);
      __syncthreads();
    }
  }
  float* c = C + (long)blockIdx.x * 64 * 64;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) c[(wave * 16 + fk * 4 + r) * 64 + j * 16 + li] = acc[j][r];
}

static void accuracy(int K, float spread) {
  std::vector<float> A(64L * K), B(64L * K);
  srand(1);
  auto rnd = [&]() { float u = (rand() % 20001 - 10000) * 1e-4f; float e = spread > 0 ? ldexpf(1.f, (rand() % (int)(2 * spread + 1)) - (int)spread) : 1.f; return u * e; };
  for (auto& v : A) v = rnd();
  for (auto& v : B) v = rnd();
  float *dA, *dB, *dC;
  CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, 64 * 64 * 4));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  std::vector<float> c0(4096), c1(4096);
  hipLaunchKernelGGL((gemm_tile<0>), dim3(1), dim3(256), 0, 0, dA, dB, dC, K, K, 0L);
  CK(hipMemcpy(c0.data(), dC, 4096 * 4, hipMemcpyDeviceToHost));
  hipLaunchKernelGGL((gemm_tile<1>), dim3(1), dim3(256), 0, 0, dA, dB, dC, K, K, 0L);
  CK(hipMemcpy(c1.data(), dC, 4096 * 4, hipMemcpyDeviceToHost));
  double e0 = 0, e1 = 0, r0 = 0, r1 = 0;
  for (int i = 0; i < 64; ++i)
    for (int j = 0; j < 64; ++j) {
      double ref = 0, mag = 0;
      for (int k = 0; k < K; ++k) { const double p = (double)A[(long)i * K + k] * B[(long)j * K + k]; ref += p; mag += fabs(p); }
      e0 = fmax(e0, fabs(c0[i * 64 + j] - ref) / mag); e1 = fmax(e1, fabs(c1[i * 64 + j] - ref) / mag);
      r0 += pow((c0[i * 64 + j] - ref) / mag, 2); r1 += pow((c1[i * 64 + j] - ref) / mag, 2);
    }
  printf("K=%5d exponent spread +-%2.0f: |err| / sum|a||b|  bf16x3 (6 MFMA) max %.2e rms %.2e   fp32 MFMA max %.2e rms %.2e\n",
         K, spread, e0, sqrt(r0 / 4096), e1, sqrt(r1 / 4096));
  CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
}

template <int MODE>
static void speed(const char* name, int K, int wg_per_cu) {
  const int tiles = 256 * wg_per_cu;
  const long stride = 0;   // all tiles read the same (L2-resident) operands: the K loop is what is timed
  float *dA, *dB, *dC;
  CK(hipMalloc(&dA, (size_t)64 * K * 4)); CK(hipMalloc(&dB, (size_t)64 * K * 4)); CK(hipMalloc(&dC, (size_t)tiles * 4096 * 4));
  CK(hipMemset(dA, 0, (size_t)64 * K * 4)); CK(hipMemset(dB, 0, (size_t)64 * K * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_tile<MODE>), dim3(tiles), dim3(256), 0, 0, dA, dB, dC, K, K, stride);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((gemm_tile<MODE>), dim3(tiles), dim3(256), 0, 0, dA, dB, dC, K, K, stride);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double fl = 2.0 * 64 * 64 * K * tiles * 10;
  printf("%-28s K=%5d WG/CU=%d  %8.1f us/launch  %7.1f TF (fp32-equivalent)\n", name, K, wg_per_cu, ms * 100, fl / (ms * 1e-3) / 1e12);
  CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
}

static void layout_check() {
  const int K = 64;
  std::vector<float> A(64L * K), B(64L * K);
  for (int i = 0; i < 64; ++i) for (int k = 0; k < K; ++k) { A[(long)i * K + k] = (float)((i * 7 + k * 3) % 5 - 2); B[(long)i * K + k] = (float)((i * 5 + k) % 7 - 3); }
  float *dA, *dB, *dC;
  CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, 64 * 64 * 4));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  std::vector<float> c0(4096);
  hipLaunchKernelGGL((gemm_tile<0>), dim3(1), dim3(256), 0, 0, dA, dB, dC, K, K, 0L);
  CK(hipMemcpy(c0.data(), dC, 4096 * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < 64; ++i) for (int j = 0; j < 64; ++j) {
    double ref = 0; for (int k = 0; k < K; ++k) ref += (double)A[(long)i * K + k] * B[(long)j * K + k];
    if (fabs(ref - c0[i * 64 + j]) > 1e-3) { if (bad < 6) printf("  C[%d][%d] = %g, expected %g\n", i, j, c0[i * 64 + j], ref); ++bad; }
  }
  printf("layout check (integer data, K=64): %d of 4096 wrong\n", bad);
  if (bad && getenv("GS_DUMP")) {
    FILE* f = fopen(getenv("GS_DUMP"), "w");
    for (int i = 0; i < 64; ++i) { for (int j = 0; j < 64; ++j) fprintf(f, "%g ", c0[i * 64 + j]); fprintf(f, "\n"); }
    fclose(f);
  }
}

int main() {
  layout_check();
  accuracy(576, 0); accuracy(4608, 0); accuracy(4608, 6); accuracy(18432, 0);
  for (int wg = 1; wg <= 2; ++wg) {
    speed<1>("fp32 MFMA, simple loop", 2304, wg);
    speed<0>("bf16x3 split, 6 MFMA", 2304, wg);
  }
  speed<1>("fp32 MFMA, simple loop", 576, 4);
  speed<0>("bf16x3 split, 6 MFMA", 576, 2);
  return 0;
}
