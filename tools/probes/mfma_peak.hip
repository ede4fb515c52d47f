// Pure-MFMA throughput probe (fp32): how close to the nominal 157.3 TF can any loop get?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, long long* clk) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  long long c1 = clock64(), w1 = wall_clock64();
  f32x4 s{0, 0, 0, 0};
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, long long* clk) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  long long c1 = clock64(), w1 = wall_clock64();
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <typename F>
void bench(const char* nm, F launch, double flop_per_wave_iter, int iters, int blocks) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double fl = flop_per_wave_iter * iters * 4.0 * blocks;
  printf("%-34s %8.1f us %7.1f TF", nm, ms * 1e3, fl / (ms * 1e-3) / 1e12);
}

int main() {
  float* out; long long* clk; CK(hipMalloc(&out, 4096 * 256 * 4)); CK(hipMalloc(&clk, 16));
  const int iters = 4000;
  long long h[2];
  for (int bpc = 1; bpc <= 4; bpc *= 2) {
    const int blocks = 256 * bpc;
#define RUN16(N) { char nm[64]; snprintf(nm, 64, "16x16x4 acc=%d blocks/CU=%d", N, bpc); \
    bench(nm, [&] { hipLaunchKernelGGL(k16<N>, dim3(blocks), dim3(256), 0, 0, out, iters, clk); }, 2048.0 * N, iters, blocks); \
    CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost)); printf("   clk %.0f MHz (cycles %lld per iter %.1f)\n", h[0] * 100.0 / h[1], h[0], (double)h[0] / iters); }
#define RUN32(N) { char nm[64]; snprintf(nm, 64, "32x32x2 acc=%d blocks/CU=%d", N, bpc); \
    bench(nm, [&] { hipLaunchKernelGGL(k32<N>, dim3(blocks), dim3(256), 0, 0, out, iters, clk); }, 4096.0 * N, iters, blocks); \
    CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost)); printf("   clk %.0f MHz (cycles %lld per iter %.1f)\n", h[0] * 100.0 / h[1], h[0], (double)h[0] / iters); }
    RUN16(1) RUN16(2) RUN16(4) RUN16(16) RUN32(1) RUN32(2) RUN32(4)
  }
  return 0;
}
