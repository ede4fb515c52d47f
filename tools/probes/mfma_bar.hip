// MFMA + barrier probe: 16 accumulators, G groups of 16 MFMAs between barriers; optional LDS reads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int G, bool BAR, bool LDSR>
__global__ __launch_bounds__(256) void kb(float* out, int iters) {
  __shared__ float sh[8192];
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
  for (int i = threadIdx.x; i < 8192; i += 256) sh[i] = i * 1e-4f;
  __syncthreads();
  float a[2] = {threadIdx.x * 1e-3f, 0.5f}, b[8];
  for (int j = 0; j < 8; ++j) b[j] = blockIdx.x * 1e-3f + j;
  const int lane = threadIdx.x & 63;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (LDSR) {
        const float* pa = sh + g * 528 + lane;
        a[0] = pa[0]; a[1] = pa[16 * 4];
#pragma unroll
        for (int j = 0; j < 8; ++j) b[j] = pa[2048 + j * 16];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
          acc[i * 8 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i * 8 + j], 0, 0, 0);
    }
    if (BAR) __syncthreads();
  }
  f32x4 s{0, 0, 0, 0};
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <int G, bool BAR, bool LDSR>
void run(const char* nm, float* out, int bpc) {
  const int blocks = 256 * bpc, iters = 20000 / G;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((kb<G, BAR, LDSR>), dim3(blocks), dim3(256), 0, 0, out, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((kb<G, BAR, LDSR>), dim3(blocks), dim3(256), 0, 0, out, iters);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double fl = 2048.0 * 16 * G * iters * 4.0 * blocks;
  printf("%-40s WG/CU=%d %9.1f us %7.1f TF\n", nm, bpc, ms * 1e3, fl / (ms * 1e-3) / 1e12);
}

int main() {
  float* out; CK(hipMalloc(&out, 4096 * 256 * 4));
  for (int bpc = 1; bpc <= 3; ++bpc) {
    run<4, false, false>("G=4 mfma only", out, bpc);
    run<4, true, false>("G=4 mfma + barrier", out, bpc);
    run<4, false, true>("G=4 mfma + ldsread", out, bpc);
    run<4, true, true>("G=4 mfma + ldsread + barrier", out, bpc);
    run<2, true, true>("G=2 mfma + ldsread + barrier", out, bpc);
    run<1, true, true>("G=1 mfma + ldsread + barrier", out, bpc);
  }
  return 0;
}
