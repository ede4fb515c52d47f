// Probe: wave tile TM x TN (16x16 MFMA tiles), 4 k-groups per barrier, LDS fragment reads like the conv kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int TM, int TN, int GPB /*groups per barrier*/, int LDSW>
__global__ __launch_bounds__(256) void kb(float* out, int iters, const float* src) {
  __shared__ float sh[2 * 2624];
  f32x4 acc[TM][TN];
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  for (int i = threadIdx.x; i < 2 * 2624; i += 256) sh[i] = (i % 97) * 1e-3f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kk = lane >> 4, li = lane & 15;
  float fa[2][TM], fb[2][TN];
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 1 << 26, 0x00020000);
  f32x4 ra{0, 0, 0, 0}, rb{0, 0, 0, 0};
  unsigned goff = (blockIdx.x * 8192u + threadIdx.x * 16u) & ((1u << 24) - 1);
  const float* As = sh; const float* Bs = sh + 1312;
  for (int i = 0; i < TM; ++i) fa[0][i] = As[kk * 80 + wave * 16 + i * 16 + li];
  for (int j = 0; j < TN; ++j) fb[0][j] = Bs[kk * 80 + j * 16 + li];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < GPB; ++g) {
      const int cur = g & 1, nxt = cur ^ 1;
#pragma unroll
      for (int q = 0; q < TM * TN; ++q) {
        const int i = q / TN, j = q % TN;
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
        if (q == 0) for (int ii = 0; ii < TM; ++ii) fa[nxt][ii] = As[(((g + 1) & 3) * 4 + kk) * 80 + 8 * ((g + 1) & 3) + (wave * TM * 16) % 64 + ii * 16 + li];
        if (q == (TM * TN > 1 ? 1 : 0)) for (int jj = 0; jj < TN; ++jj) fb[nxt][jj] = Bs[(((g + 1) & 3) * 4 + kk) * 80 + 8 * ((g + 1) & 3) + (jj * 16) % 64 + li];
        if (LDSW == 1 && g == 1 && q == 2 % (TM * TN)) {  // stage store like the conv kernel: 2 x 16 B per thread
          *reinterpret_cast<f32x4*>(&sh[2624 + (threadIdx.x * 4) % 1300]) = f32x4{fa[cur][0], 1.f, 2.f, 3.f};
          *reinterpret_cast<f32x4*>(&sh[2624 + 1312 + (threadIdx.x * 4) % 1300]) = f32x4{fb[cur][0], 1.f, 2.f, 3.f};
        }
        if (LDSW == 2 && g == 1 && q == 2 % (TM * TN)) {  // registers loaded one step ago -> LDS, then reload
          *reinterpret_cast<f32x4*>(&sh[2624 + (threadIdx.x * 4) % 1300]) = ra;
          *reinterpret_cast<f32x4*>(&sh[2624 + 1312 + (threadIdx.x * 4) % 1300]) = rb;
          goff = (goff + 8192u * 2048u) & ((1u << 24) - 1);
          ra = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, goff, 0, 0));
          rb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, goff + 4096u, 0, 0));
        }
        if (LDSW == 3 && g == 0 && q == 2 % (TM * TN)) {  // LDS-DMA: global -> LDS, no VGPRs, no ds_write
          goff = (goff + 8192u * 2048u) & ((1u << 24) - 1);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(sh + 2624 + wave * 256), 16, goff, 0, 0, 0);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(sh + 2624 + 1312 + wave * 256), 16, goff + 4096u, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (LDSW == 3) __builtin_amdgcn_s_waitcnt(0x0f70 & 0x3f70);
    __syncthreads();
  }
  f32x4 s{0, 0, 0, 0};
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + ra[0] + rb[0];
}

template <int TM, int TN, int GPB, int LDSW>
void run(const char* nm, float* out, int bpc, const float* src) {
  const int blocks = 256 * bpc, iters = 4000 * 16 / (TM * TN * GPB);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((kb<TM, TN, GPB, LDSW>), dim3(blocks), dim3(256), 0, 0, out, iters, src);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((kb<TM, TN, GPB, LDSW>), dim3(blocks), dim3(256), 0, 0, out, iters, src);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double fl = 2048.0 * TM * TN * GPB * iters * 4.0 * blocks;
  printf("%-44s WG/CU=%d %9.1f us %7.1f TF\n", nm, bpc, ms * 1e3, fl / (ms * 1e-3) / 1e12);
}

int main() {
  float* out; CK(hipMalloc(&out, 8192 * 256 * 4));
  float* src; CK(hipMalloc(&src, 1 << 26)); CK(hipMemset(src, 0, 1 << 26));
  for (int bpc = 1; bpc <= 5; bpc += 2) {
    run<1, 4, 4, 0>("64x64 tile (1x4/wave), barrier per 4 groups", out, bpc, src);
    run<1, 4, 4, 1>("  + stage stores (ds_write of registers)", out, bpc, src);
    run<1, 4, 4, 2>("  + global load -> VGPR -> ds_write", out, bpc, src);
    run<1, 4, 4, 3>("  + LDS-DMA (buffer_load ... lds)", out, bpc, src);
    run<2, 8, 4, 2>("128x128 (2x8/wave) global -> VGPR -> ds_write", out, bpc, src);
    run<2, 8, 4, 3>("128x128 (2x8/wave) LDS-DMA", out, bpc, src);
  }
  return 0;
}
