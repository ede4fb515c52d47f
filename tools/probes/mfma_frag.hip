// Probe: LDS fragment traffic of the 64x64 fp32 conv tile (4 waves, wave tile 16 x 64, BK = 16).
//   scheme 0 = the r02 kernel: k-major float images; per k-group 1 + 4 ds_read_b32; stage stores
//              4 transposing ds_write_b32 (A) + 1 ds_write_b128 (B)
//   scheme 1 = 16-byte fragments: A as [k/4][m] chunks (one ds_read_b128 per K step supplies the four
//              k-groups: group e contracts channels {e, 4+e, 8+e, 12+e}), B k-major with the wave's
//              four column tiles interleaved (tile j owns columns 4*l + j): one ds_read_b128 per
//              k-group; stage stores one ds_write_b128 each
// Both: global -> VGPR -> LDS staging one K step ahead, double-buffered stages, one barrier per step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int PA = 80, PB = 80;            // scheme 0 pitches (floats)
constexpr int A0_SZ = 16 * PA + 32, B0_SZ = 16 * PB + 32;
constexpr int PQ = 66 * 4;                 // scheme 1: floats per k/4 plane of A (64 chunks + 2 pad)
constexpr int A1_SZ = 4 * PQ, PB1 = 64, B1_SZ = 16 * PB1;
constexpr int STAGE = 2720;                // >= both

template <int SCHEME, int STORES>
__global__ __launch_bounds__(256) void kb(float* out, int iters, const float* src) {
  __shared__ __attribute__((aligned(16))) float sh[2 * STAGE];
  f32x4 acc[4];
  for (int j = 0; j < 4; ++j) acc[j] = f32x4{0, 0, 0, 0};
  for (int i = threadIdx.x; i < 2 * STAGE; i += 256) sh[i] = (i % 97) * 1e-3f;
  __syncthreads();
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int kk = lane >> 4, li = lane & 15;
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 1 << 26, 0x00020000);
  f32x4 ra0{0, 0, 0, 0}, rb0{0, 0, 0, 0}, ra1{0, 0, 0, 0}, rb1{0, 0, 0, 0};
  unsigned goff = (blockIdx.x * 8192u + t * 16u) & ((1u << 24) - 1);
  // one K step: computes from cb, stores the set loaded two steps ago into nb, refills that set
  auto step = [&](f32x4& ra, f32x4& rb, const float* cb, float* nb) __attribute__((always_inline)) {
    if (SCHEME == 0) {
      const float* As = cb; const float* Bs = cb + A0_SZ;
      float fa[2], fb[2][4];
      fa[0] = As[kk * PA + wave * 16 + li];
      for (int j = 0; j < 4; ++j) fb[0][j] = Bs[kk * PB + j * 16 + li];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cur = g & 1, nxt = cur ^ 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[cur], fb[cur][j], acc[j], 0, 0, 0);
          if (g < 3 && j == 0) fa[nxt] = As[((g + 1) * 4 + kk) * PA + 8 * (g + 1) + wave * 16 + li];
          if (g < 3 && j == 1)
            for (int jj = 0; jj < 4; ++jj)
              fb[nxt][jj] = Bs[((g + 1) * 4 + kk) * PB + 8 * (g + 1) + jj * 16 + li];
          if (g == 2 && j == 2) {
            goff = (goff + 8192u * 2048u) & ((1u << 24) - 1);
            if (STORES) {
              ra = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, goff, 0, 0));
              rb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, goff + 4096u, 0, 0));
            }
          }
          if (STORES && g == 1 && j == 2) {
            const int i = t >> 2, kq = t & 3;
#pragma unroll
            for (int e = 0; e < 4; ++e) nb[(kq * 4 + e) * PA + 8 * kq + i] = ra[e];
          }
          if (STORES && g == 1 && j == 3) {
            const int k = t >> 4, n4 = t & 15;
            *reinterpret_cast<f32x4*>(&nb[A0_SZ + k * PB + 8 * (k >> 2) + n4 * 4]) = rb;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
      const float* As = cb; const float* Bs = cb + A1_SZ;
      f32x4 fa = *reinterpret_cast<const f32x4*>(&As[kk * PQ + (wave * 16 + li) * 4]);
      f32x4 fb[2];
      fb[0] = *reinterpret_cast<const f32x4*>(&Bs[(4 * kk + 0) * PB1 + li * 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int cur = e & 1, nxt = cur ^ 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[e], fb[cur][j], acc[j], 0, 0, 0);
          if (e < 3 && j == 0)
            fb[nxt] = *reinterpret_cast<const f32x4*>(&Bs[(4 * kk + e + 1) * PB1 + li * 4]);
          if (e == 2 && j == 2) {
            goff = (goff + 8192u * 2048u) & ((1u << 24) - 1);
            if (STORES) {
              ra = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, goff, 0, 0));
              rb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, goff + 4096u, 0, 0));
            }
          }
          if (STORES && e == 1 && j == 2) {
            const int i = t >> 2, kq = t & 3;
            *reinterpret_cast<f32x4*>(&nb[kq * PQ + i * 4]) = ra;
          }
          if (STORES && e == 1 && j == 3) {
            const int k = t >> 4, n4 = t & 15;
            *reinterpret_cast<f32x4*>(&nb[A1_SZ + k * PB1 + n4 * 4]) = rb;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    __syncthreads();
  };
  for (int it = 0; it < iters; it += 2) {
    step(ra0, rb0, sh, sh + STAGE);
    step(ra1, rb1, sh + STAGE, sh);
  }
  f32x4 s{0, 0, 0, 0};
  for (int j = 0; j < 4; ++j) s += acc[j];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + ra0[0] + rb0[0] + ra1[0] + rb1[0];
}

template <int SCHEME, int STORES>
void run(const char* nm, float* out, int bpc, const float* src) {
  const int blocks = 256 * bpc, iters = 4000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((kb<SCHEME, STORES>), dim3(blocks), dim3(256), 0, 0, out, iters, src);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((kb<SCHEME, STORES>), dim3(blocks), dim3(256), 0, 0, out, iters, src);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double fl = 2048.0 * 16 * iters * 4.0 * blocks;
  printf("%-58s WG/CU=%d %9.1f us %7.1f TF\n", nm, bpc, ms * 1e3, fl / (ms * 1e-3) / 1e12);
}

int main() {
  float* out; CK(hipMalloc(&out, 8192 * 256 * 4));
  float* src; CK(hipMalloc(&src, 1 << 26)); CK(hipMemset(src, 0, 1 << 26));
  for (int bpc = 1; bpc <= 3; ++bpc) {
    run<0, 0>("b32 fragments (r02), reads only", out, bpc, src);
    run<1, 0>("b128 fragments, reads only", out, bpc, src);
    run<0, 1>("b32 fragments (r02) + global->VGPR->LDS staging", out, bpc, src);
    run<1, 1>("b128 fragments + global->VGPR->LDS staging", out, bpc, src);
  }
  return 0;
}
