// Shared helpers for the gfx950 kernels of the GAIA-seg supernet hot path.
// (MI355X only: 64-lane waves, 256 CUs in 8 XCDs; no other target is supported.)
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include "gaiaseg_hip.h"

namespace gs {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;      // CDNA wavefront
constexpr int kNumCUDefault = 256;   // MI355X (8 XCDs x 32 CUs)
// Compute units of the current device (hipDeviceProp::multiProcessorCount), read once; the planners'
// "workgroups per CU" arithmetic uses it.  Without a device (host-only planning queries in the build
// container) the MI355X figure is assumed.
inline int num_cu() {
  static const int v = [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
        prop.multiProcessorCount > 0)
      return prop.multiProcessorCount;
    (void)hipGetLastError();
    return kNumCUDefault;
  }();
  return v;
}
constexpr int kNumXCD = 8;

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Returns 0 or the positive hipError_t of the most recent launch on this thread.
inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? GS_OK : static_cast<int>(e);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Grid size for a memory-bound grid-stride kernel: enough blocks to fill the chip
// (256 CUs x 8 blocks), never more than the work needs.
inline int stream_grid(int64_t work_items, int block) {
  int64_t g = ceil_div(work_items, block);
  const int64_t cap = static_cast<int64_t>(num_cu()) * 8;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return static_cast<int>(g);
}

// Bijective XCD remap (blocks b and b+8 share an XCD under round-robin dispatch): gives every
// XCD a contiguous chunk of the tile sequence so neighbouring tiles hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg / kNumXCD, r = nwg % kNumXCD;
  const int xcd = bid % kNumXCD, idx = bid / kNumXCD;
  const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + idx;
}

// Wave-level sum (64 lanes).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
// Wave-level sum on the VALU's data-parallel-primitive lanes (no LDS traffic, unlike the
// ds_bpermute behind __shfl_down): an inclusive scan inside each row of 16 lanes, then the two
// row broadcasts; lane 63 holds the total, returned as a wave-uniform value.  The association order
// differs from wave_sum's (both are fixed, so results stay run-to-run reproducible).
__device__ __forceinline__ float wave_sum_dpp(float v) {
  auto shr = [](float x, auto ctrl, auto rows) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x),
                                                                 decltype(ctrl)::value,
                                                                 decltype(rows)::value, 0xf, true));
  };
  using std::integral_constant;
  v += shr(v, integral_constant<int, 0x111>{}, integral_constant<int, 0xf>{});   // row_shr:1
  v += shr(v, integral_constant<int, 0x112>{}, integral_constant<int, 0xf>{});   // row_shr:2
  v += shr(v, integral_constant<int, 0x114>{}, integral_constant<int, 0xf>{});   // row_shr:4
  v += shr(v, integral_constant<int, 0x118>{}, integral_constant<int, 0xf>{});   // row_shr:8
  v += shr(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{});   // row_bcast:15
  v += shr(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{});   // row_bcast:31
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Host-side diagnostic counters in double that several host threads add to (the forward thread and
// the autograd thread both launch convolutions): compare-and-swap on the bit pattern.
static inline double flops_load(const double* p) {
  const unsigned long long b = __atomic_load_n(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED);
  return __builtin_bit_cast(double, b);
}
static inline void flops_store(double* p, double v) {
  __atomic_store_n(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED);
}
static inline void flops_add(double* p, double v) {
  unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
  unsigned long long cur = __atomic_load_n(q, __ATOMIC_RELAXED), next;
  do {
    next = __builtin_bit_cast(unsigned long long, __builtin_bit_cast(double, cur) + v);
  } while (!__atomic_compare_exchange_n(q, &cur, next, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
}

}  // namespace gs
