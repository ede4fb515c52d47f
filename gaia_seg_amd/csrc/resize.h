// Bilinear source-index arithmetic shared by the resize and the fused cross-entropy kernels.
// Follows ATen's area_pixel_compute_scale / area_pixel_compute_source_index in fp32, so the
// interpolation weights equal the CPU reference's (F.interpolate == mmseg.ops.resize,
// gaiaseg/models/decode_heads/dynamic_fcn_head.py:141-145) bit for bit.
#pragma once
#include <hip/hip_runtime.h>

namespace gs {

struct Lerp {
  int i0, i1;    // the two source indices (i1 == i0 at the far border)
  float l0, l1;  // their weights
};

__host__ __device__ __forceinline__ float resize_scale(int in, int out, int align) {
  if (align) return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  return (float)in / (float)out;
}

__device__ __forceinline__ Lerp lerp_coord(int dst, float scale, int in, int align) {
  float src;
  if (align) {
    src = scale * (float)dst;
  } else {
    src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
  }
  Lerp r;
  r.i0 = (int)src;
  if (r.i0 > in - 1) r.i0 = in - 1;
  r.i1 = r.i0 + ((r.i0 < in - 1) ? 1 : 0);
  r.l1 = src - (float)r.i0;
  r.l0 = 1.f - r.l1;
  return r;
}

// Conservative range [lo, hi] of destination indices whose footprint can touch source index i
// (the exact weights are recomputed with lerp_coord, so a wider range is harmless).
__device__ __forceinline__ void dst_range(int i, float scale, int out, int& lo, int& hi) {
  if (scale <= 0.f) { lo = 0; hi = out - 1; return; }
  const float inv = 1.f / scale;
  lo = (int)floorf(((float)i - 1.f) * inv - 1.5f);
  hi = (int)ceilf(((float)i + 1.5f) * inv + 0.5f);
  if (lo < 0) lo = 0;
  if (hi > out - 1) hi = out - 1;
}

// weight with which destination coordinate `l` contributes to source index i
__device__ __forceinline__ float adj_weight(const Lerp& l, int i) {
  return (l.i0 == i ? l.l0 : 0.f) + (l.i1 == i ? l.l1 : 0.f);
}

}  // namespace gs
