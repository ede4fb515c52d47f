// Training input pipeline as ONE gather kernel per sample.
//
// Replaces the per-sample CPU transforms of the reference's train_pipeline
// (configs/_dynamic_/models/pspnet_ar50to101v2_gsync.py:60-75, executed by mmseg / mmcv on
// DataLoader workers): Resize(ratio_range) -> RandomCrop -> RandomFlip -> PhotoMetricDistortion ->
// Normalize(to_rgb) -> Pad(0 / 255) -> DefaultFormatBundle.  The random decisions are drawn on the
// host (gaia_seg_amd/datasets/gpu_pipeline.py) exactly as the CPU transforms draw them; this kernel
// applies them.  Every output pixel of the fixed-size crop gathers its source: the flip and the crop
// offset move the coordinate, the resize is a bilinear (image) / nearest (label) fetch from the
// ORIGINAL uint8 image — the resized image (up to 2048x4096x3) never exists —, the photometric
// distortion and the normalisation run on the fetched pixel, and positions outside the cropped
// region get the pad values.  HBM traffic per sample: the touched part of the source image once
// (uint8) + 3 x 4 B + 8 B per output pixel.
//
// Arithmetic contract (the oracle's numpy restatement follows the same steps; mmcv executes them
// with OpenCV, which is not in the image — see oracle/pipeline.py for what that leaves unpinned):
//   resize   image: half-pixel centres, src = (dst + 0.5) * (in / out) - 0.5, edge-clamped,
//            bilinear in float, rounded to the nearest integer (uint8 like cv2.INTER_LINEAR);
//            label: nearest, src = min(floor(dst * in / out), in - 1)       (cv2.INTER_NEAREST)
//   photometric distortion (mmseg PhotoMetricDistortion.__call__ order):
//            brightness (+delta), contrast first (mode 1), saturation and hue in HSV, contrast last
//            (mode 0); after every step "convert" = clip to [0, 255] and truncate to uint8
//   normalize (x - mean) / std on RGB order when to_rgb
#include "common.h"

// no fused multiply-add here: every step is specified as separately rounded fp32 operations (what
// numpy does), so that the uint8 roundings fall on the same side as in the oracle
#pragma clang fp contract(off)

namespace gs {

__device__ __forceinline__ float u8_convert(float v) {   // mmseg PhotoMetricDistortion.convert
  v = fminf(fmaxf(v, 0.f), 255.f);
  return floorf(v);
}

// 8-bit HSV as OpenCV defines it for uint8 images: H in [0, 180), S, V in [0, 255]
__device__ __forceinline__ void bgr2hsv_u8(float b, float g, float r, float& h, float& s, float& v) {
  v = fmaxf(b, fmaxf(g, r));
  const float mn = fminf(b, fminf(g, r));
  const float diff = v - mn;
  s = v > 0.f ? rintf(diff * 255.f / v) : 0.f;
  float hh = 0.f;
  if (diff > 0.f) {
    if (v == r) hh = (g - b) / diff;
    else if (v == g) hh = 2.f + (b - r) / diff;
    else hh = 4.f + (r - g) / diff;
    hh *= 30.f;                       // degrees / 2
    if (hh < 0.f) hh += 180.f;
  }
  h = rintf(hh);
  if (h >= 180.f) h -= 180.f;
}

__device__ __forceinline__ void hsv2bgr_u8(float h, float s, float v, float& b, float& g, float& r) {
  const float hf = h / 30.f;          // sector 0..6
  const float sf = s / 255.f;
  int sector = (int)floorf(hf);
  const float f = hf - (float)sector;
  if (sector >= 6) sector -= 6;
  const float p = v * (1.f - sf), q = v * (1.f - sf * f), t = v * (1.f - sf * (1.f - f));
  float rr, gg, bb;
  switch (sector) {
    case 0: rr = v; gg = t; bb = p; break;
    case 1: rr = q; gg = v; bb = p; break;
    case 2: rr = p; gg = v; bb = t; break;
    case 3: rr = p; gg = q; bb = v; break;
    case 4: rr = t; gg = p; bb = v; break;
    default: rr = v; gg = p; bb = q; break;
  }
  b = rintf(bb); g = rintf(gg); r = rintf(rr);
}

__global__ __launch_bounds__(256) void seg_augment_kernel(const gs_augment_desc d,
                                                          const uint8_t* __restrict__ img,
                                                          const uint8_t* __restrict__ label,
                                                          float* __restrict__ out_img,
                                                          int64_t* __restrict__ out_label) {
  const long plane = (long)d.out_h * d.out_w;
  const float sy = (float)d.src_h / (float)d.res_h, sx = (float)d.src_w / (float)d.res_w;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < plane;
       i += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % d.out_w), oy = (int)(i / d.out_w);
    if (oy >= d.crop_h || ox >= d.crop_w) {            // Pad(size, pad_val, seg_pad_val)
      out_img[i] = d.pad_val;
      out_img[plane + i] = d.pad_val;
      out_img[2 * plane + i] = d.pad_val;
      if (out_label) out_label[i] = d.seg_pad_val;
      continue;
    }
    const int cx = d.flip ? d.crop_w - 1 - ox : ox;    // RandomFlip (horizontal) of the crop
    const int ry = oy + d.crop_y, rx = cx + d.crop_x;  // position in the resized image
    // ---- label: nearest ----
    if (out_label) {
      const int ly = min((int)floorf((float)ry * sy), d.src_h - 1);
      const int lx = min((int)floorf((float)rx * sx), d.src_w - 1);
      out_label[i] = label[(long)ly * d.src_w + lx];
    }
    // ---- image: bilinear, rounded to uint8 ----
    float fy = ((float)ry + 0.5f) * sy - 0.5f, fx = ((float)rx + 0.5f) * sx - 0.5f;
    int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
    float wy = fy - (float)y0, wx = fx - (float)x0;
    if (y0 < 0) { y0 = 0; wy = 0.f; }
    if (x0 < 0) { x0 = 0; wx = 0.f; }
    int y1 = y0 + 1, x1 = x0 + 1;
    if (y1 > d.src_h - 1) { y1 = d.src_h - 1; if (y0 > d.src_h - 1) { y0 = d.src_h - 1; } }
    if (x1 > d.src_w - 1) { x1 = d.src_w - 1; if (x0 > d.src_w - 1) { x0 = d.src_w - 1; } }
    const uint8_t* p00 = img + ((long)y0 * d.src_w + x0) * 3;
    const uint8_t* p01 = img + ((long)y0 * d.src_w + x1) * 3;
    const uint8_t* p10 = img + ((long)y1 * d.src_w + x0) * 3;
    const uint8_t* p11 = img + ((long)y1 * d.src_w + x1) * 3;
    float c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float top = (float)p00[k] + ((float)p01[k] - (float)p00[k]) * wx;
      const float bot = (float)p10[k] + ((float)p11[k] - (float)p10[k]) * wx;
      c[k] = rintf(top + (bot - top) * wy);
    }
    // channel order of the source: BGR (cv2.imread) unless src_is_rgb
    float b = d.src_is_rgb ? c[2] : c[0], g = c[1], r = d.src_is_rgb ? c[0] : c[2];
    // ---- PhotoMetricDistortion ----
    if (d.pm_enable) {
      if (d.pm_brightness) { b = u8_convert(b + d.pm_delta); g = u8_convert(g + d.pm_delta); r = u8_convert(r + d.pm_delta); }
      if (d.pm_contrast && d.pm_contrast_first) {
        b = u8_convert(b * d.pm_alpha); g = u8_convert(g * d.pm_alpha); r = u8_convert(r * d.pm_alpha);
      }
      if (d.pm_saturation) {     // each of the two is its own uint8 BGR -> HSV -> BGR round trip
        float h, s, v;
        bgr2hsv_u8(b, g, r, h, s, v);
        s = u8_convert(s * d.pm_sat_alpha);
        hsv2bgr_u8(h, s, v, b, g, r);
      }
      if (d.pm_hue) {
        float h, s, v;
        bgr2hsv_u8(b, g, r, h, s, v);
        h = h + (float)d.pm_hue_delta;      // integer delta, uint8 H modulo 180
        h = h - 180.f * floorf(h / 180.f);
        hsv2bgr_u8(h, s, v, b, g, r);
      }
      if (d.pm_contrast && !d.pm_contrast_first) {
        b = u8_convert(b * d.pm_alpha); g = u8_convert(g * d.pm_alpha); r = u8_convert(r * d.pm_alpha);
      }
    }
    // ---- Normalize (to_rgb -> planes R, G, B) + DefaultFormatBundle (CHW float) ----
    const float ch0 = d.to_rgb ? r : b, ch2 = d.to_rgb ? b : r;
    out_img[i] = (ch0 - d.mean[0]) / d.std[0];
    out_img[plane + i] = (g - d.mean[1]) / d.std[1];
    out_img[2 * plane + i] = (ch2 - d.mean[2]) / d.std[2];
  }
}

}  // namespace gs

using namespace gs;

extern "C" int gs_seg_augment(const gs_augment_desc* d, const uint8_t* img, const uint8_t* label,
                              float* out_img, int64_t* out_label, void* stream) {
  if (!d || !img || !out_img) return GS_E_NULL;
  if (out_label && !label) return GS_E_NULL;
  if (d->src_h <= 0 || d->src_w <= 0 || d->res_h <= 0 || d->res_w <= 0 || d->out_h <= 0 ||
      d->out_w <= 0)
    return GS_E_BADARG;
  if (d->crop_h < 0 || d->crop_w < 0 || d->crop_h > d->out_h || d->crop_w > d->out_w)
    return GS_E_BADARG;
  if (d->crop_y < 0 || d->crop_x < 0 || d->crop_y + d->crop_h > d->res_h ||
      d->crop_x + d->crop_w > d->res_w)
    return GS_E_BADARG;
  for (int k = 0; k < 3; ++k)
    if (!(d->std[k] > 0.f)) return GS_E_BADARG;
  const long plane = (long)d->out_h * d->out_w;
  hipLaunchKernelGGL(seg_augment_kernel, dim3(stream_grid(plane, 256)), dim3(256), 0,
                     as_stream(stream), *d, img, label, out_img, out_label);
  return launch_status();
}
