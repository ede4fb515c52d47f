"""ORACLE (test infrastructure — never imported by the product path).

numpy restatement of the reference's train_pipeline transforms applied with GIVEN random decisions
(configs/_dynamic_/models/pspnet_ar50to101v2_gsync.py:60-75): Resize -> RandomCrop -> RandomFlip ->
PhotoMetricDistortion -> Normalize -> Pad -> DefaultFormatBundle, each as its own full-image pass
like the CPU transforms.

PARITY UNPINNED against OpenCV: mmcv executes Resize and the BGR<->HSV conversions with cv2, which is
not in this image (and mmcv / mmseg are absent from /root/reference).  What is restated here is the
documented arithmetic — INTER_LINEAR with half-pixel centres and rounding to uint8, INTER_NEAREST as
floor(dst * in / out), 8-bit HSV with H in [0, 180) — in float32 instead of cv2's 11-bit (resize) and
12-bit (HSV) fixed point, so single pixels may differ from cv2 by one uint8 level.  The order of the
transforms, of the uint8 "convert" roundings of PhotoMetricDistortion, the crop / flip / pad index
arithmetic and the normalisation constants are the reference's."""
import numpy as np

f32 = np.float32


def resize_bilinear_u8(img, rh, rw):
    h, w = img.shape[:2]
    sy, sx = f32(h) / f32(rh), f32(w) / f32(rw)
    fy = (np.arange(rh, dtype=f32) + f32(0.5)) * sy - f32(0.5)
    fx = (np.arange(rw, dtype=f32) + f32(0.5)) * sx - f32(0.5)
    y0 = np.floor(fy).astype(np.int64)
    x0 = np.floor(fx).astype(np.int64)
    wy = (fy - y0.astype(f32)).astype(f32)
    wx = (fx - x0.astype(f32)).astype(f32)
    wy[y0 < 0] = 0
    wx[x0 < 0] = 0
    y0 = np.clip(y0, 0, h - 1)
    x0 = np.clip(x0, 0, w - 1)
    y1 = np.minimum(y0 + 1, h - 1)
    x1 = np.minimum(x0 + 1, w - 1)
    im = img.astype(f32)
    wx_ = wx[None, :, None]
    wy_ = wy[:, None, None]
    top = im[y0][:, x0] + (im[y0][:, x1] - im[y0][:, x0]) * wx_
    bot = im[y1][:, x0] + (im[y1][:, x1] - im[y1][:, x0]) * wx_
    return np.rint(top + (bot - top) * wy_).astype(f32)


def resize_nearest(lab, rh, rw):
    h, w = lab.shape
    sy, sx = f32(h) / f32(rh), f32(w) / f32(rw)
    ys = np.minimum(np.floor(np.arange(rh, dtype=f32) * sy).astype(np.int64), h - 1)
    xs = np.minimum(np.floor(np.arange(rw, dtype=f32) * sx).astype(np.int64), w - 1)
    return lab[ys][:, xs]


def convert(x, alpha=1.0, beta=0.0):
    """PhotoMetricDistortion.convert: float32 * alpha + beta, clip, truncate to uint8."""
    return np.floor(np.clip(x.astype(f32) * f32(alpha) + f32(beta), 0, 255)).astype(f32)


def bgr2hsv(img):
    b, g, r = img[..., 0], img[..., 1], img[..., 2]
    v = np.maximum(b, np.maximum(g, r))
    mn = np.minimum(b, np.minimum(g, r))
    diff = v - mn
    with np.errstate(divide="ignore", invalid="ignore"):
        s = np.where(v > 0, np.rint(diff * f32(255) / v), f32(0)).astype(f32)
        hr = (g - b) / diff
        hg = f32(2) + (b - r) / diff
        hb = f32(4) + (r - g) / diff
    hh = np.where(v == r, hr, np.where(v == g, hg, hb)).astype(f32) * f32(30)
    hh = np.where(hh < 0, hh + f32(180), hh)
    hh = np.where(diff > 0, hh, f32(0)).astype(f32)
    h = np.rint(hh)
    h = np.where(h >= 180, h - 180, h).astype(f32)
    return np.stack([h, s, v], axis=-1)


def hsv2bgr(hsv):
    h, s, v = hsv[..., 0], hsv[..., 1], hsv[..., 2]
    hf = h / f32(30)
    sf = s / f32(255)
    sector = np.floor(hf).astype(np.int64)
    f = (hf - sector.astype(f32)).astype(f32)
    sector = np.where(sector >= 6, sector - 6, sector)
    p = v * (f32(1) - sf)
    q = v * (f32(1) - sf * f)
    t = v * (f32(1) - sf * (f32(1) - f))
    r = np.choose(sector, [v, q, p, p, t, v])
    g = np.choose(sector, [t, v, v, q, p, p])
    b = np.choose(sector, [p, p, t, v, v, q])
    return np.rint(np.stack([b, g, r], axis=-1)).astype(f32)


def train_sample(img_bgr, label, p, crop_size=(512, 1024), mean=(123.675, 116.28, 103.53),
                 std=(58.395, 57.12, 57.375), to_rgb=True, pad_val=0.0, seg_pad_val=255):
    """img_bgr uint8 [H,W,3], label uint8 [H,W], p = the random decisions (draw_train_params).
    Returns (img float32 [3, oh, ow], label int64 [oh, ow])."""
    im = resize_bilinear_u8(img_bgr, p["res_h"], p["res_w"])                       # Resize
    lb = resize_nearest(label, p["res_h"], p["res_w"])
    y, x, ch, cw = p["crop_y"], p["crop_x"], p["crop_h"], p["crop_w"]               # RandomCrop
    im, lb = im[y:y + ch, x:x + cw], lb[y:y + ch, x:x + cw]
    if p["flip"]:                                                                   # RandomFlip
        im, lb = im[:, ::-1], lb[:, ::-1]
    if p.get("pm_enable"):                                                          # PhotoMetric...
        if p["pm_brightness"]:
            im = convert(im, beta=p["pm_delta"])
        if p["pm_contrast"] and p["pm_contrast_first"]:
            im = convert(im, alpha=p["pm_alpha"])
        if p["pm_saturation"]:
            hsv = bgr2hsv(im)
            hsv[..., 1] = convert(hsv[..., 1], alpha=p["pm_sat_alpha"])
            im = hsv2bgr(hsv)
        if p["pm_hue"]:
            hsv = bgr2hsv(im)
            hsv[..., 0] = np.mod(hsv[..., 0] + f32(p["pm_hue_delta"]), f32(180))
            im = hsv2bgr(hsv)
        if p["pm_contrast"] and not p["pm_contrast_first"]:
            im = convert(im, alpha=p["pm_alpha"])
    if to_rgb:                                                                      # Normalize
        im = im[..., ::-1]
    im = (im.astype(f32) - np.asarray(mean, f32)) / np.asarray(std, f32)
    oh, ow = crop_size                                                              # Pad
    out = np.full((oh, ow, 3), pad_val, f32)
    out[:ch, :cw] = im
    lab = np.full((oh, ow), seg_pad_val, np.int64)
    lab[:ch, :cw] = lb
    return np.ascontiguousarray(out.transpose(2, 0, 1)), lab                        # FormatBundle
