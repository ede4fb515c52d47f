"""GPU input pipeline for supernet training.

The reference trains on ``train_pipeline`` of configs/_dynamic_/models/pspnet_ar50to101v2_gsync.py:
60-75 — LoadImageFromFile, LoadAnnotations, Resize(img_scale=(2048, 1024), ratio_range=(0.5, 2.0)),
RandomCrop(crop_size, cat_max_ratio=0.75), RandomFlip(0.5), PhotoMetricDistortion, Normalize, Pad,
DefaultFormatBundle — run by mmseg / mmcv (absent) on DataLoader worker processes, two per GPU
(cfg :118), on the CPU.  At 145 images/s per MI355X that is ~1 200 augmented 1024x512 crops per
second for an 8-GPU node: here the transforms run on the GPU instead.  The host only draws the random
decisions — with numpy's RandomState in the order the CPU transforms consume them
(``draw_train_params``) — and one gather kernel per sample (``gs_seg_augment``,
csrc/augment.hip) produces the normalised fp32 crop and the int64 label map straight from the
decoded uint8 image on the device.

Input: an iterable of (img uint8 [H, W, 3] BGR as cv2.imread gives it, label uint8 [H, W]) device
or host tensors; decoding (PNG / JPEG) is outside this stage.
"""
import ctypes

import numpy as np
import torch

from ..hip import lib as _lib
from ..hip.runtime import current_stream_ptr


def rescale_size(h, w, scale):
    """mmcv.rescale_size for a (long, short) scale tuple: keep the aspect ratio, fit both edges."""
    max_long, max_short = max(scale), min(scale)
    f = min(max_long / max(h, w), max_short / min(h, w))
    return int(h * f + 0.5), int(w * f + 0.5)


def draw_train_params(rng, h, w, cfg, label=None):
    """The random decisions of one sample, drawn in the order mmseg's transforms consume
    ``np.random``: Resize.random_sample_ratio, RandomCrop.get_crop_bbox (+ up to 10 redraws for
    cat_max_ratio), RandomFlip, PhotoMetricDistortion (brightness, mode, contrast / saturation / hue).

    ``label``: host or device uint8 [h, w] tensor, needed only when cat_max_ratio < 1."""
    p = {}
    lo, hi = cfg.get("ratio_range", (0.5, 2.0))
    ratio = rng.random_sample() * (hi - lo) + lo
    img_scale = cfg.get("img_scale", (2048, 1024))
    scale = (int(img_scale[0] * ratio), int(img_scale[1] * ratio))
    rh, rw = rescale_size(h, w, scale)
    p["res_h"], p["res_w"] = rh, rw
    ch, cw = cfg.get("crop_size", (512, 1024))

    def bbox():
        mh, mw = max(rh - ch, 0), max(rw - cw, 0)
        oy = rng.randint(0, mh + 1)
        ox = rng.randint(0, mw + 1)
        return oy, ox, min(ch, rh - oy), min(cw, rw - ox)
    box = bbox()
    cat_max = cfg.get("cat_max_ratio", 1.0)
    if cat_max < 1.0 and label is not None:
        ignore = cfg.get("ignore_index", 255)
        for _ in range(10):
            if _dominant_fraction(label, h, w, rh, rw, box, ignore) < cat_max:
                break
            box = bbox()
    p["crop_y"], p["crop_x"], p["crop_h"], p["crop_w"] = box
    p["flip"] = bool(rng.rand() < cfg.get("flip_ratio", 0.5))
    pm = cfg.get("photometric", True)
    p["pm_enable"] = bool(pm)
    if pm:
        bd = cfg.get("brightness_delta", 32)
        c_lo, c_hi = cfg.get("contrast_range", (0.5, 1.5))
        s_lo, s_hi = cfg.get("saturation_range", (0.5, 1.5))
        hd = cfg.get("hue_delta", 18)
        p["pm_brightness"] = bool(rng.randint(2))
        p["pm_delta"] = float(rng.uniform(-bd, bd)) if p["pm_brightness"] else 0.0
        mode = rng.randint(2)
        p["pm_contrast_first"] = mode == 1
        p["pm_contrast"], p["pm_alpha"] = False, 1.0
        if mode == 1:
            p["pm_contrast"] = bool(rng.randint(2))
            p["pm_alpha"] = float(rng.uniform(c_lo, c_hi)) if p["pm_contrast"] else 1.0
        p["pm_saturation"] = bool(rng.randint(2))
        p["pm_sat_alpha"] = float(rng.uniform(s_lo, s_hi)) if p["pm_saturation"] else 1.0
        p["pm_hue"] = bool(rng.randint(2))
        p["pm_hue_delta"] = int(rng.randint(-hd, hd)) if p["pm_hue"] else 0
        if mode == 0:
            p["pm_contrast"] = bool(rng.randint(2))
            p["pm_alpha"] = float(rng.uniform(c_lo, c_hi)) if p["pm_contrast"] else 1.0
    return p


def _dominant_fraction(label, h, w, rh, rw, box, ignore):
    """Share of the most frequent non-ignored class inside the crop window of the nearest-resized
    label map (RandomCrop's cat_max_ratio test); 1.0 when fewer than two classes are present."""
    oy, ox, ch, cw = box
    dev = label.device if isinstance(label, torch.Tensor) else "cpu"
    lab = torch.as_tensor(label)
    ys = torch.clamp((torch.arange(oy, oy + ch, device=dev, dtype=torch.float32) * (h / rh)).floor().long(), max=h - 1)
    xs = torch.clamp((torch.arange(ox, ox + cw, device=dev, dtype=torch.float32) * (w / rw)).floor().long(), max=w - 1)
    win = lab[ys][:, xs].reshape(-1).long()
    cnt = torch.bincount(win[win != ignore], minlength=1)
    cnt = cnt[cnt > 0]
    if cnt.numel() <= 1:
        return 1.0
    return float(cnt.max()) / float(cnt.sum())


class GpuTrainPipeline:
    """Batches of dict(img fp32 [N,3,H,W], gt_semantic_seg int64 [N,1,H,W], img_metas) from decoded
    uint8 samples, all transforms on the device."""

    def __init__(self, crop_size=(512, 1024), img_scale=(2048, 1024), ratio_range=(0.5, 2.0),
                 cat_max_ratio=0.75, flip_ratio=0.5, photometric=True,
                 mean=(123.675, 116.28, 103.53), std=(58.395, 57.12, 57.375), to_rgb=True,
                 pad_val=0.0, seg_pad_val=255, ignore_index=255, seed=None, device="cuda",
                 src_is_rgb=False):
        self.cfg = dict(crop_size=tuple(crop_size), img_scale=tuple(img_scale),
                        ratio_range=tuple(ratio_range), cat_max_ratio=cat_max_ratio,
                        flip_ratio=flip_ratio, photometric=photometric, ignore_index=ignore_index)
        self.mean, self.std, self.to_rgb = tuple(mean), tuple(std), bool(to_rgb)
        self.pad_val, self.seg_pad_val = float(pad_val), int(seg_pad_val)
        self.rng = np.random.RandomState(seed)
        self.device = torch.device(device)
        self.src_is_rgb = bool(src_is_rgb)   # decoded by Pillow (RGB) instead of cv2.imread (BGR)

    def descriptor(self, h, w, p, src_is_rgb=False):
        d = _lib.AugmentDesc()
        d.src_h, d.src_w, d.src_is_rgb = h, w, 1 if src_is_rgb else 0
        d.res_h, d.res_w = p["res_h"], p["res_w"]
        d.crop_y, d.crop_x, d.crop_h, d.crop_w = p["crop_y"], p["crop_x"], p["crop_h"], p["crop_w"]
        d.out_h, d.out_w = self.cfg["crop_size"]
        d.flip = 1 if p["flip"] else 0
        d.pm_enable = 1 if p.get("pm_enable") else 0
        for k in ("pm_brightness", "pm_contrast", "pm_contrast_first", "pm_saturation", "pm_hue"):
            setattr(d, k, 1 if p.get(k) else 0)
        d.pm_delta, d.pm_alpha = p.get("pm_delta", 0.0), p.get("pm_alpha", 1.0)
        d.pm_sat_alpha, d.pm_hue_delta = p.get("pm_sat_alpha", 1.0), p.get("pm_hue_delta", 0)
        d.to_rgb = 1 if self.to_rgb else 0
        for k in range(3):
            d.mean[k], d.std[k] = self.mean[k], self.std[k]
        d.pad_val, d.seg_pad_val = self.pad_val, self.seg_pad_val
        return d

    def sample(self, img, label, out_img, out_label, params=None, src_is_rgb=None):
        """Augment one decoded sample into row slices of the batch tensors; returns the params."""
        if src_is_rgb is None:
            src_is_rgb = self.src_is_rgb
        img = img.to(self.device, non_blocking=True).contiguous()
        label = label.to(self.device, non_blocking=True).contiguous()
        if img.dtype != torch.uint8 or label.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
            raise TypeError("expected uint8 img [H, W, 3] and uint8 label [H, W]")
        h, w = int(img.shape[0]), int(img.shape[1])
        p = params if params is not None else draw_train_params(self.rng, h, w, self.cfg, label)
        d = self.descriptor(h, w, p, src_is_rgb)
        _lib.check(_lib.load().gs_seg_augment(ctypes.byref(d), img.data_ptr(), label.data_ptr(),
                                              out_img.data_ptr(), out_label.data_ptr(),
                                              current_stream_ptr()), "gs_seg_augment")
        return p

    def batch(self, samples, params=None):
        """samples: list of (img, label[, filename]); returns the train_step batch dict."""
        n = len(samples)
        oh, ow = self.cfg["crop_size"]
        imgs = torch.empty((n, 3, oh, ow), dtype=torch.float32, device=self.device)
        gts = torch.empty((n, 1, oh, ow), dtype=torch.int64, device=self.device)
        metas = []
        for i, s in enumerate(samples):
            p = self.sample(s[0], s[1], imgs[i], gts[i, 0], None if params is None else params[i])
            metas.append(dict(ori_shape=tuple(s[0].shape), img_shape=(p["crop_h"], p["crop_w"], 3),
                              pad_shape=(oh, ow, 3), flip=p["flip"], flip_direction="horizontal",
                              scale_factor=p["res_h"] / s[0].shape[0],
                              filename=s[2] if len(s) > 2 else "sample_%d" % i))
        return dict(img=imgs, img_metas=metas, gt_semantic_seg=gts)

    def test_batch(self, samples, img_scale):
        """The single-scale test pipeline (LoadImageFromFile, Resize(keep_ratio) to ``img_scale``,
        Normalize; configs/_dynamic_/models/pspnet_ar50to101v2_gsync.py:76-93) for samples of ONE
        size: img fp32 [N, 3, h', w'] through the same kernel (no crop, flip or distortion), and the
        label maps at their ORIGINAL size (predictions are rescaled to ``ori_shape`` before they are
        compared)."""
        h, w = int(samples[0][0].shape[0]), int(samples[0][0].shape[1])
        for s in samples:
            if tuple(s[0].shape[:2]) != (h, w):
                raise ValueError("test_batch needs samples of one size (got %s and %s)"
                                 % ((h, w), tuple(s[0].shape[:2])))
        rh, rw = rescale_size(h, w, img_scale) if img_scale is not None else (h, w)
        n = len(samples)
        imgs = torch.empty((n, 3, rh, rw), dtype=torch.float32, device=self.device)
        scratch = torch.empty((rh, rw), dtype=torch.int64, device=self.device)
        gts, metas = [], []
        p = dict(res_h=rh, res_w=rw, crop_y=0, crop_x=0, crop_h=rh, crop_w=rw, flip=False)
        for i, s in enumerate(samples):
            img = s[0].to(self.device, non_blocking=True).contiguous()
            label = s[1]
            dummy = label if label is not None else torch.zeros((h, w), dtype=torch.uint8)
            dummy = dummy.to(self.device, non_blocking=True).contiguous()
            d = self.descriptor(h, w, p, self.src_is_rgb)
            d.out_h, d.out_w = rh, rw
            _lib.check(_lib.load().gs_seg_augment(ctypes.byref(d), img.data_ptr(), dummy.data_ptr(),
                                                  imgs[i].data_ptr(), scratch.data_ptr(),
                                                  current_stream_ptr()), "gs_seg_augment")
            if label is not None:
                gts.append(dummy.to(torch.int64))
            metas.append(dict(ori_shape=(h, w, 3), img_shape=(rh, rw, 3), pad_shape=(rh, rw, 3),
                              flip=False, flip_direction="horizontal", scale_factor=rh / h,
                              filename=s[2] if len(s) > 2 else "sample_%d" % i))
        out = dict(img=imgs, img_metas=metas)
        if gts:
            out["gt_semantic_seg"] = torch.stack(gts).unsqueeze(1)
        return out
