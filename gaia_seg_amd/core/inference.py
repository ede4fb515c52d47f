"""Test-time driver of the segmentor: whole-image and sliding-window inference through ONE fused
epilogue kernel (``gs_slide_fuse``, csrc/inference.hip).

Behaviour of the reference (mmseg EncoderDecoder, restated at gaiaseg/models/segmentors/
dynamic_distiller.py:416-540): per window run encode_decode, add the up-sampled logits into a
full-size ``preds`` tensor, count, divide, resize to ``ori_shape``, softmax, flip back, argmax;
``aug_test`` averages the probabilities of the augmented views.  What this driver does instead:

* the window list is a product of row and column origins (``window_axes``) — pinned against the
  reference's own loop by tests/golden/ref_pure_functions.json;
* all windows go through the network as ONE batch (eval-mode BatchNorm is batch-independent, so the
  results equal the reference's one-window-at-a-time loop) in passes of at most ``max_pass_pixels``;
* the head's low-resolution logits of every window stay resident and the kernel gathers them per
  output pixel: accumulate / count / divide / rescale / softmax / flip / argmax in one launch, writing
  only the label map (or the probabilities when the caller wants them).
"""
import ctypes

import torch

from ..hip import lib as _lib
from ..hip.runtime import current_stream_ptr, require_gpu_tensor


def _axis_origins(length, crop, stride):
    """Origins of the windows along one axis: a regular grid of ``stride`` whose last window is
    snapped back inside the image (dynamic_distiller.py:423-437); one window when crop >= length."""
    n = max(length - crop + stride - 1, 0) // stride + 1
    out = []
    for i in range(n):
        end = min(i * stride + crop, length)
        out.append(max(end - crop, 0))
    return out


def window_axes(h_img, w_img, crop_size, stride):
    """(row origins, column origins, window height, window width) of slide inference."""
    h_crop, w_crop = int(crop_size[0]), int(crop_size[1])
    ys = _axis_origins(h_img, h_crop, int(stride[0]))
    xs = _axis_origins(w_img, w_crop, int(stride[1]))
    return ys, xs, min(h_crop, h_img), min(w_crop, w_img)


def slide_windows(h_img, w_img, crop_size, stride):
    """The reference's window list [(y1, y2, x1, x2)], row-major."""
    ys, xs, hc, wc = window_axes(h_img, w_img, crop_size, stride)
    return [(y, y + hc, x, x + wc) for y in ys for x in xs]


def _padded_nhwc(logits):
    """The [B, h, w, ld] storage behind a head's logical NCHW output (channels-last, class stride
    padded to a float4 multiple), or a padded copy when the tensor is laid out otherwise."""
    b, c, h, w = logits.shape
    nhwc = logits.detach().permute(0, 2, 3, 1)
    ld = nhwc.stride(2)
    ok = (nhwc.stride(3) == 1 and ld % 4 == 0 and ld >= c and nhwc.stride(1) == w * ld
          and nhwc.stride(0) == h * w * ld and nhwc.data_ptr() % 16 == 0
          and nhwc.untyped_storage().nbytes() // 4 - nhwc.storage_offset() >= b * h * w * ld)
    if ok:
        return nhwc.as_strided((b, h, w, ld), nhwc.stride())
    ld = (c + 3) // 4 * 4
    full = torch.zeros((b, h, w, ld), dtype=torch.float32, device=logits.device)
    full[..., :c].copy_(nhwc)
    return full


class FusedInference:
    """Runs ``logits_fn(batch) -> [B, C, h, w]`` over the windows of an image and fuses the epilogue."""

    def __init__(self, num_classes, align_corners=False, max_pass_pixels=16 << 20):
        self.num_classes = num_classes
        self.align_corners = bool(align_corners)
        self.max_pass_pixels = max_pass_pixels

    def window_logits(self, logits_fn, img, ys, xs, hc, wc):
        """Padded low-resolution logits [ny*nx*N, hl, wl, ld] of all windows, window-major."""
        n = img.shape[0]
        wins = [(y, x) for y in ys for x in xs]
        if len(wins) == 1 and (hc, wc) == tuple(img.shape[2:]):
            return _padded_nhwc(logits_fn(img))
        per_pass = max(1, self.max_pass_pixels // max(1, n * hc * wc))
        chunks = []
        for i in range(0, len(wins), per_pass):
            crops = torch.cat([img[:, :, y:y + hc, x:x + wc] for y, x in wins[i:i + per_pass]])
            chunks.append(_padded_nhwc(logits_fn(crops)))
        return chunks[0] if len(chunks) == 1 else torch.cat(chunks)

    def __call__(self, logits_fn, img, mode="whole", crop_size=None, stride=None, out_size=None,
                 flip=None, probs_in=None, want_probs=False, want_labels=True):
        """Returns (labels int64 [N, Ho, Wo] or None, probabilities fp32 [N, C, Ho, Wo] or None).
        ``probs_in`` (same shape as the probabilities) is added before the argmax / the store."""
        require_gpu_tensor(img, "image")
        n, _, h_img, w_img = img.shape
        if mode == "slide":
            ys, xs, hc, wc = window_axes(h_img, w_img, crop_size, stride)
        elif mode == "whole":
            ys, xs, hc, wc = [0], [0], h_img, w_img
        else:
            raise ValueError("test_cfg.mode must be 'slide' or 'whole', got %r" % (mode,))
        low = self.window_logits(logits_fn, img, ys, xs, hc, wc)
        if low.shape[0] != len(ys) * len(xs) * n:
            raise RuntimeError("decode head returned %d maps for %d windows x %d images"
                               % (low.shape[0], len(ys) * len(xs), n))
        ho, wo = (h_img, w_img) if out_size is None else (int(out_size[0]), int(out_size[1]))
        d = _lib.SlideDesc()
        d.N, d.C, d.ld = n, self.num_classes, low.shape[3]
        d.hl, d.wl, d.hc, d.wc = low.shape[1], low.shape[2], hc, wc
        d.H, d.W, d.Ho, d.Wo = h_img, w_img, ho, wo
        d.ny, d.nx = len(ys), len(xs)
        d.align_corners = 1 if self.align_corners else 0
        d.flip = {None: 0, False: 0, "horizontal": 1, "vertical": 2}[flip]
        d.reserved = 0
        dev = img.device
        labels = torch.empty((n, ho, wo), dtype=torch.int64, device=dev) if want_labels else None
        probs = None
        if want_probs:
            probs = torch.empty((n, self.num_classes, ho, wo), dtype=torch.float32, device=dev)
        if probs_in is not None:
            if tuple(probs_in.shape) != (n, self.num_classes, ho, wo) or not probs_in.is_contiguous():
                raise ValueError("probs_in must be a contiguous [N, C, Ho, Wo] tensor")
        ay = (ctypes.c_int32 * len(ys))(*ys)
        ax = (ctypes.c_int32 * len(xs))(*xs)
        _lib.check(_lib.load().gs_slide_fuse(
            ctypes.byref(d), ay, ax, low.data_ptr(),
            probs_in.data_ptr() if probs_in is not None else None,
            probs.data_ptr() if probs is not None else None,
            labels.data_ptr() if labels is not None else None, current_stream_ptr()), "gs_slide_fuse")
        return labels, probs
