"""ORACLE (test infrastructure — never imported by the product path).

CPU restatement of the test-time path of the segmentor: gaiaseg/models/segmentors/
dynamic_distiller.py:252-262 (encode_decode), :416-459 (slide_inference), :461-473
(whole_inference), :475-508 (inference: softmax + flip back), :510-521 (simple_test) and :523-540
(aug_test: mean of the per-augmentation probabilities).  `slide_windows` is pinned bit-exactly by
tests/golden/ref_pure_functions.json, which holds window lists produced by the reference's own loop
(tests/golden/make_ref_pure_fixtures.py)."""
import torch
import torch.nn.functional as F

from . import ops as O


def slide_windows(h_img, w_img, crop_size, stride):
    """Window list of dynamic_distiller.py:423-437 as (y1, y2, x1, x2), row-major: a regular grid
    of `stride`, the last window of every row / column snapped back inside the image."""
    h_crop, w_crop = crop_size
    h_stride, w_stride = stride
    h_grids = max(h_img - h_crop + h_stride - 1, 0) // h_stride + 1
    w_grids = max(w_img - w_crop + w_stride - 1, 0) // w_stride + 1
    wins = []
    for hi in range(h_grids):
        for wi in range(w_grids):
            y2 = min(hi * h_stride + h_crop, h_img)
            x2 = min(wi * w_stride + w_crop, w_img)
            wins.append((max(y2 - h_crop, 0), y2, max(x2 - w_crop, 0), x2))
    return wins


def slide_inference(model, img, ori_shape, crop_size, stride, rescale):
    n, _, h_img, w_img = img.shape
    align = model.decode_head.align_corners
    preds = None
    count = img.new_zeros((n, 1, h_img, w_img))
    for y1, y2, x1, x2 in slide_windows(h_img, w_img, crop_size, stride):
        logit = model.encode_decode(img[:, :, y1:y2, x1:x2])
        if preds is None:
            preds = img.new_zeros((n, logit.shape[1], h_img, w_img))
        preds = preds + F.pad(logit, (x1, w_img - x2, y1, h_img - y2))
        count[:, :, y1:y2, x1:x2] += 1
    assert int((count == 0).sum()) == 0
    preds = preds / count
    if rescale:
        preds = O.resize(preds, size=tuple(ori_shape[:2]), mode="bilinear", align_corners=align)
    return preds


def whole_inference(model, img, ori_shape, rescale):
    logit = model.encode_decode(img)
    if rescale:
        logit = O.resize(logit, size=tuple(ori_shape[:2]), mode="bilinear",
                         align_corners=model.decode_head.align_corners)
    return logit


def inference(model, img, img_meta, test_cfg, rescale=True):
    """Probabilities [N, C, H, W] of one (possibly flipped) view."""
    mode = test_cfg["mode"]
    assert mode in ("slide", "whole")
    if mode == "slide":
        logit = slide_inference(model, img, img_meta["ori_shape"], test_cfg["crop_size"],
                                test_cfg["stride"], rescale)
    else:
        logit = whole_inference(model, img, img_meta["ori_shape"], rescale)
    out = F.softmax(logit, dim=1)
    if img_meta.get("flip", False):
        d = img_meta["flip_direction"]
        assert d in ("horizontal", "vertical")
        out = out.flip(dims=(3,)) if d == "horizontal" else out.flip(dims=(2,))
    return out


def simple_test(model, img, img_meta, test_cfg, rescale=True):
    return inference(model, img, img_meta, test_cfg, rescale).argmax(dim=1)


def aug_test(model, imgs, img_metas, test_cfg, rescale=True):
    assert rescale
    prob = inference(model, imgs[0], img_metas[0], test_cfg, rescale)
    for i in range(1, len(imgs)):
        prob = prob + inference(model, imgs[i], img_metas[i], test_cfg, rescale)
    return (prob / len(imgs)).argmax(dim=1)
