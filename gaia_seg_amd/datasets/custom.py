"""File-backed segmentation datasets with mmseg's ``CustomDataset`` contract, and the translation of
an mmseg transform list into the settings of the GPU input pipeline.

The reference's configs name their data by ``type`` strings that mmseg resolves
(configs/_dynamic_/models/pspnet_ar50to101v2_gsync.py:95-135: ``data.train = [dict(
type='CityscapesDataset19', data_root=..., img_dir='leftImg8bit/train', ann_dir='gtFine/train',
pipeline=train_pipeline)]``; ``CityscapesDataset19`` itself is defined nowhere in the tree, SURVEY.md
§8d — it is the 19-class Cityscapes of mmseg).  What a dataset has to provide for this path is small:
the sorted list of (image, label map) files and the decoded uint8 arrays; every transform after the
decode runs on the GPU (gpu_pipeline.py).

Contract restated from mmseg's CustomDataset ([3P], SURVEY.md App. A style recollection):
  * ``img_dir`` / ``ann_dir`` are joined to ``data_root`` unless absolute;
  * without ``split`` every file under ``img_dir`` (recursively) that ends with ``img_suffix`` is a
    sample, its label map is ``ann_dir / name.replace(img_suffix, seg_map_suffix)``; with ``split`` the
    file lists one sample name (no suffix) per line; samples are ordered by file name;
  * ``reduce_zero_label``: label 0 becomes 255 and every other label drops by one;
  * images are what ``cv2.imread`` gives (uint8, three channels), label maps are read unchanged.
Decoding uses Pillow (the only image library in the image): RGB order, which the augment kernel takes
through its ``src_is_rgb`` switch instead of a channel swap on the host.
"""
import os

import numpy as np
import torch

from ..core.registry import Registry, build_from_cfg

DATASETS = Registry("dataset")


def _scan(root, suffix):
    out = []
    for base, _dirs, files in os.walk(root, followlinks=True):
        rel = os.path.relpath(base, root)
        for f in files:
            if f.endswith(suffix):
                out.append(f if rel == "." else os.path.join(rel, f))
    return sorted(out)


@DATASETS.register_module()
class CustomDataset:
    CLASSES = None
    PALETTE = None

    def __init__(self, pipeline=None, img_dir=None, img_suffix=".jpg", ann_dir=None,
                 seg_map_suffix=".png", split=None, data_root=None, test_mode=False,
                 ignore_index=255, reduce_zero_label=False, classes=None, palette=None):
        if img_dir is None:
            raise ValueError("img_dir is required")
        self.pipeline = list(pipeline or [])
        self.img_suffix, self.seg_map_suffix = img_suffix, seg_map_suffix
        self.test_mode = bool(test_mode)
        self.ignore_index = ignore_index
        self.reduce_zero_label = bool(reduce_zero_label)
        if classes is not None:
            self.CLASSES = tuple(classes)
        if palette is not None:
            self.PALETTE = [tuple(p) for p in palette]
        if data_root is not None:
            if not os.path.isabs(img_dir):
                img_dir = os.path.join(data_root, img_dir)
            if ann_dir is not None and not os.path.isabs(ann_dir):
                ann_dir = os.path.join(data_root, ann_dir)
            if split is not None and not os.path.isabs(split):
                split = os.path.join(data_root, split)
        self.data_root, self.img_dir, self.ann_dir, self.split = data_root, img_dir, ann_dir, split
        self.img_infos = self.load_annotations()

    def load_annotations(self):
        if not os.path.isdir(self.img_dir):
            raise FileNotFoundError("img_dir %r does not exist" % (self.img_dir,))
        infos = []
        if self.split is not None:
            with open(self.split) as f:
                names = [ln.strip() for ln in f if ln.strip()]
            for n in names:
                info = dict(filename=n + self.img_suffix)
                if self.ann_dir is not None:
                    info["ann"] = dict(seg_map=n + self.seg_map_suffix)
                infos.append(info)
        else:
            for img in _scan(self.img_dir, self.img_suffix):
                info = dict(filename=img)
                if self.ann_dir is not None:
                    info["ann"] = dict(seg_map=img[:-len(self.img_suffix)] + self.seg_map_suffix)
                infos.append(info)
        return sorted(infos, key=lambda i: i["filename"])

    def __len__(self):
        return len(self.img_infos)

    def image_path(self, idx):
        return os.path.join(self.img_dir, self.img_infos[idx]["filename"])

    def label_path(self, idx):
        ann = self.img_infos[idx].get("ann")
        return None if ann is None or self.ann_dir is None else os.path.join(self.ann_dir, ann["seg_map"])

    def read(self, idx):
        """(img uint8 [H, W, 3] RGB, label uint8 [H, W] or None, file name) as host tensors."""
        from PIL import Image
        path = self.image_path(idx)
        with Image.open(path) as im:
            img = np.array(im.convert("RGB"), dtype=np.uint8)
        label = None
        lp = self.label_path(idx)
        if lp is not None and not self.test_mode_without_labels():
            with Image.open(lp) as im:
                label = np.array(im)
            if label.ndim != 2:
                raise ValueError("label map %r is not single-channel (shape %s)" % (lp, label.shape))
            if label.dtype != np.uint8:
                if label.max(initial=0) > 255:
                    raise ValueError("label map %r has ids above 255" % (lp,))
                label = label.astype(np.uint8)
            if label.shape != img.shape[:2]:
                raise ValueError("image %s and label map %s differ in size" % (img.shape[:2], label.shape))
            if self.reduce_zero_label:
                label = label.copy()
                label[label == 0] = 255
                label -= 1
                label[label == 254] = 255
            label = torch.from_numpy(np.ascontiguousarray(label))
        return torch.from_numpy(img), label, path

    def test_mode_without_labels(self):
        return self.test_mode and self.ann_dir is None

    def get_gt_seg_maps(self):
        """Label maps of every sample in dataset order (mmseg's evaluate() input)."""
        return [self.read(i)[1].numpy() for i in range(len(self))]


_CITYSCAPES_CLASSES = ("road", "sidewalk", "building", "wall", "fence", "pole", "traffic light",
                       "traffic sign", "vegetation", "terrain", "sky", "person", "rider", "car",
                       "truck", "bus", "train", "motorcycle", "bicycle")
_CITYSCAPES_PALETTE = [(128, 64, 128), (244, 35, 232), (70, 70, 70), (102, 102, 156), (190, 153, 153),
                       (153, 153, 153), (250, 170, 30), (220, 220, 0), (107, 142, 35), (152, 251, 152),
                       (70, 130, 180), (220, 20, 60), (255, 0, 0), (0, 0, 142), (0, 0, 70),
                       (0, 60, 100), (0, 80, 100), (0, 0, 230), (119, 11, 32)]


@DATASETS.register_module(name=["CityscapesDataset", "CityscapesDataset19"])
class CityscapesDataset(CustomDataset):
    """19-class Cityscapes: ``*_leftImg8bit.png`` images, ``*_gtFine_labelTrainIds.png`` label maps."""
    CLASSES = _CITYSCAPES_CLASSES
    PALETTE = _CITYSCAPES_PALETTE

    def __init__(self, img_suffix="_leftImg8bit.png", seg_map_suffix="_gtFine_labelTrainIds.png", **kw):
        super().__init__(img_suffix=img_suffix, seg_map_suffix=seg_map_suffix, **kw)


def build_dataset(cfg, default_args=None):
    if isinstance(cfg, (list, tuple)):
        if len(cfg) != 1:
            raise NotImplementedError("concatenated datasets (%d entries): pass one" % len(cfg))
        cfg = cfg[0]
    return build_from_cfg(dict(cfg), DATASETS, default_args)


# ---- mmseg transform list -> GPU pipeline settings ---------------------------------------------
_IGNORED = ("LoadImageFromFile", "LoadAnnotations", "DefaultFormatBundle", "Collect", "ImageToTensor")


def train_pipeline_kwargs(pipeline):
    """Keyword arguments of GpuTrainPipeline from a training transform list (the order of the
    reference's list — Resize, RandomCrop, RandomFlip, PhotoMetricDistortion, Normalize, Pad — is the
    order the fused kernel applies; anything else in the list is an error, not a silent skip)."""
    kw = dict(ratio_range=(1.0, 1.0), cat_max_ratio=1.0, flip_ratio=0.0, photometric=False)
    seen = []
    for t in pipeline:
        t = dict(t)
        name = t.pop("type")
        seen.append(name)
        if name in _IGNORED:
            continue
        if name == "Resize":
            if t.get("keep_ratio", True) is not True:
                raise NotImplementedError("Resize(keep_ratio=False)")
            kw["img_scale"] = tuple(t["img_scale"])
            if t.get("ratio_range") is not None:
                kw["ratio_range"] = tuple(t["ratio_range"])
        elif name == "RandomCrop":
            kw["crop_size"] = tuple(t["crop_size"])
            kw["cat_max_ratio"] = t.get("cat_max_ratio", 1.0)
        elif name == "RandomFlip":
            kw["flip_ratio"] = t.get("flip_ratio", t.get("prob", 0.0)) or 0.0
            if t.get("direction", "horizontal") != "horizontal":
                raise NotImplementedError("RandomFlip(direction=%r)" % t["direction"])
        elif name == "PhotoMetricDistortion":
            if t:
                raise NotImplementedError("PhotoMetricDistortion with non-default ranges %r" % (t,))
            kw["photometric"] = True
        elif name == "Normalize":
            kw["mean"], kw["std"] = tuple(t["mean"]), tuple(t["std"])
            kw["to_rgb"] = bool(t.get("to_rgb", True))
        elif name == "Pad":
            if t.get("size") is None:
                raise NotImplementedError("Pad(size_divisor=...)")
            kw["pad_size"] = tuple(t["size"])
            kw["pad_val"], kw["seg_pad_val"] = t.get("pad_val", 0), t.get("seg_pad_val", 255)
        else:
            raise NotImplementedError("transform %r has no GPU counterpart in this build" % name)
    want = ["Resize", "RandomCrop", "RandomFlip", "PhotoMetricDistortion", "Normalize", "Pad"]
    order = [n for n in seen if n in want]
    if order != [n for n in want if n in order]:
        raise NotImplementedError("transform order %r (the fused kernel applies %r)" % (order, want))
    if "img_scale" not in kw or "crop_size" not in kw:
        raise ValueError("a training pipeline needs Resize(img_scale=...) and RandomCrop(crop_size=...)")
    pad = kw.pop("pad_size", None)
    if pad is not None and tuple(pad) != tuple(kw["crop_size"]):
        raise NotImplementedError("Pad size %r != crop size %r" % (pad, kw["crop_size"]))
    return kw


def eval_pipeline_kwargs(pipeline):
    """(img_scale, mean, std, to_rgb) of a single-scale, no-flip MultiScaleFlipAug test list."""
    out = dict(img_scale=None, mean=(123.675, 116.28, 103.53), std=(58.395, 57.12, 57.375), to_rgb=True)
    for t in pipeline:
        t = dict(t)
        name = t.pop("type")
        if name == "MultiScaleFlipAug":
            if t.get("flip", False) or t.get("img_ratios") is not None:
                raise NotImplementedError("multi-scale / flip test-time augmentation in the val loader")
            out["img_scale"] = tuple(t["img_scale"])
            for s in t.get("transforms", []):
                if s.get("type") == "Normalize":
                    out.update(mean=tuple(s["mean"]), std=tuple(s["std"]), to_rgb=bool(s.get("to_rgb", True)))
        elif name == "Normalize":
            out.update(mean=tuple(t["mean"]), std=tuple(t["std"]), to_rgb=bool(t.get("to_rgb", True)))
    return out
