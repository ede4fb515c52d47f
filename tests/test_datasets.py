"""File-backed datasets and their loaders (host logic; no GPU).

Reference surface: the ``data`` block of configs/_dynamic_/models/pspnet_ar50to101v2_gsync.py:52-135
(``CityscapesDataset19`` with ``img_dir`` / ``ann_dir`` / ``pipeline``), mmseg's CustomDataset file
pairing, ``build_dataloader(..., dist, seed, drop_last=True)`` at gaiaseg/apis/train.py:74-84."""
import os

import numpy as np
import pytest
import torch

from gaia_seg_amd.datasets import (DATASETS, build_dataset, epoch_indices, eval_pipeline_kwargs,
                                   train_pipeline_kwargs)

TRAIN_PIPELINE = [
    dict(type="LoadImageFromFile"),
    dict(type="LoadAnnotations"),
    dict(type="Resize", img_scale=(2048, 1024), ratio_range=(0.5, 2.0)),
    dict(type="RandomCrop", crop_size=(512, 1024), cat_max_ratio=0.75),
    dict(type="RandomFlip", flip_ratio=0.5),
    dict(type="PhotoMetricDistortion"),
    dict(type="Normalize", mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True),
    dict(type="Pad", size=(512, 1024), pad_val=0, seg_pad_val=255),
    dict(type="DefaultFormatBundle"),
    dict(type="Collect", keys=["img", "gt_semantic_seg"]),
]
TEST_PIPELINE = [
    dict(type="LoadImageFromFile"),
    dict(type="MultiScaleFlipAug", img_scale=(2048, 1024), flip=False,
         transforms=[dict(type="Resize", keep_ratio=True), dict(type="RandomFlip"),
                     dict(type="Normalize", mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375],
                          to_rgb=True),
                     dict(type="ImageToTensor", keys=["img"]), dict(type="Collect", keys=["img"])]),
]


def make_cityscapes(root, cities=(("aachen", 3), ("bochum", 2)), size=(24, 40), split="train", seed=0):
    """A Cityscapes-shaped tree of tiny PNGs; returns {relative image name: (rgb, label)}."""
    from PIL import Image
    rng = np.random.RandomState(seed)
    truth = {}
    for city, n in cities:
        os.makedirs(os.path.join(root, "leftImg8bit", split, city), exist_ok=True)
        os.makedirs(os.path.join(root, "gtFine", split, city), exist_ok=True)
        for i in range(n):
            stem = "%s_%06d_000019" % (city, i)
            rgb = rng.randint(0, 256, size + (3,), dtype=np.uint8)
            lab = rng.randint(0, 19, size, dtype=np.uint8)
            lab[rng.rand(*size) < 0.1] = 255
            Image.fromarray(rgb, "RGB").save(os.path.join(root, "leftImg8bit", split, city,
                                                          stem + "_leftImg8bit.png"))
            Image.fromarray(lab, "L").save(os.path.join(root, "gtFine", split, city,
                                                        stem + "_gtFine_labelTrainIds.png"))
            # files a scan must not pick up
            Image.fromarray(lab, "L").save(os.path.join(root, "gtFine", split, city,
                                                        stem + "_gtFine_labelIds.png"))
            truth[os.path.join(city, stem + "_leftImg8bit.png")] = (rgb, lab)
    return truth


def test_registered_names_of_the_reference_configs():
    for name in ("CityscapesDataset19", "CityscapesDataset", "CustomDataset"):
        assert name in DATASETS
    assert DATASETS.get("CityscapesDataset19") is DATASETS.get("CityscapesDataset")
    assert len(DATASETS.get("CityscapesDataset").CLASSES) == 19


def test_cityscapes_pairs_files_in_name_order_and_decodes_them(tmp_path):
    truth = make_cityscapes(str(tmp_path))
    ds = build_dataset([dict(type="CityscapesDataset19", data_root=str(tmp_path),
                             img_dir="leftImg8bit/train", ann_dir="gtFine/train", pipeline=TRAIN_PIPELINE)])
    assert len(ds) == 5
    names = [i["filename"] for i in ds.img_infos]
    assert names == sorted(truth)
    for k, name in enumerate(names):
        img, label, path = ds.read(k)
        assert path.endswith(name) and os.path.isabs(path)
        assert ds.label_path(k).endswith(name.replace("_leftImg8bit.png", "_gtFine_labelTrainIds.png"))
        rgb, lab = truth[name]
        assert img.dtype == torch.uint8 and tuple(img.shape) == rgb.shape
        assert np.array_equal(img.numpy(), rgb)          # RGB order, as decoded
        assert label.dtype == torch.uint8 and np.array_equal(label.numpy(), lab)
    maps = ds.get_gt_seg_maps()
    assert len(maps) == 5 and np.array_equal(maps[0], truth[names[0]][1])


def test_custom_dataset_split_file_and_reduce_zero_label(tmp_path):
    from PIL import Image
    os.makedirs(tmp_path / "img")
    os.makedirs(tmp_path / "ann")
    rng = np.random.RandomState(1)
    for n in ("a", "b", "c"):
        Image.fromarray(rng.randint(0, 256, (8, 9, 3), dtype=np.uint8), "RGB").save(tmp_path / "img" / (n + ".jpg"))
        Image.fromarray(np.arange(72, dtype=np.uint8).reshape(8, 9) % 4, "L").save(tmp_path / "ann" / (n + ".png"))
    (tmp_path / "split.txt").write_text("c\na\n\n")
    ds = build_dataset(dict(type="CustomDataset", data_root=str(tmp_path), img_dir="img", ann_dir="ann",
                            split="split.txt", reduce_zero_label=True, classes=("x", "y", "z")))
    assert [i["filename"] for i in ds.img_infos] == ["a.jpg", "c.jpg"]      # only the split, by name
    assert ds.CLASSES == ("x", "y", "z")
    _, label, _ = ds.read(0)
    raw = np.arange(72, dtype=np.uint8).reshape(8, 9) % 4
    want = np.where(raw == 0, 255, raw - 1).astype(np.uint8)
    assert np.array_equal(label.numpy(), want)
    with pytest.raises(FileNotFoundError):
        build_dataset(dict(type="CustomDataset", img_dir=str(tmp_path / "nope")))
    with pytest.raises(KeyError):
        build_dataset(dict(type="NoSuchDataset", img_dir="x"))


def test_label_and_image_of_different_size_are_refused(tmp_path):
    from PIL import Image
    os.makedirs(tmp_path / "img")
    os.makedirs(tmp_path / "ann")
    Image.fromarray(np.zeros((8, 9, 3), np.uint8), "RGB").save(tmp_path / "img" / "a.png")
    Image.fromarray(np.zeros((8, 8), np.uint8), "L").save(tmp_path / "ann" / "a.png")
    ds = build_dataset(dict(type="CustomDataset", img_dir=str(tmp_path / "img"), img_suffix=".png",
                            ann_dir=str(tmp_path / "ann")))
    with pytest.raises(ValueError):
        ds.read(0)


def test_pipeline_lists_of_the_reference_config_translate():
    kw = train_pipeline_kwargs(TRAIN_PIPELINE)
    assert kw == dict(ratio_range=(0.5, 2.0), cat_max_ratio=0.75, flip_ratio=0.5, photometric=True,
                      img_scale=(2048, 1024), crop_size=(512, 1024),
                      mean=(123.675, 116.28, 103.53), std=(58.395, 57.12, 57.375), to_rgb=True,
                      pad_val=0, seg_pad_val=255)
    tk = eval_pipeline_kwargs(TEST_PIPELINE)
    assert tk["img_scale"] == (2048, 1024) and tk["to_rgb"] is True and tk["mean"][0] == 123.675
    # what the fused kernel cannot do is an error, not a silent skip
    with pytest.raises(NotImplementedError):
        train_pipeline_kwargs(TRAIN_PIPELINE[:4] + [dict(type="RandomRotate", prob=0.5, degree=10)] + TRAIN_PIPELINE[4:])
    with pytest.raises(NotImplementedError):      # Normalize before the photometric distortion
        train_pipeline_kwargs(TRAIN_PIPELINE[:5] + [TRAIN_PIPELINE[6], TRAIN_PIPELINE[5]] + TRAIN_PIPELINE[7:])
    with pytest.raises(NotImplementedError):
        train_pipeline_kwargs([dict(type="Resize", img_scale=(64, 32)), dict(type="RandomCrop", crop_size=(16, 16)),
                               dict(type="Pad", size=(32, 32))])
    with pytest.raises(NotImplementedError):
        eval_pipeline_kwargs([dict(type="MultiScaleFlipAug", img_scale=(64, 32), flip=True, transforms=[])])


def test_epoch_indices_follow_distributed_sampler():
    from torch.utils.data.distributed import DistributedSampler
    for n, world in ((11, 4), (5, 8), (16, 2), (7, 1)):
        data = list(range(n))
        for epoch in (0, 3):
            for rank in range(world):
                ref = DistributedSampler(data, num_replicas=world, rank=rank, shuffle=True, seed=7)
                ref.set_epoch(epoch)
                assert epoch_indices(n, epoch, 7, rank, world) == list(ref)
        assert epoch_indices(n, 0, 0, 0, world, shuffle=False)[0] == 0
    assert epoch_indices(0, 0, 0, 0, 2) == []


class _FakePipeline:
    """Stands in for GpuTrainPipeline (which needs the GPU): records what the loader hands over."""

    def __init__(self, **kw):
        self.kw = kw

    def batch(self, samples):
        return [s[2] for s in samples]

    def test_batch(self, samples, img_scale):
        return ([s[2] for s in samples], img_scale)


def test_train_loader_shards_shuffles_prefetches_and_drops_the_last_batch(tmp_path, monkeypatch):
    import gaia_seg_amd.datasets.loader as loader_mod
    make_cityscapes(str(tmp_path), cities=(("a", 4), ("b", 3)))
    ds = build_dataset(dict(type="CityscapesDataset", data_root=str(tmp_path), img_dir="leftImg8bit/train",
                            ann_dir="gtFine/train", pipeline=TRAIN_PIPELINE))
    monkeypatch.setattr(loader_mod, "GpuTrainPipeline", _FakePipeline)
    streams = []
    for rank in range(2):
        ld = loader_mod.FileBatchLoader(ds, 2, train_pipeline_kwargs(ds.pipeline), workers_per_gpu=3, seed=5,
                                        rank=rank, world=2, device="cpu")
        assert ld.pipeline.kw["src_is_rgb"] is True and ld.pipeline.kw["crop_size"] == (512, 1024)
        got = [next(ld) for _ in range(6)]            # three epochs of two batches each (4 samples / rank)
        ld.close()
        want = []
        for epoch in range(3):
            idx = epoch_indices(7, epoch, 5, rank, 2)
            assert len(idx) == 4
            want += [[ds.image_path(i) for i in idx[k:k + 2]] for k in (0, 2)]
        assert got == want
        streams.append(got)
    assert streams[0] != streams[1]
    # the same (seed, rank) gives the same stream again
    ld = loader_mod.FileBatchLoader(ds, 2, train_pipeline_kwargs(ds.pipeline), seed=5, rank=1, world=2, device="cpu")
    assert [next(ld) for _ in range(6)] == streams[1]
    ld.close()
    with pytest.raises(ValueError):      # 7 samples over 4 ranks: two per rank, a batch of 3 never fills
        loader_mod.FileBatchLoader(ds, 3, train_pipeline_kwargs(ds.pipeline), rank=0, world=4, device="cpu")


def test_eval_loader_walks_its_shard_in_order_and_cycles(tmp_path, monkeypatch):
    import gaia_seg_amd.datasets.loader as loader_mod
    make_cityscapes(str(tmp_path), cities=(("a", 3), ("b", 2)), split="val")
    ds = build_dataset(dict(type="CityscapesDataset19", data_root=str(tmp_path), img_dir="leftImg8bit/val",
                            ann_dir="gtFine/val", pipeline=TEST_PIPELINE))
    monkeypatch.setattr(loader_mod, "GpuTrainPipeline", _FakePipeline)
    tk = eval_pipeline_kwargs(ds.pipeline)
    ld = loader_mod.FileEvalLoader(ds, 2, tk["img_scale"], tk["mean"], tk["std"], tk["to_rgb"], rank=1, world=2,
                                   device="cpu")
    assert len(ld) == 1                                   # samples 1, 3 of 5
    a, b = next(ld), next(ld)
    ld.close()
    assert a == ([ds.image_path(1), ds.image_path(3)], (2048, 1024)) and b == a
