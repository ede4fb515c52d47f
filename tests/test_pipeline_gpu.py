"""GPU input pipeline (gs_seg_augment) against the oracle's numpy restatement of the reference's
train_pipeline transforms, on the same random decisions."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _sample(h, w, seed):
    rng = np.random.RandomState(seed)
    # smooth-ish image with saturated and grey regions (exercises every HSV sector, s = 0, v = 0)
    base = rng.randint(0, 256, size=(h // 8 + 1, w // 8 + 1, 3)).astype(np.uint8)
    img = np.kron(base, np.ones((8, 8, 1), np.uint8))[:h, :w].copy()
    img[:10] = 0
    img[10:20] = 255
    img[20:30] = img[20:30, :, :1]            # grey rows
    img[rng.rand(h, w) < 0.3] = rng.randint(0, 256, size=3)
    lab = rng.randint(0, 19, size=(h, w)).astype(np.uint8)
    lab[rng.rand(h, w) < 0.05] = 255
    return img, lab


CASES = [
    dict(res=(300, 500), crop=(20, 30, 128, 192), flip=False),                          # no resize
    dict(res=(451, 752), crop=(100, 200, 128, 192), flip=True, pm=dict(b=12.5)),       # upscale
    dict(res=(150, 250), crop=(0, 10, 128, 192), flip=False, pm=dict(c=1.37, first=True, s=0.6)),
    dict(res=(97, 161), crop=(0, 0, 97, 161), flip=True, pm=dict(h=-17, c=0.55, first=False)),  # pad
    dict(res=(600, 1000), crop=(472, 808, 128, 192), flip=False, pm=dict(b=-31.0, s=1.49, h=11)),
    dict(res=(300, 500), crop=(5, 7, 128, 100), flip=True, pm=dict(s=1.2, h=90)),       # pad right
]


@pytest.mark.parametrize("case", CASES)
def test_seg_augment_matches_oracle(hip_lib, case):
    from gaia_seg_amd.datasets import GpuTrainPipeline
    from oracle.pipeline import train_sample
    img, lab = _sample(300, 500, seed=1)
    pm = case.get("pm")
    p = dict(res_h=case["res"][0], res_w=case["res"][1], crop_y=case["crop"][0], crop_x=case["crop"][1],
             crop_h=case["crop"][2], crop_w=case["crop"][3], flip=case["flip"], pm_enable=pm is not None)
    if pm is not None:
        p.update(pm_brightness="b" in pm, pm_delta=pm.get("b", 0.0), pm_contrast="c" in pm,
                 pm_alpha=pm.get("c", 1.0), pm_contrast_first=pm.get("first", False),
                 pm_saturation="s" in pm, pm_sat_alpha=pm.get("s", 1.0), pm_hue="h" in pm,
                 pm_hue_delta=pm.get("h", 0))
    want_img, want_lab = train_sample(img, lab, p, crop_size=(128, 192))
    pipe = GpuTrainPipeline(crop_size=(128, 192))
    batch = pipe.batch([(torch.from_numpy(img), torch.from_numpy(lab))], params=[p])
    got_img = batch["img"][0].cpu().numpy()
    got_lab = batch["gt_semantic_seg"][0, 0].cpu().numpy()
    assert np.array_equal(got_lab, want_lab)
    diff = np.abs(got_img - want_img)
    # the same fp32 operations in the same order: equal up to the last bit of the final division
    assert float(diff.max()) <= 1e-6, (float(diff.max()), int((diff > 1e-6).sum()))
    assert batch["img_metas"][0]["flip"] == case["flip"]


def test_pipeline_draws_and_trains(hip_lib):
    """Seeded end to end: params drawn like the CPU transforms draw them, a batch of two samples,
    and the batch is a valid train_step input."""
    from gaia_seg_amd.datasets import GpuTrainPipeline
    pipe = GpuTrainPipeline(crop_size=(64, 96), img_scale=(256, 128), seed=3)
    samples = [tuple(torch.from_numpy(a) for a in _sample(128, 256, seed=s)) for s in (5, 6)]
    batch = pipe.batch(samples)
    assert tuple(batch["img"].shape) == (2, 3, 64, 96) and batch["img"].dtype == torch.float32
    assert tuple(batch["gt_semantic_seg"].shape) == (2, 1, 64, 96)
    assert batch["gt_semantic_seg"].dtype == torch.int64
    lab = batch["gt_semantic_seg"].cpu()
    assert int(((lab < 0) | ((lab > 18) & (lab != 255))).sum()) == 0
    assert bool(torch.isfinite(batch["img"]).all())


# ---- file-backed datasets through the GPU pipeline ------------------------------------------------
def _small_pipeline(crop=(64, 96), scale=(160, 96)):
    return [dict(type="LoadImageFromFile"), dict(type="LoadAnnotations"),
            dict(type="Resize", img_scale=scale, ratio_range=(0.5, 2.0)),
            dict(type="RandomCrop", crop_size=crop, cat_max_ratio=0.75),
            dict(type="RandomFlip", flip_ratio=0.5), dict(type="PhotoMetricDistortion"),
            dict(type="Normalize", mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True),
            dict(type="Pad", size=crop, pad_val=0, seg_pad_val=255),
            dict(type="DefaultFormatBundle"), dict(type="Collect", keys=["img", "gt_semantic_seg"])]


def test_file_loader_batches_equal_the_oracle_pipeline_on_the_decoded_files(hip_lib, tmp_path):
    """CityscapesDataset19 config dict -> build_dataloader -> batches: every sample equals the oracle's
    numpy transforms applied to the decoded PNG (converted to the BGR order cv2.imread gives, which is
    what the oracle restates) with the decisions the loader's pipeline drew; the stream is the seeded
    DistributedSampler order (configs/_dynamic_/models/pspnet_ar50to101v2_gsync.py:95-118)."""
    from gaia_seg_amd.apis.train import build_dataloader
    from gaia_seg_amd.datasets import draw_train_params, epoch_indices
    from oracle.pipeline import train_sample
    from test_datasets import make_cityscapes
    truth = make_cityscapes(str(tmp_path), cities=(("a", 3), ("b", 3)), size=(96, 160))
    cfg = [dict(type="CityscapesDataset19", data_root=str(tmp_path), img_dir="leftImg8bit/train",
                ann_dir="gtFine/train", pipeline=_small_pipeline())]
    ld = build_dataloader(cfg, 2, seed=4, device="cuda", workers_per_gpu=2)
    names = sorted(truth)
    rng = np.random.RandomState(4 * 1000003)          # the loader's pipeline seed at rank 0
    order = epoch_indices(6, 0, 4, 0, 1) + epoch_indices(6, 1, 4, 0, 1)
    k = 0
    for _ in range(5):                                 # crosses the epoch boundary (3 batches / epoch)
        batch = next(ld)
        assert tuple(batch["img"].shape) == (2, 3, 64, 96) and batch["img"].is_cuda
        for j in range(2):
            rgb, lab = truth[names[order[k]]]
            k += 1
            p = draw_train_params(rng, 96, 160, ld.pipeline.cfg, torch.from_numpy(lab))
            want_img, want_lab = train_sample(rgb[:, :, ::-1].copy(), lab, p, crop_size=(64, 96))
            assert batch["img_metas"][j]["filename"].endswith(names[order[k - 1]])
            assert np.array_equal(batch["gt_semantic_seg"][j, 0].cpu().numpy(), want_lab)
            assert float(np.abs(batch["img"][j].cpu().numpy() - want_img).max()) <= 1e-6
    # the second epoch came out of the device cache: six files decoded once each (plus what the
    # prefetcher had in flight when the first epoch ended)
    assert len(ld._pre.cache) == 6 and ld._pre.decoded <= 6 + 4
    before = ld._pre.decoded
    for _ in range(3):
        next(ld)
    assert ld._pre.decoded == before
    ld.close()
    off = build_dataloader(cfg, 2, seed=4, device="cuda", device_cache_gb=0)
    for _ in range(4):
        next(off)
    assert not off._pre.cache and off._pre.decoded == 8
    off.close()


def test_eval_loader_normalises_whole_images_and_keeps_original_labels(hip_lib, tmp_path):
    from gaia_seg_amd.apis.train import build_dataloader
    from oracle.pipeline import train_sample
    from test_datasets import TEST_PIPELINE, make_cityscapes
    truth = make_cityscapes(str(tmp_path), cities=(("a", 2), ("b", 1)), size=(64, 128), split="val")
    pipe = [dict(TEST_PIPELINE[0]), dict(TEST_PIPELINE[1], img_scale=(128, 64))]
    cfg = dict(type="CityscapesDataset", data_root=str(tmp_path), img_dir="leftImg8bit/val",
               ann_dir="gtFine/val", pipeline=pipe)
    ld = build_dataloader(cfg, 2, device="cuda", train=False)
    names = sorted(truth)
    b0, b1 = next(ld), next(ld)
    assert tuple(b0["img"].shape) == (2, 3, 64, 128) and tuple(b1["img"].shape) == (2, 3, 64, 128)
    seen = [m["filename"] for m in b0["img_metas"] + b1["img_metas"]]
    assert [s.split(os.sep)[-1] for s in seen] == [names[i].split(os.sep)[-1] for i in (0, 1, 2, 0)]   # cycles
    p = dict(res_h=64, res_w=128, crop_y=0, crop_x=0, crop_h=64, crop_w=128, flip=False, pm_enable=False)
    for j in range(2):
        rgb, lab = truth[names[j]]
        want_img, _ = train_sample(rgb[:, :, ::-1].copy(), lab, p, crop_size=(64, 128))
        assert float(np.abs(b0["img"][j].cpu().numpy() - want_img).max()) <= 1e-6
        assert np.array_equal(b0["gt_semantic_seg"][j, 0].cpu().numpy(), lab.astype(np.int64))
        assert b0["img_metas"][j]["ori_shape"] == (64, 128, 3)
    ld.close()


def test_train_cli_from_a_file_backed_config(hip_lib, tmp_path):
    """tools/train_supernet.py with the in-tree FCN supernet config whose ``data`` block names
    CityscapesDataset19 directories, as the reference's config does
    (configs/_dynamic_/models/pspnet_ar50to101v2_gsync.py:95-135): trains and runs the cross-arch
    evaluation from PNG files."""
    import subprocess
    import sys
    from test_datasets import make_cityscapes
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    data = str(tmp_path / "cityscapes")
    make_cityscapes(data, cities=(("a", 3), ("b", 2)), size=(192, 320))
    make_cityscapes(data, cities=(("v", 2),), size=(192, 320), split="val", seed=9)
    norm = "dict(type='Normalize', mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)"
    cfg = tmp_path / "fcn_files.py"
    cfg.write_text(
        "_base_ = [%r]\n" % os.path.join(root, "configs", "supernet", "fcn_ar50to101v2.py") +
        "crop_size = (128, 256)\n"
        "data = dict(_delete_=True, samples_per_gpu=2, workers_per_gpu=2,\n"
        "    train=[dict(type='CityscapesDataset19', data_root=%r, img_dir='leftImg8bit/train',\n" % data +
        "                ann_dir='gtFine/train', pipeline=[dict(type='LoadImageFromFile'), dict(type='LoadAnnotations'),\n"
        "                    dict(type='Resize', img_scale=(320, 192), ratio_range=(0.5, 2.0)),\n"
        "                    dict(type='RandomCrop', crop_size=(128, 256), cat_max_ratio=0.75),\n"
        "                    dict(type='RandomFlip', flip_ratio=0.5), dict(type='PhotoMetricDistortion'), %s,\n" % norm +
        "                    dict(type='Pad', size=(128, 256), pad_val=0, seg_pad_val=255),\n"
        "                    dict(type='DefaultFormatBundle'), dict(type='Collect', keys=['img', 'gt_semantic_seg'])])],\n"
        "    val=dict(type='CityscapesDataset19', data_root=%r, img_dir='leftImg8bit/val', ann_dir='gtFine/val',\n" % data +
        "             pipeline=[dict(type='LoadImageFromFile'), dict(type='MultiScaleFlipAug', img_scale=(320, 192), flip=False,\n"
        "                 transforms=[dict(type='Resize', keep_ratio=True), dict(type='RandomFlip'), %s,\n" % norm +
        "                             dict(type='ImageToTensor', keys=['img']), dict(type='Collect', keys=['img'])])]))\n")
    cmd = [sys.executable, os.path.join(root, "tools", "train_supernet.py"), str(cfg), "--work-dir",
           str(tmp_path / "work"), "--seed", "0", "--max-iters", "4", "--cfg-options", "log_config.interval=2",
           "checkpoint_config.interval=100", "evaluation.interval=4", "evaluation.num_batches=1"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    out = res.stderr + res.stdout
    assert "Iter [4/4]" in out and "decode.loss_seg" in out
    assert out.count("mIoU") >= 3 and "R101" in out, out[-2000:]
