"""GPU input pipeline (gs_seg_augment) against the oracle's numpy restatement of the reference's
train_pipeline transforms, on the same random decisions."""
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
