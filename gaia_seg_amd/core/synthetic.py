"""Seeded synthetic batches with the shapes and value ranges of the reference pipeline
(SURVEY.md §8d): img ~ N(0,1) (mean/std-normalised images), labels uniform over the classes with
5 % of the pixels and a border band set to 255 (what Pad(seg_pad_val=255) produces,
configs/_dynamic_/models/pspnet_ar50to101v2_gsync.py:72)."""
import torch


def make_batch(n, h, w, num_classes=19, seed=0, device="cpu", border=8, ignore_frac=0.05):
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(n, 3, h, w, generator=g)
    gt = torch.randint(0, num_classes, (n, 1, h, w), generator=g)
    gt[torch.rand(n, 1, h, w, generator=g) < ignore_frac] = 255
    if border > 0:
        gt[:, :, -border:, :] = 255
        gt[:, :, :, -border:] = 255
    metas = [dict(ori_shape=(h, w, 3), img_shape=(h, w, 3), pad_shape=(h, w, 3), flip=False,
                  scale_factor=1.0, filename="synthetic_%d" % i) for i in range(n)]
    return dict(img=img.to(device), img_metas=metas, gt_semantic_seg=gt.to(device))


class SyntheticLoader:
    """Endless iterable of seeded batches (a different seed per iteration and per rank)."""

    def __init__(self, samples_per_gpu, size, num_classes=19, seed=0, rank=0, device="cuda",
                 pool=4):
        self.batches = [make_batch(samples_per_gpu, size[0], size[1], num_classes,
                                   seed * 1000003 + rank * 1009 + i, device) for i in range(pool)]
        self.i = 0

    def __iter__(self):
        return self

    def __next__(self):
        b = self.batches[self.i % len(self.batches)]
        self.i += 1
        return b
