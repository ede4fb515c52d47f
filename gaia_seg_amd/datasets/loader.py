"""Batch loaders over file-backed datasets: decode on host threads, everything else on the GPU.

Replaces mmseg's ``build_dataloader(dataset, samples_per_gpu, workers_per_gpu, ..., dist, seed,
drop_last=True)`` (call site gaiaseg/apis/train.py:74-84) + ``IterLoader`` for this path:

  * sharding and shuffling follow torch's ``DistributedSampler(shuffle=True, seed)``: epoch e draws
    ``randperm(len, generator seeded seed + e)``, pads it to a multiple of the world size by wrapping
    around and gives rank r every world-th index from r; batches of ``samples_per_gpu`` with the
    incomplete last batch of an epoch dropped; the iterator never ends (IterBasedRunner);
  * ``workers_per_gpu`` host THREADS decode ahead of the consumer (Pillow releases the GIL while it
    inflates a PNG); the decoded uint8 arrays go to the device as they are (an image of 2048 x 1024 is
    6 MB) and ``GpuTrainPipeline`` produces the normalised crop and the label map there;
  * the random decisions of the augmentation are drawn by the pipeline's own RandomState in consumption
    order, so a (seed, rank) pair fixes the whole stream of batches;
  * decoded samples stay on the device as uint8 up to ``device_cache_bytes`` (default 32 GiB of the
    288 GB of HBM: all of Cityscapes' 2975 training images are 24 GB), so from the second epoch on the
    host does no decoding at all.  One decode thread gives 10-13 images/s on 2048 x 1024 PNGs and an
    MI355X trains 160 crops/s, i.e. without the cache the first epoch wants ~14 ``workers_per_gpu``.
"""
import math
from collections import deque
from concurrent.futures import ThreadPoolExecutor

import torch

from .gpu_pipeline import GpuTrainPipeline


def epoch_indices(n, epoch, seed, rank, world, shuffle=True):
    """This rank's sample indices of one epoch (DistributedSampler layout, drop_last=False)."""
    if n <= 0:
        return []
    if shuffle:
        g = torch.Generator()
        g.manual_seed(int(seed) + int(epoch))
        idx = torch.randperm(n, generator=g).tolist()
    else:
        idx = list(range(n))
    total = int(math.ceil(n / world)) * world
    pad = total - n
    if pad:
        idx += (idx * int(math.ceil(pad / n)))[:pad]
    return idx[rank:total:world]


class _Prefetcher:
    """Decodes dataset samples on a thread pool, in the order asked for, a bounded distance ahead;
    keeps the decoded uint8 tensors on the device while they fit the cache budget."""

    def __init__(self, dataset, workers, depth, device="cpu", cache_bytes=0):
        self.dataset = dataset
        self.pool = ThreadPoolExecutor(max_workers=max(1, workers), thread_name_prefix="gs-decode")
        self.depth = max(1, depth)
        self.pending = deque()
        self.device = torch.device(device)
        self.cache, self.cache_left = {}, int(cache_bytes) if self.device.type == "cuda" else 0
        self.decoded = 0          # samples that went through the decoder (diagnostics, tests)

    def fill(self, index_iter):
        while len(self.pending) < self.depth:
            i = next(index_iter)
            hit = self.cache.get(i)
            self.pending.append(_Done((i, hit)) if hit is not None
                                else self.pool.submit(lambda k=i: (k, self.dataset.read(k))))

    def get(self, index_iter):
        self.fill(index_iter)
        i, sample = self.pending.popleft().result()
        if i is not None and i not in self.cache:
            self.decoded += 1
            nbytes = sample[0].numel() + (sample[1].numel() if sample[1] is not None else 0)
            if nbytes <= self.cache_left:
                sample = (sample[0].to(self.device, non_blocking=True),
                          None if sample[1] is None else sample[1].to(self.device, non_blocking=True),
                          sample[2])
                self.cache[i] = sample
                self.cache_left -= nbytes
        self.fill(index_iter)
        return sample

    def close(self):
        for f in self.pending:
            f.cancel()
        self.pending.clear()
        self.pool.shutdown(wait=False)


class FileBatchLoader:
    """Endless training batches dict(img, img_metas, gt_semantic_seg) on the device."""

    def __init__(self, dataset, samples_per_gpu, pipeline_kwargs, workers_per_gpu=2, seed=0, rank=0,
                 world=1, device="cuda", shuffle=True, device_cache_bytes=32 << 30):
        if len(dataset) == 0:
            raise ValueError("empty dataset (%s)" % getattr(dataset, "img_dir", "?"))
        self.dataset, self.bs = dataset, int(samples_per_gpu)
        self.seed, self.rank, self.world, self.shuffle = int(seed or 0), rank, world, shuffle
        if len(epoch_indices(len(dataset), 0, self.seed, rank, world, shuffle)) < self.bs:
            raise ValueError("%d samples over %d rank(s) give less than one batch of %d"
                             % (len(dataset), world, self.bs))
        self.pipeline = GpuTrainPipeline(seed=self.seed * 1000003 + rank * 1009, device=device,
                                         src_is_rgb=True, ignore_index=dataset.ignore_index,
                                         **pipeline_kwargs)
        self.epoch = 0
        self._indices = self._index_stream()
        self._pre = _Prefetcher(dataset, workers_per_gpu, 2 * self.bs, device, device_cache_bytes)

    def _index_stream(self):
        while True:
            idx = epoch_indices(len(self.dataset), self.epoch, self.seed, self.rank, self.world,
                                self.shuffle)
            for k in range(len(idx) // self.bs * self.bs):     # drop_last
                yield idx[k]
            self.epoch += 1

    def __iter__(self):
        return self

    def __next__(self):
        samples = [self._pre.get(self._indices) for _ in range(self.bs)]
        for s in samples:
            if s[1] is None:
                raise ValueError("training sample %s has no label map" % (s[2],))
        return self.pipeline.batch(samples)

    def close(self):
        self._pre.close()


class FileEvalLoader:
    """Validation batches dict(img, img_metas, gt_semantic_seg) in dataset order, this rank's shard
    (sample i on rank i % world), cycling: the cross-arch evaluation hook draws ``num_batches`` per
    anchor (core/evaluation.py)."""

    def __init__(self, dataset, samples_per_gpu, img_scale, mean, std, to_rgb=True, workers_per_gpu=2,
                 rank=0, world=1, device="cuda", device_cache_bytes=8 << 30):
        if len(dataset) == 0:
            raise ValueError("empty dataset (%s)" % getattr(dataset, "img_dir", "?"))
        self.dataset, self.bs, self.img_scale = dataset, int(samples_per_gpu), img_scale
        self.rank, self.world = rank, world
        self.pipeline = GpuTrainPipeline(mean=mean, std=std, to_rgb=to_rgb, device=device,
                                         src_is_rgb=True, photometric=False, flip_ratio=0.0)
        self._indices = self._index_stream()
        self._pre = _Prefetcher(dataset, workers_per_gpu, 2 * self.bs, device, device_cache_bytes)

    def shard(self):
        return list(range(len(self.dataset)))[self.rank::self.world] or [0]

    def _index_stream(self):
        while True:
            for i in self.shard():
                yield i

    def __len__(self):
        return int(math.ceil(len(self.shard()) / self.bs))

    def __iter__(self):
        return self

    def __next__(self):
        samples = [self._pre.get(self._indices)]
        while len(samples) < self.bs:                      # a batch holds samples of one size
            nxt = self._pre.get(self._indices)
            if tuple(nxt[0].shape) != tuple(samples[0][0].shape):
                self._pre.pending.appendleft(_Done((None, nxt)))     # (already counted / cached)
                break
            samples.append(nxt)
        return self.pipeline.test_batch(samples, self.img_scale)

    def close(self):
        self._pre.close()


class _Done:
    """A decoded sample put back at the head of the prefetch queue."""

    def __init__(self, value):
        self.value = value

    def result(self):
        return self.value

    def cancel(self):
        return False
