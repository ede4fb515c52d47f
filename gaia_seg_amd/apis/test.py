"""Test-time drivers — host-side mirror of gaiaseg/apis/test.py:30-186 and the BatchNorm
re-calibration switches of gaiaseg/apis/train.py:177-184 / tools/test_supernet.py:190-198.

What differs by design: label maps stay on the device until a rank's shard is complete, and the
per-rank result lists are exchanged as python objects over the gloo host group
(``core.dist.gather_objects``) instead of pickled bytes staged through padded CUDA tensors
(collect_results_gpu, :155-186) or a shared tmp dir (collect_results_cpu, :113-152); the ordering
contract is the reference's: rank r holds samples r, r + world, ... and rank 0 returns them
interleaved and truncated to the dataset size."""
import torch
from torch.nn.modules.batchnorm import _BatchNorm

from ..core import dist as gdist


def apply_bn_calibration(model, calib_cfg, phase):
    """``cfg.caliberate_bn`` (sic — the reference's key).

    phase 'train' — ``reset_stats``: running_mean <- 0, running_var <- 1 before the run
                    (gaiaseg/apis/train.py:177-184);
    phase 'test'  — ``use_minibatch_stats``: drop the running statistics so that eval-mode BN
                    normalises with the statistics of the batch it sees
                    (tools/test_supernet.py:190-198)."""
    if not calib_cfg:
        return 0
    n = 0
    for m in model.modules():
        if not isinstance(m, _BatchNorm):
            continue
        if phase == "train" and calib_cfg.get("reset_stats", False):
            with torch.no_grad():
                m.running_mean.zero_()
                m.running_var.fill_(1)
            n += 1
        elif phase == "test" and calib_cfg.get("use_minibatch_stats", False):
            m.running_mean = None
            m.running_var = None
            m.track_running_stats = False
            # DynamicBatchNorm2d caches one BNParams view per mode holding the OLD buffers (ops.conv_bn
            # decides "batch statistics?" from that view): an eval forward before the calibration
            # would otherwise keep normalising with the running statistics
            m.__dict__.pop("_bnp_cache", None)
            n += 1
    return n


def _batch_results(model, data, **kw):
    with torch.no_grad():
        return model(return_loss=False, **data, **kw)


def single_gpu_test(model, data_loader, **kw):
    """gaiaseg/apis/test.py:30-88 without the visualisation branch: list of label maps."""
    model.eval()
    results = []
    for data in data_loader:
        out = _batch_results(model, data, **kw)
        results.extend(out if isinstance(out, list) else [out])
    return results


def collect_results(result_part, size):
    """Rank 0 gets the results of all ranks in dataset order (sample i lives on rank i % world at
    position i // world, the DistributedSampler layout); other ranks get None."""
    if not gdist.is_dist():
        return list(result_part)[:size]
    parts = gdist.gather_objects(list(result_part))
    if gdist.rank() != 0:
        return None
    ordered = []
    longest = max(len(p) for p in parts)
    for i in range(longest):
        for p in parts:
            if i < len(p):
                ordered.append(p[i])
    return ordered[:size]      # the sampler may have padded the last round


def multi_gpu_test(model, data_loader, size=None, **kw):
    """gaiaseg/apis/test.py:91-130: every rank evaluates its shard, rank 0 returns all results."""
    part = single_gpu_test(model, data_loader, **kw)
    if size is None:
        size = len(part) * gdist.world_size()
    return collect_results(part, size)
