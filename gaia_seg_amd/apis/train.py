"""train_segmentor — host-side mirror of gaiaseg/apis/train.py:47-186.

Same call contract (model, train_sampler, val_sampler, dataset, cfg, distributed, validate,
timestamp, meta) and the same feature switches read with cfg.get (manipulate_arch :142,
lr_scaler :103-113, resume_from / load_from :172-175).  What differs by design:
  * the model is not wrapped in MMDistributedDataParallel: parameters and gradients live in flat
    arenas and a bucketed RCCL all-reduce over slices of the gradient arena is driven by the
    backward tape (core/dist.py);
  * the optimizer is the fused flat-arena SGD kernel (cfg.optimizer must be SGD, as in the
    in-tree config);
  * datasets: the reference's CityscapesDataset19 is not defined anywhere and
    no dataset exists offline, so ``type='SyntheticSegDataset'`` (seeded tensors of the pipeline's
    output shapes) is the built-in loader; any iterable of
    dict(img, img_metas, gt_semantic_seg) batches is accepted.
"""
import random

import numpy as np
import torch

from ..core import dist as gdist
from ..core.param_arena import ParamArena
from ..core.runner import (ArenaOptimizerHook, CheckpointHook, FixedLrUpdaterHook,
                           IterBasedRunner, ManipulateArchHook, PolyLrUpdaterHook, TextLoggerHook)
from ..core.synthetic import SyntheticLoader


def set_random_seed(seed, deterministic=False):
    """gaiaseg/apis/train.py:30-45.  (The HIP kernels are bit-reproducible regardless of
    ``deterministic``: no float atomics are used anywhere on the path.)"""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def build_dataloader(dataset_cfg, samples_per_gpu, seed=0, device="cuda", num_classes=19,
                     workers_per_gpu=2, train=True, device_cache_gb=None):
    """mmseg ``build_dataset`` + ``build_dataloader(..., dist, seed, drop_last=True)``
    (gaiaseg/apis/train.py:74-84, tools/train_supernet.py:197) for this path: a config dict naming a
    registered file-backed dataset (``CityscapesDataset19`` of the in-tree configs, ``CityscapesDataset``,
    ``CustomDataset``) becomes a loader whose transforms run on the GPU; ``SyntheticSegDataset`` gives
    the seeded synthetic batches; anything else that is already an iterable of batches passes through."""
    if isinstance(dataset_cfg, (list, tuple)):
        if len(dataset_cfg) != 1:
            raise NotImplementedError("concatenated datasets (%d entries)" % len(dataset_cfg))
        dataset_cfg = dataset_cfg[0]
    if isinstance(dataset_cfg, dict):
        t = dataset_cfg.get("type")
        if t == "SyntheticSegDataset":
            return SyntheticLoader(samples_per_gpu, tuple(dataset_cfg["size"]),
                                   dataset_cfg.get("num_classes", num_classes), seed=seed,
                                   rank=gdist.rank(), device=device)
        from ..datasets import (DATASETS, FileBatchLoader, FileEvalLoader, build_dataset,
                                eval_pipeline_kwargs, train_pipeline_kwargs)
        if t not in DATASETS:
            raise NotImplementedError("dataset type %r is not registered (have %s and "
                                      "SyntheticSegDataset)" % (t, sorted(DATASETS.module_dict)))
        ds = build_dataset(dataset_cfg)
        # decoded uint8 samples kept in HBM (``data.device_cache_gb``, a key of this build; None = the
        # loaders' defaults, 0 = off)
        extra = {} if device_cache_gb is None else dict(device_cache_bytes=int(device_cache_gb * (1 << 30)))
        if train:
            return FileBatchLoader(ds, samples_per_gpu, train_pipeline_kwargs(ds.pipeline),
                                   workers_per_gpu=workers_per_gpu, seed=seed, rank=gdist.rank(),
                                   world=gdist.world_size(), device=device, **extra)
        tk = eval_pipeline_kwargs(ds.pipeline)
        return FileEvalLoader(ds, samples_per_gpu, tk["img_scale"], tk["mean"], tk["std"], tk["to_rgb"],
                              workers_per_gpu=workers_per_gpu, rank=gdist.rank(),
                              world=gdist.world_size(), device=device, **extra)
    return dataset_cfg  # already an iterable of batches


def train_segmentor(model, train_sampler, val_sampler, dataset, cfg, distributed=False,
                    validate=False, timestamp=None, meta=None, logger=None):
    device = torch.device("cuda", torch.cuda.current_device())
    model = model.to(device)
    arena = ParamArena(model)
    gdist.sync_module_states(model, arena)   # the DDP wrap-time broadcast (:88-96)
    reducer = gdist.GradReducer(arena.flat_grad, arena.segments,
                                bucket_bytes=cfg.get("bucket_bytes", 64 << 20))
    opt = dict(cfg.optimizer)
    if opt.pop("type", "SGD") != "SGD":
        raise NotImplementedError("only SGD (the in-tree config) has a fused arena kernel")
    lr = opt["lr"]
    lr_scaler = cfg.get("lr_scaler")      # gaiaseg/apis/train.py:103-113
    if lr_scaler is not None:
        total_batch = cfg.data["samples_per_gpu"] * gdist.world_size()
        if lr_scaler.get("policy", "linear") == "linear":
            lr = lr_scaler["base_lr"] * total_batch
        else:
            lr = lr_scaler["base_lr"] * total_batch ** lr_scaler.get("temperature", 0.5)
    runner = IterBasedRunner(model, arena, reducer, base_lr=lr, momentum=opt.get("momentum", 0.0),
                             weight_decay=opt.get("weight_decay", 0.0),
                             max_iters=cfg.runner["max_iters"], work_dir=cfg.get("work_dir"),
                             logger=logger, meta=meta)
    if cfg.get("manipulate_arch", True):  # :142-146
        runner.register_hook(ManipulateArchHook(train_sampler))
    lrc = dict(cfg.get("lr_config") or dict(policy="fixed"))
    policy = lrc.pop("policy", "fixed")
    runner.register_hook(PolyLrUpdaterHook(**lrc) if policy == "poly" else FixedLrUpdaterHook())
    if cfg.get("optimizer_config", {}) and dict(cfg.optimizer_config).get("grad_clip"):
        raise NotImplementedError("grad_clip")
    runner.register_hook(ArenaOptimizerHook())
    ck = cfg.get("checkpoint_config")
    if ck:
        runner.register_hook(CheckpointHook(**dict(ck)))
    lg = cfg.get("log_config")
    if lg:
        runner.register_hook(TextLoggerHook(interval=lg.get("interval", 50), logger=logger))
    if validate and cfg.get("evaluation"):
        # gaiaseg/apis/train.py:150-170: (Dist)CrossArchEvalHook over the val anchors
        from ..core.evaluation import CrossArchEvalHook
        ev = dict(cfg.evaluation)
        val_cfg = cfg.data.get("val") or cfg.data["train"]
        val_loader = build_dataloader(val_cfg, cfg.data["samples_per_gpu"], seed=12345,
                                      device=device, train=cfg.data.get("val") is None,
                                      workers_per_gpu=cfg.data.get("workers_per_gpu", 2),
                                      device_cache_gb=cfg.data.get("device_cache_gb"))
        runner.register_hook(CrossArchEvalHook(val_loader, val_sampler,
                                               interval=ev.get("interval", 8000),
                                               num_batches=ev.get("num_batches", 4),
                                               num_classes=model.num_classes, logger=logger))
    if cfg.get("resume_from"):
        runner.resume(cfg.resume_from)
    elif cfg.get("load_from"):
        runner.load_checkpoint(cfg.load_from)
    from .test import apply_bn_calibration
    apply_bn_calibration(model, cfg.get("caliberate_bn"), "train")   # gaiaseg/apis/train.py:177-184
    loader = build_dataloader(dataset, cfg.data["samples_per_gpu"], seed=cfg.get("seed") or 0,
                              device=device, workers_per_gpu=cfg.data.get("workers_per_gpu", 2),
                              device_cache_gb=cfg.data.get("device_cache_gb"))
    runner.run([loader], cfg.get("workflow", [("train", 1)]))
    return runner
