#!/usr/bin/env python
"""Supernet training entry point — same CLI as the reference's tools/train_supernet.py:36-96 and
the same sequence (:99-214): Config -> cfg-options -> dist init -> work_dir / logger / seed ->
build_segmentor -> build_model_sampler x2 -> dataset -> train_segmentor.

Launch one process per GPU, e.g.
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \
        tools/train_supernet.py configs/supernet/pspnet_ar50to101v2.py --launcher pytorch
"""
import argparse
import logging
import os
import os.path as osp
import sys
import time

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from gaia_seg_amd import __version__  # noqa: E402
from gaia_seg_amd.apis import set_random_seed, train_segmentor  # noqa: E402
from gaia_seg_amd.core.config import Config, DictAction  # noqa: E402
from gaia_seg_amd.core.model_space import build_model_sampler  # noqa: E402
from gaia_seg_amd.models import build_segmentor  # noqa: E402


def parse_args():
    parser = argparse.ArgumentParser(description="Train a segmentor supernet")
    parser.add_argument("config", help="train config file path")
    parser.add_argument("--work-dir", help="the dir to save logs and models")
    parser.add_argument("--load-from", help="the checkpoint file to load weights from")
    parser.add_argument("--resume-from", help="the checkpoint file to resume from")
    parser.add_argument("--no-validate", action="store_true",
                        help="whether not to evaluate the checkpoint during training")
    group_gpus = parser.add_mutually_exclusive_group()
    group_gpus.add_argument("--gpus", type=int, help="number of gpus to use (non-distributed)")
    group_gpus.add_argument("--gpu-ids", type=int, nargs="+", help="ids of gpus to use")
    parser.add_argument("--seed", type=int, default=None, help="random seed")
    parser.add_argument("--deterministic", action="store_true")
    parser.add_argument("--options", nargs="+", default=None, help="custom options (deprecated)")
    parser.add_argument("--cfg-options", nargs="+", default=None,
                        help="override settings in the config, key=value pairs")
    parser.add_argument("--launcher", choices=["none", "pytorch", "slurm", "mpi"], default="none")
    parser.add_argument("--local_rank", "--local-rank", type=int, default=0)
    parser.add_argument("--max-iters", type=int, default=None, help="override runner.max_iters")
    args = parser.parse_args()
    if "LOCAL_RANK" not in os.environ:
        os.environ["LOCAL_RANK"] = str(args.local_rank)
    return args


def init_dist(launcher, backend="nccl"):
    if launcher != "pytorch":
        raise NotImplementedError("launcher %s: use torch.distributed.run (pytorch)" % launcher)
    local_rank = int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    pg_opts = None
    if backend == "nccl":   # RCCL's streams at the training stream's priority (core/runner.py TRAIN_PRIORITY)
        try:
            pg_opts = dist.ProcessGroupNCCL.Options()
            pg_opts.is_high_priority_stream = os.environ.get("GS_TRAIN_PRIORITY", "1") != "0"
        except (AttributeError, TypeError):
            pg_opts = None
    if pg_opts is not None:
        dist.init_process_group(backend=backend, device_id=torch.device("cuda", local_rank), pg_options=pg_opts)
    else:
        dist.init_process_group(backend=backend, device_id=torch.device("cuda", local_rank))


def get_root_logger(log_file=None, log_level=logging.INFO):
    logger = logging.getLogger("gaia_seg_amd")
    if not logger.handlers:
        rank = dist.get_rank() if dist.is_initialized() else 0
        handlers = [logging.StreamHandler()]
        if rank == 0 and log_file is not None:
            handlers.append(logging.FileHandler(log_file, "w"))
        fmt = logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s")
        for h in handlers:
            h.setFormatter(fmt)
            logger.addHandler(h)
        logger.setLevel(log_level if rank == 0 else logging.ERROR)
    return logger


def main():
    args = parse_args()
    cfg = Config.fromfile(args.config)
    options = DictAction.parse(args.cfg_options or args.options)
    if options:
        cfg.merge_from_dict(options)
    if args.max_iters is not None:
        cfg.merge_from_dict({"runner.max_iters": args.max_iters})
    if args.work_dir is not None:
        cfg.work_dir = args.work_dir
    elif cfg.get("work_dir", None) is None:
        cfg.work_dir = osp.join("./work_dirs", osp.splitext(osp.basename(args.config))[0])
    if args.load_from is not None:
        cfg.load_from = args.load_from
    if args.resume_from is not None:
        cfg.resume_from = args.resume_from
    cfg.gpu_ids = range(1) if args.gpus is None and args.gpu_ids is None else (
        args.gpu_ids if args.gpu_ids is not None else range(args.gpus))

    if args.launcher == "none":
        distributed = False
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
    else:
        distributed = True
        init_dist(args.launcher, **dict(cfg.get("dist_params") or dict(backend="nccl")))

    os.makedirs(osp.abspath(cfg.work_dir), exist_ok=True)
    cfg.dump(osp.join(cfg.work_dir, osp.basename(args.config)))
    timestamp = time.strftime("%Y%m%d_%H%M%S", time.localtime())
    logger = get_root_logger(osp.join(cfg.work_dir, "%s.log" % timestamp), cfg.get("log_level", "INFO"))
    meta = dict(env_info="gaia_seg_amd %s, torch %s, hip %s" % (__version__, torch.__version__,
                                                              torch.version.hip))
    logger.info("Distributed training: %s" % distributed)
    logger.info("Config:\n%s" % cfg.pretty_text)
    if args.seed is not None:
        logger.info("Set random seed to %s, deterministic: %s" % (args.seed, args.deterministic))
        set_random_seed(args.seed, deterministic=args.deterministic)
    cfg.seed = args.seed
    meta["seed"] = args.seed
    meta["exp_name"] = osp.basename(args.config)

    model = build_segmentor(cfg.model, train_cfg=cfg.get("train_cfg"), test_cfg=cfg.get("test_cfg"))
    logger.info("parameters: %.2f M" % (sum(p.numel() for p in model.parameters()) / 1e6))
    train_sampler = build_model_sampler(cfg.train_sampler)
    val_sampler = build_model_sampler(cfg.val_sampler)
    if args.seed is not None:
        train_sampler.seed(args.seed)
    meta.update(version=__version__, config=cfg.pretty_text,
                CLASSES=tuple("class_%d" % i for i in range(model.num_classes)))
    model.CLASSES = meta["CLASSES"]
    runner = train_segmentor(model, train_sampler, val_sampler, cfg.data["train"], cfg,
                             distributed=distributed, validate=(not args.no_validate),
                             timestamp=timestamp, meta=meta, logger=logger)
    logger.info("finished %d iterations" % runner.iter)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
