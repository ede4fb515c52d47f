#!/usr/bin/env python
"""Headline benchmark: supernet train images/sec at 1024x512 (BASELINE.json config 2):
FCN decode head + aux FCN head on the dynamic R50..R101 supernet, bs 2 per GPU, one randomly sampled
subnet per step (the reference's train sampler), fp32, synthetic data, random-init weights.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

A step = sample arch -> zero grads of the active ranges -> forward (HIP kernels) -> backward
(HIP kernels) with the bucketed RCCL all-reduce of the active gradient ranges overlapped ->
fused SGD(momentum, weight decay) with poly LR.  Nothing is skipped inside the timed region.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     : the dynamic 3x3 bottleneck conv forward, timed live with HIP events on
                 the launch stream inside the timed steps: achieved = sum(2*M*N*K FLOPs) / sum(time)
                 against the fp32 MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md) — fp32 because the
                 reference path is fp32 and the parity bar is 1e-3 rel fp32;
  cpu_baseline : the CPU oracle (oracle/model.py, PyTorch-CPU) timed on the host cores on a bounded
                 sample of the same workload (N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--config", default=os.path.join(ROOT, "configs/supernet/fcn_ar50to101v2.py"))
    ap.add_argument("--arch", default="sample",
                    help="'sample' = one subnet per step from the train sampler (config of record); "
                         "or an anchor name: MAX MIN R50 R77 R101")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-size", default="512x1024",
                    help="HxW of the bounded CPU sample (bs 1, R50 anchor)")
    ap.add_argument("--no-k3-timer", action="store_true")
    ap.add_argument("--crop", default=None,
                    help="HxW override of the crop size (diagnostics only, e.g. 64x128 makes the GPU "
                         "work negligible and exposes the host cost per step; the headline "
                         "number is always the config's 512x1024)")
    return ap.parse_args()


def host_cores():
    """CPU cores this process may actually use: the cgroup quota if there is one (the GPU box
    gives 16 of its 256 hardware threads), else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def k3_traffic():
    """HBM bytes per K3 launch from the PMC passes (FETCH_SIZE x2 + WRITE_SIZE, separate
    rocprofv3 --pmc runs of this script; see profiles/r01_k3_traffic.json).  Counters cannot be
    read from inside the process, so this is the committed measurement, or None."""
    path = os.path.join(ROOT, "profiles", "r01_k3_traffic.json")
    try:
        with open(path) as f:
            return json.load(f)["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(cfg, size, seed):
    """One fwd+bwd+SGD step of the oracle on the host cores: a bounded sample (bs 1, R50 anchor)."""
    import torch
    from gaia_seg_amd.core.dynamic import fold_dict
    from gaia_seg_amd.core.synthetic import make_batch
    from oracle.model import OEncoderDecoder
    h, w = size
    torch.manual_seed(seed)
    model_cfg = {k: v for k, v in cfg.model.to_dict().items() if k != "type"} \
        if hasattr(cfg.model, "to_dict") else {k: v for k, v in dict(cfg.model).items() if k != "type"}
    orc = OEncoderDecoder(**model_cfg).train()
    r50 = {"arch.backbone.stem.width": 64, "arch.backbone.body.width": [64, 128, 256, 512],
           "arch.backbone.body.depth": [3, 4, 6, 3]}
    orc.manipulate_arch(fold_dict(r50)["arch"])
    opt = torch.optim.SGD(orc.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4)
    batch = make_batch(1, h, w, seed=seed)
    cores = host_cores()
    torch.set_num_threads(cores)

    def step():
        opt.zero_grad(set_to_none=True)
        loss, _ = orc.parse_losses(orc.forward_train(batch["img"], batch["gt_semantic_seg"]))
        loss.backward()
        opt.step()
    step()  # warm-up (oneDNN primitive creation)
    t0 = time.time()
    n = 0
    while True:
        step()
        n += 1
        if time.time() - t0 > 8.0 or n >= 3:
            break
    dt = (time.time() - t0) / n
    # images/sec scaled to the benchmark resolution by pixel count (conv work is linear in pixels)
    scale = (h * w) / (512.0 * 1024.0)
    return dict(value=round(scale / dt, 4), unit="images/sec", cores=cores, kind="port",
                sample="oracle (PyTorch-CPU fp32) fwd+bwd+SGD, R50 anchor, bs 1 at %dx%d, %d timed "
                       "step(s) of %.2f s, scaled by pixel count to 512x1024" % (h, w, n, dt))


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs MI355X GPUs (the HIP path has no CPU fallback)")
    # GS_BENCH_BACKEND=gloo rehearses the multi-rank code path on a box with fewer GPUs than ranks
    # (ranks share devices, collectives go through the host); the measured path is always RCCL
    backend = os.environ.get("GS_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d (using WORLD_SIZE)" % (args.gpus, world),
              file=sys.stderr)

    import random
    from gaia_seg_amd.core import dist as gdist
    from gaia_seg_amd.core.config import Config
    from gaia_seg_amd.core.model_space import build_model_sampler
    from gaia_seg_amd.core.param_arena import ParamArena
    from gaia_seg_amd.core.runner import (ArenaOptimizerHook, IterBasedRunner, ManipulateArchHook,
                                          PolyLrUpdaterHook)
    from gaia_seg_amd.core.synthetic import SyntheticLoader
    from gaia_seg_amd.hip import lib, ops
    from gaia_seg_amd.models import build_segmentor

    lib.load()
    cfg = Config.fromfile(args.config)
    torch.manual_seed(args.seed)
    random.seed(args.seed)
    model = build_segmentor(cfg.model, train_cfg=cfg.get("train_cfg"), test_cfg=cfg.get("test_cfg"))
    model = model.to(dev).train()
    arena = ParamArena(model)
    reducer = gdist.GradReducer(arena.flat_grad, arena.segments)
    opt = cfg.optimizer
    runner = IterBasedRunner(model, arena, reducer, base_lr=opt["lr"], momentum=opt["momentum"],
                             weight_decay=opt["weight_decay"], max_iters=cfg.runner["max_iters"])
    sampler = build_model_sampler(cfg.train_sampler)
    sampler.seed(args.seed)
    if args.arch == "sample":
        runner.register_hook(ManipulateArchHook(sampler))
    else:
        anchors = {a["name"]: a for a in sampler.model_samplers[0].anchors}
        runner.set_arch(anchors[args.arch])
    lrc = dict(cfg.lr_config)
    lrc.pop("policy", None)
    runner.register_hook(PolyLrUpdaterHook(**lrc))
    runner.register_hook(ArenaOptimizerHook())
    runner.call_hook("before_run")

    bs = cfg.data["samples_per_gpu"]
    size = tuple(cfg.crop_size)
    if args.crop:
        size = tuple(int(v) for v in args.crop.split("x"))
    loader = SyntheticLoader(bs, size, num_classes=19, seed=args.seed, rank=rank, device=dev)

    for _ in range(args.warmup):
        runner.train_iter(next(loader))
    torch.cuda.synchronize()

    if runner.host_prof is not None:
        runner.host_prof.clear()
    # the timed steps always see draws 1..K of the seeded train sampler, whatever the warm-up
    # length was (the subnet mix decides the step time: the number must not depend on --warmup)
    sampler.seed(args.seed)
    timer = None
    if not args.no_k3_timer:
        # HIP events on the launch stream around every bottleneck-conv2 forward (conv + its split-K
        # reduce), recorded inside the library (gs_k3_timer_*, include/gaiaseg_hip.h)
        timer = lib.load()
        lib.check(timer.gs_k3_timer_enable(1), "gs_k3_timer_enable")
    arch_log_start = runner.iter
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    prof = None
    if os.environ.get("GS_CPROFILE"):   # diagnostics: host profile of the timed loop (main thread)
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        runner.train_iter(next(loader))
    if prof is not None:
        prof.disable()
        import pstats
        pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(40)
        from gaia_seg_amd.hip import runtime as _rt
        if _rt.BACKWARD_PROFILE is not None:
            print("---- backward thread ----", file=sys.stderr)
            pstats.Stats(_rt.BACKWARD_PROFILE, stream=sys.stderr).sort_stats("tottime").print_stats(30)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if timer is not None:
        lib.check(timer.gs_k3_timer_enable(0), "gs_k3_timer_enable")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if runner.host_prof:
        hp = runner.host_prof
        n = max(hp.pop("iters", 1), 1)
        print("host ms/iter: " + ", ".join("%s %.2f" % (k, 1e3 * v / n) for k, v in hp.items()),
              file=sys.stderr)
    loss = float(runner.outputs["loss"].detach())
    if rank == 0:
        imgs = world * bs * args.steps
        out = {
            "metric": "supernet train images/sec at 1024x512",
            "value": round(imgs / elapsed, 3),
            "unit": "images/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "fp32",
            "data": "synthetic",
            "config": {
                "workload": "%s + aux FCN on the dynamic R50..R101 ResNet supernet (%s), %dx%d "
                            "crops, bs %d/GPU, arch=%s, fwd+bwd+SGD, random-init weights"
                            % (cfg.model["decode_head"]["type"], os.path.basename(args.config),
                               size[1], size[0], bs, args.arch),
                "global_batch": world * bs,
                "parallelism": "dp%d" % world,
                "last_loss": round(loss, 5),
            },
        }
        hooks = [h for h in runner.hooks if isinstance(h, ManipulateArchHook)]
        if hooks:
            out["config"]["archs"] = hooks[0].history[arch_log_start:]
        if timer is not None:
            import ctypes
            c_n, c_ms, c_fl = ctypes.c_int64(0), ctypes.c_double(0.0), ctypes.c_double(0.0)
            lib.check(timer.gs_k3_timer_read(ctypes.byref(c_n), ctypes.byref(c_ms), ctypes.byref(c_fl)),
                      "gs_k3_timer_read")
            if c_n.value:
                launches, ms, flops = c_n.value, c_ms.value, c_fl.value
                achieved = flops / (ms * 1e-3) / 1e12
                out["roofline"] = {
                    "kernel": "igemm_rows_fast_kernel<BM,BN,false,3,0,1> + splitk_reduce_kernel<false,1> (dynamic 3x3 bottleneck conv "
                              "forward, incl. its split-K reduce where used)",
                    "bound": "mfma", "achieved": round(achieved, 2),
                    "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": k3_traffic(),
                    "launches": launches, "avg_launch_us": round(1e3 * ms / launches, 2),
                    "algorithmic_gflop_per_launch": round(flops / launches / 1e9, 3),
                }
        if world == 1 and not args.no_cpu_baseline:
            h, w = (int(v) for v in args.cpu_baseline_size.split("x"))
            out["cpu_baseline"] = cpu_baseline(cfg, (h, w), args.seed)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
