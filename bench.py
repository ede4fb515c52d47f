#!/usr/bin/env python
"""Headline benchmark: supernet train images/sec at 1024x512 (BASELINE.json config 2):
FCN decode head + aux FCN head on the dynamic R50..R101 supernet, bs 2 per GPU, one randomly sampled
subnet per step (the reference's train sampler), fp32, synthetic data, random-init weights.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

A step = sample arch -> zero grads of the active ranges -> forward (HIP kernels) -> backward
(HIP kernels) with the bucketed RCCL all-reduce of the active gradient ranges overlapped ->
fused SGD(momentum, weight decay) with poly LR.  Nothing is skipped inside the timed region.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process only launches
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD (before anything touches
the GPU) and exits with its code; the ranks are one process per GPU over RCCL.

Rank 0 prints ONE JSON line.  `value` comes from K un-instrumented steps.  Besides the contract
fields the line carries
  check        : before any timing, the first timed step's subnet and batch are run once from the
                 initial weights (dropout off: device RNG streams cannot match the CPU) and the
                 losses compared with the committed oracle values (tests/golden/bench_check.json,
                 made by tests/golden/make_bench_check.py); a mismatch > 1e-3 aborts the benchmark;
  roofline     : the dynamic 3x3 bottleneck conv forward, timed with HIP events on the
                 launch stream in a SEPARATE instrumented pass over the same K draws (so the
                 instrumentation never sits inside `value`): achieved = sum(2*M*N*K FLOPs) / sum(time)
                 against the fp32 MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md) — fp32 because the
                 reference path is fp32 and the parity bar is 1e-3 rel fp32;
  roofline_step: algorithmic conv FLOPs of the K timed steps (fwd + dgrad + wgrad) / their wall time;
  cpu_baseline : the CPU oracle (oracle/model.py, PyTorch-CPU) timed on the host cores on a bounded
                 sample of the same workload (N=1 only): median of >= 5 steps.
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
# The bf16x3 K loop (six bf16 MFMAs per fp32 product, csrc/igemm_core.h x3_k_loop) is bounded by LDS
# bandwidth, not by the MFMA pipe: 6 B per operand element -> 213 fp32-equivalent TFLOP/s at 100 % of the
# LDS read rate (profiles/r02_bf16x3_probe.md).  A launch on that loop is priced against this figure.
X3_LDS_BOUND_TFLOPS = 213.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--config", default=os.path.join(ROOT, "configs/supernet/fcn_ar50to101v2.py"))
    ap.add_argument("--arch", default="sample",
                    help="'sample' = one subnet per step from the train sampler (config of record); "
                         "or an anchor name: MAX MIN R50 R77 R101")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--repeats", type=int, default=5,
                    help="the un-instrumented K-step pass is repeated this many times over the SAME K "
                         "draws; `value` / `ms_per_step` are the MEDIAN pass, min / max go to `config`")
    ap.add_argument("--data", default="synthetic", choices=["synthetic", "pipeline"],
                    help="'pipeline' (diagnostics): every timed step's batch is produced by the GPU input "
                         "pipeline (gs_seg_augment, datasets/gpu_pipeline.py) from resident uint8 "
                         "2048x1024 images, so the input stage's cost per step is on record")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-size", default="512x1024",
                    help="HxW of the bounded CPU sample (config batch size, R50 anchor)")
    ap.add_argument("--no-check", action="store_true",
                    help="skip the oracle check of the first step (diagnostics only)")
    ap.add_argument("--no-k3-timer", action="store_true")
    ap.add_argument("--step-graphs", action="store_true",
                    help="replay recurring subnets' training step from a captured HIP graph "
                         "(IterBasedRunner.train_iter; off by default: slower on ROCm 7.2, see "
                         "DESIGN.md); diagnostics / A-B only")
    ap.add_argument("--plan-only", action="store_true",
                    help="no GPU work: print the data-parallel exchange plan of --gpus N ranks (gradient "
                         "buckets, bytes and collective launches per step for MIN / R50 / MAX and the "
                         "seeded --steps-draw mix, SyncBN collectives per step, expected all-reduce time "
                         "at a stated bus bandwidth) as one JSON object and exit")
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
    rocprofv3 --pmc runs of this script on the sampled mix; tools/pmc_summary.py hbm ->
    profiles/rNN_k3_traffic.json).  Counters cannot be read from inside the process, so this is the
    latest committed measurement, or None."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_k3_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                return json.load(f)["hbm_bytes_per_launch"], os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def _plain(obj):
    if isinstance(obj, dict):
        return {k: _plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_plain(v) for v in obj]
    return obj


R50 = {"arch.backbone.stem.width": 64, "arch.backbone.body.width": [64, 128, 256, 512],
       "arch.backbone.body.depth": [3, 4, 6, 3]}


def cpu_baseline(cfg, size, seed, bs):
    """fwd+bwd+SGD steps of the oracle on the host cores: a bounded sample of the workload (the
    config's batch size and crop, R50 anchor instead of the sampled mix), median of >= 5 steps."""
    import statistics
    import torch
    from gaia_seg_amd.core.dynamic import fold_dict
    from gaia_seg_amd.core.synthetic import make_batch
    from oracle.model import OEncoderDecoder
    h, w = size
    torch.manual_seed(seed)
    orc = OEncoderDecoder(**{k: v for k, v in _plain(cfg.model).items() if k != "type"}).train()
    orc.manipulate_arch(fold_dict(R50)["arch"])
    opt = torch.optim.SGD(orc.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4)
    batch = make_batch(bs, h, w, seed=seed)
    cores = host_cores()
    torch.set_num_threads(cores)

    def step():
        opt.zero_grad(set_to_none=True)
        loss, _ = orc.parse_losses(orc.forward_train(batch["img"], batch["gt_semantic_seg"]))
        loss.backward()
        opt.step()
    for _ in range(2):   # warm-up (oneDNN primitive creation, allocator)
        step()
    times = []
    t_all = time.time()
    while len(times) < 5 or (time.time() - t_all < 12.0 and len(times) < 9):
        t0 = time.time()
        step()
        times.append(time.time() - t0)
    dt = statistics.median(times)
    scale = (h * w) / (512.0 * 1024.0)   # conv work is linear in pixels (1.0 at the default size)
    return dict(value=round(bs * scale / dt, 4), unit="images/sec", cores=cores, kind="port",
                sample="oracle (PyTorch-CPU fp32, oneDNN) fwd+bwd+SGD, R50 anchor, bs %d at %dx%d: "
                       "median %.2f s/step over %d timed steps after 2 warm-up steps"
                       % (bs, h, w, dt, len(times)))


class PipelineLoader:
    """--data pipeline: every batch comes out of the GPU input pipeline (Resize(ratio 0.5-2) ->
    RandomCrop -> RandomFlip -> PhotoMetricDistortion -> Normalize -> Pad, one gs_seg_augment launch
    per sample) from a pool of resident uint8 2048x1024 "decoded" images (Cityscapes' size), on the
    training stream, inside the timed step."""

    def __init__(self, bs, size, seed, rank, dev, pool=6):
        import torch
        from gaia_seg_amd.datasets.gpu_pipeline import GpuTrainPipeline
        g = torch.Generator().manual_seed(seed * 7919 + rank)
        self.samples = []
        for _ in range(pool):
            img = torch.randint(0, 256, (1024, 2048, 3), generator=g, dtype=torch.uint8).to(dev)
            lab = torch.randint(0, 19, (1024, 2048), generator=g, dtype=torch.uint8).to(dev)
            self.samples.append((img, lab))
        self.pipe = GpuTrainPipeline(crop_size=size, seed=seed + rank, device=dev, cat_max_ratio=1.0)
        self.bs, self.i = bs, 0

    def __iter__(self):
        return self

    def __next__(self):
        n = len(self.samples)
        batch = self.pipe.batch([self.samples[(self.i + j) % n] for j in range(self.bs)])
        self.i += self.bs
        return batch


def launch_ranks(n):
    """--gpus N without a launcher: start N ranks as child processes (one per GPU, RCCL) and exit
    with their code.  Nothing in this parent touches the GPU (a process that has initialised HIP
    must never exec another program on this pool)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)]
    cmd += sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def first_step_check(model, cfg_path, seed, size, bs, arch_meta, batch, dev, strict=False):
    """Run the first timed step's subnet and batch once from the initial weights with dropout off and
    compare the losses with the committed oracle values.  Returns the `check` entry; raises on a
    mismatch.  strict (the default configuration, seed and crop): a missing or stale fixture is an
    error too, so the gate cannot disappear silently; diagnostic overrides report "skipped".  (The BN running statistics it moves and the gradients it leaves are part of the
    warm-up state; nothing of it is timed.)"""
    import torch
    from gaia_seg_amd.core.dynamic import fold_dict
    key = "%s|seed%d|%dx%d|bs%d|%s" % (os.path.basename(cfg_path), seed, size[0], size[1], bs,
                                       arch_meta.get("name", "random"))
    path = os.path.join(ROOT, "tests", "golden", "bench_check.json")
    try:
        with open(path) as f:
            gold = json.load(f).get(key)
    except (OSError, ValueError):
        gold = None
    if gold is None:
        if strict:
            raise SystemExit("bench.py: no committed oracle value for the default configuration (%s); "
                             "regenerate tests/golden/bench_check.json or pass --no-check" % key)
        return dict(status="skipped", reason="no committed oracle value for %s" % key)
    csum = float(sum(p.detach().double().abs().sum() for p in model.parameters()))
    if abs(csum - gold["param_abs_sum"]) > 1e-9 * gold["param_abs_sum"]:
        msg = ("initial weights differ from the fixture's (init order / RNG stream / layout changed): "
               "abs-sum %.9g vs %.9g" % (csum, gold["param_abs_sum"]))
        if strict:
            raise SystemExit("bench.py: %s; regenerate tests/golden/bench_check.json (tests/golden/"
                             "make_bench_check.py) or pass --no-check" % msg)
        return dict(status="skipped", reason=msg)
    heads = [h for h in (model.decode_head, getattr(model, "auxiliary_head", None)) if h is not None]
    saved_h = [h.dropout for h in heads]
    for h in heads:
        h.dropout = None
    try:
        model.manipulate_arch(fold_dict(arch_meta)["arch"])
        out = model.train_step(batch, None)
        out["loss"].backward()
        torch.cuda.synchronize()
    finally:
        for h, d in zip(heads, saved_h):
            h.dropout = d
    got = {k: float(v) for k, v in out["log_vars"].items()}
    worst = 0.0
    for k, want in gold["log_vars"].items():
        if k.endswith("acc_seg"):
            continue
        worst = max(worst, abs(got[k] - want) / max(abs(want), 1e-12))
    res = dict(status="ok", what="first timed step's subnet (%s) and batch from the initial weights, "
               "dropout off, vs committed CPU-oracle losses (tests/golden/bench_check.json)"
               % arch_meta.get("name", "random"),
               hip_loss=round(got["loss"], 6), oracle_loss=round(gold["log_vars"]["loss"], 6),
               max_rel_err=float("%.3g" % worst), tol=1e-3)
    if not worst < 1e-3:
        raise SystemExit("bench.py: the HIP path's losses differ from the oracle's on the first step: "
                         "%s vs %s" % (got, gold["log_vars"]))
    return res


def step_flops(model, sampler, seed, steps, size, bs, fixed_meta):
    """Algorithmic conv FLOPs of the timed steps (forward + dgrad + wgrad = 3x forward, the stem
    conv has no dgrad) and the K3 launches' algorithmic FLOPs / bytes: replays the seeded draws on the
    host after the timing."""
    from gaia_seg_amd.core.dynamic import fold_dict
    from gaia_seg_amd.core.flops import _conv, model_flops
    total = 0.0
    k3 = dict(launches=0, flops=0.0, bytes=0.0)
    if fixed_meta is None:
        sampler.seed(seed)
    for _ in range(steps):
        meta = fixed_meta if fixed_meta is not None else sampler.sample()
        model.manipulate_arch(fold_dict(meta)["arch"])
        f = model_flops(model, size[0], size[1])
        b = model.backbone
        if b.deep_stem:
            first = _conv(b.stem[0], 3, size[0], size[1])
        else:
            first = _conv(b.conv1, 3, size[0], size[1])
        total += bs * (3.0 * f["total"] - first[0])   # the very first conv has no data gradient
        # K3 = conv2 of every active bottleneck: algorithmic bytes x + W + y once
        c, h, w = first[2], first[3], first[4]
        if b.deep_stem:
            for i in (3, 6):
                _, _, c, h, w = _conv(b.stem[i], c, h, w)
        mp = b.maxpool
        h = (h + 2 * mp.padding - mp.kernel_size) // mp.stride + 1
        w = (w + 2 * mp.padding - mp.kernel_size) // mp.stride + 1
        for name in b.res_layers:
            for blk in getattr(b, name).active_blocks():
                _, _, c1, h1, w1 = _conv(blk.conv1, c, h, w)
                f2, _, c2, h2, w2 = _conv(blk.conv2, c1, h1, w1)
                _, _, c, h, w = _conv(blk.conv3, c2, h2, w2)
                k3["launches"] += 1
                k3["flops"] += bs * f2
                k3["bytes"] += 4.0 * (bs * h1 * w1 * c1 + 9 * c1 * c2 + bs * h2 * w2 * c2)
    return total, k3


# Ring all-reduce over xGMI: every GPU has 7 links x ~153 GB/s (one per peer); a ring uses one link per
# direction and hop, RCCL runs several rings / a tree over the mesh.  The figure below is the ASSUMED
# large-message bus bandwidth the plan prices the exchange with (RCCL all-reduce, 8 GPUs, >= 64 MB
# messages); replace it with the measured one once a node is available.
PLAN_BUS_GBPS = 300.0


def plan_only(args):
    """--plan-only: what the data-parallel exchange of one step looks like at --gpus N, from host
    arithmetic alone (the bucket planner of core/dist.py over the arena layout)."""
    import torch
    from gaia_seg_amd.core.config import Config
    from gaia_seg_amd.core.dist import GradReducer
    from gaia_seg_amd.core.dynamic import fold_dict
    from gaia_seg_amd.core.model_space import build_model_sampler
    from gaia_seg_amd.core.param_arena import _ALIGN, arena_layout
    from gaia_seg_amd.core.bricks import DynamicBatchNorm2d
    from gaia_seg_amd.hip.runtime import round_up
    from gaia_seg_amd.models import build_segmentor
    cfg = Config.fromfile(args.config)
    torch.manual_seed(args.seed)
    model = build_segmentor(cfg.model, train_cfg=cfg.get("train_cfg"), test_cfg=cfg.get("test_cfg")).train()
    layout, total = arena_layout(model)
    segments = {id(p_): (o, round_up(max(n, 1), _ALIGN)) for _, p_, _, o, n in layout}
    reducer = GradReducer(torch.zeros(1), segments)      # planner only: nothing is reduced
    n = max(args.gpus, 1)
    sampler = build_model_sampler(cfg.train_sampler)
    anchors = {a["name"]: dict(a) for a in sampler.model_samplers[0].anchors}

    def one(meta):
        model.manipulate_arch(fold_dict(meta)["arch"])
        params = [p_ for p_ in model.active_parameters() if p_.requires_grad]
        plan = reducer._plan(params, None)
        runs = [r for b in plan for r in b["runs"]]
        nbytes = 4 * sum(b_ - a_ for a_, b_ in runs)
        payload = 4 * sum(segments[id(p_)][1] for p_ in params)
        sync = [m for m in model.modules() if isinstance(m, DynamicBatchNorm2d) and m.sync is not None]
        t_ms = 2.0 * (n - 1) / n * nbytes / (PLAN_BUS_GBPS * 1e9) * 1e3 if n > 1 else 0.0
        return dict(buckets=len(plan), collective_launches=len(plan),
                    runs=len(runs), allreduce_bytes=nbytes, active_parameter_bytes=payload,
                    padded_hole_bytes=nbytes - payload,
                    bucket_bytes=[4 * sum(b_ - a_ for a_, b_ in b["runs"]) for b in plan],
                    syncbn_layers=len(sync), syncbn_collectives=2 * len(sync),
                    expected_allreduce_ms=round(t_ms, 3))
    out = {"plan_only": True, "n_gpus": n, "config": os.path.basename(args.config),
           "arena_bytes": 4 * total, "bucket_cap_bytes": 4 * reducer.bucket_elems,
           "assumed_bus_bandwidth_GBps": PLAN_BUS_GBPS,
           "expected_allreduce_ms_is": "2 (N-1)/N x bytes / bus bandwidth: ring all-reduce at the assumed "
                                       "large-message bus bandwidth; buckets are issued during backward on "
                                       "the weight-gradient stream, so all but the last bucket overlap it",
           "per_step": {}}
    for name in ("MIN", "R50", "MAX"):
        if name in anchors:
            out["per_step"][name] = one(anchors[name])
    sampler.seed(args.seed)
    mix = [one(sampler.sample()) for _ in range(args.steps)]
    keys = ("buckets", "collective_launches", "allreduce_bytes", "active_parameter_bytes",
            "syncbn_collectives", "expected_allreduce_ms")
    out["per_step"]["mix_of_%d_draws_mean" % args.steps] = {
        k: round(sum(m[k] for m in mix) / len(mix), 3) for k in keys}
    print(json.dumps(out))


def main():
    args = parse_args()
    if args.plan_only:
        return plan_only(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus)   # never returns
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
    elif world > torch.cuda.device_count():
        raise SystemExit("bench.py: %d ranks but %d GPU(s) visible (RCCL needs one GPU per rank)"
                         % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            # the training step runs on a high-priority stream (core/runner.py TRAIN_PRIORITY): RCCL's own
            # streams get the same priority, so that a bucket's all-reduce is dispatched when it is issued
            # and not only when the compute queue runs dry
            pg_opts = None
            try:
                pg_opts = dist.ProcessGroupNCCL.Options()
                pg_opts.is_high_priority_stream = os.environ.get("GS_TRAIN_PRIORITY", "1") != "0"
            except (AttributeError, TypeError):
                pg_opts = None
            if pg_opts is not None:
                dist.init_process_group("nccl", device_id=dev, pg_options=pg_opts)
            else:
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
    from gaia_seg_amd.core.synthetic import SyntheticLoader, make_batch
    from gaia_seg_amd.hip import lib
    from gaia_seg_amd.models import build_segmentor

    L = lib.load()
    cfg = Config.fromfile(args.config)
    torch.manual_seed(args.seed)
    random.seed(args.seed)
    model = build_segmentor(cfg.model, train_cfg=cfg.get("train_cfg"), test_cfg=cfg.get("test_cfg"))
    model = model.to(dev).train()
    arena = ParamArena(model)
    gdist.sync_module_states(model, arena)   # the DDP wrap-time broadcast (gaiaseg/apis/train.py:88-96)
    reducer = gdist.GradReducer(arena.flat_grad, arena.segments)
    opt = cfg.optimizer
    runner = IterBasedRunner(model, arena, reducer, base_lr=opt["lr"], momentum=opt["momentum"],
                             weight_decay=opt["weight_decay"], max_iters=cfg.runner["max_iters"])
    sampler = build_model_sampler(cfg.train_sampler)
    sampler.seed(args.seed)
    fixed_meta = None
    if args.arch != "sample":
        anchors = {a["name"]: a for a in sampler.model_samplers[0].anchors}
        fixed_meta = dict(anchors[args.arch])
        if cfg.get("stem_anchors"):   # deep-stem (v1c / OS8) supernet: a width per stem conv
            fixed_meta["arch.backbone.stem.width"] = list(cfg["stem_anchors"][args.arch])
    elif cfg.model["backbone"].get("deep_stem"):
        raise SystemExit("bench.py: the deep-stem config needs --arch <anchor> (the in-tree train sampler "
                         "draws one stem width, the deep stem takes three)")

    data_cfg = cfg.get("data") or {}
    bs = data_cfg.get("samples_per_gpu", 2)
    size = tuple(cfg.get("crop_size") or (512, 1024))
    if args.crop:
        size = tuple(int(v) for v in args.crop.split("x"))

    # ---- parity gate: the first timed step's subnet and batch against the committed oracle losses
    check = dict(status="skipped", reason="--no-check")
    if not args.no_check:
        first_meta = fixed_meta if fixed_meta is not None else sampler.sample()
        sampler.seed(args.seed)
        b0 = make_batch(bs, size[0], size[1], 19, args.seed * 1000003, dev)   # rank 0's first batch
        default_cfg = os.path.join(ROOT, "configs/supernet/fcn_ar50to101v2.py")
        strict = (os.path.abspath(args.config) == default_cfg and args.seed == 0 and not args.crop
                  and args.arch in ("sample", "R50"))
        check = first_step_check(model, args.config, args.seed, size, bs, first_meta, b0, dev, strict)
        arena.zero_grad()
        del b0

    if fixed_meta is None:
        runner.register_hook(ManipulateArchHook(sampler))
    else:
        runner.set_arch(fixed_meta)
    lrc = dict(cfg.lr_config)
    lrc.pop("policy", None)
    runner.register_hook(PolyLrUpdaterHook(**lrc))
    runner.register_hook(ArenaOptimizerHook())
    runner.call_hook("before_run")
    if args.data == "pipeline":
        loader = PipelineLoader(bs, size, args.seed, rank, dev)
    else:
        loader = SyntheticLoader(bs, size, num_classes=19, seed=args.seed, rank=rank, device=dev)
    if args.step_graphs:
        runner.graphs_enabled = True
    # Start-up, untimed: capture the HIP graphs of the train sampler's named anchor subnets (they
    # recur; random subnets never do and stay eager).  Each capture is one ordinary training step on
    # a synthetic batch, like a warm-up step.  One rank only: see IterBasedRunner.train_iter.
    graphs_built = 0
    if runner.graphs_enabled and world == 1:
        metas = [fixed_meta] if fixed_meta is not None else list(sampler.model_samplers[0].anchors)
        graphs_built = runner.prepare_graphs(metas, make_batch(bs, size[0], size[1], 19, 12345, dev))

    if os.environ.get("GS_MAIN_PRIORITY"):   # diagnostics: the training stream above the side stream
        hp = torch.cuda.Stream(device=dev, priority=-1)
        hp.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(hp)
    for _ in range(args.warmup):
        runner.train_iter(next(loader))
    torch.cuda.synchronize()

    def timed_pass(instrumented):
        """K steps over draws 1..K of the seeded train sampler (whatever the warm-up length was:
        the subnet mix decides the step time), bracketed by barrier + synchronize."""
        sampler.seed(args.seed)
        if instrumented:
            lib.check(L.gs_k3_timer_enable(1), "gs_k3_timer_enable")
        runner.graphs_paused = bool(instrumented)   # HIP-event timers cannot sit inside a graph
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            runner.train_iter(next(loader))
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if instrumented:
            lib.check(L.gs_k3_timer_enable(0), "gs_k3_timer_enable")
        runner.graphs_paused = False
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    call_prof = None
    if runner.host_prof is not None:
        runner.host_prof.clear()
        call_prof = lib.enable_call_profile()
        for rec in call_prof.values():
            rec[0], rec[1] = 0, 0.0
    arch_log_start = runner.iter
    bytes0 = reducer.bytes_reduced
    reducer.collectives = 0
    prof = None
    if os.environ.get("GS_CPROFILE"):   # diagnostics: host profile of the timed loop (main thread)
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    seg0 = torch.cuda.memory_stats(dev).get("segment.all.allocated", 0)
    import ctypes
    L.gs_debug_conv_launch_flops(None, 1)
    L.gs_debug_k3_flops(None, 1)
    if runner.step_events is not None:
        runner.step_events.clear()
    passes = [timed_pass(False)]                     # <- `value`: nothing but the training steps
    if runner.step_events is not None:
        print("step events (mean ms from step begin, first pass): "
              + ", ".join("%s %.3f" % kv for kv in sorted(runner.step_event_summary().items(),
                                                          key=lambda kv: kv[1])), file=sys.stderr)
        runner.step_events = None
    kl = (ctypes.c_double * 15)()
    L.gs_debug_conv_launch_flops(kl, 1)              # which MFMA path carried the first pass's FLOPs
    k3kl = (ctypes.c_double * 5)()
    L.gs_debug_k3_flops(k3kl, 1)
    seg_new = torch.cuda.memory_stats(dev).get("segment.all.allocated", 0) - seg0
    bytes_per_step = (reducer.bytes_reduced - bytes0) / max(args.steps, 1)
    collectives_per_step = reducer.collectives / max(args.steps, 1) if world > 1 else 0.0
    for _ in range(max(args.repeats, 1) - 1):        # same K draws again (the weights keep training)
        passes.append(timed_pass(False))
    elapsed = sorted(passes)[len(passes) // 2]       # the median pass: exactly K timed steps
    if prof is not None:
        prof.disable()
        import pstats
        pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(40)
        from gaia_seg_amd.hip import runtime as _rt
        if _rt.BACKWARD_PROFILE is not None:
            print("---- backward thread ----", file=sys.stderr)
            pstats.Stats(_rt.BACKWARD_PROFILE, stream=sys.stderr).sort_stats("tottime").print_stats(30)
    arch_log_end = arch_log_start + args.steps
    loss = float(runner.outputs["loss"].detach())
    if runner.host_prof:
        hp = runner.host_prof
        n = max(hp.pop("iters", 1), 1)
        print("host ms/iter: " + ", ".join("%s %.2f" % (k, 1e3 * v / n) for k, v in hp.items()),
              file=sys.stderr)
        if call_prof:
            tot = sum(v[1] for v in call_prof.values())
            cnt = sum(v[0] for v in call_prof.values())
            print("  inside the C-ABI: %.2f ms/iter over %.0f calls/iter; top: " % (1e3 * tot / n, cnt / n)
                  + ", ".join("%s %.0fx%.1fus" % (k[3:], v[0] / n, 1e6 * v[1] / max(v[0], 1))
                              for k, v in sorted(call_prof.items(), key=lambda kv: -kv[1][1])[:12]),
                  file=sys.stderr)
    k3 = None
    if not args.no_k3_timer:
        # separate pass over the same draws with HIP events on the launch stream around every
        # bottleneck-conv2 forward (gs_k3_timer_*, include/gaiaseg_hip.h)
        elapsed_instr = timed_pass(True)
        c_n, c_ms, c_fl = ctypes.c_int64(0), ctypes.c_double(0.0), ctypes.c_double(0.0)
        lib.check(L.gs_k3_timer_read(ctypes.byref(c_n), ctypes.byref(c_ms), ctypes.byref(c_fl)),
                  "gs_k3_timer_read")
        if c_n.value:
            k3 = (c_n.value, c_ms.value, c_fl.value, elapsed_instr)

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
            "data": "synthetic" if args.data == "synthetic" else "synthetic uint8 images through the GPU input pipeline",
            "config": {
                "workload": "%s + aux FCN on the dynamic R50..R101 ResNet supernet (%s), %dx%d "
                            "crops, bs %d/GPU, arch=%s, fwd+bwd+SGD, random-init weights"
                            % (cfg.model["decode_head"]["type"], os.path.basename(args.config),
                               size[1], size[0], bs, args.arch),
                "global_batch": world * bs,
                "parallelism": "dp%d" % world,
                "allreduce_bytes_per_step_per_rank": int(bytes_per_step),
                "allreduce_launches_per_step": round(collectives_per_step, 1),
                "last_loss": round(loss, 5),
                "passes": {"repeats": len(passes), "value_is": "median pass",
                           "images_per_sec_min": round(imgs / max(passes), 3),
                           "images_per_sec_max": round(imgs / min(passes), 3),
                           "ms_per_step_all": [round(1e3 * p / args.steps, 3) for p in passes]},
                "input": ("GPU input pipeline (gs_seg_augment from resident uint8 2048x1024 images) "
                          "inside every timed step" if args.data == "pipeline" else
                          "resident synthetic fp32 batches"),
                "device_mallocs_in_timed_steps": int(seg_new),
                "contraction": "fp32 MFMA (forward, weight gradient); stride-1 data gradients and the "
                               "split-K 3x3 forwards: six bf16 MFMAs over an exact three-way bf16 split "
                               "of both operands, fp32 accumulation (GS_X3=0 / GS_X3_FWD=0 turn them "
                               "off; the unsplit forwards stay on the fp32 MFMA for parity margin: "
                               "DESIGN.md section 11); the measured FLOP shares are in "
                               "roofline.flop_share_by_k_loop and roofline_step.flop_share_by_k_loop",
                "step_graphs": dict(runner.graph_stats, built_at_startup=graphs_built,
                                    what="HIP-graph replay of recurring subnets' whole training "
                                         "step (same kernels as the eager step); counts cover "
                                         "start-up, warm-up and the timed steps"),
            },
            "check": check,
            "distributed": {
                "backend": (dist.get_backend() if world > 1 else None),
                "world_size": (dist.get_world_size() if world > 1 else 1),
                "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version())
                if hasattr(torch.cuda, "nccl") else None,
                "collectives_per_step": round(collectives_per_step, 1),
                "allreduce_bytes_per_step_per_rank": int(bytes_per_step),
                "early_optimizer_steps": runner.early_steps,
                "training_stream": ("high priority (-1); weight-gradient / optimizer streams 0; RCCL streams high"
                                    if runner.TRAIN_PRIORITY else "the caller's stream"),
            },
        }
        hooks = [h for h in runner.hooks if isinstance(h, ManipulateArchHook)]
        if hooks:
            out["config"]["archs"] = hooks[0].history[arch_log_start:arch_log_end]
        fl, k3_alg = step_flops(model, sampler, args.seed, args.steps, size, bs, fixed_meta)
        fl *= world
        # share of the launched contraction FLOPs per K loop (library counters over the first pass)
        kl_tot = sum(kl) or 1.0
        x3_share = sum(kl[o * 5 + 3] for o in range(3)) / kl_tot
        fwd_tot = sum(kl[0:5]) or 1.0
        blended = 1.0 / ((1.0 - x3_share) / FP32_MFMA_PEAK_TFLOPS + x3_share / X3_LDS_BOUND_TFLOPS)
        ach_step = fl / elapsed / 1e12 / world
        out["roofline_step"] = {
            "what": "all convolutions of the %d timed steps, fwd + dgrad + wgrad (BN / pooling / loss "
                    "/ SGD carry no credited FLOPs)" % args.steps,
            "bound": "mfma", "achieved": round(ach_step, 2),
            "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s per GPU",
            "frac": round(ach_step / FP32_MFMA_PEAK_TFLOPS, 4),
            "algorithmic_gflop_per_step_per_gpu": round(fl / args.steps / world / 1e9, 1),
            "flop_share_by_k_loop": {"fp32_mfma": round(1.0 - x3_share, 4), "bf16x3": round(x3_share, 4)},
            "blended_bound": round(blended, 1),
            "frac_of_blended_bound": round(ach_step / blended, 4),
            "bound_note": "`frac` divides fp32-equivalent FLOPs by the fp32 MFMA peak (157.3); the share "
                          "of the FLOPs launched on the bf16x3 loop (stride-1 data gradients) is bounded "
                          "by LDS bandwidth at %.0f fp32-equivalent TFLOP/s instead, so the step's "
                          "bound is the harmonic blend above" % X3_LDS_BOUND_TFLOPS}
        if k3 is not None:
            launches, ms, flops, el_i = k3
            achieved = flops / (ms * 1e-3) / 1e12
            k3_x3 = k3kl[3] / (sum(k3kl) or 1.0)
            k3_bound = 1.0 / ((1.0 - k3_x3) / FP32_MFMA_PEAK_TFLOPS + k3_x3 / X3_LDS_BOUND_TFLOPS)
            traffic, traffic_src = k3_traffic()
            out["roofline"] = {
                "kernel": "igemm_rows_fast_kernel<..., KS=3, ROLE=1> (split-K slabs combined inside the launch): "
                          "the dynamic 3x3 bottleneck conv forward",
                "timed_with": ("marker HIP events recorded on the launch stream in front of and behind each launch "
                               "(GS_K3_TIMER_MARKERS: includes the gap between the markers and the dispatch)"
                               if os.environ.get("GS_K3_TIMER_MARKERS") else
                               "one pair of HIP events per launch, attached to the kernel's own dispatch on its stream "
                               "(hipExtLaunchKernelGGL start / stop events: the dispatch's begin / end timestamps, the "
                               "duration a rocprofv3 kernel trace reports; GS_K3_TIMER_MARKERS=1 gives the r01-r03 "
                               "marker-event figure, 2-3 us more per launch)"),
                "bound": "mfma", "achieved": round(achieved, 2),
                "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                "traffic_source": traffic_src and "%s (separate rocprofv3 --pmc passes of this "
                                                  "command)" % traffic_src,
                "launches": launches, "avg_launch_us": round(1e3 * ms / launches, 2),
                "algorithmic_gflop_per_launch": round(flops / launches / 1e9, 3),
                "algorithmic_bytes_per_launch": int(k3_alg["bytes"] / max(k3_alg["launches"], 1)),
                "algorithmic_check": "host replay of the %d draws: %d launches, %.3f GF per launch"
                                     % (args.steps, k3_alg["launches"],
                                        k3_alg["flops"] / max(k3_alg["launches"], 1) / 1e9),
                "flop_share_by_k_loop": {"fp32_mfma": round(1.0 - k3_x3, 4), "bf16x3": round(k3_x3, 4)},
                "blended_bound": round(k3_bound, 1),
                "frac_of_blended_bound": round(achieved / k3_bound, 4),
                "bound_note": "`frac` = fp32-equivalent FLOPs / the 157.3 TF fp32 MFMA peak; the K3 launches "
                              "dispatched to the bf16x3 loop (the split-K launches of stages 3-4) are bounded by LDS "
                              "bandwidth at %.0f fp32-equivalent TF instead: blended bound above"
                              % X3_LDS_BOUND_TFLOPS,
                "measured_in": "a separate instrumented pass over the same %d draws (%.3f ms/step; "
                               "`value` is from the un-instrumented pass)"
                               % (args.steps, 1e3 * el_i / args.steps),
            }
        if world == 1 and not args.no_cpu_baseline:
            h, w = (int(v) for v in args.cpu_baseline_size.split("x"))
            out["cpu_baseline"] = cpu_baseline(cfg, (h, w), args.seed, bs)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
