"""Two-rank data-parallel training on ONE MI355X (gloo process group over device tensors): the
N>1 code path end to end — arch broadcast, bucketed all-reduce of the active gradient ranges driven
by the backward tape, SyncBN statistics exchange in the heads, 1/world folded into the fused SGD.
RCCL needs one GPU per rank: a two-rank RCCL test runs when two GPUs are visible, and a ONE-rank RCCL
group exercises the backend-specific branch of the reducer (the grouped launch of a bucket's runs)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(head_norm, seed=0, backbone_norm=None):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util_models import fcn_head, model_cfg, psp_head
    from gaia_seg_amd.models import build_segmentor
    cfg = model_cfg(psp_head(), aux=True) if backbone_norm is None else \
        model_cfg(psp_head(), aux=True, norm=backbone_norm)
    for h in ("decode_head", "auxiliary_head"):
        cfg[h]["norm_cfg"] = dict(type=head_norm, requires_grad=True)
    torch.manual_seed(seed)
    return build_segmentor(cfg)


def _runner(model, lr=0.05):
    from gaia_seg_amd.core.dist import GradReducer
    from gaia_seg_amd.core.param_arena import ParamArena
    from gaia_seg_amd.core.runner import ArenaOptimizerHook, IterBasedRunner, ManipulateArchHook
    from gaia_seg_amd.core.model_space import build_model_sampler
    from gaia_seg_amd.core import dist as gdist
    arena = ParamArena(model)
    gdist.sync_module_states(model, arena)
    runner = IterBasedRunner(model, arena, GradReducer(arena.flat_grad, arena.segments, bucket_bytes=1 << 20),
                             base_lr=lr, momentum=0.9, weight_decay=5e-4, max_iters=100)
    sampler = build_model_sampler(dict(type="anchor", anchors=[
        {"name": "sub", "arch.backbone.stem.width": 16, "arch.backbone.body.width": [16, 48, 64, 96],
         "arch.backbone.body.depth": [1, 2, 2, 1]},
        {"name": "max", "arch.backbone.stem.width": 32, "arch.backbone.body.width": [32, 64, 96, 128],
         "arch.backbone.body.depth": [2, 2, 3, 2]}]))
    sampler.seed(7 + (dist.get_rank() if dist.is_initialized() else 0))   # only rank 0's draw counts
    runner.register_hook(ManipulateArchHook(sampler))
    runner.register_hook(ArenaOptimizerHook())
    return runner, arena


def _norm_checksum(model):
    """(sum, abs-sum) over the BatchNorm gamma / beta only: sensitive to their gradients' scale."""
    from torch.nn.modules.batchnorm import _BatchNorm
    s = a = 0.0
    for m in model.modules():
        if isinstance(m, _BatchNorm):
            for p in (m.weight, m.bias):
                s += p.detach().double().sum().item()
                a += (p.detach().double() - (1.0 if p is m.weight else 0.0)).abs().sum().item()
    return s, a


def _worker(rank, world, port, head_norm, q, seed_per_rank=False, backbone_norm=None, steps=3,
            backend="gloo"):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":       # RCCL: one GPU per rank
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
    from gaia_seg_amd.core.synthetic import make_batch
    model = _build(head_norm, seed=rank if seed_per_rank else 0, backbone_norm=backbone_norm).cuda().train()
    runner, arena = _runner(model)
    names = []
    for it in range(steps):
        batch = make_batch(2, 64, 96, seed=100 * it + rank, device="cuda", border=2)
        out = runner.train_iter(batch)
        names.append(runner.arch_name)
    torch.cuda.synchronize()
    q.put((rank, arena.flat_param.double().sum().item(), arena.flat_param.abs().double().sum().item(),
           names, float(out["log_vars"]["loss"]), runner.reducer.bytes_reduced, _norm_checksum(model)))
    dist.destroy_process_group()


def _spawn(head_norm, **kw):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, head_norm, q), kwargs=kw) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return out


def test_two_ranks_stay_in_lockstep_with_syncbn_heads():
    r0, r1 = _spawn("SyncBN")
    assert r0[3] == r1[3]                                 # same subnet sequence on both ranks
    assert r0[1] == r1[1] and r0[2] == r1[2]              # bit-identical parameters after 3 steps
    assert r0[4] == r1[4]                                 # log vars are rank-averaged
    assert r0[5] == r1[5] > 0


def test_two_rank_step_equals_gradient_average():
    """With rank-local BN everywhere, 3 data-parallel steps on 2 ranks equal a single process that
    averages the two per-rank gradients (accumulate, grad_scale 1/2)."""
    r0, _ = _spawn("BN")
    from gaia_seg_amd.core.dynamic import fold_dict
    from gaia_seg_amd.core.synthetic import make_batch
    model = _build("BN").cuda().train()
    runner, arena = _runner(model)
    hook = runner.hooks[0]
    for it in range(3):
        hook.before_train_iter(runner)                    # same seeded draw as rank 0
        arena.zero_grad(runner.active_ranges)
        for rank in range(2):
            batch = make_batch(2, 64, 96, seed=100 * it + rank, device="cuda", border=2)
            out = model.train_step(batch, None)
            out["loss"].backward()
            # gradients are WRITTEN (not accumulated) by the kernels: keep a running sum
            if rank == 0:
                g0 = arena.flat_grad.clone()
            else:
                arena.flat_grad.add_(g0)
        arena.sgd_step(runner.active_ranges, 0.05, 0.9, 5e-4, 0.5)
        runner.iter += 1
    torch.cuda.synchronize()
    ref_sum = arena.flat_param.double().sum().item()
    ref_abs = arena.flat_param.abs().double().sum().item()
    assert hook.history == r0[3]
    assert abs(ref_sum - r0[1]) <= 1e-6 * ref_abs and abs(ref_abs - r0[2]) <= 1e-6 * ref_abs


def test_ranks_built_from_different_seeds_start_from_rank0_weights():
    """The reference launch passes no --seed: MMDistributedDataParallel's wrap-time broadcast makes
    the replicas equal (gaiaseg/apis/train.py:88-96).  Here: sync_module_states."""
    r0, r1 = _spawn("SyncBN", seed_per_rank=True)
    assert r0[1] == r1[1] and r0[2] == r1[2]
    ref, _ = _spawn("SyncBN")                             # both ranks seeded like rank 0
    assert r0[1] == ref[1] and r0[2] == ref[2]


def test_two_rank_syncbn_equals_one_process_on_the_concatenated_batch():
    """With every BatchNorm synchronised over the world, 2 ranks x bs 2 is the same function as one
    process on the 4 images: global batch statistics, mean of the two per-rank losses = loss of the
    concatenation (equal sizes), gradients averaged.  Pins the VALUES of the SyncBN gamma / beta
    gradients (torch.nn.SyncBatchNorm + DDP: (1/world) * sum of the per-rank sums)."""
    r0, _ = _spawn("SyncBN", backbone_norm="SyncBN", steps=2)
    from gaia_seg_amd.core.synthetic import make_batch
    model = _build("SyncBN", backbone_norm="SyncBN").cuda().train()
    runner, arena = _runner(model)
    for it in range(2):
        parts = [make_batch(2, 64, 96, seed=100 * it + rank, device="cuda", border=2) for rank in range(2)]
        batch = dict(img=torch.cat([p["img"] for p in parts]),
                     gt_semantic_seg=torch.cat([p["gt_semantic_seg"] for p in parts]),
                     img_metas=parts[0]["img_metas"] + parts[1]["img_metas"])
        runner.train_iter(batch)
    torch.cuda.synchronize()
    ref_sum = arena.flat_param.double().sum().item()
    ref_abs = arena.flat_param.abs().double().sum().item()
    assert runner.hooks[0].history == r0[3]
    assert abs(ref_sum - r0[1]) <= 2e-6 * ref_abs and abs(ref_abs - r0[2]) <= 2e-6 * ref_abs
    ns, na = _norm_checksum(model)      # na = total distance the BN affine parameters moved
    assert na > 1e-3
    assert abs(ns - r0[6][0]) <= 1e-3 * na and abs(na - r0[6][1]) <= 1e-3 * na


# (no multi-GPU node has been available to this build so far: this is the only test of the suite that
# has never executed.  A plain skip on one GPU, a plain pass / fail on a node -- r03 advisor finding.)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank")
def test_two_ranks_over_rccl_match_the_gloo_run():
    """The same two-rank training over RCCL (xGMI) as over gloo: identical subnet sequence, ranks in
    lockstep, and the parameters equal the gloo run's up to the all-reduce's summation order."""
    n0, n1 = _spawn("SyncBN", backend="nccl")
    assert n0[3] == n1[3] and n0[1] == n1[1] and n0[2] == n1[2] and n0[5] == n1[5] > 0
    g0, _ = _spawn("SyncBN")
    assert n0[3] == g0[3]
    assert abs(n0[1] - g0[1]) <= 1e-6 * g0[2] and abs(n0[2] - g0[2]) <= 1e-6 * g0[2]


def _one_rank_rccl(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        from gaia_seg_amd.core.dist import GradReducer
        flat = torch.arange(4096, dtype=torch.float32, device="cuda")
        want = flat.clone()
        red = GradReducer(flat, {}, bucket_bytes=1 << 20)
        assert red._coalescing()
        runs = [(0, 512), (1024, 1536), (3072, 4096)]
        red._issue(runs)                       # three runs -> ONE grouped RCCL launch
        assert red.collectives == 1 and len(red._works) == 1
        red._issue([(2048, 2560)])             # a single run: the plain call
        assert red.collectives == 2
        for w in red._works:
            w.wait()
        torch.cuda.synchronize()
        assert torch.equal(flat, want)         # the sum over one rank
        assert red.bytes_reduced == 4 * (512 + 512 + 1024 + 512)
        dist.destroy_process_group()
        q.put("ok")
    except BaseException as e:   # noqa: BLE001
        q.put("%s: %s" % (type(e).__name__, e))
        raise


def test_bucket_runs_go_out_as_one_grouped_rccl_launch():
    """The RCCL branch of GradReducer._issue (torch's coalescing manager -> ProcessGroupNCCL
    allreduce_coalesced) on a one-rank RCCL group: RCCL has never run anywhere else in this
    pipeline (no multi-GPU node so far), so at least the API path is executed on real hardware."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_rccl, args=(_free_port(), q))
    p.start()
    p.join(180)
    assert not p.is_alive()
    assert q.get(timeout=5) == "ok"
