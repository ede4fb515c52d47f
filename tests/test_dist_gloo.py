"""Multi-process (world_size 2, gloo, CPU) tests of the data-parallel host logic: the bucketed
gradient reducer over a flat arena, the arch broadcast of ManipulateArchHook, SyncBN statistics
merging.  RCCL itself is only exercised on the GPU box; the code path is the same."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _worker_reducer(rank, world, port, q):
    _init(rank, world, port)
    from gaia_seg_amd.core.dist import GradReducer
    sizes = (64, 128, 64, 256, 64, 512)
    params = [torch.nn.Parameter(torch.zeros(n)) for n in sizes]
    segs, off = {}, 0
    for p in params:
        segs[id(p)] = (off, p.numel())
        off += p.numel()
    flat = torch.zeros(off)
    g = torch.Generator().manual_seed(100 + rank)
    local = torch.randn(off, generator=g)
    flat.copy_(local)
    red = GradReducer(flat, segs, bucket_bytes=4 * 300)
    active = [params[i] for i in (0, 1, 3, 5)]          # params 2 and 4 are depth-skipped blocks
    red.begin(active, key=("arch", 1))
    for p in reversed(active):                           # backward order
        p._gs_grad_ready(p)
    red.finish()
    # expected: sum over ranks on the active ranges, untouched elsewhere
    both = [torch.randn(off, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
    want = local.clone()
    for i in (0, 1, 3, 5):
        o, n = segs[id(params[i])]
        want[o:o + n] = sum(b[o:o + n] for b in both)
    ok = bool(torch.allclose(flat, want))
    # one large bucket with holes at the skipped blocks: three runs, three calls, same result
    flat.copy_(local)
    red2 = GradReducer(flat, segs, bucket_bytes=4 * 4000)
    plan = red2._plan(active, key=("arch", 2))
    grouped = len(plan) == 1 and len(plan[0]["runs"]) == 3
    red2.begin(active, key=("arch", 2))
    for p in reversed(active):
        p._gs_grad_ready(p)
    red2.finish()
    ok = ok and grouped and bool(torch.allclose(flat, want))
    # padded holes (hole_frac = 1: everything between the runs travels along) with the debug check on:
    # zero holes pass, a stale value in a hole is caught before it would be summed over the ranks
    for stale in (False, True):
        flat.copy_(local)
        for i in (2, 4):
            o, n = segs[id(params[i])]
            flat[o:o + n] = 0.0
        if stale:
            flat[segs[id(params[2])][0] + 3] = 1.0
        red3 = GradReducer(flat, segs, bucket_bytes=4 * 4000, hole_frac=1.0)
        red3.check_holes = True
        red3.begin(active, key=None)
        caught = False
        try:
            for p in reversed(active):
                p._gs_grad_ready(p)
            red3.finish()
        except AssertionError:
            caught = True
        ok = ok and (caught == stale)
        if caught:      # keep the ranks' collectives paired: the other rank raised at the same point
            red3._active, red3._works = None, []
    q.put((rank, ok, red.bytes_reduced))
    dist.destroy_process_group()


def _worker_hook(rank, world, port, q):
    _init(rank, world, port)
    from gaia_seg_amd.core.model_space import build_model_sampler
    from gaia_seg_amd.core.runner import ManipulateArchHook

    class FakeRunner:
        def __init__(self):
            self.metas = []

        def set_arch(self, meta):
            self.metas.append(meta)

    sampler = build_model_sampler(dict(type="anchor", anchors=[
        {"name": "A", "arch.backbone.body.depth": [1, 1]}, {"name": "B", "arch.backbone.body.depth": [2, 2]},
        {"name": "C", "arch.backbone.body.depth": [3, 3]}]))
    sampler.seed(1234 + rank)                            # different RNG streams on purpose
    hook, runner = ManipulateArchHook(sampler), FakeRunner()
    for _ in range(12):
        hook.before_train_iter(runner)
    q.put((rank, [m["name"] for m in runner.metas]))
    dist.destroy_process_group()


def _worker_syncbn(rank, world, port, q):
    _init(rank, world, port)
    from gaia_seg_amd.hip.ops import _sync_stats
    torch.manual_seed(0)
    full = torch.randn(6, 8) * 2 + 1                      # 6 samples x 8 channels, split 4 / 2
    mine = full[:4] if rank == 0 else full[4:]
    shift = mine[0].clone()
    d = mine - shift
    sums = torch.cat([d.sum(0), (d * d).sum(0), shift])
    merged, count = _sync_stats(sums, float(mine.shape[0]), 8, dist.group.WORLD)
    mean = merged[16:24] + merged[:8] / count
    var = merged[8:16] / count - (merged[:8] / count) ** 2
    ok = (count == 6.0 and torch.allclose(mean, full.mean(0), atol=1e-5)
          and torch.allclose(var, full.var(0, unbiased=False), atol=1e-5))
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def _run(worker):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(out)


def test_bucketed_reducer_sums_active_ranges_only():
    out = _run(_worker_reducer)
    assert all(ok for _, ok, _ in out), out
    assert out[0][2] == out[1][2] == 4 * (64 + 128 + 256 + 512)


def test_manipulate_arch_hook_broadcasts_rank0_draw():
    out = _run(_worker_hook)
    assert out[0][1] == out[1][1] and len(set(out[0][1])) > 1


def test_syncbn_statistics_merge_matches_global_batch():
    out = _run(_worker_syncbn)
    assert all(ok for _, ok in out), out


def _worker_collect(rank, world, port, q):
    _init(rank, world, port)
    from gaia_seg_amd.apis.test import collect_results
    size = 7                                  # 7 samples over 2 ranks: the sampler pads rank 1
    mine = ["sample%d" % i for i in range(rank, 8, world)]   # rank r holds r, r + world, ...
    got = collect_results(mine, size)
    q.put((rank, got))
    dist.destroy_process_group()


def test_collect_results_restores_dataset_order():
    """gaiaseg/apis/test.py:113-186: rank 0 gets every rank's results interleaved in dataset order
    and truncated to the dataset size; other ranks get None."""
    out = _run(_worker_collect)
    assert out[0][1] == ["sample%d" % i for i in range(7)]
    assert out[1][1] is None
