"""Host-only dry run of a training step (no GPU): every C-ABI entry point is replaced by a no-op stub,
so only the Python orchestration runs, on CPU tensors.  Used as a subprocess by
tests/test_host_logic.py (the stubs patch the library module globally) and by hand to profile the
host cost of a step:  python tests/host_dry_run.py [steps] [--profile N]

Prints one JSON line: {"ms_per_step": ..., "unreachable_per_step": ...} — the second number is what
gc.collect() finds after one step: anything above zero is a reference cycle that keeps device memory
of a finished step alive until the cyclic collector runs."""
import gc
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gaia_seg_amd.hip import lib, ops, runtime  # noqa: E402


class _Stub:
    def __getattr__(self, name):
        if name.endswith("_bytes"):
            f = lambda *a: 1 << 20      # noqa: E731
        elif name.endswith("_supported"):
            f = lambda *a: 1            # noqa: E731
        else:
            f = lambda *a: 0            # noqa: E731
        setattr(self, name, f)
        return f


def main():
    stub = _Stub()
    lib._lib = stub
    lib.load = lambda: stub
    ops._L = lambda: stub
    runtime.require_gpu_tensor = lambda t, what: None
    runtime.current_stream_ptr = lambda: 0
    ops.current_stream_ptr = lambda: 0
    ops.SIDE_WGRAD = False
    from gaia_seg_amd.core.config import Config
    from gaia_seg_amd.core.dynamic import fold_dict
    from gaia_seg_amd.core.synthetic import make_batch
    from gaia_seg_amd.models import build_segmentor
    import gaia_seg_amd.models.losses.cross_entropy_loss as cel
    cel.require_gpu_tensor = lambda t, what: None
    cel.current_stream_ptr = lambda: 0
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    name = next((a for a in sys.argv[1:] if a.endswith(".py")), "configs/supernet/fcn_ar50to101v2.py")
    cfg = Config.fromfile(os.path.join(root, name))
    torch.manual_seed(0)
    model = build_segmentor(cfg.model, train_cfg=cfg.get("train_cfg"),
                            test_cfg=cfg.get("test_cfg")).train()
    r50 = {"arch.backbone.stem.width": 64, "arch.backbone.body.width": [64, 128, 256, 512],
           "arch.backbone.body.depth": [3, 4, 6, 3]}
    model.manipulate_arch(fold_dict(r50)["arch"])
    batch = make_batch(2, 64, 128, 19, 0, torch.device("cpu"))

    def step():
        out = model.train_step(batch, None)
        out["loss"].backward()

    for _ in range(3):
        step()
    gc.collect()
    step()
    unreachable = gc.collect()
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        best = min(best, (time.perf_counter() - t0) / n)
    if "--profile" in sys.argv:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(n):
            step()
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(int(sys.argv[sys.argv.index("--profile") + 1]))
    print(json.dumps({"ms_per_step": round(best * 1e3, 2), "unreachable_per_step": unreachable}))


if __name__ == "__main__":
    main()
