#!/usr/bin/env python
"""Fused resize + cross-entropy kernels alone on the GPU, at the two head shapes of the FCN supernet
(decode head on the stage-4 feature, 16x32 -> 512x1024; auxiliary head on stage 3, 32x64 -> 512x1024;
N = 2, 19 classes, 5 % ignored pixels): us per forward / backward call (HIP events, 30 calls each).

    python tools/bench_ce.py [--md out.md]"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaia_seg_amd.hip import lib  # noqa: E402
from gaia_seg_amd.models.losses.cross_entropy_loss import _ce_desc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--md", default=None)
    ap.add_argument("--iters", type=int, default=30)
    args = ap.parse_args()
    L = lib.load()
    dev = torch.device("cuda")
    st = torch.cuda.current_stream().cuda_stream
    rows = []
    for n, H, W, h, w in [(2, 512, 1024, 16, 32), (2, 512, 1024, 32, 64), (2, 512, 1024, 64, 128),
                          (4, 769, 769, 193, 193)]:
        torch.manual_seed(0)
        logits = torch.randn(n, 19, h, w, device=dev).contiguous(memory_format=torch.channels_last)
        label = torch.randint(0, 19, (n, H, W), device=dev)
        label[torch.rand(n, H, W, device=dev) < 0.05] = 255
        d = _ce_desc(logits, (H, W), 255, False)
        lse = torch.empty((n, H, W), device=dev)
        out = torch.empty(2, device=dev)
        ws = torch.empty(max(L.gs_ce_workspace_bytes(ctypes.byref(d)), 256), dtype=torch.uint8, device=dev)
        ld = 20
        buf = torch.empty((n, h, w, ld), device=dev)
        nb = L.gs_ce_backward_workspace_bytes(ctypes.byref(d), ld)
        wsb = torch.empty(max(nb, 256), dtype=torch.uint8, device=dev)
        fwd = lambda: L.gs_ce_forward_scaled(ctypes.byref(d), logits.data_ptr(), label.data_ptr(), None, None,
                                             lse.data_ptr(), 1.0 / label.numel(), 100.0 / label.numel(),
                                             out.data_ptr(), ws.data_ptr(), ws.numel(), st)
        bwd = lambda: L.gs_ce_backward_ws(ctypes.byref(d), logits.data_ptr(), label.data_ptr(), None, None,
                                          lse.data_ptr(), 1.0 / label.numel(), buf.data_ptr(), ld,
                                          wsb.data_ptr() if nb else None, nb, st)
        for name, fn in (("forward", fwd), ("backward", bwd)):
            for _ in range(3):
                lib.check(fn(), name)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / args.iters
            mpx = n * H * W / 1e6
            rows.append(("%dx%dx%d -> %dx%d" % (n, h, w, H, W), name, "tile" if (name == "backward" and nb) else "-",
                         us, mpx * 19 / us * 1e3))
            print("%-28s %-8s %-5s %8.1f us  %7.1f G softmax terms/s" % rows[-1], flush=True)
    if args.md:
        with open(args.md, "w") as f:
            f.write("| logits -> labels | pass | form | us | G softmax terms / s |\n|---|---|---|---|---|\n")
            for r in rows:
                f.write("| %s | %s | %s | %.1f | %.1f |\n" % r)


if __name__ == "__main__":
    main()
