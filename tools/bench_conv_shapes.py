#!/usr/bin/env python
"""Per-shape timing of the three conv entry points on the supernet's GEMM shapes at 1024x512, bs 2,
kernels ALONE on the GPU (HIP events on the launch stream, 30 launches each):

    python tools/bench_conv_shapes.py [--ops fwd,dgrad,wgrad] [--ks 1,3] [--md out.md]

Per shape: us per call (conv + its split-K reduce), fp32-equivalent TFLOP/s = 2*M*N*K / time and the
fraction of the fp32 MFMA peak (157.3), algorithmic MB (operands once) and the GB/s that implies, the
K loop that ran (gs_debug_last_conv_launch).  The 1x1 shapes include the first block of a stage
(conv1 at the previous resolution, the strided shortcut); widths are the R50 / MAX / MIN anchors'."""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaia_seg_amd.hip import lib  # noqa: E402

KLOOP = {0: "generic", 1: "fp32", 2: "fp32x2", 3: "bf16x3", 4: "stream"}
RES = [(128, 256), (64, 128), (32, 64), (16, 32)]


def shapes(ks_list, widths):
    out = []   # (name, N, H, W, Ci, Co, k, stride)
    for wname, ws in widths:
        for s, (h, w) in enumerate(RES):
            wd = ws[s]
            if 3 in ks_list:
                out.append(("%s s%d conv2 3x3 %d" % (wname, s + 1, wd), 2, h, w, wd, wd, 3, 1))
            if 1 in ks_list:
                out.append(("%s s%d conv3 1x1 %d->%d" % (wname, s + 1, wd, 4 * wd), 2, h, w, wd, 4 * wd, 1, 1))
                out.append(("%s s%d conv1 1x1 %d->%d" % (wname, s + 1, 4 * wd, wd), 2, h, w, 4 * wd, wd, 1, 1))
                if s > 0:   # first block of the stage: conv1 reads the previous stage's output
                    ph, pw = RES[s - 1]
                    out.append(("%s s%d.0 conv1 1x1 %d->%d" % (wname, s + 1, 4 * ws[s - 1], wd), 2, ph, pw,
                                4 * ws[s - 1], wd, 1, 1))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ops", default="fwd,dgrad,wgrad")
    ap.add_argument("--ks", default="1,3")
    ap.add_argument("--anchors", default="R50")
    ap.add_argument("--md", default=None)
    ap.add_argument("--iters", type=int, default=30)
    args = ap.parse_args()
    L = lib.load()
    dev = torch.device("cuda")
    st = torch.cuda.current_stream().cuda_stream
    allw = {"R50": [64, 128, 256, 512], "MAX": [80, 160, 320, 640], "MIN": [48, 96, 192, 384]}
    widths = [(a, allw[a]) for a in args.anchors.split(",")]
    ks_list = [int(v) for v in args.ks.split(",")]
    rows = []
    for name, n, h, w, ci, co, k, stride in shapes(ks_list, widths):
        p = k // 2
        ho, wo = (h + 2 * p - k) // stride + 1, (w + 2 * p - k) // stride + 1
        x = torch.randn(n, h, w, ci, device=dev)
        y = torch.randn(n, ho, wo, co, device=dev)
        wt = torch.randn(k, k, ci, co, device=dev) * 0.05
        dx = torch.zeros_like(x)
        dw = torch.zeros_like(wt)
        d = lib.ConvDesc(N=n, H=h, W=w, Ci=ci, Co=co, Ci_max=ci, Co_ld=co, KH=k, KW=k, stride=stride,
                         pad=p, dil=1, Ho=ho, Wo=wo, x_sn=h * w * ci, x_sh=w * ci, x_sw=ci, x_sc=1,
                         ldy=co, ld_add=0, role=0, reserved=0, in_affine=None)
        need = L.gs_conv2d_workspace_bytes(ctypes.byref(d))
        ws = torch.empty(max(need, 256), dtype=torch.uint8, device=dev)
        calls = {
            "fwd": lambda: L.gs_conv2d_forward(ctypes.byref(d), x.data_ptr(), wt.data_ptr(), None, None,
                                               y.data_ptr(), ws.data_ptr(), ws.numel(), st),
            "dgrad": lambda: L.gs_conv2d_dgrad(ctypes.byref(d), y.data_ptr(), wt.data_ptr(), dx.data_ptr(),
                                               0, ws.data_ptr(), ws.numel(), st),
            "wgrad": lambda: L.gs_conv2d_wgrad(ctypes.byref(d), x.data_ptr(), y.data_ptr(), dw.data_ptr(),
                                               ws.data_ptr(), ws.numel(), st),
        }
        fl = 2.0 * n * ho * wo * ci * co * k * k
        by = 4.0 * (x.numel() + y.numel() + wt.numel())
        for op in args.ops.split(","):
            run = calls[op]
            for _ in range(5):
                lib.check(run(), op)
            rec = lib.DebugLaunch()
            L.gs_debug_last_conv_launch(ctypes.byref(rec))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / args.iters
            rows.append((name, op, us, fl / us / 1e6, fl / us / 1e6 / 157.3, by / 1e6, by / us / 1e3,
                         "%s 64x%d s%d" % (KLOOP.get(rec.kloop, "?"), rec.bn, rec.splits)))
            print("%-34s %-6s %7.1f us %6.1f TF %5.2f  %6.1f MB %6.0f GB/s  %s" % rows[-1], flush=True)
    if args.md:
        with open(args.md, "w") as f:
            f.write("| shape | op | us | TFLOP/s | of fp32 peak | algorithmic MB | GB/s | kernel |\n"
                    "|---|---|---|---|---|---|---|---|\n")
            for r in rows:
                f.write("| %s | %s | %.1f | %.1f | %.2f | %.1f | %.0f | %s |\n" % r)


if __name__ == "__main__":
    main()
