#!/usr/bin/env python
"""Times gs_conv2d_wgrad on the supernet's 3x3 / 1x1 weight-gradient shapes (1024x512, bs 2).

    python tools/bench_wgrad.py            # all-taps kernel where eligible
    GS_NO_WGRAD_TAPS=1 python tools/bench_wgrad.py   # tap-major kernel (r01)
Prints us per call and TFLOP/s (2*M*N*K) per shape; HIP events on the launch stream."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaia_seg_amd.hip import lib  # noqa: E402

SHAPES = []   # (name, N, H, W, Ci, Co, k)
for stage, (h, w) in enumerate([(128, 256), (64, 128), (32, 64), (16, 32)]):
    for width in ([48, 64, 80], [96, 128, 160], [192, 256, 320], [384, 512, 640])[stage]:
        SHAPES.append(("s%d 3x3 w%d" % (stage + 1, width), 2, h, w, width, width, 3))
for stage, (h, w) in enumerate([(128, 256), (64, 128), (32, 64), (16, 32)]):
    width = (64, 128, 256, 512)[stage]
    SHAPES.append(("s%d 1x1 %d->%d" % (stage + 1, 4 * width, width), 2, h, w, 4 * width, width, 1))
    SHAPES.append(("s%d 1x1 %d->%d" % (stage + 1, width, 4 * width), 2, h, w, width, 4 * width, 1))


def main():
    L = lib.load()
    dev = torch.device("cuda")
    st = torch.cuda.current_stream().cuda_stream
    for name, n, h, w, ci, co, k in SHAPES:
        x = torch.randn(n, h, w, ci, device=dev)
        dy = torch.randn(n, h, w, co, device=dev)
        dw = torch.zeros(k, k, ci, co, device=dev)
        d = lib.ConvDesc()
        d.N, d.H, d.W, d.Ci, d.Co = n, h, w, ci, co
        d.Ci_max, d.Co_ld, d.KH, d.KW = ci, co, k, k
        d.stride, d.pad, d.dil, d.Ho, d.Wo = 1, k // 2, 1, h, w
        d.x_sn, d.x_sh, d.x_sw, d.x_sc = h * w * ci, w * ci, ci, 1
        d.ldy = co
        need = L.gs_conv2d_workspace_bytes(ctypes.byref(d))
        ws = torch.empty(max(need, 256), dtype=torch.uint8, device=dev)

        def run():
            lib.check(L.gs_conv2d_wgrad(ctypes.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(),
                                        ws.data_ptr(), ws.numel(), st), "wgrad")
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 30
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        fl = 2.0 * n * h * w * ci * co * k * k
        print("%-22s %8.1f us  %6.1f TF  (slab %5.1f MB)" % (name, us, fl / us / 1e6, need / 2**20))


if __name__ == "__main__":
    main()
