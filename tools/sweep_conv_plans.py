#!/usr/bin/env python
"""Sweep tile / split-K plans of the dynamic conv kernels over the supernet's GEMM shapes
(1024x512 crops, bs 2) and report, per shape and pass, the planner's choice against the best
measured plan.  Writes JSON; the planner rules in csrc/igemm_core.h (make_plan) are fitted to it.

    python tools/sweep_conv_plans.py [--out gpurun_out/plan_sweep.json] [--quick]
"""
import argparse
import ctypes
import json
import os.path as osp
import sys

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import torch  # noqa: E402

from gaia_seg_amd.core.bricks import DynamicConv2d  # noqa: E402
from gaia_seg_amd.hip import lib, ops  # noqa: E402
from gaia_seg_amd.hip.runtime import Act, Tape  # noqa: E402


def shapes(quick):
    out = []
    for stage, (h, w) in enumerate([(128, 256), (64, 128), (32, 64), (16, 32)]):
        for base in ((64,) if quick else (48, 64, 80)):
            c = base << stage
            out.append(("s%d 3x3 %d" % (stage + 1, c), c, c, 3, h, w))
            out.append(("s%d 1x1 %d->%d" % (stage + 1, 4 * c, c), 4 * c, c, 1, h, w))
            out.append(("s%d 1x1 %d->%d" % (stage + 1, c, 4 * c), c, 4 * c, 1, h, w))
    out.append(("fcn 3x3 2048->512", 2048, 512, 3, 16, 32))
    out.append(("aux 3x3 1024->256", 1024, 256, 3, 32, 64))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/plan_sweep.json")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--modes", default="fwd,dgrad,wgrad")
    args = ap.parse_args()
    dev = torch.device("cuda")
    L = lib.load()
    st = torch.cuda.current_stream().cuda_stream
    big_ws = torch.empty(96 << 20, dtype=torch.uint8, device=dev)  # kMaxSlabBytes
    results = []
    for name, ci, co, k, h, w in shapes(args.quick):
        m = DynamicConv2d(ci, co, k, padding=k // 2, bias=False).to(dev)
        x = Act(torch.randn(2, h, w, ci, device=dev), True)
        y = m.forward_act(Tape(enabled=False), x)
        gy = torch.randn_like(y.t)
        gx = torch.empty_like(x.t)
        gw = ops.ensure_grad(m.weight)
        d = ops._conv_desc(x, m.weight, co, 1, k // 2, 1, y.ld)
        M = 2 * h * w
        for mode in args.modes.split(","):
            if mode == "wgrad":
                gm, gn, gk = k * k * ci, co, M
            elif mode == "dgrad":
                gm, gn, gk = M, ci, k * k * co
            else:
                gm, gn, gk = M, co, k * k * ci
            flops = 2.0 * gm * gn * gk

            def call():
                if mode == "fwd":
                    return L.gs_conv2d_forward(ctypes.byref(d), x.ptr, m.weight.data_ptr(), None, None,
                                               y.ptr, big_ws.data_ptr(), big_ws.numel(), st)
                if mode == "dgrad":
                    return L.gs_conv2d_dgrad(ctypes.byref(d), gy.data_ptr(), m.weight.data_ptr(),
                                             gx.data_ptr(), 0, big_ws.data_ptr(), big_ws.numel(), st)
                return L.gs_conv2d_wgrad(ctypes.byref(d), x.ptr, gy.data_ptr(), gw.data_ptr(),
                                         big_ws.data_ptr(), big_ws.numel(), st)

            def timed(n=12):
                rc = call()
                if rc != 0:
                    return None
                for _ in range(2):
                    call()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    call()
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) * 1e3 / n

            L.gs_debug_force_plan(0, 0, 0)
            base = timed()
            cands = {}
            nk = (gk + 15) // 16
            bns = sorted({bn for bn in (32, 48, 64, 80, 96, 128)
                          if -(-gn // bn) * bn <= 1.35 * gn or bn == 32}, reverse=True)
            for bm in (64, 128):
                if bm == 128 and gm < 128:
                    continue
                for bn in bns:
                    tiles = -(-gm // bm) * -(-gn // bn)
                    for s in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 64, 128, 256):
                        if s > 1 and (s * tiles > 4096 or nk // s < 2 or s * gm * gn * 4 > (96 << 20)):
                            continue
                        if s == 1 and tiles > 16384:
                            continue
                        L.gs_debug_force_plan(bm, bn, s)
                        t = timed(8)
                        if t is not None:
                            cands["%dx%d/%d" % (bm, bn, s)] = round(t, 2)
            L.gs_debug_force_plan(0, 0, 0)
            best = min(cands, key=cands.get)
            rec = dict(shape=name, mode=mode, M=gm, N=gn, K=gk, planner_us=round(base, 2),
                       best=best, best_us=cands[best], cands=cands)
            results.append(rec)
            print("%-22s %-5s M=%-6d N=%-5d K=%-6d planner %6.1f us (%5.1f TF) | best %-11s %6.1f us (%5.1f TF) %+5.1f%%"
                  % (name, mode, gm, gn, gk, base, flops / base / 1e6, best, cands[best],
                     flops / cands[best] / 1e6, 100 * (base / cands[best] - 1)), flush=True)
    with open(args.out, "w") as f:
        json.dump(results, f, indent=1)
    tot_p = sum(r["planner_us"] for r in results)
    tot_b = sum(r["best_us"] for r in results)
    print("sum planner %.1f us, sum best %.1f us (%.1f%%)" % (tot_p, tot_b, 100 * (tot_p / tot_b - 1)))


if __name__ == "__main__":
    main()
