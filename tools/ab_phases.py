"""Stride-2 data gradients / transposed convolutions: the fused four-phase kernel (conv_phase.hip) against one gather launch per phase --
agreement (rel-L2) and time of the pass, same box.  Usage: python tools/ab_phases.py [--n 16] [--small]"""
import argparse
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "automatic-sem-image-segmentation_amd"
E = importlib.import_module(PKG + ".engine")
LY = importlib.import_module(PKG + ".layers")
L = importlib.import_module(PKG + "._lib")

# name, k, cin, cout, stride, padding, transposed, h = w, which pass is the phase-type one
LAYERS_ALL = [("g_up2 T3x3 s2 128->64 (forward)", 3, 128, 64, 2, "same", True, 256, "fwd"),
          ("g_up1 T3x3 s2 256->128 (forward)", 3, 256, 128, 2, "same", True, 128, "fwd"),
          ("g_up0 T3x3 s2 512->256 (forward)", 3, 512, 256, 2, "same", True, 64, "fwd"),
          ("g_down0 3x3 s2 64->128 (data gradient)", 3, 64, 128, 2, "same", False, 512, "dgrad"),
          ("g_down1 3x3 s2 128->256 (data gradient)", 3, 128, 256, 2, "same", False, 256, "dgrad"),
          ("g_down2 3x3 s2 256->512 (data gradient)", 3, 256, 512, 2, "same", False, 128, "dgrad"),
          ("d_down0 4x4 s2 128->256 (data gradient)", 4, 128, 256, 2, "valid", False, 255, "dgrad"),
          ("d_down1 4x4 s2 256->512 (data gradient)", 4, 256, 512, 2, "valid", False, 126, "dgrad")]


LAYERS = [l for l in LAYERS_ALL if not os.environ.get("AB_ONLY") or os.environ["AB_ONLY"] in l[0]]


def rel_l2(a, b):
    return float((a - b).double().norm() / b.double().norm().clamp_min(1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=16)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--small", action="store_true", help="n = 2, quarter-size maps (correctness only)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for name, k, cin, cout, stride, padding, transposed, hw, which in LAYERS:
        n = 2 if a.small else a.n
        hw = max(hw // 4, 24) if a.small else hw
        g = torch.Generator().manual_seed(7)
        wt = torch.empty((k, k, cout, cin) if transposed else (k, k, cin, cout)).uniform_(-0.05, 0.05, generator=g)
        xt = torch.randn((n, hw, hw, cin), generator=g)
        res, out = [], []
        for fused in (0, 1):
            with L.config(phases_fused=(int(os.environ.get('AB_FLOOR', '1')) if fused else 0)):
                arena = E.ParamArena(dev)
                conv = LY.Conv2D(arena, "c", k, cin, cout, stride=stride, padding=padding, use_bias=False, transposed=transposed)
                arena.materialize()
                arena["c/kernel"].copy_(wt)
                x = E.Act(xt.to(dev), requires_grad=(which == "dgrad"))
                ts = []
                for it in range(a.iters + 2):
                    tape = E.Tape()
                    tape.param_grads = False
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    if which == "fwd":
                        e0.record()
                        y = conv(tape, x)
                        e1.record()
                        val = y
                    else:
                        y = conv(tape, x)
                        gt, _ = y.grad_target()
                        gy = torch.Generator(device=dev).manual_seed(5)
                        gt.t.normal_(generator=gy)
                        x.grad, x.grad_init = None, False
                        torch.cuda.synchronize()
                        e0.record()
                        tape.backward()
                        e1.record()
                        val = x.get_grad()
                    torch.cuda.synchronize()
                    if it >= 2:
                        ts.append(e0.elapsed_time(e1) * 1e3)
                res.append(sorted(ts)[len(ts) // 2])
                out.append(val.dense().clone())
                if os.environ.get("AB_PROF"):
                    lib = L.load()
                    lib.ss_prof_reset(); lib.ss_prof_enable(1)
                    tape = E.Tape(); tape.param_grads = False
                    if which == "fwd":
                        conv(tape, x)
                    else:
                        y = conv(tape, x); gt, _ = y.grad_target(); gt.t.normal_(); x.grad, x.grad_init = None, False
                        tape.backward()
                    torch.cuda.synchronize(); lib.ss_prof_enable(0)
                    for kname, v in L.prof_summary().items():
                        print(f"      [{'fused' if fused else 'per phase'}] {kname}: {v['launches']} x {v['total_ms'] / v['launches'] * 1e3:.1f} us")
        print(f"{name:42s} n={n} {hw}x{hw}: per phase {res[0]:8.1f} us   fused {res[1]:8.1f} us   rel-L2 {rel_l2(out[1], out[0]):.2e}   max|ref| {float(out[0].abs().max()):.3g}")


if __name__ == "__main__":
    main()
