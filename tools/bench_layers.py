"""Time forward and backward (data + weight gradient) of the non-trunk CycleGAN conv layers at 512x512, batch N (HIP events).
Usage: python tools/bench_layers.py [--n 8]"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "automatic-sem-image-segmentation_amd"
E = importlib.import_module(PKG + ".engine")
LY = importlib.import_module(PKG + ".layers")

# name, k, cin, cout, stride, padding, transposed, input h=w
LAYERS = [
    ("g_stem7", 7, 1, 64, 1, ("reflect", 3), False, 512),
    ("g_down1", 3, 64, 128, 2, "same", False, 512),
    ("g_down2", 3, 128, 256, 2, "same", False, 256),
    ("g_down3", 3, 256, 512, 2, "same", False, 128),
    ("g_trunk", 3, 512, 512, 1, ("reflect", 1), False, 64),
    ("g_up1", 3, 512, 256, 2, "same", True, 64),
    ("g_up2", 3, 256, 128, 2, "same", True, 128),
    ("g_up3", 3, 128, 64, 2, "same", True, 256),
    ("g_head7", 7, 64, 1, 1, ("reflect", 3), False, 512),
    ("d_c1", 4, 1, 128, 2, "valid", False, 512),
    ("d_c2", 4, 128, 256, 2, "valid", False, 255),
    ("d_c3", 4, 256, 512, 2, "valid", False, 126),
    ("d_out", 4, 512, 1, 1, "valid", False, 62),
]


def timeit(fn, iters):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    print(f"{'layer':10s} {'GFLOP':>8s} {'fwd ms':>8s} {'TF/s':>7s} {'bwd ms':>8s} {'TF/s':>7s}")
    for name, k, cin, cout, s, pad, tr, hw in LAYERS:
        arena = E.ParamArena(dev)
        conv = LY.Conv2D(arena, "c", k, cin, cout, stride=s, padding=pad, transposed=tr)
        arena.materialize()
        arena["c/kernel"].uniform_(-0.05, 0.05)
        x = E.Act(torch.randn((a.n, hw, hw, cin), device=dev), requires_grad=True)
        tape = E.Tape()
        y = conv(tape, x)
        macs = y.n * y.h * y.w * cout * k * k * cin if not tr else x.n * x.h * x.w * cin * k * k * cout
        gf = 2.0 * macs / 1e9
        f_ms = timeit(lambda: conv(E.Tape(enabled=False), x), a.iters)

        def bwd():
            t = E.Tape()
            yy = conv(t, x)
            gt, _ = yy.grad_target()
            x.grad_init = False
            t.backward()
        fb_ms = timeit(bwd, a.iters)
        b_ms = max(fb_ms - f_ms, 1e-6)
        print(f"{name:10s} {gf:8.1f} {f_ms:8.3f} {gf / f_ms:7.1f} {b_ms:8.3f} {2 * gf / b_ms:7.1f}")


if __name__ == "__main__":
    main()
