"""Micro-benchmarks of single C-ABI ops at the 512x512 / batch-8 shapes of the CycleGAN step (HIP-event timing).
Usage: python tools/bench_kernels.py [trunk_fwd|trunk_dgrad|trunk_wgrad|norm|all] [--iters N]"""
import argparse
import ctypes
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "automatic-sem-image-segmentation_amd"
E = importlib.import_module(PKG + ".engine")
LY = importlib.import_module(PKG + ".layers")
L = importlib.import_module(PKG + "._lib")


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
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--n", type=int, default=8)
    ap.add_argument("--hw", type=int, default=64)
    ap.add_argument("--c", type=int, default=512)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = L.load()
    n, hw, c = a.n, a.hw, a.c
    arena = E.ParamArena(dev)
    conv = LY.Conv2D(arena, "c", 3, c, c, padding=("reflect", 1))
    norm = LY.Norm(arena, "n", c, "instance")
    arena.materialize()
    arena["c/kernel"].uniform_(-0.05, 0.05)
    arena["n/gamma"].fill_(1.0)
    x = E.Act(torch.randn((n, hw, hw, c), device=dev))
    gflop = 2.0 * n * hw * hw * c * c * 9 / 1e9
    res = {}
    if a.what in ("trunk_fwd", "all"):
        tape = E.Tape(enabled=False)
        ms = timeit(lambda: conv(tape, x), a.iters)
        res["trunk_fwd"] = (ms, gflop / ms)
    if a.what in ("trunk_dgrad", "trunk_wgrad", "all"):
        tape = E.Tape()
        y = conv(tape, x)
        d = conv.desc(x, y)
        dy = E.Act(torch.randn_like(y.t))
        dx = E.Act(torch.empty_like(x.t))
        w = arena["c/kernel"]
        gw = arena.grad("c/kernel")
        if a.what in ("trunk_dgrad", "all"):
            ws = E.workspace(lib.ss_conv2d_workspace_bytes(ctypes.byref(d), L.PASS_BWD_DATA), dev)
            ms = timeit(lambda: L.check(lib.ss_conv2d_bwd_data(ctypes.byref(d), dy.ptr, E._p(w), dx.ptr, 0, E._p(ws), ws.numel(), E._stream()), "dgrad"), a.iters)
            res["trunk_dgrad"] = (ms, gflop / ms)
        if a.what in ("trunk_wgrad", "all"):
            ws = E.workspace(lib.ss_conv2d_workspace_bytes(ctypes.byref(d), L.PASS_BWD_WEIGHT), dev)
            ms = timeit(lambda: L.check(lib.ss_conv2d_bwd_weight(ctypes.byref(d), x.ptr, dy.ptr, E._p(gw), None, 1, E._p(ws), ws.numel(), E._stream()), "wgrad"), a.iters)
            res["trunk_wgrad"] = (ms, gflop / ms)
    if a.what in ("norm", "all"):
        tape = E.Tape()
        y = norm(tape, x, act="relu")
        ms = timeit(lambda: norm(E.Tape(enabled=False), x, act="relu"), a.iters)
        nbytes = x.t.numel() * 4
        res["instnorm_fwd"] = (ms, 3 * nbytes / ms / 1e6)   # GB/s at the 3-pass algorithmic traffic
        gt, _ = y.grad_target()
        gt.t.normal_()

        def bwd():
            y.grad_init = True
            x.grad = None
            x.grad_init = False
            tape.ops[-1]()
        ms = timeit(bwd, a.iters)
        res["instnorm_bwd"] = (ms, 5 * nbytes / ms / 1e6)
    for k, (ms, rate) in res.items():
        unit = "GB/s" if k.startswith("instnorm") else "TFLOP/s"
        print(f"{k:14s} {ms:8.3f} ms   {rate:8.1f} {unit}")


if __name__ == "__main__":
    main()
