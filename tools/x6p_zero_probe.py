"""Is the Winograd GEMM bound by stalls or by the chip's power budget?  The same launches (trunk convolution 512 -> 512 on 64 x 64
maps: 36 GEMMs of M = 256 n, N = K = 512) on random operands and on all-zero operands: the instruction stream, the addresses and the
bytes moved are identical, only the bit activity of the data differs.  Times from the library's HIP-event recorder.
Usage: python tools/x6p_zero_probe.py [--iters N]"""
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


def run(lib, dev, n, iters, xmode, wmode):
    arena = E.ParamArena(dev)
    conv = LY.Conv2D(arena, "c", 3, 512, 512, padding=("reflect", 1))
    arena.materialize()
    g = torch.Generator(device="cpu").manual_seed(7)
    w = arena["c/kernel"]
    if wmode == "random":
        w.copy_(torch.empty(w.shape).uniform_(-0.05, 0.05, generator=g))
    elif wmode == "const":
        w.fill_(0.03125)
    else:
        w.zero_()
    if xmode == "random":
        xt = torch.randn((n, 64, 64, 512), generator=g).to(dev)
    elif xmode == "const":
        xt = torch.full((n, 64, 64, 512), 0.5, device=dev)
    else:
        xt = torch.zeros((n, 64, 64, 512), device=dev)
    x = E.Act(xt)
    for _ in range(3):
        conv(E.Tape(enabled=False), x)
    torch.cuda.synchronize()
    lib.ss_prof_reset()
    lib.ss_prof_enable(1)
    for _ in range(iters):
        conv(E.Tape(enabled=False), x)
    torch.cuda.synchronize()
    lib.ss_prof_enable(0)
    prof = L.prof_summary()
    return {k: round(v["avg_ms"] * 1e3, 1) for k, v in prof.items() if k.startswith("gemm_x6p")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = L.load()
    for n in (8, 16):
        for xm, wm in (("random", "random"), ("zero", "random"), ("random", "zero"), ("zero", "zero"), ("const", "const"), ("random", "random")):
            print(f"n={n} x={xm:6s} w={wm:6s}: {run(lib, dev, n, a.iters, xm, wm)}", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
