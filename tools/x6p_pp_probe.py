"""A/B of the ping-pong x3h Winograd GEMM (gemm_x6p_pp_kernel, config key x6p_pp) against the one-phase kernel on the SAME
problems: results must be bit-identical (same MFMA order per accumulator), times come from the library's HIP-event recorder.
Usage: python tools/x6p_pp_probe.py [--iters N] [--quick]"""
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


def run(lib, dev, n, hw, cin, cout, iters, reflect=True):
    arena = E.ParamArena(dev)
    conv = LY.Conv2D(arena, "c", 3, cin, cout, padding=("reflect", 1) if reflect else "same")
    arena.materialize()
    g = torch.Generator(device="cpu").manual_seed(n * 1000 + hw + cin)
    arena["c/kernel"].copy_(torch.empty(arena["c/kernel"].shape).uniform_(-0.05, 0.05, generator=g))
    x = E.Act(torch.randn((n, hw, hw, cin), generator=g).to(dev))
    dyt = torch.randn((n, hw, hw, cout), generator=g).to(dev)
    res = {}
    for pp in (0, 1, 2, 3):
        with L.config(x6p_pp=pp, x6p=2):
            tape = E.Tape()
            y = conv(tape, x)
            d = conv.desc(x, y)
            dy = E.Act(dyt.clone())
            dx = E.Act(torch.empty_like(x.t))
            w = arena["c/kernel"]
            ws = E.workspace(lib.ss_conv2d_workspace_bytes(ctypes.byref(d), L.PASS_BWD_DATA), dev)

            def bwd():
                L.check(lib.ss_conv2d_bwd_data(ctypes.byref(d), dy.ptr, E._p(w), dx.ptr, 0, E._p(ws), ws.numel(), E._stream()), "dgrad")

            bwd()
            torch.cuda.synchronize()
            res[pp] = [y.t.clone(), dx.t.clone()]
            lib.ss_prof_reset()
            lib.ss_prof_enable(1)
            for _ in range(iters):
                conv(E.Tape(enabled=False), x)
                bwd()
            torch.cuda.synchronize()
            lib.ss_prof_enable(0)
            prof = L.prof_summary()
            res[pp].append({k: (round(v["avg_ms"] * 1e3, 1), v["launches"]) for k, v in prof.items() if k.startswith("gemm_x6p")})
    same_y = all(torch.equal(res[0][0], res[k][0]) for k in (1, 2, 3))
    same_dx = all(torch.equal(res[0][1], res[k][1]) for k in (1, 2, 3))
    finite = bool(torch.isfinite(res[1][0]).all() and torch.isfinite(res[1][1]).all())
    print(f"n={n} hw={hw} {cin}->{cout}: y_equal={same_y} dx_equal={same_dx} finite={finite}  one-phase {res[0][2]}  pp(M) {res[1][2]}  pp(3+3) {res[2][2]}  pp(C) {res[3][2]}", flush=True)
    if not (same_y and same_dx):
        dlt = (res[0][0] - res[1][0]).abs().max().item(), (res[0][1] - res[1][1]).abs().max().item()
        print("   max |diff| y, dx:", dlt, " max|y|", res[0][0].abs().max().item(), flush=True)
    return same_y and same_dx and finite


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = L.load()
    ok = True
    shapes = [(16, 64, 512, 512), (8, 64, 512, 512)]
    if not a.quick:
        # ragged M (tiles not a multiple of 256), N = 64 / 192 (edge slabs), K = 64 (two chunks), non-reflect padding, one sample
        shapes += [(1, 64, 512, 512), (2, 40, 256, 256), (3, 36, 64, 64), (1, 20, 128, 192), (5, 28, 64, 320), (2, 64, 256, 128)]
    for (n, hw, ci, co) in shapes:
        ok &= run(lib, dev, n, hw, ci, co, a.iters if ci >= 512 else 3, reflect=(hw % 8 == 0))
    for rep in range(3 if not a.quick else 1):          # repeated runs of the headline shape: a race would show as run-to-run differences
        ok &= run(lib, dev, 8, 64, 512, 512, 3)
    print("ALL_OK" if ok else "MISMATCH", flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
