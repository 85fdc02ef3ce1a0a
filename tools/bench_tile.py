"""Per-layer timing of the MultiResUNet's small-channel stride-1 convolutions (forward, data gradient, weight gradient) through the
C ABI, with the LDS-staged tile kernels (conv_tile.hip) on and off, against the HBM floor of each pass.
Usage: python tools/bench_tile.py [--n 8] [--iters 20] [--only fwd|dgrad|wgrad]"""
import argparse
import ctypes
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "automatic-sem-image-segmentation_amd"
E = importlib.import_module(PKG + ".engine")
L = importlib.import_module(PKG + "._lib")

# name, k, cin, cout, h=w  (UNet_Segmentation.py:451-503 at 512x512 tiles)
LAYERS = [
    ("mrb1.sc 1x1 1->25", 1, 1, 25, 512), ("mrb1.3 1->4", 3, 1, 4, 512), ("mrb1.5 4->8", 3, 4, 8, 512), ("mrb1.7 8->13", 3, 8, 13, 512),
    ("rp1.sc 1x1 25->16", 1, 25, 16, 512), ("rp1.3 25->16", 3, 25, 16, 512), ("rp1.sc 1x1 16->16", 1, 16, 16, 512),
    ("rp1.3 16->16", 3, 16, 16, 512), ("mrb9.sc 1x1 32->25", 1, 32, 25, 512), ("mrb9.3 32->4", 3, 32, 4, 512),
    ("out 1x1 25->1", 1, 25, 1, 512),
    ("mrb2.sc 1x1 25->51", 1, 25, 51, 256), ("mrb2.3 25->8", 3, 25, 8, 256), ("mrb2.5 8->17", 3, 8, 17, 256), ("mrb2.7 17->26", 3, 17, 26, 256),
    ("rp2.3 51->32", 3, 51, 32, 256), ("rp2.3 32->32", 3, 32, 32, 256), ("mrb8.sc 1x1 64->105", 1, 64, 105, 256),
    ("mrb8.3 64->17", 3, 64, 17, 256), ("mrb8.5 17->35", 3, 17, 35, 256), ("mrb8.7 35->53", 3, 35, 53, 256),
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
    return e0.elapsed_time(e1) / iters * 1e3     # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    lib = L.load()
    dev = torch.device("cuda:0")
    print(f"{'layer':22s} {'pass':6s} {'floor us':>9s} {'tile us':>9s} {'old us':>9s} {'tile TB/s':>10s}")
    tot = {}
    for name, k, cin, cout, hw in LAYERS:
        x = torch.randn((a.n, hw, hw, cin), device=dev)
        dy = torch.randn((a.n, hw, hw, cout), device=dev)
        y = torch.empty_like(dy)
        dx = torch.empty_like(x)
        w = torch.randn((k, k, cin, cout), device=dev) * 0.1
        dw = torch.zeros_like(w)
        pad = (k - 1) // 2
        d = L.ConvDesc(a.n, hw, hw, cin, cin, hw, hw, cout, cout, k, k, 1, pad, pad, L.PAD_ZERO, 0, L.ACT_NONE, 0.0, L.ALGO_AUTO)
        pix = a.n * hw * hw
        passes = {
            "fwd": (L.PASS_FWD, 4.0 * pix * (cin + cout),
                    lambda ws: lib.ss_conv2d_fwd(ctypes.byref(d), x.data_ptr(), w.data_ptr(), None, y.data_ptr(), ws.data_ptr(), ws.numel(), None)),
            "dgrad": (L.PASS_BWD_DATA, 4.0 * pix * (cin + cout),
                      lambda ws: lib.ss_conv2d_bwd_data(ctypes.byref(d), dy.data_ptr(), w.data_ptr(), dx.data_ptr(), 0, ws.data_ptr(), ws.numel(), None)),
            "wgrad": (L.PASS_BWD_WEIGHT, 4.0 * pix * (cin + cout),
                      lambda ws: lib.ss_conv2d_bwd_weight(ctypes.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, 0, ws.data_ptr(), ws.numel(), None)),
        }
        for pname, (pass_, nbytes, fn) in passes.items():
            if a.only and a.only != pname:
                continue
            res = []
            for tile in (1, 0):
                with L.config(tile_conv=tile):
                    ws = torch.empty(max(int(lib.ss_conv2d_workspace_bytes(ctypes.byref(d), pass_)), 1 << 20), dtype=torch.uint8, device=dev)
                    rc = fn(ws)
                    assert rc == 0, (name, pname, rc, lib.ss_last_error())
                    res.append(timeit(lambda: fn(ws), a.iters))
            floor = nbytes / 6.3e12 * 1e6
            print(f"{name:22s} {pname:6s} {floor:9.1f} {res[0]:9.1f} {res[1]:9.1f} {nbytes / res[0] / 1e6:10.2f}")
            t = tot.setdefault(pname, [0.0, 0.0, 0.0])
            t[0] += floor; t[1] += res[0]; t[2] += res[1]
    for pname, t in tot.items():
        print(f"{'TOTAL':22s} {pname:6s} {t[0]:9.1f} {t[1]:9.1f} {t[2]:9.1f}")


if __name__ == "__main__":
    main()
