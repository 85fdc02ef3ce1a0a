"""Time InstanceNorm / BatchNorm forward and forward+backward for the small tensors of a per-GPU batch-1/2 step (HIP events).
Usage: SS_NORM_FUSED_PIX=<max pixels per group for the one-launch kernels, 0 = off> python tools/bench_norm.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "automatic-sem-image-segmentation_amd"
E = importlib.import_module(PKG + ".engine")
LY = importlib.import_module(PKG + ".layers")

# kind, n, h=w, c, act
import sys as _sys
SHAPES_BIG = [("instance", 16, 64, 512, "relu"), ("instance", 8, 64, 512, "relu"), ("instance", 16, 64, 512, None), ("batch", 8, 512, 16, "relu"), ("batch", 8, 256, 51, "relu")] + [("instance", n, 128, 256, "relu") for n in (4, 8)] + [("instance", n, 256, 128, "relu") for n in (1, 2, 4, 8)] + [("batch", 8, 512, 25, "relu"), ("batch", 2, 512, 25, "relu")]
SHAPES = SHAPES_BIG if "--big" in _sys.argv else [
    ("instance", 1, 64, 512, "relu"), ("instance", 2, 64, 512, "relu"), ("instance", 1, 32, 512, "relu"),
    ("instance", 1, 128, 256, "relu"), ("instance", 1, 64, 256, "lrelu"), ("instance", 1, 62, 512, "lrelu"),
    ("instance", 1, 16, 512, "relu"), ("instance", 4, 32, 512, "relu"),
    ("batch", 1, 64, 128, "relu"), ("batch", 1, 32, 213, "relu"), ("batch", 1, 64, 35, "relu"), ("batch", 1, 32, 426, None),
]


def timeit(fn, iters=50):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    print(f"SS_NORM_FUSED_PIX={os.environ.get('SS_NORM_FUSED_PIX', '(default)')}")
    print(f"{'kind':9s} {'n':>2s} {'hw':>4s} {'c':>4s} {'fwd us':>8s} {'fwd+bwd us':>11s}")
    for kind, n, hw, c, act in SHAPES:
        arena = E.ParamArena(dev)
        norm = LY.Norm(arena, "n", c, kind)
        arena.materialize()
        arena["n/gamma"].fill_(1.0)
        x = E.Act(torch.randn((n, hw, hw, c), device=dev), requires_grad=True)
        y = E.Act.empty(n, hw, hw, c, dev)
        if "--cycle" in sys.argv:          # a different input / output buffer every call (16 of each): nothing is cache- or TLB-resident
            xs = [E.Act(torch.randn((n, hw, hw, c), device=dev)) for _ in range(16)]
            ys = [E.Act.empty(n, hw, hw, c, dev) for _ in range(16)]
            k = [0]

            def fcyc():
                k[0] = (k[0] + 1) % 16
                norm(E.Tape(enabled=False), xs[k[0]], act=act, act_alpha=0.2, out=ys[k[0]])
            print(f"{kind:9s} {n:2d} {hw:4d} {c:4d} {timeit(fcyc):8.1f}  (cycling 16 buffers)")
            continue
        f_us = timeit(lambda: norm(E.Tape(enabled=False), x, act=act, act_alpha=0.2, out=y))

        def fb():
            t = E.Tape()
            yy = norm(t, x, act=act, act_alpha=0.2, out=y)
            yy.grad = None
            yy.grad_target()
            x.grad_init = False
            t.backward()
        fb_us = timeit(fb)
        print(f"{kind:9s} {n:2d} {hw:4d} {c:4d} {f_us:8.1f} {fb_us:11.1f}")


if __name__ == "__main__":
    main()
