"""Diagnostic: errors of the norm passes against a float64 ground truth, with and without the fused finalize (norm_fuse_fin)."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B = "automatic-sem-image-segmentation_amd"
E = importlib.import_module(B + ".engine"); LY = importlib.import_module(B + ".layers"); L = importlib.import_module(B + "._lib")
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)


def truth(x, gy, gam, bet, kind, act, res):
    x = x.double().requires_grad_(True); gam = gam.double().requires_grad_(True); bet = bet.double().requires_grad_(True)
    r = res.double().requires_grad_(True) if res is not None else None
    dims = (1, 2) if kind == "instance" else (0, 1, 2)
    m = x.mean(dims, keepdim=True); v = (x * x).mean(dims, keepdim=True) - m * m
    eps = 1e-5 if kind == "instance" else 1e-3
    z = (x - m) / torch.sqrt(v + eps) * gam + bet
    if r is not None:
        z = z + r
    y = torch.relu(z) if act == "relu" else z
    y.backward(gy.double())
    return y.detach(), x.grad, gam.grad, bet.grad, m.detach().flatten(), (1 / torch.sqrt(v + eps)).detach().flatten()


def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30))


for kind, c, shp, act, use_res, off in (("instance", 4, (4, 64, 64), "relu", False, 0.3), ("instance", 4, (3, 64, 64), "relu", False, 0.3), ("instance", 8, (4, 64, 64), "relu", False, 0.3),
                                         ("instance", 4, (2, 64, 64), "relu", False, 0.3), ("instance", 4, (2, 64, 64), "relu", False, 3.0),
                                         ("instance", 256, (1, 128, 128), "relu", False, 0.3), ("instance", 256, (1, 128, 128), None, True, 0.3),
                                         ("batch", 51, (1, 256, 256), "relu", False, 0.5), ("batch", 17, (2, 64, 64), "relu", False, 0.5),
                                         ("instance", 128, (1, 256, 256), "relu", False, 0.3), ("instance", 64, (2, 128, 128), "relu", False, 1.0)):
    n, h, w = shp
    x = torch.randn((n, h, w, c), generator=g) + off
    gy = torch.randn((n, h, w, c), generator=g)
    res = torch.randn((n, h, w, c), generator=g) if use_res else None
    gam = torch.rand(c, generator=g) + 0.5; bet = torch.rand(c, generator=g) - 0.5
    T = truth(x, gy, gam, bet, kind, act, res)
    for fuse in (0, 1, 2, 3):
        with L.config(norm_fuse_fin=fuse):
            arena = E.ParamArena(dev); layer = LY.Norm(arena, "n", c, kind); arena.materialize()
            arena["n/gamma"].copy_(gam); arena["n/beta"].copy_(bet)
            tape = E.Tape(); xa = E.Act(x.to(dev)); ra = E.Act(res.to(dev)) if use_res else None
            y = layer(tape, xa, act=act, residual=ra)
            gt, _ = y.grad_target(); gt.t.copy_(gy.to(dev)); arena.zero_grad(); tape.backward()
            out = (y.dense().cpu(), xa.get_grad().dense().cpu(), arena.grad("n/gamma").cpu(), arena.grad("n/beta").cpu())
            print(kind, c, shp, act, "res" if use_res else "", "off", off, "fuse", fuse,
                  "y %.2e dx %.2e dgamma %.2e dbeta %.2e" % tuple(rel(a, b) for a, b in zip(out, T[:4])), flush=True)
