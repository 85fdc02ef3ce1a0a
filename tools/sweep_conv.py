"""Sweep conv shapes (fwd / dgrad / wgrad) against torch CPU fp64 to find shape-dependent errors."""
import importlib, os, sys, itertools
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ops as O
PKG = "automatic-sem-image-segmentation_amd"
E = importlib.import_module(PKG + ".engine"); LY = importlib.import_module(PKG + ".layers")
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
def rel(a, b): return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
bad = 0
shapes = [(2, 40, 48), (1, 80, 96), (2, 20, 24), (2, 10, 12), (2, 5, 6), (3, 33, 47)]
chans = [(64, 105, 1), (64, 17, 3), (17, 35, 3), (35, 53, 3), (105, 16, 1), (128, 212, 1), (212, 128, 3), (426, 71, 3), (25, 51, 1), (8, 17, 3), (32, 25, 1)]
for (n, h, w), (cin, cout, k) in itertools.product(shapes, chans):
    arena = E.ParamArena(dev); layer = LY.Conv2D(arena, "c", k, cin, cout, padding="same"); arena.materialize()
    wc = (torch.rand((k, k, cin, cout), generator=g, dtype=torch.float64) - 0.5); xc = torch.rand((n, h, w, cin), generator=g, dtype=torch.float64) - 0.5
    arena["c/kernel"].copy_(wc.float())
    xr = xc.clone().requires_grad_(True); wr = wc.clone().requires_grad_(True)
    yr = O.conv2d(xr, wr, None, 1, "same"); gy = torch.rand(yr.shape, generator=g, dtype=torch.float64) - 0.5; yr.backward(gy)
    tape = E.Tape(); x = E.Act(xc.float().to(dev)); y = layer(tape, x)
    gt, _ = y.grad_target(); gt.t.copy_(gy.float().to(dev)); arena.zero_grad(); tape.backward(); torch.cuda.synchronize()
    e = (rel(y.dense().cpu().double().numpy(), yr.detach().numpy()), rel(x.get_grad().dense().cpu().double().numpy(), xr.grad.numpy()),
         rel(arena.grad("c/kernel").cpu().double().numpy(), wr.grad.numpy()))
    flag = "BAD" if max(e) > 1e-5 else ""
    bad += bool(flag)
    if flag or os.environ.get("VERBOSE"): print((n, h, w), (cin, cout, k), ["%.1e" % v for v in e], flag)
print("bad:", bad)
