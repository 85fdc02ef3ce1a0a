"""Second sweep: strides, paddings, transposed convs and channel counts that select the x6 / Winograd / one-channel kernels,
odd spatial sizes, against the torch CPU fp64 oracle (fwd / dgrad / wgrad rel-L2).  Prints only failures unless VERBOSE=1."""
import importlib, itertools, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ops as O
PKG = "automatic-sem-image-segmentation_amd"
E = importlib.import_module(PKG + ".engine"); LY = importlib.import_module(PKG + ".layers")
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
def rel(a, b): return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
def oracle(x, w, k, s, pad, tr):
    if tr: return O.conv2d_transpose(x, w, None, s)
    if isinstance(pad, tuple): return O.conv2d(O.reflection_pad(x, (2 * pad[1], 2 * pad[1])), w, None, s, "valid")
    return O.conv2d(x, w, None, s, pad)
cases = []
for (n, h, w) in [(2, 70, 66), (1, 129, 131), (3, 45, 52)]:
    for (cin, cout) in [(32, 64), (64, 40), (96, 160), (128, 128)]:
        cases += [(n, h, w, cin, cout, 3, 1, ("reflect", 1), False), (n, h, w, cin, cout, 3, 1, "same", False),
                  (n, h, w, cin, cout, 3, 2, "same", False), (n, h, w, cin, cout, 4, 2, "valid", False),
                  (n, h // 2, w // 2, cin, cout, 3, 2, "same", True), (n, h, w, cin, cout, 1, 1, "same", False)]
for (n, h, w) in [(1, 130, 140), (2, 150, 129)]:
    cases += [(n, h, w, 1, 32, 7, 1, ("reflect", 3), False), (n, h, w, 24, 1, 7, 1, ("reflect", 3), False),
              (n, h, w, 1, 16, 3, 1, "same", False), (n, h, w, 8, 1, 4, 1, "valid", False), (n, h, w, 1, 48, 4, 1, "valid", False)]
bad = 0
for (n, h, w, cin, cout, k, s, pad, tr) in cases:
    arena = E.ParamArena(dev); layer = LY.Conv2D(arena, "c", k, cin, cout, stride=s, padding=pad, transposed=tr); arena.materialize()
    wshape = (k, k, cout, cin) if tr else (k, k, cin, cout)
    wc = (torch.rand(wshape, generator=g, dtype=torch.float64) - 0.5); xc = torch.rand((n, h, w, cin), generator=g, dtype=torch.float64) - 0.5
    arena["c/kernel"].copy_(wc.float())
    xr = xc.clone().requires_grad_(True); wr = wc.clone().requires_grad_(True)
    yr = oracle(xr, wr, k, s, pad, tr); gy = torch.rand(yr.shape, generator=g, dtype=torch.float64) - 0.5; yr.backward(gy)
    tape = E.Tape(); x = E.Act(xc.float().to(dev)); y = layer(tape, x)
    assert tuple(y.t.shape) == tuple(yr.shape), (y.t.shape, yr.shape)
    gt, _ = y.grad_target(); gt.t.copy_(gy.float().to(dev)); arena.zero_grad(); tape.backward(); torch.cuda.synchronize()
    e = (rel(y.dense().cpu().double().numpy(), yr.detach().numpy()), rel(x.get_grad().dense().cpu().double().numpy(), xr.grad.numpy()),
         rel(arena.grad("c/kernel").cpu().double().numpy(), wr.grad.numpy()))
    flag = "BAD" if max(e) > 2e-5 else ""
    bad += bool(flag)
    if flag or os.environ.get("VERBOSE"): print((n, h, w, cin, cout, k, s, pad, tr), ["%.1e" % v for v in e], flag)
print("cases:", len(cases), "bad:", bad)
