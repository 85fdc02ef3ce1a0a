"""Sub-pixel phases of the stride-2 data gradients / transposed convolutions in ONE launch (config key gconv_phases) vs one launch per
phase: results must be bit-identical; times from HIP events."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "automatic-sem-image-segmentation_amd"
E = importlib.import_module(PKG + ".engine")
LY = importlib.import_module(PKG + ".layers")
L = importlib.import_module(PKG + "._lib")
dev = torch.device("cuda:0")
CASES = [("g_up3", 3, 128, 64, 2, "same", True, 256, 8), ("g_up2", 3, 256, 128, 2, "same", True, 128, 8), ("g_up1", 3, 512, 256, 2, "same", True, 64, 8),
         ("g_down1", 3, 64, 128, 2, "same", False, 512, 8), ("g_down2", 3, 128, 256, 2, "same", False, 256, 8), ("d_c2", 4, 128, 256, 2, "valid", False, 255, 8),
         ("d_c3", 4, 256, 512, 2, "valid", False, 126, 8), ("odd", 3, 64, 96, 2, "same", False, 67, 3), ("up_odd", 3, 96, 64, 2, "same", True, 33, 2)]


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


ok = True
for name, k, cin, cout, s, pad, tr, hw, n in CASES:
    g = torch.Generator().manual_seed(hw + cin)
    wt = torch.empty((k, k, cout, cin) if tr else (k, k, cin, cout)).uniform_(-0.05, 0.05, generator=g)
    xt = torch.randn((n, hw, hw, cin), generator=g)
    res, tm = {}, {}
    for ph in (0, 1):
        with L.config(gconv_phases=ph):
            arena = E.ParamArena(dev)
            conv = LY.Conv2D(arena, "c", k, cin, cout, stride=s, padding=pad, transposed=tr)
            arena.materialize()
            arena["c/kernel"].copy_(wt)
            x = E.Act(xt.to(dev), requires_grad=True)
            tape = E.Tape()
            y = conv(tape, x)
            gt, _ = y.grad_target()
            gdy = torch.Generator().manual_seed(7)
            gt.t.copy_(torch.randn(gt.t.shape, generator=gdy).to(dev))
            arena.zero_grad()
            tape.backward()
            torch.cuda.synchronize()
            res[ph] = (y.dense().clone(), x.get_grad().dense().clone())
            tm[ph] = (timeit(lambda: conv(E.Tape(enabled=False), x)),)
    same = torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    ok &= same
    print(f"{name:8s} fwd ms per-phase {tm[0][0]:.3f}  joint {tm[1][0]:.3f}   bit-identical y/dx: {same}", flush=True)
print("ALL_OK" if ok else "MISMATCH")
