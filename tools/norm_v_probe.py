"""BatchNorm forward / forward+backward at the MultiResUNet's full-resolution shapes: channel counts that are multiples of 4 (16-byte
accesses, V = 4) against their odd neighbours (4-byte accesses, V = 1).  Prints us and the algorithmic TB/s (fwd: 3 passes over the
tensor -- statistics read, apply read + write; bwd: 5 -- statistics read dy + x, apply read dy + x, write dx)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "automatic-sem-image-segmentation_amd"
E = importlib.import_module(PKG + ".engine"); LY = importlib.import_module(PKG + ".layers")
dev = torch.device("cuda:0")


def timeit(fn, iters=30):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for n, hw, c in [(8, 512, 16), (8, 512, 17), (8, 512, 24), (8, 512, 25), (8, 512, 13), (8, 512, 8), (8, 256, 52), (8, 256, 51), (8, 256, 36), (8, 256, 35), (8, 128, 104), (8, 128, 105)]:
    arena = E.ParamArena(dev); norm = LY.Norm(arena, "n", c, "batch"); arena.materialize(); arena["n/gamma"].fill_(1.0)
    xs = [E.Act(torch.randn((n, hw, hw, c), device=dev), requires_grad=True) for _ in range(6)]
    ys = [E.Act.empty(n, hw, hw, c, dev) for _ in range(6)]
    k = [0]

    def f():
        k[0] = (k[0] + 1) % 6
        norm(E.Tape(enabled=False), xs[k[0]], act="relu", out=ys[k[0]])

    def fb():
        k[0] = (k[0] + 1) % 6
        t = E.Tape(); x = xs[k[0]]
        yy = norm(t, x, act="relu", out=ys[k[0]]); yy.grad = None; yy.grad_target(); x.grad_init = False
        t.backward()
    fu, fbu = timeit(f), timeit(fb)
    mb = n * hw * hw * c * 4 / 1e6
    print(f"n={n} {hw}x{hw} c={c:4d} ({mb:6.1f} MB)  fwd {fu:7.1f} us = {3 * mb / fu:5.2f} TB/s   fwd+bwd {fbu:7.1f} us  bwd alone {fbu - fu:7.1f} us = {5 * mb / (fbu - fu):5.2f} TB/s", flush=True)
