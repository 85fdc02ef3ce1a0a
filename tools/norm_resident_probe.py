"""InstanceNorm backward at the trunk's shapes: one-pass register-resident form (ss_config norm_bwd_resident = 1) against the statistics +
apply kernels (0), HIP events, cycling over 6 buffer sets (nothing cache-resident)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "automatic-sem-image-segmentation_amd"
E = importlib.import_module(PKG + ".engine"); LY = importlib.import_module(PKG + ".layers"); L = importlib.import_module(PKG + "._lib")
dev = torch.device("cuda:0")


def timeit(fn, iters=30):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for n, hw, c in [(8, 128, 256), (16, 128, 256), (1, 128, 256), (4, 64, 256)]:
    arena = E.ParamArena(dev); norm = LY.Norm(arena, "n", c, "instance"); arena.materialize(); arena["n/gamma"].fill_(1.0)
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
    fu = timeit(f)
    row = []
    for mode in (0, 1):
        L.config_set("norm_bwd_resident", mode)
        row.append(timeit(fb) - fu)
    L.config_set("norm_bwd_resident", 1)
    mb = n * hw * hw * c * 4 / 1e6
    print(f"n={n} {hw}x{hw} c={c} ({mb:.0f} MB per tensor): backward two-pass {row[0]:7.1f} us, one-pass {row[1]:7.1f} us ({3 * mb / row[1]:.2f} TB/s on 3 passes); timeouts {L.load().ss_norm_resident_timeouts()}", flush=True)
