"""Do equally-aligned tensors cost bandwidth?  InstanceNorm apply + residual (read x, read residual, write y) and its backward on the trunk
shape at n = 16 (3 x 134 MB: beyond the 256 MiB Infinity Cache), with the three tensors (a) as the caching allocator hands them out
(power-of-two sizes: identical offsets modulo 128 MiB) and (b) carved out of one buffer at skewed offsets.  Usage: python tools/skew_probe.py"""
import importlib, os, sys, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "automatic-sem-image-segmentation_amd"
E = importlib.import_module(PKG + ".engine"); LY = importlib.import_module(PKG + ".layers"); L = importlib.import_module(PKG + "._lib")
dev = torch.device("cuda:0")
lib = L.load()


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for n in (8, 16):
    shape = (n, 64, 64, 512)
    numel = n * 64 * 64 * 512
    for skew in (0, 1 << 12, (1 << 12) + 256, 1 << 16, (1 << 20) + (1 << 12)):
        big = torch.empty(3 * numel + 3 * (skew // 4) + 1024, dtype=torch.float32, device=dev)
        views = []
        for i in range(3):
            off = i * (numel + skew // 4)
            views.append(big[off:off + numel].view(shape))
        x, r, y = views
        x.normal_(); r.normal_()
        d = L.NormDesc(n, 64, 64, 512, 512, 512, 512, n, 1e-5, L.ACT_NONE if hasattr(L, "ACT_NONE") else 0, 0.0, dtype=0)
        mean = torch.zeros(n * 512, device=dev); rstd = torch.ones(n * 512, device=dev)
        gamma = torch.ones(512, device=dev); beta = torch.zeros(512, device=dev)
        fn = lambda: L.check(lib.ss_norm_apply(ctypes.byref(d), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(gamma.data_ptr()), ctypes.c_void_p(beta.data_ptr()),
                                               ctypes.c_void_p(r.data_ptr()), ctypes.c_void_p(y.data_ptr()), ctypes.c_void_p(mean.data_ptr()),
                                               ctypes.c_void_p(rstd.data_ptr()), E._stream()), "apply")
        us = timeit(fn)
        print(f"n={n} skew {skew:8d} B: norm_apply+residual {us:7.1f} us = {3 * numel * 4 / us / 1e6:.2f} TB/s")
        del big, views, x, r, y


# cold / producer-hot variants of the n = 16 case: events around the apply launch only
n = 16
shape = (n, 64, 64, 512); numel = n * 64 * 64 * 512
x, r, y, src = (torch.empty(shape, device=dev) for _ in range(4))
x.normal_(); r.normal_(); src.normal_()
flush = torch.empty(768 << 20, dtype=torch.uint8, device=dev)
d = L.NormDesc(n, 64, 64, 512, 512, 512, 512, n, 1e-5, 0, 0.0, dtype=0)
mean = torch.zeros(n * 512, device=dev); rstd = torch.ones(n * 512, device=dev); gamma = torch.ones(512, device=dev); beta = torch.zeros(512, device=dev)
amax = torch.zeros(4096, dtype=torch.int32, device=dev)
def apply(with_amax=False):
    d.y_amax = amax.data_ptr() if with_amax else None
    L.check(lib.ss_norm_apply(ctypes.byref(d), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(gamma.data_ptr()), ctypes.c_void_p(beta.data_ptr()),
                              ctypes.c_void_p(r.data_ptr()), ctypes.c_void_p(y.data_ptr()), ctypes.c_void_p(mean.data_ptr()), ctypes.c_void_p(rstd.data_ptr()), E._stream()), "apply")
for name, pre, am in (("back to back", lambda: None, False), ("after a flush of 768 MB (all cold)", lambda: flush.fill_(1), False),
                      ("x written just before (copy), residual cold", lambda: (flush.fill_(1), x.copy_(src)), False),
                      ("all cold + amax slot", lambda: flush.fill_(1), True)):
    ts = []
    for _ in range(8):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); apply(am); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print(f"n=16 {name}: {ts[len(ts) // 2]:.1f} us")

# n = 16 as two launches over 8 samples each (InstanceNorm: groups are samples): does keeping a launch inside the Infinity Cache pay?
d8 = L.NormDesc(8, 64, 64, 512, 512, 512, 512, 8, 1e-5, 0, 0.0, dtype=0)
half = 8 * 64 * 64 * 512
def apply_halves():
    for h in (1, 0):          # back to front, as the kernel itself walks
        o = h * half * 4
        L.check(lib.ss_norm_apply(ctypes.byref(d8), ctypes.c_void_p(x.data_ptr() + o), ctypes.c_void_p(gamma.data_ptr()), ctypes.c_void_p(beta.data_ptr()),
                                  ctypes.c_void_p(r.data_ptr() + o), ctypes.c_void_p(y.data_ptr() + o), ctypes.c_void_p(mean.data_ptr() + h * 8 * 512 * 4),
                                  ctypes.c_void_p(rstd.data_ptr() + h * 8 * 512 * 4), E._stream()), "apply")
for name, pre in (("back to back", lambda: None), ("after a flush of 768 MB (all cold)", lambda: flush.fill_(1)),
                  ("x written just before (copy), residual cold", lambda: (flush.fill_(1), x.copy_(src)))):
    ts = []
    for _ in range(8):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); apply_halves(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print(f"n=16 as 2 x 8, {name}: {ts[len(ts) // 2]:.1f} us")
