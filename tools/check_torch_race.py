"""Does the packed-fp32 hazard (see csrc/Makefile) also hit OTHER people's kernels?  Run torch elementwise kernels (a + b, a * b + c,
the arithmetic an RCCL sum performs) on one stream while a generator forward + backward (bf16-MFMA x6 kernels) runs on another, and
compare bit for bit with the results obtained alone."""
import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "automatic-sem-image-segmentation_amd"
E = importlib.import_module(PKG + ".engine"); NETS = importlib.import_module(PKG + ".nets")
dev = torch.device("cuda:0")
F, S, N = 32, 256, 4
gb = NETS.ResnetGenerator(filters=F, device="cuda:0", seed=2)
b = (torch.rand((2 * N, S, S, 1), device=dev) * 2 - 1); gyb = torch.randn((2 * N, S, S, 1), device=dev)


def other():
    t = E.Tape(); o = gb(E.Act(b, requires_grad=False), True, t); g, _ = o.grad_target(); g.t.copy_(gyb); gb.zero_grad(); t.backward()


x = torch.randn(64 * 1024 * 1024 // 4, device=dev); y = torch.randn_like(x); z = torch.randn_like(x)
ops = {"add": lambda: x + y, "fma": lambda: torch.addcmul(z, x, y), "sum": lambda: (x * y).sum(), "mul_scalar": lambda: x * 0.125}
ref = {k: f().clone() for k, f in ops.items()}
other(); torch.cuda.synchronize()
s1, s2 = E.side_streams(dev)
bad = {k: 0 for k in ops}
for it in range(10):
    cur = torch.cuda.current_stream(); s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s2):
        other()
    outs = []
    with torch.cuda.stream(s1):
        for _ in range(8):
            outs.append({k: f() for k, f in ops.items()})
    torch.cuda.synchronize()
    for o in outs:
        for k in ops:
            bad[k] += int(not torch.equal(o[k], ref[k]))
print("torch kernels concurrent with x6 kernels, mismatching runs of 80:", bad)
