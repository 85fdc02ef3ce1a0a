"""Run the same conv layer forward + backward concurrently on two HIP streams (different layers / inputs per stream) and compare
bit for bit with the results obtained alone: a cross-stream race inside the library shows up as mismatches."""
import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "automatic-sem-image-segmentation_amd"
E = importlib.import_module(PKG + ".engine"); LY = importlib.import_module(PKG + ".layers")
dev = torch.device("cuda:0")


def mk(k, cin, cout, s, pad, tr=False):
    a = E.ParamArena(dev); c = LY.Conv2D(a, "c", k, cin, cout, stride=s, padding=pad, transposed=tr); a.materialize()
    a["c/kernel"].uniform_(-0.05, 0.05)
    return a, c


def run(a, c, x, gy):
    tape = E.Tape(); x.grad = None; x.grad_init = False
    y = c(tape, x); gt, _ = y.grad_target(); gt.t.copy_(gy); a.zero_grad(); tape.backward()
    return y.dense().clone(), x.get_grad().dense().clone(), a.grad("c/kernel").clone()


for (k, cin, cout, s, pad, hw, tr) in [(7, 1, 32, 1, ("reflect", 3), 256, False), (7, 32, 1, 1, ("reflect", 3), 256, False), (3, 64, 128, 2, "same", 128, False), (3, 128, 128, 1, ("reflect", 1), 64, False),
                                       (4, 64, 128, 2, "valid", 127, False), (3, 128, 64, 2, "same", 64, True)]:
    a1, c1 = mk(k, cin, cout, s, pad, tr); a2, c2 = mk(k, cin, cout, s, pad, tr)
    x1 = E.Act(torch.randn((4, hw, hw, cin), device=dev), requires_grad=True); x2 = E.Act(torch.randn((4, hw, hw, cin), device=dev), requires_grad=True)
    oh = c1(E.Tape(enabled=False), x1)
    g1 = torch.randn_like(oh.t); g2 = torch.randn_like(oh.t)
    r1 = run(a1, c1, x1, g1); r2 = run(a2, c2, x2, g2)
    torch.cuda.synchronize()
    s1, s2 = E.side_streams(dev)
    bad = [0, 0, 0]
    for it in range(20):
        s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s1):
            o1 = run(a1, c1, x1, g1)
        with torch.cuda.stream(s2):
            o2 = run(a2, c2, x2, g2)
        torch.cuda.synchronize()
        for i in range(3):
            bad[i] += int(not torch.equal(o1[i], r1[i])) + int(not torch.equal(o2[i], r2[i]))
    print((k, cin, cout, s, pad, hw, tr), "mismatches (y, dx, dw) of 40:", bad)
