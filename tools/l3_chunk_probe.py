"""Does running a trunk layer over sample chunks (working set below the 256 MiB Infinity Cache) beat one pass over the whole batch?
Times conv (Winograd trunk) forward / forward+backward and InstanceNorm forward+backward on (8, 128, 128, 256), whole batch
vs chunks of 4 / 2 / 1 samples, one stream.  Usage (GPU box): python tools/l3_chunk_probe.py"""
import importlib
import sys, os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BASE = "automatic-sem-image-segmentation_amd"
E, LY, L = (importlib.import_module(f"{BASE}.{m}") for m in ("engine", "layers", "_lib"))
dev = torch.device("cuda:0")


def timed(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    n, h, w, c = 8, 128, 128, 256
    arena = E.ParamArena(dev)
    conv = LY.Conv2D(arena, "c", 3, c, c, padding=("reflect", 1), use_bias=False)
    norm = LY.Norm(arena, "n", c, "instance")
    arena.materialize()
    arena["c/kernel"].normal_(0, 0.02)
    arena["n/gamma"].fill_(1.0)
    xt = torch.randn((n, h, w, c), device=dev)
    gt = torch.randn((n, h, w, c), device=dev)

    def run(chunk, what, bwd):
        def f():
            for n0 in range(0, n, chunk):
                tape = E.Tape(enabled=bwd)
                x = E.Act(xt[n0:n0 + chunk], requires_grad=bwd)
                y = x
                if "conv" in what:
                    y = conv(tape, y)
                if "norm" in what:
                    y = norm(tape, y, act="relu")
                if bwd:
                    g, _ = y.grad_target()
                    g.t.copy_(gt[n0:n0 + chunk])
                    arena.zero_grad() if False else None
                    tape.backward()
        return f

    for what in ("conv", "norm", "conv+norm"):
        for bwd in (False, True):
            row = []
            for chunk in (8, 4, 2, 1):
                row.append(f"chunk {chunk}: {timed(run(chunk, what, bwd)):8.1f} us")
            print(f"{what:10s} {'fwd+bwd' if bwd else 'fwd    '}  " + "   ".join(row), flush=True)


main()
