"""A/B of the interleaved-fragment-read variant of the one-phase Winograd GEMM (tile_dbg & 4) on the trunk shapes: bit-identical
results, times from the library's HIP-event recorder; with the phase-skipping bits for the breakdown."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "automatic-sem-image-segmentation_amd"
E = importlib.import_module(PKG + ".engine")
LY = importlib.import_module(PKG + ".layers")
L = importlib.import_module(PKG + "._lib")
dev = torch.device("cuda:0")
lib = L.load()
for n in (8, 16):
    arena = E.ParamArena(dev)
    conv = LY.Conv2D(arena, "c", 3, 512, 512, padding=("reflect", 1))
    arena.materialize()
    arena["c/kernel"].uniform_(-0.05, 0.05)
    x = E.Act(torch.randn((n, 64, 64, 512), device=dev))
    outs = {}
    for rep in range(2):
        for ilv in (0, 1):
            row = []
            for dbg in (0, 2, 32):
                with L.config(tile_dbg=dbg, gemm_ilv=ilv):
                    y = conv(E.Tape(enabled=False), x)
                    torch.cuda.synchronize()
                    if dbg == 0:
                        outs[ilv] = y.t.clone()
                    lib.ss_prof_reset(); lib.ss_prof_enable(1)
                    for _ in range(10):
                        conv(E.Tape(enabled=False), x)
                    torch.cuda.synchronize()
                    lib.ss_prof_enable(0)
                    p = L.prof_summary()
                    row.append((dbg, [round(v["avg_ms"] * 1e3, 1) for k, v in p.items() if k.startswith("gemm_x6p")]))
            print(f"n={n} ilv={ilv}: " + "  ".join(f"dbg{d}={t}" for d, t in row), flush=True)
    print("   bit-identical:", torch.equal(outs[0], outs[1]), flush=True)
