"""Phase-skipping breakdown of the two x3h Winograd GEMM kernels (tile_dbg bits: 32 = no C stores, 64 = no B DMA, 128 = no A DMA):
results are wrong with any bit set; only the times matter."""
import ctypes, importlib, os, sys
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
    for pp in (0, 1, 3):
        row = []
        for dbg in (0, 8, 32, 192, 224):
            with L.config(x6p_pp=pp, tile_dbg=dbg):
                conv(E.Tape(enabled=False), x)
                torch.cuda.synchronize()
                lib.ss_prof_reset(); lib.ss_prof_enable(1)
                for _ in range(10):
                    conv(E.Tape(enabled=False), x)
                torch.cuda.synchronize()
                lib.ss_prof_enable(0)
                p = L.prof_summary()
                row.append((dbg, [round(v["avg_ms"] * 1e3, 1) for k, v in p.items() if k.startswith("gemm_x6p")]))
        print(f"n={n} pp={pp}: " + "  ".join(f"dbg{d}={t}" for d, t in row), flush=True)
