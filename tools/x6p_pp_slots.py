"""Slot-phase timing of the ping-pong GEMM (tile_dbg & 512): cycle sums of waves 0 / 4 of workgroup 0 over one launch."""
import ctypes, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "automatic-sem-image-segmentation_amd"
E = importlib.import_module(PKG + ".engine")
LY = importlib.import_module(PKG + ".layers")
L = importlib.import_module(PKG + "._lib")
dev = torch.device("cuda:0")
lib = L.load()
names = ["dma_issue(+epi)", "frag_reads", "vmcnt_wait", "barrier_M", "mfma_issue", "barrier_C"]
for n in (8, 16):
    arena = E.ParamArena(dev)
    conv = LY.Conv2D(arena, "c", 3, 512, 512, padding=("reflect", 1))
    arena.materialize()
    arena["c/kernel"].uniform_(-0.05, 0.05)
    x = E.Act(torch.randn((n, 64, 64, 512), device=dev))
    for dbg in (0, 32, 192, 224):
        with L.config(x6p_pp=1, tile_dbg=dbg | 512):
            for _ in range(3):
                conv(E.Tape(enabled=False), x)
            torch.cuda.synchronize()
            out = (ctypes.c_ulonglong * 16)()
            lib.ss_dbg_x6p_slots(out)
            for g in (0, 1):
                v = [out[g * 8 + k] for k in range(6)]
                tot = sum(v)
                print(f"n={n} dbg={dbg:3d} group{g}: total {tot} cycles; " + "  ".join(f"{nm} {100.0 * a / max(tot, 1):.1f}%" for nm, a in zip(names, v)), flush=True)
