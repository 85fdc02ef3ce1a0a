"""Winograd weight-gradient GEMM (gemm_tn_x3h) on the trunk shape: time by config (gemm_ilv, saved operand on / off)."""
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
    for save in (True,):
        for ilv in (0, 16, 4, 0, 16):
            LY.SAVE_OPERAND = save
            with L.config(tile_dbg=ilv):
                arena = E.ParamArena(dev)
                conv = LY.Conv2D(arena, "c", 3, 512, 512, padding=("reflect", 1), use_bias=False)
                arena.materialize()
                arena["c/kernel"].uniform_(-0.05, 0.05)
                x = E.Act(torch.randn((n, 64, 64, 512), device=dev), requires_grad=False)
                dyt = torch.randn((n, 64, 64, 512), device=dev)

                def step():
                    tape = E.Tape()
                    y = conv(tape, x)
                    gt, _ = y.grad_target()
                    gt.t.copy_(dyt)
                    tape.backward()
                step(); step()
                torch.cuda.synchronize()
                lib.ss_prof_reset(); lib.ss_prof_enable(1)
                for _ in range(6):
                    step()
                torch.cuda.synchronize()
                lib.ss_prof_enable(0)
                p = L.prof_summary()
                print(f"n={n} save={int(save)} ilv={ilv}: " + "  ".join(f"{k} {v['avg_ms'] * 1e3:.1f}us" for k, v in p.items() if k.startswith("gemm_")), flush=True)
LY.SAVE_OPERAND = True
