"""Where wgrad_x6_kernel's time goes: the weight gradient of the stride-2 / 4x4 layers timed alone (HIP events) with the kernel's
phase-skipping measurement bits (ss_config tile_dbg: 1 = no global loads, 2 = no split / LDS stores, 4 = no fragment reads / MFMAs)."""
import ctypes, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "automatic-sem-image-segmentation_amd"
E = importlib.import_module(PKG + ".engine"); LY = importlib.import_module(PKG + ".layers"); L = importlib.import_module(PKG + "._lib")
lib = L.load(); dev = torch.device("cuda:0")
LAYERS = [("g_down1", 3, 64, 128, 2, "same", False, 512), ("g_down2", 3, 128, 256, 2, "same", False, 256), ("d_c2", 4, 128, 256, 2, "same", False, 256),
          ("d_c3", 4, 256, 512, 2, "same", False, 128), ("g_up1", 3, 256, 128, 2, "same", True, 128)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for name, k, cin, cout, s, pad, tr, hw in LAYERS:
    arena = E.ParamArena(dev); conv = LY.Conv2D(arena, "c", k, cin, cout, stride=s, padding=pad, transposed=tr); arena.materialize()
    x = E.Act(torch.randn((n, hw, hw, cin), device=dev)); oh, ow = conv.out_hw(hw, hw)
    dy = E.Act(torch.randn((n, oh, ow, cout), device=dev))
    d = conv.desc(x, dy)
    gw = arena.grad("c/kernel")
    row = []
    for dbg in (0, 1, 2, 4, 3, 6, 5):
        L.config_set("tile_dbg", dbg)
        nb = lib.ss_conv2d_workspace_bytes(ctypes.byref(d), L.PASS_BWD_WEIGHT)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        d.x_amax, d.x_amax_valid = x.amax_slot(), 0
        d.dy_amax, d.dy_amax_valid = dy.amax_slot(), 0
        def run():
            L.check(lib.ss_conv2d_bwd_weight(ctypes.byref(d), x.ptr, dy.ptr, gw.data_ptr(), None, 0, ws.data_ptr(), ws.numel(), None), "bwd_weight")
        run(); d.x_amax_valid = d.dy_amax_valid = 1; run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record(); torch.cuda.synchronize()
        row.append("%d:%.0f" % (dbg, e0.elapsed_time(e1) * 100))
    L.config_set("tile_dbg", 0)
    print(name, "us per weight gradient (incl. the reduce launch) by tile_dbg:", " ".join(row), flush=True)
