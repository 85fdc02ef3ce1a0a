"""conv_wgrad_stage.hip against wgrad_x6_kernel (ss_config wgrad_stage 1 / 0) and against float64: values and time per weight gradient."""
import ctypes, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "automatic-sem-image-segmentation_amd"
E = importlib.import_module(PKG + ".engine"); LY = importlib.import_module(PKG + ".layers"); L = importlib.import_module(PKG + "._lib")
lib = L.load(); dev = torch.device("cuda:0")
LAYERS = [("g_down1", 3, 64, 128, 2, "same", False, 512, 8), ("g_down2", 3, 128, 256, 2, "same", False, 256, 8), ("d_c2", 4, 128, 256, 2, "same", False, 256, 8),
          ("d_c3", 4, 256, 512, 2, "same", False, 128, 8), ("g_up1", 3, 256, 128, 2, "same", True, 128, 8), ("g_up2", 3, 128, 64, 2, "same", True, 256, 8),
          ("d_c2_valid", 4, 128, 256, 2, "valid", False, 255, 8), ("d_c3_valid", 4, 256, 512, 2, "valid", False, 126, 8),
          ("small_down", 3, 32, 128, 2, "same", False, 64, 2), ("small_4x4", 4, 64, 64, 2, "same", False, 96, 3)]
g = torch.Generator().manual_seed(0)
for name, k, cin, cout, s, pad, tr, hw, n in LAYERS:
    arena = E.ParamArena(dev); conv = LY.Conv2D(arena, "c", k, cin, cout, stride=s, padding=pad, transposed=tr); arena.materialize()
    hh, ww = hw, hw + (32 if name.startswith("small") else 0)
    x = E.Act((torch.randn((n, hh, ww, cin), generator=g) * 0.7).to(dev)); oh, ow = conv.out_hw(hh, ww)
    dy = E.Act((torch.randn((n, oh, ow, cout), generator=g) * 0.3).to(dev))
    d = conv.desc(x, dy)
    gw = arena.grad("c/kernel")
    out, tim = {}, {}
    for mode in (0, 1):
        L.config_set("wgrad_stage", mode)
        nb = lib.ss_conv2d_workspace_bytes(ctypes.byref(d), L.PASS_BWD_WEIGHT)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        d.x_amax, d.x_amax_valid = x.amax_slot(), 0
        d.dy_amax, d.dy_amax_valid = dy.amax_slot(), 0
        def run():
            L.check(lib.ss_conv2d_bwd_weight(ctypes.byref(d), x.ptr, dy.ptr, gw.data_ptr(), None, 0, ws.data_ptr(), ws.numel(), None), "bwd_weight")
        run(); d.x_amax_valid = d.dy_amax_valid = 1; run(); torch.cuda.synchronize()
        out[mode] = gw.detach().cpu().double().clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record(); torch.cuda.synchronize()
        tim[mode] = e0.elapsed_time(e1) * 100
    L.config_set("wgrad_stage", 1)
    rel = float((out[1] - out[0]).abs().max() / out[0].abs().max())
    msg = ""
    if name.startswith("small"):          # float64 truth through torch's CPU convolution
        xc = x.t.cpu().double().permute(0, 3, 1, 2).requires_grad_(False)
        w = torch.zeros((cout, cin, k, k), dtype=torch.float64, requires_grad=True)
        pt = max(k - 1 - ((hh - 1) % s), 0) // 2; pl = max(k - 1 - ((ww - 1) % s), 0) // 2
        xp = torch.nn.functional.pad(xc, (pl, k, pt, k))
        yy = torch.nn.functional.conv2d(xp, w, stride=s)[:, :, :oh, :ow]
        (yy * dy.t.cpu().double().permute(0, 3, 1, 2)).sum().backward()
        truth = w.grad.permute(2, 3, 1, 0)
        msg = " vs float64: staged %.2e, wgrad_x6 %.2e" % (float((out[1] - truth).abs().max() / truth.abs().max()), float((out[0] - truth).abs().max() / truth.abs().max()))
    print(f"{name:10s} wgrad_x6 {tim[0]:6.0f} us  staged {tim[1]:6.0f} us  max|diff| / max|dw| {rel:.2e}{msg}", flush=True)
