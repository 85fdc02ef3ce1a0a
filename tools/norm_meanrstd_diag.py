"""Diagnostic: mean / rstd that ss_norm_fwd leaves for the backward pass, fused finalize off / on, against float64."""
import ctypes, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B = "automatic-sem-image-segmentation_amd"
L = importlib.import_module(B + "._lib"); E = importlib.import_module(B + ".engine")
lib = L.load(); dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
for (n, h, w, c), groups in (((2, 64, 64, 4), 2), ((4, 64, 64, 4), 4), ((4, 64, 64, 4), 1), ((3, 40, 40, 256), 3), ((2, 64, 64, 51), 1)):
    x = (torch.randn((n, h, w, c), generator=g) * 0.7 + 0.4).to(dev)
    gam = torch.ones(c, device=dev); bet = torch.zeros(c, device=dev)
    xd = x.double().view(groups, -1, c)
    m64 = xd.mean(1); v64 = (xd * xd).mean(1) - m64 * m64
    eps = 1e-5
    for fuse in (0, 1):
        with L.config(norm_fuse_fin=fuse):
            d = L.NormDesc(n, h, w, c, c, c, 0, groups, eps, L.ACT_RELU, 0.0)
            y = torch.empty_like(x); mean = torch.full((groups * c,), 7.0, device=dev); rstd = torch.full((groups * c,), 7.0, device=dev)
            ws = torch.empty(lib.ss_norm_workspace_bytes(ctypes.byref(d)), dtype=torch.uint8, device=dev)
            L.check(lib.ss_norm_fwd(ctypes.byref(d), x.data_ptr(), gam.data_ptr(), bet.data_ptr(), None, y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), None, None,
                                    0.99, ws.data_ptr(), ws.numel(), None), "norm_fwd")
            torch.cuda.synchronize()
            em = float((mean.double().view(groups, c) - m64).abs().max()); er = float((rstd.double().view(groups, c) * torch.sqrt(v64 + eps) - 1).abs().max())
            yt = torch.relu((xd - m64[:, None]) / torch.sqrt(v64 + eps)[:, None]).view(n, h, w, c)
            print((n, h, w, c), "groups", groups, "fuse", fuse, "mean err %.2e rstd rel err %.2e y err %.2e" % (em, er, float((y.double() - yt).abs().max())), flush=True)
