"""Host-side cost of the per-GPU-batch-1 steps (CycleGAN + MultiResUNet, 512 x 512): time the host needs to ISSUE a step (metrics reads off)
next to the wall time, and a cProfile of where it goes.  Usage: python tools/prof_host_b1.py [--size 512] [--batch 1]"""
import argparse, cProfile, pstats, importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "automatic-sem-image-segmentation_amd"
ap = argparse.ArgumentParser(); ap.add_argument("--size", type=int, default=512); ap.add_argument("--batch", type=int, default=1)
a = ap.parse_args()
N = importlib.import_module(PKG + ".nets"); UN = importlib.import_module(PKG + ".UNet_Segmentation"); OPT = importlib.import_module(PKG + ".optim")
CG = importlib.import_module(PKG + ".CycleGAN"); E = importlib.import_module(PKG + ".engine")
dev = torch.device("cuda:0")
ga, gb = N.ResnetGenerator(filters=64, device=dev, seed=1), N.ResnetGenerator(filters=64, device=dev, seed=2)
da, db = N.PatchDiscriminator(filters=128, device=dev, seed=3), N.PatchDiscriminator(filters=128, device=dev, seed=4)
model = CG.CycleGanModel(ga, gb, da, db, image_pool_a=CG.ImagePool(2, 50), image_pool_b=CG.ImagePool(2, 50))
model.compile(OPT.Adam(2e-4, beta_1=0.5), OPT.Adam(2e-4, beta_1=0.5), OPT.Adam(2e-4, beta_1=0.5), OPT.Adam(2e-4, beta_1=0.5))
um = UN.UNetModel(N.MultiResUNet(16, device=dev, seed=5), 9.0, OPT.Adam(1e-3))
x = E.Act((torch.rand((a.batch, a.size, a.size, 1)) * 2 - 1).to(dev), requires_grad=False)
y = E.Act(((torch.rand((a.batch, a.size, a.size, 1)) > 0.9).float() * 2 - 1).to(dev), requires_grad=False)
ux, uy = (x.t + 1) / 2, (y.t + 1) / 2
for name, fn in (("cyclegan", lambda: model.train_step((x, y))), ("unet", lambda: um.train_step((ux, uy)))):
    model.sync_metrics = um.sync_metrics = True
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t) * 100
    model.sync_metrics = um.sync_metrics = False          # no device->host read: the host runs ahead, its own time shows
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10): fn()
    host = (time.perf_counter() - t) * 100
    torch.cuda.synchronize()
    print(f"{name}: wall {wall:.2f} ms per step, host issue {host:.2f} ms per step")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5): fn()
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
