"""How much of a train step's wall time is the host busy issuing launches (vs blocked in the one device->host read per model)?
Usage: python tools/host_busy.py [--batch 1] [--size 512]"""
import argparse
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "automatic-sem-image-segmentation_amd"
E = importlib.import_module(PKG + ".engine")
CG = importlib.import_module(PKG + ".CycleGAN")
UN = importlib.import_module(PKG + ".UNet_Segmentation")
NETS = importlib.import_module(PKG + ".nets")
OPT = importlib.import_module(PKG + ".optim")

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda:0")
ga, gb = NETS.ResnetGenerator(filters=64, device=dev, seed=1), NETS.ResnetGenerator(filters=64, device=dev, seed=2)
da, db = NETS.PatchDiscriminator(filters=128, device=dev, seed=3), NETS.PatchDiscriminator(filters=128, device=dev, seed=4)
unet = NETS.MultiResUNet(16, device=dev, seed=5)
model = CG.CycleGanModel(ga, gb, da, db, image_pool_a=CG.ImagePool(2, 50), image_pool_b=CG.ImagePool(2, 50))
model.compile(*[OPT.Adam(2e-4, beta_1=0.5) for _ in range(4)])
um = UN.UNetModel(unet, 9.0, OPT.Adam(1e-3))
x = torch.rand((a.batch, a.size, a.size, 1), device=dev) * 2 - 1
y = torch.rand((a.batch, a.size, a.size, 1), device=dev) * 2 - 1
A, B = E.Act(x, requires_grad=False), E.Act(y, requires_grad=False)
ux, uy = (x + 1) / 2, ((y + 1) / 2 > 0.9).float()

blocked = [0.0]
_cpu = torch.Tensor.cpu


def timed_cpu(self, *args, **kw):
    t = time.perf_counter()
    r = _cpu(self, *args, **kw)
    blocked[0] += time.perf_counter() - t
    return r


torch.Tensor.cpu = timed_cpu
for which, fn in (("cyclegan", lambda: model.train_step((A, B))), ("unet", lambda: um.train_step((ux, uy)))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    blocked[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(a.steps):
        fn()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / a.steps * 1e3
    blk = blocked[0] / a.steps * 1e3
    print(f"{which:9s} batch {a.batch} {a.size}^2: wall {wall:7.2f} ms/step, host blocked on the GPU {blk:7.2f} ms, host busy {wall - blk:7.2f} ms")
