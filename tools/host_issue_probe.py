"""How much of a train step is host issue time?  Wraps Tensor.cpu (the only device->host read of a step: its metrics) and reports,
per CycleGAN + UNet step, the host time until the first blocking read vs the whole step.  Usage: python tools/host_issue_probe.py [global_batch]"""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (synthetic_tiles)

BASE = "automatic-sem-image-segmentation_amd"
E, NETS, CG, UN, OPT = (importlib.import_module(f"{BASE}.{m}") for m in ("engine", "nets", "CycleGAN", "UNet_Segmentation", "optim"))
GB = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
ga = NETS.ResnetGenerator(filters=64, device=dev, seed=1)
gb = NETS.ResnetGenerator(filters=64, device=dev, seed=2)
da = NETS.PatchDiscriminator(filters=128, device=dev, seed=3)
db = NETS.PatchDiscriminator(filters=128, device=dev, seed=4)
unet = NETS.MultiResUNet(16, device=dev, seed=5)
model = CG.CycleGanModel(ga, gb, da, db, image_pool_a=CG.ImagePool(2, 50), image_pool_b=CG.ImagePool(2, 50))
model.compile(OPT.Adam(2e-4, beta_1=0.5), OPT.Adam(2e-4, beta_1=0.5), OPT.Adam(2e-4, beta_1=0.5), OPT.Adam(2e-4, beta_1=0.5))
umodel = UN.UNetModel(unet, 9.0, OPT.Adam(1e-3))
a_all, b_all = bench.synthetic_tiles(GB, 512, 1234)
a = E.Act(a_all.to(dev).contiguous(), requires_grad=False)
b = E.Act(b_all.to(dev).contiguous(), requires_grad=False)
ux, uy = ((a.t + 1) / 2).contiguous(), ((b.t + 1) / 2).contiguous()

first_read = []
orig_cpu = torch.Tensor.cpu


def cpu(self, *args, **kw):
    if self.is_cuda and not first_read:
        first_read.append(time.perf_counter())
    return orig_cpu(self, *args, **kw)


torch.Tensor.cpu = cpu
for what, fn in (("cyclegan", lambda: model.train_step((a, b))), ("unet", lambda: umodel.train_step((ux, uy)))):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    rows = []
    for _ in range(8):
        first_read.clear()
        t0 = time.perf_counter()
        fn()
        t1 = time.perf_counter()
        rows.append(((first_read[0] - t0) * 1e3 if first_read else float("nan"), (t1 - t0) * 1e3))
    rows.sort(key=lambda r: r[1])
    med = rows[len(rows) // 2]
    print(f"global batch {GB}: {what:9s} host until first device read {med[0]:7.2f} ms   step {med[1]:7.2f} ms", flush=True)
