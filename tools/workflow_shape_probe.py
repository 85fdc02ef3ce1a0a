"""The two train steps at the workflow's default shape (batch 5 x 384 x 384, StartProcess.py defaults) the way the training loops run them
-- numpy batches from the loader, metrics read every step -- against the same steps without the per-step device->host read and with
device-resident inputs: what the host side of a step costs.  Usage: python tools/workflow_shape_probe.py [batch] [size]"""
import importlib, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BASE = "automatic-sem-image-segmentation_amd"
E, NETS, CG, UN, OPT = (importlib.import_module(f"{BASE}.{m}") for m in ("engine", "nets", "CycleGAN", "UNet_Segmentation", "optim"))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 5
S = int(sys.argv[2]) if len(sys.argv) > 2 else 384
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
ga = NETS.ResnetGenerator(filters=64, device=dev, seed=1, use_skip_connection=False)
gb = NETS.ResnetGenerator(filters=64, device=dev, seed=2, use_skip_connection=False)
da = NETS.PatchDiscriminator(filters=128, device=dev, seed=3)
db = NETS.PatchDiscriminator(filters=128, device=dev, seed=4)
model = CG.CycleGanModel(ga, gb, da, db, image_pool_a=CG.ImagePool(B, 50), image_pool_b=CG.ImagePool(B, 50), lambda_identity_a=0.5, lambda_identity_b=0.5)
model.compile(OPT.Adam(2e-4, beta_1=0.5), OPT.Adam(2e-4, beta_1=0.5), OPT.Adam(2e-4, beta_1=0.5), OPT.Adam(2e-4, beta_1=0.5))
umodel = UN.UNetModel(NETS.MultiResUNet(16, device=dev, seed=5), 9.0, OPT.Adam(1e-3))
batches = [((rng.random((B, S, S, 1), dtype=np.float32) * 2 - 1), ((rng.random((B, S, S, 1)) > 0.8).astype(np.float32) * 2 - 1)) for _ in range(4)]
ubatches = [((a + 1) / 2, (b + 1) / 2) for a, b in batches]


def run(m, data, steps, sync, resident):
    if resident:
        data = [tuple(E.Act(torch.from_numpy(t).to(dev).contiguous(), requires_grad=False) for t in pair) for pair in data]
    m.sync_metrics = sync
    for i in range(4):
        m.train_step(data[i % 4])
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(steps):
        m.train_step(data[i % 4])
    torch.cuda.synchronize()
    m.sync_metrics = True
    return (time.perf_counter() - t) / steps * 1e3


for name, m, data in (("cyclegan", model, batches), ("unet", umodel, ubatches)):
    for sync, resident in ((True, False), (False, False), (False, True), (True, False)):
        print(f"{name:9s} batch {B} x {S}: metrics read each step={sync!s:5} inputs on device={resident!s:5}: {run(m, data, 30, sync, resident):7.2f} ms per step", flush=True)
