"""MultiResUNet train steps on 16-bit activation storage at the workflow's shape (batch 5, 384 x 384): loss / mae per step, first non-finite value.
Usage: python tools/unet16_nan_probe.py [f16|bf16] [steps]   (environment switches select kernel variants)"""
import importlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "automatic-sem-image-segmentation_amd"
E = importlib.import_module(PKG + ".engine"); UN = importlib.import_module(PKG + ".UNet_Segmentation"); NETS = importlib.import_module(PKG + ".nets")
OPT = importlib.import_module(PKG + ".optim")
dt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[sys.argv[1] if len(sys.argv) > 1 else "f16"]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda:0")
net = NETS.MultiResUNet(16, device=dev, seed=5, act_dtype=dt)
model = UN.UNetModel(net, 9.0, OPT.Adam(1e-3))
g = torch.Generator().manual_seed(3)
data = os.environ.get("PROBE_DATA")
if data:          # real SEM tiles (a staged copy of the publication's images), pseudo masks by threshold: the workflow's value ranges
    from PIL import Image
    files = sorted(os.listdir(os.path.join(data, "Input_Images")))[:8]
    imgs = [np.asarray(Image.open(os.path.join(data, "Input_Images", f)).convert("L"), dtype=np.float32) / 255.0 for f in files]
    tiles = [im[r:r + 384, c:c + 384] for im in imgs for r in (0, 328) for c in (0, 320, 640)]
    X = torch.from_numpy(np.stack(tiles))[..., None]
    Y = (X > X.mean() + 0.1).float()
else:
    X = torch.rand((40, 384, 384, 1), generator=g)
    Y = (torch.rand((40, 384, 384, 1), generator=g) > 0.86).float()
for i in range(steps):
    sel = torch.randint(0, X.shape[0], (5,), generator=g)
    x, y = X[sel], Y[sel]
    m = model.train_step((x.to(dev).to(dt), y.to(dev).to(dt)))
    if i % 20 and all(np.isfinite(v) for v in m.values()):
        continue
    bad = int((~torch.isfinite(net.arena.params)).sum())
    print(i, {k: round(float(v), 5) for k, v in m.items()}, "non-finite params:", bad, flush=True)
    if any(not np.isfinite(v) for v in m.values()):
        break
