"""How close do the MultiResUNet's convolution outputs (pre-BatchNorm) and gradients come to fp16's largest finite value (65504)?  fp32 storage, the
workflow's shape (batch 5, 384 x 384), real SEM tiles with threshold masks (PROBE_DATA=<dir with Input_Images/>) or random data; the gradients are
multiplied by the fp16 loss scale (1024) before the comparison.  Prints the largest |value| per layer over the run and the step it occurred at."""
import importlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "automatic-sem-image-segmentation_amd"
E = importlib.import_module(PKG + ".engine"); UN = importlib.import_module(PKG + ".UNet_Segmentation"); NETS = importlib.import_module(PKG + ".nets")
OPT = importlib.import_module(PKG + ".optim"); LY = importlib.import_module(PKG + ".layers")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
net = NETS.MultiResUNet(16, device=dev, seed=5)
model = UN.UNetModel(net, 9.0, OPT.Adam(1e-3))
peak, step_no = {}, [0]
orig = LY.Conv2D.__call__


def tracked(self, tape, x, out=None):
    y = orig(self, tape, x, out=out)
    v = float(y.dense().abs().max())
    if v > peak.get(self.name, (0.0, 0))[0]:
        peak[self.name] = (v, step_no[0])
    return y


LY.Conv2D.__call__ = tracked
g = torch.Generator().manual_seed(3)
data = os.environ.get("PROBE_DATA")
if data:
    from PIL import Image
    files = sorted(os.listdir(os.path.join(data, "Input_Images")))[:8]
    imgs = [np.asarray(Image.open(os.path.join(data, "Input_Images", f)).convert("L"), dtype=np.float32) / 255.0 for f in files]
    tiles = [im[r:r + 384, c:c + 384] for im in imgs for r in (0, 328) for c in (0, 320, 640)]
    X = torch.from_numpy(np.stack(tiles))[..., None]
    Y = (X > X.mean() + 0.1).float()
else:
    X = torch.rand((40, 384, 384, 1), generator=g)
    Y = (torch.rand((40, 384, 384, 1), generator=g) > 0.86).float()
gmax = (0.0, 0)
for i in range(steps):
    step_no[0] = i
    sel = torch.randint(0, X.shape[0], (5,), generator=g)
    model.train_step((X[sel].to(dev), Y[sel].to(dev)))
    gm = float(net.arena.grads.abs().max())
    if gm > gmax[0]:
        gmax = (gm, i)
top = sorted(peak.items(), key=lambda kv: -kv[1][0])[:8]
print("largest |conv output| by layer (value, step):", [(k, round(v[0], 1), v[1]) for k, v in top])
print("largest |weight gradient| (unscaled):", gmax, " x 1024 =", gmax[0] * 1024)
