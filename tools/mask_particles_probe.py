"""Step 2's particle source: time WGAN._sample_particles(3000) (one simulated mask's worth) for several generator call sizes.
Usage: python tools/mask_particles_probe.py"""
import importlib, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
W = importlib.import_module("automatic-sem-image-segmentation_amd.WassersteinGAN")
dev = torch.device("cuda:0")
gen = W.WganGenerator(64, 64, 128, device=dev, seed=1)
wf = W.WGAN.__new__(W.WGAN)
wf.train_images = np.zeros((8, 64, 64, 1), dtype="float32")
wf.batch_size, wf.n_z, wf.device = 64, 128, dev
wf.model = lambda z, training=False: gen(z, training=training)
for chunk in (64, 256, 512, 1024, 3000, 64):
    wf.sample_chunk = chunk
    wf._sample_particles(3000)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        out = wf._sample_particles(3000)
    torch.cuda.synchronize()
    print(f"sample_chunk {chunk}: {(time.perf_counter() - t) / 3 * 1e3:.1f} ms per 3000 particles; out {out.shape} {out.dtype}", flush=True)
