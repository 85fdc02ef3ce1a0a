"""Golden values of skimage.filters.threshold_li (scikit-image 0.18.3), the default threshold of the reference's filter_gan_masks
(HelperFunctions.py:8,163).  Run with the interpreter that has scikit-image:  /opt/conda/bin/python3.9 tests/golden/make_li_goldens.py"""
import os

import numpy as np
from skimage.filters import threshold_li

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(11)
out = {}
for i in range(6):
    h, w = rng.integers(20, 60), rng.integers(20, 60)
    base = rng.normal(40 + 10 * i, 12, (h, w))
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(4):
        cy, cx, r = rng.integers(0, h), rng.integers(0, w), rng.integers(3, 9)
        base[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] += 110
    img8 = np.clip(base, 0, 255).astype(np.uint8)
    out[f"u8_{i}"] = img8
    out[f"u8_{i}_li"] = np.array(threshold_li(img8))
    imgf = (base / 255.0).astype(np.float32)
    out[f"f32_{i}"] = imgf
    out[f"f32_{i}_li"] = np.array(threshold_li(imgf))
np.savez_compressed(os.path.join(HERE, "threshold_li.npz"), **out)
print({k: float(v) for k, v in out.items() if k.endswith("_li")})
