"""Golden vectors for the post-processing row: threshold -> EDT -> gaussian -> peak_local_max ->
marker watershed with lines -> 8-to-4 connectivity.

Run with the interpreter that HAS scikit-image (0.18.3) and scipy (1.7.1) in this container:

    /opt/conda/bin/python3.9 tests/golden/make_postproc_goldens.py

The reference's own functions are imported from /root/reference and executed literally (``Measurements.Measure.segment``
Measurements.py:263-305 and ``HelperFunctions.segment`` HelperFunctions.py:155-160) with an empty ``cv2`` stub (cv2 is not
installed; neither function touches it).  Outputs: tests/golden/postproc_segment.npz -- inputs and expected outputs only.
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference/Releases/Version 1.2.0"
HERE = os.path.dirname(os.path.abspath(__file__))


class _Auto(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Auto(self.__name__ + "." + name)

    def __call__(self, *a, **k):
        return None


def synth(rng, h, w, n_blobs, rmin, rmax, noise):
    """SEM-like probability map (uint8): overlapping bright ellipses on a dark background, blurred, plus noise."""
    from scipy import ndimage
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.zeros((h, w), np.float64)
    for _ in range(n_blobs):
        cy, cx = rng.uniform(0, h), rng.uniform(0, w)
        a, b = rng.uniform(rmin, rmax), rng.uniform(rmin, rmax)
        th = rng.uniform(0, np.pi)
        u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
        v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
        img = np.maximum(img, ((u / a) ** 2 + (v / b) ** 2 <= 1.0) * rng.uniform(0.6, 1.0))
    img = ndimage.gaussian_filter(img, 1.5) + rng.normal(0, noise, img.shape)
    return np.clip(img * 255.0, 0, 255).astype(np.uint8)


def main():
    sys.modules.setdefault("cv2", _Auto("cv2"))
    sys.modules.setdefault("tqdm", types.SimpleNamespace(tqdm=lambda x, **k: x))
    sys.path.insert(0, REF)
    import Measurements as ME      # reference module
    import HelperFunctions as HF   # reference module
    from skimage.feature import peak_local_max
    from skimage.filters import threshold_otsu
    from scipy import ndimage

    rng = np.random.default_rng(20240917)
    out = {}
    cases = [(96, 128, 14, 8, 18, 0.03, 9), (160, 160, 30, 6, 16, 0.05, 9), (128, 200, 10, 14, 30, 0.02, 9),
             (200, 150, 40, 5, 12, 0.04, 5), (64, 64, 3, 10, 20, 0.02, 9), (256, 256, 60, 7, 20, 0.04, 9)]
    for i, (h, w, nb, r0, r1, noise, md) in enumerate(cases):
        img = synth(rng, h, w, nb, r0, r1, noise)
        out[f"c{i}_image"] = img
        out[f"c{i}_min_distance"] = np.int64(md)
        out[f"c{i}_otsu"] = np.float64(threshold_otsu(img))
        # intermediates of Measure.segment restated with the same library calls, for localising a mismatch
        mask = img > threshold_otsu(img)
        dist = ndimage.gaussian_filter(ndimage.distance_transform_edt(mask), sigma=1)
        out[f"c{i}_distance"] = dist
        out[f"c{i}_peaks"] = peak_local_max(dist, min_distance=md).astype(np.int64)
        # the reference's own functions
        out[f"c{i}_seg_ws"] = ME.Measure.segment(img, -1, True, md, darkBackground=True)
        out[f"c{i}_seg_nows"] = ME.Measure.segment(img, -1, False, md, darkBackground=True)
        out[f"c{i}_seg_fixed_thr"] = ME.Measure.segment(img, 100, True, md, darkBackground=True)
        out[f"c{i}_hf_segment"] = HF.segment(img, -1, True, md, use_four_connectivity=True)
    # degenerate inputs (Measurements.py:286-287): constant mask returns before the watershed
    flat = np.full((32, 32), 7, np.uint8)
    out["flat_image"] = flat
    out["flat_seg"] = ME.Measure.segment(flat, 3, True, 9, darkBackground=True)
    np.savez_compressed(os.path.join(HERE, "postproc_segment.npz"), **out)
    print("wrote", os.path.join(HERE, "postproc_segment.npz"), {k: getattr(v, "shape", None) for k, v in out.items() if "seg_ws" in k})


if __name__ == "__main__":
    main()
