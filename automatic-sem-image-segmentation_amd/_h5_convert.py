"""Stand-alone HDF5 <-> npz converter for the ``model.weights.h5`` member of a Keras-3 ``.keras`` archive.

Runs under any interpreter that has numpy + h5py (in this image: /opt/conda/bin/python3.9; the main interpreter has no h5py) --
keras_io.py calls it as a subprocess when h5py cannot be imported in-process.  Uses nothing but numpy and h5py.

    python _h5_convert.py to_h5  in.npz  out.h5      # npz keys are HDF5 dataset paths ('gen_a/layers/conv2d/vars/0')
    python _h5_convert.py to_npz in.h5   out.npz
"""
import sys

import numpy as np


def to_h5(src, dst):
    import h5py
    z = np.load(src)
    with h5py.File(dst, "w") as f:
        for key in z.files:
            f.create_dataset(key, data=z[key])


def to_npz(src, dst):
    import h5py
    out = {}
    with h5py.File(src, "r") as f:
        def visit(name, obj):
            if isinstance(obj, h5py.Dataset):
                out[name] = np.asarray(obj)
        f.visititems(visit)
    np.savez(dst, **out)


if __name__ == "__main__":
    mode, src, dst = sys.argv[1:4]
    {"to_h5": to_h5, "to_npz": to_npz}[mode](src, dst)
