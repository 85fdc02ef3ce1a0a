"""``.keras`` checkpoint exchange: the Keras-3 archive layout the reference writes and reads with ``model.save('…/model.keras')`` /
``keras.models.load_model`` (CycleGAN.py:221,228; UNet_Segmentation.py:287,303) -- a zip of ``config.json``, ``metadata.json`` and
``model.weights.h5``.

The weight file follows Keras 3.5's ``H5IOStore`` naming as far as it can be restated without Keras (it is not installable here and
the reference ships no ``.keras`` / ``.h5`` file -- ``.MISSING_LARGE_BLOBS`` -- so this layout is UNPINNED): every layer
that owns variables is a group ``<model path>/layers/<keras layer name>/vars/<i>`` with Keras' automatic layer names (``conv2d``,
``conv2d_1`` …, ``conv2d_transpose``, ``group_normalization``, ``batch_normalization``; one counter per layer class, continuing across
the models of one session in build order) and the layer's variables in Keras order (kernel, bias | gamma, beta | [gamma,] beta,
moving_mean, moving_variance); optimizer state as ``<optimizer attribute>/vars/<i>`` (iterations, then momentum / velocity per variable).
``config.json`` records the builder arguments of THIS framework (class_name + config), not a Keras functional graph: a Keras-built
model of the same architecture can ``load_weights`` the member ``model.weights.h5``; ``keras.models.load_model`` of the whole archive
would need the functional-graph JSON, which is not written.

HDF5 access: ``h5py`` in-process when importable, else the stand-alone converter ``_h5_convert.py`` under an interpreter that has it
(``SS_H5PY_PYTHON``, default /opt/conda/bin/python3.9).  Neither available -> a clear error (plain ``.npz`` remains available by
giving a path that ends in ``.npz``).
"""
import io
import json
import os
import subprocess
import tempfile
import time
import zipfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class KerasIOError(RuntimeError):
    pass


# ---- HDF5 <-> {path: array} ---------------------------------------------------------------------------------------------------
def _h5py_python():
    return os.environ.get("SS_H5PY_PYTHON", "/opt/conda/bin/python3.9")


def _write_h5(arrays, h5_path):
    try:
        import h5py                                  # noqa: F401
        from . import _h5_convert
        with tempfile.TemporaryDirectory() as td:
            npz = os.path.join(td, "w.npz")
            np.savez(npz, **arrays)
            _h5_convert.to_h5(npz, h5_path)
        return
    except ImportError:
        pass
    py = _h5py_python()
    if not os.path.exists(py):
        raise KerasIOError(f"writing model.weights.h5 needs h5py: not importable here and no interpreter at {py} (set SS_H5PY_PYTHON), "
                           f"or save to a path ending in .npz")
    with tempfile.TemporaryDirectory() as td:
        npz = os.path.join(td, "w.npz")
        np.savez(npz, **arrays)
        r = subprocess.run([py, os.path.join(_HERE, "_h5_convert.py"), "to_h5", npz, h5_path], capture_output=True, text=True)
        if r.returncode != 0:
            raise KerasIOError("HDF5 conversion failed: " + r.stderr[-2000:])


def _read_h5(h5_path):
    try:
        import h5py                                  # noqa: F401
        from . import _h5_convert
        with tempfile.TemporaryDirectory() as td:
            npz = os.path.join(td, "w.npz")
            _h5_convert.to_npz(h5_path, npz)
            z = np.load(npz)
            return {k: z[k] for k in z.files}
    except ImportError:
        pass
    py = _h5py_python()
    if not os.path.exists(py):
        raise KerasIOError(f"reading model.weights.h5 needs h5py: not importable here and no interpreter at {py} (set SS_H5PY_PYTHON)")
    with tempfile.TemporaryDirectory() as td:
        npz = os.path.join(td, "w.npz")
        r = subprocess.run([py, os.path.join(_HERE, "_h5_convert.py"), "to_npz", h5_path, npz], capture_output=True, text=True)
        if r.returncode != 0:
            raise KerasIOError("HDF5 conversion failed: " + r.stderr[-2000:])
        z = np.load(npz)
        return {k: z[k] for k in z.files}


# ---- variable names of this framework -> Keras layer groups ---------------------------------------------------------------------
class NameCounters:
    """Keras' per-class automatic layer names: 'conv2d', 'conv2d_1', … (one instance per saved archive = one 'session')."""

    def __init__(self):
        self.n = {}

    def next(self, kind):
        i = self.n.get(kind, 0)
        self.n[kind] = i + 1
        return kind if i == 0 else f"{kind}_{i}"


def layer_groups(net, counters):
    """[(keras layer name, [variable names of `net` in Keras order])] in creation order.  A new Keras layer starts at every kernel
    (Conv2D / Conv2DTranspose, its bias follows), at every gamma, and at a beta that does not follow a gamma (scale=False)."""
    groups, prev = [], None
    transposed = {f"{c.name}/kernel" for c in _convs(net) if c.transposed}
    dense = {f"{c.name}/kernel" for c in _convs(net) if getattr(c, "keras_kind", None) == "dense"}      # keras.layers.Dense as 1x1 conv
    names = [s[0] for s in net.arena.specs]
    kinds = [n.rsplit("/", 1)[-1] for n in names]
    i = 0
    while i < len(names):
        k = kinds[i]
        if k == "kernel":
            g = [names[i]]
            if i + 1 < len(names) and kinds[i + 1] == "bias":
                g.append(names[i + 1])
            groups.append((counters.next("dense" if names[i] in dense else ("conv2d_transpose" if names[i] in transposed else "conv2d")), g))
            i += len(g)
        elif k in ("gamma", "beta"):
            g = [names[i]]
            j = i + 1
            if k == "gamma" and j < len(names) and kinds[j] == "beta":
                g.append(names[j]); j += 1
            while j < len(names) and kinds[j] in ("moving_mean", "moving_variance"):
                g.append(names[j]); j += 1
            batch = any(x.endswith(("moving_mean", "moving_variance")) for x in g)
            groups.append((counters.next("batch_normalization" if batch else "group_normalization"), g))
            i = j
        else:
            raise KerasIOError(f"unexpected variable kind {names[i]}")
        prev = k
    return groups


def _convs(net):
    from .layers import Conv2D
    out, seen = [], set()

    def walk(o):
        if id(o) in seen:
            return
        seen.add(id(o))
        if isinstance(o, Conv2D):
            out.append(o)
        elif isinstance(o, (list, tuple)):
            for v in o:
                walk(v)
        elif hasattr(o, "__dict__") and not isinstance(o, type):
            for k, v in vars(o).items():
                if k not in ("arena", "device"):
                    walk(v)
    walk(net)
    return out


def net_arrays(net, prefix, counters):
    """{hdf5 path: array} of one network under `prefix` ('' or 'gen_a/')."""
    weights = dict(zip([s[0] for s in net.arena.specs], net.get_weights()))
    out = {}
    for lname, vs in layer_groups(net, counters):
        for i, v in enumerate(vs):
            w = weights[v]
            if lname.startswith("dense") and w.ndim == 4:          # Keras stores a Dense kernel as (inputs, units)
                w = w.reshape(w.shape[2], w.shape[3])
            out[f"{prefix}layers/{lname}/vars/{i}"] = w
    return out


def load_net_arrays(net, prefix, counters, arrays):
    order = [s[0] for s in net.arena.specs]
    got = {}
    for lname, vs in layer_groups(net, counters):
        for i, v in enumerate(vs):
            key = f"{prefix}layers/{lname}/vars/{i}"
            if key not in arrays:
                raise KerasIOError(f"{key} missing from model.weights.h5")
            got[v] = arrays[key]
    net.set_weights([got[n] for n in order])


def optimizer_arrays(opt, net, prefix):
    """Keras Adam state of `net`'s trainable variables: vars/0 = iterations, then (momentum, velocity) per variable."""
    if opt is None or net.arena.m is None:
        return {}
    out = {f"{prefix}vars/0": np.asarray(opt.iterations, dtype=np.int64)}
    m, v = net.arena.m.detach().cpu().numpy(), net.arena.v.detach().cpu().numpy()
    i = 1
    for name, shape, trainable, off in net.arena.specs:
        if not trainable:
            continue
        size = int(np.prod(shape))
        out[f"{prefix}vars/{i}"] = m[off:off + size].reshape(shape).copy()
        out[f"{prefix}vars/{i + 1}"] = v[off:off + size].reshape(shape).copy()
        i += 2
    return out


def load_optimizer_arrays(opt, net, prefix, arrays):
    import torch
    if opt is None or f"{prefix}vars/0" not in arrays:
        return False
    opt.iterations = int(arrays[f"{prefix}vars/0"])
    m, v = net.arena.m, net.arena.v
    i = 1
    for name, shape, trainable, off in net.arena.specs:
        if not trainable:
            continue
        size = int(np.prod(shape))
        m[off:off + size].copy_(torch.as_tensor(arrays[f"{prefix}vars/{i}"], dtype=torch.float32).reshape(-1))
        v[off:off + size].copy_(torch.as_tensor(arrays[f"{prefix}vars/{i + 1}"], dtype=torch.float32).reshape(-1))
        i += 2
    return True


# ---- the archive ---------------------------------------------------------------------------------------------------------------
def write_archive(path, arrays, class_name, config):
    """zip{config.json, metadata.json, model.weights.h5} at `path` (written atomically)."""
    with tempfile.TemporaryDirectory() as td:
        h5 = os.path.join(td, "model.weights.h5")
        _write_h5(arrays, h5)
        tmp = path + ".tmp"
        with zipfile.ZipFile(tmp, "w", zipfile.ZIP_STORED) as z:
            z.writestr("config.json", json.dumps({"module": "automatic-sem-image-segmentation_amd", "class_name": class_name,
                                                  "config": config, "registered_name": class_name}, indent=1))
            z.writestr("metadata.json", json.dumps({"keras_version": "3.5.0-layout", "writer": "automatic-sem-image-segmentation_amd",
                                                    "date_saved": time.strftime("%Y-%m-%d@%H:%M:%S")}))
            z.write(h5, "model.weights.h5")
        os.replace(tmp, path)


def read_archive(path):
    """-> (class_name, config, {hdf5 path: array})."""
    with zipfile.ZipFile(path) as z:
        cfg = json.loads(z.read("config.json"))
        with tempfile.TemporaryDirectory() as td:
            z.extract("model.weights.h5", td)
            arrays = _read_h5(os.path.join(td, "model.weights.h5"))
    return cfg.get("class_name"), cfg.get("config", {}), arrays
