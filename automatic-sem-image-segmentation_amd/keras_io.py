"""``.keras`` checkpoint exchange: the Keras-3 archive layout the reference writes and reads with ``model.save('…/model.keras')`` /
``keras.models.load_model`` (CycleGAN.py:221,228; UNet_Segmentation.py:287,303) -- a zip of ``config.json``, ``metadata.json`` and
``model.weights.h5``.

The weight file follows Keras 3.5's ``H5IOStore`` naming as far as it can be restated without Keras (it is not installable here and
the reference ships no ``.keras`` / ``.h5`` file -- ``.MISSING_LARGE_BLOBS`` -- so this layout is UNPINNED): every layer
that owns variables is a group ``<model path>/layers/<keras layer name>/vars/<i>`` with Keras' automatic layer names (``conv2d``,
``conv2d_1`` …, ``conv2d_transpose``, ``group_normalization``, ``batch_normalization``; one counter per layer class, RESTARTED for
every container -- Keras' ``_save_container_state`` names the layers of each saved model from a fresh set of snake-case class
counters) and the layer's variables in Keras order (kernel, bias | gamma, beta | [gamma,] beta, moving_mean, moving_variance);
optimizer state as ``<optimizer attribute>/vars/<i>`` in the order Keras' ``BaseOptimizer`` tracks its variables: 0 = iterations,
1 = learning_rate, then Adam's momentums (one per trainable variable), then its velocities.  Archives of earlier versions of this
file (one counter shared by all networks; iterations followed by interleaved momentum / velocity pairs) still load.
``config.json`` records the builder arguments of THIS framework (class_name + config), not a Keras functional graph: a Keras-built
model of the same architecture can ``load_weights`` the member ``model.weights.h5``; ``keras.models.load_model`` of the whole archive
would need the functional-graph JSON, which is not written.

HDF5 access: ``h5py`` in-process when importable, else the stand-alone converter ``_h5_convert.py`` under an interpreter that has it
(``SS_H5PY_PYTHON``, default /opt/conda/bin/python3.9).  Neither available: ``hdf5_available()`` is False (the workflows check it
BEFORE training starts and warn), ``write_archive`` then stores the same arrays as ``<path>.npz`` next to where the archive would
have been instead of losing a finished training, and ``read_archive`` falls back to that file.
"""
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


def hdf5_available():
    """True when model.weights.h5 can be written / read here: h5py in this interpreter, or an interpreter at SS_H5PY_PYTHON that has it."""
    try:
        import h5py                                  # noqa: F401
        return True
    except ImportError:
        return os.path.exists(_h5py_python())


def warn_if_no_hdf5(who):
    """Called by the workflows BEFORE a training starts: without HDF5 access the final `model.keras` will be written as
    `model.keras.npz` (same arrays, loadable by the same load functions) -- better known now than after the last epoch."""
    if not hdf5_available():
        import warnings
        warnings.warn(f"{who}: no h5py here and no interpreter at {_h5py_python()} (SS_H5PY_PYTHON): '.keras' archives will be "
                      f"written as '<path>.npz' (arrays + config), which the load functions of this package read back")


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


def net_arrays(net, prefix, counters=None):
    """{hdf5 path: array} of one network (= one Keras container) under `prefix` ('' or 'gen_a/').  Layer names come from a FRESH set
    of per-class counters: Keras restarts them for every container it saves (`counters` is only for reading old archives)."""
    weights = dict(zip([s[0] for s in net.arena.specs], net.get_weights()))
    out = {}
    for lname, vs in layer_groups(net, counters if counters is not None else NameCounters()):
        for i, v in enumerate(vs):
            w = weights[v]
            if lname.startswith("dense") and w.ndim == 4:          # Keras stores a Dense kernel as (inputs, units)
                w = w.reshape(w.shape[2], w.shape[3])
            out[f"{prefix}layers/{lname}/vars/{i}"] = w
    return out


def load_net_arrays(net, prefix, arrays, legacy_counters=None):
    """Inverse of net_arrays.  `legacy_counters`: a NameCounters shared by the networks of one archive, advanced in the order the
    first version of this file saved them -- tried when the per-container names do not resolve (old archives)."""
    order = [s[0] for s in net.arena.specs]
    legacy = layer_groups(net, legacy_counters) if legacy_counters is not None else None      # always advance the shared counters
    for groups in (layer_groups(net, NameCounters()), legacy):
        if groups is None:
            continue
        keys = [(v, f"{prefix}layers/{lname}/vars/{i}") for lname, vs in groups for i, v in enumerate(vs)]
        if all(k in arrays for _, k in keys):
            got = {v: arrays[k] for v, k in keys}
            net.set_weights([got[n] for n in order])
            return
    missing = [k for _, k in keys if k not in arrays]
    raise KerasIOError(f"{missing[0]} (and {len(missing) - 1} more) missing from model.weights.h5")


def optimizer_arrays(opt, net, prefix):
    """Keras Adam state of `net`'s trainable variables in the order ``keras.optimizers.Adam`` tracks it: vars/0 = iterations,
    vars/1 = learning_rate, then one momentum per trainable variable, then one velocity per trainable variable."""
    if opt is None or net.arena.m is None:
        return {}
    out = {f"{prefix}vars/0": np.asarray(opt.iterations, dtype=np.int64),
           f"{prefix}vars/1": np.asarray(float(opt.learning_rate), dtype=np.float32)}
    m, v = net.arena.m.detach().cpu().numpy(), net.arena.v.detach().cpu().numpy()
    train = [(shape, off) for _, shape, trainable, off in net.arena.specs if trainable]
    n = len(train)
    for j, (shape, off) in enumerate(train):
        size = int(np.prod(shape))
        out[f"{prefix}vars/{2 + j}"] = m[off:off + size].reshape(shape).copy()
        out[f"{prefix}vars/{2 + n + j}"] = v[off:off + size].reshape(shape).copy()
    return out


def load_optimizer_arrays(opt, net, prefix, arrays):
    """Restores iterations, learning rate and the Adam slots.  Accepts the Keras order written by ``optimizer_arrays`` and the layout
    of this file's first version (no learning_rate entry, momentum / velocity interleaved) -- told apart by the entry count."""
    import torch
    if opt is None or f"{prefix}vars/0" not in arrays:
        return False
    opt.iterations = int(arrays[f"{prefix}vars/0"])
    m, v = net.arena.m, net.arena.v
    train = [(shape, off) for _, shape, trainable, off in net.arena.specs if trainable]
    n = len(train)
    count = sum(1 for k in arrays if k.startswith(f"{prefix}vars/"))
    if count == 2 + 2 * n:
        opt.learning_rate = float(arrays[f"{prefix}vars/1"])
        slots = [(2 + j, 2 + n + j) for j in range(n)]
    elif count == 1 + 2 * n:
        slots = [(1 + 2 * j, 2 + 2 * j) for j in range(n)]
    else:
        raise KerasIOError(f"{prefix}: {count} optimizer entries do not fit {n} trainable variables")
    for (shape, off), (im, iv) in zip(train, slots):
        size = int(np.prod(shape))
        m[off:off + size].copy_(torch.as_tensor(arrays[f"{prefix}vars/{im}"], dtype=torch.float32).reshape(-1))
        v[off:off + size].copy_(torch.as_tensor(arrays[f"{prefix}vars/{iv}"], dtype=torch.float32).reshape(-1))
    return True


# ---- the archive ---------------------------------------------------------------------------------------------------------------
def write_archive(path, arrays, class_name, config):
    """zip{config.json, metadata.json, model.weights.h5} at `path` (written atomically).  Without HDF5 access (no h5py here, no
    interpreter that has it) the SAME arrays + config go to `path + '.npz'` with a warning: a save at the end of a long training
    must not be what loses it; read_archive finds that file."""
    with tempfile.TemporaryDirectory() as td:
        h5 = os.path.join(td, "model.weights.h5")
        try:
            _write_h5(arrays, h5)
        except KerasIOError as e:
            import warnings
            fb = path + ".npz"
            np.savez(fb, __class_name__=np.array(class_name), __config__=np.array(json.dumps(config)), **arrays)
            warnings.warn(f"{e}; wrote the archive's arrays to {fb} instead")
            return fb
        tmp = path + ".tmp"
        with zipfile.ZipFile(tmp, "w", zipfile.ZIP_STORED) as z:
            z.writestr("config.json", json.dumps({"module": "automatic-sem-image-segmentation_amd", "class_name": class_name,
                                                  "config": config, "registered_name": class_name}, indent=1))
            z.writestr("metadata.json", json.dumps({"keras_version": "3.5.0-layout", "writer": "automatic-sem-image-segmentation_amd",
                                                    "date_saved": time.strftime("%Y-%m-%d@%H:%M:%S")}))
            z.write(h5, "model.weights.h5")
        os.replace(tmp, path)
    return path


def read_archive(path):
    """-> (class_name, config, {hdf5 path: array}).  Falls back to `path + '.npz'` (write_archive's fallback) when `path` is absent."""
    if not os.path.exists(path) and os.path.exists(path + ".npz"):
        z = np.load(path + ".npz")
        arrays = {k: z[k] for k in z.files if not k.startswith("__")}
        return str(z["__class_name__"]), json.loads(str(z["__config__"])), arrays
    with zipfile.ZipFile(path) as z:
        cfg = json.loads(z.read("config.json"))
        with tempfile.TemporaryDirectory() as td:
            z.extract("model.weights.h5", td)
            arrays = _read_h5(os.path.join(td, "model.weights.h5"))
    return cfg.get("class_name"), cfg.get("config", {}), arrays
