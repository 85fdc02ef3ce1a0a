"""Pin the WGAN-GP network topologies by executing the reference's own builders (WassersteinGAN.py:548-681).

Run in the build container only (needs /root/reference; never on the GPU box):

    python tests/golden/make_wgan_goldens.py

``WGAN.get_generator_model`` / ``get_discriminator_model`` are plain functional-API Python: they run LITERALLY under the eager
layer-level ``keras`` stand-in of make_topology_goldens.py, extended here by Dense / Flatten / Reshape / Dropout.  Dropout is
deterministic: pass 1 logs (rate, shape) of every Dropout call, pass 2 multiplies by keep masks drawn from numpy PCG64 streams
(``golden_keep_masks``) -- the masks are a function of (seed, index, shape, rate), only their checksum is stored.  Weights as in
make_topology_goldens.golden_weights.  Writes tests/golden/wgan_topology.npz (inputs, outputs, variable names / shapes, dropout log).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_topology_goldens as M      # noqa: E402


def golden_keep_masks(log, seed):
    """Keep masks (1 = kept) for a Dropout log [(rate, shape), ...]: shared by this script and tests/test_wgan_*.py."""
    out = []
    for i, (rate, shape) in enumerate(log):
        rng = np.random.default_rng([seed, 7000 + i])
        out.append((rng.random(size=tuple(shape)) >= rate).astype(np.float32))
    return out


class DropState:
    log = []          # (rate, shape) per call, in call order
    keep = None       # list of torch keep masks for pass 2


def extend_stub():
    keras = sys.modules["keras"]
    Layer, Stub = keras.layers.Layer, M.Stub
    counters = {}

    def uname(kind):
        counters[kind] = counters.get(kind, -1) + 1
        return f"{kind}_{counters[kind]}" if counters[kind] else kind

    class Dense(Layer):
        def __init__(self, units, use_bias=True, activation=None, **kw):
            assert activation is None
            self.units, self.use_bias, self.name = units, use_bias, uname("dense")

        def call(self, x):
            assert x.dim() == 2
            w = Stub.var(f"{self.name}/kernel", (x.shape[-1], self.units))
            b = Stub.var(f"{self.name}/bias", (self.units,)) if self.use_bias else None
            y = x @ w
            return y + b if b is not None else y

    class Flatten(Layer):
        def call(self, x):
            return x.reshape(x.shape[0], -1)          # channels-last row-major, like keras.layers.Flatten

    class Reshape(Layer):
        def __init__(self, target_shape, **kw):
            self.t = tuple(target_shape)

        def call(self, x):
            return x.reshape((x.shape[0],) + self.t)

    class Dropout(Layer):
        def __init__(self, rate, **kw):
            self.rate = rate

        def call(self, x, training=None):
            if not Stub.training:
                return x
            i = len(DropState.log)
            DropState.log.append((float(self.rate), tuple(x.shape)))
            if DropState.keep is None:
                return x
            return x * DropState.keep[i] * (1.0 / (1.0 - self.rate))

    for cls in (Dense, Flatten, Reshape, Dropout):
        setattr(keras.layers, cls.__name__, cls)
    return counters


def run(builder, x, seed, counters_list, want_inference=True):
    def reset():
        for c in counters_list:
            c.clear()
        DropState.log = []
    reset()
    DropState.keep = None
    M.Stub.reset(x, True, None)
    builder()
    specs, log = list(M.Stub.specs), list(DropState.log)
    ws = M.golden_weights(specs, seed)
    keep = golden_keep_masks(log, seed)
    preset = [torch.from_numpy(w) for w in ws]
    reset()
    DropState.keep = [torch.from_numpy(k) for k in keep]
    M.Stub.reset(x, True, preset)
    y = builder()
    out = {"x": x.numpy(), "y_train": y.detach().numpy(), "names": np.array([s[0] for s in specs]),
           "trainable": np.array([s[2] for s in specs]), "shapes": np.array([",".join(map(str, s[1])) for s in specs]),
           "seed": np.array(seed), "checksum": np.array(sum(float(np.sum(w.astype(np.float64))) for w in ws)),
           "drop_rates": np.array([r for r, _ in log]), "drop_shapes": np.array([",".join(map(str, s)) for _, s in log]),
           "drop_checksum": np.array(sum(float(k.sum()) for k in keep))}
    for i, v in M.Stub.moving_updates:
        out[f"moving_after/{i}"] = v.detach().numpy()
    if want_inference:
        reset()
        M.Stub.reset(x, False, preset)
        out["y_infer"] = builder().detach().numpy()
    return out


def main():
    c1 = M.install_layer_stub()
    c2 = extend_stub()
    import importlib
    for name in ("matplotlib", "matplotlib.pyplot", "tqdm", "tqdm.autonotebook", "opensimplex", "cv2"):
        try:
            importlib.import_module(name)
        except Exception:
            sys.modules[name] = types.ModuleType(name)
            sys.modules[name].tqdm = lambda it=None, **k: it
    import WassersteinGAN as RW          # /root/reference/Releases/Version 1.2.0/WassersteinGAN.py

    g = torch.Generator().manual_seed(11)
    res = {}
    for i, (h, w) in enumerate(((64, 64), (32, 48))):
        wf = RW.WGAN.__new__(RW.WGAN)
        wf.train_images = np.zeros((4, h, w, 1), dtype="float32")
        wf.n_z = 16
        z = torch.randn((6, wf.n_z), generator=g)
        res[f"gen_{h}x{w}"] = run(lambda wf=wf: wf.get_generator_model().outputs, z, 500 + i, (c1, c2))
        x = torch.rand((3, h, w, 1), generator=g) * 2 - 1
        res[f"critic_{h}x{w}"] = run(lambda wf=wf: wf.get_discriminator_model().outputs, x, 600 + i, (c1, c2))
        for k in (f"gen_{h}x{w}", f"critic_{h}x{w}"):
            print(k, res[k]["y_train"].shape, len(res[k]["names"]), "variables", list(res[k]["drop_rates"]))
    flat = {f"{case}/{k}": v for case, d in res.items() for k, v in d.items()}

    # --- host side: the training set WGAN.__init__ builds (WassersteinGAN.py:334-361) and step 0 (HelperFunctions.py:241-287) ------
    import random
    import tempfile
    from PIL import Image
    import HelperFunctions as RH
    rng = np.random.default_rng(21)
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(os.path.join(td, "Input_Masks"))
        for k, (h, w) in enumerate(((20, 28), (20, 28), (20, 28))):          # equal sizes: the reference stacks them with np.array
            m = (rng.random((h, w)) > 0.5).astype("uint8") * 255
            m[0, 0], m[-1, -1] = 0, 255
            Image.fromarray(m).save(os.path.join(td, "Input_Masks", f"mask_{k}.tif"))
            flat[f"trainset/mask_{k}"] = m
        wf = RW.WGAN(root_dir=td)
        flat["trainset/train_images"] = wf.train_images
        print("train_images", wf.train_images.shape)
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "Input_Images")
        os.makedirs(src)
        for sub in ("trainA", "testA"):
            os.makedirs(os.path.join(td, "2_CycleGAN", "data", sub))
        for k, (h, w) in enumerate(((70, 90), (70, 90), (70, 90))):            # equal sizes (np.array stacking in the loader)
            im = (rng.random((h, w)) * 255).astype("uint8")
            im[: h // 2] //= 4                                   # a dark half: some tiles fail the background filter
            Image.fromarray(im).save(os.path.join(src, f"sem_{k}.tif"))
            flat[f"step0/sem_{k}"] = im
        random.seed(5)
        RH.prepare_images_cycle_gan(td, src, tile_size_w=32, tile_size_h=32, num_simulated_masks=30, dark_background=True)
        for sub in ("trainA", "testA"):
            names = sorted(os.listdir(os.path.join(td, "2_CycleGAN", "data", sub)))
            flat[f"step0/{sub}/names"] = np.array(names)
            flat[f"step0/{sub}/sums"] = np.array([int(np.asarray(Image.open(os.path.join(td, "2_CycleGAN", "data", sub, n_)), dtype=np.int64).sum())
                                                  for n_ in names])
        print("step0", len(flat["step0/trainA/names"]), "tiles,", len(flat["step0/testA/names"]), "test tiles")

    path = os.path.join(HERE, "wgan_topology.npz")
    np.savez_compressed(path, **flat)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
