"""Pin the network TOPOLOGIES and the data FEEDERS by executing the reference's own builder code.

Run in the build container only (needs /root/reference; never on the GPU box):

    python tests/golden/make_topology_goldens.py

``CycleGAN.get_resnet_generator`` / ``get_discriminator`` (CycleGAN.py:323-451, incl. the skip-connection, resize-convolution and
sigmoid branches) and ``UNet.multi_res_unet`` (UNet_Segmentation.py:401-562) are plain functional-API Python: they are imported from
/root/reference and executed LITERALLY under a layer-level ``keras`` stand-in whose ``keras.layers.*`` are thin eager callables over
``oracle.ops`` (Keras itself is not installable here).  What this pins: layer order, widths (``int(w*0.167)`` ..., the hard-coded
``32*8`` decoder widths), padding / stride / bias / norm options of every layer, the pre-pad / crop arithmetic, and the order in
which variables are created.  What it cannot pin: the arithmetic INSIDE the Keras layers (oracle/ops.py restates it, SURVEY K-list).

The stand-in is eager: ``Input`` returns the actual batch, every layer call computes immediately and takes its variables from a
queue (creation order), so ``builder(input, weights) -> output`` is a pure function; ``keras.models.Model`` just records the output.

Weights are NOT stored (the UNet has 2.4 M): they are a deterministic function of (seed, index, shape) -- ``golden_weights`` below,
numpy PCG64 streams, stable across numpy versions -- and a float64 checksum is stored to detect drift.  Only inputs, outputs, variable
names / shapes / trainable flags are committed (npz).  Nothing of the reference is copied.

Also here: vectors for the feeders ``CycleGAN.DataLoader`` (CycleGAN.py:454-479) and ``UNet_Segmentation.ImageDataset / DataLoader /
DataSet`` (UNet_Segmentation.py:21-144), produced by the reference's own classes on synthetic files.
"""
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/Releases/Version 1.2.0"
sys.path.insert(0, REPO)


def golden_weights(specs, seed):
    """Deterministic test weights for a list of (name, shape, trainable): kernels ~ U(-a, a) with a Glorot-like bound, gamma ~
    U(0.5, 1.5), beta / bias / moving_mean ~ U(-0.3, 0.3), moving_variance ~ U(0.5, 1.5).  Shared by generator script and tests."""
    out = []
    for i, (name, shape, _tr) in enumerate(specs):
        rng = np.random.default_rng([seed, i])
        kind = name.rsplit("/", 1)[-1]
        if kind == "kernel":
            rf = int(np.prod(shape[:-2]))
            a = float(np.sqrt(6.0 / (shape[-2] * rf + shape[-1] * rf)))
            w = rng.uniform(-a, a, size=shape)
        elif kind in ("gamma", "moving_variance"):
            w = rng.uniform(0.5, 1.5, size=shape)
        else:
            w = rng.uniform(-0.3, 0.3, size=shape)
        out.append(w.astype(np.float32))
    return out


class Stub:
    """State of the eager stand-in: the batch ``Input`` returns, the variable queue, the training flag."""
    input = None
    training = True
    preset = None        # list of torch tensors to hand out in creation order (None: zeros are created, shapes recorded)
    specs = []           # (name, shape, trainable) in creation order
    values = []          # the tensors handed out
    moving_updates = []  # (index, new value) for BN moving statistics

    @classmethod
    def reset(cls, x, training, preset):
        cls.input, cls.training, cls.preset = x, training, preset
        cls.specs, cls.values, cls.moving_updates = [], [], []

    @classmethod
    def var(cls, name, shape, trainable=True):
        i = len(cls.specs)
        cls.specs.append((name, tuple(int(s) for s in shape), trainable))
        v = cls.preset[i] if cls.preset is not None else torch.zeros(shape, dtype=cls.input.dtype)
        assert tuple(v.shape) == tuple(shape), (name, v.shape, shape)
        cls.values.append(v)
        return v


def install_layer_stub():
    from oracle import ops

    class _Auto(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            m = _Auto(self.__name__ + "." + name)
            setattr(self, name, m)
            return m

        def __call__(self, *a, **k):
            return _Auto("call")

    keras = _Auto("keras")
    counters = {}

    def uname(kind):
        counters[kind] = counters.get(kind, -1) + 1
        return f"{kind}_{counters[kind]}" if counters[kind] else kind

    def pair(v):
        return (v, v) if isinstance(v, int) else tuple(v)

    class Layer:
        def __init__(self, **kwargs):
            pass

        def __call__(self, *a, **k):
            return self.call(*a, **k)

    class Conv2D(Layer):
        def __init__(self, filters, kernel_size, strides=(1, 1), padding="valid", use_bias=True, kernel_initializer=None,
                     activation=None, name=None, **kw):
            self.filters, self.k, self.s = filters, pair(kernel_size), pair(strides)
            self.padding, self.use_bias, self.activation = padding, use_bias, activation
            assert self.k[0] == self.k[1] and self.s[0] == self.s[1] and activation is None
            self.name = name or uname("conv2d")

        def call(self, x):
            w = Stub.var(f"{self.name}/kernel", (self.k[0], self.k[1], x.shape[-1], self.filters))
            b = Stub.var(f"{self.name}/bias", (self.filters,)) if self.use_bias else None
            return ops.conv2d(x, w, b, stride=self.s[0], padding=self.padding)

    class Conv2DTranspose(Layer):
        def __init__(self, filters, kernel_size, strides=(1, 1), padding="valid", use_bias=True, kernel_initializer=None, name=None, **kw):
            self.filters, self.k, self.s, self.use_bias = filters, pair(kernel_size), pair(strides), use_bias
            assert padding == "same"
            self.name = name or uname("conv2d_transpose")

        def call(self, x):
            w = Stub.var(f"{self.name}/kernel", (self.k[0], self.k[1], self.filters, x.shape[-1]))
            b = Stub.var(f"{self.name}/bias", (self.filters,)) if self.use_bias else None
            return ops.conv2d_transpose(x, w, b, stride=self.s[0])

    class GroupNormalization(Layer):
        def __init__(self, groups=32, axis=-1, epsilon=1e-3, center=True, scale=True, gamma_initializer="ones", **kw):
            assert groups == -1 and axis == 3
            self.eps, self.center, self.scale = epsilon, center, scale
            self.name = uname("group_normalization")

        def call(self, x, training=None):
            c = x.shape[-1]
            g = Stub.var(f"{self.name}/gamma", (c,)) if self.scale else torch.ones(c, dtype=x.dtype)
            b = Stub.var(f"{self.name}/beta", (c,)) if self.center else torch.zeros(c, dtype=x.dtype)
            return ops.instance_norm(x, g, b, eps=self.eps)

    class BatchNormalization(Layer):
        def __init__(self, axis=-1, momentum=0.99, epsilon=1e-3, center=True, scale=True, **kw):
            assert axis in (3, -1) and center
            self.momentum, self.eps, self.scale = momentum, epsilon, scale
            self.name = uname("batch_normalization")

        def call(self, x, training=None):
            if x.dim() == 2:            # features of a Dense layer: statistics over the batch axis only
                return self.call(x.reshape(x.shape[0], 1, 1, -1)).reshape(x.shape)
            c = x.shape[-1]
            g = Stub.var(f"{self.name}/gamma", (c,)) if self.scale else None
            b = Stub.var(f"{self.name}/beta", (c,))
            i_mm = len(Stub.specs)
            mm = Stub.var(f"{self.name}/moving_mean", (c,), trainable=False)
            mv = Stub.var(f"{self.name}/moving_variance", (c,), trainable=False)
            y, nmm, nmv = ops.batch_norm(x, g, b, mm, mv, Stub.training, momentum=self.momentum, eps=self.eps)
            Stub.moving_updates += [(i_mm, nmm), (i_mm + 1, nmv)]
            return y

    class Activation(Layer):
        def __init__(self, activation, name=None, **kw):
            self.fn = {"relu": torch.relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid,
                       "softmax": lambda x: torch.softmax(x, dim=-1)}[activation]

        def call(self, x):
            return self.fn(x)

    class LeakyReLU(Layer):
        def __init__(self, negative_slope=0.3, **kw):
            self.alpha = negative_slope

        def call(self, x):
            return ops.leaky_relu(x, self.alpha)

    class MaxPooling2D(Layer):
        def __init__(self, pool_size=(2, 2), **kw):
            assert pair(pool_size) == (2, 2)

        def call(self, x):
            return ops.max_pool2x2(x)

    class UpSampling2D(Layer):
        def __init__(self, size=(2, 2), **kw):
            assert pair(size) == (2, 2)

        def call(self, x):
            return x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)

    class Cropping2D(Layer):
        def __init__(self, cropping, **kw):
            self.c = cropping

        def call(self, x):
            (t, b), (l, r) = self.c
            return x[:, t:x.shape[1] - b, l:x.shape[2] - r, :]

    def Input(shape=None, name=None, **kw):
        assert tuple(Stub.input.shape[1:]) == tuple(shape), (Stub.input.shape, shape)
        return Stub.input

    class Model:
        def __init__(self, inputs=None, outputs=None, name=None, **kw):
            self.inputs, self.outputs, self.name = inputs, outputs, name

    def ops_pad(x, pad, mode="constant"):
        assert mode == "reflect" and tuple(pad[0]) == (0, 0) and tuple(pad[3]) == (0, 0)
        (t, b), (l, r) = pad[1], pad[2]
        return torch.nn.functional.pad(x.permute(0, 3, 1, 2), (l, r, t, b), mode="reflect").permute(0, 2, 3, 1)

    keras.layers.Layer = Layer
    for cls in (Conv2D, Conv2DTranspose, GroupNormalization, BatchNormalization, Activation, LeakyReLU, MaxPooling2D,
                UpSampling2D, Cropping2D):
        setattr(keras.layers, cls.__name__, cls)
    keras.layers.Input = Input
    keras.layers.add = lambda xs: xs[0] + xs[1]
    keras.layers.concatenate = lambda xs, axis=-1: torch.cat(list(xs), dim=axis)
    keras.models.Model = Model
    keras.Model = type("KModel", (), {"__init__": lambda self, *a, **k: None})
    keras.ops.pad = ops_pad
    keras.saving.register_keras_serializable = lambda *a, **k: (lambda cls: cls)
    keras.utils.Sequence = type("Sequence", (), {"__init__": lambda self, **k: None})
    keras.callbacks.Callback = type("Callback", (), {})
    keras.losses.MeanAbsoluteError = lambda: (lambda t, p: (t - p).abs().mean())
    keras.losses.MeanSquaredError = lambda: (lambda t, p: ((t - p) ** 2).mean())
    sys.modules["keras"] = keras
    for name in ("cv2", "skimage", "skimage.filters", "skimage.segmentation", "skimage.feature",
                 "skimage.measure", "skimage.morphology", "opensimplex"):
        sys.modules.setdefault(name, _Auto(name))
    os.environ["KERAS_BACKEND"] = "torch"
    sys.path.insert(0, REF)
    return counters


def run_builder(builder, x, seed, counters, want_inference=False):
    """builder() executes reference code on Stub.input.  Pass 1 records the variable specs, pass 2 runs with golden weights."""
    counters.clear()
    Stub.reset(x, True, None)
    builder()
    specs = list(Stub.specs)
    ws = golden_weights(specs, seed)
    preset = [torch.from_numpy(w) for w in ws]
    counters.clear()
    Stub.reset(x, True, preset)
    y = builder()
    out = {"x": x.numpy(), "y_train": y.detach().numpy(),
           "names": np.array([s[0] for s in specs]), "trainable": np.array([s[2] for s in specs]),
           "shapes": np.array([",".join(map(str, s[1])) for s in specs]),
           "seed": np.array(seed), "checksum": np.array(sum(float(np.sum(w.astype(np.float64))) for w in ws))}
    for i, v in Stub.moving_updates:
        out[f"moving_after/{i}"] = v.detach().numpy()
    if want_inference:
        counters.clear()
        Stub.reset(x, False, preset)
        out["y_infer"] = builder().detach().numpy()
    return out


def main():
    counters = install_layer_stub()
    import CycleGAN as RCG          # /root/reference/Releases/Version 1.2.0/CycleGAN.py
    import UNet_Segmentation as RUN  # /root/reference/Releases/Version 1.2.0/UNet_Segmentation.py

    g = torch.Generator().manual_seed(5)
    res = {}

    def workflow(shape, **attrs):
        wf = RCG.CycleGAN.__new__(RCG.CycleGAN)
        wf.image_shape = shape
        wf.kernel_init, wf.gamma_init = None, "ones"
        wf.use_skip_connection = wf.use_resize_convolution = False
        wf.gaussian_noise_value = 0.0
        for k, v in attrs.items():
            setattr(wf, k, v)
        return wf

    # --- generators: StartProcess configuration + each builder branch (CycleGAN.py:360-423) -------------------------------------
    gen_cases = {
        "gen_default_32": dict(shape=(32, 32, 1), n=2, attrs={}, kw={}),
        "gen_prepad_36x44": dict(shape=(36, 44, 1), n=1, attrs={}, kw={}),                 # CycleGAN.py:365-367, output not cropped
        "gen_skip_32": dict(shape=(32, 32, 1), n=1, attrs=dict(use_skip_connection=True), kw={}),
        "gen_resize_32": dict(shape=(32, 32, 1), n=1, attrs=dict(use_resize_convolution=True), kw={}),
        "gen_sigmoid_32": dict(shape=(32, 32, 1), n=1, attrs={}, kw=dict(use_binary_crossentropy=True)),
    }
    for i, (name, c) in enumerate(gen_cases.items()):
        wf = workflow(c["shape"], **c["attrs"])
        x = torch.rand((c["n"],) + c["shape"], generator=g) * 2 - 1

        def build(wf=wf, c=c):
            return wf.get_resnet_generator(name="generator_A", filters=4, num_downsampling_blocks=3, num_residual_blocks=9,
                                           num_upsample_blocks=3, **c["kw"]).outputs
        res[name] = run_builder(build, x, 100 + i, counters)
        print(name, res[name]["y_train"].shape, len(res[name]["names"]), "variables")

    # --- discriminator (CycleGAN.py:425-451), padding='valid' as create_model passes it (CycleGAN.py:148) -------------------------
    for i, (name, shape, nd) in enumerate((("disc_valid_64", (64, 64, 1), 2), ("disc_valid_134x130_nd3", (134, 130, 1), 3))):
        wf = workflow(shape)
        x = torch.rand((2,) + shape, generator=g) * 2 - 1

        def build(wf=wf, nd=nd):
            return wf.get_discriminator(name="discriminator_A", num_downsampling_blocks=nd, filters=8, padding="valid").outputs
        res[name] = run_builder(build, x, 200 + i, counters)
        print(name, res[name]["y_train"].shape, len(res[name]["names"]), "variables")

    # --- MultiResUNet (UNet_Segmentation.py:505-562), real widths (conv_filters = 16) -----------------------------------------------
    for i, (name, shape, n) in enumerate((("unet_32", (32, 32, 1), 2), ("unet_pad_40x36", (40, 36, 1), 1))):
        x = torch.rand((n,) + shape, generator=g)

        def build():
            return RUN.UNet.multi_res_unet(Stub.input, output_channels=1, conv_filters=16)
        res[name] = run_builder(build, x, 300 + i, counters, want_inference=True)
        print(name, res[name]["y_train"].shape, len(res[name]["names"]), "variables")

    # --- multi-class head (UNet_Segmentation.py:558-560): Conv2D(k, 1x1, bias) + softmax instead of conv2d_bn(1, sigmoid) ---------
    x = torch.rand((2, 32, 32, 1), generator=g)

    def build_mc():
        return RUN.UNet.multi_res_unet(Stub.input, output_channels=3, conv_filters=16)
    res["unet_32_softmax3"] = run_builder(build_mc, x, 310, counters, want_inference=True)
    print("unet_32_softmax3", res["unet_32_softmax3"]["y_train"].shape, len(res["unet_32_softmax3"]["names"]), "variables")

    flat = {f"{case}/{k}": v for case, d in res.items() for k, v in d.items()}
    path = os.path.join(HERE, "topology_goldens.npz")
    np.savez_compressed(path, **flat)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")

    # --- feeders ---------------------------------------------------------------------------------------------------------------
    out = {}
    rng = np.random.default_rng(9)
    a = rng.random((11, 4, 4, 1)).astype("float32")
    b = rng.random((7, 4, 4, 1)).astype("float32")
    dl = RCG.DataLoader(a.copy(), b.copy(), batch_size=3)
    out["cg/a"], out["cg/b"], out["cg/len"] = a, b, np.array(len(dl))
    np.random.seed(42)
    for ep in range(2):
        for idx in range(len(dl)):
            xa, xb = dl[idx]
            out[f"cg/ep{ep}/a{idx}"], out[f"cg/ep{ep}/b{idx}"] = xa.copy(), xb.copy()   # views of arrays shuffled in place later
        dl.on_epoch_end()
    # UNet: files on disk (names only matter for the split / ids); images 8x6 uint8
    import tempfile
    from PIL import Image
    with tempfile.TemporaryDirectory() as td:
        idir, mdir = os.path.join(td, "imgs"), os.path.join(td, "masks")
        os.makedirs(idir), os.makedirs(mdir)
        names = [f"img_{k:02d}.tif" for k in range(7)]
        srcs = {}
        for nme in names:
            im = (rng.random((6, 8)) * 255).astype("uint8")
            mk = ((rng.random((6, 8)) > 0.6) * 255).astype("uint8")
            Image.fromarray(im).save(os.path.join(idir, nme))
            Image.fromarray(mk).save(os.path.join(mdir, nme))
            srcs[nme] = (im, mk)
        out["un/names"] = np.array(names)
        out["un/imgs"] = np.stack([srcs[n_][0] for n_ in names])
        out["un/masks"] = np.stack([srcs[n_][1] for n_ in names])
        for subset in ("train", "val"):
            ds = RUN.ImageDataset(idir, mdir)
            ds.initialize_images(subset)
            out[f"un/{subset}/ids"] = np.array(ds.image_ids)
            out[f"un/{subset}/files"] = np.array([os.path.basename(ds.image_info[i_]["image_path"]) for i_ in ds.image_ids])
            ld = RUN.DataLoader(ds, batch_size=3, shuffle=True)
            out[f"un/{subset}/len"] = np.array(len(ld))
            np.random.seed(7)
            for ep in range(2):
                for idx in range(len(ld)):
                    x_, y_ = ld[idx]
                    out[f"un/{subset}/ep{ep}/x{idx}"], out[f"un/{subset}/ep{ep}/y{idx}"] = x_, y_
                ld.on_epoch_end()
    xs = rng.random((10, 4, 4, 1)).astype("float32")
    ys = (rng.random((10, 4, 4, 1)) > 0.5).astype("float32")
    dset = RUN.DataSet(xs.copy(), ys.copy(), batch_size=4, shuffle=True)
    out["ds/x"], out["ds/y"], out["ds/len"] = xs, ys, np.array(len(dset))
    random.seed(3)
    for ep in range(2):
        for idx in range(len(dset)):
            x_, y_ = dset[idx]
            out[f"ds/ep{ep}/x{idx}"], out[f"ds/ep{ep}/y{idx}"] = np.array(x_), np.array(y_)
        dset.on_epoch_end()
    path = os.path.join(HERE, "feeder_goldens.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
