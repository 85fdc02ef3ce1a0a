"""Generate the committed golden vectors by executing the REFERENCE's own control flow.

Run in the build container only (needs /root/reference; never on the GPU box):

    python tests/golden/make_goldens.py

What is pinned:
* ``CycleGanModel.train_step_torch`` (CycleGAN.py:615-710), ``CycleGAN.generator_loss_fn`` /
  ``discriminator_loss_fn`` (CycleGAN.py:301-308) and ``ImagePool`` (CycleGAN.py:908-964) are
  imported from /root/reference and executed literally, under a test-only ``keras`` stub
  (Keras itself is not installed).  The nets handed to the reference step are the oracle's
  torch restatements; the four optimizers are the oracle's Keras-Adam restatement.
* ``HelperFunctions.tile_image`` / ``stitch_image`` / ``load_and_preprocess_images`` are executed
  literally for tiling / normalisation vectors.

Nothing of the reference is copied: only inputs and outputs are stored (npz).
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


def install_keras_stub():
    """~40-line stand-in for the parts of the keras namespace CycleGAN.py touches at import
    time and inside train_step_torch.  Test-only; lives in this container only."""

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

    class Model(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def compile(self, **k):
            pass

    class Mean:
        def __init__(self, name):
            self.name, self.total, self.count = name, 0.0, 0

        def update_state(self, v):
            self.total += float(v)
            self.count += 1

        def result(self):
            return self.total / max(self.count, 1)

    class _Loss:
        def __init__(self, fn):
            self.fn = fn

        def __call__(self, y_true, y_pred):
            return self.fn(y_true, y_pred)

    keras.Model = Model
    keras.metrics.Mean = Mean
    keras.losses.MeanAbsoluteError = lambda: _Loss(lambda t, p: (t - p).abs().mean())
    keras.losses.MeanSquaredError = lambda: _Loss(lambda t, p: ((t - p) ** 2).mean())
    keras.ops.ones_like = torch.ones_like
    keras.ops.zeros_like = torch.zeros_like
    keras.ops.expand_dims = lambda x, axis: torch.unsqueeze(x, axis)
    keras.ops.copy = lambda x: x.clone()
    keras.ops.concatenate = lambda xs, axis=0: torch.cat(list(xs), axis)
    keras.saving.register_keras_serializable = lambda *a, **k: (lambda cls: cls)
    keras.layers.Layer = type("Layer", (), {})
    keras.utils.Sequence = type("Sequence", (), {"__init__": lambda self, **k: None})
    keras.callbacks.Callback = type("Callback", (), {})
    sys.modules["keras"] = keras
    for name in ("cv2", "skimage", "skimage.filters", "skimage.segmentation", "skimage.feature",
                 "skimage.measure", "skimage.morphology", "opensimplex"):
        sys.modules.setdefault(name, _Auto(name))
    os.environ["KERAS_BACKEND"] = "torch"
    sys.path.insert(0, REF)


def gen_cyclegan_step(ref_cg, out_path, n, size, filters, steps, seed):
    from oracle import nets, ops

    dt = torch.float32
    ga = nets.ResnetGenerator(filters=filters, dtype=dt, seed=seed + 1)
    gb = nets.ResnetGenerator(filters=filters, dtype=dt, seed=seed + 2)
    da = nets.PatchDiscriminator(filters=2 * filters, dtype=dt, seed=seed + 3)
    db = nets.PatchDiscriminator(filters=2 * filters, dtype=dt, seed=seed + 4)
    init = {f"init/{nm}/{i}": w for nm, net in (("gen_a", ga), ("gen_b", gb), ("disc_a", da), ("disc_b", db))
            for i, w in enumerate(net.get_weights())}

    # reference objects: workflow instance only for its loss functions (no directory scan)
    wf = ref_cg.CycleGAN.__new__(ref_cg.CycleGAN)
    wf.label_smoothing_factor = 0.0
    import keras
    wf.adv_loss_fn = keras.losses.MeanSquaredError()
    # pools exactly as CycleGAN.__init__ builds them: batch_size frozen at 2 (CycleGAN.py:23,107-108)
    pool_a = ref_cg.ImagePool(batch_size=2, pool_size=3)
    pool_b = ref_cg.ImagePool(batch_size=2, pool_size=3)
    model = ref_cg.CycleGanModel(ga, gb, da, db, image_pool_a=pool_a, image_pool_b=pool_b,
                                 lambda_cycle_a=10, lambda_cycle_b=10, lambda_identity_a=0.5, lambda_identity_b=0.5)
    model.compile(gen_a_optimizer=ops.KerasAdam(2e-4, 0.5), gen_b_optimizer=ops.KerasAdam(2e-4, 0.5),
                  disc_x_optimizer=ops.KerasAdam(2e-4, 0.5), disc_y_optimizer=ops.KerasAdam(2e-4, 0.5),
                  gen_loss_fn=wf.generator_loss_fn, disc_loss_fn=wf.discriminator_loss_fn)

    g = torch.Generator().manual_seed(seed)
    random.seed(seed)
    out = dict(init)
    out["meta"] = np.array([n, size, filters, steps, seed], dtype=np.int64)
    for s in range(steps):
        real_a = torch.rand((n, size, size, 1), generator=g) * 2 - 1
        real_b = (torch.rand((n, size, size, 1), generator=g) > 0.8).float() * 2 - 1
        out[f"step{s}/real_a"] = real_a.numpy()
        out[f"step{s}/real_b"] = real_b.numpy()
        metrics = model.train_step_torch((real_a, real_b))
        out[f"step{s}/metrics"] = np.array([metrics[k] for k in sorted(metrics)], dtype=np.float64)
        out[f"step{s}/metric_names"] = np.array(sorted(metrics))
    for nm, net in (("gen_a", ga), ("gen_b", gb), ("disc_a", da), ("disc_b", db)):
        for i, w in enumerate(net.get_weights()):
            out[f"final/{nm}/{i}"] = w
    np.savez_compressed(out_path, **out)
    print("wrote", out_path, os.path.getsize(out_path) // 1024, "KiB")


def gen_pool_trace(ref_cg, out_path):
    random.seed(0)
    pool = ref_cg.ImagePool(batch_size=2, pool_size=5)
    g = torch.Generator().manual_seed(7)
    ins, outs = [], []
    for step in range(30):
        n = 3 if step % 4 else 2   # 3 exercises the frozen loop bound of 2 (only the first 2 images are used)
        x = torch.rand((n, 2, 2, 1), generator=g)
        y = pool.query(x.clone())
        ins.append(x.numpy())
        outs.append(np.asarray(y.numpy()))
    np.savez_compressed(out_path, n_steps=30, **{f"in{i}": a for i, a in enumerate(ins)},
                        **{f"out{i}": a for i, a in enumerate(outs)})
    print("wrote", out_path)


def gen_helper_vectors(out_path):
    import HelperFunctions as HF  # reference module, cv2/skimage stubbed
    rng = np.random.default_rng(3)
    out = {}
    for name, (h, w, th, tw) in {"a": (70, 100, 32, 32), "b": (64, 64, 32, 32), "c": (50, 90, 48, 32)}.items():
        img = rng.random((h, w, 1)).astype("float32")
        tiles = HF.tile_image(img, tw, th, min_overlap=2)
        out[f"tile_{name}/img"] = img
        out[f"tile_{name}/tiles"] = tiles
        out[f"tile_{name}/shape"] = np.array([h, w, th, tw])
        for mode in (0, 1, 2):
            out[f"tile_{name}/stitched{mode}"] = HF.stitch_image(tiles, w, h, min_overlap=2, manage_overlap_mode=mode)
    # 8 -> 4 connectivity (HelperFunctions.py:144-152), pure numpy in the reference
    for i in range(4):
        m = (rng.random((24, 31)) > (0.35 + 0.1 * i)).astype("uint8") * 255
        out[f"conn_{i}/in"] = m.copy()
        out[f"conn_{i}/out"] = HF.eight_to_four_connected(m.copy())
    # load_and_preprocess_images (HelperFunctions.py:294-329) on synthetic files written to a temp dir
    import tempfile
    from PIL import Image
    with tempfile.TemporaryDirectory() as td:
        arr8 = (rng.random((40, 56)) * 255).astype("uint8")
        arr8[:3, :3] = 255
        Image.fromarray(arr8).save(os.path.join(td, "a.tif"))
        out["load/src"] = arr8
        out["load/m11"] = HF.load_and_preprocess_images(td, normalization_range=(-1, 1))
        out["load/unet"] = HF.load_and_preprocess_images(td, normalization_range=(0, 1), contrast_optimization_range=(0.5, 99.5))
        out["load/mask"] = HF.load_and_preprocess_images(td, normalization_range=(0, 1), threshold_value=0.5)
    np.savez_compressed(out_path, **out)
    print("wrote", out_path)


def main():
    install_keras_stub()
    import CycleGAN as ref_cg  # the reference module, /root/reference/Releases/Version 1.2.0/CycleGAN.py
    gen_cyclegan_step(ref_cg, os.path.join(HERE, "cyclegan_step_n5_s64_f4.npz"), n=5, size=64, filters=4, steps=3, seed=11)
    gen_cyclegan_step(ref_cg, os.path.join(HERE, "cyclegan_step_n2_s64_f4.npz"), n=2, size=64, filters=4, steps=2, seed=23)
    gen_pool_trace(ref_cg, os.path.join(HERE, "image_pool_trace.npz"))
    try:
        gen_helper_vectors(os.path.join(HERE, "helper_tiling.npz"))
    except Exception as e:  # helper vectors are for the "next" rows; do not block the step goldens
        print("helper vectors skipped:", repr(e))


if __name__ == "__main__":
    main()
