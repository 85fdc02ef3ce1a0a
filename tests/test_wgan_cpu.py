"""WGAN-GP (row f4): the oracle's generator / critic (oracle/wgan.py) held to vectors produced by the REFERENCE's own
``get_generator_model`` / ``get_discriminator_model`` (WassersteinGAN.py:548-681) under the layer-level keras stand-in
(tests/golden/make_wgan_goldens.py), plus host-side pieces of the workflow (training-set construction, step-0 tiling)."""
import importlib
import importlib.util
import os

import numpy as np
import pytest
import torch

from oracle import wgan as OW

HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(HERE, "golden", name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


golden_weights = _load("make_topology_goldens").golden_weights


def golden_keep_masks(log, seed):            # = make_wgan_goldens.golden_keep_masks (that module needs /root/reference to import)
    out = []
    for i, (rate, shape) in enumerate(log):
        rng = np.random.default_rng([seed, 7000 + i])
        out.append((rng.random(size=tuple(shape)) >= rate).astype(np.float32))
    return out


@pytest.fixture(scope="module")
def topo(golden_dir):
    return np.load(os.path.join(golden_dir, "wgan_topology.npz"))


def case_specs(z, case):
    names = [str(s) for s in z[f"{case}/names"]]
    shapes = [tuple(int(v) for v in str(s).split(",")) for s in z[f"{case}/shapes"]]
    trainable = [bool(t) for t in z[f"{case}/trainable"]]
    return list(zip(names, shapes, trainable))


def case_weights(z, case):
    specs = case_specs(z, case)
    ws = golden_weights(specs, int(z[f"{case}/seed"]))
    assert abs(sum(float(np.sum(w.astype(np.float64))) for w in ws) - float(z[f"{case}/checksum"])) < 1e-6, "weight generator drifted"
    return specs, ws


def case_keep(z, case):
    log = [(float(r), tuple(int(v) for v in str(s).split(","))) for r, s in zip(z[f"{case}/drop_rates"], z[f"{case}/drop_shapes"])]
    keep = golden_keep_masks(log, int(z[f"{case}/seed"]))
    assert abs(sum(float(k.sum()) for k in keep) - float(z[f"{case}/drop_checksum"])) < 1e-6, "mask generator drifted"
    return log, keep


@pytest.mark.parametrize("hw", [(64, 64), (32, 48)])
def test_oracle_generator_matches_reference_builder(topo, hw):
    case = f"gen_{hw[0]}x{hw[1]}"
    specs, ws = case_weights(topo, case)
    net = OW.WganGenerator(hw[0], hw[1], n_z=16)
    assert [tuple(v.shape) for v in net.variables] == [s[1] for s in specs]
    assert [v.trainable for v in net.variables] == [s[2] for s in specs]
    assert [v.name.rsplit("/", 1)[-1] for v in net.variables] == [s[0].rsplit("/", 1)[-1] for s in specs]
    net.set_weights(ws)
    z = torch.from_numpy(topo[f"{case}/x"])
    with torch.no_grad():
        y_inf = net(z, False)
        y = net(z, True)
    np.testing.assert_allclose(y.numpy(), topo[f"{case}/y_train"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(y_inf.numpy(), topo[f"{case}/y_infer"], rtol=1e-4, atol=2e-5)
    for i, v in enumerate(net.variables):
        if not v.trainable:
            np.testing.assert_allclose(v.value.numpy(), topo[f"{case}/moving_after/{i}"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("hw", [(64, 64), (32, 48)])
def test_oracle_critic_matches_reference_builder(topo, hw):
    case = f"critic_{hw[0]}x{hw[1]}"
    specs, ws = case_weights(topo, case)
    log, keep = case_keep(topo, case)
    assert [r for r, _ in log] == [OW.DROP_CONV, OW.DROP_CONV, OW.DROP_FLAT]
    net = OW.WganCritic(hw[0], hw[1])
    assert [tuple(v.shape) for v in net.variables] == [s[1] for s in specs]
    assert [v.name.rsplit("/", 1)[-1] for v in net.variables] == [s[0].rsplit("/", 1)[-1] for s in specs]
    net.set_weights(ws)
    x = torch.from_numpy(topo[f"{case}/x"])
    masks = dict(zip(("drop1", "drop2", "flat"), (torch.from_numpy(k) for k in keep)))
    with torch.no_grad():
        y = net(x, True, masks)
        y_inf = net(x, False)
    np.testing.assert_allclose(y.numpy(), topo[f"{case}/y_train"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(y_inf.numpy(), topo[f"{case}/y_infer"], rtol=1e-4, atol=2e-5)


def test_gradient_penalty_double_backward_by_finite_differences():
    """The analytic route the HIP path takes (the critic is piecewise linear in its input: d gp / d W through the backward chain
    only) equals autograd's create_graph route -- checked here on the oracle in float64 against central differences of gp(W)."""
    torch.manual_seed(0)
    d = OW.WganCritic(16, 16, dtype=torch.float64, seed=3)
    g = OW.WganGenerator(16, 16, n_z=4, dtype=torch.float64, seed=4)
    step = OW.WganStep(g, d)
    real = torch.rand(2, 16, 16, 1, dtype=torch.float64) * 2 - 1
    fake = (torch.rand(2, 16, 16, 1, dtype=torch.float64) * 2 - 1).requires_grad_(True)      # as a generator output is
    alpha = torch.randn(2, 1, 1, 1, dtype=torch.float64)
    gp, _ = step.gradient_penalty(real, fake, alpha, None)
    w = d.p("conv1/kernel")
    (gw,) = torch.autograd.grad(gp, w)
    idx = (2, 2, 5, 7)
    eps = 1e-6
    with torch.no_grad():
        w[idx] += eps
    gp_p, _ = step.gradient_penalty(real, fake, alpha, None)
    with torch.no_grad():
        w[idx] -= 2 * eps
    gp_m, _ = step.gradient_penalty(real, fake, alpha, None)
    fd = float((gp_p - gp_m) / (2 * eps))
    assert abs(fd - float(gw[idx])) < 1e-6 * max(1.0, abs(fd))


# ---- host side of the workflow -----------------------------------------------------------------------------------------------
BASE = "automatic-sem-image-segmentation_amd"


def test_step0_tiling_writes_the_reference_files(topo, tmp_path):
    """prepare_images_cycle_gan (HelperFunctions.py:241-287) on the synthetic SEM images of the golden file, python's ``random``
    seeded like the generator script: same tile files, same pixel sums, same five test tiles."""
    import random
    from PIL import Image
    H = importlib.import_module(BASE + ".HelperFunctions")
    src = tmp_path / "Input_Images"
    src.mkdir()
    for sub in ("trainA", "testA"):
        (tmp_path / "2_CycleGAN" / "data" / sub).mkdir(parents=True)
    for k in range(3):
        Image.fromarray(topo[f"step0/sem_{k}"]).save(src / f"sem_{k}.tif")
    random.seed(5)
    H.prepare_images_cycle_gan(str(tmp_path), str(src), tile_size_w=32, tile_size_h=32, num_simulated_masks=30, dark_background=True)
    for sub in ("trainA", "testA"):
        d = tmp_path / "2_CycleGAN" / "data" / sub
        names = sorted(os.listdir(d))
        assert names == [str(n) for n in topo[f"step0/{sub}/names"]]
        sums = [int(np.asarray(Image.open(d / n), dtype=np.int64).sum()) for n in names]
        assert sums == [int(v) for v in topo[f"step0/{sub}/sums"]]


def test_training_set_construction_matches_reference(topo, tmp_path, monkeypatch):
    """WGAN.__init__ (WassersteinGAN.py:334-361): threshold at 0.5, [-1, 1], the four flips, zero padding to multiples of 16
    (with the reference's width-from-height slip)."""
    from PIL import Image
    W = importlib.import_module(BASE + ".WassersteinGAN")
    (tmp_path / "Input_Masks").mkdir()
    for k in range(3):
        Image.fromarray(topo[f"trainset/mask_{k}"]).save(tmp_path / "Input_Masks" / f"mask_{k}.tif")
    monkeypatch.setattr(W.D, "local_device", lambda: torch.device("cpu"))
    wf = W.WGAN(root_dir=str(tmp_path))
    assert wf.train_images.dtype == np.float32
    np.testing.assert_array_equal(wf.train_images, topo["trainset/train_images"])
    assert (wf.batch_size, wf.epochs, wf.n_z) == (64, 1000, 128)


def test_rotation_matrix_and_warp_follow_the_opencv_definitions():
    W = importlib.import_module(BASE + ".WassersteinGAN")
    m = W._rotation_matrix_2d((10.0, 6.0), 90.0, 1.0)
    np.testing.assert_allclose(m @ np.array([10.0, 6.0, 1.0]), [10.0, 6.0], atol=1e-12)          # the centre is fixed
    # +90 degrees is counter-clockwise in image coordinates (y down): the point right of the centre moves up
    np.testing.assert_allclose(m @ np.array([12.0, 6.0, 1.0]), [10.0, 4.0], atol=1e-12)
    m2 = W._rotation_matrix_2d((0.0, 0.0), 0.0, 2.0)
    src = np.zeros((8, 8), np.uint8)
    src[2:4, 1:3] = 200
    out = W._warp_affine(src, m2, (16, 16))
    assert out.shape == (16, 16) and out[4:7, 2:5].min() == 200 and out[12:, :].max() == 0          # scaled by two about the origin
    ident = W._warp_affine(src, np.array([[1.0, 0, 3], [0, 1.0, 1]]), (12, 10))
    assert ident.shape == (10, 12) and np.array_equal(ident[1:9, 3:11], src)                       # pure translation, (width, height) order


def test_gradient_noise_is_smooth_and_spans_its_range():
    W = importlib.import_module(BASE + ".WassersteinGAN")
    n = W._gradient_noise2array(np.arange(0, 4, 4 / 200), np.arange(0, 4, 4 / 240), np.random.default_rng(0))
    assert n.shape == (200, 240) and np.isfinite(n).all()
    assert n.max() - n.min() > 0.8 and np.abs(np.diff(n, axis=0)).max() < 0.1 and np.abs(np.diff(n, axis=1)).max() < 0.1


def _stub_particles(count):
    import random
    yy, xx = np.mgrid[0:32, 0:32]
    out = np.zeros((count, 32, 32), np.uint8)
    for k in range(count):
        r = random.randint(6, 11)
        out[k] = (((yy - 16) ** 2 + (xx - 16) ** 2) < r * r) * 255
    return out


def test_mask_simulation_in_worker_processes_writes_the_sequential_masks(tmp_path, monkeypatch):
    """WGAN.simulate_masks draws where / how large / how rotated in the main process (the reference's order) and lets worker processes put
    the particles onto the canvases -- the placement consumes no random numbers, so the files must be the ones the inline loop writes."""
    import random
    from PIL import Image
    W = importlib.import_module(f"{BASE}.WassersteinGAN")
    outs = {}
    for workers in (1, 2):
        monkeypatch.setenv("SS_MASK_WORKERS", str(workers))
        wf = W.WGAN.__new__(W.WGAN)
        wf.train_images = np.zeros((8, 32, 32, 1), dtype="float32")
        wf.batch_size, wf.n_z, wf.model = 64, 16, object()          # the generator is stubbed: particles are seeded discs
        wf.generate_dir = str(tmp_path / f"trainB_{workers}")
        wf._sample_particles = _stub_particles
        random.seed(3)
        np.random.seed(3)
        wf.simulate_masks(no_of_images=4, use_perlin_noise=True, use_normal_distribution=True, max_overlap=0.5, img_width=96, img_height=80)
        outs[workers] = [np.asarray(Image.open(os.path.join(wf.generate_dir, f"{i:05d}.tif"))) for i in range(4)]
    for a, b in zip(outs[1], outs[2]):
        assert a.shape == (80, 96) and set(np.unique(a)) <= {0, 255} and a.any()
        np.testing.assert_array_equal(a, b)


def test_particle_cleaning_opens_with_the_9x9_square():
    """_MaskCanvas.place cleans a warped particle by hole filling and a 9 x 9 opening (WassersteinGAN.py:523-524), evaluated as row and
    column passes: the pixels of scipy's binary_opening with the full square, also for shapes that touch the border."""
    from scipy import ndimage
    W = importlib.import_module(f"{BASE}.WassersteinGAN")
    rng = np.random.default_rng(0)
    for _ in range(60):
        h, w = rng.integers(10, 90, 2)
        x = ndimage.gaussian_filter(rng.random((h, w)), rng.uniform(1, 4)) > 0.5
        want = ndimage.binary_opening(x, structure=np.ones((9, 9)))
        got = x
        for op in (ndimage.binary_erosion, ndimage.binary_dilation):
            got = op(op(got, structure=W._ROW9), structure=W._COL9)
        np.testing.assert_array_equal(got, want)
    # and through place(): a 30 x 30 square with a one-pixel spur and a hole -- the spur goes, the hole is filled, the rim is eroded by 2
    c = W._MaskCanvas(64, 64, 96, 96, 1.0)
    p = np.zeros((64, 64), np.uint8)
    p[17:47, 17:47] = 255
    p[30, 47:60] = 255
    p[30:33, 30:33] = 0
    assert c.place(p, 20, 20, 0.0, 1.0, 0.5)
    ys, xs = np.nonzero(c.img)
    assert (ys.min(), ys.max(), xs.min(), xs.max()) == (20 + 19, 20 + 44, 20 + 19, 20 + 44) and c.img.sum() == 26 * 26
