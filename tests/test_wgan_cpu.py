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
