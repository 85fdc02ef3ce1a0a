"""Pin the oracle against vectors produced by the reference's own control flow
(tests/golden/make_goldens.py ran CycleGanModel.train_step_torch / ImagePool from
/root/reference under a test-only keras stub).  CPU only."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import nets, steps


def _load_nets(z, filters):
    ga = nets.ResnetGenerator(filters=filters)
    gb = nets.ResnetGenerator(filters=filters)
    da = nets.PatchDiscriminator(filters=2 * filters)
    db = nets.PatchDiscriminator(filters=2 * filters)
    for nm, net in (("gen_a", ga), ("gen_b", gb), ("disc_a", da), ("disc_b", db)):
        net.set_weights([z[f"init/{nm}/{i}"] for i in range(len(net.variables))])
    return ga, gb, da, db


@pytest.mark.parametrize("fname", ["cyclegan_step_n5_s64_f4.npz", "cyclegan_step_n2_s64_f4.npz"])
def test_cyclegan_step_matches_reference_control_flow(golden_dir, fname):
    z = np.load(os.path.join(golden_dir, fname))
    n, size, filters, n_steps, seed = (int(v) for v in z["meta"])
    ga, gb, da, db = _load_nets(z, filters)
    random.seed(seed)
    step = steps.CycleGanStep(ga, gb, da, db, steps.ImagePool(2, 3), steps.ImagePool(2, 3))
    for s in range(n_steps):
        m = step.train_step((torch.from_numpy(z[f"step{s}/real_a"]), torch.from_numpy(z[f"step{s}/real_b"])))
        names = [str(x) for x in z[f"step{s}/metric_names"]]
        assert sorted(m) == names
        got = np.array([m[k] for k in names])
        np.testing.assert_allclose(got, z[f"step{s}/metrics"], rtol=1e-6, atol=1e-7)
    for nm, net in (("gen_a", ga), ("gen_b", gb), ("disc_a", da), ("disc_b", db)):
        for i, w in enumerate(net.get_weights()):
            np.testing.assert_allclose(w, z[f"final/{nm}/{i}"], rtol=1e-5, atol=1e-7, err_msg=f"{nm}/{i}")


def test_image_pool_trace(golden_dir):
    z = np.load(os.path.join(golden_dir, "image_pool_trace.npz"))
    random.seed(0)
    pool = steps.ImagePool(batch_size=2, pool_size=5)
    for i in range(int(z["n_steps"])):
        out = pool.query(torch.from_numpy(z[f"in{i}"]).clone())
        np.testing.assert_array_equal(out.numpy(), z[f"out{i}"])
        assert out.shape[0] == 2  # frozen loop bound: only the first two images are ever returned


def test_summed_generator_gradients_semantics():
    """torch-backend quirk (CycleGAN.py:664-665): each generator is updated with d(L_a+L_b)/dtheta."""
    torch.manual_seed(0)
    ga = nets.ResnetGenerator(filters=2, num_residual_blocks=1, seed=1)
    gb = nets.ResnetGenerator(filters=2, num_residual_blocks=1, seed=2)
    da = nets.PatchDiscriminator(filters=4, seed=3)
    db = nets.PatchDiscriminator(filters=4, seed=4)
    real_a = torch.rand(2, 64, 64, 1) * 2 - 1
    real_b = torch.rand(2, 64, 64, 1) * 2 - 1
    st = steps.CycleGanStep(ga, gb, da, db)
    w0 = [w.copy() for w in ga.get_weights()]
    st.train_step((real_a, real_b))
    # first Adam step moves every weight with non-zero grad by ~lr regardless of scale
    moved = [np.abs(a - b).max() for a, b in zip(ga.get_weights(), w0)]
    assert max(moved) > 1e-4


def test_unet_param_count_and_shapes():
    net = nets.MultiResUNet(16)
    total = sum(int(np.prod(v.shape)) for v in net.variables)
    trainable = sum(int(np.prod(v.shape)) for v in net.trainable_weights)
    assert total == 2429491 and trainable == 2414297
    y = net(torch.rand(1, 32, 32, 1), training=True)
    assert y.shape == (1, 32, 32, 1)


def test_generator_discriminator_param_counts():
    g = nets.ResnetGenerator(64)
    d = nets.PatchDiscriminator(128)
    assert sum(int(np.prod(v.shape)) for v in g.variables) == 45591425
    assert sum(int(np.prod(v.shape)) for v in d.variables) == 2633345
