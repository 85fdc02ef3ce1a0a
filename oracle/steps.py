"""Oracle train steps and image buffer (test infrastructure; see oracle/__init__.py).

``ImagePool``            -- CycleGAN.py:908-964 (incl. the frozen ``batch_size`` loop bound).
``CycleGanStep``         -- CycleGanModel.train_step_torch, CycleGAN.py:615-710.
``UNetStep``             -- Keras default train_step as driven by UNet_Segmentation.py:277-283,
                            loss UNet_Segmentation.py:379-384, compile :393-395.
"""
import random

import torch

from . import ops

METRIC_NAMES = ("d_a", "d_b", "d_fake_a", "d_fake_b", "d_real_a", "d_real_b", "g_a", "g_b",
                "g_adv_a", "g_adv_b", "g_cyc_a", "g_cyc_b", "g_id_a", "g_id_b")  # CycleGAN.py:547-560


class ImagePool:
    """History buffer of generated images.  ``rng`` defaults to the module-level python RNG
    exactly as the reference uses it (CycleGAN.py:955-957)."""

    def __init__(self, batch_size, pool_size=50, rng=random):
        self.pool_size = pool_size
        self.batch_size = batch_size
        self.rng = rng
        self.num_imgs = 0
        self.images = []

    def query(self, images):
        if self.pool_size == 0:
            return images
        out = []
        for index in range(self.batch_size):
            if index >= images.shape[0]:
                # The reference relies on a ValueError from images[index] (CycleGAN.py:945-948);
                # torch raises IndexError instead.  Stopping the loop is the documented intent.
                break
            image = images[index:index + 1]
            if self.num_imgs < self.pool_size:
                self.num_imgs += 1
                self.images.append(image)
                out.append(image)
            else:
                p = self.rng.uniform(0, 1)
                if p > 0.5:
                    rid = self.rng.randint(0, self.pool_size - 1)
                    tmp = self.images[rid].clone()
                    self.images[rid] = image
                    out.append(tmp)
                else:
                    out.append(image)
        return torch.cat(out, 0)


class CycleGanStep:
    """One optimisation step of G_a, G_b, D_a, D_b with the torch-backend semantics:
    both generator losses are back-propagated into both generators (summed grads, CycleGAN.py:664-665)."""

    def __init__(self, gen_a, gen_b, disc_a, disc_b, pool_a=None, pool_b=None,
                 lambda_cycle_a=10.0, lambda_cycle_b=10.0, lambda_identity_a=0.5, lambda_identity_b=0.5,
                 learning_rate=2e-4, beta_1=0.5, label_smoothing_factor=0.0):
        self.gen_a, self.gen_b, self.disc_a, self.disc_b = gen_a, gen_b, disc_a, disc_b
        self.pool_a = pool_a if pool_a is not None else ImagePool(1, 0)
        self.pool_b = pool_b if pool_b is not None else ImagePool(1, 0)
        self.lca, self.lcb, self.lia, self.lib = lambda_cycle_a, lambda_cycle_b, lambda_identity_a, lambda_identity_b
        self.use_identity = lambda_identity_a > 0 or lambda_identity_b > 0
        self.ls = label_smoothing_factor
        self.opt = {k: ops.KerasAdam(learning_rate, beta_1) for k in ("gen_a", "gen_b", "disc_a", "disc_b")}
        self.sums = {k: 0.0 for k in METRIC_NAMES}
        self.count = 0

    # CycleGAN.py:301-308
    def _gen_loss(self, fake):
        return ops.mse(torch.ones_like(fake) * (1.0 - self.ls) + self.ls / 2, fake)

    def _disc_loss(self, real, fake):
        real_loss = ops.mse(torch.ones_like(real) * (1.0 - self.ls) + self.ls / 2, real)
        fake_loss = ops.mse(torch.zeros_like(fake) * (1.0 - self.ls) + self.ls / 2, fake)
        return (real_loss + fake_loss) * 0.5, real_loss, fake_loss

    def reset_metrics(self):
        self.sums = {k: 0.0 for k in METRIC_NAMES}
        self.count = 0

    def train_step(self, batch):
        real_a, real_b = batch
        ga, gb, da, db = self.gen_a, self.gen_b, self.disc_a, self.disc_b
        fake_b = ga(real_a, True)
        fake_a = gb(real_b, True)
        cycled_a = gb(fake_b, True)
        cycled_b = ga(fake_a, True)
        if self.use_identity:
            same_a = gb(real_a, True)
            same_b = ga(real_b, True)
        disc_fake_a = da(fake_a, True)
        disc_fake_b = db(fake_b, True)
        adv_a = self._gen_loss(disc_fake_b)
        adv_b = self._gen_loss(disc_fake_a)
        cyc_a = ops.mae(real_b, cycled_b) * self.lca
        cyc_b = ops.mae(real_a, cycled_a) * self.lcb
        if self.use_identity:
            id_a = ops.mae(real_b, same_b) * self.lca * self.lia
            id_b = ops.mae(real_a, same_a) * self.lcb * self.lib
        else:
            id_a = id_b = torch.zeros((), dtype=real_a.dtype)
        total_a = adv_a + cyc_a + id_a
        total_b = adv_b + cyc_b + id_b
        ga.zero_grad()
        gb.zero_grad()
        total_a.backward(retain_graph=True)
        total_b.backward(retain_graph=True)
        with torch.no_grad():
            self.opt["gen_a"].apply([v.value.grad for v in ga.trainable_weights], ga.trainable_weights)
            self.opt["gen_b"].apply([v.value.grad for v in gb.trainable_weights], gb.trainable_weights)

        disc_real_a = da(real_a, True)
        disc_fake_a = da(self.pool_a.query(fake_a.detach().clone()), True)
        disc_real_b = db(real_b, True)
        disc_fake_b = db(self.pool_b.query(fake_b.detach().clone()), True)
        d_a, d_a_real, d_a_fake = self._disc_loss(disc_real_a, disc_fake_a)
        d_b, d_b_real, d_b_fake = self._disc_loss(disc_real_b, disc_fake_b)
        da.zero_grad()
        db.zero_grad()
        d_a.backward()
        d_b.backward()
        with torch.no_grad():
            self.opt["disc_a"].apply([v.value.grad for v in da.trainable_weights], da.trainable_weights)
            self.opt["disc_b"].apply([v.value.grad for v in db.trainable_weights], db.trainable_weights)

        vals = dict(d_a=d_a, d_b=d_b, d_fake_a=d_a_fake, d_fake_b=d_b_fake, d_real_a=d_a_real,
                    d_real_b=d_b_real, g_a=total_a, g_b=total_b, g_adv_a=adv_a, g_adv_b=adv_b,
                    g_cyc_a=cyc_a, g_cyc_b=cyc_b, g_id_a=id_a, g_id_b=id_b)
        self.count += 1
        for k in METRIC_NAMES:
            self.sums[k] += float(vals[k].detach())
        # keras.metrics.Mean: running mean since the last reset (CycleGAN.py:695-710)
        return {k: self.sums[k] / self.count for k in METRIC_NAMES}


class UNetStep:
    """fwd(training=True) -> weighted BCE -> backward -> Keras Adam; metrics loss / mae / acc (T12)."""

    def __init__(self, net, weighting, learning_rate=1e-3):
        self.net = net
        self.weighting = weighting
        self.opt = ops.KerasAdam(learning_rate)

    def train_step(self, batch):
        x, y = batch
        self.net.zero_grad()
        p = self.net(x, True)
        loss = ops.weighted_bce(y, p, self.weighting)
        loss.backward()
        tw = self.net.trainable_weights
        with torch.no_grad():
            self.opt.apply([v.value.grad for v in tw], tw)
            mae = (y - p).abs().mean()
            acc = ((p > 0.5).to(y.dtype) == y).to(y.dtype).mean()
        return {"loss": float(loss.detach()), "mae": float(mae), "acc": float(acc)}, p.detach()
