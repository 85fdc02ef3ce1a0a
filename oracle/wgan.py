"""Oracle for the WGAN-GP mask synthesiser (test infrastructure; see oracle/__init__.py).

Restates, on top of oracle/ops.py and torch autograd:

``WganGenerator``   -- WGAN.get_generator_model, WassersteinGAN.py:644-681 (``upsample_block`` :615-642)
``WganCritic``      -- WGAN.get_discriminator_model, WassersteinGAN.py:569-613 (``conv_block`` :548-567)
``WganStep``        -- WGAN_GP.train_step_torch, WassersteinGAN.py:177-234, with ``gradient_penalty`` :86-117, the losses
                       :687-697 and the optimizers of ``create_model`` :702-703.

Everything the reference draws from Keras' global RNG inside a step (latent vectors, the interpolation factor -- a NORMAL variate,
WassersteinGAN.py:97 -- and the Dropout keep masks) is an explicit argument here, so that the HIP path can be compared on identical
draws.  Pinned by tests/golden/wgan_topology.npz: the reference's own builder functions executed under the layer-level keras
stand-in of tests/golden/make_topology_goldens.py.
"""
import torch

from . import ops
from .nets import Net

DROP_CONV, DROP_FLAT = 0.3, 0.2          # WassersteinGAN.py:589,599,609


def dropout(x, keep, rate):
    """keras.layers.Dropout(rate) in training mode with an explicit keep mask (1 = kept): x * keep / (1 - rate)."""
    return x if keep is None else x * keep * (1.0 / (1.0 - rate))


class WganGenerator(Net):
    """noise (n, n_z) -> Dense(h/8 * w/8 * 256, no bias) -> BN -> LeakyReLU(0.2) -> Reshape(h/8, w/8, 256) ->
    3 x [UpSampling2D(2) -> Conv2D(3x3, same, no bias) -> BN -> LeakyReLU(0.2) | tanh] with 128, 64, 1 filters."""

    def __init__(self, height, width, n_z=128, dtype=torch.float32, seed=0):
        super().__init__(dtype, seed)
        self.h8, self.w8, self.n_z = height // 8, width // 8, n_z
        units = height // 8 * width // 8 * 4 * 8 * 8         # the reference's expression, left to right (WassersteinGAN.py:646)
        assert units == self.h8 * self.w8 * 256, "the reference's Reshape only works for sizes divisible by 8"
        self.add_kernel("dense/kernel", (n_z, units))
        self._bn_make("bn0", units)
        cin = 256
        for i, f in enumerate((128, 64, 1)):
            self.add_kernel(f"up{i}/kernel", (3, 3, cin, f))
            self._bn_make(f"up{i}.bn", f)
            cin = f

    def _bn_make(self, name, c):
        self.add_ones(f"{name}/gamma", (c,))
        self.add_zeros(f"{name}/beta", (c,))
        self.add_zeros(f"{name}/moving_mean", (c,), trainable=False)
        self.add_ones(f"{name}/moving_variance", (c,), trainable=False)

    def _bn(self, name, x, training):
        mm, mv = self._by_name[f"{name}/moving_mean"], self._by_name[f"{name}/moving_variance"]
        y, nmm, nmv = ops.batch_norm(x, self.p(f"{name}/gamma"), self.p(f"{name}/beta"), mm.value, mv.value, training)
        if training:
            mm.value, mv.value = nmm.detach(), nmv.detach()
        return y

    def forward(self, z, training=True):
        n = z.shape[0]
        x = (z @ self.p("dense/kernel")).reshape(n, 1, 1, -1)
        x = ops.leaky_relu(self._bn("bn0", x, training), 0.2)
        x = x.reshape(n, self.h8, self.w8, 256)
        for i in range(3):
            x = x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
            x = ops.conv2d(x, self.p(f"up{i}/kernel"), None, stride=1, padding="same")
            x = self._bn(f"up{i}.bn", x, training)
            x = ops.leaky_relu(x, 0.2) if i < 2 else torch.tanh(x)
        return x


class WganCritic(Net):
    """4 x [Conv2D(5x5, stride 2, same, bias) -> LeakyReLU(0.2) (-> Dropout(0.3) after blocks 1 and 2)] with 64..512 filters ->
    Flatten -> Dropout(0.2) -> Dense(1)."""

    FILTERS = (64, 128, 256, 512)

    def __init__(self, height, width, channels=1, dtype=torch.float32, seed=0):
        super().__init__(dtype, seed)
        cin = channels
        for i, f in enumerate(self.FILTERS):
            self.add_kernel(f"conv{i}/kernel", (5, 5, cin, f))
            self.add_zeros(f"conv{i}/bias", (f,))
            cin = f
        hh, ww = height, width
        for _ in range(4):
            hh, ww = -(-hh // 2), -(-ww // 2)
        self.flat = hh * ww * 512
        self.add_kernel("dense/kernel", (self.flat, 1))
        self.add_zeros("dense/bias", (1,))

    def forward(self, x, training=True, keep=None):
        """keep: {'drop1', 'drop2', 'flat'} -> keep masks (shapes of the tensors they multiply), or None for no dropout."""
        keep = keep or {}
        for i in range(4):
            x = ops.conv2d(x, self.p(f"conv{i}/kernel"), self.p(f"conv{i}/bias"), stride=2, padding="same")
            x = ops.leaky_relu(x, 0.2)
            if i in (1, 2) and training:
                x = dropout(x, keep.get(f"drop{i}"), DROP_CONV)
        x = x.reshape(x.shape[0], -1)
        if training:
            x = dropout(x, keep.get("flat"), DROP_FLAT)
        return x @ self.p("dense/kernel") + self.p("dense/bias")

    def __call__(self, x, training=True, keep=None):
        return self.forward(x, training, keep)


class WganStep:
    """WGAN_GP.train_step_torch.  ``draws`` of train_step: dict with
    'z'     : d_steps + 1 latent batches (the last one feeds the generator update),
    'alpha' : d_steps tensors (n, 1, 1, 1),
    'keep_fake', 'keep_real', 'keep_gp' : d_steps dropout mask dicts each, 'keep_gen' : one (None entries = no dropout)."""

    def __init__(self, generator, critic, d_steps=3, gp_weight=10.0, learning_rate=0.0002, beta_1=0.5, beta_2=0.9):
        self.g, self.d, self.d_steps, self.gp_weight = generator, critic, d_steps, gp_weight
        self.g_opt = ops.KerasAdam(learning_rate, beta_1, beta_2)
        self.d_opt = ops.KerasAdam(learning_rate, beta_1, beta_2)

    def gradient_penalty(self, real, fake, alpha, keep):
        interpolated = real + alpha * (fake - real)
        pred = self.d(interpolated, True, keep)
        grads = torch.autograd.grad(outputs=pred, inputs=interpolated, grad_outputs=torch.ones_like(pred), create_graph=True,
                                    retain_graph=True)[0]
        norm = torch.sqrt(torch.sum(grads * grads, dim=[1, 2, 3]))
        return torch.mean((norm - 1.0) ** 2), norm

    def train_step(self, real, draws):
        out = {}
        for i in range(self.d_steps):
            fake = self.g(draws["z"][i], True)
            fake_logits = self.d(fake, True, draws["keep_fake"][i])
            real_logits = self.d(real, True, draws["keep_real"][i])
            d_cost = fake_logits.mean() - real_logits.mean()
            gp, gn = self.gradient_penalty(real, fake, draws["alpha"][i], draws["keep_gp"][i])
            d_loss = d_cost + gp * self.gp_weight
            self.d.zero_grad()
            d_loss.backward()
            tw = self.d.trainable_weights
            out.setdefault("d_grads", []).append([v.value.grad.detach().clone() for v in tw])
            with torch.no_grad():
                self.d_opt.apply([v.value.grad for v in tw], tw)
        generated = self.g(draws["z"][self.d_steps], True)
        g_loss = -self.d(generated, True, draws["keep_gen"]).mean()
        self.g.zero_grad()
        g_loss.backward()
        tw = self.g.trainable_weights
        out["g_grads"] = [v.value.grad.detach().clone() for v in tw]
        with torch.no_grad():
            self.g_opt.apply([v.value.grad for v in tw], tw)
        out.update(d_loss=float(d_cost.detach()), d_total_loss=float(d_loss.detach()), g_loss=float(g_loss.detach()),
                   grad_penalty=float(gp.detach()), grad_norm=float(gn.detach().mean()), generated=generated.detach())
        return out
