"""WGAN-GP particle-mask synthesiser on the HIP engine -- mirrors ``Releases/Version 1.2.0/WassersteinGAN.py``.

``WganGenerator`` / ``WganCritic``  = ``WGAN.get_generator_model`` / ``get_discriminator_model`` (WassersteinGAN.py:569-681)
``WGAN_GP``                         = the model class (WassersteinGAN.py:26-257): ``compile`` + ``train_step`` with
                                      ``discriminator_extra_steps`` critic updates and the gradient penalty (:86-117)
``WGAN``                            = the workflow class (WassersteinGAN.py:288-545, 683-724): training set from ``Input_Masks``,
                                      ``start_training``, ``simulate_masks``

Gradient penalty without a second-order autograd: the critic is piecewise linear in its input (convolutions, LeakyReLU, Dropout
masks, Dense), so g = d critic(x^) / d x^ is the BACKWARD chain  dz_l = m_l .* da_l,  da_{l-1} = conv_l^T(dz_l)  (m_l = LeakyReLU
slope x Dropout keep of the forward at x^), a computation that is linear in every weight W_l.  Its adjoint is again made of the
existing convolution passes:  with t_0 = d gp / d g,
    dW_l += bwd_weight(x = t_{l-1}, dy = dz_l),      t_l = m_l .* conv_l(t_{l-1})   (forward convolution, no bias)
and for the Dense head dW += bwd_weight(x = keep .* t_4, dy = 1).  This is exactly what torch.autograd computes with
``create_graph=True`` (WassersteinGAN.py:109): the masks have zero derivative almost everywhere, biases get no gradient from the
penalty.  tests/test_wgan_gpu.py holds it to the oracle (oracle/wgan.py, autograd double backward) on identical draws.
"""
import ctypes
import math
import os
import random
import time

import numpy as np
import torch

from . import _lib as L
from . import dist as D
from . import layers as LY
from .engine import Act, Tape, _p, _stream, cat_batch, workspace
from .layers import Conv2D, Norm
from .nets import Network
from .optim import Adam

DROP_CONV, DROP_FLAT = 0.3, 0.2          # WassersteinGAN.py:589,599,609
METRIC_NAMES = ("d_loss", "d_total_loss", "g_loss", "grad_penalty", "grad_norm")          # WassersteinGAN.py:47-57


class WganGenerator(Network):
    """noise (n, n_z) -> Dense(h/8 * w/8 * 256, no bias) -> BN -> LeakyReLU(0.2) -> Reshape -> 3 x [UpSampling2D -> Conv2D(3x3,
    same, no bias) -> BN -> LeakyReLU(0.2) | tanh], 128 / 64 / 1 filters (WassersteinGAN.py:644-681)."""

    def __init__(self, height, width, n_z=128, device="cuda", seed=0):
        super().__init__(device)
        A = self.arena
        self.h8, self.w8, self.n_z = height // 8, width // 8, n_z
        units = height // 8 * width // 8 * 4 * 8 * 8          # the reference's expression, evaluated left to right
        assert units == self.h8 * self.w8 * 256, "the reference's Reshape needs sizes divisible by 8"
        self.dense = Conv2D(A, "dense", 1, n_z, units)          # Dense on (n, 1, 1, n_z): a 1x1 convolution
        self.dense.keras_kind = "dense"
        self.bn0 = Norm(A, "bn0", units, "batch")
        self.ups = []
        cin = 256
        for i, f in enumerate((128, 64, 1)):
            self.ups.append((Conv2D(A, f"up{i}", 3, cin, f, padding="same"), Norm(A, f"up{i}.bn", f, "batch")))
            cin = f
        self._finish(seed)

    def forward(self, tape, z, training=True):
        assert (z.h, z.w, z.c) == (1, 1, self.n_z)
        x = self.bn0(tape, self.dense(tape, z), act="lrelu", act_alpha=0.2, training=training)
        x = LY.reshape(tape, x, self.h8, self.w8, 256)
        for i, (conv, bn) in enumerate(self.ups):
            x = conv(tape, LY.upsample2x(tape, x))
            x = bn(tape, x, act="lrelu" if i < 2 else "tanh", act_alpha=0.2, training=training)
        return x

    def __call__(self, z, training=True, tape=None):
        if not isinstance(z, Act):
            z = torch.as_tensor(z, dtype=torch.float32).to(self.device)
            z = Act(z.reshape(z.shape[0], 1, 1, -1).contiguous(), requires_grad=False)
        return super().__call__(z, training, tape)


class WganCritic(Network):
    """4 x [Conv2D(5x5, stride 2, same, bias) -> LeakyReLU(0.2) (-> Dropout(0.3) after blocks 1 and 2)], 64..512 filters ->
    Flatten -> Dropout(0.2) -> Dense(1) (WassersteinGAN.py:569-613)."""

    FILTERS = (64, 128, 256, 512)

    def __init__(self, height, width, channels=1, device="cuda", seed=0):
        super().__init__(device)
        A = self.arena
        self.convs = []
        cin = channels
        for i, f in enumerate(self.FILTERS):
            self.convs.append(Conv2D(A, f"conv{i}", 5, cin, f, stride=2, padding="same", use_bias=True, act="lrelu", act_alpha=0.2))
            cin = f
        hh, ww = height, width
        for _ in range(4):
            hh, ww = -(-hh // 2), -(-ww // 2)
        self.fh, self.fw = hh, ww
        self.flat = hh * ww * 512
        self.dense = Conv2D(A, "dense", 1, self.flat, 1, use_bias=True)
        self.dense.keras_kind = "dense"
        self._finish(seed)

    def draw_keep(self, n):
        """Dropout keep masks for a batch of n samples, from torch's device generator (keras: ``random.uniform >= rate``)."""
        def mask(shape, rate):
            return Act((torch.rand(shape, device=self.device) >= rate).to(torch.float32), requires_grad=False)
        h1, w1 = self.convs[1].out_hw(*self.convs[0].out_hw(self.in_hw[0], self.in_hw[1]))
        h2, w2 = self.convs[2].out_hw(h1, w1)
        return {"drop1": mask((n, h1, w1, 128), DROP_CONV), "drop2": mask((n, h2, w2, 256), DROP_CONV),
                "flat": mask((n, 1, 1, self.flat), DROP_FLAT)}

    in_hw = (64, 64)

    def forward(self, tape, x, training=True, keep=None, trace=None):
        """keep: {'drop1', 'drop2', 'flat'} -> Act keep masks (None / missing entry: that Dropout is the identity).
        trace (list): receives the post-activation output of every conv block -- what the gradient-penalty chain needs."""
        keep = (keep or {}) if training else {}
        for i, conv in enumerate(self.convs):
            x = conv(tape, x)
            if trace is not None:
                trace.append(x)
            if i in (1, 2):
                x = LY.dropout(tape, x, _as_act(keep.get(f"drop{i}"), x), DROP_CONV)
        x = LY.reshape(tape, x, 1, 1, self.flat)
        x = LY.dropout(tape, x, _as_act(keep.get("flat"), x), DROP_FLAT)
        return self.dense(tape, x)

    def __call__(self, x, training=True, tape=None, keep=None, trace=None):
        if not isinstance(x, Act):
            x = Act(torch.as_tensor(x, dtype=torch.float32).to(self.device).contiguous(), requires_grad=False)
        self.in_hw = (x.h, x.w)
        return self.forward(tape if tape is not None else Tape(enabled=False), x, training, keep, trace)

    # ---- gradient penalty ------------------------------------------------------------------------------------------------
    def gradient_penalty(self, interpolated, keep, coef, want_grads=True):
        """norms (device tensor, one per sample) of g = d critic / d interpolated, and -- accumulated into this network's gradient
        arena -- d/dW of  coef * sum_i (norm_i - 1)^2  (WassersteinGAN.py:86-117; see the module docstring for the derivation)."""
        lib = L.load()
        keep = keep or {}
        x0 = interpolated
        n = x0.n
        trace = []
        # forward at the interpolated images up to the last conv block (the head's output is not needed, only its weights)
        h = x0
        for i, conv in enumerate(self.convs):
            h = conv(Tape(enabled=False), h)
            trace.append(h)
            if i in (1, 2):
                h = LY.dropout(Tape(enabled=False), h, _as_act(keep.get(f"drop{i}"), h), DROP_CONV)
        drop = {1: _as_act(keep.get("drop1"), trace[1]), 2: _as_act(keep.get("drop2"), trace[2])}
        kflat = _as_act(keep.get("flat"), Act(trace[3].t.view(n, 1, 1, self.flat), requires_grad=False))
        ins = [x0, trace[0], trace[1], trace[2]]          # geometry of every conv's input (Dropout keeps the shape)

        # ---- backward chain: g --------------------------------------------------------------------------------------------
        ones = Act(torch.ones((n, 1, 1, 1), dtype=torch.float32, device=self.device), requires_grad=False)
        flat_in = Act(trace[3].t.view(n, 1, 1, self.flat), requires_grad=False)
        dh = flat_in.like(requires_grad=False)
        self._bwd_data(self.dense, flat_in, ones, dh)
        da = _mul(dh, kflat, 1.0 / (1.0 - DROP_FLAT)) if kflat is not None else dh
        da = Act(da.t.view(trace[3].t.shape), requires_grad=False)
        dz = [None] * 4
        for i in (3, 2, 1, 0):
            y = trace[i]
            if i in (1, 2) and drop[i] is not None:
                da = _mul(da, drop[i], 1.0 / (1.0 - DROP_CONV))
            dzi = y.like(requires_grad=False)
            L.check(lib.ss_act_bwd_t(y.dt, L.ACT_LRELU, 0.2, da.ptr, da.cs, y.ptr, y.cs, dzi.ptr, dzi.cs, y.rows, y.c, _stream()), "act_bwd")
            dz[i] = dzi
            da = ins[i].like(requires_grad=False)
            self._bwd_data(self.convs[i], ins[i], dzi, da)
        g = da
        norms = torch.empty(n, dtype=torch.float32, device=self.device)
        gbar = g.like(requires_grad=False) if want_grads else None
        L.check(lib.ss_wgan_gp_grad(g.ptr, n, g.h * g.w * g.c, float(coef), gbar.ptr if want_grads else None, _p(norms), _stream()),
                "wgan_gp_grad")
        if not want_grads:
            return norms
        # ---- adjoint chain: weight gradients --------------------------------------------------------------------------------
        t = gbar
        for i in range(4):
            conv, y = self.convs[i], trace[i]
            self._bwd_weight(conv, t, dz[i])
            s = y.like(requires_grad=False)
            self._fwd_linear(conv, t, s)
            tl = y.like(requires_grad=False)
            L.check(lib.ss_act_bwd_t(y.dt, L.ACT_LRELU, 0.2, s.ptr, s.cs, y.ptr, y.cs, tl.ptr, tl.cs, y.rows, y.c, _stream()), "act_bwd")
            if i in (1, 2) and drop[i] is not None:
                tl = _mul(tl, drop[i], 1.0 / (1.0 - DROP_CONV))
            t = tl
        t = Act(t.t.view(n, 1, 1, self.flat), requires_grad=False)
        if kflat is not None:
            t = _mul(t, kflat, 1.0 / (1.0 - DROP_FLAT))
        self._bwd_weight(self.dense, t, ones)
        return norms

    # raw passes of a layer on caller-chosen tensors (the tape ops in layers.Conv2D only cover the ordinary backward)
    def _desc(self, conv, x, y):
        d = L.ConvDesc.from_buffer_copy(conv.desc(x, y))
        d.act, d.x_amax, d.dy_amax, d.x_amax_valid, d.dy_amax_valid, d.w_cache = L.ACT_NONE, None, None, 0, 0, None
        return d

    def _bwd_data(self, conv, x_like, dy, dx):
        lib = L.load()
        d = self._desc(conv, x_like, dy)
        ws = workspace(lib.ss_conv2d_workspace_bytes(ctypes.byref(d), L.PASS_BWD_DATA), self.device)
        L.check(lib.ss_conv2d_bwd_data(ctypes.byref(d), dy.ptr, _p(self.arena[f"{conv.name}/kernel"]), dx.ptr, 0, _p(ws), ws.numel(), _stream()),
                f"conv2d_bwd_data[{conv.name}] (gradient penalty)")

    def _bwd_weight(self, conv, x, dy):
        lib = L.load()
        d = self._desc(conv, x, dy)
        ws = workspace(lib.ss_conv2d_workspace_bytes(ctypes.byref(d), L.PASS_BWD_WEIGHT), self.device)
        L.check(lib.ss_conv2d_bwd_weight(ctypes.byref(d), x.ptr, dy.ptr, _p(self.arena.grad(f"{conv.name}/kernel")), None, 1, _p(ws), ws.numel(),
                                         _stream()), f"conv2d_bwd_weight[{conv.name}] (gradient penalty)")

    def _fwd_linear(self, conv, x, y):
        lib = L.load()
        d = self._desc(conv, x, y)
        ws = workspace(lib.ss_conv2d_workspace_bytes(ctypes.byref(d), L.PASS_FWD), self.device)
        L.check(lib.ss_conv2d_fwd(ctypes.byref(d), x.ptr, _p(self.arena[f"{conv.name}/kernel"]), None, y.ptr, _p(ws), ws.numel(), _stream()),
                f"conv2d_fwd[{conv.name}] (gradient penalty)")


def _as_act(keep, like):
    if keep is None or isinstance(keep, Act):
        return keep
    t = torch.as_tensor(keep, dtype=torch.float32).to(like.device).reshape(like.t.shape).contiguous()
    return Act(t, requires_grad=False)


def _mul(a, b, scale):
    out = a.like(requires_grad=False)
    L.check(L.load().ss_mul_t(a.dt, float(scale), a.ptr, a.cs, b.ptr, b.cs, out.ptr, out.cs, a.rows, a.c, _stream()), "mul")
    return out


def discriminator_loss(real_img, fake_img):
    """WGAN.discriminator_loss (WassersteinGAN.py:687-691): mean(fake logits) - mean(real logits)."""
    return float(np.mean(_np(fake_img)) - np.mean(_np(real_img)))


def generator_loss(fake_img):
    """WGAN.generator_loss (WassersteinGAN.py:694-696)."""
    return float(-np.mean(_np(fake_img)))


def _np(x):
    if isinstance(x, Act):
        x = x.dense()
    if isinstance(x, torch.Tensor):
        x = x.detach().float().cpu().numpy()
    return np.asarray(x)


class WGAN_GP:
    """keras.Model surface of the reference class: ``WGAN_GP(discriminator, generator, latent_dim, discriminator_extra_steps=3,
    gp_weight=10.0)``, ``compile(d_optimizer, g_optimizer, d_loss_fn, g_loss_fn)``, ``train_step(real_images) -> metrics``,
    ``__call__(latent) -> images``."""

    def __init__(self, discriminator, generator, latent_dim, discriminator_extra_steps=3, gp_weight=10.0, **kwargs):
        self.discriminator, self.generator, self.latent_dim = discriminator, generator, latent_dim
        self.d_steps, self.gp_weight = discriminator_extra_steps, gp_weight
        self.d_optimizer = self.g_optimizer = self.d_loss_fn = self.g_loss_fn = None
        self.device = generator.device
        self.keep_grads = False          # tests: keep every update's gradients in self.grad_log
        self.grad_log = {"d": [], "g": []}
        self.reset_metrics()
        D.broadcast_params([self.generator, self.discriminator])

    def compile(self, d_optimizer, g_optimizer, d_loss_fn=discriminator_loss, g_loss_fn=generator_loss, **kwargs):
        self.d_optimizer, self.g_optimizer, self.d_loss_fn, self.g_loss_fn = d_optimizer, g_optimizer, d_loss_fn, g_loss_fn

    def reset_metrics(self):
        self._sums, self._count = np.zeros(len(METRIC_NAMES)), 0

    def __call__(self, inputs, training=False):
        return self.generator(inputs, training)

    def draw(self, n):
        """Everything one step takes from the random generator (keras.random.normal / Dropout in the reference)."""
        dev, ds = self.device, self.d_steps
        return {"z": [torch.randn((n, self.latent_dim), device=dev) for _ in range(ds + 1)],
                "alpha": [torch.randn((n, 1, 1, 1), device=dev) for _ in range(ds)],
                "keep_fake": [self.discriminator.draw_keep(n) for _ in range(ds)],
                "keep_real": [self.discriminator.draw_keep(n) for _ in range(ds)],
                "keep_gp": [self.discriminator.draw_keep(n) for _ in range(ds)],
                "keep_gen": self.discriminator.draw_keep(n)}

    def train_step(self, real_images, draws=None):
        """WGAN_GP.train_step_torch (WassersteinGAN.py:177-234).  ``draws`` (tests): the step's random numbers, see draw()."""
        if isinstance(real_images, tuple):
            real_images = real_images[0]
        lib = L.load()
        gen, crit = self.generator, self.discriminator
        real = real_images if isinstance(real_images, Act) else Act(
            torch.as_tensor(real_images, dtype=torch.float32).to(self.device).contiguous(), requires_grad=False)
        n = real.n
        crit.in_hw = (real.h, real.w)
        world = D.world_size()
        if draws is None:
            draws = self.draw(n)
        inv_n = 1.0 / n
        for i in range(self.d_steps):
            # the generator runs in training mode (its BatchNorm moving statistics advance) but nothing is back-propagated into it:
            # the reference discards those gradients (only the critic's variables are handed to the optimizer)
            fake = gen(draws["z"][i], True)
            fake = Act(fake.t, requires_grad=False)
            tape = Tape()
            both = Act(cat_batch([fake.t, real.t]), requires_grad=False)
            keep = _cat_keep(draws["keep_fake"][i], draws["keep_real"][i], self.device)
            logits = crit(both, True, tape, keep)
            lg = logits.dense().reshape(2 * n)
            # d_cost = mean(fake) - mean(real): dlogit = +1/n for the fake half, -1/n for the real half
            gt, _ = logits.grad_target()
            L.check(lib.ss_fill(gt.ptr, inv_n, n, _stream()), "fill")
            L.check(lib.ss_fill(ctypes.c_void_p(gt.t.data_ptr() + 4 * n), -inv_n, n, _stream()), "fill")
            crit.zero_grad()
            D.begin_backward([crit])
            tape.backward()
            alpha = torch.as_tensor(draws["alpha"][i], dtype=torch.float32).to(self.device).reshape(n).contiguous()
            inter = real.like(requires_grad=False)
            L.check(lib.ss_wgan_interpolate(real.ptr, fake.ptr, _p(alpha), inter.ptr, n, real.h * real.w * real.c, _stream()), "interpolate")
            norms = crit.gradient_penalty(inter, draws["keep_gp"][i], self.gp_weight * inv_n)
            D.all_reduce_grads([crit])
            if self.keep_grads:
                self.grad_log["d"].append(crit.get_gradients())
            self.d_optimizer.apply(crit, 1.0 / world)
            if i == self.d_steps - 1:          # the reference's trackers see the LAST critic step's values (WassersteinGAN.py:227-231)
                lgh = lg.float().cpu().numpy().astype(np.float64)
                nrm = norms.cpu().numpy().astype(np.float64)
                d_cost = float(lgh[:n].mean() - lgh[n:].mean())
                gp = float(np.mean((nrm - 1.0) ** 2))
                gn = float(nrm.mean())
        # generator update
        tape = Tape()
        generated = gen(draws["z"][self.d_steps], True, tape)
        tape.param_grads = False
        logits = crit(generated, True, tape, draws["keep_gen"])
        tape.param_grads = True
        gt, _ = logits.grad_target()
        L.check(lib.ss_fill(gt.ptr, -inv_n, n, _stream()), "fill")
        gen.zero_grad()
        D.begin_backward([gen])
        tape.backward()
        D.all_reduce_grads([gen])
        if self.keep_grads:
            self.grad_log["g"].append(gen.get_gradients())
        self.g_optimizer.apply(gen, 1.0 / world)
        g_loss = float(-logits.dense().float().cpu().numpy().astype(np.float64).mean())
        vals = D.mean_scalars(np.array([d_cost, d_cost + gp * self.gp_weight, g_loss, gp, gn]))
        self._sums += vals
        self._count += 1
        self.last = dict(zip(METRIC_NAMES, (float(v) for v in vals)))
        return {k: float(self._sums[j] / self._count) for j, k in enumerate(METRIC_NAMES)}

    @staticmethod
    def to_numpy_array(x):
        return _np(x).copy()


def _cat_keep(a, b, device):
    out = {}
    for k in ("drop1", "drop2", "flat"):
        ka, kb = (a or {}).get(k), (b or {}).get(k)
        if ka is None and kb is None:
            continue
        assert ka is not None and kb is not None, "dropout masks for both halves of the batched critic pass, or for neither"
        ta = ka.t if isinstance(ka, Act) else torch.as_tensor(ka, dtype=torch.float32).to(device)
        tb = kb.t if isinstance(kb, Act) else torch.as_tensor(kb, dtype=torch.float32).to(device)
        if ta.dim() == 2:
            ta = ta.reshape(ta.shape[0], 1, 1, -1)
        out[k] = Act(cat_batch([ta.contiguous(), tb.reshape((tb.shape[0],) + tuple(ta.shape[1:])).contiguous()]), requires_grad=False)
    return out


class GANMonitor:
    """The reference's callback (WassersteinGAN.py:259-285): every ``output_epochs`` epochs a 3-column sheet of ``num_img`` generated
    masks, written as ``Epoch_#####.png`` (the reference draws the same grid with matplotlib)."""

    def __init__(self, output_dir, num_img=9, latent_dim=128, output_epochs=100):
        self.num_img, self.latent_dim, self.epochs, self.output_dir = num_img, latent_dim, output_epochs, output_dir

    def on_epoch_end(self, model, epoch, logs=None):
        if epoch % int(self.epochs) == 0:
            self.plot_reconstruction(model, epoch, nex=self.num_img)

    def plot_reconstruction(self, model, epoch, nex=9):
        from PIL import Image
        z = torch.randn((nex, self.latent_dim), device=model.device)
        samples = WGAN_GP.to_numpy_array(model(z))
        cols = 3
        rows = math.ceil(nex / float(cols))
        h, w = samples.shape[1], samples.shape[2]
        sheet = np.full((rows * (h + 4) + 4, cols * (w + 4) + 4), 255, np.uint8)
        for i, s in enumerate(samples):
            r, c = divmod(i, cols)
            sheet[4 + r * (h + 4):4 + r * (h + 4) + h, 4 + c * (w + 4):4 + c * (w + 4) + w] = (s * 127.5 + 127.5)[:, :, 0].astype('uint8')
        os.makedirs(self.output_dir, exist_ok=True)
        Image.fromarray(sheet).save(os.path.join(self.output_dir, 'Epoch_{:05d}.png'.format(epoch)))


_ROW9, _COL9 = np.ones((1, 9), dtype=bool), np.ones((9, 1), dtype=bool)


def _rotation_matrix_2d(center, angle, scale):
    """cv2.getRotationMatrix2D (WassersteinGAN.py:510): angle in degrees, positive = counter-clockwise, origin top-left."""
    a = scale * math.cos(math.radians(angle))
    b = scale * math.sin(math.radians(angle))
    return np.array([[a, b, (1 - a) * center[0] - b * center[1]], [-b, a, b * center[0] + (1 - a) * center[1]]], dtype=np.float64)


def _warp_affine(src, m, dsize):
    """cv2.warpAffine(src, M, (width, height)) with its defaults (bilinear, constant border 0) for a uint8 image: M maps source to
    destination, every destination pixel samples the source at M^-1 (x, y).  OpenCV evaluates this in fixed point (coordinates to
    1/32 pixel); this is the real-arithmetic form, rounded to nearest -- the result is thresholded at 127 right after
    (WassersteinGAN.py:522), so the two can differ on isolated edge pixels only (cv2 is not installed here: unpinned)."""
    from scipy import ndimage
    width, height = dsize
    full = np.vstack([m, [0.0, 0.0, 1.0]])
    inv = np.linalg.inv(full)
    xs, ys = np.arange(width, dtype=np.float64)[None, :], np.arange(height, dtype=np.float64)[:, None]          # broadcast: one row, one column
    sx = inv[0, 0] * xs + inv[0, 1] * ys + inv[0, 2]
    sy = inv[1, 0] * xs + inv[1, 1] * ys + inv[1, 2]
    out = ndimage.map_coordinates(src.astype(np.float64), [sy, sx], order=1, mode='constant', cval=0.0)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def _gradient_noise2array(ys, xs, rng):
    """Smooth 2-D gradient noise on the grid ys x xs (values roughly in [-1, 1]) -- the role of ``opensimplex.noise2array`` after
    ``opensimplex.random_seed()`` (WassersteinGAN.py:424-426).  The reference seeds it from the clock, so only its character
    (band-limited, one feature per unit of the coordinates) matters: classic Perlin noise with a random gradient lattice."""
    x0, y0 = np.floor(xs).astype(int), np.floor(ys).astype(int)
    nx, ny = int(x0.max()) + 2, int(y0.max()) + 2
    ang = rng.uniform(0, 2 * np.pi, (ny, nx))
    gx, gy = np.cos(ang), np.sin(ang)
    fx, fy = (xs - x0)[None, :], (ys - y0)[:, None]

    def corner(dx, dy):          # lattice gradients at (y0 + dy, x0 + dx): the index is separable -- a row take, then a column take
        rows, cols = y0 + dy, x0 + dx
        return gx[rows][:, cols] * (fx - dx) + gy[rows][:, cols] * (fy - dy)

    def fade(t):
        return t * t * t * (t * (t * 6 - 15) + 10)
    u, v = fade(fx), fade(fy)
    top = corner(0, 0) * (1 - u) + corner(1, 0) * u
    bot = corner(0, 1) * (1 - u) + corner(1, 1) * u
    return (top * (1 - v) + bot * v) * math.sqrt(2.0)


class _MaskCanvas:
    """The working image of one simulated mask (WassersteinGAN.py:405-417,519-533): the requested size plus a margin of 3 d, with
    d = the diagonal of the largest (scaled) particle, so that particles may hang over the edges of the final crop.  Positions are
    drawn over (W, H) = size + 2 d; the smooth noise field covers size + 3 d and is indexed [x, y] like the reference's."""

    def __init__(self, particle_h, particle_w, img_height, img_width, max_scaling):
        self.ph, self.pw, self.img_h, self.img_w = particle_h, particle_w, img_height, img_width
        self.d = math.ceil(math.sqrt((max_scaling * particle_h) ** 2 + (max_scaling * particle_w) ** 2))
        self.H, self.W = img_height + 2 * self.d, img_width + 2 * self.d
        self.img = np.zeros((img_height + 3 * self.d, img_width + 3 * self.d), dtype='uint8')

    def clear(self):
        self.img[:] = 0

    def noise_field(self, frequency):
        """Band-limited noise in [-1, 1], shape (img_w + 3 d, img_h + 3 d) (opensimplex.noise2array(x=iy, y=ix): WassersteinGAN.py:419-424)."""
        ix = np.arange(0, frequency, frequency / (self.img_w + 3 * self.d))
        iy = np.arange(0, frequency, frequency / (self.img_h + 3 * self.d))
        f = _gradient_noise2array(ix, iy, np.random.default_rng(np.random.randint(0, 2 ** 31 - 1)))
        f -= np.min(f)
        f /= np.max(f) / 2
        return f - 1

    def grid_positions(self, kind, spacing_factor, jitter_factor):
        """Hexagonal (every other row shifted by half a cell) or cubic lattice over (W, H), jittered and clipped (WassersteinGAN.py:426-458)."""
        sx, sy = int(spacing_factor * self.pw), int(spacing_factor * self.ph)
        if kind == 'HEXAGONAL':
            shift = int(spacing_factor * self.pw / 2)
            pts = []
            for k, y in enumerate(range(0, self.H, sy)):
                for x in range(0, self.W, sx):
                    if x + k % 2 * shift > self.W:
                        break
                    pts.append((x + k % 2 * shift, y))
            # the reference pre-sizes its arrays (rows x columns + 1): the unused tail stays at the origin and takes part in what follows
            size = math.ceil(self.H / (spacing_factor * self.ph)) * math.ceil(self.W / (spacing_factor * self.pw)) + 1
            pos = np.zeros((max(size, len(pts)), 2), dtype='int32')
            if pts:
                pos[:len(pts)] = np.asarray(pts, dtype='int32')
            pos_x, pos_y = pos[:, 0], pos[:, 1]
        else:
            pos_y, pos_x = np.mgrid[0:self.H:sy, 0:self.W:sy]          # sic (WassersteinGAN.py:452): both spacings from the particle height
            pos_x, pos_y = pos_x.flatten(), pos_y.flatten()
        pos_x = pos_x + np.random.randint(int(-jitter_factor * self.pw), int(jitter_factor * self.pw), pos_x.size)
        pos_y = pos_y + np.random.randint(int(-jitter_factor * self.ph), int(jitter_factor * self.ph), pos_y.size)
        return np.clip(pos_x, 0, self.W), np.clip(pos_y, 0, self.H)

    @staticmethod
    def positions_from_field(noise, level, count):
        """``count`` distinct positions, uniformly among the cells where the field exceeds ``level`` (WassersteinGAN.py:461-468)."""
        weight = (noise > level).astype('float32')
        weight /= np.sum(weight)
        values = np.random.choice(noise.ravel(), count, replace=False, p=weight.ravel())
        cells = np.asarray(np.nonzero(np.isin(noise, values))).transpose()
        np.random.shuffle(cells)
        pos_x, pos_y = cells.transpose()
        return pos_x, pos_y

    def place(self, particle, y0, x0, rotation, scaling, max_overlap):
        """Rotate / scale the particle about its centre into its bounding box, clean it (threshold, fill holes, 9x9 opening), and put
        it at (y0, x0): pixels under its footprint are cleared, its 2-pixel erosion is set -- a dark rim separates neighbours.
        Skipped when the erosion overlaps what is already there by more than ``max_overlap`` of its area (WassersteinGAN.py:500-526)."""
        from scipy import ndimage
        height, width = particle.shape
        centre = (width / 2, height / 2)
        m = _rotation_matrix_2d(centre, rotation, scaling)
        abs_cos, abs_sin = abs(m[0, 0]), abs(m[0, 1])
        bound_w = int(width * abs_sin + height * abs_cos)
        bound_h = int(width * abs_cos + height * abs_sin)
        m[0, 2] += bound_h / 2 - centre[0]
        m[1, 2] += bound_w / 2 - centre[1]
        shape = _warp_affine(particle, m, (bound_h, bound_w)) > 127
        # 9 x 9 opening as its separable form (a square's erosion / dilation is the row pass followed by the column pass, also at the
        # zero border): the same pixels at 18 instead of 81 structure elements per pixel and pass
        shape = ndimage.binary_fill_holes(shape)
        for op in (ndimage.binary_erosion, ndimage.binary_dilation):
            shape = op(op(shape, structure=_ROW9), structure=_COL9)
        core = ndimage.binary_erosion(shape, iterations=2)
        if not np.any(core):
            return False
        win = self.img[y0:y0 + shape.shape[0], x0:x0 + shape.shape[1]]
        if max_overlap is not None and np.sum(np.logical_and(win, core).astype('int32')) > max_overlap * np.sum(core.astype('uint8')):
            return False
        win -= np.logical_and(win, shape).astype('uint8')
        win += core.astype('uint8')
        return True

    def crop(self):
        a, b = int((self.img.shape[0] - self.img_h) / 2), int((self.img.shape[1] - self.img_w) / 2)
        return self.img[a:a + self.img_h, b:b + self.img_w]


def _place_job(job):
    """One simulated mask from its drawn ingredients (worker process of WGAN.simulate_masks): the placement consumes no random
    numbers, so the masks are the ones the sequential loop would write."""
    from PIL import Image
    ctor, particles, pos_y, pos_x, rotations, scalings, max_overlap, path = job
    canvas = _MaskCanvas(*ctor)
    for j in range(len(particles)):
        canvas.place(particles[j], int(pos_y[j]), int(pos_x[j]), rotations[j], scalings[j], max_overlap)
    Image.fromarray(canvas.crop() * 255).save(path)
    return path


class WGAN:
    """Workflow class of step 1 (WassersteinGAN.py:288-545, 683-724): same constructor, attributes and on-disk contract
    (``Input_Masks`` -> ``1_WGAN/{Models,Output_Images}/<timestamp>`` -> simulated masks in ``2_CycleGAN/data/trainB``)."""

    def __init__(self, root_dir, allow_memory_growth=True, use_gpus_no=(0, )):
        from . import HelperFunctions
        self.root_dir = os.path.join(root_dir, '1_WGAN')
        self.input_dir = os.path.join(root_dir, 'Input_Masks')
        self.output_dir = os.path.join(self.root_dir, 'Output_Images')
        self.model_dir = os.path.join(self.root_dir, 'Models')
        self.generate_dir = os.path.join(root_dir, '2_CycleGAN', 'data', 'trainB')
        self.batch_size, self.epochs, self.n_z, self.model = 64, 1000, 128, None
        self.allow_memory_growth, self.use_gpus_no = allow_memory_growth, use_gpus_no
        D.init_from_env()
        self.device = D.local_device()

        self.train_images = []
        max_image_height = max_image_width = 0
        images = HelperFunctions.load_and_preprocess_images(input_dir_or_filelist=self.input_dir, threshold_value=0.5,
                                                            normalization_range=(-1, 1), output_channels=1, contrast_optimization_range=None)
        for image in images:
            max_image_height = max([max_image_height, image.shape[0]])
            max_image_width = max([max_image_height, image.shape[1]])          # sic (WassersteinGAN.py:343): the running HEIGHT
            self.train_images += [image.copy(), np.fliplr(image.copy()), np.flipud(image.copy()), np.flipud(np.fliplr(image.copy()))]
        # sizes that can be halved four times (zero padding, centred)
        if max_image_height % 2 ** 4 != 0:
            max_image_height = (max_image_height // (2 ** 4) + 1) * 2 ** 4
        if max_image_width % 2 ** 4 != 0:
            max_image_width = (max_image_width // (2 ** 4) + 1) * 2 ** 4
        for i, image in enumerate(self.train_images):
            if image.shape[0] < max_image_height or image.shape[1] < max_image_width:
                img = np.zeros((max_image_height, max_image_width, 1), dtype='float32')
                t, l = (max_image_height - image.shape[0]) // 2, (max_image_width - image.shape[1]) // 2
                img[t:t + image.shape[0], l:l + image.shape[1], :] = image[:, :, :]
                self.train_images[i] = img
        self.train_images = np.asarray(self.train_images, dtype='float32')
        self.prefix = time.strftime('%Y-%m-%d_%H-%M-%S', time.localtime())

    # ---- networks / model (WassersteinGAN.py:569-724) -----------------------------------------------------------------------------
    def get_discriminator_model(self):
        return WganCritic(self.train_images.shape[1], self.train_images.shape[2], self.train_images.shape[3], device=self.device)

    def get_generator_model(self):
        return WganGenerator(self.train_images.shape[1], self.train_images.shape[2], self.n_z, device=self.device)

    discriminator_loss = staticmethod(discriminator_loss)
    generator_loss = staticmethod(generator_loss)

    def create_model(self):
        model = WGAN_GP(discriminator=self.get_discriminator_model(), generator=self.get_generator_model(), latent_dim=self.n_z,
                        discriminator_extra_steps=3)
        model.compile(d_optimizer=Adam(learning_rate=0.0002, beta_1=0.5, beta_2=0.9), g_optimizer=Adam(learning_rate=0.0002, beta_1=0.5, beta_2=0.9),
                      g_loss_fn=self.generator_loss, d_loss_fn=self.discriminator_loss)
        return model

    def start_training(self):
        """``model.fit(train_images, batch_size, epochs, callbacks=[GANMonitor(every 20 epochs), CSVLogger])`` + ``model.save``
        (WassersteinGAN.py:366-380): per epoch a fresh permutation (Keras shuffles array inputs), batches of ``batch_size`` with a
        ragged last batch, metrics = running means over the epoch, one CSV row per epoch (Keras CSVLogger: ',' and sorted keys)."""
        os.makedirs(os.path.join(self.model_dir, self.prefix), exist_ok=True)
        os.makedirs(os.path.join(self.output_dir, self.prefix), exist_ok=True)
        from . import keras_io
        keras_io.warn_if_no_hdf5('WGAN.start_training')          # the only save happens after the last epoch: say so NOW
        self.model = self.create_model()
        cbk = GANMonitor(output_dir=os.path.join(self.output_dir, self.prefix), num_img=9, latent_dim=self.n_z, output_epochs=20)
        log_path = os.path.join(self.model_dir, self.prefix, 'training_log.csv')
        rank, world = D.rank(), D.world_size()
        n = len(self.train_images)
        if world > 1:
            # data parallel: every rank must cut its shard out of the SAME permutation (one RandomState seeded by rank 0's draw),
            # the generator's BatchNorm needs whole-global-batch statistics as in the single-device reference, and a global batch
            # that does not split evenly is an error, not silently dropped samples
            D.check_batch_divisible(self.batch_size, world, 'WGAN.batch_size')
            D.enable_sync_bn(True)
            seed = torch.tensor([np.random.randint(0, 2 ** 31 - 1)], dtype=torch.int64, device=self.device)
            torch.distributed.broadcast(seed, 0)
            perm_rng = np.random.RandomState(int(seed.item()))
        else:
            perm_rng = np.random
        for epoch in range(self.epochs):
            self.model.reset_metrics()
            order = perm_rng.permutation(n)
            logs = {}
            for start in range(0, n, self.batch_size):
                idx = order[start:start + self.batch_size]
                if world > 1:
                    per = len(idx) // world
                    if per * world != len(idx):
                        D.warn_once(f'WGAN: the last batch of an epoch ({len(idx)} samples) does not split over {world} ranks: '
                                    f'{len(idx) - per * world} sample(s) per epoch are skipped')
                    if per == 0:
                        continue
                    idx = idx[rank * per:(rank + 1) * per]
                logs = self.model.train_step(self.train_images[idx])
            if rank == 0:
                new = not os.path.exists(log_path)
                with open(log_path, 'a') as f:
                    if new:
                        f.write(','.join(['epoch'] + sorted(logs)) + '\n')
                    f.write(','.join([str(epoch)] + [repr(logs[k]) for k in sorted(logs)]) + '\n')
                cbk.on_epoch_end(self.model, epoch, logs)
        if rank == 0:
            self.save_model(os.path.join(self.model_dir, self.prefix, 'model.keras'))
        return self.model

    # ---- checkpoint exchange --------------------------------------------------------------------------------------------------
    def save_model(self, path):
        """``model.save(path)``: .keras archive (generator/ + discriminator/ + both Adam states; keras_io.py) or .npz."""
        from . import keras_io
        m = self.model
        if path.endswith('.npz'):
            arrays = {f"generator/{n_}": w for n_, w in zip(m.generator.variable_names, m.generator.get_weights())}
            arrays.update({f"discriminator/{n_}": w for n_, w in zip(m.discriminator.variable_names, m.discriminator.get_weights())})
            np.savez(path, **arrays)
            return
        arrays = keras_io.net_arrays(m.discriminator, "discriminator/")
        arrays.update(keras_io.net_arrays(m.generator, "generator/"))
        arrays.update(keras_io.optimizer_arrays(m.d_optimizer, m.discriminator, "d_optimizer/"))
        arrays.update(keras_io.optimizer_arrays(m.g_optimizer, m.generator, "g_optimizer/"))
        keras_io.write_archive(path, arrays, "WGAN_GP", dict(latent_dim=self.n_z, image_shape=list(self.train_images.shape[1:])))

    def load_model(self, path):
        from . import keras_io
        if self.model is None:
            self.model = self.create_model()
        m = self.model
        if path.endswith('.npz'):
            z = np.load(path)
            m.generator.set_weights([z[f"generator/{n_}"] for n_ in m.generator.variable_names])
            m.discriminator.set_weights([z[f"discriminator/{n_}"] for n_ in m.discriminator.variable_names])
            return m
        _cls, _cfg, arrays = keras_io.read_archive(path)
        legacy = keras_io.NameCounters()
        keras_io.load_net_arrays(m.discriminator, "discriminator/", arrays, legacy)
        keras_io.load_net_arrays(m.generator, "generator/", arrays, legacy)
        keras_io.load_optimizer_arrays(m.d_optimizer, m.discriminator, "d_optimizer/", arrays)
        keras_io.load_optimizer_arrays(m.g_optimizer, m.generator, "g_optimizer/", arrays)
        return m

    # ---- mask simulation (WassersteinGAN.py:382-545) ------------------------------------------------------------------------------
    def _sample_particles(self, count):
        """``count`` generated particle masks as uint8 images (WassersteinGAN.py:487-503: batches of ``batch_size`` latent vectors,
        inference mode)."""
        out = np.zeros((count, self.train_images.shape[1], self.train_images.shape[2]), dtype='uint8')
        if count == 0:
            return out
        # the latent vectors are drawn ``batch_size`` rows at a time (the stream of the per-batch loop); the generator then runs on
        # ``sample_chunk`` of them per call: a mask needs ~3 000 particles, and 47 calls of 64 with a device->host read each made the
        # drawing loop (not the placement workers) the pace of step 2.  Inference mode: a particle does not depend on its batch.
        z = torch.cat([torch.randn((min(self.batch_size, count - j), self.n_z), device=self.device) for j in range(0, count, self.batch_size)])
        chunk = max(int(getattr(self, "sample_chunk", 512)), 1)
        for j in range(0, count, chunk):
            y = self.model(z[j:j + chunk], training=False)
            y = y.dense() if isinstance(y, Act) else y
            if isinstance(y, torch.Tensor):
                # grey levels on the device: float32 y * 127.5 + 127.5 truncated to uint8, the arithmetic of the host form below, and a
                # quarter of the bytes to read back (a mask's particles are 49 MB as float32)
                out[j:j + chunk] = (y.detach().float() * 127.5 + 127.5).to(torch.uint8)[:, :, :, 0].cpu().numpy()
            else:
                out[j:j + chunk] = (np.asarray(y, dtype='float32') * 127.5 + 127.5)[:, :, :, 0].astype('uint8')
        return out

    def simulate_masks(self, no_of_images=1, min_no_of_particles=100, max_no_of_particles=150, use_normal_distribution=False, sigma=0.10,
                       mu=1.0, min_scaling=0.75, max_scaling=1.25, use_perlin_noise=True, perlin_noise_threshold=0.5,
                       perlin_noise_frequency=4, use_random_rotation='DISABLE', max_overlap=0.01, grid_type='DISABLE',
                       grid_spacing_factor=0.125, grid_noise_factor=0.05, img_width=384, img_height=384):
        """Step 2 of the workflow (WassersteinGAN.py:382-545): ``no_of_images`` binary masks of generated particles.  Per mask: where
        (grid or free placement, optionally thinned / weighted by a smooth noise field), how large and how rotated every particle
        is, then the particles one by one onto a canvas with a margin, rejecting those that overlap too much."""
        from PIL import Image
        from shutil import copy
        if self.model is None:
            self.load_model(os.path.join(self.model_dir, os.listdir(self.model_dir)[-1], 'model.keras'))
        if use_normal_distribution:
            min_scaling, max_scaling_used = mu - 3 * sigma, mu + 3 * sigma
        else:
            max_scaling_used = max_scaling
        if max_overlap is not None and grid_type not in ('HEXAGONAL', 'CUBIC'):
            grid_type = 'HEXAGONAL'          # overlap control only exists on the grid paths (WassersteinGAN.py:407-408)
        ctor = (self.train_images.shape[1], self.train_images.shape[2], img_height, img_width, max_scaling)
        canvas = _MaskCanvas(*ctor)
        os.makedirs(self.generate_dir, exist_ok=True)
        # Where / how large / how rotated is drawn here, in the reference's order, and the particles come from the generator on the
        # device; putting ~3 000 particles onto a canvas (affine warp, hole filling, opening, erosion, overlap test per particle) is
        # 1 - 5 s of pure host work per mask and draws nothing: it runs in worker processes (SS_MASK_WORKERS; default: the usable cores --
        # affinity and cgroup quota -- less one; 1 = inline), so 1 000 masks take a minute instead of the better part of an hour.
        from . import HelperFunctions
        workers = int(os.environ.get("SS_MASK_WORKERS", HelperFunctions.default_workers()))
        # (HelperFunctions.JobPool: a worker process that dies is noticed and its masks are placed inline instead of hanging the step)
        pool = HelperFunctions.JobPool(_place_job, workers if (workers > 1 and no_of_images >= 4) else 1)
        try:
            for i in range(no_of_images):
                count = random.randint(min_no_of_particles, max_no_of_particles) if grid_type not in ('HEXAGONAL', 'CUBIC') else 0
                noise = canvas.noise_field(perlin_noise_frequency) if (use_perlin_noise or use_random_rotation == 'PERLIN') else None
                level = 2 * perlin_noise_threshold - 1
                if grid_type in ('HEXAGONAL', 'CUBIC'):
                    pos_x, pos_y = canvas.grid_positions(grid_type, grid_spacing_factor, grid_noise_factor)
                    if use_perlin_noise:          # thin the grid where the field is low (indexed [x, y], as the reference does)
                        keep = [q for q in range(len(pos_x)) if noise[pos_x[q], pos_y[q]] > level]
                        pos_x, pos_y = [pos_x[q] for q in keep], [pos_y[q] for q in keep]
                    count = len(pos_x)
                elif use_perlin_noise:
                    pos_x, pos_y = canvas.positions_from_field(noise, level, count)
                else:
                    pos_x, pos_y = np.random.randint(0, canvas.W, count), np.random.randint(0, canvas.H, count)
                scalings = np.random.normal(mu, sigma, count) if use_normal_distribution else np.random.uniform(min_scaling, max_scaling_used, count)
                scalings = np.clip(scalings, min_scaling, max_scaling_used)
                if use_random_rotation == 'RANDOM':
                    rotations = np.random.randint(0, 360, count)
                elif use_random_rotation == 'PERLIN':
                    rotations = noise[np.asarray(pos_y, dtype=int), np.asarray(pos_x, dtype=int)] * 180
                else:
                    rotations = np.zeros(count)
                job = (ctor, self._sample_particles(count), np.asarray(pos_y), np.asarray(pos_x), np.asarray(rotations), np.asarray(scalings),
                       max_overlap, os.path.join(self.generate_dir, '{:05d}.tif'.format(i)))
                pool.submit(job, in_flight=3 * workers)          # bound what is in flight (a job carries ~12 MB of particles)
        finally:
            pool.close()

        # five random files for testing (WassersteinGAN.py:539-545)
        input_files = [f for f in os.listdir(self.generate_dir) if '.tif' in f or '.png' in f or '.bmp' in f]
        output_dir = os.path.join(self.generate_dir, '..', 'testB')
        os.makedirs(output_dir, exist_ok=True)
        for f in random.sample(input_files, min(5, len(input_files))):
            copy(os.path.join(self.generate_dir, f), output_dir)
