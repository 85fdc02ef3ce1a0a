"""Oracle network topologies (test infrastructure; see oracle/__init__.py).

Restates ``CycleGAN.get_resnet_generator`` (CycleGAN.py:360-423 with
``residual_block`` :323-337, ``downsample`` :339-345, ``upsample`` :347-358),
``CycleGAN.get_discriminator`` (CycleGAN.py:425-451) and
``UNet.multi_res_unet`` (UNet_Segmentation.py:401-562) on top of oracle/ops.py.

Weights are kept in Keras variable layout and creation order, so that
``get_weights()`` lines up with ``keras.Model.get_weights()``.
"""
import math
import torch

from . import ops


class Var:
    """Mimics a Keras variable wrapper: ``.value`` is the torch tensor (CycleGAN.py:668)."""

    def __init__(self, name, value, trainable=True):
        self.name = name
        self.value = value
        self.trainable = trainable

    @property
    def shape(self):
        return tuple(self.value.shape)


def glorot_uniform(shape, gen, dtype):
    """keras GlorotUniform: limit = sqrt(6 / (fan_in + fan_out)); fans from the kernel shape."""
    rf = 1
    for d in shape[:-2]:
        rf *= d
    fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    limit = math.sqrt(6.0 / (fan_in + fan_out))
    u = torch.rand(shape, generator=gen, dtype=torch.float64)
    return ((u * 2.0 - 1.0) * limit).to(dtype)


class Net:
    """Base: ordered list of Var; callable as ``net(x, training=True)`` on NHWC tensors."""

    def __init__(self, dtype=torch.float32, seed=0):
        self.dtype = dtype
        self.gen = torch.Generator().manual_seed(seed)
        self.variables = []
        self._by_name = {}

    # -- variable creation ------------------------------------------------
    def _add(self, name, value, trainable=True):
        value = value.to(self.dtype)
        if trainable:
            value.requires_grad_(True)
        v = Var(name, value, trainable)
        self.variables.append(v)
        self._by_name[name] = v
        return v

    def add_kernel(self, name, shape):
        return self._add(name, glorot_uniform(shape, self.gen, self.dtype))

    def add_zeros(self, name, shape, trainable=True):
        return self._add(name, torch.zeros(shape), trainable)

    def add_ones(self, name, shape, trainable=True):
        return self._add(name, torch.ones(shape), trainable)

    def p(self, name):
        return self._by_name[name].value

    # -- Keras-like surface -----------------------------------------------
    @property
    def trainable_weights(self):
        return [v for v in self.variables if v.trainable]

    @property
    def non_trainable_weights(self):
        return [v for v in self.variables if not v.trainable]

    def get_weights(self):
        return [v.value.detach().cpu().numpy().copy() for v in self.variables]

    def set_weights(self, arrays):
        assert len(arrays) == len(self.variables)
        with torch.no_grad():
            for v, a in zip(self.variables, arrays):
                v.value.copy_(torch.as_tensor(a, dtype=self.dtype).reshape(v.value.shape))

    def zero_grad(self):
        for v in self.variables:
            v.value.grad = None

    def __call__(self, x, training=True):
        return self.forward(x, training)


class ResnetGenerator(Net):
    """CycleGAN.get_resnet_generator, default branch (no skip connection, no resize-conv, tanh)."""

    def __init__(self, filters=64, num_downsampling_blocks=3, num_residual_blocks=9,
                 num_upsample_blocks=3, channels=1, dtype=torch.float32, seed=0, use_skip_connection=False,
                 use_resize_convolution=False, sigmoid_output=False):
        super().__init__(dtype, seed)
        self.nd, self.nr, self.nu = num_downsampling_blocks, num_residual_blocks, num_upsample_blocks
        self.skip, self.resize, self.sigmoid_output = use_skip_connection, use_resize_convolution, sigmoid_output
        f = filters
        self.add_kernel("c7_in/kernel", (7, 7, channels, f))
        self._gn("c7_in", f)
        for i in range(self.nd):
            self.add_kernel(f"down{i}/kernel", (3, 3, f, 2 * f))
            f *= 2
            self._gn(f"down{i}", f)
        for i in range(self.nr):
            for j in range(2):
                self.add_kernel(f"res{i}.{j}/kernel", (3, 3, f, f))
                self._gn(f"res{i}.{j}", f)
        for i in range(self.nu):
            if use_resize_convolution:
                self.add_kernel(f"up{i}/kernel", (3, 3, f, f // 2))  # CycleGAN.py:351 plain Conv2D
            else:
                self.add_kernel(f"up{i}/kernel", (3, 3, f // 2, f))  # Conv2DTranspose: (kh,kw,out,in)
            f //= 2
            self._gn(f"up{i}", f)
        self.add_kernel("c7_out/kernel", (7, 7, f, channels))
        self.add_zeros("c7_out/bias", (channels,))
        if use_skip_connection:      # CycleGAN.py:396-415, creation order
            self.add_kernel("skip.sc1x1/kernel", (1, 1, channels, f))
            self._gn("skip.sc1x1", f)
            self.add_kernel("skip.3/kernel", (3, 3, channels, f))
            self._gn("skip.3", f)
            self._gn("skip.sum", f)
            self.add_kernel("skip.out1x1/kernel", (1, 1, f + channels, channels))

    def _gn(self, name, c):
        self.add_ones(f"{name}/gamma", (c,))
        self.add_zeros(f"{name}/beta", (c,))

    def _in(self, name, x):
        return ops.instance_norm(x, self.p(f"{name}/gamma"), self.p(f"{name}/beta"))

    def forward(self, x, training=True):
        m = 2 ** self.nd
        ph = (m - x.shape[1] % m) % m
        pw = (m - x.shape[2] % m) % m
        img_input = x
        x = ops.reflection_pad(x, (pw, ph))
        x = ops.reflection_pad(x, (6, 6))
        x = torch.relu(self._in("c7_in", ops.conv2d(x, self.p("c7_in/kernel"))))
        for i in range(self.nd):
            x = ops.conv2d(x, self.p(f"down{i}/kernel"), stride=2, padding="same")
            x = torch.relu(self._in(f"down{i}", x))
        for i in range(self.nr):
            y = ops.reflection_pad(x, (2, 2))
            y = torch.relu(self._in(f"res{i}.0", ops.conv2d(y, self.p(f"res{i}.0/kernel"))))
            y = ops.reflection_pad(y, (2, 2))
            y = self._in(f"res{i}.1", ops.conv2d(y, self.p(f"res{i}.1/kernel")))
            x = x + y
        for i in range(self.nu):
            if self.resize:
                x = x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)     # UpSampling2D nearest
                x = ops.conv2d(ops.reflection_pad(x, (2, 2)), self.p(f"up{i}/kernel"))
            else:
                x = ops.conv2d_transpose(x, self.p(f"up{i}/kernel"), stride=2)
            x = torch.relu(self._in(f"up{i}", x))
        x = ops.reflection_pad(x, (6, 6))
        x = ops.conv2d(x, self.p("c7_out/kernel"), self.p("c7_out/bias"))
        if self.skip:
            sc = torch.relu(self._in("skip.sc1x1", ops.conv2d(img_input, self.p("skip.sc1x1/kernel"))))
            out = torch.relu(self._in("skip.3", ops.conv2d(ops.reflection_pad(img_input, (2, 2)), self.p("skip.3/kernel"))))
            out = torch.relu(self._in("skip.sum", sc + out))
            x = ops.conv2d(torch.cat([out, x], dim=3), self.p("skip.out1x1/kernel"))
        return torch.sigmoid(x) if self.sigmoid_output else torch.tanh(x)


class PatchDiscriminator(Net):
    """CycleGAN.get_discriminator with gaussian_noise_value == 0 (StartProcess.py:96)."""

    def __init__(self, filters=128, num_downsampling_blocks=2, channels=1, padding="valid",
                 dtype=torch.float32, seed=0):
        super().__init__(dtype, seed)
        self.nd, self.padding = num_downsampling_blocks, padding
        f = filters
        self.add_kernel("c4_in/kernel", (4, 4, channels, f))
        self.add_zeros("c4_in/bias", (f,))
        for i in range(self.nd):
            self.add_kernel(f"down{i}/kernel", (4, 4, f, 2 * f))
            f *= 2
            self.add_ones(f"down{i}/gamma", (f,))
            self.add_zeros(f"down{i}/beta", (f,))
        self.add_kernel("c4_out/kernel", (4, 4, f, 1))
        self.add_zeros("c4_out/bias", (1,))

    def forward(self, x, training=True):
        x = ops.conv2d(x, self.p("c4_in/kernel"), self.p("c4_in/bias"), stride=2, padding=self.padding)
        x = ops.leaky_relu(x, 0.2)
        for i in range(self.nd):
            s = 2 if i < 3 else 1
            x = ops.conv2d(x, self.p(f"down{i}/kernel"), stride=s, padding=self.padding)
            x = ops.instance_norm(x, self.p(f"down{i}/gamma"), self.p(f"down{i}/beta"))
            x = ops.leaky_relu(x, 0.2)
        return ops.conv2d(x, self.p("c4_out/kernel"), self.p("c4_out/bias"), stride=1, padding=self.padding)


class MultiResUNet(Net):
    """UNet.multi_res_unet (UNet_Segmentation.py:505-562): output_channels == 1 -> conv2d_bn(1, 1x1, sigmoid) head (:556-557),
    otherwise Conv2D(output_channels, 1x1, bias) + softmax over the channels (:558-560)."""

    ALPHA = 1.67

    def __init__(self, conv_filters=16, dtype=torch.float32, seed=0, output_channels=1):
        super().__init__(dtype, seed)
        self.output_channels = output_channels
        self.filters = f = conv_filters
        self._n = 0
        cin = 1
        c1 = self._mrb_make("mrb1", f, cin)
        self._rp_make("rp1", f, 4, c1)
        c2 = self._mrb_make("mrb2", f * 2, c1)
        self._rp_make("rp2", f * 2, 3, c2)
        c3 = self._mrb_make("mrb3", f * 4, c2)
        self._rp_make("rp3", f * 4, 2, c3)
        c4 = self._mrb_make("mrb4", f * 8, c3)
        self._rp_make("rp4", f * 8, 1, c4)
        c5 = self._mrb_make("mrb5", f * 16, c4)
        self._upT_make("up6T", c5, f * 8)
        c6 = self._mrb_make("mrb6", 32 * 8, f * 8 + f * 8)
        self._upT_make("up7T", c6, f * 4)
        c7 = self._mrb_make("mrb7", 32 * 4, f * 4 + f * 4)
        self._upT_make("up8T", c7, f * 2)
        c8 = self._mrb_make("mrb8", 32 * 2, f * 2 + f * 2)
        self._upT_make("up9T", c8, f)
        c9 = self._mrb_make("mrb9", f, f + f)
        if output_channels == 1:
            self._cbn_make("out1x1", 1, c9, 1)
        else:
            self.add_kernel("out1x1/kernel", (1, 1, c9, output_channels))
            self.add_zeros("out1x1/bias", (output_channels,))

    # ---- parameter construction (Keras creation order) ------------------
    @classmethod
    def widths(cls, u):
        w = cls.ALPHA * u
        return int(w * 0.167), int(w * 0.333), int(w * 0.5)

    def _bn_make(self, name, c, scale):
        if scale:
            self.add_ones(f"{name}/gamma", (c,))
        self.add_zeros(f"{name}/beta", (c,))
        self.add_zeros(f"{name}/moving_mean", (c,), trainable=False)
        self.add_ones(f"{name}/moving_variance", (c,), trainable=False)

    def _cbn_make(self, name, k, cin, cout):
        self.add_kernel(f"{name}/kernel", (k, k, cin, cout))
        self._bn_make(f"{name}/bn", cout, scale=False)

    def _mrb_make(self, name, u, cin):
        a, b, c = self.widths(u)
        self._cbn_make(f"{name}.sc1x1", 1, cin, a + b + c)
        self._cbn_make(f"{name}.3", 3, cin, a)
        self._cbn_make(f"{name}.5", 3, a, b)
        self._cbn_make(f"{name}.7", 3, b, c)
        self._bn_make(f"{name}.bn_a", a + b + c, scale=True)
        self._bn_make(f"{name}.bn_b", a + b + c, scale=True)
        return a + b + c

    def _rp_make(self, name, filters, length, cin):
        for i in range(length):
            self._cbn_make(f"{name}.{i}.sc", 1, cin, filters)
            self._cbn_make(f"{name}.{i}.3", 3, cin, filters)
            self._bn_make(f"{name}.{i}.bn", filters, scale=True)
            cin = filters

    def _upT_make(self, name, cin, cout):
        self.add_kernel(f"{name}/kernel", (2, 2, cout, cin))
        self.add_zeros(f"{name}/bias", (cout,))

    # ---- forward --------------------------------------------------------
    def _bn(self, name, x, training, scale):
        gamma = self.p(f"{name}/gamma") if scale else None
        mm, mv = self._by_name[f"{name}/moving_mean"], self._by_name[f"{name}/moving_variance"]
        y, nmm, nmv = ops.batch_norm(x, gamma, self.p(f"{name}/beta"), mm.value, mv.value, training)
        if training:
            mm.value, mv.value = nmm.detach(), nmv.detach()
        return y

    def _cbn(self, name, x, training, act):
        x = ops.conv2d(x, self.p(f"{name}/kernel"), padding="same")
        x = self._bn(f"{name}/bn", x, training, scale=False)
        if act == "relu":
            x = torch.relu(x)
        elif act == "sigmoid":
            x = torch.sigmoid(x)
        return x

    def _mrb(self, name, x, t):
        sc = self._cbn(f"{name}.sc1x1", x, t, None)
        a = self._cbn(f"{name}.3", x, t, "relu")
        b = self._cbn(f"{name}.5", a, t, "relu")
        c = self._cbn(f"{name}.7", b, t, "relu")
        out = torch.cat([a, b, c], dim=3)
        out = self._bn(f"{name}.bn_a", out, t, scale=True)
        out = torch.relu(sc + out)
        return self._bn(f"{name}.bn_b", out, t, scale=True)

    def _rp(self, name, length, x, t):
        for i in range(length):
            sc = self._cbn(f"{name}.{i}.sc", x, t, None)
            out = self._cbn(f"{name}.{i}.3", x, t, "relu")
            x = self._bn(f"{name}.{i}.bn", torch.relu(sc + out), t, scale=True)
        return x

    def _upT(self, name, x):
        return ops.conv2d_transpose(x, self.p(f"{name}/kernel"), self.p(f"{name}/bias"), stride=2)

    def forward(self, x, training=True):
        t = training
        ph = (16 - x.shape[1] % 16) % 16
        pw = (16 - x.shape[2] % 16) % 16
        x = ops.reflection_pad(x, (pw, ph))
        m1 = self._mrb("mrb1", x, t)
        p1 = ops.max_pool2x2(m1)
        m1 = self._rp("rp1", 4, m1, t)
        m2 = self._mrb("mrb2", p1, t)
        p2 = ops.max_pool2x2(m2)
        m2 = self._rp("rp2", 3, m2, t)
        m3 = self._mrb("mrb3", p2, t)
        p3 = ops.max_pool2x2(m3)
        m3 = self._rp("rp3", 2, m3, t)
        m4 = self._mrb("mrb4", p3, t)
        p4 = ops.max_pool2x2(m4)
        m4 = self._rp("rp4", 1, m4, t)
        m5 = self._mrb("mrb5", p4, t)
        m6 = self._mrb("mrb6", torch.cat([self._upT("up6T", m5), m4], dim=3), t)
        m7 = self._mrb("mrb7", torch.cat([self._upT("up7T", m6), m3], dim=3), t)
        m8 = self._mrb("mrb8", torch.cat([self._upT("up8T", m7), m2], dim=3), t)
        m9 = self._mrb("mrb9", torch.cat([self._upT("up9T", m8), m1], dim=3), t)
        H, W = m9.shape[1], m9.shape[2]
        m9 = m9[:, ph // 2: H - (ph // 2 + ph % 2), pw // 2: W - (pw // 2 + pw % 2), :]
        if self.output_channels == 1:
            return self._cbn("out1x1", m9, t, "sigmoid")
        return torch.softmax(ops.conv2d(m9, self.p("out1x1/kernel"), self.p("out1x1/bias"), stride=1, padding="valid"), dim=-1)
