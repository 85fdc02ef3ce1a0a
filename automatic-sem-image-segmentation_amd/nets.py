"""Network topologies of the hot path, composed from layers.py.

* ``ResnetGenerator``   -- CycleGAN.get_resnet_generator, CycleGAN.py:360-423 (blocks :323-358)
* ``PatchDiscriminator``-- CycleGAN.get_discriminator,   CycleGAN.py:425-451
* ``MultiResUNet``      -- UNet.multi_res_unet,          UNet_Segmentation.py:401-562

Variables are declared in Keras creation order with Keras shapes, so ``get_weights()`` /
``set_weights()`` exchange lists with ``keras.Model.get_weights()`` of the reference networks.
"""
import math

import numpy as np
import torch

from . import _lib as L
from .engine import Act, ParamArena, Tape, convert
from . import layers
from .layers import Conv2D, Norm, add, crop, maxpool2x2, reflect_pad, upsample2x


class Network:
    def __init__(self, device, act_dtype=torch.float32):
        self.device = torch.device(device)
        self.arena = ParamArena(self.device)
        self.training_runs = 0
        # storage type of the ACTIVATIONS (and of what is saved for backward): float32 = the reference's precision; bfloat16 /
        # float16 = mixed precision (BASELINE configs 2 and 5): weights, gradients of weights, optimizer state, normalisation
        # statistics and all accumulation stay fp32.  Inputs are converted on entry, outputs are returned in this type.
        self.act_dtype = L.torch_dtype(act_dtype)

    # ---- Keras-like weight surface -------------------------------------------------------------
    def _finish(self, seed):
        self.arena.materialize()
        self.init_weights(seed)

    def init_weights(self, seed=0):
        """Glorot-uniform kernels (keras default / GlorotUniform, CycleGAN.py:125), zeros bias/beta/moving_mean,
        ones gamma/moving_variance.  Keras' stateful SeedGenerator stream is not reproducible (SURVEY K-list 9);
        parity tests always load explicit weights."""
        gen = torch.Generator().manual_seed(seed)
        for name, shape, trainable, _ in self.arena.specs:
            v = self.arena[name]
            if name.endswith("/kernel"):
                rf = int(np.prod(shape[:-2]))
                limit = math.sqrt(6.0 / (shape[-2] * rf + shape[-1] * rf))
                u = torch.rand(shape, generator=gen, dtype=torch.float64)
                v.copy_(((u * 2.0 - 1.0) * limit).to(torch.float32))
            elif name.endswith("/gamma") or name.endswith("/moving_variance"):
                v.fill_(1.0)
            else:
                v.zero_()

    @property
    def variable_names(self):
        return [s[0] for s in self.arena.specs]

    def get_weights(self):
        return [self.arena[name].detach().cpu().numpy().copy() for name, *_ in self.arena.specs]

    def set_weights(self, arrays):
        assert len(arrays) == len(self.arena.specs), (len(arrays), len(self.arena.specs))
        for (name, shape, *_), a in zip(self.arena.specs, arrays):
            self.arena[name].copy_(torch.as_tensor(np.asarray(a), dtype=torch.float32).reshape(shape))

    def get_gradients(self):
        return {name: self.arena.grad(name).detach().cpu().numpy().copy()
                for name, _, trainable, _ in self.arena.specs if trainable}

    def zero_grad(self):
        self.arena.zero_grad()

    def count_params(self):
        return sum(int(np.prod(s[1])) for s in self.arena.specs)

    def __call__(self, x, training=True, tape=None):
        """x: Act or NHWC torch tensor.  Returns the output Act; backward closures go to ``tape``."""
        if not isinstance(x, Act):
            x = Act(x.contiguous(), requires_grad=False)
        x = convert(x, self.act_dtype)
        return self.forward(tape if tape is not None else Tape(enabled=False), x, training)


class ResnetGenerator(Network):
    """CycleGAN.get_resnet_generator (CycleGAN.py:360-423).  Defaults = the StartProcess configuration (no skip
    connection, transposed-conv upsampling, tanh output); the builder's other branches are options:
    ``use_skip_connection`` (CycleGAN.py:396-415), ``use_resize_convolution`` (CycleGAN.py:348-351),
    ``sigmoid_output`` (use_binary_crossentropy generator A, CycleGAN.py:417-418)."""

    def __init__(self, filters=64, num_downsampling_blocks=3, num_residual_blocks=9, num_upsample_blocks=3,
                 channels=1, device="cuda", seed=0, algo=L.ALGO_AUTO, use_skip_connection=False,
                 use_resize_convolution=False, sigmoid_output=False, act_dtype=torch.float32, checkpoint_blocks=False):
        super().__init__(device, act_dtype)
        # activation checkpointing of the residual trunk (BASELINE config 5): forward keeps only every block's INPUT; backward
        # recomputes the block (conv, InstanceNorm, ReLU, conv, InstanceNorm: CycleGAN.py:323-337) before back-propagating through it
        self.checkpoint_blocks = checkpoint_blocks
        A = self.arena
        self.nd, self.nr, self.nu = num_downsampling_blocks, num_residual_blocks, num_upsample_blocks
        self.filters = filters
        self.use_skip_connection, self.use_resize_convolution = use_skip_connection, use_resize_convolution
        self.sigmoid_output = sigmoid_output
        final_act = "sigmoid" if sigmoid_output else "tanh"
        f = filters
        self.c7_in = Conv2D(A, "c7_in", 7, channels, f, padding=("reflect", 3), algo=algo)
        self.in_c7 = Norm(A, "c7_in", f, "instance")
        self.down = []
        for i in range(self.nd):
            conv = Conv2D(A, f"down{i}", 3, f, 2 * f, stride=2, padding="same", algo=algo)
            f *= 2
            self.down.append((conv, Norm(A, f"down{i}", f, "instance")))
        self.res = []
        for i in range(self.nr):
            c0 = Conv2D(A, f"res{i}.0", 3, f, f, padding=("reflect", 1), algo=algo)
            n0 = Norm(A, f"res{i}.0", f, "instance")
            c1 = Conv2D(A, f"res{i}.1", 3, f, f, padding=("reflect", 1), algo=algo)
            n1 = Norm(A, f"res{i}.1", f, "instance")
            self.res.append((c0, n0, c1, n1))
        self.up = []
        for i in range(self.nu):
            if use_resize_convolution:   # UpSampling2D -> ReflectionPadding2D -> Conv2D(3x3, valid)
                conv = Conv2D(A, f"up{i}", 3, f, f // 2, padding=("reflect", 1), algo=algo)
            else:
                conv = Conv2D(A, f"up{i}", 3, f, f // 2, stride=2, transposed=True, algo=algo)
            f //= 2
            self.up.append((conv, Norm(A, f"up{i}", f, "instance")))
        self.c7_out = Conv2D(A, "c7_out", 7, f, channels, padding=("reflect", 3), use_bias=True,
                             act=None if use_skip_connection else final_act, algo=algo)
        if use_skip_connection:
            self.sk_sc = Conv2D(A, "skip.sc1x1", 1, channels, f, algo=algo)
            self.sk_sc_n = Norm(A, "skip.sc1x1", f, "instance")
            self.sk_c3 = Conv2D(A, "skip.3", 3, channels, f, padding=("reflect", 1), algo=algo)
            self.sk_c3_n = Norm(A, "skip.3", f, "instance")
            self.sk_n = Norm(A, "skip.sum", f, "instance")
            self.sk_out = Conv2D(A, "skip.out1x1", 1, f + channels, channels, act=final_act, algo=algo)
        self._finish(seed)

    def forward(self, tape, x, training=True):
        m = 2 ** self.nd
        ph, pw = (m - x.h % m) % m, (m - x.w % m) % m
        img_input = x
        if ph or pw:
            if self.use_skip_connection:
                raise NotImplementedError("skip connection + pre-padding concatenates mismatching sizes in the reference too")
            # CycleGAN.py:365-367: reflect pre-pad to a multiple of 2**n_down; the output is NOT cropped back (CycleGAN.py:394)
            x = reflect_pad(tape, x, pw, ph)
        h = self.in_c7(tape, self.c7_in(tape, x), act="relu")
        for conv, norm in self.down:
            h = norm(tape, conv(tape, h), act="relu")
        for blk in self.res:
            h = self._res_block_ckpt(tape, blk, h) if (self.checkpoint_blocks and tape.enabled) else self._res_block(tape, blk, h)
        for conv, norm in self.up:
            if self.use_resize_convolution:
                h = upsample2x(tape, h)
            h = norm(tape, conv(tape, h), act="relu")
        if not self.use_skip_connection:
            return self.c7_out(tape, h)
        f = self.sk_sc.cout
        cat = x.like(c=f + self.c7_out.cout)       # concatenate([out, x]) without a copy
        self.c7_out(tape, h, out=cat.slice(f, self.c7_out.cout))
        sc = self.sk_sc_n(tape, self.sk_sc(tape, img_input), act="relu")
        o3 = self.sk_c3_n(tape, self.sk_c3(tape, img_input), act="relu")
        self.sk_n(tape, add(tape, sc, o3), act="relu", out=cat.slice(0, f))
        return self.sk_out(tape, cat)


    @staticmethod
    def _res_block(tape, blk, h):
        c0, n0, c1, n1 = blk
        # conv -> InstanceNorm -> relu -> pad -> conv (CycleGAN.py:327-333): the first norm's apply pass is deferred into the second
        # convolution's operand load where that convolution can normalise while loading ("fused InstanceNorm + conv")
        y = n0(tape, c0(tape, h), act="relu", defer_to=c1)
        return n1(tape, c1(tape, y), residual=h)

    def _res_block_ckpt(self, tape, blk, h):
        """Forward without a tape (nothing saved but the block's input and output); the recorded backward closure re-runs the
        block on a local tape and back-propagates through it.  Parameter gradients accumulate into the arena as usual; the
        variables' uses are announced here (the recomputation tape does not count them again), so the overlapped gradient
        exchange still sees a bucket complete only after the last real use."""
        from .engine import Tape as _Tape
        param_grads = tape.param_grads
        out = self._res_block(_Tape(enabled=False), blk, h)
        c0, n0, c1, n1 = blk
        names = [f"{c0.name}/kernel", f"{n0.name}/gamma", f"{n0.name}/beta", f"{c1.name}/kernel", f"{n1.name}/gamma", f"{n1.name}/beta"]
        if param_grads and tape.count_uses:
            self.arena.note_use(names)

        def backward():
            dy = out.get_grad()
            if dy is None:
                return
            inner = _Tape(param_grads=param_grads, count_uses=False)
            h2 = Act(h.t, h.c0, h.c, requires_grad=h.requires_grad)      # same storage, own gradient slot
            out2 = self._res_block(inner, blk, h2)
            out2.grad, out2.grad_init = dy, True                         # the recomputed output receives the saved gradient
            inner.backward()
            if h.requires_grad:
                gh = h2.get_grad()
                if gh is not None:
                    from .layers import add_grad
                    add_grad(h, gh)

        tape.record(backward)
        return out


class PatchDiscriminator(Network):
    """CycleGAN.get_discriminator (CycleGAN.py:425-451), padding='valid' (CycleGAN.py:148).  ``gaussian_noise_value`` > 0 puts a
    GaussianNoise layer in front of every convolution (training mode only); StartProcess.py:96 leaves it at 0."""

    def __init__(self, filters=128, num_downsampling_blocks=2, channels=1, padding="valid", device="cuda", seed=0,
                 algo=L.ALGO_AUTO, gaussian_noise_value=0.0, act_dtype=torch.float32):
        super().__init__(device, act_dtype)
        A = self.arena
        self.filters, self.nd = filters, num_downsampling_blocks
        self.gaussian_noise_value = float(gaussian_noise_value)
        f = filters
        self.c4_in = Conv2D(A, "c4_in", 4, channels, f, stride=2, padding=padding, use_bias=True, act="lrelu",
                            act_alpha=0.2, algo=algo)
        self.down = []
        for i in range(num_downsampling_blocks):
            s = 2 if i < 3 else 1
            conv = Conv2D(A, f"down{i}", 4, f, 2 * f, stride=s, padding=padding, algo=algo)
            f *= 2
            self.down.append((conv, Norm(A, f"down{i}", f, "instance")))
        self.c4_out = Conv2D(A, "c4_out", 4, f, 1, stride=1, padding=padding, use_bias=True, algo=algo)
        self._finish(seed)

    def forward(self, tape, x, training=True):
        from .layers import gaussian_noise
        sd = self.gaussian_noise_value
        h = self.c4_in(tape, gaussian_noise(tape, x, sd, training))
        for conv, norm in self.down:
            h = norm(tape, conv(tape, gaussian_noise(tape, h, sd, training)), act="lrelu", act_alpha=0.2)
        return self.c4_out(tape, gaussian_noise(tape, h, sd, training))


class _ConvBN:
    """UNet.conv2d_bn: conv(no bias, 'same') + BatchNormalization(scale=False) [+ activation]."""

    def __init__(self, arena, name, k, cin, cout, algo):
        self.conv = Conv2D(arena, name, k, cin, cout, padding="same", algo=algo)
        self.bn = Norm(arena, f"{name}/bn", cout, "batch", scale=False)

    def __call__(self, tape, x, act, training, residual=None, out=None):
        return self.bn(tape, self.conv(tape, x), act=act, residual=residual, out=out, training=training)


class _MultiResBlock:
    """UNet.multi_res_block (UNet_Segmentation.py:451-474)."""

    def __init__(self, arena, name, u, cin, algo):
        w = MultiResUNet.ALPHA * u
        a, b, c = int(w * 0.167), int(w * 0.333), int(w * 0.5)
        self.widths = (a, b, c)
        self.cout = a + b + c
        self.sc = _ConvBN(arena, f"{name}.sc1x1", 1, cin, self.cout, algo)
        self.c3 = _ConvBN(arena, f"{name}.3", 3, cin, a, algo)
        self.c5 = _ConvBN(arena, f"{name}.5", 3, a, b, algo)
        self.c7 = _ConvBN(arena, f"{name}.7", 3, b, c, algo)
        self.bn_a = Norm(arena, f"{name}.bn_a", self.cout, "batch")
        self.bn_b = Norm(arena, f"{name}.bn_b", self.cout, "batch")

    def __call__(self, tape, x, training, out=None):
        a, b, c = self.widths
        cat = x.like(c=self.cout)     # concat buffer: producers write their slices
        if layers.SYNC_BN is not None and training:
            # data parallel: the shortcut's and the first 3x3's BatchNorms read statistics of two tensors that exist at the same time --
            # one packed exchange each way instead of two (layers.sync_norm_group)
            sc, s3 = layers.sync_norm_group(tape, [(self.sc.bn, self.sc.conv(tape, x), dict(act=None)),
                                                   (self.c3.bn, self.c3.conv(tape, x), dict(act="relu", out=cat.slice(0, a)))])
        else:
            sc = self.sc(tape, x, None, training)
            s3 = self.c3(tape, x, "relu", training, out=cat.slice(0, a))
        s5 = self.c5(tape, s3, "relu", training, out=cat.slice(a, b))
        self.c7(tape, s5, "relu", training, out=cat.slice(a + b, c))
        h = self.bn_a(tape, cat, act="relu", residual=sc, training=training)   # relu(shortcut + BN(cat))
        return self.bn_b(tape, h, training=training, out=out)


class _ResPath:
    """UNet.res_path (UNet_Segmentation.py:476-503)."""

    def __init__(self, arena, name, filters, length, cin, algo):
        self.stages = []
        for i in range(length):
            sc = _ConvBN(arena, f"{name}.{i}.sc", 1, cin, filters, algo)
            c3 = _ConvBN(arena, f"{name}.{i}.3", 3, cin, filters, algo)
            bn = Norm(arena, f"{name}.{i}.bn", filters, "batch")
            self.stages.append((sc, c3, bn))
            cin = filters

    def __call__(self, tape, x, training, out=None):
        for i, (sc, c3, bn) in enumerate(self.stages):
            if layers.SYNC_BN is not None and training:
                # forward statistics of the pair in one exchange; backward layer by layer (o's gradient is written by sc's backward)
                y3, y1 = c3.conv(tape, x), sc.conv(tape, x)
                o = y3.like()
                o, s = layers.sync_norm_group(tape, [(c3.bn, y3, dict(act="relu", out=o)), (sc.bn, y1, dict(act="relu", residual=o))],
                                              pack_backward=False)
            else:
                o = c3(tape, x, "relu", training)
                s = sc(tape, x, "relu", training, residual=o)          # relu(BN(conv1x1(x)) + o)
            x = bn(tape, s, training=training, out=out if i == len(self.stages) - 1 else None)
        return x


class MultiResUNet(Network):
    """UNet.multi_res_unet (UNet_Segmentation.py:505-562).  ``output_channels == 1``: conv2d_bn(1, 1x1, sigmoid) head (:556-557);
    otherwise Conv2D(output_channels, 1x1, bias) + softmax over the channels (:558-560).  Decoder block widths follow the
    reference's hard-coded 32*8 / 32*4 / 32*2 (UNet_Segmentation.py:543,546,549), independent of ``conv_filters``."""

    ALPHA = 1.67

    def __init__(self, conv_filters=16, device="cuda", seed=0, algo=L.ALGO_AUTO, act_dtype=torch.float32, output_channels=1):
        super().__init__(device, act_dtype)
        self.output_channels = output_channels
        A = self.arena
        f = self.filters = conv_filters
        self.mrb1 = _MultiResBlock(A, "mrb1", f, 1, algo)
        self.rp1 = _ResPath(A, "rp1", f, 4, self.mrb1.cout, algo)
        self.mrb2 = _MultiResBlock(A, "mrb2", f * 2, self.mrb1.cout, algo)
        self.rp2 = _ResPath(A, "rp2", f * 2, 3, self.mrb2.cout, algo)
        self.mrb3 = _MultiResBlock(A, "mrb3", f * 4, self.mrb2.cout, algo)
        self.rp3 = _ResPath(A, "rp3", f * 4, 2, self.mrb3.cout, algo)
        self.mrb4 = _MultiResBlock(A, "mrb4", f * 8, self.mrb3.cout, algo)
        self.rp4 = _ResPath(A, "rp4", f * 8, 1, self.mrb4.cout, algo)
        self.mrb5 = _MultiResBlock(A, "mrb5", f * 16, self.mrb4.cout, algo)
        self.up6 = Conv2D(A, "up6T", 2, self.mrb5.cout, f * 8, stride=2, use_bias=True, transposed=True, algo=algo)
        self.mrb6 = _MultiResBlock(A, "mrb6", 32 * 8, f * 16, algo)
        self.up7 = Conv2D(A, "up7T", 2, self.mrb6.cout, f * 4, stride=2, use_bias=True, transposed=True, algo=algo)
        self.mrb7 = _MultiResBlock(A, "mrb7", 32 * 4, f * 8, algo)
        self.up8 = Conv2D(A, "up8T", 2, self.mrb7.cout, f * 2, stride=2, use_bias=True, transposed=True, algo=algo)
        self.mrb8 = _MultiResBlock(A, "mrb8", 32 * 2, f * 4, algo)
        self.up9 = Conv2D(A, "up9T", 2, self.mrb8.cout, f, stride=2, use_bias=True, transposed=True, algo=algo)
        self.mrb9 = _MultiResBlock(A, "mrb9", f, f * 2, algo)
        if output_channels == 1:
            self.head = _ConvBN(A, "out1x1", 1, self.mrb9.cout, 1, algo)
        else:
            self.head = Conv2D(A, "out1x1", 1, self.mrb9.cout, output_channels, use_bias=True, algo=algo)
        self._finish(seed)

    def forward(self, tape, x, training=True):
        ph, pw = (16 - x.h % 16) % 16, (16 - x.w % 16) % 16
        x = reflect_pad(tape, x, pw, ph)                      # UNet_Segmentation.py:520-522
        f, t = self.filters, training
        dev = x.device
        # The four ResPaths depend on one encoder block each and are needed only where the decoder concatenates them: with
        # tape.branch_streams they run on streams of their own beside the deeper part of the network, in the forward pass and in the
        # replay (engine.Branch); otherwise inline, right before the concatenation's consumer (the order of the plain program).
        bs = tape.branch_streams if (tape.enabled and x.device.type == "cuda") else None

        def res_path(k, rp, m, dst):
            """Start ResPath k (input m, output into the concat slice dst); returns what join() needs."""
            if bs is None:
                return lambda: rp(tape, m, t, out=dst)
            br = tape.fork(bs[k % len(bs)])
            with br:
                rp(tape, m, t, out=dst)
            return br.join

        m1 = self.mrb1(tape, x, t)
        p1 = maxpool2x2(tape, m1)
        # skip concatenations [upT(deeper), ResPath(encoder)]: both producers write into the concat buffer
        cat9 = m1.like(c=f * 2)
        j1 = res_path(0, self.rp1, m1, cat9.slice(f, f))
        m2 = self.mrb2(tape, p1, t)
        p2 = maxpool2x2(tape, m2)
        cat8 = m2.like(c=f * 4)
        j2 = res_path(1, self.rp2, m2, cat8.slice(f * 2, f * 2))
        m3 = self.mrb3(tape, p2, t)
        p3 = maxpool2x2(tape, m3)
        cat7 = m3.like(c=f * 8)
        j3 = res_path(2, self.rp3, m3, cat7.slice(f * 4, f * 4))
        m4 = self.mrb4(tape, p3, t)
        p4 = maxpool2x2(tape, m4)
        cat6 = m4.like(c=f * 16)
        j4 = res_path(3, self.rp4, m4, cat6.slice(f * 8, f * 8))
        m5 = self.mrb5(tape, p4, t)
        j4()
        self.up6(tape, m5, out=cat6.slice(0, f * 8))
        m6 = self.mrb6(tape, cat6, t)
        j3()
        self.up7(tape, m6, out=cat7.slice(0, f * 4))
        m7 = self.mrb7(tape, cat7, t)
        j2()
        self.up8(tape, m7, out=cat8.slice(0, f * 2))
        m8 = self.mrb8(tape, cat8, t)
        j1()
        self.up9(tape, m8, out=cat9.slice(0, f))
        m9 = self.mrb9(tape, cat9, t)
        m9 = crop(tape, m9, ph // 2, ph // 2 + ph % 2, pw // 2, pw // 2 + pw % 2)     # Cropping2D, UNet_Segmentation.py:554
        if self.output_channels == 1:
            return self.head(tape, m9, "sigmoid", t)
        from .layers import softmax
        return softmax(tape, self.head(tape, m9))
