"""Oracle layer arithmetic (test infrastructure; see oracle/__init__.py).

Every function takes and returns NHWC tensors and Keras-layout weights, like
the Keras layers the reference calls; torch autograd provides the backward.
The Keras-internal semantics restated here are the SURVEY.md section-8 K-list.
"""
import torch
import torch.nn.functional as F


def _nchw(x):
    return x.permute(0, 3, 1, 2)


def _nhwc(x):
    return x.permute(0, 2, 3, 1)


def reflection_pad(x, padding):
    """ReflectionPadding2D.call -- CycleGAN.py:495-506 / UNet_Segmentation.py:578-589.

    ``padding=(pad_w_total, pad_h_total)``; before = p//2, after = p//2 + p%2.
    """
    pw, ph = padding
    if pw == 0 and ph == 0:
        return x
    pad = (pw // 2, pw // 2 + pw % 2, ph // 2, ph // 2 + ph % 2)
    return _nhwc(F.pad(_nchw(x), pad, mode="reflect"))


def same_pad_amounts(size, k, s):
    """Keras/TF 'same' padding for one spatial dim (K-list 2).

    total = k - 1 - ((size - 1) % s); before = total // 2; after = (total + 1) // 2.
    """
    total = max(k - 1 - ((size - 1) % s), 0)
    return total // 2, (total + 1) // 2


def conv2d(x, kernel, bias=None, stride=1, padding="valid"):
    """keras.layers.Conv2D on the torch backend.  kernel: (kh, kw, cin, cout) (K-list 1)."""
    kh, kw = kernel.shape[0], kernel.shape[1]
    xc = _nchw(x)
    if padding == "same":
        pt, pb = same_pad_amounts(x.shape[1], kh, stride)
        pl, pr = same_pad_amounts(x.shape[2], kw, stride)
        xc = F.pad(xc, (pl, pr, pt, pb))
    elif padding != "valid":
        raise ValueError(padding)
    w = kernel.permute(3, 2, 0, 1).contiguous()
    y = F.conv2d(xc, w, bias, stride=stride)
    return _nhwc(y)


def conv2d_transpose(x, kernel, bias=None, stride=2):
    """keras.layers.Conv2DTranspose(padding='same') on the torch backend (K-list 1, 3).

    kernel: (kh, kw, cout, cin).  k=3,s=2 -> torch padding=1, output_padding=1;
    k=2,s=2 -> padding=0, output_padding=0.  No kernel flip.
    """
    k = kernel.shape[0]
    assert kernel.shape[1] == k
    output_padding = stride - k % 2
    torch_padding = max(-((k % 2 - k + output_padding) // 2), 0)
    torch_output_padding = 2 * torch_padding + k % 2 - k + output_padding
    w = kernel.permute(3, 2, 0, 1).contiguous()  # (cin, cout, kh, kw)
    y = F.conv_transpose2d(_nchw(x), w, bias, stride=stride, padding=torch_padding,
                           output_padding=torch_output_padding)
    return _nhwc(y)


def instance_norm(x, gamma, beta, eps=1e-5):
    """GroupNormalization(groups=-1, axis=3) == per-sample per-channel norm (T5, K-list 4).

    var = E[x^2] - E[x]^2 (ops.moments on torch); y = x*inv + (beta - mean*inv), inv = rsqrt(var+eps)*gamma.
    """
    mean = x.mean(dim=(1, 2), keepdim=True)
    var = (x * x).mean(dim=(1, 2), keepdim=True) - mean * mean
    inv = torch.rsqrt(var + eps) * gamma
    return x * inv + (beta - mean * inv)


def batch_norm(x, gamma, beta, moving_mean, moving_var, training, momentum=0.99, eps=1e-3):
    """keras.layers.BatchNormalization(axis=3) (T10, K-list 5).

    Returns (y, new_moving_mean, new_moving_var).  ``gamma`` may be None (scale=False).
    Training: biased batch variance over (N,H,W), also used for the moving update.
    """
    if training:
        mean = x.mean(dim=(0, 1, 2))
        var = (x * x).mean(dim=(0, 1, 2)) - mean * mean
        new_mm = moving_mean * momentum + mean.detach() * (1.0 - momentum)
        new_mv = moving_var * momentum + var.detach() * (1.0 - momentum)
    else:
        mean, var = moving_mean, moving_var
        new_mm, new_mv = moving_mean, moving_var
    inv = torch.rsqrt(var + eps)
    if gamma is not None:
        inv = inv * gamma
    return x * inv + (beta - mean * inv), new_mm, new_mv


def max_pool2x2(x):
    return _nhwc(F.max_pool2d(_nchw(x), 2))


def leaky_relu(x, alpha=0.2):
    return F.leaky_relu(x, alpha)


# ---- losses (Keras 'sum_over_batch_size' = mean over every element; K-list 7) ----

def mse(y_true, y_pred):
    return ((y_true - y_pred) ** 2).mean()


def mae(y_true, y_pred):
    return (y_true - y_pred).abs().mean()


def weighted_bce_multi(y_true, y_pred, weighting, eps=1e-7):
    """The reference's loss closure for a multi-channel output (UNet_Segmentation.py:379-384): BinaryCrossentropy(reduction none)
    averages over the LAST axis, the result is broadcast back over the channels and weighted with y_true * (w - 1) + 1."""
    p = y_pred.clamp(eps, 1.0 - eps)
    bce = -(y_true * torch.log(p) + (1.0 - y_true) * torch.log(1.0 - p)).mean(dim=-1, keepdim=True)
    weights = y_true * (weighting - 1.0) + 1.0
    return (bce * weights).mean()


def weighted_bce(y_true, y_pred, weighting, eps=1e-7):
    """UNet_Segmentation.py:379-384 with Keras' BinaryCrossentropy(reduction='none')."""
    p = y_pred.clamp(eps, 1.0 - eps)
    bce = -(y_true * torch.log(p) + (1.0 - y_true) * torch.log(1.0 - p))
    bce = bce.mean(dim=-1, keepdim=True)
    w = y_true * (weighting - 1.0) + 1.0
    return (bce * w).mean()


class KerasAdam:
    """keras.optimizers.Adam as applied by ``optimizer.apply(grads, variables)`` (T8, K-list 6)."""

    def __init__(self, learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.learning_rate = learning_rate
        self.beta_1, self.beta_2, self.epsilon = beta_1, beta_2, epsilon
        self.iterations = 0
        self.m = None
        self.v = None

    def apply(self, grads, variables):
        params = [getattr(v, "value", v) for v in variables]
        if self.m is None:
            self.m = [torch.zeros_like(p) for p in params]
            self.v = [torch.zeros_like(p) for p in params]
        t = self.iterations + 1
        lr = float(self.learning_rate)
        alpha = lr * (1.0 - self.beta_2 ** t) ** 0.5 / (1.0 - self.beta_1 ** t)
        with torch.no_grad():
            for p, g, m, v in zip(params, grads, self.m, self.v):
                # keras/src/optimizers/adam.py update_step: assign_add forms, (1 - beta) formed in python double precision
                m.add_((g - m) * (1.0 - self.beta_1))
                v.add_((g * g - v) * (1.0 - self.beta_2))
                p.sub_(m * alpha / (v.sqrt() + self.epsilon))
        self.iterations = t
