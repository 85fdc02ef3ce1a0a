"""keras.optimizers.Adam semantics on a flat parameter arena (CycleGAN.py:168-171,668-669,690-692;
UNet_Segmentation.py:393): one fused HIP launch per network and step (ss_adam_keras)."""
import math

from . import _lib as L
from .engine import _p, _stream


class Adam:
    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, weight_decay=None):
        self.learning_rate = learning_rate
        self.beta_1, self.beta_2, self.epsilon = beta_1, beta_2, epsilon
        # Keras applies `variable -= variable * weight_decay * lr` when weight_decay is not None; the reference
        # only ever passes 0.0 (UNet_Segmentation.py:393), which is a no-op.
        if weight_decay not in (None, 0, 0.0):
            raise NotImplementedError("weight_decay != 0 is not on the reference path")
        self.iterations = 0

    def next_alpha(self):
        """Advance the iteration counter and return this step's alpha = lr * sqrt(1 - beta_2^t) / (1 - beta_1^t), t = iterations + 1."""
        t = self.iterations + 1
        self.iterations = t
        return float(self.learning_rate) * math.sqrt(1.0 - self.beta_2 ** t) / (1.0 - self.beta_1 ** t)

    def apply(self, net, grad_scale=1.0, alpha_dev=None):
        """Equivalent of ``optimizer.apply(grads, net.trainable_weights)`` with the grads in ``net.arena.grads``.
        alpha_dev = a 1-element fp32 device tensor: the launch reads alpha from there when it RUNS (hipGraph replays: the caller
        writes ``next_alpha()`` into it before every replay and this call does not touch the iteration counter)."""
        lib = L.load()
        a = net.arena
        a.join_refresh()          # a weight-cache refresh on a side stream may still be reading the old values
        if alpha_dev is None:
            L.check(lib.ss_adam_keras(_p(a.params), _p(a.grads), _p(a.m), _p(a.v), a.n_train, self.next_alpha(),
                                      self.beta_1, self.beta_2, self.epsilon, float(grad_scale), _stream()), "ss_adam_keras")
        else:
            L.check(lib.ss_adam_keras_dev(_p(a.params), _p(a.grads), _p(a.m), _p(a.v), a.n_train, _p(alpha_dev),
                                          self.beta_1, self.beta_2, self.epsilon, float(grad_scale), _stream()), "ss_adam_keras_dev")
        a.touch()
