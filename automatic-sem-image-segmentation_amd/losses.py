"""Loss kernels writing device scalars + gradients (CycleGAN.py:301-308,644-650; UNet_Segmentation.py:379-384)."""

from . import _lib as L
from .engine import _p, _stream, workspace


def _dense(a):
    assert a.cs == a.c and a.c0 == 0, "losses operate on dense activations"


def mse_const(pred, target, grad_scale, loss_slot, want_grad=True):
    """loss_slot[0] = mean((target - pred)^2); pred.grad = grad_scale * d/dpred."""
    lib = L.load()
    _dense(pred)
    g = None
    if want_grad:
        g, acc = pred.grad_target()
        assert acc == 0
    count = pred.rows * pred.c
    ws = workspace(lib.ss_loss_workspace_bytes(count), pred.device)
    L.check(lib.ss_loss_mse_const_t(pred.dt, pred.ptr, count, float(target), float(grad_scale), _p(loss_slot),
                                  g.ptr if g is not None else None, _p(ws), ws.numel(), _stream()), "ss_loss_mse_const")


def mae(truth, pred, grad_scale, loss_slot, want_grad=True):
    lib = L.load()
    _dense(pred)
    _dense(truth)
    g = None
    if want_grad:
        g, acc = pred.grad_target()
        assert acc == 0
    count = pred.rows * pred.c
    ws = workspace(lib.ss_loss_workspace_bytes(count), pred.device)
    assert truth.dt == pred.dt
    L.check(lib.ss_loss_mae_t(pred.dt, truth.ptr, pred.ptr, count, float(grad_scale), _p(loss_slot),
                            g.ptr if g is not None else None, _p(ws), ws.numel(), _stream()), "ss_loss_mae")


def weighted_bce(truth, pred, weighting, grad_scale, out3, want_grad=True):
    lib = L.load()
    _dense(pred)
    _dense(truth)
    g = None
    if want_grad:
        g, acc = pred.grad_target()
        assert acc == 0
    count = pred.rows * pred.c
    ws = workspace(lib.ss_loss_workspace_bytes(count), pred.device)
    assert truth.dt == pred.dt
    if pred.c > 1:          # multi-class head: the reference's closure averages the BCE over the channels first (UNet_Segmentation.py:379-384)
        assert truth.c == pred.c, "one-hot targets, one channel per class"
        L.check(lib.ss_loss_weighted_bce_mc_t(pred.dt, truth.ptr, pred.ptr, pred.rows, pred.c, float(weighting), float(grad_scale), _p(out3),
                                              g.ptr if g is not None else None, _p(ws), ws.numel(), _stream()), "ss_loss_weighted_bce_mc")
        return
    L.check(lib.ss_loss_weighted_bce_t(pred.dt, truth.ptr, pred.ptr, count, float(weighting), float(grad_scale), _p(out3),
                                     g.ptr if g is not None else None, _p(ws), ws.numel(), _stream()), "ss_loss_weighted_bce")
