"""Layer operations on top of the C ABI: each op runs its forward kernels and records the backward closure
on the tape.  Mirrors the Keras layers the reference composes (Conv2D / Conv2DTranspose /
GroupNormalization / BatchNormalization / MaxPooling2D / activations; CycleGAN.py:323-451,
UNet_Segmentation.py:401-562) with the fusions chosen for MI355X: reflection padding lives in the conv
gather, activation and residual add live in the norm apply pass, bias+activation in the conv epilogue.
"""
import ctypes
import os

import torch

from . import _lib as L
from .engine import current_stream_obj, TIMER, Act, DeferredNorm, _p, _stream, workspace, zero_

CONV_STATS = os.environ.get("SS_CONV_STATS", "1") != "0"          # 0: norms always run their own statistics pass (measurement)
NORM_AMAX = os.environ.get("SS_NORM_AMAX", "1") != "0"            # 0: convolutions scan their operands for the x3h scales themselves (measurement)
WEIGHT_CACHE = os.environ.get("SS_WEIGHT_CACHE", "1") != "0"      # 0: every pass derives its weight operands itself (measurement)
# Norm(..., defer=True) leaves the apply pass to the consuming convolution's operand load where that convolution can
# (ss_conv2d_fuses_in_norm); 0: every norm writes its output (measurement / A-B tests: both routes give the same bits)
FUSE_IN_NORM = os.environ.get("SS_FUSE_IN_NORM", "1") != "0"
# 1: a convolution whose path can (ss_conv2d_saved_operand_bytes) keeps the transformed input operand of its forward pass for its weight
# gradient (ss_conv_desc::saved_operand): HBM capacity for one input transform per layer and step; 0: the weight gradient transforms x again
SAVE_OPERAND = os.environ.get("SS_SAVE_OPERAND", "1") != "0"

# Cross-rank BatchNorm statistics (data parallel): set by dist.enable_sync_bn() to a callable that all-reduces (SUM) a
# float32 device tensor in place and returns the world size.  None = per-process statistics (single GPU).
SYNC_BN = None
# the gradient of an un-activated residual add is handed over as a buffer instead of copied (SS_RESIDUAL_GRAD_ALIAS=0: copied)
RESIDUAL_GRAD_ALIAS = os.environ.get("SS_RESIDUAL_GRAD_ALIAS", "1") != "0"

ACTS = {None: L.ACT_NONE, "relu": L.ACT_RELU, "lrelu": L.ACT_LRELU, "tanh": L.ACT_TANH, "sigmoid": L.ACT_SIGMOID}


def same_pad(size, k, s):
    """Keras/TF 'same' (SURVEY K-list 2): total = k-1-((size-1)%s); before = total//2."""
    total = max(k - 1 - ((size - 1) % s), 0)
    return total // 2, (total + 1) // 2


class Conv2D:
    """keras.layers.Conv2D / Conv2DTranspose parameters + geometry.

    padding: 'valid' | 'same' | ('reflect', p)  -- ('reflect', p) = ReflectionPadding2D((2p,2p)) followed by a
    'valid' conv (CycleGAN.py:326-327,368-372,392-393), fused into the gather.
    """

    def __init__(self, arena, name, k, cin, cout, stride=1, padding="valid", use_bias=False, act=None,
                 act_alpha=0.0, transposed=False, algo=L.ALGO_AUTO):
        self.arena, self.name = arena, name
        self.k, self.cin, self.cout, self.stride = k, cin, cout, stride
        self.padding, self.use_bias, self.act, self.act_alpha = padding, use_bias, act, act_alpha
        self.transposed, self.algo = transposed, (L.default_algo() if algo == L.ALGO_AUTO else algo)
        arena.declare(f"{name}/kernel", (k, k, cout, cin) if transposed else (k, k, cin, cout))
        if use_bias:
            arena.declare(f"{name}/bias", (cout,))
        self._desc_cache = {}
        self._amax_cache = {}
        self._wc = None             # weight cache of this layer (ss_wcache + device buffer), see _attach_wcache
        arena.derived.append(self)
        self.profile_tag = None     # set by bench.py to time this layer's forward launch with HIP events

    def out_hw(self, h, w):
        k, s = self.k, self.stride
        if self.transposed:
            return h * s, w * s
        if self.padding == "valid":
            return (h - k) // s + 1, (w - k) // s + 1
        if self.padding == "same":
            return -(-h // s), -(-w // s)
        p = self.padding[1]
        return h + 2 * p - k + 1, w + 2 * p - k + 1

    def desc(self, x, y):
        key = (x.n, x.h, x.w, x.cs, y.cs, x.dt)
        d = self._desc_cache.get(key)
        if d is None:
            k, s = self.k, self.stride
            pt = pl = 0
            mode = L.PAD_ZERO
            if self.transposed:
                # torch-backend alignment of Conv2DTranspose(padding='same') (SURVEY K-list 3)
                output_padding = s - k % 2
                pt = pl = max(-((k % 2 - k + output_padding) // 2), 0)
            elif self.padding == "same":
                pt, pl = same_pad(x.h, k, s)[0], same_pad(x.w, k, s)[0]
            elif self.padding != "valid":
                pt = pl = self.padding[1]
                mode = L.PAD_REFLECT
            assert x.dt == y.dt, "a convolution reads and writes activations of one storage type"
            d = L.ConvDesc(x.n, x.h, x.w, self.cin, x.cs, y.h, y.w, self.cout, y.cs, k, k, s, pt, pl, mode,
                           1 if self.transposed else 0, ACTS[self.act], float(self.act_alpha), self.algo, dtype=x.dt)
            self._desc_cache[key] = d
        return d

    def __call__(self, tape, x, out=None):
        lib = L.load()
        assert x.c == self.cin, (self.name, x.c, self.cin)
        oh, ow = self.out_hw(x.h, x.w)
        y = out if out is not None else x.like(h=oh, w=ow, c=self.cout)
        assert (y.h, y.w, y.c) == (oh, ow, self.cout)
        param_grads = tape.param_grads
        # x may be a norm output whose apply pass was deferred to this layer's operand load (engine.DeferredNorm): taken when the
        # forward pass -- and, if a weight gradient will be asked for, that pass too -- can normalise while loading; otherwise the
        # tensor is materialised now (ss_norm_apply) and this is a plain convolution
        fused = None
        if isinstance(x, DeferredNorm) and not x.materialized:
            dq = self.desc(x.pre, y)
            if self._fuses_in_norm(dq, L.PASS_FWD) and (not (param_grads and tape.enabled) or self._fuses_in_norm(dq, L.PASS_BWD_WEIGHT)):
                fused = x
            else:
                x.tensor()
        xin = fused.pre if fused is not None else x          # what the kernels read
        d = self.desc(xin, y)
        self._set_in_norm(d, fused)
        w = self.arena[f"{self.name}/kernel"]
        b = self.arena[f"{self.name}/bias"] if self.use_bias else None
        nb = lib.ss_conv2d_workspace_bytes(ctypes.byref(d), L.PASS_FWD)
        ws = workspace(nb, x.device)
        e0 = TIMER.start() if (TIMER.enabled and self.profile_tag) else None
        uses = self._uses_amax(d, L.PASS_FWD)
        if (uses & 1) or fused is not None:        # this pass needs (fused: produces) max|x|: into x's slot unless an earlier pass already has
            d.x_amax, d.x_amax_valid = x.amax_slot(), x.amax_state()
        else:
            d.x_amax, d.x_amax_valid = None, 0
        d.dy_amax, d.dy_amax_valid = None, 0
        wst = self._attach_wcache(d, L.PASS_FWD)
        # statistics of the output for a following norm, taken in the epilogue that writes y (where the path can: Winograd forward)
        sc = self._stats_chunks(d) if (CONV_STATS and y.parent is None and y.c0 == 0 and y.c == y.cs) else 0
        if sc:
            st = torch.empty(y.n * sc * self.cout * 2, dtype=torch.float32, device=y.device)
            d.y_stats = st.data_ptr()
        else:
            d.y_stats = None
        # the transformed input operand of this pass, kept for its weight gradient where the path can hand it over (Winograd x3h
        # forward -> pre-split-plane weight gradient): one input transform per layer and step less, paid in HBM capacity
        sv = None
        if SAVE_OPERAND and param_grads and tape.enabled:
            nsv = self._saved_bytes(d)
            if nsv:
                sv = torch.empty(nsv, dtype=torch.uint8, device=y.device)
        d.saved_operand = sv.data_ptr() if sv is not None else None
        L.check(lib.ss_conv2d_fwd(ctypes.byref(d), xin.ptr, _p(w), _p(b), y.ptr, _p(ws), ws.numel(), _stream()),
                f"conv2d_fwd[{self.name}]")
        d.saved_operand = None
        if sc:
            y.stats = (st, sc)
        self._wcache_done(wst)
        if (uses & 1) or fused is not None:
            x.amax_valid = True
        if e0 is not None:
            TIMER.stop(e0, self.profile_tag, x.n)
        pnames = [f"{self.name}/kernel"] + ([f"{self.name}/bias"] if self.use_bias else [])
        if param_grads and tape.enabled and tape.count_uses:
            self.arena.note_use(pnames)

        def backward():
            dy = y.get_grad()
            if dy is None:
                return
            if self.act is not None:
                dz = y.like(requires_grad=False)
                L.check(lib.ss_act_bwd_t(y.dt, ACTS[self.act], float(self.act_alpha), dy.ptr, dy.cs, y.ptr, y.cs,
                                         dz.ptr, dz.cs, y.rows, y.c, _stream()), "act_bwd")
                dy = dz
            # the descriptor's out_cstride must describe the dy buffer actually passed
            dd = d if dy.cs == y.cs else self.desc_with_out_cs(d, dy.cs)
            if param_grads:
                nbw = lib.ss_conv2d_workspace_bytes(ctypes.byref(dd), L.PASS_BWD_WEIGHT)
                wsw = workspace(nbw, x.device)
                gw = self.arena.grad(f"{self.name}/kernel")
                gb = self.arena.grad(f"{self.name}/bias") if self.use_bias else None
                uses = self._uses_amax(dd, L.PASS_BWD_WEIGHT)
                self._set_in_norm(dd, fused)
                dd.x_amax, dd.x_amax_valid = (x.amax_slot(), x.amax_state()) if uses & 1 else (None, 0)
                dd.dy_amax, dd.dy_amax_valid = (dy.amax_slot(), dy.amax_state()) if uses & 2 else (None, 0)
                dd.saved_operand = sv.data_ptr() if sv is not None else None
                # (a pass that would have to COMPUTE a maximum into a shared slot stays on the chain: the data gradient trusts the slot)
                amax_ready = (not (uses & 1) or x.amax_valid) and (not (uses & 2) or dy.amax_valid)
                side = tape.wgrad_stream if amax_ready else None
                if side is not None:
                    # off the chain: dW is read by nobody in backward.  The side stream starts behind everything issued so far (dy is
                    # final), works in its own scratch buffer, and the tensors it reads may not be recycled under it (dy can be a
                    # temporary of this closure).  The launch takes the side stream's handle directly: entering torch's stream context
                    # costs ~25 us per convolution, which paces the per-GPU-batch-1 steps (host-bound there).
                    side.wait_stream(current_stream_obj())
                    sraw = side.cuda_stream
                    wss = workspace(nbw, x.device, sraw)
                    L.check(lib.ss_conv2d_bwd_weight(ctypes.byref(dd), xin.ptr, dy.ptr, _p(gw), _p(gb), 1, _p(wss), wss.numel(),
                                                     ctypes.c_void_p(sraw)), f"conv2d_bwd_weight[{self.name}] (side stream)")
                    dy.t.record_stream(side)
                    xin.t.record_stream(side)
                    if sv is not None:
                        sv.record_stream(side)
                    if self.arena.grad_hook is not None:          # a gradient exchange launched from here orders itself behind THIS stream
                        with torch.cuda.stream(side):
                            self.arena.note_done(pnames)
                    else:
                        self.arena.note_done(pnames)
                else:
                    L.check(lib.ss_conv2d_bwd_weight(ctypes.byref(dd), xin.ptr, dy.ptr, _p(gw), _p(gb), 1, _p(wsw), wsw.numel(),
                                                     _stream()), f"conv2d_bwd_weight[{self.name}]")
                dd.saved_operand = None
                if uses & 1:
                    x.amax_valid = True
                if uses & 2:
                    dy.amax_valid = True
                if side is None:
                    self.arena.note_done(pnames)
            if x.requires_grad:
                dx, accum = x.grad_target()
                ddx = dd if dx.cs == dd.in_cstride else self.desc_with_in_cs(dd, dx.cs)
                nbd = lib.ss_conv2d_workspace_bytes(ctypes.byref(ddx), L.PASS_BWD_DATA)
                wsd = workspace(nbd, x.device)
                uses = self._uses_amax(ddx, L.PASS_BWD_DATA)
                self._set_in_norm(ddx, None)
                ddx.x_amax, ddx.x_amax_valid = None, 0
                ddx.dy_amax, ddx.dy_amax_valid = (dy.amax_slot(), dy.amax_state()) if uses & 2 else (None, 0)
                wst = self._attach_wcache(ddx, L.PASS_BWD_DATA)
                L.check(lib.ss_conv2d_bwd_data(ctypes.byref(ddx), dy.ptr, _p(w), dx.ptr, accum, _p(wsd), wsd.numel(),
                                               _stream()), f"conv2d_bwd_data[{self.name}]")
                self._wcache_done(wst)
                if uses & 2:
                    dy.amax_valid = True

        tape.record(backward)
        return y

    @staticmethod
    def _set_in_norm(d, x):
        """Point `d` at the statistics of the deferred norm `x` (engine.DeferredNorm) or switch the fused input norm off (None).
        Descriptors are cached per geometry and shared between calls: set before EVERY launch."""
        if x is None:
            d.in_norm_groups = 0
            d.in_norm_mean = d.in_norm_rstd = d.in_norm_gamma = d.in_norm_beta = None
            return
        d.in_norm_mean, d.in_norm_rstd = x.mean.data_ptr(), x.rstd.data_ptr()
        d.in_norm_gamma = x.gamma.data_ptr() if x.gamma is not None else None
        d.in_norm_beta = x.beta.data_ptr()
        d.in_norm_groups, d.in_norm_act, d.in_norm_alpha = x.groups, x.act_code, float(x.act_alpha)

    def _saved_bytes(self, d):
        """ss_conv2d_saved_operand_bytes(d), cached per geometry and configuration."""
        key = (d.n, d.ih, d.iw, d.in_cstride, d.out_cstride, d.dtype, "saved", L.CONFIG_EPOCH)
        u = self._amax_cache.get(key)
        if u is None:
            u = self._amax_cache[key] = int(L.load().ss_conv2d_saved_operand_bytes(ctypes.byref(d)))
        return u

    def _fuses_in_norm(self, d, pass_):
        """ss_conv2d_fuses_in_norm(d, pass), cached per geometry and configuration."""
        key = (d.n, d.ih, d.iw, d.in_cstride, d.out_cstride, d.dtype, "in_norm", pass_, L.CONFIG_EPOCH)
        u = self._amax_cache.get(key)
        if u is None:
            u = self._amax_cache[key] = bool(FUSE_IN_NORM and L.load().ss_conv2d_fuses_in_norm(ctypes.byref(d), pass_))
        return u

    def will_fuse_in_norm(self, pre, tape):
        """Would this layer, fed the DEFERRED normalisation of the dense activation `pre` (any storage type), normalise in its operand load (forward
        and, when `tape` asks for weight gradients, the weight gradient too)?  Norm(..., defer_to=layer) asks before it skips its
        apply pass: a consumer that cannot fuse gets the ordinary norm (same kernels, same bits as without any deferral)."""
        if not FUSE_IN_NORM or pre.c != self.cin or self.transposed:
            return False
        oh, ow = self.out_hw(pre.h, pre.w)
        key = (pre.n, pre.h, pre.w, pre.cs, self.cout, pre.dt)
        d = self._desc_cache.get(key)
        if d is None:          # same construction as desc(): geometry only, the output is dense
            class _Y:          # noqa: N801  (shape carrier)
                h, w, cs, dt = oh, ow, self.cout, pre.dt
            d = self.desc(pre, _Y)
        if not self._fuses_in_norm(d, L.PASS_FWD):
            return False
        return not (tape.param_grads and tape.enabled) or self._fuses_in_norm(d, L.PASS_BWD_WEIGHT)

    def _attach_wcache(self, d, pass_):
        """Point `d` at this layer's weight cache (include/semseg_hip.h ss_wcache): Winograd-transformed / transposed / split weight
        planes are derived once per weight VERSION (optimizer step, set_weights, any in-place write torch knows of, configuration
        switch) instead of once per use; every descriptor geometry of the layer shares the cache.  Returns the state to hand to
        _wcache_done after the launch, or None when the layer keeps no cache."""
        st = self._wc
        if st is False or not WEIGHT_CACHE:
            d.w_cache = None
            return None
        ver = (self.arena.weights_key(), L.CONFIG_EPOCH)
        if st is None:
            lib = L.load()
            nb = lib.ss_conv2d_wcache_bytes(ctypes.byref(d), L.PASS_FWD) + lib.ss_conv2d_wcache_bytes(ctypes.byref(d), L.PASS_BWD_DATA)
            if nb == 0:
                self._wc = False
                d.w_cache = None
                return None
            buf = torch.empty(nb, dtype=torch.uint8, device=self.arena.device)
            st = self._wc = dict(buf=buf, c=L.WCache(buf.data_ptr(), nb), ver=ver, sync=dict(ev=None, synced=set()), users={})
        c = st["c"]
        if st["ver"] != ver:
            c.count, c.used = 0, 0          # = ss_wcache_invalidate
            st["ver"], st["sync"] = ver, dict(ev=None, synced=set())
        ukey = (d.n, d.ih, d.iw, d.in_cstride, d.out_cstride, d.dtype, pass_)
        if ukey not in st["users"]:
            du = L.ConvDesc.from_buffer_copy(d)          # private copy for refresh_wcache: no activation-side pointers
            du.x_amax, du.dy_amax, du.x_amax_valid, du.dy_amax_valid = None, None, 0, 0
            self._set_in_norm(du, None)
            st["users"][ukey] = (du, pass_)
        cur = _stream().value or 0
        sync = st["sync"]
        if sync["ev"] is not None and cur not in sync["synced"]:
            current_stream_obj().wait_event(sync["ev"])      # entries filled on another stream: order this stream after them
            sync["synced"].add(cur)
        d.w_cache = ctypes.addressof(c)
        st["fills0"], st["cur"] = c.fills, cur
        return st

    @staticmethod
    def _wcache_done(st):
        if st is not None and st["c"].fills != st["fills0"]:
            ev = torch.cuda.Event()
            ev.record(current_stream_obj())
            st["sync"] = dict(ev=ev, synced={st["cur"]})

    def refresh_wcache(self, sync):
        """Recompute on the CURRENT stream every operand this layer kept for the previous weight version (fill-only calls of the
        passes that used the cache).  Called through ParamArena.refresh_derived() before concurrent streams fork, so that no
        stream has to wait for another one's first use of a layer.  Returns the number of passes refreshed."""
        st = self._wc
        if not st or not WEIGHT_CACHE:
            return 0
        ver = (self.arena.weights_key(), L.CONFIG_EPOCH)
        if st["ver"] == ver:
            return 0
        lib = L.load()
        c = st["c"]
        c.count, c.used = 0, 0
        st["ver"], st["sync"] = ver, sync
        w = self.arena[f"{self.name}/kernel"]
        c.fill_only = 1
        try:
            for d, pass_ in st["users"].values():
                d.w_cache = ctypes.addressof(c)
                nb = lib.ss_conv2d_workspace_bytes(ctypes.byref(d), pass_)
                ws = workspace(nb, self.arena.device)
                if pass_ == L.PASS_FWD:
                    L.check(lib.ss_conv2d_fwd(ctypes.byref(d), None, _p(w), None, None, _p(ws), ws.numel(), _stream()),
                            f"conv2d_fwd[{self.name}] (weight refresh)")
                else:
                    L.check(lib.ss_conv2d_bwd_data(ctypes.byref(d), None, _p(w), None, 0, _p(ws), ws.numel(), _stream()),
                            f"conv2d_bwd_data[{self.name}] (weight refresh)")
        finally:
            c.fill_only = 0
        return len(st["users"])

    def _stats_chunks(self, d):
        key = (d.n, d.ih, d.iw, d.in_cstride, d.out_cstride, d.dtype, "stats", L.CONFIG_EPOCH)
        u = self._amax_cache.get(key)
        if u is None:
            u = self._amax_cache[key] = int(L.load().ss_conv2d_stats_chunks(ctypes.byref(d)))
        return u

    def _uses_amax(self, d, pass_):
        """ss_conv2d_uses_amax(d, pass), cached per geometry: bit 0 = the pass reads (and leaves in the slot) max|x|, bit 1 = max|dy|."""
        # dtype / transposed are part of the answer (native 16-bit paths need no maximum): one layer can see both storage types
        key = (d.n, d.ih, d.iw, d.in_cstride, d.out_cstride, d.oh, d.ow, d.dtype, d.transposed, pass_, L.CONFIG_EPOCH)
        u = self._amax_cache.get(key)
        if u is None:
            u = L.load().ss_conv2d_uses_amax(ctypes.byref(d), pass_)
            self._amax_cache[key] = u
        return u

    @staticmethod
    def desc_with_out_cs(d, cs):
        d2 = L.ConvDesc.from_buffer_copy(d)
        d2.out_cstride = cs
        return d2

    @staticmethod
    def desc_with_in_cs(d, cs):
        d2 = L.ConvDesc.from_buffer_copy(d)
        d2.in_cstride = cs
        return d2


class Norm:
    """InstanceNorm (groups = n) or BatchNorm (groups = 1) with fused activation / residual add."""

    def __init__(self, arena, name, c, kind, scale=True, eps=None, momentum=0.99):
        assert kind in ("instance", "batch")
        self.arena, self.name, self.c, self.kind, self.scale = arena, name, c, kind, scale
        self.eps = eps if eps is not None else (1e-5 if kind == "instance" else 1e-3)
        self.momentum = momentum
        self._amax_cache = {}
        if scale:
            arena.declare(f"{name}/gamma", (c,))
        arena.declare(f"{name}/beta", (c,))
        if kind == "batch":
            arena.declare(f"{name}/moving_mean", (c,), trainable=False)
            arena.declare(f"{name}/moving_variance", (c,), trainable=False)

    def _reports_amax(self, d):
        key = (d.n, d.h, d.w, d.groups, d.dtype, L.CONFIG_EPOCH)
        r = self._amax_cache.get(key)
        if r is None:
            r = self._amax_cache[key] = bool(L.load().ss_norm_reports_amax(ctypes.byref(d)))
        return r

    def __call__(self, tape, x, act=None, act_alpha=0.0, residual=None, out=None, training=True, defer_to=None):
        """defer_to = the ONE convolution layer the result goes to: when that layer can normalise in its operand load
        (Conv2D.will_fuse_in_norm) and the case allows it (no residual, no caller-provided output, relu / leaky-relu /
        no activation, per-process statistics) only the statistics are taken and an engine.DeferredNorm is returned; otherwise this
        is the ordinary norm."""
        lib = L.load()
        assert x.c == self.c
        sync_now = SYNC_BN if self.kind == "batch" else None
        defer = bool(defer_to is not None and FUSE_IN_NORM and residual is None and out is None and act in (None, "relu", "lrelu")
                     and sync_now is None and (self.kind == "instance" or training)
                     and x.parent is None and x.c0 == 0 and defer_to.will_fuse_in_norm(x, tape))
        y = out if out is not None else (None if defer else x.like())
        groups = x.n if self.kind == "instance" else 1
        assert (y is None or x.dt == y.dt) and (residual is None or residual.dt == x.dt)
        d = L.NormDesc(x.n, x.h, x.w, x.c, x.cs, y.cs if y is not None else x.c, residual.cs if residual is not None else 0, groups,
                       float(self.eps), ACTS[act], float(act_alpha), dtype=x.dt)
        gamma = self.arena[f"{self.name}/gamma"] if self.scale else None
        beta = self.arena[f"{self.name}/beta"]
        rp = residual.ptr if residual is not None else None
        if self.kind == "batch" and not training:
            L.check(lib.ss_norm_infer(ctypes.byref(d), x.ptr, _p(gamma), _p(beta),
                                      _p(self.arena[f"{self.name}/moving_mean"]),
                                      _p(self.arena[f"{self.name}/moving_variance"]), rp, y.ptr, _stream()), "norm_infer")
            return y
        mean = torch.empty(groups * x.c, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        mm = mv = None
        if self.kind == "batch":
            mm, mv = self.arena[f"{self.name}/moving_mean"], self.arena[f"{self.name}/moving_variance"]
        nb = lib.ss_norm_workspace_bytes(ctypes.byref(d))
        ws = workspace(nb, x.device)
        sync = SYNC_BN if self.kind == "batch" else None
        count = x.rows // groups
        # the norm reports max|y| while it writes y (the next convolution's x3h scale): fp32, whole-tensor outputs only -- a channel
        # slice of a concat buffer is read by its consumers together with its neighbours, under another view
        reports = NORM_AMAX and sync is None and self._reports_amax(d)
        want_amax = reports and y is not None and y.parent is None and y.c0 == 0 and y.c == y.cs and y.amax is None
        if want_amax:
            d.y_amax = y.amax_slot()
        if sync is None and x.stats is not None and x.parent is None:
            d.x_stats, d.x_stats_chunks = x.stats[0].data_ptr(), x.stats[1]
        if sync is None:
            L.check(lib.ss_norm_fwd(ctypes.byref(d), x.ptr, _p(gamma), _p(beta), rp, y.ptr if y is not None else None, _p(mean), _p(rstd),
                                    _p(mm), _p(mv), float(self.momentum), _p(ws), ws.numel(), _stream()),
                    f"norm_fwd[{self.name}]")
            if want_amax:
                y.amax_valid = True
            if y is None:          # statistics only: the apply pass belongs to the consumer (or to DeferredNorm.tensor())
                def materialize(dn):
                    out_ = x.like()
                    dm = L.NormDesc.from_buffer_copy(d)
                    dm.y_cstride, dm.y_amax, dm.x_stats, dm.x_stats_chunks = out_.cs, None, None, 0
                    rep = NORM_AMAX and self._reports_amax(dm) and dn.amax is None
                    if rep:
                        dm.y_amax = out_.amax_slot()
                    L.check(lib.ss_norm_apply(ctypes.byref(dm), x.ptr, _p(gamma), _p(beta), None, out_.ptr, _p(mean), _p(rstd), _stream()),
                            f"norm_apply[{self.name}]")
                    if rep:
                        out_.amax_valid = True
                        dn.amax, dn.amax_valid = out_.amax, True
                    elif dn.amax is not None:
                        out_.amax, out_.amax_valid = dn.amax, dn.amax_valid
                    return out_
                y = DeferredNorm(x, mean, rstd, gamma, beta, groups, ACTS[act], float(act_alpha), materialize)
        else:
            sums = torch.empty(groups * x.c * 2, dtype=torch.float32, device=x.device)
            L.check(lib.ss_norm_fwd_stats(ctypes.byref(d), x.ptr, _p(sums), _p(ws), ws.numel(), _stream()), "norm_fwd_stats")
            count = count * sync(sums)
            L.check(lib.ss_norm_fwd_finish(ctypes.byref(d), x.ptr, _p(gamma), _p(beta), rp, y.ptr, _p(sums), count, _p(mean), _p(rstd),
                                           _p(mm), _p(mv), float(self.momentum), _stream()), "norm_fwd_finish")
        param_grads = tape.param_grads
        pnames = ([f"{self.name}/gamma"] if self.scale else []) + [f"{self.name}/beta"]
        if param_grads and tape.enabled and tape.count_uses:
            self.arena.note_use(pnames)

        def backward():
            dy = y.get_grad()
            if dy is None:
                return
            dx, accum = x.grad_target()
            dres, racc = (None, 0)
            if residual is not None and residual.requires_grad:
                if (RESIDUAL_GRAD_ALIAS and act is None and sync is None and type(residual) is Act and residual.parent is None
                        and not residual.grad_init and y.parent is None and dy.c0 == 0 and dy.c == dy.cs == residual.cs
                        and residual.c0 == 0 and dy.t.shape == residual.t.shape and dy.t.dtype == residual.t.dtype):
                    # y = norm(x) + residual without an activation (the ResNet block's add, CycleGAN.py:336): d loss / d residual IS dy.
                    # This op is the first writer of the residual's gradient and the last reader of dy's buffer apart from its own
                    # kernels, so the buffer is handed over instead of copied (one tensor write less per block; the blocks of a trunk
                    # pass ONE gradient buffer down, each data gradient accumulating into it behind this op's reads on the same stream)
                    residual.grad, residual.grad_init = dy, True
                else:
                    dres, racc = residual.grad_target()
            db = L.NormDesc.from_buffer_copy(d)
            db.res_cstride = dres.cs if dres is not None else 0
            db.y_amax = None
            # max|dx| for the producing convolution's data / weight gradient: only when this op writes dx whole (a later
            # accumulation into the same gradient clears the flag, engine.Act.grad_target)
            dx_amax = reports and not accum and x.parent is None and dx.c0 == 0 and dx.c == dx.cs and dx.amax is None
            db.dx_amax = dx.amax_slot() if dx_amax else None
            ws2 = workspace(lib.ss_norm_workspace_bytes(ctypes.byref(db)), x.device)
            ggam = self.arena.grad(f"{self.name}/gamma") if (self.scale and param_grads) else None
            gbet = self.arena.grad(f"{self.name}/beta") if param_grads else None
            if sync is None:
                # relu / leaky-relu without a residual: the kernels recompute the mask from x instead of reading y
                # (none: the kernels never look at y)
                yp = None if (act in (None, "relu", "lrelu") and residual is None) else y.ptr
                L.check(lib.ss_norm_bwd(ctypes.byref(db), dy.ptr, dy.cs, x.ptr, yp, _p(gamma), _p(beta), _p(mean), _p(rstd),
                                        dx.ptr, dx.cs, accum, dres.ptr if dres is not None else None, racc,
                                        _p(ggam), _p(gbet), 1, _p(ws2), ws2.numel(), _stream()), f"norm_bwd[{self.name}]")
                if dx_amax:
                    dx.amax_valid = True
            else:
                lsums = torch.empty(groups * x.c * 2, dtype=torch.float32, device=x.device)
                L.check(lib.ss_norm_bwd_stats(ctypes.byref(db), dy.ptr, dy.cs, x.ptr, y.ptr, _p(mean), _p(rstd), _p(lsums),
                                              _p(ws2), ws2.numel(), _stream()), "norm_bwd_stats")
                gsums = lsums.clone()
                sync(gsums)
                L.check(lib.ss_norm_bwd_finish(ctypes.byref(db), dy.ptr, dy.cs, x.ptr, y.ptr, _p(gamma), _p(mean), _p(rstd),
                                               _p(gsums), _p(lsums), count, dx.ptr, dx.cs, accum,
                                               dres.ptr if dres is not None else None, racc, _p(ggam), _p(gbet), 1,
                                               _p(ws2), ws2.numel(), _stream()), "norm_bwd_finish")
            if param_grads:
                self.arena.note_done(pnames)

        tape.record(backward)
        return y


def sync_norm_group(tape, items, pack_backward=True):
    """Several BatchNorm layers whose statistics do not depend on one another, under cross-rank statistics (SYNC_BN): ONE all-reduce of
    the packed (sum, sum of squares) vectors in forward instead of one per layer, and -- where the layers' output gradients are final at
    the same point of the replay (`pack_backward`: a MultiRes block's shortcut and first 3x3, UNet_Segmentation.py:455-458) -- one for
    the packed backward sums as well.  A ResPath stage's pair (:483-488) packs in forward only: the 3x3 branch's output gradient IS the
    shortcut norm's residual gradient, which its finish pass writes.

    items: [(norm, x, dict(act=, act_alpha=, residual=, out=))] in the order the layers' apply passes have to run (a later item may take
    an earlier item's output as its residual).  Returns the outputs.  The arithmetic is ss_norm_{fwd,bwd}_stats / _finish, as in
    Norm.__call__ under SYNC_BN: the results are the unpacked ones bit for bit (the all-reduce sums the same numbers)."""
    lib = L.load()
    sync = SYNC_BN
    assert sync is not None and all(nm.kind == "batch" for nm, _, _ in items)
    dev = items[0][1].device
    offs, tot = [], 0
    for nm, x, _ in items:
        assert x.c == nm.c
        offs.append(tot)
        tot += 2 * x.c
    buf = torch.empty(tot, dtype=torch.float32, device=dev)
    st = []
    for (nm, x, kw), off in zip(items, offs):
        residual, out = kw.get("residual"), kw.get("out")
        y = out if out is not None else x.like()
        d = L.NormDesc(x.n, x.h, x.w, x.c, x.cs, y.cs, residual.cs if residual is not None else 0, 1, float(nm.eps),
                       ACTS[kw.get("act")], float(kw.get("act_alpha", 0.0)), dtype=x.dt)
        ws = workspace(lib.ss_norm_workspace_bytes(ctypes.byref(d)), dev)
        sums = buf[off:off + 2 * x.c]
        L.check(lib.ss_norm_fwd_stats(ctypes.byref(d), x.ptr, _p(sums), _p(ws), ws.numel(), _stream()), "norm_fwd_stats")
        st.append(dict(nm=nm, x=x, y=y, d=d, sums=sums, residual=residual, act=kw.get("act")))
    world = sync(buf)                                   # the group's one forward exchange
    param_grads = tape.param_grads
    for e in st:
        nm, x, y, d, residual = e["nm"], e["x"], e["y"], e["d"], e["residual"]
        e["count"] = x.rows * world
        e["mean"] = torch.empty(x.c, dtype=torch.float32, device=dev)
        e["rstd"] = torch.empty_like(e["mean"])
        e["gamma"] = nm.arena[f"{nm.name}/gamma"] if nm.scale else None
        e["beta"] = nm.arena[f"{nm.name}/beta"]
        L.check(lib.ss_norm_fwd_finish(ctypes.byref(d), x.ptr, _p(e["gamma"]), _p(e["beta"]), residual.ptr if residual is not None else None, y.ptr,
                                       _p(e["sums"]), e["count"], _p(e["mean"]), _p(e["rstd"]), _p(nm.arena[f"{nm.name}/moving_mean"]),
                                       _p(nm.arena[f"{nm.name}/moving_variance"]), float(nm.momentum), _stream()), "norm_fwd_finish")
        e["pnames"] = ([f"{nm.name}/gamma"] if nm.scale else []) + [f"{nm.name}/beta"]
        if param_grads and tape.enabled and tape.count_uses:
            nm.arena.note_use(e["pnames"])

    def bwd_stats(e, lsums):
        dy = e["y"].get_grad()
        if dy is None:
            return False
        x, d, residual = e["x"], e["d"], e["residual"]
        e["dy"] = dy
        e["dx"], e["accum"] = x.grad_target()
        e["dres"], e["racc"] = residual.grad_target() if (residual is not None and residual.requires_grad) else (None, 0)
        db = e["db"] = L.NormDesc.from_buffer_copy(d)
        db.res_cstride = e["dres"].cs if e["dres"] is not None else 0
        e["ws2"] = workspace(lib.ss_norm_workspace_bytes(ctypes.byref(db)), dev)
        e["lsums"] = lsums
        L.check(lib.ss_norm_bwd_stats(ctypes.byref(db), dy.ptr, dy.cs, x.ptr, e["y"].ptr, _p(e["mean"]), _p(e["rstd"]), _p(lsums),
                                      _p(e["ws2"]), e["ws2"].numel(), _stream()), "norm_bwd_stats")
        return True

    def bwd_finish(e, gsums):
        nm, x, dy, dx, dres, db = e["nm"], e["x"], e["dy"], e["dx"], e["dres"], e["db"]
        ggam = nm.arena.grad(f"{nm.name}/gamma") if (nm.scale and param_grads) else None
        gbet = nm.arena.grad(f"{nm.name}/beta") if param_grads else None
        L.check(lib.ss_norm_bwd_finish(ctypes.byref(db), dy.ptr, dy.cs, x.ptr, e["y"].ptr, _p(e["gamma"]), _p(e["mean"]), _p(e["rstd"]),
                                       _p(gsums), _p(e["lsums"]), e["count"], dx.ptr, dx.cs, e["accum"],
                                       dres.ptr if dres is not None else None, e["racc"], _p(ggam), _p(gbet), 1,
                                       _p(e["ws2"]), e["ws2"].numel(), _stream()), "norm_bwd_finish")
        if param_grads:
            nm.arena.note_done(e["pnames"])

    def backward():
        if pack_backward:
            lbuf = torch.empty(tot, dtype=torch.float32, device=dev)
            live = [e for e, off in zip(st, offs) if bwd_stats(e, lbuf[off:off + 2 * e["x"].c])]
            if not live:
                return
            if len(live) < len(st):          # a layer without an output gradient contributes nothing: its slot must not travel as garbage
                for e, off in zip(st, offs):
                    if e not in live:
                        zero_(lbuf[off:off + 2 * e["x"].c])
            gbuf = lbuf.clone()
            sync(gbuf)                          # the group's one backward exchange
            for e, off in zip(st, offs):
                if e in live:
                    bwd_finish(e, gbuf[off:off + 2 * e["x"].c])
            return
        for e in reversed(st):                  # dependent output gradients: layer by layer, last applied first
            lsums = torch.empty(2 * e["x"].c, dtype=torch.float32, device=dev)
            if bwd_stats(e, lsums):
                gsums = lsums.clone()
                sync(gsums)
                bwd_finish(e, gsums)

    tape.record(backward)
    return [e["y"] for e in st]


def maxpool2x2(tape, x):
    lib = L.load()
    y = x.like(h=x.h // 2, w=x.w // 2)
    L.check(lib.ss_maxpool2x2_fwd_t(x.dt, x.ptr, x.cs, y.ptr, y.cs, x.n, x.h, x.w, x.c, _stream()), "maxpool_fwd")

    def backward():
        dy = y.get_grad()
        if dy is None or not x.requires_grad:
            return
        dx, accum = x.grad_target()
        L.check(lib.ss_maxpool2x2_bwd_t(x.dt, dy.ptr, dy.cs, x.ptr, x.cs, dx.ptr, dx.cs, accum, x.n, x.h, x.w, x.c, _stream()),
                "maxpool_bwd")

    tape.record(backward)
    return y


def add_grad(dst_act, src):
    """Accumulate the dense gradient view ``src`` into dst_act's gradient (fan-out of an activation)."""
    lib = L.load()
    dg, accum = dst_act.grad_target()
    L.check(lib.ss_axpby_t(src.dt, 1.0, src.ptr, src.cs, 1.0 if accum else 0.0, dg.ptr if accum else None, dg.cs,
                         dg.ptr, dg.cs, dst_act.rows, dst_act.c, _stream()), "axpby")


def gaussian_noise(tape, x, stddev, training=True):
    """keras.layers.GaussianNoise(stddev): additive zero-mean noise in training mode only (CycleGAN.py:427,438,446); identity
    otherwise.  The samples come from torch's device generator (the reference's come from Keras' seed generator: the streams
    cannot match, only the distribution); the gradient passes through unchanged."""
    if not training or stddev <= 0:
        return x
    lib = L.load()
    noise = Act(torch.randn((x.n, x.h, x.w, x.c), dtype=torch.float32, device=x.device).to(x.dtype), requires_grad=False)
    y = x.like(requires_grad=x.requires_grad)
    L.check(lib.ss_axpby_t(x.dt, 1.0, x.ptr, x.cs, float(stddev), noise.ptr, noise.cs, y.ptr, y.cs, x.rows, x.c, _stream()), "axpby")

    def backward():
        dy = y.get_grad()
        if dy is not None and x.requires_grad:
            add_grad(x, dy)

    tape.record(backward)
    return y


def dropout(tape, x, keep, rate):
    """keras.layers.Dropout(rate) in training mode with an explicit keep mask (WassersteinGAN.py:566-567,621): y = x * keep / (1 - rate).
    ``keep``: Act of x's shape holding 0 / 1, or None for "no dropout" (identity).  The caller draws the mask (torch's device
    generator; the reference's comes from Keras' seed generator: the streams cannot match, only the distribution)."""
    if keep is None:
        return x
    lib = L.load()
    sc = 1.0 / (1.0 - float(rate))
    y = x.like(requires_grad=x.requires_grad)
    L.check(lib.ss_mul_t(x.dt, sc, x.ptr, x.cs, keep.ptr, keep.cs, y.ptr, y.cs, x.rows, x.c, _stream()), "mul")

    def backward():
        dy = y.get_grad()
        if dy is None or not x.requires_grad:
            return
        dx, accum = x.grad_target()
        if accum:
            tmp = x.like(requires_grad=False)
            L.check(lib.ss_mul_t(x.dt, sc, dy.ptr, dy.cs, keep.ptr, keep.cs, tmp.ptr, tmp.cs, x.rows, x.c, _stream()), "mul")
            L.check(lib.ss_axpby_t(x.dt, 1.0, tmp.ptr, tmp.cs, 1.0, dx.ptr, dx.cs, dx.ptr, dx.cs, x.rows, x.c, _stream()), "axpby")
        else:
            L.check(lib.ss_mul_t(x.dt, sc, dy.ptr, dy.cs, keep.ptr, keep.cs, dx.ptr, dx.cs, x.rows, x.c, _stream()), "mul")

    tape.record(backward)
    return y


def reshape(tape, x, h, w, c):
    """keras.layers.Reshape / Flatten on a dense channels-last activation: the same memory under another (h, w, c) -- no kernel."""
    assert x.c0 == 0 and x.c == x.cs and x.h * x.w * x.c == h * w * c, "reshape needs a dense activation of the same size"
    y = Act(x.t.view(x.n, h, w, c), requires_grad=x.requires_grad)

    def backward():
        dy = y.get_grad()
        if dy is None or not x.requires_grad:
            return
        assert dy.c0 == 0 and dy.c == dy.cs
        add_grad(x, Act(dy.t.view(x.t.shape), requires_grad=False))

    tape.record(backward)
    return y


def batch_split(tape, x, sizes):
    """Views of consecutive sample ranges of ``x`` (no copy).  Used to run one network pass over several input batches at once
    (CycleGAN: the "fake" and "identity" passes of a generator share weights and are per-sample independent); backward gathers
    the parts' gradients into x's gradient."""
    assert x.parent is None and x.c0 == 0 and x.c == x.cs and sum(sizes) == x.n
    parts, n0 = [], 0
    for n in sizes:
        parts.append(Act(x.t[n0:n0 + n], requires_grad=x.requires_grad))
        n0 += n

    def backward():
        grads = [q.get_grad() for q in parts]
        if not x.requires_grad or all(g is None for g in grads):
            return
        lib = L.load()
        dx, accum = x.grad_target()
        if not accum:
            zero_(dx.t)
        n0 = 0
        for q, g in zip(parts, grads):
            if g is not None:
                dst = Act(dx.t[n0:n0 + q.n], dx.c0, dx.c, False)
                L.check(lib.ss_axpby_t(g.dt, 1.0, g.ptr, g.cs, 1.0, dst.ptr, dst.cs, dst.ptr, dst.cs, q.rows, q.c, _stream()), "axpby")
            n0 += q.n

    tape.record(backward)
    return parts


def reflect_pad(tape, x, pad_w_total, pad_h_total):
    """ReflectionPadding2D as a standalone op (CycleGAN.py:482-506): total padding split p//2 before, p//2 + p%2 after."""
    if pad_w_total == 0 and pad_h_total == 0:
        return x
    lib = L.load()
    pt, pb = pad_h_total // 2, pad_h_total // 2 + pad_h_total % 2
    pl, pr = pad_w_total // 2, pad_w_total // 2 + pad_w_total % 2
    y = x.like(h=x.h + pt + pb, w=x.w + pl + pr, requires_grad=x.requires_grad)
    L.check(lib.ss_reflect_pad2d_fwd_t(x.dt, x.ptr, x.cs, y.ptr, y.cs, x.n, x.h, x.w, x.c, pt, pb, pl, pr, _stream()), "reflect_pad_fwd")

    def backward():
        dy = y.get_grad()
        if dy is None or not x.requires_grad:
            return
        dx, accum = x.grad_target()
        L.check(lib.ss_reflect_pad2d_bwd_t(x.dt, dy.ptr, dy.cs, dx.ptr, dx.cs, accum, x.n, x.h, x.w, x.c, pt, pb, pl, pr, _stream()), "reflect_pad_bwd")

    tape.record(backward)
    return y


def crop(tape, x, top, bottom, left, right):
    """keras.layers.Cropping2D(((top, bottom), (left, right)))."""
    if top == 0 and bottom == 0 and left == 0 and right == 0:
        return x
    lib = L.load()
    oh, ow = x.h - top - bottom, x.w - left - right
    y = x.like(h=oh, w=ow)
    L.check(lib.ss_crop2d_fwd_t(x.dt, x.ptr, x.cs, y.ptr, y.cs, x.n, x.h, x.w, x.c, top, left, oh, ow, _stream()), "crop_fwd")

    def backward():
        dy = y.get_grad()
        if dy is None or not x.requires_grad:
            return
        dx, accum = x.grad_target()
        L.check(lib.ss_crop2d_bwd_t(x.dt, dy.ptr, dy.cs, dx.ptr, dx.cs, accum, x.n, x.h, x.w, x.c, top, left, oh, ow, _stream()), "crop_bwd")

    tape.record(backward)
    return y


def upsample2x(tape, x):
    lib = L.load()
    y = x.like(h=2 * x.h, w=2 * x.w)
    L.check(lib.ss_upsample2x_fwd_t(x.dt, x.ptr, x.cs, y.ptr, y.cs, x.n, x.h, x.w, x.c, _stream()), "upsample_fwd")

    def backward():
        dy = y.get_grad()
        if dy is None or not x.requires_grad:
            return
        dx, accum = x.grad_target()
        L.check(lib.ss_upsample2x_bwd_t(x.dt, dy.ptr, dy.cs, dx.ptr, dx.cs, accum, x.n, x.h, x.w, x.c, _stream()), "upsample_bwd")

    tape.record(backward)
    return y


def softmax(tape, x):
    """keras.layers.Activation('softmax') over the channels of every pixel (UNet_Segmentation.py:560)."""
    lib = L.load()
    y = x.like()
    L.check(lib.ss_softmax_fwd_t(x.dt, x.ptr, x.cs, y.ptr, y.cs, x.rows, x.c, _stream()), "softmax_fwd")

    def backward():
        dy = y.get_grad()
        if dy is None or not x.requires_grad:
            return
        dx, accum = x.grad_target()
        assert not accum, "softmax input has one consumer"
        L.check(lib.ss_softmax_bwd_t(x.dt, dy.ptr, dy.cs, y.ptr, y.cs, dx.ptr, dx.cs, x.rows, x.c, _stream()), "softmax_bwd")

    tape.record(backward)
    return y


def add(tape, a, b, out=None):
    """keras.layers.add([a, b])."""
    lib = L.load()
    y = out if out is not None else a.like()
    L.check(lib.ss_axpby_t(a.dt, 1.0, a.ptr, a.cs, 1.0, b.ptr, b.cs, y.ptr, y.cs, a.rows, a.c, _stream()), "axpby")

    def backward():
        dy = y.get_grad()
        if dy is None:
            return
        for t in (a, b):
            if t.requires_grad:
                add_grad(t, dy)

    tape.record(backward)
    return y
