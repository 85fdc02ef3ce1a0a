"""Host-side execution engine: NHWC activation views, a flat parameter arena per network and a minimal
reverse-mode tape.  PyTorch is used only for device memory and streams; every arithmetic step is a call
into libsemseg_hip.so (include/semseg_hip.h).

Layout decisions (MI355X-first):
* activations: fp32 NHWC, channel-strided views so that Keras ``concatenate`` never copies -- producers write
  straight into their slice of the concatenated buffer (UNet_Segmentation.py:469,542-551);
* parameters: ONE flat fp32 arena per network (16-byte aligned variables, Keras layouts) with matching flat
  gradient / Adam-m / Adam-v arenas -> the optimizer is a single fused launch and the data-parallel gradient
  exchange is a single bucketed all-reduce over the arena.
"""
import ctypes
import os

import torch

from . import _lib as L

_WS = {}
_WS_RETIRED = []
# 1 (default): a network's weight-derived operands are refreshed from ONE recorded plan per step (ParamArena._refresh_batched);
# 0: layer by layer (measurement / A-B: the operands are bit-identical)
WPREP_BATCH = os.environ.get("SS_WPREP_BATCH", "1") != "0"
_WC_WINO_KINDS = (1, 2, 3, 4, 5)          # csrc/common.h SS_WC_WINO_*: cache entries of the Winograd paths


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_DEV_INDEX = None


def _stream():
    """hipStream_t of torch's CURRENT stream on this process's device (every launch goes there).  Asked thousands of times per
    step: ``torch.cuda.current_stream()`` builds a Stream object (~8 us, it was a quarter of the host time of a batch-1 step);
    the raw query is ~0.2 us.  The device index is the process's device (one process per GPU) and is looked up once."""
    global _DEV_INDEX
    if _RAW_STREAM is None:
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    if _DEV_INDEX is None:
        _DEV_INDEX = torch.cuda.current_device()
    return ctypes.c_void_p(_RAW_STREAM(_DEV_INDEX))


_STREAM_OBJS = {}


def current_stream_obj():
    """torch's CURRENT stream as a Stream object, from a table keyed by the raw handle: ``torch.cuda.current_stream()`` builds a new
    object per call (~8 us with its device-index checks; the per-GPU-batch-1 MultiResUNet step is paced by the host: 14 ms to issue
    11 ms of kernels, a fifth of it in such calls)."""
    raw = _stream().value or 0
    key = (_DEV_INDEX if _DEV_INDEX is not None else torch.cuda.current_device(), raw)      # handle 0 = the default stream of EVERY device
    obj = _STREAM_OBJS.get(key)
    if obj is None:
        obj = _STREAM_OBJS[key] = torch.cuda.current_stream()
    return obj


def stream_obj_of(device, raw):
    """The registered Stream object of a raw handle on ``device`` (side_streams / current_stream_obj register them), or None."""
    return _STREAM_OBJS.get((device.index if device.index is not None else torch.cuda.current_device(), raw or 0))


def raw_stream(stream):
    """hipStream_t of a torch Stream object as the C ABI takes it."""
    return ctypes.c_void_p(stream.cuda_stream)


_SIDE = {}


def side_streams(device, n=2):
    """``n`` side streams per device for independent kernel chains (CycleGanModel.train_step runs the A->B->A and B->A->B
    generator chains and the two discriminator chains concurrently: at small per-GPU batches a single chain cannot fill 256 CUs)."""
    key = (device.type, device.index)
    have = _SIDE.get(key, ())
    if len(have) < n:
        have = have + tuple(torch.cuda.Stream(device=device) for _ in range(n - len(have)))
        _SIDE[key] = have
        for st in have:
            _STREAM_OBJS.setdefault((st.device.index, st.cuda_stream), st)
    return have[:n]


def workspace(nbytes, device, stream=None):
    """Grow-only scratch buffer shared by all calls on a (device, stream): use is stream-ordered, so concurrent chains on
    different streams get different buffers.  stream = the raw handle (int) of the stream the caller launches on when that is not
    torch's current one."""
    key = (device.type, device.index, ((_stream().value or 0) if stream is None else stream) if device.type == "cuda" else 0)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None and stream is not None:
            # The buffer is allocated from the pool of torch's CURRENT stream but used on `stream`: kernels still running there may be
            # writing the old one (split-K partials of a weight gradient), and the caching allocator would hand its block to the next
            # current-stream allocation as soon as the last reference drops.  Tell it about the other user first.
            user = stream_obj_of(device, stream)
            if user is not None:
                buf.record_stream(user)
            else:                    # a stream nobody registered: keep the old buffer alive for good (grow-only, a few per process)
                _WS_RETIRED.append(buf)
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


class capture_scope:
    """While a train step is CAPTURED into a hipGraph (torch.cuda.graph): the per-stream scratch of this module -- workspaces, amax slot
    pools -- is taken from fresh tables, so that everything the graph's kernels address was allocated inside the capture, from the
    graph's private memory pool, and the zeroing of the slot pools is part of the graph (slots are raised by atomic maxima: every
    replay must start from zero).  The eager tables come back afterwards; the captured ones stay referenced from `kept` (the owner of
    the graph holds it), so nothing the graph addresses is ever regrown or recycled under it."""

    def __enter__(self):
        global _WS, _SLOT_POOLS
        self._saved = (_WS, _SLOT_POOLS)
        _WS, _SLOT_POOLS = {}, {}
        self.kept = None
        return self

    def __exit__(self, *exc):
        global _WS, _SLOT_POOLS
        self.kept = (_WS, _SLOT_POOLS)
        _WS, _SLOT_POOLS = self._saved
        return False


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class KernelTimer:
    """HIP-event timing of tagged kernel launches on the current stream (bench.py roofline leg)."""

    def __init__(self):
        self.enabled = False
        self.events = []

    def start(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def stop(self, e0, tag, units=1):
        """``units`` = work units of this launch (samples in the batch): launches of one tag may differ in size."""
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.events.append((tag, e0, e1, units))

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for tag, e0, e1, units in self.events:
            n, tot, u = out.get(tag, (0, 0.0, 0))
            out[tag] = (n + 1, tot + e0.elapsed_time(e1), u + units)
        self.events = []
        return {k: {"launches": n, "avg_ms": tot / n, "total_ms": tot, "units": u} for k, (n, tot, u) in out.items()}


TIMER = KernelTimer()


_SLOT_POOLS = {}
_SLOT_POOL_SIZE = 1024
_SLOT_WORDS = 1024          # SS_AMAX_SLOT_BYTES / 4: 16 stripes, one per 256-byte line (include/semseg_hip.h)


def _amax_slot(device):
    """One zeroed amax slot from a pool of the CURRENT stream (one 4 MiB memset per 1024 slots instead of a fill per activation; a
    pool is zeroed on the stream that first uses it, so its slots are only handed to work on that stream).  Slots are never
    reused: a pool lives as long as any activation holds one of its slots."""
    key = (device, _stream().value or 0)
    pool = _SLOT_POOLS.get(key)
    if pool is None or pool[1] >= _SLOT_POOL_SIZE:
        pool = _SLOT_POOLS[key] = [torch.zeros(_SLOT_POOL_SIZE * _SLOT_WORDS, dtype=torch.int32, device=device), 0]
    i = pool[1]
    pool[1] = i + 1
    return pool[0][i * _SLOT_WORDS:(i + 1) * _SLOT_WORDS]


def zero_(t):
    """Zero a device tensor on the current stream through the library (no torch kernel on the product path)."""
    L.check(L.load().ss_zero(ctypes.c_void_p(t.data_ptr()), t.numel() * t.element_size(), _stream()), "ss_zero")
    return t


def zeros_like(t):
    return zero_(torch.empty_like(t))


def cat_batch(tensors):
    """Concatenate NHWC tensors of one shape family along the batch axis (the batched generator / discriminator passes of the
    CycleGAN step): one allocation + one library copy per part instead of torch.cat."""
    n = sum(int(x.shape[0]) for x in tensors)
    out = torch.empty((n,) + tuple(tensors[0].shape[1:]), dtype=tensors[0].dtype, device=tensors[0].device)
    lib, dt = L.load(), L.dtype_of(tensors[0])
    off = 0
    for x in tensors:
        assert x.is_contiguous() and x.shape[1:] == tensors[0].shape[1:] and x.dtype == tensors[0].dtype
        k = int(x.shape[0])
        if k:
            c = int(x.shape[-1])
            L.check(lib.ss_copy_t(dt, ctypes.c_void_p(x.data_ptr()), c, ctypes.c_void_p(out[off:off + k].data_ptr()), c, x.numel() // c, c, _stream()),
                    "ss_copy")
        off += k
    return out


class Act:
    """An NHWC activation view: channels [c0, c0+c) of a base tensor [n,h,w,cs]."""

    __slots__ = ("t", "c0", "c", "grad", "grad_init", "requires_grad", "parent", "amax", "amax_valid", "amax_dirty", "dt", "stats")

    def __init__(self, t, c0=0, c=None, requires_grad=True, parent=None):
        assert t.dim() == 4 and t.is_contiguous()
        self.dt = L.dtype_of(t)          # ss_dtype of the storage: fp32 (default) or bf16 / fp16 mixed-precision storage
        self.t, self.c0 = t, c0
        self.c = t.shape[3] - c0 if c is None else c
        self.grad = None
        self.grad_init = False
        self.requires_grad = requires_grad
        self.parent = parent
        self.amax = None          # amax slot: bit pattern of max|view| once a producer / conv pass has computed it (x3h scale)
        self.amax_valid = False
        self.amax_dirty = False   # the slot may hold a STALE maximum (invalidated after a producer reported one): a pass clears it before scanning
        self.stats = None         # (tensor, chunks per sample): partial (sum, sum of squares) per sample and channel from the producing conv

    # geometry
    @property
    def n(self): return self.t.shape[0]
    @property
    def h(self): return self.t.shape[1]
    @property
    def w(self): return self.t.shape[2]
    @property
    def cs(self): return self.t.shape[3]
    @property
    def rows(self): return self.t.shape[0] * self.t.shape[1] * self.t.shape[2]
    @property
    def device(self): return self.t.device
    @property
    def ptr(self): return ctypes.c_void_p(self.t.data_ptr() + self.t.element_size() * self.c0)
    @property
    def dtype(self): return self.t.dtype

    def amax_slot(self):
        """Caller-owned slot for ss_conv_desc.x_amax / dy_amax: the first conv pass that needs this tensor's maximum leaves it
        here, later passes over the same (unchanged) tensor reuse it instead of scanning the tensor again."""
        if self.amax is None:
            self.amax = _amax_slot(self.t.device)
        return ctypes.c_void_p(self.amax.data_ptr())

    def amax_state(self):
        """ss_conv_desc::*_amax_valid for this view's slot: 1 = holds the maximum, 2 = zero (fresh from the zeroed pool, never written:
        the pass scans into it without a memset dispatch), 0 = stale contents (the pass clears it first)."""
        return 1 if self.amax_valid else (0 if self.amax_dirty else 2)

    def amax_invalidate(self):
        """The tensor changes under a reported maximum (a second writer accumulates into it)."""
        if self.amax_valid or self.amax is not None:
            self.amax_dirty = True
        self.amax_valid = False

    @staticmethod
    def empty(n, h, w, c, device, requires_grad=True, dtype=torch.float32):
        return Act(torch.empty((n, h, w, c), dtype=dtype, device=device), requires_grad=requires_grad)

    def like(self, n=None, h=None, w=None, c=None, requires_grad=True):
        """A new activation of this one's storage type and device (dimensions default to this one's)."""
        return Act.empty(self.n if n is None else n, self.h if h is None else h, self.w if w is None else w,
                         self.c if c is None else c, self.device, requires_grad, self.t.dtype)

    def slice(self, c0, c):
        """Channel slice sharing storage (and, lazily, gradient storage) with this buffer."""
        return Act(self.t, self.c0 + c0, c, self.requires_grad, parent=self if self.parent is None else self.parent)

    def dense(self):
        """Contiguous torch tensor of this view (copy when strided) -- for inspection / tests."""
        return self.t[..., self.c0:self.c0 + self.c].contiguous()

    def grad_target(self):
        """(grad view to write, accumulate flag) for a backward op producing d loss / d self."""
        if self.parent is not None:
            p = self.parent
            if p.grad is None:
                p.grad = Act(zeros_like(p.t), requires_grad=False)
                p.grad_init = True
            elif not p.grad_init:
                zero_(p.grad.t)
                p.grad_init = True
            p.grad.amax_invalidate()
            return Act(p.grad.t, self.c0, self.c, False), 1
        if self.grad is None:
            self.grad = Act(torch.empty_like(self.t), self.c0, self.c, False)
            self.grad_init = True
            return self.grad, 0
        if not self.grad_init:
            self.grad_init = True
            self.grad.amax_invalidate()
            return self.grad, 0
        self.grad.amax_invalidate()          # a second writer accumulates: a maximum reported by the first one is stale
        return self.grad, 1

    def get_grad(self):
        """Gradient accumulated so far (None when nothing flowed into this activation)."""
        if self.parent is not None:
            p = self.parent
            if p.grad is None or not p.grad_init:
                return None
            return Act(p.grad.t, self.c0, self.c, False)
        return self.grad if self.grad_init else None


class DeferredNorm(Act):
    """The output of a normalisation layer whose APPLY pass has not run: only the statistics exist (layers.Norm(..., defer=True)).
    A convolution that can normalise in its operand load (ss_conv2d_fuses_in_norm: the second half of "fused InstanceNorm + conv",
    CycleGAN.py:327-333) reads `pre` with `in_norm()`; any other consumer touching `.ptr` / `.t` gets the tensor materialised by
    ss_norm_apply first (same arithmetic, so both routes give the same bits).  Gradients flow into this object like into any
    activation: the norm's backward closure reads them."""

    __slots__ = ("pre", "mean", "rstd", "gamma", "beta", "groups", "act_code", "act_alpha", "_y", "_materialize")

    def __init__(self, pre, mean, rstd, gamma, beta, groups, act_code, act_alpha, materialize):
        # geometry / storage type of the (future) normalised tensor = the pre-norm view's; dense output
        self.dt = pre.dt
        self.t = None
        self.c0, self.c = 0, pre.c
        self.grad, self.grad_init, self.requires_grad, self.parent = None, False, True, None
        self.amax, self.amax_valid, self.amax_dirty, self.stats = None, False, False, None
        self.pre, self.mean, self.rstd, self.gamma, self.beta = pre, mean, rstd, gamma, beta
        self.groups, self.act_code, self.act_alpha = groups, act_code, act_alpha
        self._y, self._materialize = None, materialize

    @property
    def materialized(self):
        return self._y is not None

    def tensor(self):
        if self._y is None:
            self._y = self._materialize(self)
            self.t = self._y.t
        return self._y

    # geometry without materialising
    @property
    def n(self): return self.pre.n
    @property
    def h(self): return self.pre.h
    @property
    def w(self): return self.pre.w
    @property
    def cs(self): return self.pre.c if self._y is None else self._y.cs
    @property
    def rows(self): return self.pre.rows
    @property
    def device(self): return self.pre.device
    @property
    def dtype(self): return self.pre.dtype
    @property
    def ptr(self): return self.tensor().ptr

    def like(self, n=None, h=None, w=None, c=None, requires_grad=True):
        return Act.empty(self.n if n is None else n, self.h if h is None else h, self.w if w is None else w,
                         self.c if c is None else c, self.device, requires_grad, self.pre.t.dtype)

    def amax_slot(self):
        """Slot for max|NORMALISED tensor| (the fused forward pass fills it, the weight gradient reads it)."""
        if self.amax is None:
            self.amax = _amax_slot(self.device)
        return ctypes.c_void_p(self.amax.data_ptr())

    def slice(self, c0, c):
        return self.tensor().slice(c0, c)

    def dense(self):
        return self.tensor().dense()

    def grad_target(self):
        if self.grad is None:
            self.grad = Act(torch.empty((self.n, self.h, self.w, self.c), dtype=self.pre.t.dtype, device=self.device), requires_grad=False)
            self.grad_init = True
            return self.grad, 0
        if not self.grad_init:
            self.grad_init = True
            self.grad.amax_invalidate()
            return self.grad, 0
        self.grad.amax_invalidate()
        return self.grad, 1


def convert(act, dtype):
    """A copy of the activation view in another storage type (ss_convert): the fp32 <-> bf16 / fp16 boundary of a network."""
    if act.t.dtype == dtype:
        return act
    out = Act.empty(act.n, act.h, act.w, act.c, act.device, act.requires_grad, dtype)
    L.check(L.load().ss_convert(act.ptr, act.dt, act.cs, out.ptr, out.dt, out.cs, act.rows, act.c, _stream()), "ss_convert")
    return out


class Tape:
    """Records backward closures in forward order; ``backward()`` replays them in reverse."""

    def __init__(self, enabled=True, param_grads=True, count_uses=True):
        self.ops = []
        self.enabled = enabled
        self.param_grads = param_grads   # False: ops recorded now skip weight gradients (input grads only)
        self.count_uses = count_uses     # False (recomputation tapes): the uses of the variables were announced at the first forward
        # Optional side stream for the WEIGHT gradients of the backward pass (layers.Conv2D): they are off the dependency chain
        # (nothing in backward reads dW), so a convolution whose weight-gradient pass touches no shared amax slot launches it there,
        # ordered behind the chain by an event, and the chain goes on with the data gradient.  The owner of the tape joins the
        # stream before the optimizer step (join_wgrad_stream).
        self.wgrad_stream = None
        # Independent branches of the graph on streams of their own (Tape.fork / Branch): `lane` is the branch being recorded (its
        # backward closures are replayed on its stream), `branch_streams` what a network may fork onto (None: run everything inline)
        self.lane = None
        self.branch_streams = None

    def record(self, fn):
        if self.enabled:
            lane = self.lane
            if lane is not None:
                fn._lane_stream = lane.stream          # replayed on the branch's stream (closures are fresh function objects)
            self.ops.append(fn)

    def fork(self, stream):
        """A Branch of the graph that runs on ``stream`` beside the rest (see Branch).  Call where the branch's inputs are final
        and every other consumer of them has been recorded."""
        return Branch(self, stream)

    def backward(self):
        # consecutive closures of one branch replay inside ONE stream context (entering / leaving torch's context costs ~25 us:
        # per closure it was 1.3 ms of host time per MultiResUNet step, which is host-bound at per-GPU batch 1)
        ctx, ctx_stream = None, None
        try:
            for fn in reversed(self.ops):
                st = getattr(fn, "_lane_stream", None)
                if st is not ctx_stream:
                    if ctx is not None:
                        ctx.__exit__(None, None, None)
                        ctx = None
                    if st is not None:
                        ctx = torch.cuda.stream(st)
                        ctx.__enter__()
                    ctx_stream = st
                fn()
        finally:
            if ctx is not None:
                ctx.__exit__(None, None, None)
        self.ops = []
        self.join_wgrad_stream()

    def join_wgrad_stream(self):
        """The current stream waits for everything issued on the weight-gradient side stream."""
        if self.wgrad_stream is not None:
            current_stream_obj().wait_stream(self.wgrad_stream)


class Branch:
    """An independent branch of the recorded graph on a HIP stream of its own, in the forward pass AND in the replay (the
    MultiResUNet's ResPaths, UNet_Segmentation.py:476-503,533-553: each depends on one encoder block's output only and is needed
    only where the decoder concatenates it, so it runs beside the whole deeper part of the network -- in backward beside the
    backward of everything between the concatenation and that encoder block).  Three points of the forward program:

        br = tape.fork(stream)    # A: the branch's inputs are final; every OTHER consumer of them has been recorded
        with br: ...              #    the branch: its kernels go to `stream` (ordered behind A), its closures replay there
        br.join()                 # B: before the first consumer of the branch's outputs

    Replay (reverse order): the op recorded at B marks "the branch's output gradients are final" (event on the replaying stream);
    the branch's closures run on `stream` behind that event, whenever the reverse order reaches them, and leave an event; the op
    recorded at A makes the replaying stream wait for it -- so the branch writes the gradient of its input FIRST (it is replayed
    before A) and the other consumers, recorded before A, accumulate behind the wait.  Scratch buffers, amax slot pools and
    weight-cache fills are per stream / event-ordered already (workspace, _amax_slot, layers.Conv2D._attach_wcache).  Tensors
    the branch allocates live on its stream's allocator pool; the ones that cross (the input gradient it writes first, buffers of
    the forking stream it reads or writes) stay alive until the tape is dropped, after the joins."""

    def __init__(self, tape, stream):
        self.tape, self.stream = tape, stream
        self.inputs_ready = current_stream_obj().record_event()
        self.fwd_done = self.grads_ready = self.bwd_done = None
        if tape.enabled:
            def join_backward():          # replayed right before the other consumers' closures (recorded before A)
                if self.bwd_done is not None:
                    current_stream_obj().wait_event(self.bwd_done)
            tape.ops.append(join_backward)

    def __enter__(self):
        assert self.tape.lane is None, "branches do not nest"
        self.stream.wait_event(self.inputs_ready)
        if self.tape.enabled:
            def end_backward():           # replayed after the branch's closures
                self.bwd_done = self.stream.record_event()
            self.tape.ops.append(end_backward)
        self.tape.lane = self
        self._ctx = torch.cuda.stream(self.stream)
        self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        self._ctx.__exit__(*exc)
        self.tape.lane = None
        self.fwd_done = self.stream.record_event()
        if self.tape.enabled:
            def begin_backward():         # replayed before the branch's closures: behind the gradients of its outputs
                if self.grads_ready is not None:
                    self.stream.wait_event(self.grads_ready)
                else:
                    self.stream.wait_stream(current_stream_obj())
            self.tape.ops.append(begin_backward)
        return False

    def join(self):
        current_stream_obj().wait_event(self.fwd_done)
        if self.tape.enabled:
            def mark_grads_ready():       # replayed right after the closures of the outputs' consumers (recorded after B)
                self.grads_ready = current_stream_obj().record_event()
            self.tape.ops.append(mark_grads_ready)


class ParamArena:
    """Flat fp32 storage for a network's variables (Keras layouts, creation order)."""

    ALIGN = 4  # floats (16 bytes)

    def __init__(self, device):
        self.device = device
        self.specs = []      # (name, shape, trainable, offset)
        self.n_train = 0
        self.n_state = 0
        self.params = None   # trainable arena
        self.grads = None
        self.m = None
        self.v = None
        self.state = None    # non-trainable arena (BN moving statistics)
        self.views = {}
        self.gviews = {}
        # gradient-ready tracking for the overlapped data-parallel all-reduce (dist.py): a variable's gradient is final
        # once every recorded backward op that accumulates into it has run; buckets fire when all their variables are.
        self.pending = {}
        self.buckets = []          # dicts: start, end (element offsets into grads), names, remaining, fired
        self.bucket_of = {}
        self.grad_hook = None      # callable(flat_grad_slice) -> async work handle, set by dist.enable_overlap()
        self.works = []
        self.version = 0           # bumped by every writer of `params` that torch's version counter does not see (HIP kernels)
        self.derived = []          # layers that keep operands derived from the weights (layers.Conv2D weight caches)
        self._derived_key = None
        self._refresh_stream = None   # side stream of a refresh_derived() whose kernels may still be reading the weights
        self._wprep_plan = None       # recorded plan of the batched refresh (_refresh_batched)
        self._wprep_unbatchable = None

    def touch(self):
        """The weight values changed behind torch's back (optimizer kernel, collective): caches derived from them are stale."""
        self.version += 1

    def refresh_derived(self, side=None):
        """Recompute what the layers keep of the previous weight version (no-op while the weights are unchanged).  A step that
        runs concurrent kernel chains calls this before forking: otherwise the chain that reaches a layer second has to wait for
        the first one's (much later) fill.  side = None: on the CURRENT stream (one event for the whole arena).  side = a stream:
        the ~100 small launches of a network run THERE, in layer order, behind everything issued so far, with one event per layer
        -- a consumer waits for its own layer's operands only (layers.Conv2D._attach_wcache), so the chains start at once and the
        rest of the refresh runs beside their first layers instead of in front of them."""
        if self.params is None:
            return
        from . import _lib as L
        key = (self.weights_key(), L.CONFIG_EPOCH)
        if key == self._derived_key:
            return
        self._derived_key = key
        if WPREP_BATCH and self.device.type == "cuda" and self._refresh_batched(side):
            return
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                sid = _stream().value or 0
                for layer in self.derived:
                    sync = dict(ev=None, synced=set())
                    if layer.refresh_wcache(sync):
                        ev = torch.cuda.Event()
                        ev.record()
                        sync["ev"] = ev
                        sync["synced"].add(sid)
            self._refresh_stream = side
            return
        sync = dict(ev=None, synced=set())
        n = sum(layer.refresh_wcache(sync) for layer in self.derived)
        if n:
            ev = torch.cuda.Event()
            ev.record()
            sync["ev"] = ev
            sync["synced"].add(_stream().value or 0)

    def _refresh_batched(self, side):
        """The same refresh as ONE recorded plan (include/semseg_hip.h ss_wprep_*): the layers' fill-only calls are recorded once --
        every operand they would launch becomes a job -- and every later refresh replays the plan as one launch per kind of operand
        (weight maxima, tap-wise transposes, split planes, Winograd planes: ~6 launches instead of ~100 per network).  The plan is keyed
        on the configuration and on the layers' cache users (a new descriptor geometry adds operands: recorded again); a recording
        the library calls incomplete (an operand kind without a recorder hook, e.g. the x6 / fp32 arithmetic modes) is executed once
        and the key is remembered as unbatchable.  Returns False when the per-layer path has to run."""
        from . import _lib as L
        lib = L.load()
        sig = (L.CONFIG_EPOCH, tuple(len(l._wc["users"]) if getattr(l, "_wc", None) else 0 for l in self.derived))
        if sig == self._wprep_unbatchable or not any(sig[1]):
            return False
        stream = side if side is not None else current_stream_obj()
        if side is not None:
            side.wait_stream(current_stream_obj())
        plan = self._wprep_plan
        ver = (self.weights_key(), L.CONFIG_EPOCH)
        with torch.cuda.stream(stream):
            if plan is None or plan["sig"] != sig:
                L.check(lib.ss_wprep_record_begin(), "ss_wprep_record_begin")
                try:
                    dummy = dict(ev=None, synced=set())
                    for layer in self.derived:
                        layer.refresh_wcache(dummy)          # recorded, not launched (operands without a hook launch at once)
                finally:
                    nbytes, njobs, complete = ctypes.c_size_t(0), ctypes.c_int32(0), ctypes.c_int32(0)
                    L.check(lib.ss_wprep_record_end(ctypes.byref(nbytes), ctypes.byref(njobs), ctypes.byref(complete)), "ss_wprep_record_end")
                host = (ctypes.c_char * max(nbytes.value, 1))()
                L.check(lib.ss_wprep_plan_write(ctypes.cast(host, ctypes.c_void_p), nbytes.value), "ss_wprep_plan_write")
                dev = torch.frombuffer(host, dtype=torch.uint8).to(self.device)
                plan = dict(sig=sig, host=host, dev=dev, bytes=nbytes.value, jobs=njobs.value)
                if complete.value and njobs.value:
                    self._wprep_plan = plan
                else:
                    self._wprep_plan, self._wprep_unbatchable = None, sig
            else:
                for layer in self.derived:                   # the directories stay as the recording built them: only the version moves
                    st = getattr(layer, "_wc", None)
                    if st:
                        st["ver"] = ver
            # two events per arena: the strided / transposed / 4 x 4 layers' operands (part 0: maxima, transposes, split planes) are ready
            # before the trunk's Winograd planes (part 1, ~2/3 of the plan's time) -- the chains' first layers need not wait for those
            hp, dp = ctypes.cast(plan["host"], ctypes.c_void_p), ctypes.c_void_p(plan["dev"].data_ptr())
            syncs = []
            for part in (0, 1):
                if plan["jobs"]:
                    L.check(lib.ss_wprep_run_part(hp, dp, plan["bytes"], part, _stream()), "ss_wprep_run_part")
                ev = torch.cuda.Event()
                ev.record()
                syncs.append(dict(ev=ev, synced={_stream().value or 0}))
            for layer in self.derived:
                st = getattr(layer, "_wc", None)
                if st:
                    c = st["c"]
                    wino = any((int(c.entry[i].tag) & 0xff) in _WC_WINO_KINDS for i in range(c.count))
                    st["sync"] = syncs[1] if wino else syncs[0]
        if side is not None:
            self._refresh_stream = side
        return True

    def join_refresh(self):
        """Order the current stream behind a refresh_derived(side=...) still in flight: called by whoever WRITES the weights next
        (the refresh kernels read them)."""
        side = self._refresh_stream
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
            self._refresh_stream = None

    def weights_key(self):
        """Changes whenever the trainable values may have changed: explicit touch() or any in-place torch op on a view of them."""
        return (self.version, self.params._version)

    def declare(self, name, shape, trainable=True):
        size = 1
        for s in shape:
            size *= s
        if trainable:
            off = self.n_train
            self.n_train = off + (size + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        else:
            off = self.n_state
            self.n_state = off + (size + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.specs.append((name, tuple(shape), trainable, off))

    def materialize(self):
        dev = self.device
        self.params = torch.zeros(max(self.n_train, 4), dtype=torch.float32, device=dev)
        self.grads = torch.zeros_like(self.params)
        self.m = torch.zeros_like(self.params)
        self.v = torch.zeros_like(self.params)
        self.state = torch.zeros(max(self.n_state, 4), dtype=torch.float32, device=dev)
        for name, shape, trainable, off in self.specs:
            size = 1
            for s in shape:
                size *= s
            if trainable:
                self.views[name] = self.params[off:off + size].view(shape)
                self.gviews[name] = self.grads[off:off + size].view(shape)
            else:
                self.views[name] = self.state[off:off + size].view(shape)

        # buckets: consecutive variables (creation order) up to BUCKET_ELEMS each
        cur = None
        for name, shape, trainable, off in self.specs:
            if not trainable:
                continue
            size = 1
            for s_ in shape:
                size *= s_
            end = off + (size + self.ALIGN - 1) // self.ALIGN * self.ALIGN
            if cur is None or end - cur["start"] > self.BUCKET_ELEMS:
                cur = dict(start=off, end=end, names=[name], remaining=0, fired=False, active=False)
                self.buckets.append(cur)
            else:
                cur["end"] = end
                cur["names"].append(name)
            self.bucket_of[name] = cur

    BUCKET_ELEMS = 8 * 1024 * 1024      # 32 MiB fp32 buckets (xGMI: few large messages)

    def note_use(self, names):
        """A backward op that accumulates into these variables' gradients was recorded."""
        for n in names:
            self.pending[n] = self.pending.get(n, 0) + 1

    def begin_backward(self):
        """Call right before replaying a tape: arms the buckets whose variables will receive gradients."""
        self.works = []
        for b in self.buckets:
            b["remaining"] = sum(1 for n in b["names"] if self.pending.get(n, 0) > 0)
            b["active"] = b["remaining"] > 0
            b["fired"] = False
            b["merged"] = False
            b["events"] = {}

    def note_done(self, names):
        """The backward op accumulating into these variables has been ENQUEUED on the current stream.  With a gradient hook installed
        (data parallel) the kernels writing one bucket may sit on several streams -- the chain, the weight-gradient side stream
        (layers.Conv2D), a branch lane, the second chain of the dual-stream CycleGAN step -- and none of them waits for the others
        before the end of backward.  So every call leaves an event on its stream (per bucket and stream: the latest), and the
        stream that completes the bucket waits for the other streams' events before the exchange is launched behind it."""
        pending, bucket_of = self.pending, self.bucket_of
        if self.grad_hook is None:
            for n in names:
                c = pending.get(n, 0) - 1
                pending[n] = c
                if c == 0:
                    bucket_of[n]["remaining"] -= 1
            return
        touched = {}
        for n in names:
            c = pending.get(n, 0) - 1
            pending[n] = c
            bk = bucket_of[n]
            if c == 0:
                bk["remaining"] -= 1
            touched[id(bk)] = bk
        st = current_stream_obj() if self.device.type == "cuda" else None
        for bk in touched.values():
            if not bk["active"] or bk["fired"]:
                continue
            if bk["remaining"] == 0:
                if st is not None:
                    for other, ev in bk["events"].items():
                        if other is not st:
                            st.wait_event(ev)
                bk["fired"] = True
                self.works.append(self._fire(bk))
            elif st is not None:
                # one reusable event per (bucket, stream), re-recorded: a fresh event per call was several hundred creations per step
                pool = bk.setdefault("event_pool", {})
                ev = pool.get(st)
                if ev is None:
                    ev = pool[st] = torch.cuda.Event()
                ev.record(st)
                bk["events"][st] = ev

    def _fire(self, b):
        """Launch the exchange of one finished bucket behind the current stream.  Dual-chain steps (CycleGAN._train_step_dual) set
        `merge_into` / `merge_from`: the second chain's share of the bucket is added to the first chain's buffer here, bucket by
        bucket, instead of over the whole arena after both chains have joined."""
        dst, src = self.grads, getattr(self, "merge_from", None)
        if src is not None:
            dst = self.merge_into
            self._merge_bucket(dst, src, b)
            b["merged"] = True
        return self.grad_hook(dst[b["start"]:b["end"]])

    def _merge_bucket(self, dst, src, b):
        """dst[bucket] += src[bucket] on the current stream."""
        off, n = b["start"] * 4, b["end"] - b["start"]
        L.check(L.load().ss_axpby(1.0, ctypes.c_void_p(dst.data_ptr() + off), 1, 1.0, ctypes.c_void_p(src.data_ptr() + off), 1,
                                  ctypes.c_void_p(dst.data_ptr() + off), 1, n, 1, _stream()), "axpby (bucket merge)")

    def __getitem__(self, name):
        return self.views[name]

    def grad(self, name):
        return self.gviews[name]

    # ---- second gradient buffer: two backward chains running on different streams must not accumulate into the same memory ----
    def _alt(self):
        if getattr(self, "grads_alt", None) is None:
            self.grads_alt = torch.zeros_like(self.grads)
            self.gviews_alt = {}
            for name, shape, trainable, off in self.specs:
                if trainable:
                    size = 1
                    for s_ in shape:
                        size *= s_
                    self.gviews_alt[name] = self.grads_alt[off:off + size].view(shape)
        return self.grads_alt

    def swap_grads(self):
        """Exchange the roles of the main and the alternate gradient buffer (host-side pointer swap)."""
        self._alt()
        self.grads, self.grads_alt = self.grads_alt, self.grads
        self.gviews, self.gviews_alt = self.gviews_alt, self.gviews

    def zero_grad_alt(self):
        lib = L.load()
        L.check(lib.ss_fill(_p(self._alt()), 0.0, self.grads_alt.numel(), _stream()), "ss_fill")

    def merge_alt_grads(self):
        """grads += grads_alt (on the current stream), except for the buckets `_fire` has merged during backward already."""
        lib = L.load()
        if not any(b.get("merged") for b in self.buckets):
            n = self.grads.numel()
            L.check(lib.ss_axpby(1.0, _p(self.grads), 1, 1.0, _p(self.grads_alt), 1, _p(self.grads), 1, n, 1, _stream()), "axpby")
            return
        for b in self.buckets:
            if not b.get("merged"):
                self._merge_bucket(self.grads, self.grads_alt, b)
                b["merged"] = True

    def zero_grad(self):
        lib = L.load()
        L.check(lib.ss_fill(_p(self.grads), 0.0, self.grads.numel(), _stream()), "ss_fill")
