"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL/xGMI (backend "nccl" on ROCm),
gloo for the CPU tests.  The reference has no working multi-GPU path on the torch backend: the
design here is new, with single-device semantics as the parity target.

Tiles are independent in every CycleGAN layer (InstanceNorm is per-sample), so the only exchange step is the
gradient all-reduce: each network keeps ONE flat gradient arena, reduced in a few large buckets (xGMI is
point-to-point, 7 links x ~153 GB/s per GPU -> few large messages, not many small ones).  Losses are means over
the GLOBAL batch: gradients are summed over ranks and scaled by 1/world inside the fused Adam kernel.
"""
import contextlib
import os

import numpy as np
import torch
import torch.distributed as dist

BUCKET_ELEMS = 8 * 1024 * 1024      # 32 MiB fp32 buckets: the size engine.ParamArena.BUCKET_ELEMS cuts the gradient arenas into

_SOLO = 0          # depth of `solo()` sections: this rank works alone, every collective of this module is skipped


def is_dist():
    return dist.is_available() and dist.is_initialized()


def world_size():
    return dist.get_world_size() if is_dist() and not _SOLO else 1


def rank():
    return dist.get_rank() if is_dist() else 0


@contextlib.contextmanager
def solo():
    """Work that only ONE rank executes (the file-producing workflow steps: StartProcess.Workflow.run_step) must not issue
    collectives -- the other ranks are already waiting in the barrier behind the step, and a broadcast from a freshly built
    model (`broadcast_params` in every `create_model`) against that barrier hangs or corrupts the group.  Inside this section
    `world_size()` is 1, so every collective here (broadcast, gradient / SyncBN / metric all-reduce) is a no-op and data
    loaders do not shard; `rank()` keeps its value."""
    global _SOLO
    _SOLO += 1
    try:
        yield
    finally:
        _SOLO -= 1


def local_device():
    if torch.cuda.is_available():
        return torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
    return torch.device("cpu")


def init_from_env(backend=None):
    """Initialise the process group from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun contract)."""
    if is_dist() or int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return
    if backend is None:
        backend = os.environ.get("SS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        torch.cuda.set_device(local_device())
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend=backend)


# MEASUREMENT ONLY (bench.py `multi_gpu.exposed_comm_ms`): skip every data-path collective, so that a step's time without its
# exchange can be subtracted from the real one.  Results are wrong while it is set (ranks drift apart); never set it in training.
SKIP_COLLECTIVES = False


class _Done:
    def wait(self):
        return True


def _all_reduce(t, async_op=False):
    if SKIP_COLLECTIVES:
        return _Done()
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op)


def all_reduce_flat(flat, bucket_elems=BUCKET_ELEMS):
    """Sum ``flat`` (1-D tensor) over ranks in large buckets, all launched before any is waited for."""
    if world_size() == 1:
        return
    works = []
    n = flat.numel()
    for off in range(0, n, bucket_elems):
        works.append(_all_reduce(flat[off:min(off + bucket_elems, n)], async_op=True))
    for w in works:
        w.wait()


def enable_overlap(nets):
    """Overlap the gradient exchange with backward: every 32 MiB bucket of a network's flat gradient arena is all-reduced
    (asynchronously, on the communication stream) as soon as the last backward op touching one of its variables has been
    enqueued; `all_reduce_grads` then only waits.  No-op for a single process."""
    if world_size() == 1:
        return
    for net in nets:
        net.arena.grad_hook = lambda flat: _all_reduce(flat, async_op=True)


def begin_backward(nets):
    for net in nets:
        net.arena.begin_backward()


def ranks_share_device():
    """True when several ranks of this node run on one GPU (CPU-less test rigs: 2 ranks on 1 device).  Multi-stream execution
    is switched off there: two processes x three streams oversubscribe the hardware queues and the GPU falls into wave
    context-switch thrashing (measured: 16-24 s per step instead of 0.56 s)."""
    local = int(os.environ.get("LOCAL_WORLD_SIZE", world_size()))
    return torch.cuda.is_available() and local > torch.cuda.device_count()


def begin_all_reduce_grads(nets):
    """Launch (asynchronously) whatever part of these networks' gradient exchange has not been launched during backward and
    return the outstanding work handles; `finish_all_reduce_grads` waits.  Lets the caller put independent work (the CycleGAN
    discriminator phase) between the two."""
    works = []
    if world_size() == 1:
        for net in nets:
            net.arena.pending = {}
        return works
    for net in nets:
        a = net.arena
        if a.grad_hook is None:
            n = a.grads.numel()
            step = 64 * 1024 * 1024 // 4
            works += [_all_reduce(a.grads[off:min(off + step, n)], async_op=True) for off in range(0, n, step)]
        else:
            works += a.works
            works += [_all_reduce(a.grads[b["start"]:b["end"]], async_op=True)
                      for b in a.buckets if b["active"] and not b["fired"]]
        a.works = []
        a.pending = {}
    return works


def finish_all_reduce_grads(works):
    for w in works:
        w.wait()


def all_reduce_grads(nets):
    """Finish the gradient exchange of these networks: wait for the buckets launched during backward and reduce any
    bucket that has not been sent (overlap disabled, or a bucket whose variables were not all used)."""
    finish_all_reduce_grads(begin_all_reduce_grads(nets))


def broadcast_params(nets, src=0):
    """Replicas start from rank 0's weights (and BN state)."""
    if world_size() == 1:
        return
    for net in nets:
        dist.broadcast(net.arena.params, src)
        dist.broadcast(net.arena.state, src)
        getattr(net.arena, "touch", lambda: None)()      # derived-operand caches (layers.Conv2D) are stale now


def mean_scalars(values):
    """Average a small numpy vector of per-rank scalar means over ranks (equal per-rank batch sizes)."""
    if world_size() == 1 or SKIP_COLLECTIVES:
        return values
    dev = local_device() if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.as_tensor(np.asarray(values, dtype=np.float64), device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return (t / world_size()).cpu().numpy()


def enable_sync_bn(enabled=True):
    """BatchNorm statistics over the GLOBAL batch (SyncBN): needed for the data-parallel MultiResUNet to reproduce the
    single-device reference, which normalises with whole-batch statistics.  Per layer: one all-reduce of
    2*C floats forward and one backward (latency-bound; 85 layers)."""
    from . import layers

    def _allreduce(t):
        if world_size() > 1:
            _all_reduce(t)
        return world_size()

    layers.SYNC_BN = _allreduce if enabled else None


def check_batch_divisible(batch_size, world, what="batch_size"):
    """Data parallel splits every global batch into ``world`` equal contiguous shards (losses are means over the global batch,
    gradients are summed and scaled by 1/world): a batch that does not divide would silently drop tiles or hand some ranks an
    empty shard while the others block in a collective.  Raise up front instead."""
    if world > 1 and (batch_size < world or batch_size % world != 0):
        raise ValueError(f"{what} = {batch_size} cannot be split evenly over {world} ranks: choose a multiple of {world} "
                         f"(the reference defaults 2 / 5 are single-device values)")


_WARNED = set()


def warn_once(msg):
    if msg not in _WARNED and rank() == 0:
        _WARNED.add(msg)
        import warnings
        warnings.warn(msg, stacklevel=2)
