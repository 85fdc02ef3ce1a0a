"""Data-parallel semantics on the GPU: 2 ranks (gloo transport, both on cuda:0 of the 1-GPU box) training on halves of a
global batch must reproduce the single-process step on the whole batch (InstanceNorm is per-sample; BatchNorm via SyncBN;
gradients summed and scaled by 1/world).  The RCCL transport itself is exercised by the driver's multi-GPU bench."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BASE = "automatic-sem-image-segmentation_amd"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
out = sys.argv[2]
B = "automatic-sem-image-segmentation_amd"
D = importlib.import_module(B + ".dist"); CG = importlib.import_module(B + ".CycleGAN"); N = importlib.import_module(B + ".nets")
OPT = importlib.import_module(B + ".optim"); UN = importlib.import_module(B + ".UNet_Segmentation")
importlib.import_module(B + ".engine").ParamArena.BUCKET_ELEMS = 4096   # several buckets even for these tiny nets
D.init_from_env("gloo")
rank, world = D.rank(), D.world_size()
dev = D.local_device()
g = torch.Generator().manual_seed(0)
a = torch.rand((4, 64, 64, 1), generator=g) * 2 - 1
b = (torch.rand((4, 64, 64, 1), generator=g) > 0.8).float() * 2 - 1
per = 4 // world
sl = slice(rank * per, (rank + 1) * per)
nets = dict(gen_a=N.ResnetGenerator(filters=4, num_residual_blocks=2, device=dev, seed=1 + 10 * rank),
            gen_b=N.ResnetGenerator(filters=4, num_residual_blocks=2, device=dev, seed=2 + 10 * rank),
            disc_a=N.PatchDiscriminator(filters=8, device=dev, seed=3 + 10 * rank), disc_b=N.PatchDiscriminator(filters=8, device=dev, seed=4 + 10 * rank))
unet = N.MultiResUNet(16, device=dev, seed=5 + 10 * rank)
D.broadcast_params(list(nets.values()) + [unet])          # rank 1 was seeded differently on purpose
D.enable_overlap(list(nets.values()) + [unet])             # bucketed all-reduce launched during backward
if world > 1:
    D.enable_sync_bn(True)
# count the generator buckets whose exchange is launched while a tape is still replaying (the overlap itself)
E = importlib.import_module(B + ".engine")
depth, fired = [0], [0]
_bw = E.Tape.backward
def bw(self):
    depth[0] += 1
    try:
        return _bw(self)
    finally:
        depth[0] -= 1
E.Tape.backward = bw
def counting(inner):
    def hook(flat):
        fired[0] += 1 if depth[0] > 0 else 0
        return inner(flat)
    return hook
for k in ("gen_a", "gen_b"):
    if nets[k].arena.grad_hook is not None:
        nets[k].arena.grad_hook = counting(nets[k].arena.grad_hook)
model = CG.CycleGanModel(nets["gen_a"], nets["gen_b"], nets["disc_a"], nets["disc_b"], image_pool_a=CG.ImagePool(2, 0), image_pool_b=CG.ImagePool(2, 0))
model.compile(OPT.Adam(2e-4, beta_1=0.5), OPT.Adam(2e-4, beta_1=0.5), OPT.Adam(2e-4, beta_1=0.5), OPT.Adam(2e-4, beta_1=0.5))
# SyncBN exchanges of one UNet step: every one goes through layers.SYNC_BN
LYm = importlib.import_module(B + ".layers")
bn_calls = [0]
if LYm.SYNC_BN is not None:
    _bn = LYm.SYNC_BN
    def counted_bn(t):
        bn_calls[0] += 1
        return _bn(t)
    LYm.SYNC_BN = counted_bn
_ms = D.mean_scalars
metric_exchanges = [0]
def counted_ms(v):
    metric_exchanges[0] += 1 if D.world_size() > 1 else 0
    return _ms(v)
D.mean_scalars = counted_ms
model.train_step((a[sl].numpy(), b[sl].numpy()))
assert metric_exchanges[0] == 0, "train_step exchanges gradients only; metrics travel once per logging interval"
m = model.global_metrics()
um = UN.UNetModel(unet, 9.0, OPT.Adam(1e-3))
u = um.global_metrics(um.train_step((((a[sl] + 1) / 2).numpy(), ((b[sl] + 1) / 2).numpy())))
unet_collectives = bn_calls[0]
assert metric_exchanges[0] == (2 if world > 1 else 0), metric_exchanges
if rank == 0:
    arrs = {f"{k}/{i}": w for k, net in nets.items() for i, w in enumerate(net.get_weights())}
    arrs.update({f"unet/{i}": w for i, w in enumerate(unet.get_weights())})
    arrs["metrics"] = np.array([m[k] for k in sorted(m)] + [u[k] for k in sorted(u)])
    arrs["gen_buckets_fired_in_backward"] = np.array(fired[0])
    arrs["gen_buckets"] = np.array(len(nets["gen_a"].arena.buckets) + len(nets["gen_b"].arena.buckets))
    arrs["unet_syncbn_exchanges"] = np.array(unet_collectives)
    np.savez(out, **arrs)
if world > 1:
    torch.distributed.barrier(); torch.distributed.destroy_process_group()
print("DONE", rank, flush=True)
'''


def _run(tmp_path, world):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    out = str(tmp_path / f"out{world}.npz")
    # SS_DUAL_STREAM=force / SS_UNET_WGRAD_STREAM=force: keep the multi-stream execution (generator buckets merged and launched from the
    # second chain's backward; UNet weight gradients on a side stream, so that a bucket's kernels sit on two streams -- ADVICE r4)
    # although both ranks share cuda:0
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29740 + world), WORLD_SIZE=str(world), SS_DUAL_STREAM="force",
               SS_UNET_WGRAD_STREAM="force")
    procs = [subprocess.Popen([sys.executable, str(script), REPO, out], env=dict(env, RANK=str(r), LOCAL_RANK="0"),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    for r, p in enumerate(procs):
        o = p.communicate(timeout=600)[0]
        assert p.returncode == 0 and f"DONE {r}" in o, o[-3000:]
    return np.load(out)


def test_two_rank_data_parallel_equals_single_process(tmp_path):
    one, two = _run(tmp_path, 1), _run(tmp_path, 2)
    np.testing.assert_allclose(two["metrics"], one["metrics"], rtol=5e-4, atol=1e-6)
    # the dual-chain step (SS_DUAL_STREAM=force) launches every generator bucket from inside the second chain's backward
    assert int(two["gen_buckets"]) >= 4 and int(two["gen_buckets_fired_in_backward"]) == int(two["gen_buckets"]), \
        (int(two["gen_buckets_fired_in_backward"]), int(two["gen_buckets"]))
    # SyncBN: 85 BatchNorm layers, 2 exchanges each unpacked (170); the 9 MultiRes blocks pack shortcut + first 3x3 both ways, the 10
    # ResPath stages pack their pair in forward: 66 forward + 76 backward
    assert int(two["unet_syncbn_exchanges"]) <= 142, int(two["unet_syncbn_exchanges"])
    num = den = 0.0
    for k in one.files:
        if k in ("metrics", "gen_buckets", "gen_buckets_fired_in_backward", "unet_syncbn_exchanges"):
            continue
        num += float(((two[k].astype(np.float64) - one[k]) ** 2).sum())
        den += float((one[k].astype(np.float64) ** 2).sum())
        assert float(np.abs(two[k] - one[k]).max()) <= 2.2e-3, k      # one Adam step: |delta| <= ~lr, both signs
    assert (num / den) ** 0.5 <= 1e-3


def test_bench_two_ranks_reports_per_rank_times_and_exposed_communication(tmp_path):
    """bench.py's N > 1 path (per-rank medians, the no-collectives leg, replica re-alignment) on 2 ranks sharing cuda:0 over gloo --
    the transport differs from the driver's RCCL run, the code path is the same."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29747", WORLD_SIZE="2", LOCAL_WORLD_SIZE="2",
               SS_DIST_BACKEND="gloo", SS_BENCH_TEST_TRANSPORT="gloo")
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--size", "64", "--global-batch", "2",
           "--filters", "8", "--no-cpu-baseline", "--tables", str(tmp_path / "tables.json")]
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              cwd=REPO) for r in range(2)]
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    line = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["config"]["per_gpu_batch"] == 1 and j["value"] > 0
    # the line itself is flat scalars (the driver's record keeps those); the per-rank table travels in the tables file
    assert j["ms_per_step_without_collectives"] > 0 and "exposed_comm_ms_per_step" in j
    assert not [k for k, v in j.items() if isinstance(v, dict) and k not in ("config", "roofline", "cpu_baseline")], "nested tables belong in tables_path"
    mg = json.load(open(tmp_path / "tables.json"))["tables"]["multi_gpu"]
    assert len(mg["per_rank_median_ms_per_step"]) == 2 and all(v > 0 for v in mg["per_rank_median_ms_per_step"])
    assert mg["median_ms_per_step_without_collectives"] > 0 and "exposed_comm_ms_per_step" in mg
    assert not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")], "only rank 0 prints the result line"
