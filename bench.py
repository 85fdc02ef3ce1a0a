#!/usr/bin/env python3
"""Benchmark of the hot path: one "step" = one CycleGAN train_step + one MultiResUNet train_step over the same
global batch of synthetic SEM tiles (BASELINE.json metric: train_step tiles/sec, CycleGAN+UNet, 512x512 bs=8).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  Every number in it is measured by THIS run (or, for `roofline.traffic`, read from the committed
rocprofv3 PMC summary `profiles/pmc_traffic.json` that `tools/pmc_traffic.py` produces from this same command, and labelled so):

  value / ms_per_step  -- the K timed steps (barrier + synchronize on both sides, max over ranks); `median_ms_per_step` beside it.
                          The line holds flat scalars and short strings only; per-kernel tables and per-leg details go to the
                          file named by `tables_path` (default profiles/bench_tables_last.json)
  cyclegan_ms / unet_ms / unet_hbm_frac -- each trainer timed alone in further steps of the same process
  per_gpu_share_ms / scaling_cap        -- the same models on ONE tile (what each GPU of an 8-GPU run computes per step);
                          scaling_cap = median_ms_per_step / per_gpu_share_ms bounds the data-parallel speed-up
  roofline             -- dominant CONTRACTION kernel class of the step: EXECUTED matrix-instruction FLOPs (every piece product of
                          the operand splits) / HIP-event time of those launches (ss_prof_*: events on the launch stream, steps run
                          on one stream) / dense peak of the instruction the kernel issues; the legs' scalars are repeated inside
                          it; every instrumented class is in the tables file (`roofline_kernels`, `hbm_bound_kernels`)
  tiles_per_s_x6_exact / tiles_per_s_fp32_mfma -- the same step under the two stricter arithmetic modes (x3h = 0: exact 3-piece
                          bf16 split, 6 products; x6 = 0: fp32 MFMA instructions only)
  cpu_baseline         -- the oracle (plain-torch CPU restatement of the same two steps) on the host cores, warm, median of the
                          timed steps, on a bounded sample (rank 0, N=1 only).  Reported baseline only.
"""
import argparse
import importlib
import json
import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")          # before the HIP runtime initialises (see _lib.py)
import statistics
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
PKG = "automatic-sem-image-segmentation_amd"

# MI355X_MICROARCH.md: dense peaks at 256 CUs x 2.4 GHz
PEAK_TFLOPS = {"f32_mfma": 157.3, "f16_mfma": 2516.6, "bf16_mfma": 2516.6}
PEAK_HBM_TBPS = 8.0
G_FWD_GF = {256: 102.29, 384: 230.15, 512: 409.16, 1024: 1636.65}
D_FWD_GF = {256: 7.88, 384: 18.32, 512: 33.09, 1024: 135.56}
U_FWD_GF = {256: 10.56, 384: 23.76, 512: 42.24, 1024: 168.94}
UNET_BYTES_PER_TILE_512 = 6.5e9                                      # fp32 activations, 512x512


def synthetic_tiles(n, size, seed):
    """SEM-like real_a (mean ~0.13, smooth noise) and mask-like real_b (~10 % discs), NHWC in [-1,1]."""
    g = torch.Generator().manual_seed(seed)
    noise = torch.randn((n, 1, size + 8, size + 8), generator=g)
    smooth = torch.nn.functional.avg_pool2d(noise, 9, stride=1) * 3.0      # box blur 9x9, renormalised to ~N(0,1)
    real_a = (0.13 + 0.18 * smooth).clamp(0, 1) * 2 - 1
    yy, xx = torch.meshgrid(torch.arange(size), torch.arange(size), indexing="ij")
    real_b = torch.full((n, 1, size, size), -1.0)
    n_discs = max(int(0.10 * size * size / (3.14159 * 8 * 8)), 1)
    for i in range(n):
        cy = torch.randint(0, size, (n_discs,), generator=g)
        cx = torch.randint(0, size, (n_discs,), generator=g)
        r = torch.randint(4, 12, (n_discs,), generator=g)
        for k in range(n_discs):
            real_b[i, 0][(yy - cy[k]) ** 2 + (xx - cx[k]) ** 2 <= r[k] ** 2] = 1.0
    return real_a.permute(0, 2, 3, 1).contiguous(), real_b.permute(0, 2, 3, 1).contiguous()


def cpu_baseline(size, batch, filters, threads, timed):
    """Oracle CycleGAN step + UNet step on the host: one warm-up step on a quarter-size tile (thread pools, allocator, oneDNN
    primitive caches), then `timed` steps at the sample size; the median is reported."""
    from oracle import nets as ON
    from oracle import steps as OS
    torch.set_num_threads(threads)
    cg = OS.CycleGanStep(ON.ResnetGenerator(filters, seed=1), ON.ResnetGenerator(filters, seed=2),
                         ON.PatchDiscriminator(2 * filters, seed=3), ON.PatchDiscriminator(2 * filters, seed=4),
                         OS.ImagePool(2, 50), OS.ImagePool(2, 50))
    un = OS.UNetStep(ON.MultiResUNet(16, seed=5), 9.0)
    wa, wb = synthetic_tiles(batch, max(size // 2, 64), 99)
    cg.train_step((wa, wb))
    un.train_step(((wa + 1) / 2, (wb + 1) / 2))
    a, b = synthetic_tiles(batch, size, 4321)
    t_cg, t_un = [], []
    for _ in range(timed):
        t0 = time.perf_counter()
        cg.train_step((a, b))
        t1 = time.perf_counter()
        un.train_step(((a + 1) / 2, (b + 1) / 2))
        t2 = time.perf_counter()
        t_cg.append(t1 - t0)
        t_un.append(t2 - t1)
    m_cg, m_un = statistics.median(t_cg), statistics.median(t_un)
    return {"value": round(batch / (m_cg + m_un), 5), "unit": "tiles/s", "cores": threads, "host_threads": os.cpu_count(),
            "kind": "port", "cyclegan_s": round(m_cg, 2), "unet_s": round(m_un, 2),
            "role": "reported baseline only: plain-torch CPU fp32 restatement (oracle/) of the two train steps",
            "sample": f"{batch} tile {size}x{size}, F={filters}: 1 warm-up at {max(size // 2, 64)}px + {timed} timed steps, median"}


def kernel_peak(name):
    """Dense peak of the matrix instruction a contraction kernel class issues."""
    if name.startswith("gemm_x6p_kernel<2") or name.endswith(",true>") or name.startswith("gconv_phases_fused_kernel"):
        return "f16_mfma"        # x3h: v_mfma_f32_32x32x16_f16
    if name.startswith("gconv_x6v2"):
        return "f16_mfma"        # gather convolutions, second structure: x3h only
    if name.startswith(("gemm_x6p", "gconv_x6", "wgrad_x6")):
        return "bf16_mfma"       # x6: v_mfma_f32_32x32x16_bf16
    if name.startswith("gemm_tn_x3h"):
        return "f16_mfma"        # Winograd weight gradient on pre-split planes: v_mfma_f32_32x32x16_f16, three piece products
    if name.startswith("twgrad_x3h_kernel"):
        return "f16_mfma"        # tile weight gradient on the fp16 matrix cores (three piece products)
    if name.startswith("tconv_kernel"):
        return "f16_mfma"        # tile kernels: v_mfma_f32_32x32x16_f16 (x3h piece products for fp32 storage, one product for 16-bit storage)
    return "f32_mfma"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=4, choices=[2, 3, 4, 5],
                    help="BASELINE.json configs: 4 (default, the headline: CycleGAN+UNet 512x512 global batch 8, fp32), 2 (UNet only, "
                         "256x256 batch 16, bf16 storage), 3 (CycleGAN only, 256x256 batch 4), 5 (CycleGAN+UNet 1024x1024 batch 8, fp16 "
                         "storage, checkpointed residual trunk)")
    ap.add_argument("--size", type=int, default=None)
    ap.add_argument("--global-batch", type=int, default=None)
    ap.add_argument("--dtype", default=None, choices=["f32", "bf16", "f16"], help="activation storage type (weights / statistics / accumulation: fp32)")
    ap.add_argument("--checkpoint", action="store_true", help="recompute the generators' residual blocks in backward")
    ap.add_argument("--filters", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the cyclegan/unet split, the arithmetic-mode runs and the profile leg")
    ap.add_argument("--cpu-sample-size", type=int, default=512)
    ap.add_argument("--cpu-sample-batch", type=int, default=1)
    ap.add_argument("--cpu-timed-steps", type=int, default=2)
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--no-rccl-smoke", action="store_true", help="do not run the 2-rank RCCL smoke when >= 2 GPUs are visible to a 1-GPU run")
    ap.add_argument("--no-share-leg", action="store_true", help="skip the per_gpu_share leg (the same models on one tile)")
    ap.add_argument("--tables", default=None, help="where the per-kernel tables go (default: profiles/bench_tables_last.json)")
    ap.add_argument("--skip-unet", action="store_true", help="diagnostics only: time the CycleGAN step alone")
    ap.add_argument("--only-unet", action="store_true", help="diagnostics only: time the UNet step alone")
    args = ap.parse_args()
    preset = {4: dict(size=512, gb=8, dtype="f32", ck=False, cg=True, un=True), 2: dict(size=256, gb=16, dtype="bf16", ck=False, cg=False, un=True),
              3: dict(size=256, gb=4, dtype="f32", ck=False, cg=True, un=False), 5: dict(size=1024, gb=8, dtype="f16", ck=True, cg=True, un=True)}[args.config]
    args.size = args.size or preset["size"]
    args.global_batch = args.global_batch or preset["gb"]
    args.dtype = args.dtype or preset["dtype"]
    args.checkpoint = args.checkpoint or preset["ck"]
    if not preset["cg"]:
        args.only_unet = True
    if not preset["un"]:
        args.skip_unet = True

    D = importlib.import_module(PKG + ".dist")
    D.init_from_env()
    rank, world = D.rank(), D.world_size()
    assert world == args.gpus or world == 1 and args.gpus == 1, (world, args.gpus)
    dev = D.local_device()
    torch.cuda.set_device(dev)
    E = importlib.import_module(PKG + ".engine")
    L = importlib.import_module(PKG + "._lib")
    CG = importlib.import_module(PKG + ".CycleGAN")
    UN = importlib.import_module(PKG + ".UNet_Segmentation")
    NETS = importlib.import_module(PKG + ".nets")
    OPT = importlib.import_module(PKG + ".optim")
    if world > 1:
        # the exchange really runs over RCCL with one rank per GPU
        assert torch.distributed.get_world_size() == args.gpus
        if os.environ.get("SS_BENCH_TEST_TRANSPORT") != "gloo":          # tests/test_dp_gpu.py runs 2 ranks on the 1-GPU box over gloo
            assert torch.distributed.get_backend() == "nccl", "the benchmark exchanges over RCCL"
            assert torch.cuda.device_count() >= args.gpus, "one GPU per rank"

    S, GB, F = args.size, args.global_batch, args.filters
    D.check_batch_divisible(GB, world, "--global-batch")
    per = GB // world

    # networks exactly as CycleGAN.create_model / UNet.create_model build them for StartProcess.py's options
    ga = NETS.ResnetGenerator(filters=F, device=dev, seed=1, act_dtype=args.dtype, checkpoint_blocks=args.checkpoint)
    gb = NETS.ResnetGenerator(filters=F, device=dev, seed=2, act_dtype=args.dtype, checkpoint_blocks=args.checkpoint)
    da = NETS.PatchDiscriminator(filters=2 * F, device=dev, seed=3, act_dtype=args.dtype)
    db = NETS.PatchDiscriminator(filters=2 * F, device=dev, seed=4, act_dtype=args.dtype)
    unet = NETS.MultiResUNet(16, device=dev, seed=5, act_dtype=args.dtype)
    D.broadcast_params([ga, gb, da, db, unet])
    D.enable_overlap([ga, gb, da, db, unet])     # bucketed gradient all-reduce launched during backward
    if world > 1:
        D.enable_sync_bn(True)      # whole-(global)-batch BatchNorm statistics = the single-device semantics of the reference
    model = CG.CycleGanModel(ga, gb, da, db, image_pool_a=CG.ImagePool(2, 50), image_pool_b=CG.ImagePool(2, 50))
    model.compile(OPT.Adam(2e-4, beta_1=0.5), OPT.Adam(2e-4, beta_1=0.5), OPT.Adam(2e-4, beta_1=0.5), OPT.Adam(2e-4, beta_1=0.5))
    umodel = UN.UNetModel(unet, 9.0, OPT.Adam(1e-3))

    # synthetic tiles, resident in HBM before the timed region
    a_all, b_all = synthetic_tiles(GB, S, 1234)
    a = E.Act(a_all[rank * per:(rank + 1) * per].to(dev).contiguous(), requires_grad=False)
    b = E.Act(b_all[rank * per:(rank + 1) * per].to(dev).contiguous(), requires_grad=False)
    ux = E.Act(((a.t + 1) / 2).contiguous(), requires_grad=False)
    uy = E.Act(((b.t + 1) / 2).contiguous(), requires_grad=False)

    # SS_OVERLAP_UNET=1: the timed steps issue the UNet step on a stream of its own BESIDE the CycleGAN step (two independent models on
    # independent tiles; 4-5 % more tiles/s).  Not the default: the headline is the sum of the two steps, as BASELINE.md section 3
    # defines the combined figure; the overlapped rate is reported as the extra leg `overlapped`.
    overlap_unet = os.environ.get("SS_OVERLAP_UNET", "0") == "1"
    u_stream = E.side_streams(dev, 12)[10] if dev.type == "cuda" else None
    # (sequential steps: the UNet's side streams are the first of engine.side_streams, shared with the CycleGAN chains that are idle
    # then -- measured 34.1 - 35.1 ms per UNet step against 36.6 - 36.9 on streams of its own created after them; beside the CycleGAN
    # step it takes streams apart from the chains')

    full_inputs = (a, b, ux, uy)

    def step(cyclegan=not args.only_unet, unet_=not args.skip_unet, overlap=overlap_unet, inputs=None):
        a, b, ux, uy = inputs if inputs is not None else full_inputs
        if overlap and cyclegan and unet_:
            # CycleGAN first (GPU-paced: its kernels outlast its issue), the UNet step -- host-paced at per-GPU batch 1: 12 ms to issue
            # 11 ms of kernels -- is issued BEHIND it on a stream of its own, so the host issues it while the GPU still works on the
            # CycleGAN chains; neither step reads its metrics before both are issued (the read blocks the host)
            cur = torch.cuda.current_stream()
            u_stream.wait_stream(cur)
            model.sync_metrics = False
            model.train_step((a, b))
            model.sync_metrics = True
            umodel.sync_metrics, keep = False, umodel.stream_indices
            umodel.stream_indices = keep if keep is not None else [11, 12, 13, 12, 13]
            with torch.cuda.stream(u_stream):
                umodel.train_step((ux.t, uy.t))
            umodel.sync_metrics, umodel.stream_indices = True, keep
            cur.wait_stream(u_stream)
            model._scalars.cpu()          # the step's one device->host read (both steps' scalars are final behind the join)
            return
        if cyclegan:
            model.train_step((a, b))
        if unet_:
            umodel.train_step((ux.t, uy.t))

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def timed(n, **kw):
        """n steps between barriers: (total seconds, per-step seconds).  Every train step ends with a device->host read of its
        metrics, so the per-step host timestamps are device-synchronous."""
        barrier()
        t0 = time.perf_counter()
        marks = [t0]
        for _ in range(n):
            step(**kw)
            marks.append(time.perf_counter())
        barrier()
        total = time.perf_counter() - t0
        return total, [y - x for x, y in zip(marks[:-1], marks[1:])]

    def max_over_ranks(v):
        if world == 1:
            return v
        t = torch.tensor([v], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return float(t.item())

    for _ in range(args.warmup):
        step()
    elapsed, per_step = timed(args.steps)
    elapsed = max_over_ranks(elapsed)
    median_ms = max_over_ranks(statistics.median(per_step)) * 1e3

    extras = {}
    if world > 1:
        # per-rank view + what the exchange costs: the same steps once more with every data-path collective skipped
        # (dist.SKIP_COLLECTIVES, measurement only -- the ranks' weights drift apart, nothing of it enters `value`)
        t = torch.zeros(world, device=dev, dtype=torch.float64)
        t[rank] = statistics.median(per_step) * 1e3
        torch.distributed.all_reduce(t)
        k = max(3, min(args.steps, 5))
        D.SKIP_COLLECTIVES = True
        step()
        _, nc_t = timed(k)
        D.SKIP_COLLECTIVES = False
        nc_ms = max_over_ranks(statistics.median(nc_t)) * 1e3
        D.broadcast_params([ga, gb, da, db, unet])          # re-align the replicas after the measurement
        step()
        extras["multi_gpu"] = {"per_rank_median_ms_per_step": [round(float(v), 3) for v in t.tolist()],
                               "median_ms_per_step_without_collectives": round(nc_ms, 3),
                               "exposed_comm_ms_per_step": round(median_ms - nc_ms, 3),
                               "collectives": "gradient all-reduce (32 MiB buckets, launched during backward) + SyncBN statistics "
                                              "(2 all-reduces of 2C floats per BatchNorm layer) + one metrics all-reduce per step, RCCL"}
    if not args.no_extras and not args.only_unet and not args.skip_unet:
        k = max(3, min(args.steps, 5))
        _, cg_t = timed(k, cyclegan=True, unet_=False)
        _, un_t = timed(k, cyclegan=False, unet_=True)
        cg_ms, un_ms = max_over_ranks(statistics.median(cg_t)) * 1e3, max_over_ranks(statistics.median(un_t)) * 1e3
        cg_tps, un_tps = GB / cg_ms * 1e3, GB / un_ms * 1e3
        ub = UNET_BYTES_PER_TILE_512 * (S / 512.0) ** 2
        extras["cyclegan"] = {"tiles_per_s": round(cg_tps, 3), "median_ms_per_step": round(cg_ms, 3), "steps": k}
        extras["unet"] = {"tiles_per_s": round(un_tps, 3), "median_ms_per_step": round(un_ms, 3), "steps": k, "bound": "hbm",
                          "algorithmic_bytes_per_tile": ub, "achieved_TBps_per_gpu": round(un_tps * ub / 1e12 / world, 4),
                          "hbm_frac": round(un_tps * ub / 1e12 / world / PEAK_HBM_TBPS, 4)}
        extras["combined"] = {"tiles_per_s": round(1.0 / (1.0 / cg_tps + 1.0 / un_tps), 3), "formula": "1/(1/cyclegan + 1/unet)"}
        if world == 1 and per > 1 and not args.no_share_leg:
            # per_gpu_share: the SAME models on ONE tile = what each GPU of an 8-GPU run of this global batch computes per step (no
            # communication).  ms_per_step / per_gpu_share_ms bounds the speed-up data parallelism can reach: the only proxy for the
            # multi-GPU curve a 1-GPU box can time.
            one = tuple(E.Act(t_.t[:1].contiguous(), requires_grad=False) for t_ in full_inputs)
            for _ in range(5):          # (new geometry: descriptors, weight-cache users, the recorded refresh plan and allocator pools settle)
                step(inputs=one)
            _, sh_t = timed(10, inputs=one)
            _, sh_cg = timed(3, cyclegan=True, unet_=False, inputs=one)
            _, sh_un = timed(3, cyclegan=False, unet_=True, inputs=one)
            extras["per_gpu_share"] = {"per_gpu_batch": 1, "median_ms_per_step": round(statistics.median(sh_t) * 1e3, 3), "steps": 10, "warmup": 5,
                                       "cyclegan_ms": round(statistics.median(sh_cg) * 1e3, 3), "unet_ms": round(statistics.median(sh_un) * 1e3, 3)}
            for _ in range(2):
                step()          # back on the full batch (allocator pools, caches)
        if world == 1:
            step(overlap=True)
            tot, ov_t = timed(k, overlap=True)
            extras["overlapped"] = {"tiles_per_s": round(GB * k / max_over_ranks(tot), 3), "median_ms_per_step": round(statistics.median(ov_t) * 1e3, 3),
                                    "steps": k, "what": "the UNet step issued on a stream of its own beside the CycleGAN step (same work, "
                                                        "same results; not the headline `value`, which runs the two steps one after the other)"}
        # the same step under the stricter arithmetic modes (explicit ss_config_set switches, same process, same buffers)
        modes = {}
        for name, cfg in () if os.environ.get("SS_BENCH_LIGHT") == "1" else (("x6_exact_bf16_split_6_products", dict(x3h=0)), ("fp32_mfma_instructions_only", dict(x6=0))):
            with L.config(**cfg):
                step()
                tot, _ = timed(3)
            modes[name] = {"config": cfg, "tiles_per_s": round(GB * 3 / max_over_ranks(tot), 3)}
        step()      # back on the default mode (re-warm caches keyed on the configuration)
        extras["arithmetic_modes"] = modes

    # roofline leg: the timed steps run two kernel chains CONCURRENTLY on two HIP streams, where an event interval also contains
    # the neighbour's kernels; the contraction kernels are therefore timed (HIP events on the launch stream, inside the library:
    # ss_prof_*) in two further steps of the same workload run on ONE stream (same process, shapes, buffers; not part of `value`)
    prof = {}
    if not args.no_extras and os.environ.get("SS_BENCH_LIGHT") != "1":
        dual, usides, ubr = model.dual_stream, umodel.wgrad_side_stream, umodel.branch_streams
        model.dual_stream = False
        umodel.wgrad_side_stream = umodel.branch_streams = False          # event intervals must hold one kernel each
        step()
        torch.cuda.synchronize()
        lib = L.load()
        lib.ss_prof_reset()
        lib.ss_prof_enable(1)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        lib.ss_prof_enable(0)
        prof = L.prof_summary()
        model.dual_stream, umodel.wgrad_side_stream, umodel.branch_streams = dual, usides, ubr
    barrier()

    # RCCL smoke: the 2-rank tests of this repository share ONE GPU over gloo (RCCL refuses two ranks on a device), so the first box
    # that shows >= 2 devices to a single-GPU run also runs two steps of this benchmark on 2 ranks over RCCL (sub-process, bounded)
    rccl_smoke = None
    if world == 1 and rank == 0 and not args.no_extras and not args.no_rccl_smoke and torch.cuda.device_count() >= 2:
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(29500 + os.getpid() % 2000), os.path.abspath(__file__), "--gpus", "2", "--steps", "2", "--warmup", "1",
               "--no-extras", "--no-cpu-baseline", "--config", str(args.config)]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            rccl_smoke = {"rc": r.returncode, "result": json.loads(line[-1]) if line else None, "stderr_tail": r.stderr[-400:] if r.returncode else ""}
        except Exception as e:          # noqa: BLE001
            rccl_smoke = {"rc": -1, "error": repr(e)}

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = GB * args.steps / elapsed
        x6, x3h = L.config_get("x6"), L.config_get("x3h")
        store = {"f32": "f32", "bf16": "bf16", "f16": "f16"}[args.dtype]
        what = "CycleGAN+UNet" if not (args.only_unet or args.skip_unet) else ("UNet" if args.only_unet else "CycleGAN")
        # ONE compact line: flat scalars and short strings only (the driver's record keeps scalars; everything tabular -- per-kernel
        # rooflines, the arithmetic string, per-leg details -- goes to `tables_path`)
        out = {"metric": f"train_step tiles/sec ({what})", "value": round(value, 4), "unit": "tiles/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
               "median_ms_per_step": round(median_ms, 3),
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": (store + " storage, " if store != "f32" else "") + ("f32" if not x6 else ("f32 via 2xf16 split (x3h)" if x3h else "f32 via 3xbf16 split (x6)")),
               "data": "synthetic",
               "config": {"workload": f"CycleGAN+MultiResUNet train_step, {S}x{S} tiles, global batch {GB}, F={F}"
                                      + (" [CycleGAN only]" if args.skip_unet else "") + (" [UNet only]" if args.only_unet else ""),
                          "tile": S, "global_batch": GB, "per_gpu_batch": per, "parallelism": f"dp{world}", "baseline_config": args.config,
                          "activation_storage": store, "checkpointed_trunk": bool(args.checkpoint)}}
        tables = {"arithmetic": (store + " activation storage, f32 accumulate; contractions on the 16-bit matrix cores with fp32-grade operand splits: "
                                 + ("x3h = 2 fp16 pieces under power-of-two scales, 3 products" if x3h else "x6 = exact 3-piece bf16 split, 6 products"))
                                if x6 else "f32 everywhere (v_mfma_f32_32x32x2_f32)"}
        tables.update(extras)
        flat = {}
        if "cyclegan" in extras:
            flat.update(cyclegan_ms=extras["cyclegan"]["median_ms_per_step"], unet_ms=extras["unet"]["median_ms_per_step"],
                        unet_hbm_frac=extras["unet"]["hbm_frac"], combined_tiles_per_s=extras["combined"]["tiles_per_s"])
        if "overlapped" in extras:
            flat["overlapped_tiles_per_s"] = extras["overlapped"]["tiles_per_s"]
        am = extras.get("arithmetic_modes") or {}
        if "x6_exact_bf16_split_6_products" in am:
            flat["tiles_per_s_x6_exact"] = am["x6_exact_bf16_split_6_products"]["tiles_per_s"]
        if "fp32_mfma_instructions_only" in am:
            flat["tiles_per_s_fp32_mfma"] = am["fp32_mfma_instructions_only"]["tiles_per_s"]
        if "per_gpu_share" in extras:
            sh = extras["per_gpu_share"]
            flat.update(per_gpu_share_ms=sh["median_ms_per_step"], per_gpu_share_cyclegan_ms=sh["cyclegan_ms"], per_gpu_share_unet_ms=sh["unet_ms"],
                        scaling_cap=round(median_ms / sh["median_ms_per_step"], 3))
        if "multi_gpu" in extras:
            flat.update(exposed_comm_ms_per_step=extras["multi_gpu"]["exposed_comm_ms_per_step"],
                        ms_per_step_without_collectives=extras["multi_gpu"]["median_ms_per_step_without_collectives"])
        if rccl_smoke is not None:
            tables["rccl_smoke_2_ranks"] = rccl_smoke
            flat["rccl_smoke_rc"] = rccl_smoke.get("rc")
        roof = None
        if prof:
            table, hbm = {}, {}
            for name, e in prof.items():
                sec = e["total_ms"] * 1e-3
                tbps = e["bytes"] / sec / 1e12 if sec > 0 else 0.0
                hrow = {"launches": e["launches"], "avg_ms": round(e["avg_ms"], 4), "total_ms_per_step": round(e["total_ms"] / 2, 3),
                        "algorithmic_MB_per_launch": round(e["bytes"] / e["launches"] / 1e6, 1), "achieved_TBps": round(tbps, 3),
                        "peak_TBps": PEAK_HBM_TBPS, "frac": round(tbps / PEAK_HBM_TBPS, 4)}
                if e["flops"] <= 0:          # streaming kernels (normalisation passes, Winograd transforms): priced against HBM only
                    hbm[name] = hrow
                    continue
                pk = kernel_peak(name)
                ach = e["flops"] / sec / 1e12 if sec > 0 else 0.0
                table[name] = {"launches": e["launches"], "avg_ms": round(e["avg_ms"], 4), "total_ms_per_step": round(e["total_ms"] / 2, 3),
                               "executed_tflop_per_launch": round(e["flops"] / e["launches"] / 1e12, 5),
                               "algorithmic_MB_per_launch": round(e["bytes"] / e["launches"] / 1e6, 1), "achieved": round(ach, 1),
                               "peak": PEAK_TFLOPS[pk], "instruction": pk, "frac": round(ach / PEAK_TFLOPS[pk], 4)}
                if name.startswith(("tconv_kernel", "twgrad")):          # the MultiResUNet's small-channel layers: bound by HBM, not the matrix pipe
                    hbm[name] = hrow
            tj = None
            try:
                tj = json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json")))
                if tj.get("workload") != [S, GB, F, world]:
                    tj = None
            except Exception:
                tj = None
            if table:
                dom = max(table, key=lambda n_: table[n_]["total_ms_per_step"])
                d = table[dom]
                traffic = None
                if tj is not None:
                    kern = tj.get("kernels", {})
                    # rocprofv3 spells the default template arguments out (gemm_x6p_kernel<2,false>), the ss_prof label does not
                    stem = dom[:-1] if dom.endswith(">") else dom
                    traffic = kern.get(dom) or next((v for k_, v in sorted(kern.items()) if k_.startswith(stem + ",")), None)
                # what the matrix pipe of THIS box delivers on real data (the chip clocks to its power budget): a register-only MFMA stream
                # on random fp16 operands, measured now (ss_probe_mfma) -- `frac` stays against the nominal peak of MI355X_MICROARCH.md
                ceiling = None
                try:
                    import ctypes
                    sc = torch.empty(65600, dtype=torch.uint8, device=dev)
                    tf, mhz = ctypes.c_double(0.0), ctypes.c_double(0.0)
                    tf0, mhz0 = ctypes.c_double(0.0), ctypes.c_double(0.0)
                    lib_ = L.load()
                    if lib_.ss_probe_mfma(1, sc.data_ptr(), sc.numel(), None, ctypes.byref(tf), ctypes.byref(mhz)) == 0 and \
                            lib_.ss_probe_mfma(0, sc.data_ptr(), sc.numel(), None, ctypes.byref(tf0), ctypes.byref(mhz0)) == 0:
                        ceiling = {"tflops_random_operands": round(tf.value, 1), "effective_mhz_random_operands": round(mhz.value),
                                   "tflops_zero_operands": round(tf0.value, 1), "effective_mhz_zero_operands": round(mhz0.value),
                                   "frac_of_nominal_peak": round(tf.value / d["peak"], 4),
                                   "dominant_kernel_frac_of_this_ceiling": round(d["achieved"] / tf.value, 4) if tf.value > 0 else None,
                                   "source": "ss_probe_mfma in this run: register-only v_mfma_f32_32x32x16_f16 stream on every CU, no memory traffic"}
                except Exception as e:          # noqa: BLE001
                    ceiling = {"error": repr(e)}
                tables["real_data_ceiling"] = ceiling
                roof = {"bound": "mfma", "kernel": dom, "achieved": d["achieved"], "peak": d["peak"], "unit": "TFLOP/s",
                        "frac": d["frac"], "avg_launch_ms": d["avg_ms"], "launches_timed": d["launches"],
                        "executed_gflop_per_launch": round(d["executed_tflop_per_launch"] * 1e3, 2),
                        "algorithmic_bytes_per_launch": round(d["algorithmic_MB_per_launch"] * 1e6),
                        "traffic": round(traffic["bytes_per_launch"]) if traffic else None,
                        "traffic_bytes_per_launch": round(traffic["bytes_per_launch"]) if traffic else None,
                        "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, own passes)" if traffic else None,
                        "real_ceiling_tflops": (ceiling or {}).get("tflops_random_operands"),
                        "definition": "executed MFMA FLOPs (all piece products) / HIP-event time on the launch stream / dense fp16 peak"}
                tables["roofline_kernels"] = table
            if hbm:
                # HBM traffic of the same kernel classes from the committed PMC passes (profiles/pmc_traffic.json: every template
                # instantiation of a class, weighted by its launches), next to the algorithmic bytes: the ratio is what is re-read
                if tj is not None:
                    for name, row in hbm.items():
                        stem = name.split("<")[0].split(" ")[0]
                        if name.startswith(("tconv_kernel", "twgrad")):
                            continue          # (their ss_prof labels carry tile shapes the counters' kernel names do not)
                        sel = [v for k_, v in tj.get("kernels", {}).items() if k_.split("<")[0] == stem
                               and ("<normalising>" in name) == (",true," in k_ and stem == "wino_input_kernel")
                               and (("<fwd>" not in name) or "," + "0," in k_) and (("<bwd>" not in name) or "," + "1," in k_)]
                        n_l = sum(v["launches"] for v in sel)
                        if n_l:
                            mb = sum(v["bytes_per_launch"] * v["launches"] for v in sel) / n_l / 1e6
                            row["pmc_traffic_MB_per_launch"] = round(mb, 1)
                            row["traffic_over_algorithmic"] = round(mb / row["algorithmic_MB_per_launch"], 3) if row["algorithmic_MB_per_launch"] else None
                tot_ms = sum(v["total_ms_per_step"] for v in hbm.values())
                tot_b = sum(v["algorithmic_MB_per_launch"] * v["launches"] / 2 for v in hbm.values()) * 1e6
                tables["hbm_bound_kernels"] = {
                    "definition": "ALGORITHMIC bytes of the timed launches (each tensor the pass has to read or write, once) / their HIP-event "
                                  "time (same two single-stream steps) / 8 TB/s (MI355X_MICROARCH.md; ~6.3 TB/s is what a streaming kernel reaches)",
                    "total_ms_per_step": round(tot_ms, 3), "achieved_TBps": round(tot_b / (tot_ms * 1e-3) / 1e12, 3) if tot_ms > 0 else None,
                    "frac": round(tot_b / (tot_ms * 1e-3) / 1e12 / PEAK_HBM_TBPS, 4) if tot_ms > 0 else None, "kernels": hbm}
                if tot_ms > 0:
                    flat.update(streaming_kernels_ms_per_step=round(tot_ms, 3), streaming_kernels_hbm_frac=tables["hbm_bound_kernels"]["frac"])
        if roof is None and args.only_unet:
            # UNet-only workloads (BASELINE config 2): the step is HBM-bound by design -- SURVEY 8(d)'s algorithmic bytes per tile over the step time
            ub = UNET_BYTES_PER_TILE_512 * (S / 512.0) ** 2 * ({"f32": 1.0}.get(store, 0.5))
            tb = value * ub / 1e12 / world
            roof = {"bound": "hbm", "kernel": "MultiResUNet train step (whole step)", "achieved": round(tb * 1e3, 1), "peak": PEAK_HBM_TBPS * 1e3,
                    "unit": "GB/s", "frac": round(tb / PEAK_HBM_TBPS, 4), "algorithmic_bytes_per_launch": round(ub * GB / world), "traffic": None,
                    "definition": "SURVEY 8(d) activation bytes per tile (x0.5 for 16-bit storage) x tiles per step / step time / 8 TB/s"}
        if roof is not None:
            roof.update(flat)          # the driver's record keeps the scalars of `roofline`: the legs' figures travel there as well
            out["roofline"] = roof
        out.update(flat)
        if S in G_FWD_GF:
            alg = (0 if args.only_unet else 18 * G_FWD_GF[S] + 16 * D_FWD_GF[S]) + (0 if args.skip_unet else 3 * U_FWD_GF[S])
            out["algorithmic_tflops_per_gpu"] = round(alg * 1e9 * value / 1e12 / world, 2)   # SURVEY 8d direct-conv FLOPs, whole step
        if world == 1 and not args.no_cpu_baseline:
            # torch CPU convs on the 256-thread host get SLOWER beyond ~16 threads (measured: 256^2 tile 2.2 s @16, 3.0 s @32, 6.6 s @64)
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample_size, args.cpu_sample_batch, F, min(os.cpu_count() or 1, args.cpu_threads),
                                               args.cpu_timed_steps)
        tpath = args.tables or os.path.join(REPO, "profiles", "bench_tables_last.json")
        try:
            with open(tpath, "w") as f:
                json.dump({"line": out, "tables": tables}, f, indent=1)
            out["tables_path"] = os.path.relpath(tpath, REPO)
        except OSError:
            out["tables_path"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
