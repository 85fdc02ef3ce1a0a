"""Generate tests/golden/cyclegan_step_{512,256}_f64.npz and cyclegan_step_1024_f32.npz: ONE complete CycleGAN train step with the
full-size networks (F = 64, 9 residual blocks, PatchGAN 128) on one tile of the headline size (512x512), of BASELINE config 3's size
(256x256) and of config 5's size (1024x1024) -- through the float64 oracle (the arbiter) and through the float32 oracle
(the noise model: how far plain fp32 arithmetic of the same step lies from float64).  The GPU suite used to compute the 256 and 1024
oracle steps on the GPU box's host cores on every run (70 s of a 170 s suite, minutes on a loaded host); now they are data.

    python tests/golden/make_fullsize_golden.py                       512x512, fp32 + fp64   (~10 min of CPU in the build container)
    python tests/golden/make_fullsize_golden.py --size 256            256x256, fp32 + fp64   (~3 min)
    python tests/golden/make_fullsize_golden.py --size 1024 --fp32    1024x1024, fp32 only   (~4 min; the 16-bit configs compare with fp32)

(needs nothing of /root/reference)

Stored (data only):
  * the 14 metrics in float64 and as the fp32 oracle computed them;
  * per network: every gradient tensor's L2 norm (float64), the fp32 oracle's per-tensor and whole-network distance to float64,
    and 100 000 sampled gradient entries of the concatenated gradient vector (float64 values; the sample positions are
    `np.random.default_rng(SAMPLE_SEED + i).choice(total, 100000, replace=False)`, re-drawn by the test) with the fp32 oracle's
    values at the same positions;
  * a CRC of the initial weights and inputs, so that the test notices if its own seeded construction ever drifts from this run's.
The GPU test (tests/test_fullsize_gpu.py::test_cyclegan_step_512_vs_committed_fp64_fixture) rebuilds inputs and weights from the
same seeds, runs the HIP step in the three arithmetic modes and compares.
"""
import os
import random
import sys
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import nets as ON          # noqa: E402
from oracle import steps as OS         # noqa: E402

S, F, SAMPLES, SAMPLE_SEED, INPUT_SEED, RNG_SEED = 512, 64, 100000, 4242, 5, 11
NETS = ("gen_a", "gen_b", "disc_a", "disc_b")


def inputs(size=S):
    g = torch.Generator().manual_seed(INPUT_SEED)
    real_a = torch.rand((1, size, size, 1), generator=g) * 2 - 1
    real_b = (torch.rand((1, size, size, 1), generator=g) > 0.9).float() * 2 - 1
    return real_a, real_b


def make_nets(dtype, filters=F):
    return dict(gen_a=ON.ResnetGenerator(filters=filters, seed=1, dtype=dtype), gen_b=ON.ResnetGenerator(filters=filters, seed=2, dtype=dtype),
                disc_a=ON.PatchDiscriminator(filters=2 * filters, seed=3, dtype=dtype),
                disc_b=ON.PatchDiscriminator(filters=2 * filters, seed=4, dtype=dtype))


def crc_of(arrays):
    c = 0
    for a in arrays:
        c = zlib.crc32(np.ascontiguousarray(a, dtype=np.float32).tobytes(), c)
    return c


def sample_positions(i, total, samples=None):
    return np.sort(np.random.default_rng(SAMPLE_SEED + i).choice(total, min(samples or SAMPLES, total), replace=False))


def main(size=S, filters=F, out_name="cyclegan_step_512_f64.npz", fp32_only=False):
    torch.set_num_threads(os.cpu_count() or 1)
    real_a, real_b = inputs(size)
    refs32 = make_nets(torch.float32, filters)
    init = {k: refs32[k].get_weights() for k in NETS}
    refs64 = refs32
    if not fp32_only:
        refs64 = make_nets(torch.float64, filters)
        for k in NETS:
            refs64[k].set_weights(init[k])
    out = {"size": size, "filters": filters, "samples": SAMPLES, "sample_seed": SAMPLE_SEED, "input_seed": INPUT_SEED, "rng_seed": RNG_SEED,
           "crc_inputs": crc_of([real_a.numpy(), real_b.numpy()])}
    for k in NETS:
        out[f"crc_init/{k}"] = crc_of(init[k])
    res = {}
    for tag, r, dt in (("32", refs32, torch.float32),) + (() if fp32_only else (("64", refs64, torch.float64),)):
        step = OS.CycleGanStep(r["gen_a"], r["gen_b"], r["disc_a"], r["disc_b"], OS.ImagePool(2, 50), OS.ImagePool(2, 50))
        random.seed(RNG_SEED)
        m = step.train_step((real_a.to(dt), real_b.to(dt)))
        res[tag] = (m, {k: [(v.name, v.value.grad.detach().double().numpy()) for v in r[k].trainable_weights] for k in NETS})
        print(tag, {kk: float(vv) for kk, vv in m.items()}, flush=True)
    out["fp32_only"] = int(fp32_only)          # then every "...64" entry below holds the fp32 oracle's value
    (m32, g32), (m64, g64) = res["32"], res["32" if fp32_only else "64"]
    names = sorted(m64)
    out["metric_names"] = np.array(names)
    out["metrics64"] = np.array([float(m64[k]) for k in names], np.float64)
    out["metrics32"] = np.array([float(m32[k]) for k in names], np.float64)
    for i, k in enumerate(NETS):
        tn = [n for n, _ in g64[k]]
        v64 = np.concatenate([g.ravel() for _, g in g64[k]])
        v32 = np.concatenate([g.ravel() for _, g in g32[k]])
        out[f"{k}/tensor_names"] = np.array(tn)
        out[f"{k}/tensor_sizes"] = np.array([g.size for _, g in g64[k]], np.int64)
        out[f"{k}/tensor_norm64"] = np.array([np.linalg.norm(g) for _, g in g64[k]], np.float64)
        out[f"{k}/tensor_absmax64"] = np.array([np.abs(g).max() for _, g in g64[k]], np.float64)
        out[f"{k}/tensor_err32"] = np.array([np.linalg.norm(a - b) for (_, a), (_, b) in zip(g32[k], g64[k])], np.float64)
        out[f"{k}/total_norm64"] = float(np.linalg.norm(v64))
        out[f"{k}/total_err32"] = float(np.linalg.norm(v32 - v64))
        pos = sample_positions(i, v64.size)
        out[f"{k}/sample64"] = v64[pos]
        out[f"{k}/sample32"] = v32[pos].astype(np.float32)
        print(k, "rel-L2 fp32 oracle vs fp64:", out[f"{k}/total_err32"] / out[f"{k}/total_norm64"], flush=True)
    path = os.path.join(HERE, out_name)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--small":      # quick self-check of the script (not committed)
        main(64, 8, "_scratch_fullsize_small.npz")
    elif "--size" in sys.argv:
        sz = int(sys.argv[sys.argv.index("--size") + 1])
        only32 = "--fp32" in sys.argv
        SAMPLES = 20000          # the headline fixture keeps 100 000 entries per network; these two are secondary shapes
        main(sz, F, f"cyclegan_step_{sz}_{'f32' if only32 else 'f64'}.npz", fp32_only=only32)
    else:
        main()
