#!/usr/bin/env python3
"""HBM traffic per launch of the contraction kernels of `bench.py`, from rocprofv3 PMC counters.

Runs the benchmark command (few steps, one stream) under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and, in a SEPARATE pass,
`--pmc WRITE_SIZE` (TCC slots: the two do not fit in one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"), sums the counters per
kernel class, applies the guide's gfx950 correction (FETCH_SIZE counts 64 B per 128-B request: x2) and writes
`profiles/pmc_traffic.json`, which bench.py attaches to `roofline.traffic` (labelled with this source).  Counters are in KB.

    cd /tmp && export TMPDIR=/tmp && python $REPO/tools/pmc_traffic.py [--size 512 --global-batch 8]

Never combines --pmc with sys/hip/hsa tracing (only --kernel-trace)."""
import argparse
import csv
import glob
import json
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    """`void (anonymous namespace)::gemm_x6p_kernel<2>(X6PParams)` -> `gemm_x6p_kernel<2>` (bench.py's kernel-class names)."""
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void\s+", "", name)
    m = re.match(r"([A-Za-z0-9_]+(<[^()]*>)?)", name)
    s = m.group(1) if m else name
    return s.replace(", ", ",")


def run_pass(counter, outdir, bench_args):
    env = dict(os.environ, SS_DUAL_STREAM="0", SS_UNET_BRANCHES="0", SS_UNET_WGRAD_STREAM="0", TMPDIR="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", outdir, "--",
           sys.executable, os.path.join(REPO, "bench.py"), "--steps", "1", "--warmup", "1", "--no-extras", "--no-cpu-baseline"] + bench_args
    subprocess.run(cmd, check=True, env=env, cwd="/tmp", stdout=subprocess.DEVNULL)
    per = {}
    for path in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                k = short(row["Kernel_Name"])
                n, tot = per.get(k, (0, 0.0))
                per[k] = (n + 1, tot + float(row["Counter_Value"]))
    return per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--global-batch", type=int, default=8)
    ap.add_argument("--filters", type=int, default=64)
    ap.add_argument("--out", default=os.path.join(REPO, "profiles", "pmc_traffic.json"))
    ap.add_argument("--scratch", default=os.path.join(REPO, "gpurun_out", "pmc"))
    args = ap.parse_args()
    bench_args = ["--size", str(args.size), "--global-batch", str(args.global_batch), "--filters", str(args.filters)]
    fetch = run_pass("FETCH_SIZE", os.path.join(args.scratch, "fetch"), bench_args)
    write = run_pass("WRITE_SIZE", os.path.join(args.scratch, "write"), bench_args)
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        nf, f = fetch.get(k, (0, 0.0))
        nw, w = write.get(k, (0, 0.0))
        n = max(nf, nw)
        if n == 0:
            continue
        # KB per launch; FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B); WRITE_SIZE uncalibrated, taken as reported
        kernels[k] = {"launches": n, "fetch_bytes_per_launch": 2.0 * f * 1024.0 / max(nf, 1), "write_bytes_per_launch": w * 1024.0 / max(nw, 1),
                      "bytes_per_launch": 2.0 * f * 1024.0 / max(nf, 1) + w * 1024.0 / max(nw, 1)}
    out = {"workload": [args.size, args.global_batch, args.filters, 1],
           "command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE | WRITE_SIZE} -- python bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline "
                      + " ".join(bench_args) + "  (SS_DUAL_STREAM=0; two separate passes)",
           "corrections": "FETCH_SIZE x 2 (gfx950, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported; counters in KB",
           "kernels": kernels}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    top = sorted(kernels.items(), key=lambda kv: -kv[1]["bytes_per_launch"] * kv[1]["launches"])[:12]
    for k, v in top:
        print(f"{k:60s} {v['launches']:5d} launches  {v['bytes_per_launch'] / 1e6:9.1f} MB/launch")


if __name__ == "__main__":
    main()
