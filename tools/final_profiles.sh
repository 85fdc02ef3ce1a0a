#!/bin/bash
# Regenerates the round's judged artifacts on the GPU box: kernel-trace summaries (batch 8, per-GPU batch 1, UNet-only), the per-shape
# table of the contraction kernels, the PMC traffic passes and the bench line.  Usage: tools/final_profiles.sh <tag>   (tag e.g. r02_b)
tag=$1
repo=$(cd "$(dirname "$0")/.." && pwd)
cd $repo
bash tools/profile_run2.sh ${tag}_b8 "gemm_ wgrad_ gconv_ tconv twgrad conv_"
bash tools/profile_run.sh ${tag}_b1 --global-batch 1
bash tools/profile_run.sh ${tag}_unet --only-unet
(cd /tmp && export TMPDIR=/tmp && python $repo/tools/pmc_traffic.py --out $repo/gpurun_out/${tag}_pmc_traffic.json > $repo/gpurun_out/${tag}_pmc.log 2>&1)
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_n1.json 2> gpurun_out/${tag}_bench.err
python bench.py --global-batch 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/${tag}_bench_b1.json 2>> gpurun_out/${tag}_bench.err
python bench.py --config 2 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/${tag}_bench_cfg2.json 2>> gpurun_out/${tag}_bench.err
python bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/${tag}_bench_cfg3.json 2>> gpurun_out/${tag}_bench.err
python bench.py --config 5 --steps 4 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/${tag}_bench_cfg5.json 2>> gpurun_out/${tag}_bench.err
bash tools/profile_run.sh ${tag}_cfg5 --config 5 --global-batch 2
bash tools/profile_run.sh ${tag}_cfg2 --config 2
