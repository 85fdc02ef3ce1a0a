#!/bin/bash
# like profile_run.sh, plus the per-shape table of the named kernel classes; usage: tools/profile_run2.sh <tag> "<substr> <substr>" [bench args]
tag=$1; subs=$2; shift; shift
repo=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
SS_DUAL_STREAM=0 SS_UNET_BRANCHES=0 SS_UNET_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -d $repo/gpurun_out/prof_$tag -- python $repo/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras "$@" > $repo/gpurun_out/prof_$tag.bench.json 2> /dev/null
db=$(find $repo/gpurun_out/prof_$tag -name "*_results.db" | head -1)
python $repo/tools/profile_summary.py "$db" "round 6, profile $tag: SS_DUAL_STREAM=0 SS_UNET_BRANCHES=0 SS_UNET_WGRAD_STREAM=0 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras $* (5 steps traced, one HIP stream)" $repo/gpurun_out/prof_${tag}_kernel_stats.md > /dev/null
python $repo/tools/profile_shapes.py "$db" $repo/gpurun_out/prof_${tag}_shapes.md $subs > /dev/null 2> $repo/gpurun_out/prof_${tag}_shapes.err
rm -rf $repo/gpurun_out/prof_$tag
