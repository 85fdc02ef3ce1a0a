#!/bin/bash
# per-(kernel, grid) table of the full step on one stream for kernels matching the given substrings: tools/prof_shapes_any.sh <tag> "sub1 sub2" [bench args]
tag=$1; subs=$2; shift; shift
repo=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
SS_DUAL_STREAM=0 SS_UNET_BRANCHES=0 SS_UNET_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -d $repo/gpurun_out/prof_$tag -- python $repo/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras "$@" > /dev/null 2> /dev/null
db=$(find $repo/gpurun_out/prof_$tag -name "*_results.db" | head -1)
python $repo/tools/profile_shapes.py "$db" $repo/gpurun_out/prof_${tag}_shapes.md $subs > /dev/null
rm -rf $repo/gpurun_out/prof_$tag
