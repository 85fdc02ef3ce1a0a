#!/bin/bash
# per-(kernel, grid) table of the UNet step on one stream: tools/prof_unet_shapes.sh <tag>   (run on the GPU box)
tag=$1
repo=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
SS_UNET_BRANCHES=0 SS_UNET_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -d $repo/gpurun_out/prof_$tag -- python $repo/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras --only-unet > $repo/gpurun_out/prof_$tag.bench.json 2> /dev/null
db=$(find $repo/gpurun_out/prof_$tag -name "*_results.db" | head -1)
python $repo/tools/profile_shapes.py "$db" $repo/gpurun_out/prof_${tag}_shapes.md norm maxpool tconv gconv > /dev/null
rm -rf $repo/gpurun_out/prof_$tag
