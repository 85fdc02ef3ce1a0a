#!/bin/bash
# per-kernel times of the one-channel 7x7 layers (tools/c1_layers_time.py) under rocprofv3; usage: tools/c1_prof.sh <tag>
tag=$1
repo=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $repo/gpurun_out/c1prof_$tag -- python $repo/tools/c1_layers_time.py > $repo/gpurun_out/c1prof_$tag.log 2>&1
db=$(find $repo/gpurun_out/c1prof_$tag -name "*_results.db" | head -1)
python $repo/tools/profile_summary.py "$db" "one-channel 7x7 layers, $tag" $repo/gpurun_out/c1prof_${tag}_kernel_stats.md > /dev/null
rm -rf $repo/gpurun_out/c1prof_$tag
