#!/bin/bash
# Same-box A/B of diagnostic builds of the library (SS_LIB_PATH): tools/ab_lib.sh "<kernel substrings>" build_alt/lib_a.so build_alt/lib_b.so ...
# Per build: the per-(kernel, grid) table of the single-stream CycleGAN step for the named kernels, then the two-stream step time.
subs=$1; shift
repo=$(cd "$(dirname "$0")/.." && pwd)
for lib in "$@"; do
  tag=$(basename $lib .so)
  export SS_LIB_PATH=$repo/$lib
  bash $repo/tools/prof_shapes_any.sh $tag "$subs" --skip-unet
  echo "== $tag"; grep -v "^$" $repo/gpurun_out/prof_${tag}_shapes.md | head -${AB_ROWS:-14} | cut -c1-150
  (cd $repo && bash tools/ab_cg.sh - -)
done
