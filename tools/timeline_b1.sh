cd /tmp && export TMPDIR=/tmp
repo=$GRAFT_REPO_ROOT
for what in "--skip-unet" "--only-unet" ""; do
  tag=$(echo "b1$what" | tr -d ' -')
  rocprofv3 --kernel-trace -d $repo/gpurun_out/tl_$tag -- python $repo/bench.py --global-batch 1 --steps 6 --warmup 4 --no-cpu-baseline --no-extras $what > $repo/gpurun_out/tl_$tag.json 2>/dev/null
  db=$(find $repo/gpurun_out/tl_$tag -name "*_results.db" | head -1)
  python $repo/tools/timeline_summary.py "$db" $repo/gpurun_out/tl_${tag}_timeline.md 100 > /dev/null
  sqlite3 "$db" "pragma table_info(kernels)" > $repo/gpurun_out/tl_cols.txt 2>/dev/null
  rm -rf $repo/gpurun_out/tl_$tag
done
cd $repo; python bench.py --global-batch 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > gpurun_out/r05_b_bench_b1.json
