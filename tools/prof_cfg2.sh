# config 2 (MultiResUNet, 256 x 256, batch 16): bf16 / fp16 storage against fp32 storage of the same shape -- bench lines + single-stream kernel stats
repo=$(cd "$(dirname "$0")/.." && pwd)
cd $repo
for dt in bf16 f16 f32; do
  python bench.py --config 2 --dtype $dt --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$dt', j['value'], j['ms_per_step'], j['median_ms_per_step'])"
done
cd /tmp && export TMPDIR=/tmp
for dt in bf16 f32; do
  SS_UNET_BRANCHES=0 SS_UNET_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -d $repo/gpurun_out/prof_cfg2_$dt -- python $repo/bench.py --config 2 --dtype $dt --steps 3 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
  db=$(find $repo/gpurun_out/prof_cfg2_$dt -name "*_results.db" | head -1)
  python $repo/tools/profile_summary.py "$db" "round 5: SS_UNET_BRANCHES=0 SS_UNET_WGRAD_STREAM=0 python bench.py --config 2 --dtype $dt --steps 3 --warmup 2 (5 steps traced, one stream)" $repo/gpurun_out/prof_cfg2_${dt}_kernel_stats.md > /dev/null
  rm -rf $repo/gpurun_out/prof_cfg2_$dt
done
