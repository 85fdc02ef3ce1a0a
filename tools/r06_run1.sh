#!/bin/bash
# round 6, first GPU call: the GPU suite, the new bench line (per_gpu_share leg), norm walk-order and fused-finalize A/Bs, batch-1 dispatch count
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r06_a_pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_a_bench_n1.json 2> gpurun_out/r06_a_bench.err
cp profiles/bench_tables_last.json gpurun_out/r06_a_bench_tables.json 2>/dev/null
( bash tools/ab_env.sh SS_NORM_ORDER "0 2 5 7" "" 2 ) > gpurun_out/r06_a_ab_norm_order.txt 2>&1
( bash tools/ab_env.sh SS_NORM_FUSE_FIN "0 1" "--global-batch 1" 3 ) > gpurun_out/r06_a_ab_fuse_b1.txt 2>&1
( bash tools/ab_env.sh SS_NORM_FUSE_FIN "0 1" "--config 3" 2 ) > gpurun_out/r06_a_ab_fuse_cfg3.txt 2>&1
( bash tools/ab_env.sh SS_NORM_FUSE_FIN "0 1" "--config 2" 2 ) > gpurun_out/r06_a_ab_fuse_cfg2.txt 2>&1
bash tools/profile_run.sh r06_a_b1 --global-batch 1
