#!/bin/bash
# round 6, GPU call 3: batched weight preparation + fp64 partials of the fused norm finalize: suite, flipped-step counts, A/B at batch 1, dispatch count
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_wcache_gpu.py tests/test_layers_gpu.py -k "wcache or batched or norm" -q 2>&1 | tail -15 ) > gpurun_out/r06_d_pytest_quick.txt
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > gpurun_out/r06_d_pytest.txt
timeout 600 python -m pytest tests/test_nets_gpu.py -k vs_reference_goldens -q -s 2>&1 | grep -E "differ from|passed|failed|elements took" > gpurun_out/r06_d_flips.txt
python tools/norm_fuse_diag.py > gpurun_out/r06_d_norm_diag.txt 2>&1
( bash tools/ab_env.sh SS_WPREP_BATCH "0 1" "--global-batch 1" 3 ) > gpurun_out/r06_d_ab_wprep_b1.txt 2>&1
( bash tools/ab_env.sh SS_WPREP_BATCH "0 1" "" 2 ) > gpurun_out/r06_d_ab_wprep_b8.txt 2>&1
bash tools/profile_run.sh r06_d_b1 --global-batch 1
