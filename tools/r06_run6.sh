#!/bin/bash
cd "$(dirname "$0")/.."
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > gpurun_out/r06_r_pytest.txt
( bash tools/ab_env.sh SS_NORM_FUSE_FIN "0 2 3" "--global-batch 1" 3 ) > gpurun_out/r06_r_ab_fuse_b1.txt 2>&1
( bash tools/ab_env.sh SS_NORM_FUSE_FIN "0 2" "--config 2" 2 ) > gpurun_out/r06_r_ab_fuse_cfg2.txt 2>&1
