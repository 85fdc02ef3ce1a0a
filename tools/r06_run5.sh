#!/bin/bash
cd "$(dirname "$0")/.."
python tools/flips_by_tensor.py 0 3 2>&1 | grep total > gpurun_out/r06_q_flips.txt
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > gpurun_out/r06_q_pytest.txt
python tools/norm_fuse_diag.py 2>&1 | grep -E "fuse (0|3)" | head -24 > gpurun_out/r06_q_diag.txt
