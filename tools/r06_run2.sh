#!/bin/bash
# round 6, GPU call 2: whole GPU suite (no -x) after the SyncBN packing / metrics-interval / flipped-step-test changes; the flipped-step
# counts of the reference-golden CycleGAN steps with and without the fused norm finalize
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > gpurun_out/r06_b_pytest.txt
for f in 0 1; do
  echo "SS_NORM_FUSE_FIN=$f" >> gpurun_out/r06_b_flips.txt
  SS_NORM_FUSE_FIN=$f timeout 600 python -m pytest tests/test_nets_gpu.py -k vs_reference_goldens -q -s 2>&1 | grep -E "differ from|passed|failed|elements took" >> gpurun_out/r06_b_flips.txt
done
