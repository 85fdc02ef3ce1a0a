#!/bin/bash
cd "$(dirname "$0")/.."
for f in 0 1 2 3; do
  echo "SS_NORM_FUSE_FIN=$f" >> gpurun_out/r06_e_flips.txt
  SS_NORM_FUSE_FIN=$f timeout 600 python -m pytest tests/test_nets_gpu.py -k "vs_reference_goldens or prepad" -q -s 2>&1 | grep -E "differ from|passed|failed|elements took|Error|assert" | cut -c1-300 >> gpurun_out/r06_e_flips.txt
done
echo "SS_DUAL_STREAM=0 fuse 3" >> gpurun_out/r06_e_flips.txt
SS_DUAL_STREAM=0 timeout 600 python -m pytest tests/test_nets_gpu.py -k "vs_reference_goldens" -q -s 2>&1 | grep -E "differ from|passed|failed" | cut -c1-300 >> gpurun_out/r06_e_flips.txt
timeout 600 python -m pytest tests/test_nets_gpu.py -k "prepad" -q 2>&1 | tail -60 > gpurun_out/r06_e_prepad.txt
timeout 600 python -m pytest tests/test_wcache_gpu.py tests/test_layers_gpu.py -k "batched or fused_finalize" -q 2>&1 | tail -5 > gpurun_out/r06_e_quick.txt
