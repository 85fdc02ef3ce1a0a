#!/bin/bash
# Same-box A/B of the per-GPU-batch-1 step (the 8-GPU share): tools/ab_b1.sh "ENV=.." ...  ("-" = default environment)
for e in "$@"; do
  if [ "$e" == "-" ]; then e=""; fi
  env $e python bench.py --global-batch 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[%s] batch 1: %.3f tiles/s  %.2f ms/step (median %.2f)' % ('$e', d['value'], d['ms_per_step'], d['median_ms_per_step']))"
done
