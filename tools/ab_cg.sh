#!/bin/bash
# Same-box A/B of the CycleGAN step alone: tools/ab_cg.sh "ENV=.." ...  ("-" = default environment)
for e in "$@"; do
  if [ "$e" == "-" ]; then e=""; fi
  env $e python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --skip-unet 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[%s] cyclegan %.3f tiles/s  %.2f ms/step (median %.2f)' % ('$e', d['value'], d['ms_per_step'], d['median_ms_per_step']))"
done
