#!/bin/bash
# Same-box A/B of one BASELINE config: tools/ab_cfg.sh <config> "ENV=.." ...  ("-" = default environment)
cfg=$1; shift
for e in "$@"; do
  if [ "$e" == "-" ]; then e=""; fi
  env $e python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[%s] config $cfg: %.3f tiles/s  %.2f ms/step (median %.2f)' % ('$e', d['value'], d['ms_per_step'], d['median_ms_per_step']))"
done
