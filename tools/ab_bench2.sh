#!/bin/bash
# Same-box A/B of bench.py with the cyclegan / unet / overlapped legs: tools/ab_bench2.sh "ENV=.." ... ("-" = default environment)
for e in "$@"; do
  if [ "$e" == "-" ]; then e=""; fi
  env $e SS_BENCH_LIGHT=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
s='[%s] %.3f tiles/s  %.2f ms/step (median %.2f)' % ('$e', d['value'], d['ms_per_step'], d['median_ms_per_step'])
if 'cyclegan' in d: s += '  cyclegan %.2f  unet %.2f' % (d['cyclegan']['median_ms_per_step'], d['unet']['median_ms_per_step'])
if 'overlapped' in d: s += '  overlapped %.2f' % d['overlapped']['median_ms_per_step']
print(s)"
done
