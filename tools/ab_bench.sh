#!/bin/bash
# Same-box A/B of bench.py under different environments: tools/ab_bench.sh "ENV1=.. ENV2=.." "ENV=.." ...   ("-" = no extra environment)
# Prints tiles/s, ms per step (cyclegan / unet split with --split as first argument)
extra="--no-extras"
if [ "$1" == "--split" ]; then extra=""; shift; fi
for e in "$@"; do
  if [ "$e" == "-" ]; then e=""; fi
  env $e python bench.py --steps 10 --warmup 3 --no-cpu-baseline $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
s='[%s] %.3f tiles/s  %.2f ms/step (median %.2f)' % ('$e', d['value'], d['ms_per_step'], d['median_ms_per_step'])
if 'cyclegan' in d: s += '  cyclegan %.2f  unet %.2f' % (d['cyclegan']['median_ms_per_step'], d['unet']['median_ms_per_step'])
if 'roofline' in d:
    r=d['roofline']; s += '  dominant %s %.4f ms frac %.4f' % (r['kernel'], r['avg_launch_ms'], r['frac'])
print(s)"
done
