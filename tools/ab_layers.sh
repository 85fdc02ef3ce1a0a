#!/bin/bash
# Same-box A/B of tools/bench_layers.py under different environments: tools/ab_layers.sh "ENV=.." "ENV=.." ...   ("-" = none)
for e in "$@"; do
  if [ "$e" == "-" ]; then e=""; fi
  echo "== [$e]"
  env $e python tools/bench_layers.py --n 8 2>/dev/null | tail -13
done
