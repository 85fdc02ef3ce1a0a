#!/bin/bash
# phase-skipping probe of wgrad_x6_kernel (tile_dbg bits: 1 no global loads, 2 no split / LDS stores, 4 no fragment reads / MFMAs); usage: tools/wgrad_x6_probe.sh
repo=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for d in 0 1 2 4 3 6 7; do
  SS_TILE_DBG=$d SS_DUAL_STREAM=0 rocprofv3 --kernel-trace --stats -d /tmp/wx_$d -- python $repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --skip-unet > /dev/null 2>&1
  db=$(find /tmp/wx_$d -name "*_results.db" | head -1)
  python $repo/tools/profile_summary.py "$db" "dbg $d" /tmp/wx_$d.md > /dev/null
  echo "dbg=$d: $(grep -E 'wgrad_x6_kernel<128' /tmp/wx_$d.md | cut -d'|' -f3-5)"
done
