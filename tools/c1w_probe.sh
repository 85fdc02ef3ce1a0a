#!/bin/bash
# phase-skipping probe of wgrad_c1_x3h_kernel (tile_dbg bits: 1 no X loads, 2 no U build, 4 no MFMAs, 8 no X split); usage: tools/c1w_probe.sh
repo=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for d in 0 1 2 4 8 6 14 15; do
  SS_TILE_DBG=$d rocprofv3 --kernel-trace --stats -d /tmp/c1w_$d -- python $repo/tools/c1_layers_time.py > /dev/null 2>&1
  db=$(find /tmp/c1w_$d -name "*_results.db" | head -1)
  python $repo/tools/profile_summary.py "$db" "dbg $d" /tmp/c1w_$d.md > /dev/null
  echo "dbg=$d in1: $(grep -E 'conv_in1_x3h' /tmp/c1w_$d.md | sed -E 's/.*conv_in1_x3h_kernel<(true|false)>[^|]*\| ([0-9]+) \| ([0-9.]+) \| ([0-9.]+).*/\1 \4us/' | tr '\n' ' ')"
  echo "dbg=$d: $(grep -E 'wgrad_c1_x3h' /tmp/c1w_$d.md | sed -E 's/.*wgrad_c1_x3h_kernel<([01])>[^|]*\| ([0-9]+) \| ([0-9.]+) \| ([0-9.]+).*/mode\1 \4us/' | tr '\n' ' ')"
done
