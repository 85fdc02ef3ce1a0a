for st in 0 1 2 4 8; do echo "== stagger $st"; SS_TILE_STAGGER=$st python tools/bench_tile.py --only fwd 2>&1 | grep -E "mrb1.3 1->4|rp1.3 16->16|rp1.3 25->16|rp2.3 32->32|TOTAL" ; done
echo "== skeleton"; SS_TILE_DBG=15 python tools/bench_tile.py --only fwd 2>&1 | grep -E "mrb1.3 1->4|rp1.3 16->16|rp1.3 25->16|rp2.3 32->32|TOTAL"
