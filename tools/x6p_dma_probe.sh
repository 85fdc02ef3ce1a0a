for dbg in 0 16 32; do
SS_TILE_DBG=$dbg SS_DUAL_STREAM=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --skip-unet 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('dbg', $dbg, {n:(v['avg_ms'],v['frac']) for n,v in k.items() if n.startswith(('gemm_x6p','gemm_tn'))})"
done
