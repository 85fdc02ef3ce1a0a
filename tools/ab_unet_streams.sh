for cfg in "" "0,1,2,3,4" "0,1,2,1,2" "0,1,1,1,1" "1,2,3,4,5"; do
  echo "SS_UNET_STREAMS=$cfg"; SS_UNET_STREAMS=$cfg python bench.py --global-batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-extras --only-unet 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['median_ms_per_step'])"
done
echo "branches off"; SS_UNET_BRANCHES=0 python bench.py --global-batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-extras --only-unet 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['median_ms_per_step'])"
echo "wgrad stream off"; SS_UNET_WGRAD_STREAM=0 python bench.py --global-batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-extras --only-unet 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['median_ms_per_step'])"
