# HIP_FORCE_DEV_KERNARG: the package's default (1, set in _lib.py before the runtime initialises) against an explicit 0, same box
for r in 1 2; do for v in 0 default; do for args in "--global-batch 1" "--only-unet" ""; do
  if [ $v = 0 ]; then export HIP_FORCE_DEV_KERNARG=0; else unset HIP_FORCE_DEV_KERNARG; fi
  echo -n "HIP_FORCE_DEV_KERNARG=$v [$args] : "; python bench.py $args --steps 15 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['median_ms_per_step'])"
done; done; done
