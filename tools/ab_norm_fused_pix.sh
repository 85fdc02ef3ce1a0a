# A/B of the one-launch norm kernels' size limit (ss_config norm_fused_pix / SS_NORM_FUSED_PIX) at per-GPU batch 1
for pix in 1024 4096 16384 65536; do
  for what in "--only-unet" "--skip-unet"; do
    echo "SS_NORM_FUSED_PIX=$pix $what"; SS_NORM_FUSED_PIX=$pix python bench.py --global-batch 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras $what 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['median_ms_per_step'])"
  done
done
