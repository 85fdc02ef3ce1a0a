#!/bin/bash
# One-off diagnosis (round 6): the workflow's steps 0-5 with fp16 activation storage (repeated with another seed until the MultiResUNet's
# training shows the NaN of the round's first fp16 run), then step 6a (one epoch) under kernel-selection switches.
cd "$(dirname "$0")/.."
R=/tmp/ssr_bisect
run() { name=$1; shift; env "$@" python tools/real_data_eval.py --data _eval_data --out gpurun_out/bisect_$name --root $R --seed 0 --steps 6a --no-instance-scores --set UNET_EPOCHS=1 --set ACTIVATION_STORAGE=f16 > gpurun_out/bisect_$name.log 2>&1; echo "$name: $(tail -1 gpurun_out/bisect_$name/3_UNet_training_log.csv)"; grep SS_NAN_TRAP gpurun_out/bisect_$name.log | cut -c1-900; }
for seed in 0 11 12; do
  python tools/real_data_eval.py --data _eval_data --out gpurun_out/bisect_base --root $R --seed $seed --steps 0,1,2,3,4,5 --set ACTIVATION_STORAGE=f16 > gpurun_out/bisect_base.log 2>&1
  echo "base seed $seed: $(tail -1 gpurun_out/bisect_base.log | cut -c1-120)"
  out=$(run all A=1); echo "$out"
  if echo "$out" | grep -q nan; then
    run trap SS_NAN_TRAP=1
    run sync_only SS_NAN_TRAP=2
    run stride1_only SS_GCONV16_RAGGED=5
    run stride1_3x3_only SS_GCONV16_RAGGED=1
    run strided_only SS_GCONV16_RAGGED=6
    run two_products SS_WINO16_PRODUCTS=3
    run loss_scale_64 SS_F16_LOSS_SCALE=64
    run ragged_off SS_GCONV16_RAGGED=0
    break
  fi
done
