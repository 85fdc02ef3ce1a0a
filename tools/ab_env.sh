# generic A/B of one environment switch on the bench: tools/ab_env.sh VAR "v0 v1" "<bench args>" [repeats]
var=$1; vals=$2; args=$3; reps=${4:-2}
for r in $(seq $reps); do for v in $vals; do
  echo -n "$var=$v $args : "; env $var=$v python bench.py $args --steps 15 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['median_ms_per_step'])"
done; done
