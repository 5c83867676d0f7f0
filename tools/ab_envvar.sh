#!/bin/bash
# bash ab_env.sh VAR VAL_A VAL_B reps
var=$1; a=$2; b=$3; reps=${4:-3}
F="--steps 50 --warmup 5 --no-cpu-baseline --profile-steps 0 --small-batch-steps 0 --fp32-steps 0 --f16x3-steps 0"
for rep in $(seq $reps); do
  for v in $a $b; do
    env $var=$v python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v value', d['value'], 'ms', d['ms_per_step'], 'launches', d['config']['launches_per_step'], 'sclk', d['sclk_mhz']['median_window']['median'])"
  done
done
