#!/bin/bash
# same-box A/B of an environment switch (experiments only): bash tools/ab_env.sh PF_X 0 4 [reps]   -> gpurun_out/ab/ab_env_<name>.txt
name=$1; a=$2; b=$3; reps=${4:-3}
mkdir -p gpurun_out/ab
F="--steps 50 --warmup 5 --no-cpu-baseline --profile-steps 0 --small-batch-steps 0 --fp32-steps 0 --f16x3-steps 0 --no-pmc"
: > gpurun_out/ab/ab_env_$name.txt
for rep in $(seq $reps); do
  for v in $a $b; do
    env $name=$v python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name=$v value', d['value'], 'ms', d['ms_per_step'], 'sclk', d['sclk_mhz']['median_window']['median'])" | tee -a gpurun_out/ab/ab_env_$name.txt
  done
done
