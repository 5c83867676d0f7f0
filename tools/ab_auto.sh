#!/bin/bash
# same-box A/B of one plan option, OFF against AUTO (the plan's own choice): bash tools/ab_auto.sh conv_wino [reps]  -> gpurun_out/ab/abauto_<option>.txt
opt=$1; reps=${2:-3}
mkdir -p gpurun_out/ab
F="--steps 50 --warmup 5 --no-cpu-baseline --profile-steps 0 --small-batch-steps 0 --fp32-steps 0 --f16x3-steps 0"
: > gpurun_out/ab/abauto_$opt.txt
for rep in $(seq $reps); do
  for v in off auto; do
    if [ $v = off ]; then O="--option $opt=0"; else O=""; fi
    python bench.py $F $O 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$opt=$v value', d['value'], 'ms', d['ms_per_step'], 'launches', d['config']['launches_per_step'], 'sclk', d['sclk_mhz']['median_window']['median'])" | tee -a gpurun_out/ab/abauto_$opt.txt
  done
done
