#!/bin/bash
# same-box alternating A/B/C of the Winograd plan option: direct plan (conv_wino=0), AUTO (no pin), every qualifying conv (conv_wino=1)
reps=${1:-3}
mkdir -p gpurun_out/ab
F="--steps 50 --warmup 5 --no-cpu-baseline --profile-steps 0 --small-batch-steps 0 --fp32-steps 0 --f16x3-steps 0"
: > gpurun_out/ab/ab_conv_wino.txt
for rep in $(seq $reps); do
  for v in "--option conv_wino=0" "" "--option conv_wino=1"; do
    python bench.py $F $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v] value', d['value'], 'ms', d['ms_per_step'], 'launches', d['config']['launches_per_step'], 'sclk', d['sclk_mhz']['median_window']['median'])" | tee -a gpurun_out/ab/ab_conv_wino.txt
  done
done
