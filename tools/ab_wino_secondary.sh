#!/bin/bash
# AUTO vs direct plan on the secondary workloads (batch 1 / 8, config 3, config 4 shape): same box, alternating
mkdir -p gpurun_out/ab
: > gpurun_out/ab/ab_conv_wino_secondary.txt
for rep in 1 2; do
  for v in "--option conv_wino=0" ""; do
    python bench.py --steps 20 --warmup 3 --windows 1 --no-cpu-baseline --profile-steps 0 --fp32-steps 0 --f16x3-steps 0 --no-long-parity $v 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); sb=d['small_batch']
print('[$v] B16', d['ms_per_step'], 'B1', sb['batch1']['eager_ms_per_step'], 'B8', sb['batch8']['eager_ms_per_step'], 'config3', d['config3']['ms_per_step'], 'config4', d['config4_shape']['ms_per_step'])" | tee -a gpurun_out/ab/ab_conv_wino_secondary.txt
  done
done
