#!/bin/bash
# same-box A/B of the in-tree library against build/exp/libpfhip_prev.so: bash tools/ab_prev.sh [reps]   -> gpurun_out/ab/ab_prev.txt
reps=${1:-3}
mkdir -p gpurun_out/ab
cp polyffusion_amd/libpfhip.so /tmp/cur.so
F="--steps 50 --warmup 5 --no-cpu-baseline --profile-steps 0 --small-batch-steps 0 --fp32-steps 0 --f16x3-steps 0 --no-pmc"
: > gpurun_out/ab/ab_prev.txt
for rep in $(seq $reps); do
  for n in prev cur; do
    if [ $n = prev ]; then cp build/exp/libpfhip_prev.so polyffusion_amd/libpfhip.so; else cp /tmp/cur.so polyffusion_amd/libpfhip.so; fi
    python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n value', d['value'], 'ms', d['ms_per_step'], 'sclk', d['sclk_mhz']['median_window']['median'])" | tee -a gpurun_out/ab/ab_prev.txt
  done
done
cp /tmp/cur.so polyffusion_amd/libpfhip.so
