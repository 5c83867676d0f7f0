#!/bin/bash
# same-box A/B of the in-tree library against build/exp/libpfhip_prev.so on the batch-16 headline AND the batch-8 / batch-1 legs:
#   bash tools/ab_prev_small.sh [reps]   -> gpurun_out/ab/ab_prev_small.txt
reps=${1:-3}
mkdir -p gpurun_out/ab
cp polyffusion_amd/libpfhip.so /tmp/cur.so
F="--steps 30 --warmup 5 --windows 3 --no-cpu-baseline --profile-steps 0 --fp32-steps 0 --f16x3-steps 0 --no-pmc --small-batch-steps 40 --no-long-parity"
: > gpurun_out/ab/ab_prev_small.txt
for rep in $(seq $reps); do
  for n in prev cur; do
    if [ $n = prev ]; then cp build/exp/libpfhip_prev.so polyffusion_amd/libpfhip.so; else cp /tmp/cur.so polyffusion_amd/libpfhip.so; fi
    python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['small_batch']; print('$n b16 steps/s', d['value'], '| b8 ms eager/graph', s['batch8']['eager_ms_per_step'], s['batch8']['graph_ms_per_step'], '| b1 ms', s['batch1']['eager_ms_per_step'], s['batch1']['graph_ms_per_step'], '| config3 ms', d['config3']['ms_per_step'])" | tee -a gpurun_out/ab/ab_prev_small.txt
  done
done
cp /tmp/cur.so polyffusion_amd/libpfhip.so
