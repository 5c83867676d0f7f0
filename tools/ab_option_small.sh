#!/bin/bash
# same-box A/B of one plan option on the small-batch legs: bash tools/ab_option_small.sh mlp_fused [reps]   (auto vs forced on)
opt=$1; reps=${2:-2}
mkdir -p gpurun_out/ab
F="--steps 20 --warmup 5 --windows 1 --no-cpu-baseline --profile-steps 0 --fp32-steps 0 --f16x3-steps 0 --no-pmc --small-batch-steps 40"
: > gpurun_out/ab/ab_small_$opt.txt
for rep in $(seq $reps); do
  for v in auto 0 1; do
    if [ $v = auto ]; then o=""; else o="--option $opt=$v"; fi
    python bench.py $F $o 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['small_batch']; print('$opt=$v b16', d['value'], 'b8 replay ms', s['batch8']['graph_replay_ms_per_step'], 'b1 replay ms', s['batch1']['graph_replay_ms_per_step'])" | tee -a gpurun_out/ab/ab_small_$opt.txt
  done
done
