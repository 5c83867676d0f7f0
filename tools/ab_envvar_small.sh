#!/bin/bash
# same-box A/B of an environment switch on the batch-16 / batch-8 / batch-1 legs: bash tools/ab_envvar_small.sh VAR A B [reps]
var=$1; a=$2; b=$3; reps=${4:-2}
F="--steps 30 --warmup 5 --windows 3 --no-cpu-baseline --profile-steps 0 --fp32-steps 0 --f16x3-steps 0 --no-pmc --small-batch-steps 40 --no-long-parity"
for rep in $(seq $reps); do
  for v in $a $b; do
    env $var=$v python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['small_batch']; print('$var=$v b16 steps/s', d['value'], '| b8 ms eager/graph', s['batch8']['eager_ms_per_step'], s['batch8']['graph_ms_per_step'], '| b1 ms', s['batch1']['eager_ms_per_step'], s['batch1']['graph_ms_per_step'], '| config3 ms', d['config3']['ms_per_step'])"
  done
done
