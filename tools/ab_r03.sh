#!/bin/bash
# same-box A/B: the round-3 tree (build/r03, a git worktree of 43fbdfc built in place) against the working tree; alternating runs
# usage: bash tools/ab_r03.sh [reps] [extra bench flags]   -> gpurun_out/ab/ab_r03.txt
reps=${1:-3}; shift
mkdir -p gpurun_out/ab
F="--steps 50 --warmup 5 --no-cpu-baseline --profile-steps 0 --small-batch-steps 0 --fp32-steps 0 --f16x3-steps 0"
: > gpurun_out/ab/ab_r03.txt
for rep in $(seq $reps); do
  (cd build/r03 && python bench.py $F "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r03  value', d['value'], 'ms', d['ms_per_step'])") | tee -a gpurun_out/ab/ab_r03.txt
  python bench.py $F "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('HEAD value', d['value'], 'ms', d['ms_per_step'], 'windows', d.get('windows_ms_per_step'), 'sclk', d.get('sclk_mhz',{}).get('per_window'))" | tee -a gpurun_out/ab/ab_r03.txt
done
