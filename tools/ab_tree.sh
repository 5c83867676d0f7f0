#!/bin/bash
# same-box A/B of the working tree against an exported older tree (build/exp/base_tree: `git archive <rev> | tar -x` + that revision's
# libpfhip.so): bash tools/ab_tree.sh [reps] ["--option name=v" for the working tree]   -> gpurun_out/ab/ab_tree.txt
reps=${1:-3}; opt=$2
mkdir -p gpurun_out/ab
F="--steps 50 --warmup 5 --no-cpu-baseline --profile-steps 0 --small-batch-steps 0 --fp32-steps 0 --f16x3-steps 0"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], "launches", d["config"]["launches_per_step"], "sclk", d["sclk_mhz"]["median_window"]["median"])'
: > gpurun_out/ab/ab_tree.txt
for rep in $(seq $reps); do
  (cd build/exp/base_tree && python bench.py $F 2>/dev/null) | python -c "$P" base | tee -a gpurun_out/ab/ab_tree.txt
  python bench.py $F 2>/dev/null | python -c "$P" tree | tee -a gpurun_out/ab/ab_tree.txt
  if [ -n "$opt" ]; then python bench.py $F $opt 2>/dev/null | python -c "$P" "tree $opt" | tee -a gpurun_out/ab/ab_tree.txt; fi
done
