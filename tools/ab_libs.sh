#!/bin/bash
# same-box A/B/C of library variants: bash tools/ab_libs.sh "cur t0w8 t01w8" [reps]   (cur = in-tree, others = build/exp/libpfhip_<name>.so)
# per variant: a parity + determinism smoke (first rep), the conv micro-bench, and the step rate; log: gpurun_out/ab/ab_libs.txt
names=$1; reps=${2:-3}
mkdir -p gpurun_out/ab
cp polyffusion_amd/libpfhip.so /tmp/cur.so
for n in $names; do [ $n = cur ] || cp build/exp/libpfhip_$n.so /tmp/$n.so; done
F="--steps 50 --warmup 5 --no-cpu-baseline --profile-steps 0 --small-batch-steps 0 --fp32-steps 0 --f16x3-steps 0"
: > gpurun_out/ab/ab_libs.txt
for rep in $(seq $reps); do
  for n in $names; do
    cp /tmp/$n.so polyffusion_amd/libpfhip.so
    if [ $rep = 1 ]; then
      echo "== $n" | tee -a gpurun_out/ab/ab_libs.txt
      timeout 900 python -m pytest tests/test_gpu_bf16x3.py tests/test_gpu_determinism.py -x -q 2>&1 | tail -2 | tee -a gpurun_out/ab/ab_libs.txt
      python tools/bench_conv.py bf16x3 r 2>&1 | grep -v amdgpu | tee -a gpurun_out/ab/ab_libs.txt
    fi
    python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n value', d['value'], 'ms', d['ms_per_step'], 'sclk', d['sclk_mhz']['median_window']['median'])" | tee -a gpurun_out/ab/ab_libs.txt
  done
done
cp /tmp/cur.so polyffusion_amd/libpfhip.so
