#!/bin/bash
# ring-depth A/B/C for the K-split 16x16-level conv tile: build/exp/libpfhip_ring{3,7}.so vs the in-tree library (ring 5)
mkdir -p gpurun_out/ring
python -m pytest tests/test_gpu_round4.py -x -q -k plan_options 2>&1 | grep -E "^E|assert|passed|failed" | head -20
cp polyffusion_amd/libpfhip.so /tmp/ring5.so
cp build/exp/libpfhip_ring3.so /tmp/ring3.so; cp build/exp/libpfhip_ring7.so /tmp/ring7.so
F="--steps 50 --warmup 5 --no-cpu-baseline --profile-steps 0 --small-batch-steps 0 --fp32-steps 0"
for rep in 1 2 3; do
  for v in ring3 ring5 ring7; do
    cp /tmp/$v.so polyffusion_amd/libpfhip.so
    if [ $rep = 1 ]; then python tools/dump_launches.py 16 > gpurun_out/ring/dump_$v.txt 2>&1; echo -n "$v r16 convs: "; grep conv3x3 gpurun_out/ring/dump_$v.txt | awk '{print $4}' | sed -n 15,23p | tr '\n' ' '; echo; fi
    python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v value', d['value'], 'ms', d['ms_per_step'], 'sclk', d['sclk_mhz']['median_window'])"
  done
done
cp /tmp/ring5.so polyffusion_amd/libpfhip.so
