#!/bin/bash
# A/B two builds of libpfhip.so on the same GPU box: tools/ab.sh build/libpfhip_prev.so  (B = the in-tree library)
set -e
cp polyffusion_amd/libpfhip.so /tmp/new.so
for rep in 1 2; do
  for v in prev new; do
    if [ $v = prev ]; then cp "$1" polyffusion_amd/libpfhip.so; else cp /tmp/new.so polyffusion_amd/libpfhip.so; fi
    echo -n "$v: "; python bench.py --steps 40 --warmup 5 --no-cpu-baseline --profile-steps 0 --small-batch-steps 0 --fp32-steps 0 --f16x3-steps 0 2>&1 | grep -o '"value": [0-9.]*'
  done
done
cp /tmp/new.so polyffusion_amd/libpfhip.so
