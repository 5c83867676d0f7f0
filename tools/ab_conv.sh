#!/bin/bash
# same-box comparison of two libraries on the layer micro-benchmark: tools/ab_conv.sh build/libpfhip_prev.so <filter>
cp polyffusion_amd/libpfhip.so /tmp/new.so
for rep in 1 2; do
  for v in prev new; do
    if [ $v = prev ]; then cp "$1" polyffusion_amd/libpfhip.so; else cp /tmp/new.so polyffusion_amd/libpfhip.so; fi
    echo "== $v"; python tools/bench_conv.py bf16x3 $2 2>&1 | grep -v amdgpu.ids | grep "^$2"
  done
done
cp /tmp/new.so polyffusion_amd/libpfhip.so
