cp polyffusion_amd/libpfhip.so /tmp/keep.so
for rep in 1 2; do
for v in base "$@"; do
  if [ $v = base ]; then cp /tmp/keep.so polyffusion_amd/libpfhip.so; else cp build/exp/libpfhip_$v.so polyffusion_amd/libpfhip.so; fi
  echo -n "$v: "; python bench.py --steps 40 --warmup 5 --no-cpu-baseline --profile-steps 0 --small-batch-steps 0 --fp32-steps 0 --f16x3-steps 0 2>&1 | grep -o '"value": [0-9.]*'
done
done
cp /tmp/keep.so polyffusion_amd/libpfhip.so
