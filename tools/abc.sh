set -e
cp polyffusion_amd/libpfhip.so /tmp/new.so
for rep in 1 2 3; do
  for v in prev new natall; do
    case $v in prev) cp build/libpfhip_prev.so polyffusion_amd/libpfhip.so;; new) cp /tmp/new.so polyffusion_amd/libpfhip.so;; natall) cp build/exp/libpfhip_natall.so polyffusion_amd/libpfhip.so;; esac
    echo -n "$v: "; python bench.py --steps 40 --warmup 5 --no-cpu-baseline --profile-steps 0 --small-batch-steps 0 --fp32-steps 0 --f16x3-steps 0 2>&1 | grep -o '"value": [0-9.]*'
  done
done
cp /tmp/new.so polyffusion_amd/libpfhip.so
