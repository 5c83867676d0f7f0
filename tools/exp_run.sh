#!/bin/bash
# run a command against each experiment library in build/exp (GPU box): tools/exp_run.sh "<cmd>" name1 name2 ...
cmd="$1"; shift
cp polyffusion_amd/libpfhip.so /tmp/pf_keep.so
echo "== base"; eval "$cmd"
for n in "$@"; do cp build/exp/libpfhip_$n.so polyffusion_amd/libpfhip.so; echo "== $n"; eval "$cmd"; done
cp /tmp/pf_keep.so polyffusion_amd/libpfhip.so
