#!/bin/bash
# Build a -DPF_TRACE copy of the library (cycle stamps in the conv / GEMM / attention kernels) as build/libpfhip_trace.so
# without touching the in-tree libpfhip.so.  Use on the GPU box:  cp build/libpfhip_trace.so polyffusion_amd/libpfhip.so
set -e
cd "$(dirname "$0")/.."
mkdir -p build/trace_obj
for f in conv_mfma conv_bf16x3 conv_wino gemm_planes_bf3 mlp_fused_bf3 attention attention_bf3 norm_stats small_kernels encoders comm unet; do
  extra=""; [ $f = attention_bf3 ] && extra="-fno-slp-vectorize"   # polyffusion_amd/build.py EXTRA_FLAGS
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $extra -DPF_TRACE -c polyffusion_amd/csrc/$f.hip -o build/trace_obj/$f.o ) &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libpfhip_trace.so build/trace_obj/*.o -ldl
ls -la build/libpfhip_trace.so
