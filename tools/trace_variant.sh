#!/bin/bash
# kernel trace of the default bench command for both element types of the split-precision kernels, same box, back to back:
#   bash tools/trace_variant.sh TAG   -> gpurun_out/prof_TAG/{bf16,f16}.md
tag=${1:-x3}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python $root/bench.py --profile-steps 0 --no-cpu-baseline --small-batch-steps 0 --fp32-steps 0 --f16x3-steps 0 --windows 1 --no-long-parity"
for v in bf16 f16 bf16 f16; do
  PF_X3=$v rocprofv3 --kernel-trace --stats --output-format rocpd -d $out/trace_$v -o bench -- $B --steps 10 --warmup 2 > $out/trace_$v.log 2>&1
  db=$(ls $out/trace_$v/*/*.db $out/trace_$v/*.db 2>/dev/null | head -1)
  python $root/tools/prof_summary.py $db 12 > $out/$v.md
  grep -ho '"value": [0-9.]*' $out/trace_$v.log | head -1
  rm -rf $out/trace_$v
done
