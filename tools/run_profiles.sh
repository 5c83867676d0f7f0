#!/bin/bash
# Collect the round's rocprofv3 evidence on a GPU box (run through gpurun from the repo root):
#   1. --kernel-trace of the default bench command        -> gpurun_out/prof_$1/trace
#   2. --pmc pass with the SQ occupancy / MFMA-busy set    -> gpurun_out/prof_$1/sq
#   3. --pmc FETCH_SIZE and 4. --pmc WRITE_SIZE (own passes, MI355X_MICROARCH.md HBM section) -> .../fetch, .../write
# Counters are never combined with sys/hip/hsa tracing.
tag=${1:-r01c}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python $root/bench.py --profile-steps 0 --no-cpu-baseline --small-batch-steps 0 --fp32-steps 0 --f16x3-steps 0 --windows 1"
rocprofv3 --kernel-trace --stats --output-format rocpd -d $out/trace -o bench -- $B --steps 10 --warmup 2 > $out/trace.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format rocpd -d $out/sq -o bench -- $B --steps 2 --warmup 1 > $out/sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format rocpd -d $out/fetch -o bench -- $B --steps 2 --warmup 1 > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format rocpd -d $out/write -o bench -- $B --steps 2 --warmup 1 > $out/write.log 2>&1
cd $root
grep -ho '"value": [0-9.]*, "unit": "steps/s".\{0,80\}' $out/*.log
ls -la $out/*/
