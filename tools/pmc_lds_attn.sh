#!/bin/bash
# LDS-side counters of the attention micro-benchmark (one --pmc pass, kernel-trace only): tools/pmc_lds_attn.sh <tag>
root=$(pwd); out=$root/gpurun_out/pmc_lds_attn_${1:-x}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format rocpd -d $out/p -o b -- python $root/tools/bench_attn.py > $out/log.txt 2>&1
cd $root
python tools/pmc_summary.py $(find $out/p -name "*.db") > $out/summary.md 2>&1
cat $out/summary.md
