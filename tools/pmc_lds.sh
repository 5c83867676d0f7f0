#!/bin/bash
# LDS-side counters of the conv micro-benchmark shapes (one --pmc pass, kernel-trace only)
root=$(pwd); out=$root/gpurun_out/pmc_lds_${2:-x}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*LDS[A-Z_0-9]*" | sort -u > $out/lds_counters.txt
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format rocpd -d $out/p -o b -- python $root/tools/bench_conv.py bf16x3 "$1" > $out/log.txt 2>&1
cd $root
python tools/pmc_summary.py $(find $out/p -name "*.db") > $out/summary.md 2>&1
cat $out/lds_counters.txt | tr '\n' ' '; echo; cat $out/summary.md
