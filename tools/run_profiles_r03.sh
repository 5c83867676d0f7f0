#!/bin/bash
# round-3 profile collection (same passes as tools/run_profiles.sh; r03a = after the transformer-block fusions)
tools/run_profiles.sh r03a
python tools/prof_summary.py $(find gpurun_out/prof_r03a/trace -name "*.db" | head -1) 12 > gpurun_out/prof_r03a/kernel_trace.md
python tools/pmc_derive.py $(find gpurun_out/prof_r03a/sq -name "*.db" | head -1) > gpurun_out/prof_r03a/pmc_derived.md
python tools/pmc_summary.py $(find gpurun_out/prof_r03a/fetch -name "*.db" | head -1) > gpurun_out/prof_r03a/fetch.md
python tools/pmc_summary.py $(find gpurun_out/prof_r03a/write -name "*.db" | head -1) > gpurun_out/prof_r03a/write.md
python tools/prof_gaps.py $(find gpurun_out/prof_r03a/trace -name "*.db" | head -1) > gpurun_out/prof_r03a/gaps.txt 2>&1
rm -rf gpurun_out/prof_r03a/*/*/  # drop the big raw databases from the merge-back (keep the summaries)
ls -la gpurun_out/prof_r03a
