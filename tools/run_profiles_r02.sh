#!/bin/bash
# round-2 profile collection (same passes as tools/run_profiles.sh; r02a = mid-round, r02b = end of round)
tools/run_profiles.sh r02b
python tools/prof_summary.py $(find gpurun_out/prof_r02b/trace -name "*.db" | head -1) 12 > gpurun_out/prof_r02b/kernel_trace.md
python tools/pmc_derive.py $(find gpurun_out/prof_r02b/sq -name "*.db" | head -1) > gpurun_out/prof_r02b/pmc_derived.md
python tools/pmc_summary.py $(find gpurun_out/prof_r02b/fetch -name "*.db" | head -1) > gpurun_out/prof_r02b/fetch.md
python tools/pmc_summary.py $(find gpurun_out/prof_r02b/write -name "*.db" | head -1) > gpurun_out/prof_r02b/write.md
python tools/prof_gaps.py $(find gpurun_out/prof_r02b/trace -name "*.db" | head -1) > gpurun_out/prof_r02b/gaps.txt 2>&1
rm -rf gpurun_out/prof_r02b/*/*/  # drop the big raw databases from the merge-back (keep the summaries)
ls -la gpurun_out/prof_r02b
