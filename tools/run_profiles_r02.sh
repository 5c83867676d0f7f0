#!/bin/bash
# round-2 profile collection (same passes as tools/run_profiles.sh, tag r02a)
tools/run_profiles.sh r02a
python tools/prof_summary.py $(find gpurun_out/prof_r02a/trace -name "*.db" | head -1) 12 > gpurun_out/prof_r02a/kernel_trace.md
python tools/pmc_derive.py $(find gpurun_out/prof_r02a/sq -name "*.db" | head -1) > gpurun_out/prof_r02a/pmc_derived.md
python tools/pmc_summary.py $(find gpurun_out/prof_r02a/fetch -name "*.db" | head -1) > gpurun_out/prof_r02a/fetch.md
python tools/pmc_summary.py $(find gpurun_out/prof_r02a/write -name "*.db" | head -1) > gpurun_out/prof_r02a/write.md
python tools/prof_gaps.py $(find gpurun_out/prof_r02a/trace -name "*.db" | head -1) > gpurun_out/prof_r02a/gaps.txt 2>&1
rm -rf gpurun_out/prof_r02a/*/*/  # drop the big raw databases from the merge-back (keep the summaries)
ls -la gpurun_out/prof_r02a
