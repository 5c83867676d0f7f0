#!/bin/bash
# round-4 profile collection (same passes as tools/run_profiles.sh, one timed window): tools/run_profiles_r04.sh [tag]
# (r04a = after hoisting the step-invariant prefix / in-kernel noise, r04b = before, r04c = after the buffer-form direct-to-LDS loads)
tag=${1:-r04a}
tools/run_profiles.sh $tag
python tools/prof_summary.py $(find gpurun_out/prof_$tag/trace -name "*.db" | head -1) 12 > gpurun_out/prof_$tag/kernel_trace.md
python tools/pmc_derive.py $(find gpurun_out/prof_$tag/sq -name "*.db" | head -1) > gpurun_out/prof_$tag/pmc_derived.md
python tools/pmc_summary.py $(find gpurun_out/prof_$tag/fetch -name "*.db" | head -1) > gpurun_out/prof_$tag/fetch.md
python tools/pmc_summary.py $(find gpurun_out/prof_$tag/write -name "*.db" | head -1) > gpurun_out/prof_$tag/write.md
python tools/prof_gaps.py $(find gpurun_out/prof_$tag/trace -name "*.db" | head -1) > gpurun_out/prof_$tag/gaps.txt 2>&1
rm -rf gpurun_out/prof_$tag/*/*/  # drop the big raw databases from the merge-back (keep the summaries)
ls -la gpurun_out/prof_$tag
