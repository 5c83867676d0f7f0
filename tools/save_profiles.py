#!/usr/bin/env python
"""Copy the summaries of one tools/run_profiles.sh collection (gpurun_out/prof_<tag>/) into profiles/ with their provenance headers.
usage: python tools/save_profiles.py <tag> "<one-line description of the state profiled>" """
import os
import re
import sys

tag, what = sys.argv[1], sys.argv[2]
src = os.path.join("gpurun_out", f"prof_{tag}")
rate = "?"
for line in open(os.path.join(src, "trace.log"), errors="replace"):
    m = re.search(r'"value": ([0-9.]+), "unit": "steps/s"', line)
    if m:
        rate = m.group(1)
B = "python bench.py --profile-steps 0 --no-cpu-baseline --small-batch-steps 0 --fp32-steps 0 --f16x3-steps 0 --windows 1"
kt = open(os.path.join(src, "kernel_trace.md")).read()
gaps = open(os.path.join(src, "gaps.txt")).read().strip().splitlines()[-2:]
with open(f"profiles/{tag}_kernel_trace_bench_b16_bf16x3.md", "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats summary, {what}\n\n")
    f.write(f"Command (tools/run_profiles.sh {tag}, pass 1): `cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format rocpd "
            f"-d gpurun_out/prof_{tag}/trace -o bench -- {B} --steps 10 --warmup 2`  (bench line of the traced run: {rate} steps/s)\n\n")
    f.write("12 steps traced (2 warm-up + 10 timed; chord-encoder launches included once). Table from tools/prof_summary.py.\n\n")
    f.write(kt)
    f.write("\nDevice timeline (tools/prof_gaps.py): " + "  ".join(g.strip() for g in gaps) + "\n")
pm = open(os.path.join(src, "pmc_derived.md")).read()
with open(f"profiles/{tag}_pmc_derived_bf16x3.md", "w") as f:
    f.write(f"# rocprofv3 --pmc derived metrics, {what}\n\n")
    f.write(f"Pass 2 of tools/run_profiles.sh {tag}: `rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY "
            f"SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format rocpd -- {B} --steps 2 --warmup 1`; derived with "
            "tools/pmc_derive.py (MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles)). Counter passes serialise kernels, so durations here "
            "are longer than in the kernel trace.\n\n")
    f.write(pm)
    for name, title in (("fetch.md", "FETCH_SIZE (pass 3)"), ("write.md", "WRITE_SIZE (pass 4)")):
        pth = os.path.join(src, name)
        if os.path.exists(pth):
            f.write(f"\n## {title}\n\n" + open(pth).read())
print("wrote profiles/%s_*" % tag)
