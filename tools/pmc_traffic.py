#!/usr/bin/env python
"""HBM bytes per launch of the 3x3 conv family from the FETCH_SIZE / WRITE_SIZE passes of tools/run_profiles.sh (the per-kernel tables
tools/pmc_summary.py wrote: average KB per dispatch).  FETCH_SIZE is doubled (gfx950 counts 128-byte requests as 64 B,
MI355X_MICROARCH.md HBM section).  usage: python tools/pmc_traffic.py <tag> > profiles/pmc_traffic_bf16x3.json"""
import json
import re
import sys

tag = sys.argv[1]


def table(path):
    out = {}
    for line in open(path):
        m = re.match(r"\| `(.*?)` \| (\d+) \| ([0-9.e+]+) \|", line)
        if m:
            out[m.group(1)] = (int(m.group(2)), float(m.group(3)))
    return out


fetch, write = table(f"gpurun_out/prof_{tag}/fetch.md"), table(f"gpurun_out/prof_{tag}/write.md")
fam = [k for k in fetch if re.match(r"conv_bf3_kernel<[23],", k)]
launches = sum(fetch[k][0] for k in fam)
f_kb = sum(fetch[k][0] * fetch[k][1] for k in fam)
w_kb = sum(write[k][0] * write[k][1] for k in fam if k in write)
print(json.dumps({
    "kernel": "conv_bf3_kernel<3,...> and <2,...> (all 3x3 instantiations incl. the fused skip projection and the parity-folded UpSample convs, bf16x3 mode)",
    "launches": launches, "fetch_size_kb_sum": f_kb, "write_size_kb_sum": w_kb,
    "hbm_bytes_per_launch": (2 * f_kb + w_kb) * 1024 / launches,
    "note": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 2 --warmup 1` (tools/run_profiles.sh {tag}); "
            "FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md HBM section); KB units x1024"}, indent=1))
