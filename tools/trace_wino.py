#!/usr/bin/env python
"""Cycle-stamp timeline of three workgroups of the pipelined Winograd kernel (needs a PF_TRACE build: tools/build_trace.sh, then
cp build/libpfhip_trace.so polyffusion_amd/libpfhip.so on the GPU box).  usage: python tools/trace_wino.py [shape filter] [B]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyffusion_amd import _lib  # noqa: E402
import tools.bench_wino as bw  # noqa: E402

lib = _lib.load()
flt = sys.argv[1] if len(sys.argv) > 1 else "r128_64_64"
B = sys.argv[2] if len(sys.argv) > 2 else "16"
sys.argv = ["x", B, flt]
bw.main()
lib.pf_debug_wino_trace_clear()
sys.argv = ["x", B, flt]
bw.SHAPES = [s for s in bw.SHAPES if s[0] == flt] or bw.SHAPES
bw.main()
torch.cuda.synchronize()
buf = (C.c_ulonglong * 768)()
lib.pf_debug_wino_trace_read(buf, 768)
a = np.array(buf[:], dtype=np.int64)
for base, tag in ((0, "block 0"), (256, "block n/2"), (512, "block last")):
    t = a[base:base + 256]
    t = t[t > 0]
    if len(t) < 8:
        continue
    d = np.diff(t)
    nst = len(d) - 5 - 6
    print(f"{tag}: {len(t)} stamps, total {t[-1] - t[0]} ticks")
    print("   prologue [setup+loads issued, stage chunk 0, stage 3 px + loads, barrier, frags 0 -> first step]:", d[:5].tolist())
    steps = d[5:5 + nst]
    print("   steps:", steps.tolist())
    print("   mean step", float(steps.mean()) if len(steps) else 0, " chunks", len(steps) / 4)
    print("   epilogue [last step, S + exchange write, barrier, read + Y + stores, stats]:", d[5 + nst:].tolist())
