#!/usr/bin/env python
"""Cycle-stamp timeline of one wave of the ping-pong 3x3 kernel (PF_TRACE build, PF_X=4)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyffusion_amd import _lib
import tools.bench_conv as bc
lib = _lib.load()
name = sys.argv[1] if len(sys.argv) > 1 else "r32_256_256"
sys.argv = ["x", "bf16x3", name]
lib.pf_debug_trace_clear()
bc.main()
torch.cuda.synchronize()
buf = (C.c_ulonglong * 8192)()
lib.pf_debug_trace_read(buf, 8192)
a = np.array(buf[:], dtype=np.int64)
for base, tag in ((0, "block0"), (2048, "block301"), (4096, "blocklast")):
    t = a[base:base + 2048]; t = t[t > 0]
    if len(t) < 8: continue
    d = np.diff(t)
    print(tag, "n", len(t), "total cycles", t[-1] - t[0])
    print("  prologue:", d[:3].tolist())
    body = d[3:-2]
    n = len(body) // 6 * 6
    b = body[:n].reshape(-1, 6)
    print("  per tap [LOAD issue, lgkm wait, B1, COMPUTE issue, vm wait, B2 (+loop top)]:")
    for r in b[:20]: print("    ", r.tolist(), "sum", int(r.sum()))
    print("  mean over taps:", b.mean(0).round(0).tolist(), "sum", round(float(b.sum(1).mean())), "taps", len(b))
    print("  tail", d[-2:].tolist())
