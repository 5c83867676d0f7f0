#!/usr/bin/env python
"""Cycle-stamp timeline of one workgroup of the bf16x3 3x3 kernel (needs a PF_TRACE build of conv_bf16x3.hip)."""
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
    print("  prologue: start->glds issued", d[0], " loadA issue", d[1], " storeA (wait+transform+write)", d[2])
    body = d[3:-2]
    n = len(body) // 4 * 4
    b = body[:n].reshape(-1, 4)
    print("  per tap [loop-top(+writeA), vmcnt wait, barrier, body] first 12 taps:")
    for r in b[:12]: print("    ", r.tolist(), "sum", int(r.sum()))
    print("  mean over taps:", b.mean(0).round(0).tolist(), "sum", round(float(b.sum(1).mean())))
    print("  epilogue cycles", d[-1])
