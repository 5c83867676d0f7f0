#!/usr/bin/env python
"""Cycle-stamp timeline of one workgroup of the bf16x3 attention kernel (needs a PF_TRACE build)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyffusion_amd import _lib
lib = _lib.load()
B, H, L = 16, 4, 1024
FORM = int(sys.argv[1]) if len(sys.argv) > 1 else -1   # pf_attention_bf16x3 form: -1 auto, 0 128-query, 1 256-query workgroups
c = H * 64
planes = (torch.randn(B * L * 3 * c * 2, device="cuda") * 0.5).to(torch.bfloat16)
out = torch.empty(B, L, c, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    lib.pf_attention_bf16x3(planes.data_ptr(), out.data_ptr(), c, None, B, H, L, FORM, st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    lib.pf_attention_bf16x3(planes.data_ptr(), out.data_ptr(), c, None, B, H, L, FORM, st)
e1.record(); torch.cuda.synchronize()
print("kernel us", e0.elapsed_time(e1) * 100)
buf = (C.c_ulonglong * 4096)()
lib.pf_debug_trace_read_attn(buf, 4096)
a = np.array(buf[:], dtype=np.int64); a = a[a > 0]
d = np.diff(a)
print("stamps", len(a), "total", a[-1] - a[0])
body = d[1:]
wide = FORM != 0   # the 256-query form stamps an iteration three times
k = int(os.environ.get("PF_TRACE_STAMPS", "3")) if wide else 5
n = len(body) // k * k
t = body[:n].reshape(-1, k)
print("per iteration [wait + barrier, rescale check + S(t+1) beside the exponentials, PV(t) beside splits / maxima]" if wide else
      "per tile [wait, barrier, S phase (+next tile's glds), softmax, PV + loop]")
for r in t[:6]: print("   ", r.tolist(), int(r.sum()))
print("mean", t.mean(0).round(0).tolist(), round(float(t.sum(1).mean())))
