#!/usr/bin/env python
"""Launch time of the bf16x3 self-attention kernel at the two attention levels of the bench configuration (B = 16, 4 heads of 64)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyffusion_amd import _lib
lib = _lib.load()
B, H = 16, 4
FORM = int(sys.argv[1]) if len(sys.argv) > 1 else -1   # pf_attention_bf16x3 form: -1 auto, 0 128-query, 1 256-query workgroups
c = H * 64
st = torch.cuda.current_stream().cuda_stream
for L in (1024, 256):
    planes = (torch.randn(B * L * 3 * c * 2, device="cuda") * 0.5).to(torch.bfloat16)
    out = torch.empty(B, L, c, device="cuda")
    for _ in range(5):
        lib.pf_attention_bf16x3(planes.data_ptr(), out.data_ptr(), c, None, B, H, L, FORM, st)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            lib.pf_attention_bf16x3(planes.data_ptr(), out.data_ptr(), c, None, B, H, L, FORM, st)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 50)
    flops = 4.0 * B * H * L * L * 64
    print(f"L={L}: {best:.1f} us  {flops / best * 1e-6:.0f} TFLOP/s fp32-equivalent", flush=True)
