#!/usr/bin/env python
"""Cycle stamps of one workgroup of the fused MLP kernel (needs the m_trace variant from tools/exp_variant.py, see below).
build:  python tools/trace_mlp.py --build      run (GPU box): cp build/exp/libpfhip_m_trace.so polyffusion_amd/libpfhip.so; python tools/trace_mlp.py [B L]"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))

if "--build" in sys.argv:
    TR = 'do { if (tr) { *tr++ = __builtin_amdgcn_s_memtime(); } } while (0)'
    subs = [
        '#define SLOT_SYNC() do { SB();', f'#define TRS() {TR}\n#define SLOT_SYNC() do {{ SB(); TRS();',
        '__builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); SB(); } while (0)', '__builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); TRS(); SB(); } while (0)',
        '  // ---- weight stream ---', '  unsigned long long* tr = (blockIdx.x == 0 && threadIdx.x == 0) ? reinterpret_cast<unsigned long long*>(const_cast<float*>(p.mean)) : nullptr;\n  ' + TR + ';\n  // ---- weight stream ---',
        '  // ---- fragment addresses ---', '  ' + TR + ';\n  // ---- fragment addresses ---',
        '    constexpr int RPW = BM / 4;   // rows per wave', '    ' + TR + ';\n    constexpr int RPW = BM / 4;   // rows per wave',
        '    float mu[RPW];', '    ' + TR + ';\n    float mu[RPW];',
        '    for (int i = 0; i < RPW; ++i) {\n      const int row = wave * RPW + i;\n      const float rs', '    for (int i = 0; i < RPW; ++i) {\n      if (i == 0) ' + TR + ';\n      const int row = wave * RPW + i;\n      const float rs',
        '  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");\n  __syncthreads();', '  ' + TR + ';\n  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");\n  __syncthreads();',
        '  MlpX e{gamma, beta, eps,', '  if (const char* tp = getenv("PF_TRACE_PTR")) p.mean = reinterpret_cast<const float*>(strtoull(tp, nullptr, 16));\n  MlpX e{gamma, beta, eps,',
        '  hipLaunchKernelGGL(kern, dim3(batch * (l / 64)), dim3(256), lds, stream, p, e);', '  hipLaunchKernelGGL(kern, dim3(batch * (l / 64)), dim3(256), lds, stream, p, e);\n  if (getenv("PF_TRACE_PTR")) { hipDeviceSynchronize(); const_cast<ConvP&>(p).mean = nullptr; }',
    ]
    subprocess.check_call([sys.executable, os.path.join(REPO, "tools", "exp_variant.py"), "mlp_fused_bf3.hip", "m_trace"] + subs)
    raise SystemExit(0)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from polyffusion_amd import _lib  # noqa: E402
from test_gpu_bf16x3 import pack3  # noqa: E402
from test_gpu_mlp_fused import _weights, C  # noqa: E402
from test_gpu_ops import dev, rnd  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("-")]
B, L = (int(args[0]), int(args[1])) if len(args) >= 2 else (16, 1024)
lib = _lib.load()
w1, b1, w2, b2, gamma, beta, w1i, b1i = _weights(300)
p1, p2, b1d, b2d, gd, bd = pack3(lib, w1i), pack3(lib, w2), dev(b1i), dev(b2), dev(gamma), dev(beta)
x = dev(rnd((B, L, C), 1)); out = torch.empty(B, L, C, device="cuda")
st = _lib.current_stream()
run = lambda: _lib.check(lib.pf_mlp_geglu_fused(x.data_ptr(), B, L, gd.data_ptr(), bd.data_ptr(), 1e-5, p1.data_ptr(), b1d.data_ptr(), p2.data_ptr(),
                                                 b2d.data_ptr(), out.data_ptr(), None, st))
for _ in range(3):
    run()
torch.cuda.synchronize()
buf = torch.zeros(4096, dtype=torch.int64, device="cuda")
os.environ["PF_TRACE_PTR"] = hex(buf.data_ptr())
run(); torch.cuda.synchronize()
del os.environ["PF_TRACE_PTR"]
t = buf.cpu().numpy(); t = t[t > 0]
d = np.diff(t)
print(f"B={B} L={L}: {len(t)} stamps, total {t[-1] - t[0]} cycles")
print("  setup + ring issue + gamma/beta/b1:", d[0], " x loads + first butterfly:", d[1], " mean + second butterfly:", d[2], " normalise + split + store:", d[3],
      " prologue wait + first fragments:", d[4])
body = d[5:]
# stamps: two per slot (before the hand-over wait, after its barrier); groups: P = 8 slots, M(j) = 12 slots x 15, F = 4 slots
pairs = body[:len(body) // 2 * 2].reshape(-1, 2)    # [hand-over, body of the same slot]
print("  P  [barrier wait, sync-to-sync]:", pairs[:8].tolist())
for j in (0, 1, 7, 14):
    m = pairs[8 + 12 * j: 8 + 12 * (j + 1)]
    print(f"  M({j}) ff1:", m[:8].tolist(), " ff2:", m[8:].tolist(), " sum", int(m.sum()))
print("  F:", pairs[8 + 12 * 15:].tolist(), " tail", body[-1] if len(body) % 2 else "")
