#!/usr/bin/env python
"""Cycle stamps of workgroup 0 of the fused pre-attention launch (preattn_fused_bf3.hip, -DPF_PRE_TRACE).
build:  python tools/trace_pre.py --build     run (GPU box): cp build/exp/libpfhip_pre_trace.so polyffusion_amd/libpfhip.so; python tools/trace_pre.py [B L] [--fold]"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))

if "--build" in sys.argv:
    env = dict(os.environ, PF_EXP_DEFS="-DPF_PRE_TRACE")
    subprocess.check_call([sys.executable, os.path.join(REPO, "tools", "exp_variant.py"), "preattn_fused_bf3.hip", "pre_trace", "namespace pf {", "namespace pf {"], env=env)
    raise SystemExit(0)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from polyffusion_amd import _lib  # noqa: E402
from test_gpu_bf16x3 import pack3  # noqa: E402
from test_gpu_ops import dev, gn_scale_shift, rnd  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("-")]
B, L = (int(args[0]), int(args[1])) if len(args) >= 2 else (16, 1024)
fold = "--fold" in sys.argv
C = 256
lib = _lib.load()
x = dev(rnd((B, L, C), 1))
gg, gb, lg, lb, b_in = (dev(rnd((C,), s)) for s in (2, 3, 4, 5, 6))
p_in, p_qkv = pack3(lib, rnd((C, C), 7, C ** -0.5)), pack3(lib, rnd((3 * C, C), 8, C ** -0.5))
sc, sh = gn_scale_shift(lib, x.view(B, 1, L, C), None, gg, gb, 1e-6)
T = 16
xt = x.view(B, T, L // T, C)
stats = torch.stack([xt.sum(2), (xt * xt).sum(2)], dim=-1).contiguous()
y = torch.empty(B, L, C, device="cuda"); planes = torch.zeros(B * L * 3 * C, device="cuda")
st = _lib.current_stream()
run = lambda: _lib.check(lib.pf_preattn_fused(x.data_ptr(), B, L, sc.data_ptr(), sh.data_ptr(), stats.data_ptr() if fold else None, T if fold else 0,
                                               gg.data_ptr() if fold else None, gb.data_ptr() if fold else None, 1e-6, p_in.data_ptr(), b_in.data_ptr(),
                                               y.data_ptr(), lg.data_ptr(), lb.data_ptr(), 1e-5, p_qkv.data_ptr(), planes.data_ptr(), st))
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record(); torch.cuda.synchronize()
print(f"B={B} L={L} fold={fold}: {e0.elapsed_time(e1) * 50:.1f} us per launch (back to back)")
buf = torch.zeros(4096, dtype=torch.int64, device="cuda")
os.environ["PF_TRACE_PTR"] = hex(buf.data_ptr())
run(); torch.cuda.synchronize()
del os.environ["PF_TRACE_PTR"]
t = buf.cpu().numpy(); t = t[t > 0]
d = np.diff(t)
names = ["ring issue + x loads issued", "GroupNorm finalize (folded)", "GN apply + plane stores",
         "proj_in: first-slot wait + barrier", "proj_in: 16 slots", "proj_in: drain",
         "exchange: bias / gamma / beta, barrier, pass-0 ring issue, acc -> LDS rows, barrier", "exchange: row reads + barrier", "exchange: LayerNorm + planes"]
for ps in range(3):
    names += [f"pass {ps}: first-slot wait + barrier", f"pass {ps}: 16 slots", f"pass {ps}: drain", f"pass {ps}: next ring issue + epilogue"]
print(f"{len(t)} stamps, total {t[-1] - t[0]} cycles")
for n, v in zip(names, d):
    print(f"  {n:55s} {int(v):7d}")
