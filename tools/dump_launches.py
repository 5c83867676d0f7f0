#!/usr/bin/env python
"""Per-launch durations of one UNet evaluation at an arbitrary batch (hipEvents around every launch): python tools/dump_launches.py [B] [precision]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyffusion_amd import synth
from polyffusion_amd.inference_sdf import synthetic_model
from polyffusion_amd.params import preset
KIND = ["conv3x3", "gemm", "attention", "gn_stats", "ln_stats", "small"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
m = synthetic_model(preset("sdf_chd8bar"))
u = m.ldm.eps_model
u.set_precision(sys.argv[2] if len(sys.argv) > 2 else "bf16x3")
for o in sys.argv[3:]:          # plan options, NAME=0|1
    u.set_option(o.split("=")[0], bool(int(o.split("=")[1])))
x = torch.from_numpy(synth.gaussian((B, 2, 128, 128), 1)).cuda()
c = m._encode_chord(torch.from_numpy(synth.chords(B, 2)).cuda())
t = torch.full((B,), 500, dtype=torch.long, device="cuda")
for _ in range(3):
    u(x, t, c)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    u(x, t, c)
e1.record(); torch.cuda.synchronize()
print(f"B={B}: {e0.elapsed_time(e1) / 20:.3f} ms per UNet evaluation, {u.n_launches(B)} launches")
u.set_profiling(True)
u(x, t, c); torch.cuda.synchronize()
agg = {}
for i, (kind, ms, fl) in enumerate(u.read_profile()):
    print(f"launch {i:3d} {KIND[kind]:10s} {ms * 1e3:8.1f} us {fl / 1e9:8.2f} GF {fl / max(ms, 1e-9) / 1e9:7.1f} TF/s")
    a = agg.setdefault(KIND[kind], [0, 0.0]); a[0] += 1; a[1] += ms
print({k: (v[0], round(v[1], 3)) for k, v in agg.items()})
