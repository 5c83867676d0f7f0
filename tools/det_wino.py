import sys, os, ctypes as C
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from polyffusion_amd import _lib, synth
from test_gpu_wino import _unet
m = _unet(); m.set_precision("bf16x3")
for B, shared in ((64, True), (64, False), (32, False), (16, False)):
    Bx = B // 2 if shared else B
    x = torch.from_numpy(synth.gaussian((Bx, 2, 128, 128), 3)).cuda()
    c = torch.from_numpy(synth.gaussian((B, 1, 512), 4)).cuda()
    t = torch.full((B,), 500, dtype=torch.long, device="cuda")
    for opt in (None, False, True):
        m.set_option("conv_wino", opt)
        ref = m(x, t, c, shared_x=shared).clone()
        bad = 0
        for _ in range(6):
            o = m(x, t, c, shared_x=shared)
            bad += int(not torch.equal(o.view(torch.int32), ref.view(torch.int32)))
        print(f"B={B} shared={shared} conv_wino={opt}: {bad}/6 runs differ", flush=True)
