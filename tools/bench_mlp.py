#!/usr/bin/env python
"""Time pf_mlp_geglu_fused against the three launches it replaces (B=16, L=1024 and L=256)."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from polyffusion_amd import _lib  # noqa: E402
from test_gpu_bf16x3 import pack3  # noqa: E402
from test_gpu_mlp_fused import _weights, C, HID  # noqa: E402
from test_gpu_ops import dev, rnd, run_conv  # noqa: E402
import ctypes as CT  # noqa: E402


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    lib = _lib.load()
    w1, b1, w2, b2, gamma, beta, w1i, b1i = _weights(300)
    p1, p2, b1d, b2d, gd, bd = pack3(lib, w1i), pack3(lib, w2), dev(b1i), dev(b2), dev(gamma), dev(beta)
    st = _lib.current_stream()
    for B, L in ((16, 1024), (16, 256), (8, 1024), (32, 1024)):
        x = dev(rnd((B, L, C), 1))
        out = torch.empty(B, L, C, device="cuda")
        fused = lambda: lib.pf_mlp_geglu_fused(x.data_ptr(), B, L, gd.data_ptr(), bd.data_ptr(), 1e-5, p1.data_ptr(), b1d.data_ptr(),
                                               p2.data_ptr(), b2d.data_ptr(), out.data_ptr(), None, st)
        lnp = torch.zeros(B * L * C, device="cuda"); gp = torch.zeros(B * L * HID, device="cuda")
        a1 = _lib.ConvArgs(); a2 = _lib.ConvArgs()
        for a, kw in ((a1, dict(x0=lnp, c0=C, batch=B, hin=1, win=L, ks=1, stride=1, ups=0, w=p1, n=2 * HID, bias=b1d, geglu=1, out=gp, ld_out=HID,
                                precision=1, a_planes=1, out_planes=gp)),
                      (a2, dict(x0=gp, c0=HID, batch=B, hin=1, win=L, ks=1, stride=1, ups=0, w=p2, n=C, bias=b2d, res=x, ld_res=C, out=out, ld_out=C,
                                precision=1, a_planes=1))):
            for k, v in kw.items():
                setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)

        def chain():
            lib.pf_ln_planes(x.data_ptr(), B * L, C, 1e-5, gd.data_ptr(), bd.data_ptr(), lnp.data_ptr(), st)
            lib.pf_conv2d(CT.byref(a1), st)
            lib.pf_conv2d(CT.byref(a2), st)
        gf = 2.0 * B * L * (C * 2 * HID + HID * C) / 1e9
        tf, tc = timeit(fused), timeit(chain)
        print(f"B={B:3d} L={L:5d}: fused {tf:7.1f} us ({gf / tf * 1e3:6.1f} TF/s)   ln+ff1+ff2 {tc:7.1f} us ({gf / tc * 1e3:6.1f} TF/s)")


if __name__ == "__main__":
    main()
