#!/usr/bin/env python
"""Micro-benchmark of pf_conv2d on the UNet's layer shapes (B=16): time per launch and effective TFLOP/s.
usage: python tools/bench_conv.py [f32|bf16x3] [filter]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyffusion_amd import _lib  # noqa: E402

# name, B, H, W, c0, c1, cout, ks, stride, ups, prologue
SHAPES = [
    ("r128_64_64", 16, 128, 128, 64, 0, 64, 3, 1, 0, 1),
    ("r128_128+64_64", 16, 128, 128, 128, 64, 64, 3, 1, 0, 1),
    ("r64_128_128", 16, 64, 64, 128, 0, 128, 3, 1, 0, 1),
    ("r64_256+128_128", 16, 64, 64, 256, 128, 128, 3, 1, 0, 1),
    ("r32_256_256", 16, 32, 32, 256, 0, 256, 3, 1, 0, 1),
    ("r32_256+256_256", 16, 32, 32, 256, 256, 256, 3, 1, 0, 1),
    ("r16_256_256", 16, 16, 16, 256, 0, 256, 3, 1, 0, 1),
    ("r16_256+256_256", 16, 16, 16, 256, 256, 256, 3, 1, 0, 1),
    ("up64_128", 16, 64, 64, 128, 0, 128, 3, 1, 1, 0),
    ("down128_64", 16, 128, 128, 64, 0, 64, 3, 2, 0, 0),
    # second conv of a channel-changing ResBlock: the 1x1 skip projection of concat(x, skip) fused in (mode 7: skip_c0, skip_c1)
    ("rs128_64_64", 16, 128, 128, 64, 0, 64, 3, 1, 0, 1, 7, 128, 64),
    ("rs64_128_128", 16, 64, 64, 128, 0, 128, 3, 1, 0, 1, 7, 256, 128),
    ("rs32_256_256", 16, 32, 32, 256, 0, 256, 3, 1, 0, 1, 7, 256, 256),
    ("g1024_256_256", 16, 1, 1024, 256, 0, 256, 1, 1, 0, 0),
    ("g1024_256_768ln", 16, 1, 1024, 256, 0, 768, 1, 1, 0, 3),
    ("g1024_1024_256", 16, 1, 1024, 1024, 0, 256, 1, 1, 0, 0),
    ("g256_256_256", 16, 1, 256, 256, 0, 256, 1, 1, 0, 0),
    ("skip128_192_64", 16, 1, 16384, 128, 64, 64, 1, 1, 0, 0),
    # A operand as pre-split bf16 hi/lo planes (gemm_planes_bf3.hip)
    ("p1024_256_256", 16, 1, 1024, 256, 0, 256, 1, 1, 0, 0, 1),
    ("p1024_1024_256", 16, 1, 1024, 1024, 0, 256, 1, 1, 0, 0, 1),
    ("p256_256_256", 16, 1, 256, 256, 0, 256, 1, 1, 0, 0, 1),
    ("p256_1024_256", 16, 1, 256, 1024, 0, 256, 1, 1, 0, 0, 1),
    # the GeGLU projection: planes in, GeGLU product out as planes (flag 2) / plain wide GEMM for comparison
    ("pff1_1024_256_2048", 16, 1, 1024, 256, 0, 2048, 1, 1, 0, 0, 2),
    ("pwide_1024_256_2048", 16, 1, 1024, 256, 0, 2048, 1, 1, 0, 0, 1),
    ("pqkv_1024_256_768", 16, 1, 1024, 256, 0, 768, 1, 1, 0, 0, 3),
    ("pff2_1024_1024_256", 16, 1, 1024, 1024, 0, 256, 1, 1, 0, 0, 4),   # planes in, planes out (+ residual)
    ("pff2nores_1024_1024_256", 16, 1, 1024, 1024, 0, 256, 1, 1, 0, 0, 5),   # planes out, no residual
    ("pnores_1024_1024_256", 16, 1, 1024, 1024, 0, 256, 1, 1, 0, 0, 6),      # fp32 out, no residual
]


def main():
    prec = 1 if (len(sys.argv) < 2 or sys.argv[1] == "bf16x3") else 0
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    lib = _lib.load()
    for name, B, H, W, c0, c1, n, ks, stride, ups, pro, *rest in SHAPES:
        planes = bool(rest and rest[0]) and rest[0] != 7
        if planes and not prec:
            continue
        if flt and flt not in name:
            continue
        cin = c0 + c1
        x0 = torch.randn(B, H, W, c0, device="cuda")
        x1 = torch.randn(B, H, W, c1, device="cuda") if c1 else None
        taps = ks * ks
        w = torch.randn(lib.pf_packed_gemm_weight_floats(n, cin, taps), device="cuda") * 0.01
        ho, wo = (H * 2, W * 2) if ups else ((H // 2, W // 2) if stride == 2 else (H, W))
        out = torch.empty(B, ho, wo, n, device="cuda")
        sc = torch.ones(B, cin, device="cuda"); sh = torch.zeros(B, cin, device="cuda")
        mean = torch.zeros(B * H * W, device="cuda"); rstd = torch.ones(B * H * W, device="cuda")
        bias = torch.zeros(n, device="cuda")
        res = torch.randn(B, ho, wo, n, device="cuda")
        a = _lib.ConvArgs()
        a.x0, a.c0, a.x1, a.c1 = x0.data_ptr(), c0, (x1.data_ptr() if c1 else 0), c1
        a.batch, a.hin, a.win, a.ks, a.stride, a.ups = B, H, W, ks, stride, ups
        a.w, a.n, a.prologue = w.data_ptr(), n, pro
        a.sc, a.sh, a.mean, a.rstd = sc.data_ptr(), sh.data_ptr(), mean.data_ptr(), rstd.data_ptr()
        a.bias, a.res, a.ld_res = bias.data_ptr(), res.data_ptr(), n
        a.out, a.ld_out, a.precision = out.data_ptr(), n, prec
        a.a_planes = int(planes)   # same bytes as fp32 [M][K]: the random bits are fine for timing
        mode = rest[0] if rest else 0
        if mode == 2:
            a.geglu, a.ld_out, a.out_planes, a.res = 1, n // 2, out.data_ptr(), 0
        if mode in (4, 5):
            a.out_planes = out.data_ptr()
        if mode in (5, 6):
            a.res = 0
        if mode == 3:
            a.qkv_planes, a.res = out.data_ptr(), 0
        if mode == 7:
            sc0, sc1 = rest[1], rest[2]
            sx0 = torch.randn(B, H, W, sc0, device="cuda"); sx1 = torch.randn(B, H, W, sc1, device="cuda")
            sw = torch.randn(lib.pf_packed_gemm_weight_floats(n, sc0 + sc1, 1), device="cuda") * 0.01
            a.skip_x0, a.skip_c0, a.skip_x1, a.skip_c1, a.skip_w, a.res = sx0.data_ptr(), sc0, sx1.data_ptr(), sc1, sw.data_ptr(), 0
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            _lib.check(lib.pf_conv2d(C.byref(a), st))
        torch.cuda.synchronize()
        if os.environ.get("PF_DET"):   # determinism check: the same launch must reproduce its output bit for bit
            out.zero_()   # (plane outputs cover only part of the buffer)
            _lib.check(lib.pf_conv2d(C.byref(a), st))
            torch.cuda.synchronize()
            ref = out.clone()
            bad = 0
            reps = max(8, int(os.environ.get("PF_DET", "8")))
            for _ in range(reps):
                out.zero_()
                _lib.check(lib.pf_conv2d(C.byref(a), st))
                torch.cuda.synchronize()
                ne = out.view(torch.int32) != ref.view(torch.int32)   # bit patterns: the random packed weights hold NaNs
                if ne.any():
                    bad += 1
                    idx = ne.flatten().nonzero().flatten()
                    print(f"   {int(ne.sum())} elements differ, first flat indices {idx[:6].tolist()} last {idx[-3:].tolist()} "
                          f"(row {int(idx[0]) // n}, col {int(idx[0]) % n}); values {out.flatten()[idx[:3]].tolist()} vs {ref.flatten()[idx[:3]].tolist()}")
            print(f"{name:18s} deterministic: {'yes' if bad == 0 else f'NO ({bad}/{reps} runs differ)'}")
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 20
        e0.record()
        for _ in range(iters):
            lib.pf_conv2d(C.byref(a), st)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        gf = 2.0 * B * ho * wo * n * (cin * taps + (rest[1] + rest[2] if mode == 7 else 0)) / 1e9
        print(f"{name:18s} {us:8.1f} us  {gf:7.2f} GF  {gf / us * 1e3:7.1f} TF/s")


if __name__ == "__main__":
    main()
