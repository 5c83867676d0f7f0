#!/usr/bin/env python
"""Same-box alternating A/B of the fused Winograd F(2x2, 3x3) form against the direct implicit-GEMM form of pf_conv2d on the stride-1 3x3
layer shapes of the 128x128 / 64x64 / 32x32 levels (B = 16 unless given): microseconds per launch (median of the rounds), direct-equivalent
TFLOP/s, and the distance between the two results.   usage: python tools/bench_wino.py [B] [filter]"""
import ctypes as C
import os
import statistics
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from polyffusion_amd import _lib  # noqa: E402

# name, H, W, c0, c1, cout, residual
SHAPES = [
    ("r128_64_64", 128, 128, 64, 0, 64, 1),
    ("r128_64+64_64", 128, 128, 64, 64, 64, 0),
    ("r128_128+64_64", 128, 128, 128, 64, 64, 0),
    ("r64_64_128", 64, 64, 64, 0, 128, 0),
    ("r64_128_128", 64, 64, 128, 0, 128, 1),
    ("r64_128+64_128", 64, 64, 128, 64, 128, 0),
    ("r64_128+128_128", 64, 64, 128, 128, 128, 0),
    ("r64_256+128_128", 64, 64, 256, 128, 128, 0),
    ("r32_128_256", 32, 32, 128, 0, 256, 0),
    ("r32_256_256", 32, 32, 256, 0, 256, 1),
    ("r32_256+256_256", 32, 32, 256, 256, 256, 0),
]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    lib = _lib.load()
    _lib.require_gpu()
    g = torch.Generator().manual_seed(0)
    for name, H, W, c0, c1, n, has_res in SHAPES:
        if flt and flt not in name:
            continue
        cin = c0 + c1
        x0 = torch.randn(B, H, W, c0, generator=g).cuda()
        x1 = torch.randn(B, H, W, c1, generator=g).cuda() if c1 else None
        w = torch.randn(n, cin, 3, 3, generator=g) * (1.0 / (cin * 9)) ** 0.5
        wd = torch.zeros(lib.pf_packed_gemm_weight_floats(n, cin, 9), dtype=torch.float32)
        _lib.check(lib.pf_pack_gemm_weight_bf16x3(w.data_ptr(), n, cin, 9, wd.data_ptr()))
        ww = torch.zeros(lib.pf_wino_weight_bytes(n, cin), dtype=torch.uint8)
        _lib.check(lib.pf_pack_wino_weight_bf16x3(w.data_ptr(), n, cin, ww.data_ptr()))
        wd, ww = wd.cuda(), ww.cuda()
        sc = (1 + 0.1 * torch.randn(B, cin, generator=g)).cuda(); sh = (0.1 * torch.randn(B, cin, generator=g)).cuda()
        bias = torch.randn(n, generator=g).cuda(); sb = torch.randn(B, n, generator=g).cuda()
        res = torch.randn(B, H, W, n, generator=g).cuda()
        outs, args = [], []
        for wino in (0, 1):
            a = _lib.ConvArgs()
            a.x0, a.c0, a.x1, a.c1 = x0.data_ptr(), c0, (x1.data_ptr() if c1 else 0), c1
            a.batch, a.hin, a.win, a.ks, a.stride, a.ups = B, H, W, 3, 1, 0
            a.w, a.n, a.prologue, a.sc, a.sh = wd.data_ptr(), n, 1, sc.data_ptr(), sh.data_ptr()
            a.bias, a.sbias, a.ld_sbias = bias.data_ptr(), sb.data_ptr(), n
            if has_res:
                a.res, a.ld_res = res.data_ptr(), n
            out = torch.empty(B, H, W, n, device="cuda")
            a.w_wino, a.wino = ww.data_ptr(), wino
            nt = lib.pf_conv_stats_tiles(C.byref(a))
            st = torch.empty(B, nt, n, 2, device="cuda")
            a.out, a.ld_out, a.precision, a.stats_out = out.data_ptr(), n, 1, st.data_ptr()
            wsb = int(lib.pf_conv_splitk_ws_bytes(C.byref(a)))
            if wsb:
                ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
                a.splitk_ws, a.splitk_ws_bytes = ws.data_ptr(), wsb
                outs.append(ws)
            outs += [out, st]
            args.append((a, out))
        stream = torch.cuda.current_stream().cuda_stream
        times = [[], []]
        for _ in range(2):
            for a, _o in args:
                _lib.check(lib.pf_conv2d(C.byref(a), stream))
        torch.cuda.synchronize()
        for rnd in range(7):
            for k, (a, _o) in enumerate(args):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    lib.pf_conv2d(C.byref(a), stream)
                e1.record(); torch.cuda.synchronize()
                times[k].append(e0.elapsed_time(e1) * 1e3 / 20)
        d, wv = statistics.median(times[0]), statistics.median(times[1])
        gf = 2.0 * B * H * W * n * cin * 9 / 1e9
        diff = (args[0][1] - args[1][1]).abs().max().item()
        hbm = (B * H * W * (cin + n * (2 if has_res else 1)) * 4) / 1e6
        print(f"{name:18s} B={B:2d} direct {d:7.1f} us ({gf / d * 1e3:6.1f} TF/s)   wino {wv:7.1f} us ({gf / wv * 1e3:6.1f} TF/s direct-equivalent)   "
              f"x{d / wv:4.2f}   |diff| {diff:.1e}   HBM floor {hbm / 5.0:5.1f} us @5TB/s", flush=True)


if __name__ == "__main__":
    main()
