#!/usr/bin/env python
"""Tile / split-K sweep of the 3x3 layer shapes at a given batch (pf_conv_args.force_tile / force_ksplit / no_pp): which form is fastest
where the library's pickers were tuned at B = 16.   usage: python tools/sweep_conv.py B [filter]
Prints, per shape, the library's own choice and every forced combination (us per launch, 20 back-to-back launches; split forms
include their reduce launch)."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import ctypes as C  # noqa: E402

import layer_launch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
flt = sys.argv[2] if len(sys.argv) > 2 else ""
SHAPES = [s for s in layer_launch.SHAPES if s[0].startswith(("r1", "r3", "r6", "rs"))] + [
    ("rs16_256_256", 16, 16, 16, 256, 0, 256, 3, 1, 0, 1, 7, 256, 256),
    ("r64_64_128", 16, 64, 64, 64, 0, 128, 3, 1, 0, 1),
    ("r32_128_256", 16, 32, 32, 128, 0, 256, 3, 1, 0, 1),
    ("r64_128+64_128", 16, 64, 64, 128, 64, 128, 3, 1, 0, 1),
    ("r32_256+128_256", 16, 32, 32, 256, 128, 256, 3, 1, 0, 1),
]


def timeit(L):
    try:
        for _ in range(3):
            L.run()
    except RuntimeError:
        return None
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        L.run(check=False)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 20


for shape in SHAPES:
    if flt and flt not in shape[0]:
        continue
    shape = (shape[0], B) + tuple(shape[2:])
    base = layer_launch.Launch(shape, 1)
    t_auto = timeit(base)
    res = []
    nchunk = (shape[4] + shape[5]) // 32
    for tile in (1, 2, 3):
        if tile == 1 and shape[6] % 128:
            continue
        for ks in (1, 2, 4, 8, 16):
            if nchunk % ks:
                continue
            for nopp in (0, 1):
                if nopp and (tile != 1 or ks != 1):
                    continue
                L = layer_launch.Launch(shape, 1)
                a = L.args
                a.force_tile, a.force_ksplit, a.no_pp = tile, ks, nopp
                wsb = int(L.lib.pf_conv_splitk_ws_bytes(C.byref(a)))
                if wsb:
                    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
                    a.splitk_ws, a.splitk_ws_bytes = ws.data_ptr(), wsb
                    L.keep.append(ws)
                elif ks > 1:
                    continue
                t = timeit(L)
                if t is not None:
                    res.append((t, tile - 1, ks, nopp))
    res.sort()
    best = res[0]
    print(f"{shape[0]:18s} B={B}: auto {t_auto:6.1f} us | best {best[0]:6.1f} (tile {best[1]} ks {best[2]}{' nopp' if best[3] else ''}) "
          f"{(t_auto / best[0] - 1) * 100:+5.1f} % | " + "  ".join(f"t{r[1]}k{r[2]}{'n' if r[3] else ''}:{r[0]:.1f}" for r in res[:8]), flush=True)
