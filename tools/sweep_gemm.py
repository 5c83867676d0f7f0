#!/usr/bin/env python
"""Tile sweep of the planes GEMMs (pf_conv_args.force_tile) at a given batch: python tools/sweep_gemm.py B"""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import layer_launch  # noqa: E402
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
SH = [s for s in layer_launch.SHAPES if s[0].startswith("p") and len(s) > 11 and s[11] in (1, 3, 4, 5, 6)] + [
    ("pqkv_256_256_768", 16, 1, 256, 256, 0, 768, 1, 1, 0, 0, 3), ("pff2_256_1024_256", 16, 1, 256, 1024, 0, 256, 1, 1, 0, 0, 4)]
def timeit(L):
    try:
        for _ in range(3): L.run()
    except RuntimeError:
        return None
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): L.run(check=False)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 30
for sh in SH:
    sh = (sh[0], B) + tuple(sh[2:])
    t_auto = timeit(layer_launch.Launch(sh, 1))
    res = []
    for tile in (1, 2, 3):
        if tile == 1 and sh[6] % 128: continue
        L = layer_launch.Launch(sh, 1); L.args.force_tile = tile
        t = timeit(L)
        if t is not None: res.append((t, tile - 1))
    res.sort()
    print(f"{sh[0]:26s} B={B}: auto {t_auto:6.1f} | " + "  ".join(f"t{r[1]}:{r[0]:.1f}" for r in res), flush=True)
