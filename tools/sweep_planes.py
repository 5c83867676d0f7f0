#!/usr/bin/env python
"""Tile sweep of the planes GEMMs (gemm_planes_bf3.hip) on the transformer-linear shapes at B = 16: the library's own tile choice against
pf_conv_args.force_tile = 1 (128 x 128), 2 (128 x 64), 3 (64 x 64); microseconds per launch, median of 5 rounds x 30 launches, alternating."""
import os
import statistics
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import layer_launch  # noqa: E402

EXTRA = [
    ("pqkv_256_256_768", 16, 1, 256, 256, 0, 768, 1, 1, 0, 0, 3),
    ("pff1_256_256_2048", 16, 1, 256, 256, 0, 2048, 1, 1, 0, 0, 2),
    ("pff2_256_1024_256", 16, 1, 256, 1024, 0, 256, 1, 1, 0, 0, 1),
]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    for shape in layer_launch.SHAPES + EXTRA:
        if layer_launch.shape_mode(shape) not in (1, 2, 3, 4, 5, 6):
            continue
        shape = (shape[0], B) + tuple(shape[2:])
        Ls = []
        for ft in (0, 1, 2, 3):
            L = layer_launch.Launch(shape, 1)
            L.args.force_tile = ft
            if L.lib.pf_conv2d(__import__("ctypes").byref(L.args), torch.cuda.current_stream().cuda_stream) != 0:
                Ls.append(None); continue
            Ls.append(L)
        torch.cuda.synchronize()
        t = [[] for _ in Ls]
        for _ in range(5):
            for k, L in enumerate(Ls):
                if L is None:
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(30):
                    L.run(check=False)
                e1.record(); torch.cuda.synchronize()
                t[k].append(e0.elapsed_time(e1) * 1e3 / 30)
        m = [statistics.median(x) if x else float("nan") for x in t]
        print(f"{shape[0]:24s} B={B:2d}  auto {m[0]:6.1f} us   128x128 {m[1]:6.1f}   128x64 {m[2]:6.1f}   64x64 {m[3]:6.1f}", flush=True)


if __name__ == "__main__":
    main()
