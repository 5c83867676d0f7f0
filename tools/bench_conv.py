#!/usr/bin/env python
"""Micro-benchmark of pf_conv2d on the UNet's layer shapes (B=16): time per launch and effective TFLOP/s.
usage: python tools/bench_conv.py [f32|bf16x3] [filter]
The launches are tests/layer_launch.py's (the determinism test owns the list); EXTRA below is for experiments."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import layer_launch  # noqa: E402

EXTRA = [   # same tuple format as layer_launch.SHAPES
    ("rs16_256_256", 16, 16, 16, 256, 0, 256, 3, 1, 0, 1, 7, 256, 256),
    ("r16b8_256_256", 8, 16, 16, 256, 0, 256, 3, 1, 0, 1),
    ("r16b8_256+256_256", 8, 16, 16, 256, 256, 256, 3, 1, 0, 1),
    ("r32b8_256_256", 8, 32, 32, 256, 0, 256, 3, 1, 0, 1),
]


def main():
    prec = 1 if (len(sys.argv) < 2 or sys.argv[1] == "bf16x3") else 0
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for shape in layer_launch.SHAPES + EXTRA:
        if not layer_launch.supported(shape, prec) or (flt and flt not in shape[0]):
            continue
        L = layer_launch.Launch(shape, prec)
        if os.environ.get("PF_DET"):
            reps = max(8, int(os.environ.get("PF_DET", "8")))
            bad, first = L.differing_runs(reps)
            if first:
                print("   " + first)
            print(f"{L.name:18s} deterministic: {'yes' if bad == 0 else f'NO ({bad}/{reps} runs differ)'}")
            continue
        for _ in range(3):
            L.run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 20
        e0.record()
        for _ in range(iters):
            L.run(check=False)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        print(f"{L.name:18s} {us:8.1f} us  {L.gflop:7.2f} GF  {L.gflop / us * 1e3:7.1f} TF/s")


if __name__ == "__main__":
    main()
