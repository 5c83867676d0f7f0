#!/usr/bin/env python
"""tests/golden/config1_full.npz: BASELINE.json configs[0] on the REAL reference - DDPM, unconditional (uncond_scale = 0: the
denoiser sees the all(-1) condition), 8-bar prmat2c image, batch 1, 10 reverse steps on the CPU, at FULL model size (sdf_chd8bar
UNet, 41 M parameters, seeded synthetic weights).  The reference's own ``Experiments.predict`` source (ast-compiled as in
tools/make_goldens_orch.py) drives the imported real ``SDFSampler``; the noise comes from a seeded tape the test regenerates, so the
fixture holds only the seed and the final image (131 KB).  Build container only; needs /root/reference."""
from __future__ import annotations

import os
import sys
import types
from typing import Optional

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from tools.make_goldens import CHD8, OUT, Tape, import_reference, ref_ldm, save  # noqa: E402
from tools.make_goldens_orch import load_orchestration  # noqa: E402

SEED = 9001


@torch.no_grad()
def main():
    R = import_reference()
    ns = dict(torch=torch, Optional=Optional, device="cpu", DiffusionSampler=object, print=lambda *a, **k: None,
              args=types.SimpleNamespace(ddim=False, ddim_steps=3, repaint_n=1))
    load_orchestration(ns)
    ldm = ref_ldm(R, CHD8)
    params = types.SimpleNamespace(out_channels=2, img_h=128, img_w=128, d_cond=512, n_steps=10)   # t_idx = n_steps - 1 = 9: 10 reverse steps
    msdf = R["sampler_sdf"]
    sampler = msdf.SDFSampler(ldm)
    tape = Tape(SEED)
    ns["torch"] = tape
    msdf.torch = tape
    try:
        cond = torch.zeros(1, 1, 512)     # ignored: uncond_scale == 0 evaluates the denoiser on uncond_cond only (sampler/__init__.py:65-66)
        out = ns["Experiments"]("sdf_chd8bar", params, sampler).predict(cond, uncond_scale=0.0)
    finally:
        ns["torch"] = torch
        msdf.torch = torch
    print("config 1: out", tuple(out.shape), "draws", len(tape.draws), "range", float(out.min()), float(out.max()))
    save("config1_full.npz", seed=SEED, n_draws=len(tape.draws), out=out.numpy(), first_draw_sum=float(tape.draws[0].sum()))


if __name__ == "__main__":
    main()
