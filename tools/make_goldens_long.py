#!/usr/bin/env python
"""tests/golden/long_ddpm.npz and long_ddim.npz: the FULL-LENGTH loops of BASELINE.json configs 2 and 3 on the REAL reference, one sample
each, at FULL model size (sdf_chd8bar UNet, 41 M parameters, seeded synthetic weights):

* ``long_ddpm``: ``Experiments.predict`` (the reference's own source, ast-compiled as in tools/make_goldens_orch.py) drives the imported
  real ``SDFSampler.paint`` (sampler_sdf.py:289-350) through all **1000** reverse steps, guidance scale 1 (one evaluation per step);
* ``long_ddim``: the same ``predict`` with ``args.ddim``, the real ``DDIMSampler`` (sampler_ddim.py:300-362) at 50 uniform steps,
  eta 0, guidance scale 5 (two evaluations per step).

The noise comes from a seeded PCG64 tape the GPU test regenerates, so the fixtures hold the seed, the condition row and the image at a few
check-points (the state after the reverse step with that number) - 131 KB each.  Build container only; needs /root/reference; ≈10 min of CPU."""
from __future__ import annotations

import os
import sys
import time
import types
from typing import Optional

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from polyffusion_amd import synth  # noqa: E402
from tools.make_goldens import CHD8, Tape, import_reference, ref_ldm, save  # noqa: E402
from tools.make_goldens_orch import load_orchestration  # noqa: E402

SEED_DDPM, SEED_DDIM = 9102, 9103
KEEP_DDPM = (750, 500, 250, 100, 0)          # state after the reverse step with this number
KEEP_DDIM = (501, 241, 1)                    # tau values (uniform 50: tau = 20 i + 1)


def run(R, ns, ldm, which, cond, seed, keep, n_steps_param, scale):
    params = types.SimpleNamespace(out_channels=2, img_h=128, img_w=128, d_cond=512, n_steps=n_steps_param)
    if which == "ddpm":
        mod = R["sampler_sdf"]
        sampler = mod.SDFSampler(ldm)
    else:
        mod = R["sampler_ddim"]
        sampler = mod.DDIMSampler(ldm, n_steps=50, ddim_discretize="uniform", ddim_eta=0.0)
    kept = {}
    inner = sampler.p_sample
    t0 = time.time()

    def p_sample(x, c, t, step, *a, **kw):
        out = inner(x, c, t, step, *a, **kw)
        s = int(step)
        if s in keep:
            kept[s] = out[0].numpy().copy()
        if s % 100 == 0:
            print(f"  {which} step {s}: {time.time() - t0:.0f} s, |x|max {float(out[0].abs().max()):.3f}", flush=True)
        return out

    sampler.p_sample = p_sample
    tape = Tape(seed)
    ns["torch"] = tape
    mod.torch = tape
    ns["args"].ddim = which == "ddim"
    ns["args"].ddim_steps = 50
    try:
        out = ns["Experiments"]("sdf_chd8bar", params, sampler).predict(cond, uncond_scale=scale)
    finally:
        ns["torch"] = torch
        mod.torch = torch
    assert sorted(kept) == sorted(keep), sorted(kept)
    last = min(keep)
    assert np.array_equal(kept[last], out.numpy())       # zero mask: the blend leaves the p_sample result as it is
    return out, kept, tape


@torch.no_grad()
def main():
    torch.set_num_threads(os.cpu_count() or 8)
    R = import_reference()
    ns = dict(torch=torch, Optional=Optional, device="cpu", DiffusionSampler=object, print=lambda *a, **k: None,
              args=types.SimpleNamespace(ddim=False, ddim_steps=50, repaint_n=1))
    load_orchestration(ns)
    ldm = ref_ldm(R, CHD8)
    which = sys.argv[1:] or ["ddim", "ddpm"]
    if "ddim" in which:
        cond = torch.from_numpy(synth.gaussian((1, 1, 512), 77))
        out, kept, tape = run(R, ns, ldm, "ddim", cond, SEED_DDIM, KEEP_DDIM, 1000, 5.0)
        print("ddim: draws", len(tape.draws), "range", float(out.min()), float(out.max()))
        save("long_ddim.npz", seed=SEED_DDIM, n_draws=len(tape.draws), cond=cond.numpy(), keep=np.asarray(KEEP_DDIM),
             first_draw_sum=float(tape.draws[0].sum()), **{f"x_{s}": kept[s] for s in KEEP_DDIM})
    if "ddpm" in which:
        cond = torch.from_numpy(synth.gaussian((1, 1, 512), 78))
        out, kept, tape = run(R, ns, ldm, "ddpm", cond, SEED_DDPM, KEEP_DDPM, 1000, 1.0)
        print("ddpm: draws", len(tape.draws), "range", float(out.min()), float(out.max()))
        save("long_ddpm.npz", seed=SEED_DDPM, n_draws=len(tape.draws), cond=cond.numpy(), keep=np.asarray(KEEP_DDPM),
             first_draw_sum=float(tape.draws[0].sum()), **{f"x_{s}": kept[s] for s in KEEP_DDPM})


if __name__ == "__main__":
    main()
