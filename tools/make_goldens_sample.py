#!/usr/bin/env python
"""Golden vectors for the sampler entry points the round-1 fixtures did not reach (row a19 and the
``repeat_noise`` / ``temperature`` / ``cond_concat`` arguments), recorded from the REAL reference samplers
(``sampler_sdf.py:194-255``, ``sampler_ddim.py:104-166``, ``p_sample`` :80-171 / :168-231) on the small UNet with a
noise tape.  Build container only (needs /root/reference); writes ``tests/golden/sample.npz`` (arrays only).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from tools.make_goldens import SMALL, Tape, import_reference, ref_ldm, save  # noqa: E402
from polyffusion_amd.arch import UNetConfig  # noqa: E402

# the cond_concat variant feeds cat([x, cond_concat], 1) to the denoiser: 2 image + 1 concat channel in, 2 out
SMALL_CC = UNetConfig(in_channels=3, out_channels=2, channels=32, n_res_blocks=1, attention_levels=(1,),
                      channel_multipliers=(1, 2), n_heads=2, tf_layers=1, d_cond=32)


@torch.no_grad()
def main():
    R = import_reference()
    msdf, mddim = R["sampler_sdf"], R["sampler_ddim"]
    rng = np.random.Generator(np.random.PCG64(77))
    B, shape = 2, [2, 2, 16, 16]
    cond = torch.from_numpy(rng.standard_normal((B, 1, 32)).astype(np.float32))
    uc = -torch.ones(B, 1, 32)
    x_last = torch.from_numpy(rng.standard_normal(shape).astype(np.float32))
    g = dict(cond=cond.numpy(), x_last=x_last.numpy())
    ldm = ref_ldm(R, SMALL)

    def taped(mod, seed, fn):
        tape = Tape(seed)
        mod.torch = tape
        try:
            out = fn()
        finally:
            mod.torch = torch
        return out.numpy(), tape.draws

    def put(tag, out, draws):
        g[f"{tag}_out"] = out
        g[f"{tag}_n_draws"] = len(draws)
        for i, d in enumerate(draws):   # draws have different shapes with repeat_noise (first: x_T, then [1,C,H,W])
            g[f"{tag}_draw{i}"] = d

    sd = msdf.SDFSampler(ldm)
    # (a) DDPM sample(): x_T drawn inside, last 5 steps, CFG 2, shared noise across the batch, temperature 0.7
    put("sdf_a", *taped(msdf, 400, lambda: sd.sample(shape, cond, repeat_noise=True, temperature=0.7, uncond_scale=2.0,
                                                     uncond_cond=uc, t_start=995)))
    # (b) DDPM sample() from a given x_last, plain
    put("sdf_b", *taped(msdf, 401, lambda: sd.sample(shape, cond, x_last=x_last.clone(), t_start=996)))
    # (c) DDIM sample(), eta 1 (draws), shared noise, temperature 1.3, last 4 of 10 steps
    dd = mddim.DDIMSampler(ldm, 10, "uniform", 1.0)
    put("ddim_c", *taped(mddim, 402, lambda: dd.sample(shape, cond, repeat_noise=True, temperature=1.3, uncond_scale=0.0,
                                                       uncond_cond=uc, t_start=6)))
    # (d) DDIM sample(), eta 0 from x_last, CFG 4
    dd0 = mddim.DDIMSampler(ldm, 10, "quad", 0.0)
    put("ddim_d", *taped(mddim, 403, lambda: dd0.sample(shape, cond, x_last=x_last.clone(), uncond_scale=4.0, uncond_cond=uc,
                                                        t_start=5)))
    # (e) DDIM paint() with orig but orig_noise=None: fresh known-region noise per step
    orig = torch.from_numpy((rng.random(shape) < 0.1).astype(np.float32))
    mask = torch.zeros(shape)
    mask[:, :, 8:] = 1
    g["orig"], g["mask"] = orig.numpy(), mask.numpy()
    dd1 = mddim.DDIMSampler(ldm, 10, "uniform", 0.5)
    put("ddim_e", *taped(mddim, 404, lambda: dd1.paint(x_last.clone(), cond, 3, orig=orig, mask=mask, orig_noise=None,
                                                       uncond_scale=1.0, uncond_cond=uc)))
    # (f)/(g) cond_concat through paint() on a 3-input-channel denoiser
    ldm_cc = ref_ldm(R, SMALL_CC)
    cc = torch.from_numpy(rng.standard_normal((B, 1, 16, 16)).astype(np.float32))
    g["cond_concat"] = cc.numpy()
    sdc = msdf.SDFSampler(ldm_cc)
    put("sdf_f", *taped(msdf, 405, lambda: sdc.paint(x_last.clone(), cond, 3, orig=orig, mask=mask, uncond_scale=2.5,
                                                     uncond_cond=uc, cond_concat=cc)))
    put("sdf_f2", *taped(msdf, 406, lambda: sdc.paint(x_last.clone(), cond, 2, uncond_scale=1.0, uncond_cond=uc,
                                                      cond_concat=cc)))   # orig=None branch (:318-328)
    ddc = mddim.DDIMSampler(ldm_cc, 10, "uniform", 0.0)
    put("ddim_g", *taped(mddim, 407, lambda: ddc.paint(x_last.clone(), cond, 3, orig=orig, mask=mask, orig_noise=x_last,
                                                       uncond_scale=2.5, uncond_cond=uc, cond_concat=cc)))
    save("sample.npz", **g)


if __name__ == "__main__":
    main()
