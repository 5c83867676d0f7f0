#!/usr/bin/env python
"""tests/golden/orchestration_concat.npz: the reference's ``Experiments.predict(cond_concat=...)`` (inference_sdf.py:202-303 with
the cond_concat of :797-803) on a small UNet whose in_channels = out_channels + the blurry image's channels.

Same technique as tools/make_goldens_orch.py (the reference's own ``predict`` source compiled from its file with ``ast``, driven by
the imported real samplers and a noise tape); arrays only.  Build container only; needs /root/reference."""
from __future__ import annotations

import os
import sys
import types
from typing import Optional

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from polyffusion_amd.arch import UNetConfig  # noqa: E402
from tools.make_goldens import OUT, Tape, import_reference, ref_ldm, save  # noqa: E402
from tools.make_goldens_orch import load_orchestration  # noqa: E402

SMALL4 = UNetConfig(in_channels=4, out_channels=2, channels=32, n_res_blocks=1, attention_levels=(1,),
                    channel_multipliers=(1, 2), n_heads=2, tf_layers=1, d_cond=32)


@torch.no_grad()
def main():
    R = import_reference()
    import utils as ref_utils
    g = {}
    rng = np.random.Generator(np.random.PCG64(2025))
    ns = dict(torch=torch, Optional=Optional, device="cpu", DiffusionSampler=object, print=lambda *a, **k: None,
              args=types.SimpleNamespace(ddim=False, ddim_steps=3, repaint_n=1))
    load_orchestration(ns)
    ldm = ref_ldm(R, SMALL4)
    B, H, W = 3, 16, 16
    params = types.SimpleNamespace(out_channels=2, img_h=H, img_w=W, d_cond=SMALL4.d_cond, n_steps=4)
    cond = torch.from_numpy(rng.standard_normal((B, 1, 32)).astype(np.float32))
    cond_mid = torch.from_numpy(rng.standard_normal((B, 1, 32)).astype(np.float32))
    img = torch.from_numpy((rng.random((B, 2, H, W)) < 0.15).astype(np.float32))
    cc = ref_utils.get_blurry_image(img.clone(), 0.25)                       # utils.py:552-567, as inference_sdf.py:802 calls it
    orig = torch.from_numpy((rng.random((B, 2, H, W)) < 0.1).astype(np.float32))
    mask = torch.ones(B, 2, H, W)
    mask[:, :, 4:12, :] = 0
    g.update(cond=cond.numpy(), cond_mid=cond_mid.numpy(), image=img.numpy(), cond_concat=cc.numpy(), orig=orig.numpy(), mask=mask.numpy())
    msdf, mddim = R["sampler_sdf"], R["sampler_ddim"]

    def run(tag, sampler_mod, sampler, seed, c, **kw):
        tape = Tape(seed)
        ns["torch"] = tape
        sampler_mod.torch = tape
        try:
            out = ns["Experiments"]("small", params, sampler).predict(c.clone(), **kw)
        finally:
            ns["torch"] = torch
            sampler_mod.torch = torch
        g[f"{tag}_out"] = out.numpy()
        g[f"{tag}_tape0"] = tape.draws[0]
        g[f"{tag}_tape"] = np.stack(tape.draws[1:]) if len(tape.draws) > 1 else np.zeros((0, 1, 2, H, W), np.float32)
        print(f"  predict[{tag}]: out {tuple(out.shape)}, {len(tape.draws) - 1} step draws")

    sd = msdf.SDFSampler(ldm)
    run("plain", msdf, sd, 400, cond, cond_concat=cc.clone())
    run("inp_cfg", msdf, sd, 401, cond, uncond_scale=2.0, orig=orig.clone(), mask=mask.clone(), cond_concat=cc.clone())
    # autoregressive: the reference hands the whole cond_concat to every batch-1 run, so only a one-segment song works
    run("autoreg1", msdf, sd, 402, cond[:1], cond_mid=cond_mid[:1].clone(), autoreg=True, cond_concat=cc[:1].clone())
    try:
        run("autoreg3", msdf, sd, 403, cond, cond_mid=cond_mid.clone(), autoreg=True, cond_concat=cc.clone())
        g["autoreg3_fails"] = 0
    except RuntimeError as e:
        g["autoreg3_fails"] = 1
        print("  predict[autoreg, B = 3, cond_concat]: the reference raises:", str(e).splitlines()[0][:100])
    ns["args"] = types.SimpleNamespace(ddim=True, ddim_steps=3, repaint_n=1)
    dd = mddim.DDIMSampler(ldm, 10, "uniform", 0.0)
    run("ddim", mddim, dd, 404, cond, uncond_scale=3.0, orig=orig.clone(), mask=mask.clone(), cond_concat=cc.clone())
    os.makedirs(OUT, exist_ok=True)
    save("orchestration_concat.npz", **g)


if __name__ == "__main__":
    main()
