#!/usr/bin/env python
"""Pin the orchestration code of ``inference_sdf.py`` (rows a27/a28).

``/root/reference/polyffusion/inference_sdf.py`` cannot be imported here (it needs lightning,
muspy and a real omegaconf at module level), but the functions on the path are pure torch.
This script parses the file with ``ast``, keeps ONLY the definitions of

    dummy_cond_input (:60-72), get_autoreg_data (:121-129), get_mask (:132-193) and
    Experiments.__init__ / Experiments.predict (:196-303)

compiles those nodes from the reference's own source text in memory and runs them - driven by
the imported REAL ``SDFSampler`` / ``DDIMSampler`` on the small UNet with a noise tape - to record
inputs and expected outputs in ``tests/golden/orchestration.npz``.  Nothing of the reference's
source is written anywhere: the fixture holds arrays only.

Usage:  python tools/make_goldens_orch.py        (build container only; needs /root/reference)
"""
from __future__ import annotations

import ast
import os
import sys
import types
from typing import Optional

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from tools.make_goldens import OUT, REF, SMALL, Tape, import_reference, ref_ldm, save  # noqa: E402

KEEP_FUNCS = {"dummy_cond_input", "get_autoreg_data", "get_mask"}
KEEP_METHODS = {"__init__", "predict"}


def load_orchestration(namespace: dict):
    """exec the selected definitions of the reference file inside ``namespace``."""
    path = os.path.join(REF, "inference_sdf.py")
    tree = ast.parse(open(path).read(), filename=path)
    body = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in KEEP_FUNCS:
            body.append(node)
        elif isinstance(node, ast.ClassDef) and node.name == "Experiments":
            node.body = [n for n in node.body if isinstance(n, ast.FunctionDef) and n.name in KEEP_METHODS]
            body.append(node)
    found = {n.name for n in body}
    assert found == KEEP_FUNCS | {"Experiments"}, found
    mod = ast.Module(body=body, type_ignores=[])
    exec(compile(mod, path, "exec"), namespace)
    return namespace


def onsets(rng, B, steps, density, empty_rows=()):
    """prmat2c-like image [B,2,steps,128]: sparse onsets (channel 0) + sustain (channel 1)."""
    img = np.zeros((B, 2, steps, 128), np.float32)
    on = rng.random((B, steps, 128)) < density
    on[:, :, :20] = False   # keep pitch 0 free: the reference uses 0 as its "no onset" sentinel
    on[:, :, 110:] = False
    for b, s in empty_rows:
        on[b, s] = False
    img[:, 0] = on
    img[:, 1] = rng.random((B, steps, 128)) < density
    return img


@torch.no_grad()
def main():
    R = import_reference()
    g = {}
    rng = np.random.Generator(np.random.PCG64(2024))

    ns = dict(torch=torch, Optional=Optional, device="cpu", DiffusionSampler=object, print=lambda *a, **k: None,
              args=types.SimpleNamespace(ddim=False, ddim_steps=3, repaint_n=1))
    load_orchestration(ns)

    # ---- get_autoreg_data (:121-129)
    a1 = torch.arange(3 * 8 * 2, dtype=torch.float32).reshape(3, 8, 2)
    a2 = torch.arange(3 * 2 * 8 * 4, dtype=torch.float32).reshape(3, 2, 8, 4)
    g["autoreg_in_dim1"], g["autoreg_out_dim1"] = a1.numpy(), ns["get_autoreg_data"](a1).numpy()
    g["autoreg_in_dim2"], g["autoreg_out_dim2"] = a2.numpy(), ns["get_autoreg_data"](a2, split_dim=2).numpy()

    # ---- get_mask (:132-193): dense case, sparse case with empty steps (incl. leading ones), the wrap-around case
    cases = {
        "dense": onsets(rng, 2, 16, 0.05),
        "sparse": onsets(rng, 2, 16, 0.01, empty_rows=[(0, 0), (0, 1), (0, 5), (1, 3), (1, 15)]),
        "one": onsets(rng, 1, 8, 0.03, empty_rows=[(0, 2), (0, 3)]),
    }
    for name, img in cases.items():
        g[f"mask_orig_{name}"] = img
        for kind in ("remaining", "below", "above"):
            g[f"mask_{kind}_{name}"] = ns["get_mask"](torch.from_numpy(img.copy()), kind).contiguous().numpy()
    img128 = onsets(rng, 2, 128, 0.02)
    g["mask_orig_bars"] = img128
    g["mask_bars_list"] = np.array([1, 6])
    g["mask_bars"] = ns["get_mask"](torch.from_numpy(img128.copy()), "bars", bar_list=[1, 6]).numpy()

    # ---- dummy_cond_input (:60-72): shapes only
    for ct in ("chord", "txt"):
        p = types.SimpleNamespace(img_h=128, img_w=128, cond_type=ct, chd_n_step=32, chd_input_dim=36)
        outs = ns["dummy_cond_input"](3, p)
        g[f"dummy_{ct}_shapes"] = np.array([list(o.shape) + [0] * (4 - o.dim()) if o is not None else [-1] * 4 for o in outs])

    # ---- Experiments.predict (:202-303) on the small UNet, driven by the real samplers and a noise tape
    ldm = ref_ldm(R, SMALL)
    B, H, W = 3, 16, 16
    params = types.SimpleNamespace(out_channels=2, img_h=H, img_w=W, d_cond=SMALL.d_cond, n_steps=4)  # t_idx = n_steps-1 = 3
    cond = torch.from_numpy(rng.standard_normal((B, 1, 32)).astype(np.float32))
    cond_mid = torch.from_numpy(rng.standard_normal((B, 1, 32)).astype(np.float32))
    orig = torch.from_numpy((rng.random((B, 2, H, W)) < 0.1).astype(np.float32))
    mask = torch.ones(B, 2, H, W)
    mask[:, :, 4:12, :] = 0  # "bars"-like: regenerate the middle
    g.update(pred_cond=cond.numpy(), pred_cond_mid=cond_mid.numpy(), pred_orig=orig.numpy(), pred_mask=mask.numpy())
    msdf, mddim = R["sampler_sdf"], R["sampler_ddim"]

    def run(tag, sampler_mod, sampler, seed, **kw):
        tape = Tape(seed)
        ns["torch"] = tape           # predict's own start-noise draw (:221)
        sampler_mod.torch = tape     # the sampler's per-step draws
        try:
            out = ns["Experiments"]("small", params, sampler).predict(cond.clone(), **kw)
        finally:
            ns["torch"] = torch
            sampler_mod.torch = torch
        g[f"pred_{tag}_out"] = out.numpy()
        g[f"pred_{tag}_tape0"] = tape.draws[0]                       # [B,2,H,W] start noise
        g[f"pred_{tag}_tape"] = (np.stack(tape.draws[1:]) if len(tape.draws) > 1
                                 else np.zeros((0, 1, 2, H, W), np.float32))  # per-step draws
        print(f"  predict[{tag}]: out {tuple(out.shape)}, {len(tape.draws) - 1} step draws")

    sd = msdf.SDFSampler(ldm)
    ns["args"] = types.SimpleNamespace(ddim=False, ddim_steps=3, repaint_n=1)
    run("plain", msdf, sd, 300)                                                                # not autoreg, generate
    run("autoreg", msdf, sd, 301, cond_mid=cond_mid.clone(), autoreg=True)                     # 2B-1 = 5 runs x 4 steps
    run("autoreg_inp", msdf, sd, 302, cond_mid=cond_mid.clone(), autoreg=True, uncond_scale=2.0,
        orig=orig.clone(), mask=mask.clone())                                                  # inpainting + CFG
    ns["args"] = types.SimpleNamespace(ddim=False, ddim_steps=3, repaint_n=2)
    run("autoreg_rp2", msdf, sd, 303, cond_mid=cond_mid.clone(), autoreg=True, orig=orig.clone(), mask=mask.clone())
    ns["args"] = types.SimpleNamespace(ddim=True, ddim_steps=3, repaint_n=1)
    dd = mddim.DDIMSampler(ldm, 10, "uniform", 0.0)   # t_idx = ddim_steps-1 = 2 -> 3 of the 10 DDIM steps
    run("autoreg_ddim", mddim, dd, 304, cond_mid=cond_mid.clone(), autoreg=True, uncond_scale=3.0,
        orig=orig.clone(), mask=mask.clone())

    # ---- get_blurry_image (utils.py:552-567): the cond_concat image of the concat_blurry variant (inference_sdf.py:797-803)
    import utils as ref_utils   # importable with the stubs of import_reference()
    img = torch.from_numpy((rng.random((2, 2, 128, 128)) < 0.05).astype(np.float32))
    g["blurry_in"] = img.numpy()
    for tag, ratio in (("r8", 1 / 8), ("r4", 0.25)):
        g[f"blurry_{tag}"] = ref_utils.get_blurry_image(img.clone(), ratio).numpy()

    os.makedirs(OUT, exist_ok=True)
    save("orchestration.npz", **g)


if __name__ == "__main__":
    main()
