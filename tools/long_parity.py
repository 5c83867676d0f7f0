"""Full-length trajectories in the two arithmetic modes, on ONE noise tape.

The headline arithmetic (bf16x3: three bf16 MFMAs per product, ~2^-17 relative per product) is narrower than the reference's
fp32.  A single denoiser evaluation is pinned to the reference (5e-5 of the 1e-3 contract); what no 10-step test says is how
that error behaves over the loops the BASELINE configs actually run - 1000 DDPM iterations (`sampler_sdf.py:289-350`), 50 DDIM
iterations with guidance scale 5 (`sampler_ddim.py:336-362`), the autoregressive chain of three 1000-step runs
(`inference_sdf.py:227-283`) - and whether the image after the reference's threshold (`utils.py:240-269`: a cell sounds when
`round(x) > 0`) is the same music.  Here both modes run the SAME loop with the SAME on-device noise (Philox keyed by seed / draw
/ element: the draws do not depend on the mode) through `Experiments.predict`, i.e. through `paint()` itself; the f32 mode is
the yardstick (pinned to the reference at 3.5e-6 per evaluation).

    python tools/long_parity.py [--configs 2,3,5] [--json out.json]

prints one JSON object per config: final max-abs / RMS difference, the growth curve (max-abs difference after every `every`
steps), and the note-level disagreement (cells whose onset / sustain decision differs, and onsets whose duration differs,
through `pf_prmat2c_durations`).  `tests/test_gpu_long_parity.py` asserts the bounds; `bench.py` prints the numbers in
`precision_note`.  Nothing here imports the oracle: it is the product path compared with itself in two modes.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from polyffusion_amd import midi, synth  # noqa: E402
from polyffusion_amd.inference_sdf import Experiments, synthetic_model  # noqa: E402
from polyffusion_amd.params import preset  # noqa: E402
from polyffusion_amd.sampler import DDIMSampler, SDFSampler  # noqa: E402


def note_disagreement(a: torch.Tensor, b: torch.Tensor) -> dict:
    """`a`, `b`: [N, 2, S, 128] images.  The reference's output step (`utils.py:240-269`) turns an image into notes: an onset where
    channel 0 rounds above 0, sustained while channel 1 does.  Counted: cells whose onset bit / sustain bit differs, onsets present
    in one image only, onsets present in both with different durations."""
    da, db = midi.durations(a), midi.durations(b)
    on_a, on_b = da > 0, db > 0
    both = on_a & on_b
    thr = lambda v: v > 0.5            # int(round(v)) > 0 on float32 (round-half-even) is exactly v > 0.5
    return {
        "cells": int(a[:, 0].numel()),
        "onset_bits_differ": int((thr(a[:, 0]) != thr(b[:, 0])).sum()),
        "sustain_bits_differ": int((thr(a[:, 1]) != thr(b[:, 1])).sum()),
        "notes_f32": int(on_a.sum()), "notes_bf16x3": int(on_b.sum()),
        "notes_in_one_only": int((on_a != on_b).sum()),
        "notes_duration_differs": int((both & (da != db)).sum()),
    }


def _stats(ref: torch.Tensor, got: torch.Tensor) -> dict:
    d = (got - ref).double()
    return {"max_abs": float(d.abs().max()), "rms": float(d.pow(2).mean().sqrt()), "ref_rms": float(ref.double().pow(2).mean().sqrt()),
            "ref_max_abs": float(ref.abs().max())}


def run_pair(model, run, every: int):
    """`run(trace)` executes one complete generation in the model's CURRENT arithmetic mode, calling `trace(step, x)` per reverse
    step; returns the final image tensor.  Executed in f32, then in bf16x3; returns the comparison."""
    u = model.ldm.eps_model
    out, curves, secs = {}, {}, {}
    before = u.precision
    try:
        for mode in ("f32", "bf16x3"):
            u.set_precision(mode)
            snaps = []
            count = [0]

            def trace(step, x):
                count[0] += 1
                if count[0] % every == 0:
                    snaps.append((count[0], int(step), x.clone()))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out[mode] = run(trace).clone()
            torch.cuda.synchronize()
            secs[mode] = time.perf_counter() - t0
            curves[mode] = snaps
    finally:
        u.set_precision(before)
    res = _stats(out["f32"], out["bf16x3"])
    res["curve_max_abs"] = [[n, step, float((b - a).abs().max())] for (n, step, a), (_, _, b) in zip(curves["f32"], curves["bf16x3"])]
    res["finite"] = bool(torch.isfinite(out["f32"]).all() and torch.isfinite(out["bf16x3"]).all())
    res["seconds"] = {k: round(v, 2) for k, v in secs.items()}
    return res, out


def _images(gen: torch.Tensor) -> torch.Tensor:
    """[..., 2, S, 128] -> [N, 2, S, 128]"""
    return gen.reshape(-1, *gen.shape[-3:]).contiguous()


def config2(model=None, batch: int = 16, n_steps: int = 1000, seed: int = 1234, every: int = 100):
    """BASELINE configs[1]: sdf_chd8bar, chord-conditioned generation, batch 16, all 1000 DDPM steps through Experiments.predict."""
    p = preset("sdf_chd8bar")
    model = model or synthetic_model(p)
    cond = model._encode_chord(torch.from_numpy(synth.chords(batch, 4242)).cuda())

    def run(trace):
        s = SDFSampler(model.ldm, seed=seed)
        s.on_step = trace
        return Experiments("sdf_chd8bar", p, s, t_idx=n_steps - 1).predict(cond, uncond_scale=1.0)
    res, out = run_pair(model, run, every)
    res["notes"] = note_disagreement(_images(out["f32"]), _images(out["bf16x3"]))
    res["what"] = f"config 2: sdf_chd8bar B={batch}, {n_steps} DDPM steps, Experiments.predict, f32 vs bf16x3 on one Philox tape"
    return res


def config3(model=None, batch: int = 32, ddim_steps: int = 50, scale: float = 5.0, seed: int = 77, every: int = 10):
    """BASELINE configs[2]: DDIM 50 steps, classifier-free guidance 5 (2B sample-evaluations per step), batch 32."""
    p = preset("sdf_chd8bar")
    model = model or synthetic_model(p)
    cond = model._encode_chord(torch.from_numpy(synth.chords(batch, 4242)).cuda())

    def run(trace):
        s = DDIMSampler(model.ldm, ddim_steps, "uniform", 0.0, seed=seed)
        s.on_step = trace
        return Experiments("sdf_chd8bar", p, s).predict(cond, uncond_scale=scale)
    res, out = run_pair(model, run, every)
    res["notes"] = note_disagreement(_images(out["f32"]), _images(out["bf16x3"]))
    res["what"] = f"config 3: sdf_chd8bar B={batch}, DDIM {ddim_steps} steps, uncond_scale {scale}, f32 vs bf16x3"
    return res


def config5(model=None, songs: int = 8, segments: int = 2, n_steps: int = 1000, seed: int = 5, every: int = 100):
    """BASELINE configs[4] per GPU: `songs` songs denoised together, the autoregressive chain of 2*segments-1 runs of n_steps DDPM
    steps each, every run inpainting the half its predecessor produced - errors of one run feed the next one's known region."""
    p = preset("sdf_chd8bar")
    model = model or synthetic_model(p)
    chd = torch.from_numpy(synth.chords(songs * segments, 99)).cuda()
    cond = model._encode_chord(chd).view(songs, segments, 1, p.d_cond)
    cond_mid = cond.flip(1).contiguous()

    def run(trace):
        s = SDFSampler(model.ldm, seed=seed)
        s.on_step = trace
        return Experiments("sdf_chd8bar", p, s, t_idx=n_steps - 1).predict_songs(cond, cond_mid, uncond_scale=1.0)
    res, out = run_pair(model, run, every)
    # [S, 2B, 2, 64, 128] half images: the note extraction walks whole columns of `steps` cells, any height works
    res["notes"] = note_disagreement(_images(out["f32"]), _images(out["bf16x3"]))
    res["what"] = (f"config 5 chain: {songs} songs x {segments} segments = {2 * segments - 1} sequential runs x {n_steps} DDPM steps, "
                   "f32 vs bf16x3")
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="2,3,5")
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    model = synthetic_model(preset("sdf_chd8bar"))
    res = {}
    for c in a.configs.split(","):
        fn = {"2": lambda: config2(model, n_steps=a.steps), "3": lambda: config3(model), "5": lambda: config5(model, n_steps=a.steps)}[c]
        res["config" + c] = fn()
        print(json.dumps({"config" + c: res["config" + c]}), flush=True)
    if a.json:
        with open(a.json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
