"""Round-5 parity additions (-m gpu):

* the BASELINE configs at the LENGTH they run - all 1000 DDPM steps of config 2 (B = 16), all 50 DDIM steps of config 3 (B = 32,
  guidance 5), one config-5 chain (8 songs, 3 sequential 1000-step runs) - in the headline arithmetic (bf16x3) against the
  exact-fp32-MFMA mode on the same noise tape, with the drift curve and the NOTE-level disagreement after the reference's
  threshold (tools/long_parity.py; reference loops: sampler_sdf.py:289-350, sampler_ddim.py:336-362, inference_sdf.py:227-283);
* the bf16x3 margin on badly-scaled weights: per-layer log-uniform [0.25, 4] weight scales, normalisation gains up to 10, 1 % of
  the input at 50 sigma - full-size UNet eps against the CPU oracle, error relative to the output scale.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import unet_ref  # noqa: E402
from polyffusion_amd import _lib  # noqa: E402
from polyffusion_amd.arch import UNetConfig  # noqa: E402
from polyffusion_amd.inference_sdf import build_unet, synthetic_model  # noqa: E402
from polyffusion_amd.params import preset  # noqa: E402
from polyffusion_amd.weights import synth_unet_state  # noqa: E402
from tools import long_parity  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


@pytest.fixture(scope="module")
def chd8bar():
    _lib.require_gpu()
    m = synthetic_model(preset("sdf_chd8bar"))
    m.ldm.eps_model.set_precision("bf16x3")
    return m


def _record(name, res):
    print(name, json.dumps(res))
    try:
        os.makedirs(OUT, exist_ok=True)
        with open(os.path.join(OUT, f"long_parity_{name}.json"), "w") as f:
            json.dump(res, f, indent=1)
    except OSError:
        pass


# Bounds.  The two modes differ by ~5e-5 per denoiser evaluation (contract 1e-3).  A reverse loop is contractive in the known /
# low-noise directions and mildly expansive elsewhere; what is asserted is that after the WHOLE loop the images still agree to the
# single-evaluation contract (1e-3 max-abs) and that the music - the thresholded piano roll - differs in at most a few cells in
# a million (cells that sit within the drift of the 0.5 threshold).
FINAL_MAX_ABS = 1e-3
NOTE_PPM = 50            # differing onset / sustain bits per million cells


def _check(res):
    assert res["finite"]
    assert res["max_abs"] < FINAL_MAX_ABS, res
    n = res["notes"]
    assert (n["onset_bits_differ"] + n["sustain_bits_differ"]) * 1e6 <= NOTE_PPM * 2 * n["cells"], n
    assert n["notes_in_one_only"] + n["notes_duration_differs"] <= max(2, n["notes_f32"] // 1000), n


def test_config2_all_1000_ddpm_steps_f32_vs_bf16x3(chd8bar):
    res = long_parity.config2(chd8bar)
    _record("config2", res)
    assert len(res["curve_max_abs"]) == 10
    _check(res)


def test_config3_all_50_ddim_steps_cfg5_f32_vs_bf16x3(chd8bar):
    res = long_parity.config3(chd8bar)
    _record("config3", res)
    assert len(res["curve_max_abs"]) == 5
    _check(res)


def test_config5_chain_3_runs_of_1000_steps_f32_vs_bf16x3(chd8bar):
    res = long_parity.config5(chd8bar)
    _record("config5", res)
    assert len(res["curve_max_abs"]) == 30
    _check(res)


# ---------------------------------------------------------------------------------------------- badly-scaled weights
def stressed_state(cfg: UNetConfig, seed: int = 0):
    """synth_unet_state with every conv / linear weight multiplied by its own log-uniform [0.25, 4] factor, every normalisation gain
    drawn log-uniform in [0.1, 10] with a random sign, every bias x 5: nothing like the N(0, 1/fan_in) the other tests use."""
    st = synth_unet_state(cfg, seed)
    rng = np.random.Generator(np.random.PCG64(seed + 4711))
    for k in st:
        v = st[k]
        if v.ndim > 1:
            st[k] = (v * np.float32(np.exp(rng.uniform(np.log(0.25), np.log(4.0))))).astype(np.float32)
        elif k.endswith(".weight"):
            g = np.exp(rng.uniform(np.log(0.1), np.log(10.0), size=v.shape)) * rng.choice([-1.0, 1.0], size=v.shape)
            st[k] = g.astype(np.float32)
        else:
            st[k] = (v * 5).astype(np.float32)
    return st


@pytest.mark.parametrize("seed", [0, 1])
def test_bf16x3_margin_on_badly_scaled_weights_and_outliers(seed):
    """Full-size sdf_chd8bar UNet, B = 2: eps in both modes against the CPU oracle on stressed weights and an input with 1 % of
    its cells at +-50 sigma.  The contract (UNet max-abs-diff < 1e-3 for unit-scale outputs) is read RELATIVE to the output scale
    here - the stressed net's eps is not unit-scale - and the f32 mode's own distance from the oracle is printed beside it."""
    p = preset("sdf_chd8bar")
    cfg = UNetConfig(d_cond=p.d_cond)
    st = stressed_state(cfg, seed)
    u = build_unet(p)
    u.load_state_dict(st)
    rng = np.random.Generator(np.random.PCG64(99 + seed))
    x = rng.standard_normal((2, 2, 128, 128)).astype(np.float32)
    hot = rng.random(x.shape) < 0.01
    x[hot] = 50.0 * np.sign(x[hot])
    c = (3.0 * rng.standard_normal((2, 1, p.d_cond))).astype(np.float32)
    t = torch.tensor([987, 12])
    x, c = torch.from_numpy(x), torch.from_numpy(c)
    w = unet_ref.to_torch(st)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    with torch.no_grad():
        ref = unet_ref.unet_forward(w, cfg, x, t, c)
    scale = ref.abs().max().item()
    rms = ref.pow(2).mean().sqrt().item()
    errs = {}
    for mode in ("f32", "bf16x3"):
        u.set_precision(mode)
        got = u(x.cuda(), t.cuda(), c.cuda()).cpu()
        assert torch.isfinite(got).all()
        errs[mode] = (got - ref).abs().max().item()
    print(f"stress seed {seed}: |eps|max {scale:.3g} rms {rms:.3g}; max-abs-diff vs oracle f32 {errs['f32']:.3g} ({errs['f32'] / scale:.2e} rel), "
          f"bf16x3 {errs['bf16x3']:.3g} ({errs['bf16x3'] / scale:.2e} rel)")
    assert errs["bf16x3"] / scale < 1e-3
    assert errs["f32"] / scale < 1e-3
