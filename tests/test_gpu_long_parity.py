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
def stressed_state(cfg: UNetConfig, seed: int = 0, w_span: float = 4.0, g_span: float = 10.0, b_mul: float = 5.0):
    """synth_unet_state with every conv / linear weight multiplied by its own log-uniform [1/w_span, w_span] factor, every
    normalisation gain drawn log-uniform in [1/g_span, g_span] with a random sign, every bias x b_mul: nothing like the
    N(0, 1/fan_in) the other tests use."""
    st = synth_unet_state(cfg, seed)
    rng = np.random.Generator(np.random.PCG64(seed + 4711))
    for k in st:
        v = st[k]
        if v.ndim > 1:
            st[k] = (v * np.float32(np.exp(rng.uniform(-np.log(w_span), np.log(w_span))))).astype(np.float32)
        elif k.endswith(".weight"):
            g = np.exp(rng.uniform(-np.log(g_span), np.log(g_span), size=v.shape)) * rng.choice([-1.0, 1.0], size=v.shape)
            st[k] = g.astype(np.float32)
        else:
            st[k] = (v * b_mul).astype(np.float32)
    return st


def stress_inputs(seed: int, d_cond: int):
    rng = np.random.Generator(np.random.PCG64(99 + seed))
    x = rng.standard_normal((2, 2, 128, 128)).astype(np.float32)
    hot = rng.random(x.shape) < 0.01
    x[hot] = 50.0 * np.sign(x[hot])                       # 1 % of the cells at +-50 sigma
    c = (3.0 * rng.standard_normal((2, 1, d_cond))).astype(np.float32)
    return torch.from_numpy(x), torch.tensor([987, 12]), torch.from_numpy(c)


@pytest.mark.parametrize("level,seed", [("harsh", 0), ("harsh", 1), ("mild", 0), ("mild", 1)])
def test_bf16x3_margin_on_badly_scaled_weights_and_outliers(level, seed):
    """Full-size sdf_chd8bar UNet, B = 2, stressed weights (harsh: VERDICT r4's spec - weight scales log-uniform [0.25, 4], gains up
    to 10 either sign, biases x 5; mild: [0.5, 2], gains up to 3, biases x 2) and an input with 1 % of its cells at +-50 sigma.
    The yardstick is the oracle in FLOAT64: a stressed net amplifies every rounding error, fp32's included (tools/stress_diag.py:
    the fp32 oracle itself sits 1e-5 ... 5e-5 relative from the float64 result on the harsh nets, 1e-6 on the ordinary synthetic
    ones; no single class of contraction carries the amplification), so an absolute bound says little without it.  Asserted:
      * the exact-fp32-MFMA mode is as good as PyTorch's fp32 (within 4x of the fp32 oracle's own distance from float64);
      * bf16x3 (unit roundoff 2^-18 against 2^-24) stays within 64x of that distance - observed 15-25x;
      * wherever the net's fp32 conditioning is moderate (fp32 oracle < 2e-5 relative from float64), bf16x3 holds the contract
        read relative to the output scale: max-abs-diff / max|eps| < 1e-3;
      * `pick_precision` (the CLI's --precision auto: both modes on probe inputs, f32 unless they agree to 3e-4 of the output scale)
        chooses f32 wherever bf16x3 misses the contract."""
    from polyffusion_amd.inference_sdf import pick_precision
    p = preset("sdf_chd8bar")
    cfg = UNetConfig(d_cond=p.d_cond)
    st = stressed_state(cfg, seed) if level == "harsh" else stressed_state(cfg, seed, 2.0, 3.0, 2.0)
    u = build_unet(p)
    u.load_state_dict(st)
    x, t, c = stress_inputs(seed, p.d_cond)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    with torch.no_grad():
        truth = unet_ref.unet_forward(unet_ref.to_torch(st, dtype=torch.float64), cfg, x.double(), t, c.double())
        o32 = unet_ref.unet_forward(unet_ref.to_torch(st), cfg, x, t, c).double()
    scale = truth.abs().max().item()
    rel = lambda v: (v.double() - truth).abs().max().item() / scale
    e_o32 = rel(o32)
    errs = {}
    for mode in ("f32", "bf16x3"):
        u.set_precision(mode)
        got = u(x.cuda(), t.cuda(), c.cuda()).cpu()
        assert torch.isfinite(got).all()
        errs[mode] = rel(got)
    choice, probe = pick_precision(u, c.cuda())
    print(f"stress {level} seed {seed}: max|eps| {scale:.3g}; relative to it, vs float64: fp32 oracle {e_o32:.2e}, f32 mode {errs['f32']:.2e}, "
          f"bf16x3 {errs['bf16x3']:.2e} ({errs['bf16x3'] / e_o32:.0f}x the fp32 oracle); pick_precision -> {choice} (probe {probe:.2e})")
    assert errs["f32"] <= 4 * e_o32 + 1e-6
    assert errs["bf16x3"] <= 64 * e_o32 + 1e-5
    if e_o32 < 2e-5:
        assert errs["bf16x3"] < 1e-3
    if errs["bf16x3"] > 1e-3:          # the contract would be violated: the probe (threshold 3e-4 on its own inputs) must have caught it
        assert choice == "f32"
    assert u.precision == choice
