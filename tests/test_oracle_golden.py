"""Pin the CPU oracle (oracle/*.py) against golden vectors made from the REAL reference.

The vectors in tests/golden/ were produced by tools/make_goldens.py, which imports the
reference from /root/reference in the build container.  Tolerance 1e-5 abs (fp32
reorder noise; the reference's own fp32-vs-fp64 gap is 1.5e-6).
"""
import numpy as np
import torch

from oracle import encoders_ref, sampler_ref, unet_ref
from polyffusion_amd import synth
from polyffusion_amd.arch import UNetConfig
from polyffusion_amd.weights import (synth_chord_encoder_state, synth_texture_encoder_state,
                                     synth_unet_state)

SMALL = UNetConfig(in_channels=2, out_channels=2, channels=32, n_res_blocks=1, attention_levels=(1,),
                   channel_multipliers=(1, 2), n_heads=2, tf_layers=1, d_cond=32)
LIN = (0.00085, 0.012)
TOL = 1e-5


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


def small_model():
    w = unet_ref.to_torch(synth_unet_state(SMALL, 0))
    return lambda x, t, c: unet_ref.unet_forward(w, SMALL, x, t, c)


def test_time_embedding_and_tables(golden):
    g = golden("tables.npz")
    e = unet_ref.time_step_embedding(torch.from_numpy(g["t"]), SMALL.channels)
    assert maxabs(e, g["time_step_embedding"]) <= 1e-6
    alpha, beta, ab = sampler_ref.beta_schedule(1000, *LIN)
    assert np.array_equal(alpha.numpy(), g["alpha"]) and np.array_equal(beta.numpy(), g["beta"])
    assert np.array_equal(ab.numpy(), g["alpha_bar"])
    s = sampler_ref.SDFSamplerRef(None, 1000, *LIN, noise_fn=None)
    for k in ("sqrt_alpha_bar", "sqrt_1m_alpha_bar", "sqrt_recip_alpha_bar", "sqrt_recip_m1_alpha_bar",
              "log_var", "mean_x0_coef", "mean_xt_coef"):
        assert np.array_equal(getattr(s, k).numpy(), g["sdf_" + k]), k
    for tag, (S, disc, eta) in dict(u50=(50, "uniform", 0.0), q50=(50, "quad", 0.0), u20e1=(20, "uniform", 1.0)).items():
        d = sampler_ref.DDIMSamplerRef(None, 1000, *LIN, n_steps=S, discretize=disc, eta=eta)
        assert np.array_equal(d.time_steps, g[f"ddim_{tag}_time_steps"])
        for k in ("ddim_alpha", "ddim_alpha_sqrt", "ddim_alpha_prev", "ddim_sigma", "ddim_sqrt_one_minus_alpha"):
            assert np.array_equal(getattr(d, k).numpy(), g[f"ddim_{tag}_{k}"]), (tag, k)


def test_small_unet(golden):
    g = golden("unet_small.npz")
    w = unet_ref.to_torch(synth_unet_state(SMALL, 0))
    x, t = torch.from_numpy(g["x"]), torch.from_numpy(g["t"])
    trace = {}
    o1 = unet_ref.unet_forward(w, SMALL, x, t, torch.from_numpy(g["cond1"]), trace=trace)
    assert maxabs(o1, g["out1"]) <= TOL
    # block-level traces: last layer of each traced block
    assert maxabs(trace["input_blocks.1.0"], g["trace.input_blocks.1"]) <= TOL
    assert maxabs(trace["middle_block.2"], g["trace.middle_block"]) <= TOL
    o4 = unet_ref.unet_forward(w, SMALL, x, t, torch.from_numpy(g["cond4"]))
    assert maxabs(o4, g["out4"]) <= TOL


def test_full_unet_chd8bar(golden):
    g = golden("unet_chd8bar_b2.npz")
    cfg = UNetConfig(d_cond=512)
    w = unet_ref.to_torch(synth_unet_state(cfg, 0))
    x = torch.from_numpy(synth.gaussian((2, 2, 128, 128), int(g["x_seed"])))
    c = torch.from_numpy(synth.gaussian((2, 1, 512), int(g["cond_seed"])))
    o = unet_ref.unet_forward(w, cfg, x, torch.from_numpy(g["t"]), c)
    assert 0.05 < float(np.abs(g["out"]).mean()) < 20  # well-scaled output => 1e-3 abs is meaningful
    assert maxabs(o, g["out"]) <= 2e-5


def test_single_steps(golden):
    g = golden("steps.npz")
    x, e, nz = (torch.from_numpy(g[k]) for k in ("x", "e_t", "noise"))
    s = sampler_ref.SDFSamplerRef(lambda *a: e, 1000, *LIN, noise_fn=lambda shape: nz)
    for step in (0, 1, 500, 999):
        xp, x0, _ = s.p_sample(x, None, step)
        assert maxabs(xp, g[f"sdf_xprev_{step}"]) <= 1e-6 * max(1.0, float(np.abs(g[f"sdf_xprev_{step}"]).max()))
        assert maxabs(x0, g[f"sdf_x0_{step}"]) <= 1e-6 * max(1.0, float(np.abs(g[f"sdf_x0_{step}"]).max()))
        assert maxabs(s.q_sample(x, step, nz), g[f"sdf_q_{step}"]) <= 1e-6
    for tag, (S, disc, eta) in dict(u50=(50, "uniform", 0.0), u20e1=(20, "uniform", 1.0)).items():
        d = sampler_ref.DDIMSamplerRef(lambda *a: e, 1000, *LIN, n_steps=S, discretize=disc, eta=eta,
                                       noise_fn=lambda shape: nz)
        for idx in (0, 1, S - 1):
            xp, p0 = d.get_x_prev_and_pred_x0(e, idx, x)
            sc = max(1.0, float(np.abs(g[f"ddim_{tag}_predx0_{idx}"]).max()))
            assert maxabs(xp, g[f"ddim_{tag}_xprev_{idx}"]) <= 1e-6 * sc
            assert maxabs(p0, g[f"ddim_{tag}_predx0_{idx}"]) <= 1e-6 * sc
            assert maxabs(d.q_sample(x, idx, nz), g[f"ddim_{tag}_q_{idx}"]) <= 1e-6
    toy = lambda x_, t_, c_: x_ * c_.mean(dim=(1, 2))[:, None, None, None] + t_[:, None, None, None].float() * 1e-3
    cc, uc, t7 = torch.from_numpy(g["cfg_c"]), -torch.ones(2, 1, 32), torch.tensor([7, 7])
    for sc in (0.0, 1.0, 5.0):
        assert maxabs(sampler_ref.get_eps(toy, x, t7, cc, sc, uc), g[f"cfg_eps_{sc}"]) <= 1e-6


class _Tape:
    def __init__(self, arr):
        self.arr, self.i = arr, 0

    def __call__(self, shape):
        a = torch.from_numpy(self.arr[self.i])
        self.i += 1
        assert tuple(a.shape) == tuple(shape)
        return a


def test_trajectories(golden):
    g = golden("trajectories.npz")
    model = small_model()
    cond, start, orig, mask = (torch.from_numpy(g[k]) for k in ("cond", "start_noise", "orig", "mask"))
    uc = -torch.ones(2, 1, 32)
    z = torch.zeros_like(start)
    tape = _Tape(g["ddpm_gen_tape"])
    s = sampler_ref.SDFSamplerRef(model, 1000, *LIN, noise_fn=tape)
    out = s.paint(s.q_sample(z, 9, start), cond, 9, orig=z, mask=z, orig_noise=start, uncond_scale=1.0, uncond_cond=uc)
    assert tape.i == len(g["ddpm_gen_tape"]) == 18  # two draws per step, none at step 0
    assert maxabs(out, g["ddpm_gen_out"]) <= 1e-4
    tape = _Tape(g["ddpm_inp_tape"])
    s = sampler_ref.SDFSamplerRef(model, 1000, *LIN, noise_fn=tape)
    out = s.paint(s.q_sample(orig, 5, start), cond, 5, orig=orig, mask=mask, orig_noise=start,
                  uncond_scale=3.0, uncond_cond=uc, repaint_n=2)
    assert tape.i == len(g["ddpm_inp_tape"])
    assert maxabs(out, g["ddpm_inp_out"]) <= 1e-4
    d = sampler_ref.DDIMSamplerRef(model, 1000, *LIN, n_steps=10, discretize="uniform", eta=0.0)
    out = d.paint(d.q_sample(orig, 4, start), cond, 4, orig=orig, mask=mask, orig_noise=start,
                  uncond_scale=5.0, uncond_cond=uc)
    assert maxabs(out, g["ddim_out"]) <= 1e-4
    tape = _Tape(g["ddim_eta1_tape"])
    d = sampler_ref.DDIMSamplerRef(model, 1000, *LIN, n_steps=10, discretize="quad", eta=1.0, noise_fn=tape)
    out = d.paint(d.q_sample(z, 9, start), cond, 9, orig=z, mask=z, orig_noise=start, uncond_scale=0.0, uncond_cond=uc)
    assert tape.i == len(g["ddim_eta1_tape"])
    assert maxabs(out, g["ddim_eta1_out"]) <= 1e-4


def test_encoders(golden):
    g = golden("encoders.npz")
    wc = unet_ref.to_torch(synth_chord_encoder_state(0))
    wt = unet_ref.to_torch(synth_texture_encoder_state(0))
    zc = encoders_ref.encode_chord(wc, torch.from_numpy(synth.chords(3, int(g["chord_seed"]))))
    zt = encoders_ref.encode_txt(wt, torch.from_numpy(synth.prmat(3, int(g["prmat_seed"]))))
    assert zc.shape == (3, 1, 512) and zt.shape == (3, 1, 1024)
    assert maxabs(zc, g["z_chord"]) <= TOL and maxabs(zt, g["z_txt"]) <= TOL


def test_autoreg_data():
    a = torch.arange(3 * 4 * 2).view(3, 4, 2).float()
    m = sampler_ref.get_autoreg_data(a, 1)
    assert torch.equal(m[0], torch.cat([a[0, 2:], a[1, :2]]))
    assert torch.equal(m[2], torch.cat([a[2, 2:], a[0, :2]]))


def test_oracle_config1_full_size_vs_reference(golden):
    """BASELINE.json configs[0] at FULL model size: the oracle's sampler loop + UNet restatement against the image the real reference
    produced (tests/golden/config1_full.npz: DDPM, uncond_scale 0, B = 1, 10 reverse steps, seeded noise tape)."""
    import numpy as np
    import torch
    from oracle import sampler_ref, unet_ref
    from polyffusion_amd.arch import UNetConfig
    from polyffusion_amd.weights import synth_unet_state
    g = golden("config1_full.npz")
    rng = np.random.Generator(np.random.PCG64(int(g["seed"])))
    draws = [rng.standard_normal((1, 2, 128, 128)).astype(np.float32) for _ in range(int(g["n_draws"]))]
    cfg = UNetConfig(d_cond=512)
    w = unet_ref.to_torch(synth_unet_state(cfg, 0))
    it = iter(draws[1:])
    s = sampler_ref.SDFSamplerRef(lambda x, t, c: unet_ref.unet_forward(w, cfg, x, t, c), 1000, 0.00085, 0.012,
                                  noise_fn=lambda shape: torch.from_numpy(next(it)).reshape(shape))
    with torch.no_grad():
        out = sampler_ref.predict(s, torch.zeros(1, 1, 512), 512, [1, 2, 128, 128], 9, torch.from_numpy(draws[0]), uncond_scale=0.0)
    assert next(it, None) is None
    assert float((out - torch.from_numpy(g["out"])).abs().max()) <= 2e-5
