#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REAL reference (build container only).

The reference's Python sources are imported read-only from /root/reference and never
copied; only input/expected-output arrays are written.  Missing third-party modules
(torchvision, labml, pretty_midi, omegaconf) are replaced by empty stubs because the
import chain touches them but the hot path never calls into them.

Usage:  python tools/make_goldens.py            (needs /root/reference)
"""
from __future__ import annotations

import os
import sys
import tempfile
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/polyffusion"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

from polyffusion_amd.arch import UNetConfig  # noqa: E402
from polyffusion_amd.weights import (  # noqa: E402
    synth_chord_encoder_state,
    synth_texture_encoder_state,
    synth_unet_state,
)
from polyffusion_amd import synth  # noqa: E402

SMALL = UNetConfig(in_channels=2, out_channels=2, channels=32, n_res_blocks=1, attention_levels=(1,),
                   channel_multipliers=(1, 2), n_heads=2, tf_layers=1, d_cond=32)
CHD8 = UNetConfig(d_cond=512)
TXT = UNetConfig(d_cond=1024)
LIN = (0.00085, 0.012)


def import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("make_goldens.py needs the reference mounted at /root/reference")
    os.chdir(tempfile.mkdtemp(prefix="pf_golden_"))  # dirs.py mkdirs ./demo ./result on import

    def stub(name, **kw):
        m = types.ModuleType(name)
        m.__dict__.update(kw)
        sys.modules[name] = m
        return m

    tv = stub("torchvision")
    tv.models = stub("torchvision.models")
    tv.transforms = stub("torchvision.transforms")
    stub("labml", monit=types.SimpleNamespace(iterate=lambda n, it: it, enum=lambda n, it: enumerate(it)))
    stub("pretty_midi")
    stub("omegaconf", OmegaConf=object)
    sys.path.insert(0, REF)
    from stable_diffusion.model.unet import UNetModel
    from stable_diffusion.latent_diffusion import LatentDiffusion
    import sampler_sdf
    import sampler_ddim
    from models.model_sdf import Polyffusion_SDF
    from dl_modules import ChordEncoder, TextureEncoder
    return dict(UNetModel=UNetModel, LatentDiffusion=LatentDiffusion, sampler_sdf=sampler_sdf,
                sampler_ddim=sampler_ddim, Polyffusion_SDF=Polyffusion_SDF,
                ChordEncoder=ChordEncoder, TextureEncoder=TextureEncoder)


def ref_unet(R, cfg: UNetConfig, seed=0):
    m = R["UNetModel"](in_channels=cfg.in_channels, out_channels=cfg.out_channels, channels=cfg.channels,
                       n_res_blocks=cfg.n_res_blocks, attention_levels=list(cfg.attention_levels),
                       channel_multipliers=list(cfg.channel_multipliers), n_heads=cfg.n_heads,
                       tf_layers=cfg.tf_layers, d_cond=cfg.d_cond)
    sd = {k: torch.from_numpy(v) for k, v in synth_unet_state(cfg, seed).items()}
    m.load_state_dict(sd, strict=True)
    return m.eval()


def ref_ldm(R, cfg, seed=0):
    return R["LatentDiffusion"](ref_unet(R, cfg, seed), None, 0.18215, 1000, LIN[0], LIN[1]).eval()


class Tape:
    """Serves torch.randn / randn_like from a pre-drawn numpy tape; everything else is torch."""

    def __init__(self, seed):
        self.rng = np.random.Generator(np.random.PCG64(seed))
        self.draws = []

    def _draw(self, shape):
        a = self.rng.standard_normal(tuple(shape)).astype(np.float32)
        self.draws.append(a)
        return torch.from_numpy(a.copy())

    def randn(self, *shape, **kw):
        if len(shape) == 1 and not isinstance(shape[0], int):
            shape = tuple(shape[0])
        return self._draw(shape)

    def randn_like(self, x, **kw):
        return self._draw(x.shape)

    def __getattr__(self, name):
        return getattr(torch, name)


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"  {name}: {os.path.getsize(path) / 1024:.1f} KiB")


@torch.no_grad()
def main():
    R = import_reference()
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)

    # G1/G5: time embedding + schedule + sampler tables
    ldm_s = ref_ldm(R, SMALL)
    t = torch.tensor([0, 1, 500, 999])
    temb = ldm_s.eps_model.time_step_embedding(t)
    sdf = R["sampler_sdf"].SDFSampler(ldm_s)
    tabs = dict(t=t.numpy(), time_step_embedding=temb.numpy(),
                alpha=ldm_s.alpha.numpy(), beta=ldm_s.beta.numpy(), alpha_bar=ldm_s.alpha_bar.numpy())
    for k in ("sqrt_alpha_bar", "sqrt_1m_alpha_bar", "sqrt_recip_alpha_bar", "sqrt_recip_m1_alpha_bar",
              "log_var", "mean_x0_coef", "mean_xt_coef"):
        tabs["sdf_" + k] = getattr(sdf, k).numpy()
    for tag, (S, disc, eta) in dict(u50=(50, "uniform", 0.0), q50=(50, "quad", 0.0), u20e1=(20, "uniform", 1.0)).items():
        dd = R["sampler_ddim"].DDIMSampler(ldm_s, S, disc, eta)
        tabs[f"ddim_{tag}_time_steps"] = np.asarray(dd.time_steps)
        for k in ("ddim_alpha", "ddim_alpha_sqrt", "ddim_alpha_prev", "ddim_sigma", "ddim_sqrt_one_minus_alpha"):
            tabs[f"ddim_{tag}_{k}"] = getattr(dd, k).numpy()
    save("tables.npz", **tabs)

    # G3: small UNet (n_cond = 1 and n_cond = 4), with a few block traces
    rng = np.random.Generator(np.random.PCG64(11))
    x = torch.from_numpy(rng.standard_normal((3, 2, 32, 32)).astype(np.float32))
    tt = torch.tensor([0, 417, 999])
    c1 = torch.from_numpy(rng.standard_normal((3, 1, 32)).astype(np.float32))
    c4 = torch.from_numpy(rng.standard_normal((3, 4, 32)).astype(np.float32))
    net = ldm_s.eps_model
    trace = {}
    hooks = []
    for nm in ("input_blocks.1", "middle_block", "output_blocks.3"):
        mod = net.get_submodule(nm)
        hooks.append(mod.register_forward_hook(lambda m, i, o, nm=nm: trace.__setitem__(nm, o.detach().numpy().copy())))
    o1 = net(x, tt, c1)
    tr1 = dict(trace)
    for h in hooks:
        h.remove()
    o4 = net(x, tt, c4)
    save("unet_small.npz", x=x.numpy(), t=tt.numpy(), cond1=c1.numpy(), cond4=c4.numpy(), out1=o1.numpy(),
         out4=o4.numpy(), **{"trace." + k: v for k, v in tr1.items()})

    # G4: full-size sdf_chd8bar (B=2) and sdf_txt (B=1); outputs only (weights are regenerated)
    full = ref_unet(R, CHD8)
    xf = synth.gaussian((2, 2, 128, 128), seed=1234)
    tf = torch.tensor([0, 999])
    cf = synth.gaussian((2, 1, 512), seed=77)
    of = full(torch.from_numpy(xf), tf, torch.from_numpy(cf))
    save("unet_chd8bar_b2.npz", t=tf.numpy(), out=of.numpy(), x_seed=1234, cond_seed=77)
    del full
    fullt = ref_unet(R, TXT)
    xt_ = synth.gaussian((1, 2, 128, 128), seed=4321)
    ct_ = synth.gaussian((1, 1, 1024), seed=78)
    ot = fullt(torch.from_numpy(xt_), torch.tensor([500]), torch.from_numpy(ct_))
    save("unet_txt_b1.npz", t=np.array([500]), out=ot.numpy(), x_seed=4321, cond_seed=78)
    del fullt

    # G6: single-step known answers with injected e_t and noise
    rng = np.random.Generator(np.random.PCG64(5))
    xs = torch.from_numpy(rng.standard_normal((2, 2, 16, 16)).astype(np.float32))
    es = torch.from_numpy(rng.standard_normal((2, 2, 16, 16)).astype(np.float32))
    nz = rng.standard_normal((2, 2, 16, 16)).astype(np.float32)
    g6 = dict(x=xs.numpy(), e_t=es.numpy(), noise=nz)
    mod = R["sampler_sdf"]
    for step in (0, 1, 500, 999):
        sdf.get_eps = lambda *a, **k: es
        tape = Tape(0)
        tape._draw = lambda shape: torch.from_numpy(nz.copy())
        mod.torch = tape
        xp, x0, _ = sdf.p_sample(xs, None, None, step)
        mod.torch = torch
        g6[f"sdf_xprev_{step}"], g6[f"sdf_x0_{step}"] = xp.numpy(), x0.numpy()
        g6[f"sdf_q_{step}"] = sdf.q_sample(xs, step, noise=torch.from_numpy(nz)).numpy()
    modd = R["sampler_ddim"]
    for tag, (S, disc, eta) in dict(u50=(50, "uniform", 0.0), u20e1=(20, "uniform", 1.0)).items():
        dd = modd.DDIMSampler(ldm_s, S, disc, eta)
        for idx in (0, 1, S - 1):
            tape = Tape(0)
            tape._draw = lambda shape: torch.from_numpy(nz.copy())
            modd.torch = tape
            xp, p0 = dd.get_x_prev_and_pred_x0(es, idx, xs, temperature=1.0, repeat_noise=False)
            modd.torch = torch
            g6[f"ddim_{tag}_xprev_{idx}"], g6[f"ddim_{tag}_predx0_{idx}"] = xp.numpy(), p0.numpy()
            g6[f"ddim_{tag}_q_{idx}"] = dd.q_sample(xs, idx, noise=torch.from_numpy(nz)).numpy()
    # CFG combine through get_eps with a toy model
    sdf2 = mod.SDFSampler(ldm_s)
    toy = lambda x, t, c: x * c.mean(dim=(1, 2))[:, None, None, None] + t[:, None, None, None].float() * 1e-3
    sdf2.model = toy
    cc = torch.from_numpy(rng.standard_normal((2, 1, 32)).astype(np.float32))
    uc = -torch.ones(2, 1, 32)
    tcfg = torch.tensor([7, 7])
    g6["cfg_c"] = cc.numpy()
    for s in (0.0, 1.0, 5.0):
        g6[f"cfg_eps_{s}"] = sdf2.get_eps(xs, tcfg, cc, uncond_scale=s, uncond_cond=uc).numpy()
    save("steps.npz", **g6)

    # G7: short trajectories on the small UNet with a noise tape
    ldm_s = ref_ldm(R, SMALL)
    rng = np.random.Generator(np.random.PCG64(21))
    B = 2
    shape = (B, 2, 16, 16)
    cond = torch.from_numpy(rng.standard_normal((B, 1, 32)).astype(np.float32))
    uc = -torch.ones(B, 1, 32)
    start = torch.from_numpy(rng.standard_normal(shape).astype(np.float32))
    orig = torch.from_numpy((rng.random(shape) < 0.1).astype(np.float32))
    mask = torch.zeros(shape)
    mask[:, :, :8] = 1.0
    g7 = dict(cond=cond.numpy(), start_noise=start.numpy(), orig=orig.numpy(), mask=mask.numpy())
    # (a) DDPM generate path: orig = mask = 0, 10 steps, scale 1
    sd = mod.SDFSampler(ldm_s)
    tape = Tape(100)
    mod.torch = tape
    z = torch.zeros(shape)
    xt = sd.q_sample(z, 9, start)
    ga = sd.paint(xt, cond, 9, orig=z, mask=z, orig_noise=start, uncond_scale=1.0, uncond_cond=uc)
    g7["ddpm_gen_out"], g7["ddpm_gen_tape"] = ga.numpy(), np.stack(tape.draws)
    # (b) DDPM inpaint with CFG 3.0 and repaint_n = 2, 6 steps
    tape = Tape(101)
    mod.torch = tape
    xt = sd.q_sample(orig, 5, start)
    gb = sd.paint(xt, cond, 5, orig=orig, mask=mask, orig_noise=start, uncond_scale=3.0, uncond_cond=uc, repaint_n=2)
    g7["ddpm_inp_out"], g7["ddpm_inp_tape"] = gb.numpy(), np.stack(tape.draws)
    mod.torch = torch
    # (c) DDIM 5 of 10 steps eta 0 with CFG 5 and inpainting mask
    dd = modd.DDIMSampler(ldm_s, 10, "uniform", 0.0)
    xt = dd.q_sample(orig, 4, start)
    gc = dd.paint(xt, cond, 4, orig=orig, mask=mask, orig_noise=start, uncond_scale=5.0, uncond_cond=uc)
    g7["ddim_out"] = gc.numpy()
    # (d) DDIM eta = 1 (draws noise), scale 0 (unconditional)
    dd1 = modd.DDIMSampler(ldm_s, 10, "quad", 1.0)
    tape = Tape(102)
    modd.torch = tape
    xt = dd1.q_sample(z, 9, start)
    gd = dd1.paint(xt, cond, 9, orig=z, mask=z, orig_noise=start, uncond_scale=0.0, uncond_cond=uc)
    modd.torch = torch
    g7["ddim_eta1_out"], g7["ddim_eta1_tape"] = gd.numpy(), np.stack(tape.draws)
    save("trajectories.npz", **g7)

    # G9: encoders + conditioning wrapper
    ce = R["ChordEncoder"](36, 512, 512)
    ce.load_state_dict({k: torch.from_numpy(v) for k, v in synth_chord_encoder_state(0).items()})
    te = R["TextureEncoder"](256, 1024, 256, 10)
    te.load_state_dict({k: torch.from_numpy(v) for k, v in synth_texture_encoder_state(0).items()})
    pm = R["Polyffusion_SDF"](ldm_s, "chord+txt", chord_enc=ce, txt_enc=te).eval()
    chord = synth.chords(3, seed=3)
    prmat = synth.prmat(3, seed=4)
    zc = pm._encode_chord(torch.from_numpy(chord))
    zt = pm._encode_txt(torch.from_numpy(prmat))
    save("encoders.npz", chord_seed=3, prmat_seed=4, z_chord=zc.numpy(), z_txt=zt.numpy())
    print("done")


if __name__ == "__main__":
    main()
