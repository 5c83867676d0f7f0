"""Drop-in for the reference ``polyffusion/inference_sdf.py`` on the denoising path.

Kept from the reference: the flag names (``inference_sdf.py:404-507``, typed properly - the
reference leaves ``--ddim_steps``/``--ddim_eta`` as strings), the
``params.yaml`` discovery rule (:518-532), model assembly (:536-557), checkpoint loading for the
legacy ``.pt`` format (:702-716), ``Experiments.predict/generate/inpaint`` incl. the
autoregressive 4-bar inpainting schedule (:202-303), ``get_autoreg_data`` (:121-129),
``dummy_cond_input`` (:60-72) and ``get_mask`` (:132-193).

Not rebuilt (out of the hot-path scope): dataset / MIDI readers, chord
extraction, Polydis comparison.  Conditions therefore come from ``--from_song_npz`` (a quantised song in the reference's data-dictionary format,
segmented by ``polyffusion_amd.datasample``), from ``--cond_npz`` (arrays
``chord`` [B,32,36] and/or ``prmat`` [B,128,128], optional ``prmat2c`` for inpainting) or from the
seeded synthetic generator (``--synthetic``); the result is written as ``.mid`` (polyffusion_amd.midi) and ``.npy`` ([B,2,128,128] piano
roll, or [2B,2,64,128] half-segments for ``--autoreg``).  ``--synthetic_weights`` replaces the
checkpoint by the deterministic weight generator (no trained weights ship with the reference).
"""
from __future__ import annotations

import os
import random
from argparse import ArgumentParser
from datetime import datetime
from typing import Optional

import numpy as np
import torch

from . import _lib, datasample, midi, synth
from .model_sdf import ChordEncoder, Polyffusion_SDF, TextureEncoder
from .params import Params, find_params, load_params, preset
from .sampler import DDIMSampler, DiffusionSampler, SDFSampler
from .unet import LatentDiffusion, UNetModel


def _dev():
    return torch.device("cuda", torch.cuda.current_device())


def dummy_cond_input(length, params):
    h, w = params.img_h, params.img_w
    prmat2c = torch.zeros([length, 2, h, w], device=_dev())
    chord = torch.zeros([length, params.chd_n_step, params.chd_input_dim], device=_dev()) if "chord" in params.cond_type else None
    prmat = torch.zeros([length, h, w], device=_dev())
    pnotree = torch.zeros([length, h, 20, 6], dtype=torch.int64, device=_dev())     # ref:inference_sdf.py:64
    return prmat2c, pnotree, chord, prmat


def get_blurry_image(img: torch.Tensor, ratio: float = 1 / 8) -> torch.Tensor:
    """``utils.py:552-567``: bicubic down by ``ratio``, nearest back up, clipped to [0, 1] - the ``cond_concat`` image of the
    ``concat_blurry`` variant (``inference_sdf.py:797-803``).  Preparation of an input, not part of the step loop: torch ops."""
    small = torch.nn.functional.interpolate(img, scale_factor=ratio, mode="bicubic")
    return torch.nn.functional.interpolate(small, scale_factor=1 / ratio, mode="nearest").clip(0, 1)


def get_autoreg_data(data: torch.Tensor, split_dim: int = 1) -> torch.Tensor:
    """(second half of item i, first half of item i+1): the half-shifted stream for odd runs."""
    steps = data.shape[split_dim]
    half_1, half_2 = data.split(steps // 2, dim=split_dim)
    return torch.cat((half_2, half_1.roll(-1, dims=0)), dim=split_dim)


def get_mask(orig: torch.Tensor, inpaint_type: str, bar_list=None) -> torch.Tensor:
    """Inpainting masks (1 = keep the original).  Vectorised restatement of the reference's row loops."""
    B = orig.shape[0]
    if inpaint_type == "remaining":
        return orig.clone()
    if inpaint_type in ("below", "above"):
        onset = orig[:, 0]
        steps, pitches = onset.shape[1], onset.shape[2]
        flat = onset.reshape(B * steps, pitches)
        if inpaint_type == "below":
            edge = flat.argmax(dim=1)            # lowest onset per step; 0 doubles as "no onset" (reference quirk)
            sentinel = 0
        else:
            edge = (pitches - 1) - flat.flip(1).argmax(dim=1)
            sentinel = pitches - 1               # "no onset" for the top edge
        nz = edge.nonzero()                      # the reference tests `!= 0` for BOTH types (:145,:167)
        if nz.numel() == 0:
            raise IndexError("get_mask: no usable onsets in the original (the reference fails here too)")
        first = int(nz[0, 0])
        edge[:first] = edge[first]
        # sequential rule of the reference: a "no onset" step copies the previous step, step 0 wraps to the last
        if int(edge[0]) == sentinel:
            edge[0] = edge[-1]
        empty = edge == sentinel
        empty[0] = False
        idx = torch.arange(B * steps, device=orig.device)
        last = torch.where(empty, torch.full_like(idx, -1), idx).cummax(0).values
        edge = edge[last]
        cols = torch.arange(pitches, device=orig.device)[None, :]
        m = (cols >= edge[:, None]) if inpaint_type == "below" else (cols <= edge[:, None])
        return m.to(orig.dtype).reshape(B, 1, steps, pitches).expand(-1, 2, -1, -1).contiguous()
    if inpaint_type == "bars":
        if bar_list is None:
            raise ValueError("get_mask('bars') needs bar_list (the reference prompts on stdin)")
        mask = torch.ones_like(orig)
        for bar in bar_list:
            mask[:, :, bar * 16: bar * 16 + 16, :] = 0
        return mask
    raise NotImplementedError(inpaint_type)


class Experiments:
    """``inference_sdf.py:196-400``.  ``t_idx`` replaces the reference's read of the module-global
    ``args`` (ddim_steps-1 or n_steps-1); ``noise`` lets tests inject the start noise."""

    def __init__(self, model_label, params, sampler: DiffusionSampler, t_idx: Optional[int] = None, repaint_n: int = 1):
        self.model_label = model_label
        self.params = params if isinstance(params, Params) else Params(params)
        self.sampler = sampler
        if t_idx is None:
            t_idx = (len(sampler.time_steps) - 1) if isinstance(sampler, DDIMSampler) else self.params.n_steps - 1
        self.t_idx = int(t_idx)
        self.repaint_n = int(repaint_n)

    @torch.no_grad()
    def predict(self, cond: torch.Tensor, cond_mid: Optional[torch.Tensor] = None, uncond_scale=1.0, autoreg=False,
                orig=None, mask=None, cond_concat=None, noise: Optional[torch.Tensor] = None):
        p, dev = self.params, cond.device
        B = cond.shape[0]
        shape = [B, p.out_channels, p.img_h, p.img_w]
        uncond_cond = -torch.ones([B, 1, p.d_cond], device=dev)
        if orig is None or mask is None:
            orig, mask = torch.zeros(shape, device=dev), torch.zeros(shape, device=dev)
        if noise is None:
            noise = self.sampler.randn(shape, dev)
        t_idx = self.t_idx
        kw = dict(uncond_scale=uncond_scale, cond_concat=cond_concat, repaint_n=self.repaint_n)
        if not autoreg:
            xt = self.sampler.q_sample(orig, t_idx, noise)
            return self.sampler.paint(xt, cond, t_idx, orig=orig, mask=mask, orig_noise=noise, uncond_cond=uncond_cond, **kw)
        assert cond_mid is not None
        half = p.img_h // 2
        orig, mask = orig.clone(), mask.clone()  # edited in place below, like the reference's views
        orig_mid, mask_mid, noise_mid = (get_autoreg_data(v, split_dim=2) for v in (orig, mask, noise))
        uc = uncond_cond[0:1]
        gen, new_half = [], None
        for idx in range(B * 2 - 1):  # inpaint a 4-bar half each time
            src = (cond_mid, orig_mid, mask_mid, noise_mid) if idx % 2 == 1 else (cond, orig, mask, noise)
            c_s, o_s, m_s, n_s = (v[idx // 2: idx // 2 + 1] for v in src)
            if idx != 0:
                o_s[:, :, 0:half, :] = new_half
                m_s[:, :, 0:half, :] = 1
            xt = self.sampler.q_sample(o_s, t_idx, n_s)
            x0 = self.sampler.paint(xt, c_s, t_idx, orig=o_s, mask=m_s, orig_noise=n_s, uncond_cond=uc, **kw)
            if idx == 0:
                gen.append(x0[:, :, 0:half, :])
            new_half = x0[:, :, half:, :]
            gen.append(new_half)
        gen = torch.cat(gen, dim=0)
        assert gen.shape[0] == B * 2
        return gen

    @torch.no_grad()
    def predict_songs(self, cond: torch.Tensor, cond_mid: torch.Tensor, uncond_scale=1.0, orig=None, mask=None,
                      cond_concat=None, noise: Optional[torch.Tensor] = None):
        """``predict(autoreg=True)`` (``inference_sdf.py:227-283``) batched ACROSS songs - BASELINE config 5.

        ``cond`` / ``cond_mid`` are ``[S, B, n_cond, d_cond]`` (S songs of B 8-bar segments); ``orig`` / ``mask`` / ``noise``,
        when given, ``[S, B, C, H, W]``.  The 2B-1 runs stay sequential within a song (run r needs the half that run r-1
        produced) but run r of all S songs is one batch of S images, so the denoiser sees batch S instead of batch 1.
        Song s gets exactly what ``predict(cond[s], cond_mid[s], autoreg=True, ...)`` computes when both are fed the same
        noise.  With the on-device generator the draws are keyed by (seed, draw counter, global song index): ranks that
        shard the songs (``sample_offset`` = first song of the rank) reproduce the unsharded run.
        Returns ``[S, 2B, C, H/2, W]``."""
        p, dev = self.params, cond.device
        S, B = cond.shape[0], cond.shape[1]
        shape = [S, B, p.out_channels, p.img_h, p.img_w]
        half = p.img_h // 2
        if cond_concat is not None:
            # the reference hands its WHOLE [B, ...] cond_concat to every batch-1 run (inference_sdf.py:259-270), which torch.cat
            # only accepts for B == 1: the concat_blurry variant can run autoregressively on one-segment songs only
            if cond_concat.dim() != 5 or cond_concat.shape[0] != S or cond_concat.shape[1] != B:
                raise RuntimeError(f"predict_songs: cond_concat must be [S, B, C, H, W] = [{S}, {B}, ...], got {tuple(cond_concat.shape)}")
            if B != 1:
                raise RuntimeError(f"Sizes of tensors must match except in dimension 1. Expected size 1 but got size {B} for tensor number 1 "
                                   "in the list. (cond_concat of a multi-segment song under --autoreg, as in the reference)")
            cond_concat = cond_concat[:, 0].contiguous()
        if orig is None or mask is None:
            orig, mask = torch.zeros(shape, device=dev), torch.zeros(shape, device=dev)
        else:
            orig, mask = orig.clone().float(), mask.clone().float()   # edited in place below
        if noise is None:
            noise = self.sampler.randn(shape, dev)
        # half-shifted streams per song: get_autoreg_data rolls along dim 0, so the segment axis goes there
        mid = lambda v: get_autoreg_data(v.transpose(0, 1), split_dim=3).transpose(0, 1)
        orig_mid, mask_mid, noise_mid = mid(orig), mid(mask), mid(noise)
        uc = -torch.ones([S, 1, p.d_cond], device=dev)
        kw = dict(uncond_scale=uncond_scale, uncond_cond=uc, cond_concat=cond_concat, repaint_n=self.repaint_n)
        gen, new_half = [], None
        for idx in range(B * 2 - 1):
            src = (cond_mid, orig_mid, mask_mid, noise_mid) if idx % 2 == 1 else (cond, orig, mask, noise)
            c_s, o_s, m_s, n_s = (v[:, idx // 2] for v in src)
            if idx != 0:
                o_s[:, :, 0:half, :] = new_half
                m_s[:, :, 0:half, :] = 1
            c_s, o_s, m_s, n_s = (v.contiguous() for v in (c_s, o_s, m_s, n_s))
            xt = self.sampler.q_sample(o_s, self.t_idx, n_s)
            x0 = self.sampler.paint(xt, c_s, self.t_idx, orig=o_s, mask=m_s, orig_noise=n_s, **kw)
            if idx == 0:
                gen.append(x0[:, :, 0:half, :])
            new_half = x0[:, :, half:, :]
            gen.append(new_half)
        return torch.stack(gen, dim=1)

    def _stamp(self, uncond_scale, autoreg, extra=""):
        return (f"{self.model_label}{extra}[scale={uncond_scale}{',autoreg' if autoreg else ''}]"
                f"_{datetime.now().strftime('%y-%m-%d_%H%M%S')}")

    def generate(self, cond, cond_mid=None, uncond_scale=1.0, autoreg=False, no_output=False, cond_concat=None,
                 output_dir="exp", **_ignored):
        gen = self.predict(cond, cond_mid, uncond_scale, autoreg, cond_concat=cond_concat)
        if not no_output:
            os.makedirs(output_dir, exist_ok=True)
            stamp = os.path.join(output_dir, self._stamp(uncond_scale, autoreg))
            np.save(stamp + ".npy", gen.cpu().numpy())
            midi.prmat2c_to_midi_file(gen, stamp + ".mid")            # ref:inference_sdf.py:335-338
        return gen

    def inpaint(self, orig, inpaint_type, cond, cond_mid=None, autoreg=False, orig_noise=None, uncond_scale=1.0,
                bar_list=None, no_output=False, cond_concat=None, output_dir="exp"):
        mask = get_mask(orig, inpaint_type, bar_list).to(orig.device)
        gen = self.predict(cond, cond_mid, uncond_scale, autoreg, orig, mask, cond_concat=cond_concat, noise=orig_noise)
        if not no_output:
            os.makedirs(output_dir, exist_ok=True)
            stamp = os.path.join(output_dir, self._stamp(uncond_scale, autoreg, f"_inp{self.repaint_n}_{inpaint_type}"))
            np.save(stamp + ".npy", gen.cpu().numpy())
            midi.prmat2c_to_midi_file(gen, stamp + ".mid", inp_mask=mask)   # ref:inference_sdf.py:385-389: generated cells on their own track
        return gen


# --------------------------------------------------------------------------------------------- assembly
def build_unet(params, device=None, x3=None) -> UNetModel:
    """`x3="f16"`: the model lives in the fp16-split build of the library (its split mode is "f16x3", UNetModel.__init__)."""
    return UNetModel(in_channels=params.in_channels, out_channels=params.out_channels, channels=params.channels,
                     attention_levels=params.attention_levels, n_res_blocks=params.n_res_blocks,
                     channel_multipliers=params.channel_multipliers, n_heads=params.n_heads, tf_layers=params.tf_layers,
                     d_cond=params.d_cond, img_h=params.img_h, img_w=params.img_w, device=device, x3=x3)


PRECISION_PROBE_TOL = 3e-4
F16_HEADROOM_MIN = 8.0       # f16x3 is kept only if 65504 / max|stored activation| on the probe is at least this


def pick_precision(unet: UNetModel, cond: torch.Tensor, tol: Optional[float] = None, n_steps: int = 1000, seed: int = 0):
    """``--precision auto``: choose the arithmetic mode by MEASURING this network.  bf16x3 (three bf16 MFMAs per product, unit
    roundoff ~2^-18) is ~3x faster than the exact-fp32-MFMA mode and sits 5e-5 from the reference on ordinarily-conditioned
    weights, but every rounding error is amplified by the network's conditioning - fp32's too - and on badly-scaled weights the
    split can exceed the 1e-3 contract (tests/test_gpu_long_parity.py, tools/stress_diag.py).  One evaluation in each mode on three
    probe samples (Gaussian x at t = n_steps-1, n_steps/2 and 0, the first rows of the run's own ``cond``); bf16x3 is kept only if
    ``max|eps_bf16x3 - eps_f32| <= tol * max|eps_f32|`` (default 3e-4: a third of the contract).  Sets the mode on ``unet`` and
    returns ``(mode, measured ratio)``.  Cost: two evaluations of a loop that runs 50-1000."""
    lib = _lib.load()
    tol = PRECISION_PROBE_TOL if tol is None else tol
    n = 3
    x = torch.empty(n, unet.cfg.in_channels, unet.img_h, unet.img_w, dtype=torch.float32, device=cond.device)
    _lib.check(lib.pf_randn(x.data_ptr(), x.numel(), int(seed) + 0x5eed, 0, 0, _lib.current_stream()), "pf_randn")
    t = torch.tensor([n_steps - 1, n_steps // 2, 0], device=cond.device)
    c = cond[torch.arange(n, device=cond.device) % cond.shape[0]].contiguous()
    unet.set_precision("f32")
    # the f32 evaluation doubles as the range measurement: the largest |value| any layer stores (pf_unet_track_absmax).  In the fp16-piece
    # build such a value overflows beyond 65504; measured in f32 nothing overflows while measuring.  A model of the fp16 build must show
    # F16_HEADROOM_MIN x headroom on the probe to be kept in f16x3 - the probe inputs are three samples, a loop sees a thousand.
    unet.track_absmax(True)
    ref = unet(x, t, c).clone()
    absmax = unet.read_absmax()
    unet.track_absmax(False)
    pick_precision.last_absmax = absmax
    unet.set_precision(unet.split_mode)     # "bf16x3", or "f16x3" for a model built in the fp16-split library
    got = unet(x, t, c)
    scale = ref.abs().max().item()
    ratio = float("inf") if not (scale > 0 and bool(torch.isfinite(got).all())) else (got - ref).abs().max().item() / scale
    if unet.split_mode == "f16x3" and not (absmax * F16_HEADROOM_MIN <= 65504.0):
        ratio = float("inf")                # too close to fp16's range (or already beyond it: nan / inf compare false)
    mode = unet.split_mode if ratio <= tol else "f32"
    unet.set_precision(mode)
    return mode, ratio


pick_precision.last_absmax = float("nan")


def build_ldm(params, unet: UNetModel) -> LatentDiffusion:
    return LatentDiffusion(linear_start=params.linear_start, linear_end=params.linear_end, n_steps=params.n_steps,
                           latent_scaling_factor=params.latent_scaling_factor, autoencoder=None, unet_model=unet)


def build_encoders(params, device=None):
    chord_enc = txt_enc = None
    if "chord" in params.cond_type and params.use_enc:
        chord_enc = ChordEncoder(params.chd_input_dim, params.chd_hidden_dim, params.chd_z_dim, device)
    if "txt" in params.cond_type and params.use_enc:
        txt_enc = TextureEncoder(params.txt_emb_size, params.txt_hidden_dim, params.txt_z_dim, params.txt_num_channel, device)
    return chord_enc, txt_enc


def build_pnotree_encoder(params, device=None):
    """The frozen PianoTree encoder of the sdf_pnotree variant (ref:inference_sdf.py:676-680; default sizes, max_simu_note 20)."""
    from .model_sdf import PianoTreeEncoder
    return PianoTreeEncoder(max_simu_note=20, device=device) if params.cond_type == "pnotree" else None


def synthetic_model(params, seed: int = 0, device=None, x3=None) -> Polyffusion_SDF:
    """Whole model with deterministic synthetic weights (same generator as the golden vectors)."""
    from .arch import UNetConfig
    from .weights import synth_chord_encoder_state, synth_texture_encoder_state, synth_unet_state
    unet = build_unet(params, device, x3)
    unet.load_state_dict(synth_unet_state(UNetConfig.from_params(params), seed))
    chord_enc, txt_enc = build_encoders(params, device)
    if chord_enc is not None:
        chord_enc.load_state_dict(synth_chord_encoder_state(seed, params.chd_input_dim, params.chd_hidden_dim, params.chd_z_dim))
    if txt_enc is not None:
        txt_enc.load_state_dict(synth_texture_encoder_state(seed, params.txt_emb_size, params.txt_hidden_dim,
                                                            params.txt_z_dim, params.txt_num_channel))
    pnotree_enc = build_pnotree_encoder(params, device)
    if pnotree_enc is not None:
        from .weights import synth_pianotree_encoder_state
        pnotree_enc.load_state_dict(synth_pianotree_encoder_state(seed))
    return Polyffusion_SDF(build_ldm(params, unet), params.cond_type, params.cond_mode, chord_enc=chord_enc, txt_enc=txt_enc,
                           pnotree_enc=pnotree_enc)


def encode_conditions(model: Polyffusion_SDF, params, chd, prmat, autoreg: bool, pnotree=None):
    cond_mid = None
    if params.cond_type == "pnotree":          # ref:inference_sdf.py:756-760
        assert pnotree is not None
        cond = model._encode_pnotree(pnotree)
        if autoreg:
            cond_mid = model._encode_pnotree(get_autoreg_data(pnotree))
    elif params.cond_type == "chord":
        assert chd is not None
        cond = model._encode_chord(chd)
        if autoreg:
            cond_mid = model._encode_chord(get_autoreg_data(chd))
    elif params.cond_type == "txt":
        assert prmat is not None
        cond = model._encode_txt(prmat)
        if autoreg:
            cond_mid = model._encode_txt(get_autoreg_data(prmat))
    elif params.cond_type == "chord+txt":
        assert chd is not None and prmat is not None
        n = min(chd.shape[0], prmat.shape[0])
        chd, prmat = chd[:n], prmat[:n]
        cond = torch.cat([model._encode_chord(chd), model._encode_txt(prmat)], dim=-1)
        if autoreg:
            cond_mid = torch.cat([model._encode_chord(get_autoreg_data(chd)), model._encode_txt(get_autoreg_data(prmat))], dim=-1)
    else:
        raise NotImplementedError(f"cond_type {params.cond_type!r} is outside the rebuilt path")
    return cond, cond_mid


def make_parser() -> ArgumentParser:
    p = ArgumentParser(description="inference a Polyffusion model (MI355X-native denoising path)")
    p.add_argument("--chkpt_path", help="the path of the checkpoint to be used")
    p.add_argument("--custom_params_path", help="params yaml/json; default <chkpt>/../../params.yaml")
    p.add_argument("--uncond_scale", type=float, default=1.0, help="unconditional scale for classifier-free guidance")
    p.add_argument("--seed", type=int, help="use a specific seed for inference")
    p.add_argument("--autoreg", action="store_true", help="autoregressively inpaint the music segments")
    p.add_argument("--from_dataset", default=None, help="choose condition from a dataset {pop909, musicalion} (validation half of its split)")
    p.add_argument("--dataset_dir", help="extension: directory of the dataset's song .npz files (default: the reference's dirs.py paths)")
    p.add_argument("--split_dir", default=datasample.TRAIN_SPLIT_DIR, help="extension: directory of the <dataset>.pickle split files")
    p.add_argument("--song_index", type=int, help="extension: index into the validation list (the reference asks on the terminal)")
    p.add_argument("--song_index2", type=int, help="extension: the same for the texture song of chord+txt models")
    p.add_argument("--inpaint_song_index", type=int, help="extension: the same for --inpaint_from_dataset")
    p.add_argument("--from_midi", help="choose condition from a midi file")
    p.add_argument("--from_midi2", help="choose condition from the 2nd midi file. Used for chord+txt conditioning")
    p.add_argument("--inpaint_from_midi", help="inpaint a midi file")
    p.add_argument("--inpaint_from_dataset", default=None, help="inpaint a song from a dataset {pop909, musicalion}")
    p.add_argument("--inpaint_pop909_use_track", help="which tracks to use as base song for inpainting, default: 0,1,2")
    p.add_argument("--inpaint_type", help="inpaint a song, type: {remaining, below, above, bars}")
    p.add_argument("--ddim", action="store_true", help="whether to use DDIM sampler")
    p.add_argument("--ddim_discretize", default="uniform", help="{uniform(default), quad}")
    p.add_argument("--ddim_eta", type=float, default=0.0, help="ddim eta, default: 0.0")
    p.add_argument("--ddim_steps", type=int, default=50, help="number of ddim sampling steps, default: 50")
    p.add_argument("--repaint_n", type=int, default=1, help="n sampling steps in RePaint")
    p.add_argument("--length", type=int, default=0, help="the generated length (in 8-bars)")
    p.add_argument("--show_image", action="store_true", help="(ignored: no image writer on this path)")
    p.add_argument("--chkpt_name", default="weights_best.pt")
    p.add_argument("--num_generate", type=int, default=1, help="the number of samples to generate")
    p.add_argument("--output_dir", default="exp", help="directory to store generated piano rolls")
    # extensions of this build
    p.add_argument("--params_preset", help="use a built-in params preset instead of a params.yaml (e.g. sdf_chd8bar)")
    p.add_argument("--synthetic_weights", action="store_true", help="deterministic synthetic weights instead of a checkpoint")
    p.add_argument("--synthetic", action="store_true", help="seeded synthetic chords / textures as conditions")
    p.add_argument("--cond_npz", help="npz with arrays chord [B,32,36] and/or prmat [B,128,128] and/or pnotree [B,128,20,6] (and prmat2c for inpainting)")
    p.add_argument("--from_song_npz", help="a quantised song in the reference's data-dictionary format (notes, start_table, db_pos, "
                   "db_pos_filter, chord - what get_data_for_single_midi / the POP909 .npz files hold): its 8-bar segments supply the "
                   "chord / texture conditions and the image to inpaint (ref:inference_sdf.py:599-610 via data/datasample.py)")
    p.add_argument("--bar_list", help="bars to inpaint for --inpaint_type bars, comma separated")
    p.add_argument("--precision", choices=["auto", "f32", "bf16x3", "f16x3", "auto-f16x3"], default="auto", help="arithmetic of the denoiser's "
                   "contractions: f32 = exact fp32 MFMA; bf16x3 = error-compensated bf16 split (about 3x faster, 5e-5 from the reference on "
                   "ordinarily-conditioned weights); f16x3 = the same split in fp16 pieces (libpfhip_f16.so: fp32-class error at ~0.97 of bf16x3's "
                   "speed, activations must stay below 65504); auto (default) = evaluate f32 and bf16x3 once on this checkpoint and keep bf16x3 "
                   "only if they agree to 3e-4 of the output scale; auto-f16x3 = the same probe for f16x3 (it catches a range overflow)")
    p.add_argument("--hip_graph", action="store_true", help="capture one reverse step as a hipGraph and replay it (same results; "
                   "removes the host-side launch cost that bounds small batches, e.g. the batch-1 runs of --autoreg)")
    return p


def _packed_blobs(params, args, parts, chord_enc, txt_enc):
    """Rank 0's half of ``load_model``: checkpoint (or synthetic) state_dicts -> packed host blobs, one per sub-model."""
    from .checkpoint import load_checkpoint, split_state_full
    states = {}
    if args.synthetic_weights:
        from .arch import UNetConfig
        from .weights import synth_chord_encoder_state, synth_pianotree_encoder_state, synth_texture_encoder_state, synth_unet_state
        if params.cond_type == "pnotree":
            states["pnotree_enc"] = synth_pianotree_encoder_state(0)
        states["unet"] = synth_unet_state(UNetConfig.from_params(params), 0)
        if chord_enc is not None:
            states["chord_enc"] = synth_chord_encoder_state(0, params.chd_input_dim, params.chd_hidden_dim, params.chd_z_dim)
        if txt_enc is not None:
            states["txt_enc"] = synth_texture_encoder_state(0, params.txt_emb_size, params.txt_hidden_dim, params.txt_z_dim,
                                                            params.txt_num_channel)
    else:
        path = args.chkpt_path
        if path and os.path.exists(f"{path}/chkpts/{args.chkpt_name}"):
            path = f"{path}/chkpts/{args.chkpt_name}"
        if not path or not (path.endswith(".pt") or path.endswith(".ckpt")):
            raise SystemExit("--chkpt_path must name a legacy .pt or a Lightning .ckpt checkpoint (or a run directory holding chkpts/)")
        states = split_state_full(load_checkpoint(path)[0])
    blobs = {}
    for name, mod, _ in parts:
        if not states.get(name):
            raise SystemExit(f"checkpoint has no {name} weights")
        blobs[name] = mod.pack_state_dict(states[name])
    return blobs


def load_model(params, args, rank: int = 0, world: int = 1) -> Polyffusion_SDF:
    """Assemble the model (``inference_sdf.py:536-557,702-734``).  Rank 0 reads the checkpoint (or generates the synthetic
    weights) and repacks it into the kernel-side blobs; with world > 1 the other ranks receive the packed blobs by ONE
    broadcast each (RCCL over xGMI) and never touch the file system - SURVEY.md 8e."""
    from . import _lib, dist as pfdist
    unet = build_unet(params, x3="f16" if getattr(args, "precision", None) in ("f16x3", "auto-f16x3") else None)
    chord_enc, txt_enc = build_encoders(params)
    parts = [("unet", unet, unet.weight_bytes())]
    if chord_enc is not None:
        parts.append(("chord_enc", chord_enc, int(_lib.load().pf_encoder_weight_bytes(chord_enc._h))))
    if txt_enc is not None:
        parts.append(("txt_enc", txt_enc, int(_lib.load().pf_encoder_weight_bytes(txt_enc._h))))
    pnotree_enc = build_pnotree_encoder(params)
    if pnotree_enc is not None:
        parts.append(("pnotree_enc", pnotree_enc, int(_lib.load().pf_encoder_weight_bytes(pnotree_enc._h))))
    dev = _dev()
    blobs, failure = {}, None
    if rank == 0:
        # every failure of the load (bad path, missing sub-model, unexpected / mis-shaped key) happens on rank 0 only while the
        # other ranks already wait for the blobs: catch it, tell them, and let every rank leave with the same message
        try:
            blobs = {name: blob.to(dev) for name, blob in _packed_blobs(params, args, parts, chord_enc, txt_enc).items()}
        except (SystemExit, Exception) as e:   # noqa: BLE001 - re-raised below on every rank
            failure = e
    if pfdist.broadcast_int(0 if failure is None else 1) != 0:
        # ONE exception type on every rank (callers that fall back - `--precision auto` - must take the same branch everywhere)
        if failure is not None:
            if isinstance(failure, SystemExit):
                raise failure
            raise SystemExit(f"could not load the model weights: {type(failure).__name__}: {failure}") from failure
        raise SystemExit("rank 0 could not load the model weights (see its message)")
    for name, mod, nbytes in parts:
        blob = blobs[name] if rank == 0 else torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
        pfdist.broadcast_blob(blob, 0)
        mod.bind_packed(blob)
    return Polyffusion_SDF(build_ldm(params, unet), params.cond_type, params.cond_mode, chord_enc=chord_enc, txt_enc=txt_enc,
                           pnotree_enc=pnotree_enc)


def make_sampler(model, args, seed: int, sample_offset: int = 0):
    """(sampler, start index) as ``inference_sdf.py:735-747,215-219`` choose them."""
    if args.ddim:
        sampler = DDIMSampler(model.ldm, args.ddim_steps, args.ddim_discretize, args.ddim_eta, seed=seed, sample_offset=sample_offset,
                              graph=getattr(args, "hip_graph", False))
        t_idx = args.ddim_steps - 1    # inference_sdf.py:218 (NOT len(time_steps)-1: 'uniform' can yield one more entry)
        if t_idx >= len(sampler.time_steps):
            raise SystemExit(f"--ddim_steps {args.ddim_steps}: the discretisation has only {len(sampler.time_steps)} steps")
        return sampler, t_idx
    return SDFSampler(model.ldm, seed=seed, sample_offset=sample_offset, graph=getattr(args, "hip_graph", False)), None


def generate_songs(model, params, args, cond, cond_mid, orig, mask, seed: int, rank: int = 0, world: int = 1, cond_concat=None):
    """This rank's share of ``args.num_generate`` independent generations of the same conditions.

    The reference loops over the songs (``inference_sdf.py:774``); here they are ONE batch and the unit of multi-GPU
    sharding: rank r takes the songs ``shard_range(num_generate, r, world)``, keyed into the noise generator by their
    GLOBAL index (``sample_offset``), so the result does not depend on the number of GPUs, and the step loop has no
    collective.  Returns (``[n_local, B, C, H, W]`` - or ``[n_local, 2B, C, H/2, W]`` with ``--autoreg`` -, the Experiments)."""
    from . import dist as pfdist
    S, B = args.num_generate, cond.shape[0]
    lo, hi = pfdist.shard_range(S, rank, world)
    n_local = hi - lo
    per_song = 1 if args.autoreg else B   # images the sampler sees per song in one paint() call
    sampler, t_idx = make_sampler(model, args, seed, lo * per_song)
    expmt = Experiments(params.model_name, params, sampler, t_idx=t_idx, repaint_n=args.repaint_n)
    rep = lambda v: None if v is None else v.unsqueeze(0).expand(n_local, *v.shape).contiguous()
    C, H, W = params.out_channels, params.img_h, params.img_w
    if n_local == 0:
        gen = torch.empty(0, 2 * B if args.autoreg else B, C, H // 2 if args.autoreg else H, W, device=cond.device)
    elif args.autoreg:
        gen = expmt.predict_songs(rep(cond), rep(cond_mid), args.uncond_scale, orig=rep(orig), mask=rep(mask),
                                  cond_concat=rep(cond_concat))   # [n,2B,C,H/2,W]
    else:
        flat = lambda v: None if v is None else rep(v).reshape(n_local * B, *v.shape[1:])
        gen = expmt.predict(flat(cond), None, args.uncond_scale, False, flat(orig), flat(mask),
                            cond_concat=flat(cond_concat)).reshape(n_local, B, C, H, W)
    return gen, expmt


def main(argv=None):
    from . import dist as pfdist
    args = make_parser().parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("inference_sdf needs an AMD GPU (no CPU fallback on this path)")
    rank, world, _ = pfdist.init_from_env()
    say = print if rank == 0 else (lambda *a, **k: None)
    if args.seed is not None:
        seed = args.seed
    else:
        # the reference seeds only on request (inference_sdf.py:510-514) and is otherwise random; the counter-based
        # generator needs SOME key, so draw one, share it across ranks and print it for reproducibility
        seed = pfdist.broadcast_int(int.from_bytes(os.urandom(4), "little"))
        say(f"seed: {seed} (pass --seed {seed} to reproduce)")
    torch.manual_seed(seed); np.random.seed(seed % (2 ** 32)); random.seed(seed)

    if args.params_preset is not None:
        params = preset(args.params_preset)
    else:
        if args.chkpt_path is None and args.custom_params_path is None:
            raise SystemExit("give --chkpt_path, --custom_params_path or --params_preset")
        try:
            params = load_params(find_params(args.chkpt_path or ".", args.custom_params_path))
        except FileNotFoundError:
            # extension: a Lightning checkpoint carries its own params (lightning_learner.py:13 save_hyperparameters)
            if not (args.chkpt_path or "").endswith(".ckpt"):
                raise
            from .checkpoint import load_lightning_ckpt
            saved = load_lightning_ckpt(args.chkpt_path)[1]
            if saved is None:
                raise
            params = Params(saved)
            params.setdefault("cond_mode", "cond")
            params.setdefault("use_enc", True)
    say(f"model_label: {params.model_name}")
    try:
        model = load_model(params, args, rank, world)
    except SystemExit as e:
        if args.precision != "auto-f16x3":
            raise
        # the fp16-split library refused the checkpoint (a weight beyond its packing's range): the exact mode of the default library instead
        say(f"f16x3 not available for this checkpoint ({e}); loading it into the default library, precision f32")
        import copy
        args = copy.copy(args)
        args.precision = "f32"
        model = load_model(params, args, rank, world)

    length = args.length
    # prmat2c_cond: the CONDITION song's image (what concat_blurry blurs, ref:inference_sdf.py:797-803 uses `prmat2c`);
    # prmat2c_inp: the song to inpaint (`prmat2c_inp`, :565-591).  Two variables, as in the reference.
    chd = prmat = prmat2c_cond = prmat2c_inp = pnotree = None
    cond_is_dummy = False
    def song_from_midi(path, tag):
        # ref:inference_sdf.py:606-612 - quantise the file, extract its chords (written next to the outputs, as the reference's exp/*.out:
        # by rank 0 only - every rank extracts the same labels and keeps them in memory)
        from . import midi_to_data
        if rank == 0:
            os.makedirs(args.output_dir, exist_ok=True)
        data = midi_to_data.get_data_for_single_midi(path, os.path.join(args.output_dir, f"chords_extracted{tag}.out") if rank == 0 else None)
        if data is None:
            raise SystemExit(f"{path}: a barline does not fall on the 16th-note grid (get_downbeat_pos_and_filter)")
        return datasample.DataSample(data).get_whole_song_data()

    def song_from_dataset(name, index, use_track=(0, 1, 2)):
        # ref:inference_sdf.py:95-118 - a song of the validation half of the dataset's split (parity unpinned: the datasets are not shipped)
        try:
            sample, fn = datasample.choose_song_from_val_dl(name, index, use_track, args.dataset_dir, args.split_dir)
        except FileNotFoundError as e:
            raise SystemExit(f"--from_dataset {name}: {e} (give --dataset_dir / --split_dir)")
        return sample.get_whole_song_data(), fn

    def need_index(flag, value):
        # choose_song_from_val_dl asks on stdin when no index is given (ref:inference_sdf.py:95-118): under a multi-rank launch every
        # rank would block on input() - the index has to come from the command line there
        if world > 1 and value is None:
            raise SystemExit(f"{flag} is required when running on more than one rank (no interactive song choice under torchrun)")
        return value

    inp_from_midi = None
    if args.inpaint_type is not None and args.inpaint_from_midi is not None:    # :569-575
        inp_from_midi = song_from_midi(args.inpaint_from_midi, "_inpaint")[0].to(_dev())
        say(f"Inpainting midi file: {args.inpaint_from_midi}")
    elif args.inpaint_type is not None and args.inpaint_from_dataset is not None:   # :576-590
        tracks = [int(v) for v in args.inpaint_pop909_use_track.split(",")] if args.inpaint_pop909_use_track else [0, 1, 2]
        (p2c, _, _, _), fn = song_from_dataset(args.inpaint_from_dataset, need_index("--inpaint_song_index", args.inpaint_song_index), tracks)
        inp_from_midi = p2c.to(_dev())
        say(f"Inpainting midi file: {fn}")
    if args.uncond_scale != 0.0 and (args.from_midi is not None or args.from_dataset is not None):
        # the reference's order (:604-624): a MIDI file wins over a dataset song when both are given
        if args.from_midi is not None:
            p2c, pnotree, chd, prmat = song_from_midi(args.from_midi, "")
            fn = args.from_midi
        else:
            if args.from_dataset == "musicalion" and "chord" in params.cond_type.split("+"):
                raise SystemExit("--from_dataset musicalion has no chord track (ref:inference_sdf.py:620 asserts cond_type != 'chord'; "
                                 "a chord+txt model would receive chd = None)")
            (p2c, pnotree, chd, prmat), fn = song_from_dataset(args.from_dataset, need_index("--song_index", args.song_index))
        chd = None if chd is None else chd.to(_dev())
        prmat, prmat2c_cond, pnotree = prmat.to(_dev()), p2c.to(_dev()), pnotree.to(_dev())
        say(f"using the {params.cond_type.split('+')[0]} of midi file: {fn}")
        if params.cond_type == "chord+txt":                                       # texture from a second song (:627-641)
            if args.from_midi2 is not None:
                prmat = song_from_midi(args.from_midi2, "")[3].to(_dev())
                say(f"using the txt of midi file: {args.from_midi2}")
            elif args.from_dataset is not None:
                (_, _, _, prmat2), fn2 = song_from_dataset(args.from_dataset, need_index("--song_index2", args.song_index2))
                prmat = prmat2.to(_dev())
                say(f"using the txt of midi file: {fn2}")
            # else (an extension; the reference raises NotImplementedError): chords and texture both from --from_midi
    elif args.from_song_npz is not None:
        p2c, pnotree, chd, prmat = datasample.DataSample.from_npz(args.from_song_npz).get_whole_song_data()
        chd, prmat, prmat2c_cond, pnotree = chd.to(_dev()), prmat.to(_dev()), p2c.to(_dev()), pnotree.to(_dev())
    elif args.uncond_scale == 0.0 and args.cond_npz is None and not args.synthetic:
        if length <= 0 and inp_from_midi is not None:
            length = inp_from_midi.shape[0]            # :596-597
        if length <= 0:
            raise SystemExit("--length is required for unconditional generation")
        prmat2c_cond, pnotree, chd, prmat = dummy_cond_input(length, params)   # zeros: what an unconditional concat_blurry model blurs
        cond_is_dummy = True
    elif args.cond_npz is not None:
        z = np.load(args.cond_npz)
        chd = torch.from_numpy(z["chord"]).float().to(_dev()) if "chord" in z else None
        prmat = torch.from_numpy(z["prmat"]).float().to(_dev()) if "prmat" in z else None
        prmat2c_cond = torch.from_numpy(z["prmat2c"]).float().to(_dev()) if "prmat2c" in z else None
        pnotree = torch.from_numpy(z["pnotree"]).long().to(_dev()) if "pnotree" in z else None
    elif args.synthetic:
        n = length if length > 0 else 1
        chd = torch.from_numpy(synth.chords(n, seed + 100)).to(_dev())
        prmat = torch.from_numpy(synth.prmat(n, seed + 200)).to(_dev())
        if params.cond_type == "pnotree":
            pnotree = torch.from_numpy(synth.pnotree(n, seed + 300)).to(_dev())
    else:
        raise SystemExit("no condition source: use --from_midi, --from_song_npz, --cond_npz, --synthetic or --uncond_scale 0 --length N")

    if params.cond_type == "pnotree" and pnotree is None:
        raise SystemExit("cond_type pnotree needs the piano-tree grid: --from_midi, --from_song_npz, pnotree in --cond_npz or --synthetic")
    cond, cond_mid = encode_conditions(model, params, chd, prmat, args.autoreg, pnotree=pnotree)
    if params.cond_mode == "uncond":
        cond = -torch.ones_like(cond)
    if length > 0:
        cond = cond[:length]
        cond_mid = cond_mid[:length] if cond_mid is not None else None
    orig = mask = None
    # the image to inpaint: --inpaint_from_midi, else (an extension of the reference, whose dataset branches are absent here) the condition song
    prmat2c_inp = inp_from_midi if inp_from_midi is not None else (None if cond_is_dummy else prmat2c_cond)
    if args.inpaint_type is not None:
        if prmat2c_inp is None:
            raise SystemExit("--inpaint_type needs the image to inpaint: --inpaint_from_midi, prmat2c in --cond_npz, or a --from_midi / --from_song_npz song")
        n = min(cond.shape[0], prmat2c_inp.shape[0])
        cond, orig = cond[:n], prmat2c_inp[:n]
        cond_mid = None if cond_mid is None else cond_mid[:n]
        bars = [int(v) for v in args.bar_list.split(",")] if args.bar_list else None
        mask = get_mask(orig, args.inpaint_type, bars).to(orig.device)

    cond_concat = None
    if params.get("concat_blurry", False):
        # inference_sdf.py:797-803 - the denoiser of this variant takes cat([x, blurry image], 1)
        if prmat2c_cond is None:
            raise SystemExit("params.concat_blurry needs the condition song's image to blur: --from_midi, --from_song_npz, prmat2c in --cond_npz, "
                             "or --uncond_scale 0 --length N (zeros)")
        cond_concat = get_blurry_image(prmat2c_cond[:cond.shape[0]], params.get("concat_ratio", 1 / 8))
        n = min(cond.shape[0], cond_concat.shape[0])
        cond, cond_concat = cond[:n], cond_concat[:n]
        cond_mid = None if cond_mid is None else cond_mid[:n]
        orig, mask = (None if v is None else v[:n] for v in (orig, mask))
    if args.precision in ("auto", "auto-f16x3"):
        def probe(mdl):
            split = mdl.ldm.eps_model.split_mode
            mode, ratio = pick_precision(mdl.ldm.eps_model, cond.reshape(-1, *cond.shape[-2:]), n_steps=params.n_steps, seed=seed)
            mode = ["f32", split][pfdist.broadcast_int(int(mode == split))]        # one decision for all ranks (rank 0's)
            mdl.ldm.eps_model.set_precision(mode)
            # printed by EVERY rank's log prefix owner (rank 0) and kept in the run's stamp: which arithmetic produced the files
            am = pick_precision.last_absmax
            say(f"precision: {mode} ({split} vs f32 on probe inputs: {ratio:.1e} of the output scale, threshold {PRECISION_PROBE_TOL:.0e}; "
                f"largest |activation| stored on the probe {am:.3g} = fp16 headroom x{65504.0 / am if am > 0 else float('inf'):.0f}"
                f"{', f16x3 needs x%g' % F16_HEADROOM_MIN if split == 'f16x3' else ''}); "
                "the probe is one evaluation on noise - for a run that must match the reference to its fp32 rounding, pass --precision f32")
            return mode
        if probe(model) == "f32" and args.precision == "auto":
            # bf16x3 does not hold this checkpoint to the contract: before paying 3x for the fp32 mode, try the fp16 split (22 mantissa bits per
            # product instead of 16, ~3 % slower than bf16x3) - the same weights loaded into the other build of the library, the same probe
            import copy
            args16 = copy.copy(args)
            args16.precision = "auto-f16x3"
            try:
                model16 = load_model(params, args16, rank, world)
            except SystemExit as e:      # e.g. a weight beyond the fp16 packing's range (load_model raises SystemExit on every rank)
                say(f"f16x3 not available for this checkpoint ({e}); staying with f32")
                model16 = None
            if model16 is not None and probe(model16) == "f16x3":
                model = model16
    else:
        model.ldm.eps_model.set_precision(args.precision)
    S, B = args.num_generate, cond.shape[0]
    say(f"generating {S} song(s) x {B} segment(s) with uncond_scale = {args.uncond_scale} on {world} GPU(s)")
    gen, expmt = generate_songs(model, params, args, cond, cond_mid, orig, mask, seed, rank, world, cond_concat=cond_concat)
    gen = pfdist.gather_rows(gen, S, rank, world)   # [S, ...] on every rank (131 KB per image; the only end-of-run exchange)
    if not bool(torch.isfinite(gen).all()):
        # the note threshold (> 0.5) would turn a NaN into silence: refuse instead.  The one known way to get here is an activation beyond
        # fp16's range in the f16x3 mode on a checkpoint the probe inputs did not exercise (include/pfhip.h pf_x3_element)
        mode = model.ldm.eps_model.precision
        raise SystemExit(f"non-finite values in the generated piano rolls (precision {mode})"
                         + ("; f16x3 overflows beyond 65504: rerun with --precision bf16x3 or f32" if mode == "f16x3" else ""))
    if rank == 0:
        os.makedirs(args.output_dir, exist_ok=True)
        for i in range(S):
            extra = "" if args.inpaint_type is None else f"_inp{args.repaint_n}_{args.inpaint_type}"
            stamp = os.path.join(args.output_dir, expmt._stamp(args.uncond_scale, args.autoreg, extra) + (f"_{i}" if S > 1 else ""))
            np.save(stamp + ".npy", gen[i].cpu().numpy())
            # generated cells on their own track when inpainting (ref:inference_sdf.py:385-389)
            midi.prmat2c_to_midi_file(gen[i], stamp + ".mid", inp_mask=None if args.autoreg else mask)
            say(f"song {i}: piano_roll {tuple(gen[i].shape)}  onsets>0.5: {int((gen[i][:, 0] > 0.5).sum())}  -> {stamp}.mid")
    pfdist.barrier()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
