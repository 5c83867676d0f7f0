"""ORACLE (test infrastructure - NOT product code).

CPU restatement of the reference reverse-diffusion loops: beta schedule,
DDPM (``sampler_sdf.py``) and DDIM (``sampler_ddim.py``) tables and steps,
classifier-free guidance, RePaint-style ``paint`` and the autoregressive
half-overlap schedule of ``inference_sdf.py``.  Noise is *injected* through a
``noise_fn(shape) -> tensor`` callable so trajectories can be pinned against
golden vectors made from the real reference with the same noise tape
(SURVEY.md Appendix C).  Paths cited relative to ``/root/reference/polyffusion``.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import numpy as np
import torch

EpsModel = Callable[[torch.Tensor, torch.Tensor, torch.Tensor], torch.Tensor]
NoiseFn = Callable[[tuple], torch.Tensor]


def beta_schedule(n_steps: int, linear_start: float, linear_end: float):
    """stable_diffusion/latent_diffusion.py:90-103 - sqrt-linear betas in f64, cast to f32."""
    beta = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_steps, dtype=torch.float64) ** 2
    alpha = 1.0 - beta
    alpha_bar = torch.cumprod(alpha, dim=0)
    return alpha.float(), beta.float(), alpha_bar.float()


def get_eps(model: EpsModel, x, t, c, uncond_scale: float, uncond_cond):
    """stable_diffusion/sampler/__init__.py:42-80 - exact-float branches on the scale."""
    if uncond_cond is None or uncond_scale == 1.0:
        return model(x, t, c)
    if uncond_scale == 0.0:
        return model(x, t, uncond_cond)
    e_u, e_c = model(torch.cat([x, x]), torch.cat([t, t]), torch.cat([uncond_cond, c])).chunk(2)
    return e_u + uncond_scale * (e_c - e_u)


class SDFSamplerRef:
    """sampler_sdf.py:37-78 tables; :80-171 p_sample; :173-192 q_sample; :257-350 paint."""

    def __init__(self, model: EpsModel, n_steps: int, linear_start: float, linear_end: float, noise_fn: NoiseFn):
        self.model, self.n_steps, self.noise_fn = model, n_steps, noise_fn
        self.alpha, self.beta, ab = beta_schedule(n_steps, linear_start, linear_end)
        self.alpha_bar = ab
        self.time_steps = np.arange(n_steps, dtype=np.int32)
        ab_prev = torch.cat([ab.new_tensor([1.0]), ab[:-1]])
        self.sqrt_alpha_bar = ab ** 0.5
        self.sqrt_1m_alpha_bar = (1.0 - ab) ** 0.5
        self.sqrt_recip_alpha_bar = ab ** -0.5
        self.sqrt_recip_m1_alpha_bar = (1 / ab - 1) ** 0.5
        var = self.beta * (1.0 - ab_prev) / (1.0 - ab)
        self.log_var = torch.log(torch.clamp(var, min=1e-20))
        self.mean_x0_coef = self.beta * (ab_prev ** 0.5) / (1.0 - ab)
        self.mean_xt_coef = (1.0 - ab_prev) * ((1 - self.beta) ** 0.5) / (1.0 - ab)

    def p_sample(self, x, c, step: int, uncond_scale=1.0, uncond_cond=None, cond_concat=None):
        t = torch.full((x.shape[0],), int(step), dtype=torch.long)
        # sampler_sdf.py:108-118 - the concat_blurry variant's denoiser sees cat([x, cond_concat], 1)
        e_t = get_eps(self.model, x if cond_concat is None else torch.cat([x, cond_concat], dim=1), t, c, uncond_scale, uncond_cond)
        x0 = self.sqrt_recip_alpha_bar[step] * x - self.sqrt_recip_m1_alpha_bar[step] * e_t
        mean = self.mean_x0_coef[step] * x0 + self.mean_xt_coef[step] * x
        noise = 0 if step == 0 else self.noise_fn(tuple(x.shape))
        x_prev = mean + (0.5 * self.log_var[step]).exp() * noise
        return x_prev, x0, e_t

    def q_sample(self, x0, index: int, noise=None):
        if noise is None:
            noise = self.noise_fn(tuple(x0.shape))
        return self.sqrt_alpha_bar[index] * x0 + self.sqrt_1m_alpha_bar[index] * noise

    def paint(self, x, cond, t_start: int, orig=None, mask=None, orig_noise=None,
              uncond_scale=1.0, uncond_cond=None, repaint_n=1, cond_concat=None):
        for step in np.flip(self.time_steps[: t_start + 1]):
            step = int(step)
            if orig is None:
                x, _, _ = self.p_sample(x, cond, step, uncond_scale, uncond_cond, cond_concat)
                continue
            x_t = x
            for u in range(repaint_n):
                noise = self.noise_fn(tuple(orig.shape)) if step > 0 else torch.zeros_like(orig)
                x_kn = self.q_sample(orig, step, noise=noise)
                x_unkn, _, _ = self.p_sample(x_t, cond, step, uncond_scale, uncond_cond, cond_concat)
                x = x_kn * mask + x_unkn * (1 - mask)
                if u < repaint_n - 1 and step > 0:
                    noise = self.noise_fn(tuple(orig.shape))
                    # quirk kept: beta (not sqrt(beta)) scales the re-noise (sampler_sdf.py:339-341)
                    x_t = (1 - self.beta[step - 1]) ** 0.5 * x + self.beta[step - 1] * noise
        return x


class DDIMSamplerRef:
    """sampler_ddim.py:40-102 tables; :168-272 step; :274-299 q_sample; :301-362 paint."""

    def __init__(self, model: EpsModel, n_steps_model: int, linear_start: float, linear_end: float,
                 n_steps: int, discretize: str = "uniform", eta: float = 0.0, noise_fn: NoiseFn = None):
        self.model, self.noise_fn = model, noise_fn
        _, _, ab = beta_schedule(n_steps_model, linear_start, linear_end)
        if discretize == "uniform":
            c = n_steps_model // n_steps
            self.time_steps = np.asarray(list(range(0, n_steps_model, c))) + 1
        elif discretize == "quad":
            self.time_steps = ((np.linspace(0, np.sqrt(n_steps_model * 0.8), n_steps)) ** 2).astype(int) + 1
        else:
            raise NotImplementedError(discretize)
        self.ddim_alpha = ab[self.time_steps].clone()
        self.ddim_alpha_sqrt = torch.sqrt(self.ddim_alpha)
        self.ddim_alpha_prev = torch.cat([ab[0:1], ab[self.time_steps[:-1]]])
        self.ddim_sigma = eta * ((1 - self.ddim_alpha_prev) / (1 - self.ddim_alpha)
                                 * (1 - self.ddim_alpha / self.ddim_alpha_prev)) ** 0.5
        self.ddim_sqrt_one_minus_alpha = (1.0 - self.ddim_alpha) ** 0.5

    def get_x_prev_and_pred_x0(self, e_t, index: int, x):
        alpha, alpha_prev = self.ddim_alpha[index], self.ddim_alpha_prev[index]
        sigma, s1m = self.ddim_sigma[index], self.ddim_sqrt_one_minus_alpha[index]
        pred_x0 = (x - s1m * e_t) / (alpha ** 0.5)
        dir_xt = (1.0 - alpha_prev - sigma ** 2).sqrt() * e_t
        noise = 0.0 if sigma == 0.0 else self.noise_fn(tuple(x.shape))
        return (alpha_prev ** 0.5) * pred_x0 + dir_xt + sigma * noise, pred_x0

    def p_sample(self, x, c, step: int, index: int, uncond_scale=1.0, uncond_cond=None, cond_concat=None):
        t = torch.full((x.shape[0],), int(step), dtype=torch.long)
        e_t = get_eps(self.model, x if cond_concat is None else torch.cat([x, cond_concat], dim=1), t, c, uncond_scale, uncond_cond)   # sampler_ddim.py:199-209
        x_prev, pred_x0 = self.get_x_prev_and_pred_x0(e_t, index, x)
        return x_prev, pred_x0, e_t

    def q_sample(self, x0, index: int, noise=None):
        if noise is None:
            noise = self.noise_fn(tuple(x0.shape))
        return self.ddim_alpha_sqrt[index] * x0 + self.ddim_sqrt_one_minus_alpha[index] * noise

    def paint(self, x, cond, t_start: int, orig=None, mask=None, orig_noise=None,
              uncond_scale=1.0, uncond_cond=None, repaint_n=1, cond_concat=None):
        time_steps = np.flip(self.time_steps[: t_start + 1])
        for i, step in enumerate(time_steps):
            index = len(time_steps) - i - 1
            x, _, _ = self.p_sample(x, cond, int(step), index, uncond_scale, uncond_cond, cond_concat)
            if orig is not None:
                x = self.q_sample(orig, index, noise=orig_noise) * mask + x * (1 - mask)
        return x


def get_autoreg_data(data: torch.Tensor, split_dim: int = 1) -> torch.Tensor:
    """inference_sdf.py:121-129 - (second half, next item's first half)."""
    steps = data.shape[split_dim]
    h1, h2 = data.split(steps // 2, dim=split_dim)
    return torch.cat((h2, h1.roll(-1, dims=0)), dim=split_dim)


def predict(sampler, cond, d_cond: int, shape: List[int], t_idx: int, noise: torch.Tensor,
            cond_mid=None, uncond_scale=1.0, autoreg=False, orig=None, mask=None, repaint_n=1, cond_concat=None):
    """inference_sdf.py:202-303 (``Experiments.predict``) with the start noise injected.  ``cond_concat`` goes to every ``paint``
    call WHOLE, as in the reference (:259-270): under ``autoreg`` only a one-segment batch fits the batch-1 runs."""
    B = cond.shape[0]
    uncond_cond = -torch.ones([B, 1, d_cond])
    if orig is None or mask is None:
        orig, mask = torch.zeros(shape), torch.zeros(shape)
    if not autoreg:
        xt = sampler.q_sample(orig, t_idx, noise)
        return sampler.paint(xt, cond, t_idx, orig=orig, mask=mask, orig_noise=noise,
                             uncond_scale=uncond_scale, uncond_cond=uncond_cond, repaint_n=repaint_n, cond_concat=cond_concat)
    half = shape[2] // 2
    orig_mid, mask_mid, noise_mid = (get_autoreg_data(v, 2) for v in (orig, mask, noise))
    uc = uncond_cond[0:1]
    gen, new_half = [], None
    for idx in range(B * 2 - 1):
        src = (cond_mid, orig_mid, mask_mid, noise_mid) if idx % 2 == 1 else (cond, orig, mask, noise)
        c_s, o_s, m_s, n_s = (v[idx // 2].unsqueeze(0) for v in src)  # views: in-place edits persist
        if idx != 0:
            o_s[:, :, 0:half, :] = new_half
            m_s[:, :, 0:half, :] = 1
        xt = sampler.q_sample(o_s, t_idx, n_s)
        x0 = sampler.paint(xt, c_s, t_idx, orig=o_s, mask=m_s, orig_noise=n_s,
                           uncond_scale=uncond_scale, uncond_cond=uc, repaint_n=repaint_n, cond_concat=cond_concat)
        if idx == 0:
            gen.append(x0[:, :, 0:half, :])
        new_half = x0[:, :, half:, :]
        gen.append(new_half)
    return torch.cat(gen, dim=0)


def get_mask(orig: torch.Tensor, inpaint_type: str, bar_list=None) -> torch.Tensor:
    """inference_sdf.py:132-193 - inpainting masks (1 = keep).  Row-by-row like the reference, incl. its quirks:
    "no onset" is recognised by the edge value (0 for ``below``, 127 for ``above``), the leading fill tests
    ``!= 0`` for both types, and row 0 falls back to row -1 (wrap-around)."""
    if inpaint_type == "remaining":
        return orig.clone()
    if inpaint_type == "bars":
        keep = torch.ones_like(orig)
        for bar in bar_list:
            keep[:, :, 16 * bar: 16 * (bar + 1), :] = 0
        return keep
    if inpaint_type not in ("below", "above"):
        raise NotImplementedError(inpaint_type)
    n, steps, pitches = orig.shape[0], orig.shape[2], orig.shape[3]
    rows = orig[:, 0].reshape(n * steps, pitches)
    if inpaint_type == "below":
        edge, none = rows.argmax(dim=1), 0
    else:
        edge, none = 127 - rows.flip(1).argmax(dim=1), 127
    lead = int(edge.nonzero()[0])
    edge[:lead] = edge[lead]
    out = torch.zeros_like(rows)
    for r in range(n * steps):
        if edge[r] == none:
            edge[r] = edge[r - 1]
        if inpaint_type == "below":
            out[r, int(edge[r]):] = 1
        else:
            out[r, : int(edge[r]) + 1] = 1
    return out.reshape(n, 1, steps, pitches).expand(-1, 2, -1, -1)
