"""Reverse-diffusion loops on the GPU: mirrors of the reference sampler classes.

* ``DiffusionSampler``  - ``stable_diffusion/sampler/__init__.py:25-80`` (``get_eps`` with
  classifier-free guidance; exact-float branches on the scale are kept).
* ``SDFSampler``        - ``sampler_sdf.py`` (tables :52-78, ``p_sample`` :80-171, ``q_sample``
  :173-192, ``sample`` :194-255, ``paint`` :257-350 incl. the RePaint quirks of Appendix A).
* ``DDIMSampler``       - ``sampler_ddim.py`` (tables :40-102, step :168-272, ``paint`` :301-362).

Same method names and argument meaning as the reference, so ``Experiments.predict`` drives
them unchanged.  Differences, all host-side plumbing:
  - per-step scalars are read from host tables and passed by value into ONE fused HIP kernel
    per step (``pf_ddpm_step`` / ``pf_ddim_step``) instead of ~20 elementwise launches and five
    device->host syncs per step;
  - noise comes from ``noise_fn(shape)`` when given (parity tests inject the reference's noise
    tape) and otherwise from the counter-based on-device generator keyed by (seed, draw counter,
    global sample index), so a batch sharded over N GPUs draws the same noise per sample as the
    unsharded batch.  Inside the loops the draws happen IN the update kernel
    (``pf_ddpm_step_rng`` / ``pf_ddim_step_rng``: bit-identical to ``pf_randn`` + step, two
    launches and the noise tensors' HBM round trips less per step);
  - what the eps model computes from ``t`` and ``cond`` alone - the time MLP with every
    ResBlock's ``emb_layers`` and, for one context token, the whole cross-attention - is hoisted
    out of the loops: ``prepare()`` builds it once per ``paint()`` / ``sample()`` call and every
    step's forward receives it (``UNetModel.forward(time_table=, cross_bias=)``).  ``p_sample`` /
    ``get_eps`` called on their own keep working unprepared.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, List, Optional

import numpy as np
import torch

from . import _lib
from .unet import LatentDiffusion

NoiseFn = Callable[[tuple], torch.Tensor]


class DiffusionSampler:
    model: LatentDiffusion

    def __init__(self, model: LatentDiffusion, noise_fn: Optional[NoiseFn] = None, seed: int = 0, sample_offset: int = 0,
                 graph: bool = False):
        self.model = model
        self.n_steps = model.n_steps
        self.noise_fn = noise_fn
        self.seed = int(seed)
        self.sample_offset = int(sample_offset)  # global index of this rank's first sample (multi-GPU sharding)
        self._draws = 0
        # classifier-free guidance evaluates the denoiser on cat([x, x]); the part of it that cannot depend on the condition is shared
        # between the halves unless this is switched off (results equal up to tile-choice rounding; tests compare both)
        self.share_cfg_prefix = True
        self._lib = _lib.load()
        self._trows = None
        # graph=True: paint() captures ONE reverse step as a hipGraph and replays it.  What varies per step
        # (table row, time-step value, noise draw counter) lives in a device-resident pf_step_state, the coefficient table is on
        # the device, x is updated in place: a replay issues no host-side launches (about 220 per step in eager mode), which is
        # what bounds small batches (batch 1 of the --autoreg CLI, batch 8 per GPU of BASELINE config 5).  Results are bit-identical
        # to the eager loop.  Used only with the on-device noise generator (an injected noise_fn is host code).
        self.graph = bool(graph)
        self._dev_cache = {}
        # captured steps, kept across paint() calls (the autoregressive schedule calls paint() 2B-1 times with the same shapes):
        # key = everything that shapes the captured launch sequence, value = the graph + the static buffers its nodes point at
        self._graphs = {}
        self.graph_captures = 0
        # on_step(step, x): called by the eager loops of paint() after every reverse step with the step's number and its result
        # (the reference's `is_show_image` hook, sampler_sdf.py:338-345, as a callback).  None = no call, no cost.  Used by the
        # full-length parity runs (tools/long_parity.py) to record how two arithmetic modes drift apart along one noise tape.
        self.on_step: Optional[Callable[[int, torch.Tensor], None]] = None

    def _step_state(self, device) -> torch.Tensor:
        key = ("state", str(device))
        if key not in self._dev_cache:
            self._dev_cache[key] = torch.zeros(2, dtype=torch.int64, device=device)   # pf_step_state {int64 index; uint64 draws}
        return self._dev_cache[key]

    def _set_state(self, st: torch.Tensor, index: int):
        _lib.check(self._lib.pf_step_state_set(st.data_ptr(), int(index), int(self._draws), _lib.current_stream()), "pf_step_state_set")

    _MAX_GRAPHS = 4

    def _graph_token(self):
        """What a captured step depends on besides its own buffers: the UNet's workspace and weight blob (addresses are baked
        into the graph nodes; the workspace is re-allocated when a larger batch comes along) and the arithmetic mode."""
        m = self.model.eps_model
        ws = getattr(m, "_ws", None)
        blob = getattr(m, "_blob_dev", None)
        return (0 if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel(), 0 if blob is None else blob.data_ptr(),
                getattr(m, "precision", None)) + (m.options_key() if hasattr(m, "options_key") else ())

    def _graph_for(self, key, inputs, make_body, reset, uncond_scale=1.0):
        """The captured step for `key`, with `inputs` (name -> tensor or None) copied into its static buffers.  A cached entry is
        dropped when the UNet's workspace / weights / mode changed since its capture.  `make_body(bufs)` returns the step closure
        over the static buffers, `reset()` re-arms the device step state (called after every buffer refresh).  Returns the entry
        {"g": graph, "bufs": {...}}.  bufs["_prep"] holds the hoisted prefix (prepare()) in static buffers of its own: it is
        recomputed IN PLACE from the refreshed cond buffers on every call, so the captured forward always reads current values."""
        ent = self._graphs.get(key)
        if ent is not None and ent["token"] != self._graph_token():
            del self._graphs[key]
            ent = None
        if ent is None:
            bufs = {k: (None if v is None else v.clone()) for k, v in inputs.items()}
            bufs["_prep"] = self.prepare(bufs["cond"], uncond_scale=uncond_scale, uncond_cond=bufs.get("uncond_cond"))

            def restore():
                bufs["x"].copy_(inputs["x"])
                reset()

            reset()
            g = self._capture(make_body(bufs), restore)
            self.graph_captures += 1
            while len(self._graphs) >= self._MAX_GRAPHS:
                self._graphs.pop(next(iter(self._graphs)))
            ent = self._graphs[key] = {"g": g, "bufs": bufs, "token": self._graph_token()}   # token AFTER the warm-up sized the workspace
        else:
            for k, v in inputs.items():
                if v is not None:
                    ent["bufs"][k].copy_(v)
            self.prepare(ent["bufs"]["cond"], uncond_scale=uncond_scale, uncond_cond=ent["bufs"].get("uncond_cond"), out=ent["bufs"]["_prep"])
            reset()
        return ent

    @staticmethod
    def _shape_key(**tensors):
        return tuple((k, None if v is None else (tuple(v.shape), str(v.dtype))) for k, v in sorted(tensors.items()))

    def _capture(self, body, restore):
        """Warm `body` up once on a side stream (sizes the workspace, primes the allocator), undo its effect with `restore`,
        then capture it.  Returns the instantiated graph."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            body()
            restore()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body()
        restore()
        return g

    # ---- noise ------------------------------------------------------------------------------------
    def randn(self, shape, device) -> torch.Tensor:
        if self.noise_fn is not None:
            return self.noise_fn(tuple(shape)).to(device=device, dtype=torch.float32).contiguous()
        out = torch.empty(tuple(shape), dtype=torch.float32, device=device)
        per_sample = int(np.prod(shape[1:]))
        _lib.check(self._lib.pf_randn(out.data_ptr(), out.numel(), self.seed, self._draws,
                                      self.sample_offset * per_sample, _lib.current_stream()), "pf_randn")
        self._draws += 1
        return out

    # ---- the step-invariant prefix of the eps model, hoisted out of the loops ------------------------------
    def prepare(self, cond: torch.Tensor, *, uncond_scale: float = 1.0, uncond_cond: Optional[torch.Tensor] = None,
                out: Optional[dict] = None) -> dict:
        """What ``get_eps(x, t, cond, uncond_scale=, uncond_cond=)`` computes WITHOUT looking at x, for every t of the schedule:
        ``time_table`` [n_steps, W] (time MLP + all ``emb_layers`` per time-step value, ``unet.py:181-182, 286-289``) and ``cross`` -
        the collapsed one-token cross-attention biases (``unet_attention.py:186-212``) of exactly the context batch ``get_eps`` feeds
        the model: ``cond``, ``uncond_cond`` or ``cat([uncond_cond, cond])`` by the reference's exact-float branches
        (``sampler/__init__.py:63-74``).  ``None`` entries (several context tokens) are computed inside every forward as before.
        Called explicitly by the loops of this class; pass the result as ``prep=`` with the SAME cond / guidance arguments.
        ``out``: a previous result whose buffers are overwritten in place (captured-graph replay)."""
        m = self.model.eps_model
        if uncond_cond is None or uncond_scale == 1.0:
            ctx = cond
        elif uncond_scale == 0.0:
            ctx = uncond_cond
        else:
            ctx = torch.cat([uncond_cond, cond])
        o = out or {}
        # one row more than the schedule has steps: the DDIM tau table is shifted by +1 (sampler_ddim.py:63-73)
        return {"time_table": m.prepare_time(self.model.n_steps + 1, out=o.get("time_table")),
                "cross": m.prepare_cond(ctx, out=o.get("cross"))}

    @staticmethod
    def _prep_kw(prep):
        return {} if prep is None else {"time_table": prep["time_table"], "cross_bias": prep["cross"]}

    # ---- eps with classifier-free guidance ----------------------------------------------------------
    def get_eps(self, x: torch.Tensor, t: torch.Tensor, c: torch.Tensor, *, uncond_scale: float,
                uncond_cond: Optional[torch.Tensor], prep: Optional[dict] = None):
        kw = self._prep_kw(prep)
        if uncond_cond is None or uncond_scale == 1.0:
            return self.model(x, t, c, **kw)
        elif uncond_scale == 0.0:
            return self.model(x, t, uncond_cond, **kw)
        # re-concatenated on every call like the reference (sampler/__init__.py:69-74): [2B,n_cond,d_cond] is a few KB, and a
        # cache keyed on addresses could serve a stale tensor after an in-place update or an allocator address reuse
        if self.share_cfg_prefix and getattr(self.model, "supports_shared_x", False):
            # the two halves see the same x and t and differ only in the condition, which enters the UNet at its first transformer
            # block: everything in front of it is evaluated once for both (UNetModel.forward(shared_x=True), pf_unet_forward_cfg)
            eps2 = self.model(x, torch.cat([t, t]), torch.cat([uncond_cond, c]), shared_x=True, **kw)
        else:
            eps2 = self.model(torch.cat([x, x]), torch.cat([t, t]), torch.cat([uncond_cond, c]), **kw)
        e_t = torch.empty_like(eps2[: x.shape[0]])   # eps has out_channels; x may carry extra cond_concat channels
        _lib.check(self._lib.pf_cfg_combine(eps2.data_ptr(), float(uncond_scale), e_t.data_ptr(), e_t.numel(),
                                            _lib.current_stream()), "pf_cfg_combine")
        return e_t

    def _t_of(self, step: int, batch: int, device) -> torch.Tensor:
        """[batch] int64 on the device, all == step: a row of a constant [n_steps, batch] table (no fill launch per step)."""
        tr = self._trows
        if tr is None or tr.shape[1] != batch or tr.device != device:
            tr = self._trows = torch.arange(self.model.n_steps + 1, dtype=torch.long, device=device).unsqueeze(1).repeat(1, batch).contiguous()
        return tr[int(step)]

    def _eps(self, x, c, step, uncond_scale, uncond_cond, cond_concat, prep=None):
        t = self._t_of(step, x.shape[0], x.device)
        xin = x if cond_concat is None else torch.cat([x, cond_concat], dim=1)
        return self.get_eps(xin, t, c, uncond_scale=uncond_scale, uncond_cond=uncond_cond, prep=prep)

    def _rng_ok(self, x, *others) -> bool:
        """The in-kernel noise path: on-device generator, whole Philox groups per tensor and per shard, and every tensor the
        update kernel reads as 16-byte vectors (x and, when present, orig / orig_noise / mask) 16-byte aligned - anything else takes
        the randn() + step path, which reads 4-byte elements.  BOTH paths read their tensors as dense buffers through raw pointers: the
        callers (``_step``, ``repaint_step``, ``p_sample``) make every tensor contiguous before they get here."""
        per_sample = x.numel() // x.shape[0]
        if self.noise_fn is not None or x.numel() % 4 != 0 or (self.sample_offset * per_sample) % 4 != 0:
            return False
        return all(v is None or (v.is_contiguous() and v.data_ptr() % 16 == 0) for v in (x,) + others)


class SDFSampler(DiffusionSampler):
    def __init__(self, model: LatentDiffusion, is_show_image: bool = False, **kw):
        super().__init__(model, **kw)
        self.time_steps = np.asarray(list(range(self.n_steps)), dtype=np.int32)
        self.is_show_image = is_show_image
        ab, beta = model.alpha_bar, model.beta  # fp32 host tensors
        ab_prev = torch.cat([ab.new_tensor([1.0]), ab[:-1]])
        self.sqrt_alpha_bar = ab ** 0.5
        self.sqrt_1m_alpha_bar = (1.0 - ab) ** 0.5
        self.sqrt_recip_alpha_bar = ab ** -0.5
        self.sqrt_recip_m1_alpha_bar = (1 / ab - 1) ** 0.5
        variance = beta * (1.0 - ab_prev) / (1.0 - ab)
        self.log_var = torch.log(torch.clamp(variance, min=1e-20))
        self.mean_x0_coef = beta * (ab_prev ** 0.5) / (1.0 - ab)
        self.mean_xt_coef = (1.0 - ab_prev) * ((1 - beta) ** 0.5) / (1.0 - ab)
        self._sigma = (0.5 * self.log_var).exp()

    def _coef_table(self, device) -> torch.Tensor:
        """[n_steps, 7] fp32 on the device: row t = the pf_ddpm_coef of step t (same floats `_coef` passes by value)."""
        key = ("ddpm_table", str(device))
        if key not in self._dev_cache:
            rows = torch.stack([self.sqrt_recip_alpha_bar, self.sqrt_recip_m1_alpha_bar, self.mean_x0_coef, self.mean_xt_coef,
                                self._sigma, self.sqrt_alpha_bar, self.sqrt_1m_alpha_bar], dim=1).to(torch.float32).contiguous()
            self._dev_cache[key] = rows.to(device)
        return self._dev_cache[key]

    def _paint_graph(self, x, cond, t_start, orig, mask, uncond_scale, uncond_cond, cond_concat):
        """Steps t_start .. 1 as replays of one captured step; returns x_1 (the caller runs step 0, which draws no noise)."""
        lib, dev, B, n = self._lib, x.device, x.shape[0], x.numel()
        table, st = self._coef_table(dev), self._step_state(dev)
        off = self.sample_offset * (n // B)
        ndraw = 2 if orig is not None else 1
        inputs = dict(x=x, cond=cond, orig=orig, mask=mask, uncond_cond=uncond_cond, cond_concat=cond_concat)

        rng = self._rng_ok(x)

        def make_body(bf):
            xb, tb = bf["x"], torch.empty(B, dtype=torch.long, device=dev)
            npz = None if rng else torch.empty_like(xb)
            nq = torch.empty_like(xb) if (bf["orig"] is not None and not rng) else None
            bf["_keep"] = (tb, npz, nq)

            def body():
                stream = _lib.current_stream()
                _lib.check(lib.pf_step_begin(st.data_ptr(), None, tb.data_ptr(), B, stream), "pf_step_begin")
                xin = xb if bf["cond_concat"] is None else torch.cat([xb, bf["cond_concat"]], dim=1)
                e_t = self.get_eps(xin, tb, bf["cond"], uncond_scale=uncond_scale, uncond_cond=bf["uncond_cond"], prep=bf["_prep"])
                if rng:   # both draws inside the update kernel (draw order of the reference: known-region noise, then the p_sample noise)
                    _lib.check(lib.pf_ddpm_step_rng_dev(xb.data_ptr(), e_t.data_ptr(), _lib.ptr(bf["orig"]), _lib.ptr(bf["mask"]), table.data_ptr(),
                                                        st.data_ptr(), self.seed, off, xb.data_ptr(), n, stream), "pf_ddpm_step_rng_dev")
                else:
                    if nq is not None:
                        _lib.check(lib.pf_randn_dev(nq.data_ptr(), n, self.seed, st.data_ptr(), 0, off, stream), "pf_randn_dev")
                    _lib.check(lib.pf_randn_dev(npz.data_ptr(), n, self.seed, st.data_ptr(), ndraw - 1, off, stream), "pf_randn_dev")
                    _lib.check(lib.pf_ddpm_step_dev(xb.data_ptr(), e_t.data_ptr(), npz.data_ptr(), _lib.ptr(nq), _lib.ptr(bf["orig"]),
                                                    _lib.ptr(bf["mask"]), table.data_ptr(), st.data_ptr(), xb.data_ptr(), n, stream),
                               "pf_ddpm_step_dev")
                _lib.check(lib.pf_step_end(st.data_ptr(), ndraw, stream), "pf_step_end")
            return body

        key = ("ddpm", str(dev), float(uncond_scale), self.seed, off) + self._shape_key(**inputs)
        ent = self._graph_for(key, inputs, make_body, lambda: self._set_state(st, t_start), uncond_scale)
        g = ent["g"]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(t_start):
            g.replay()
        e1.record()
        self.last_replay = (e0, e1, t_start)     # bench.py reads the per-step replay time from these events
        self._draws += ndraw * t_start
        return ent["bufs"]["x"].clone()   # the static buffer is overwritten by the next paint() with these shapes

    def _coef(self, step: int) -> _lib.DdpmCoef:
        return _lib.DdpmCoef(float(self.sqrt_recip_alpha_bar[step]), float(self.sqrt_recip_m1_alpha_bar[step]),
                             float(self.mean_x0_coef[step]), float(self.mean_xt_coef[step]), float(self._sigma[step]),
                             float(self.sqrt_alpha_bar[step]), float(self.sqrt_1m_alpha_bar[step]))

    @torch.no_grad()
    def p_sample(self, x, c, t, step: int, repeat_noise: bool = False, temperature: float = 1.0,
                 uncond_scale: float = 1.0, uncond_cond=None, cond_concat=None, return_x0: bool = True, prep: Optional[dict] = None):
        """Returns (x_prev, x0, e_t) like the reference (sampler_sdf.py:80-171).  ``x0`` costs two extra elementwise
        launches; the loops of this class (``sample`` / ``paint``) pass ``return_x0=False`` and get ``None`` for it.
        ``prep``: the result of ``prepare(c, uncond_scale=, uncond_cond=)`` when the caller hoisted it."""
        step = int(step)
        x = x.contiguous()
        e_t = self._eps(x, c, step, uncond_scale, uncond_cond, cond_concat, prep)
        coef = self._coef(step)
        x_prev = torch.empty_like(x)
        if step != 0 and not repeat_noise and temperature == 1.0 and self._rng_ok(x):
            # the draw happens inside the update kernel: same values as randn() + pf_ddpm_step, one launch
            per_sample = x.numel() // x.shape[0]
            _lib.check(self._lib.pf_ddpm_step_rng(x.data_ptr(), e_t.data_ptr(), None, None, C.byref(coef), self.seed, 0, self._draws,
                                                  self.sample_offset * per_sample, x_prev.data_ptr(), x.numel(), _lib.current_stream()),
                       "pf_ddpm_step_rng")
            self._draws += 1
            x0 = (coef.c_recip * x - coef.c_recipm1 * e_t) if return_x0 else None
            return x_prev, x0, e_t
        noise = None
        if step != 0:
            noise = self.randn((1, *x.shape[1:]) if repeat_noise else x.shape, x.device)
            if repeat_noise:
                noise = noise.expand_as(x).contiguous()
            if temperature != 1.0:
                noise = noise * temperature
        _lib.check(self._lib.pf_ddpm_step(x.data_ptr(), e_t.data_ptr(), _lib.ptr(noise), None, None, None, C.byref(coef),
                                          x_prev.data_ptr(), x.numel(), _lib.current_stream()), "pf_ddpm_step")
        x0 = (coef.c_recip * x - coef.c_recipm1 * e_t) if return_x0 else None
        return x_prev, x0, e_t

    @torch.no_grad()
    def q_sample(self, x0: torch.Tensor, index: int, noise: Optional[torch.Tensor] = None):
        if noise is None:
            noise = self.randn(x0.shape, x0.device)
        out = torch.empty_like(x0)
        _lib.check(self._lib.pf_axpby(x0.contiguous().data_ptr(), noise.contiguous().data_ptr(),
                                      float(self.sqrt_alpha_bar[index]), float(self.sqrt_1m_alpha_bar[index]),
                                      out.data_ptr(), out.numel(), _lib.current_stream()), "pf_axpby")
        return out

    @torch.no_grad()
    def sample(self, shape: List[int], cond, repeat_noise=False, temperature=1.0, x_last=None, uncond_scale=1.0,
               uncond_cond=None, t_start: int = 0):
        x = x_last if x_last is not None else self.randn(shape, cond.device)
        prep = self.prepare(cond, uncond_scale=uncond_scale, uncond_cond=uncond_cond)
        for step in np.flip(self.time_steps)[t_start:]:
            x, _, _ = self.p_sample(x, cond, None, int(step), repeat_noise=repeat_noise, temperature=temperature,
                                    uncond_scale=uncond_scale, uncond_cond=uncond_cond, return_x0=False, prep=prep)
        return x

    def repaint_step(self, x_t, cond, step: int, orig, mask, *, uncond_scale: float = 1.0, uncond_cond=None, cond_concat=None,
                     prep: Optional[dict] = None):
        """ONE iteration of ``paint``'s loop body with a known region (sampler_sdf.py:307-336): eps, the reverse step on the unknown
        region, the forward-noised known region, the blend.  Two noise draws for step > 0 - known-region noise first, then the
        ``p_sample`` noise, the reference's order - made inside the update kernel when the on-device generator is in use.
        ``bench.py`` times exactly this method."""
        lib, step, n = self._lib, int(step), x_t.numel()
        coef = self._coef(step)
        x_t, orig, mask = x_t.contiguous(), orig.contiguous(), mask.contiguous()     # the update kernels read dense buffers
        e_t = self._eps(x_t, cond, step, uncond_scale, uncond_cond, cond_concat, prep)
        x = torch.empty_like(x_t)
        if step > 0 and self._rng_ok(x_t, orig, mask):
            _lib.check(lib.pf_ddpm_step_rng(x_t.data_ptr(), e_t.data_ptr(), orig.data_ptr(), mask.data_ptr(), C.byref(coef), self.seed,
                                            self._draws, self._draws + 1, self.sample_offset * (n // x_t.shape[0]), x.data_ptr(), n,
                                            _lib.current_stream()), "pf_ddpm_step_rng")
            self._draws += 2
            return x
        # injected noise tape (parity tests) / step 0: the draws are tensors.  The tape's order is the reference's: q before the eps
        # evaluation, p after it - randn() only counts draws, so drawing both here keeps the tape aligned
        noise_q = self.randn(orig.shape, x_t.device) if step > 0 else None
        noise_p = self.randn(x_t.shape, x_t.device) if step > 0 else None
        _lib.check(lib.pf_ddpm_step(x_t.data_ptr(), e_t.data_ptr(), _lib.ptr(noise_p), _lib.ptr(noise_q), orig.data_ptr(), mask.data_ptr(),
                                    C.byref(coef), x.data_ptr(), n, _lib.current_stream()), "pf_ddpm_step")
        return x

    @torch.no_grad()
    def paint(self, x, cond, t_start: int, orig=None, mask=None, orig_noise=None, uncond_scale: float = 1.0,
              uncond_cond=None, cond_concat=None, repaint_n: int = 1):
        """RePaint-style loop.  ``orig_noise`` is ignored exactly like the reference (fresh noise per step)."""
        lib, stream = self._lib, _lib.current_stream
        x = x.contiguous()
        if orig is not None:
            assert mask is not None
            orig, mask = orig.contiguous().float(), mask.contiguous().float()
        n = x.numel()
        steps = np.flip(self.time_steps[: t_start + 1])
        if self.graph and self.noise_fn is None and repaint_n == 1 and t_start >= 1 and self.on_step is None:
            x = self._paint_graph(x, cond, int(t_start), orig, mask, uncond_scale, uncond_cond, cond_concat)
            steps = steps[-1:]          # step 0 (no noise) runs eagerly below
        # everything the denoiser derives from (t, cond) alone, once for the whole loop
        prep = self.prepare(cond, uncond_scale=uncond_scale, uncond_cond=uncond_cond)
        for step in steps:
            step = int(step)
            if orig is None:
                x, _, _ = self.p_sample(x, cond, None, step, uncond_scale=uncond_scale, uncond_cond=uncond_cond,
                                        cond_concat=cond_concat, return_x0=False, prep=prep)
                if self.on_step is not None:
                    self.on_step(step, x)
                continue
            x_t = x
            for u in range(repaint_n):
                x = self.repaint_step(x_t, cond, step, orig, mask, uncond_scale=uncond_scale, uncond_cond=uncond_cond,
                                      cond_concat=cond_concat, prep=prep)
                if u < repaint_n - 1 and step > 0:
                    noise = self.randn(orig.shape, x.device)
                    b = self.model.beta[step - 1]
                    x_t = torch.empty_like(x)
                    _lib.check(lib.pf_axpby(x.data_ptr(), noise.data_ptr(), float((1 - b) ** 0.5), float(b),
                                            x_t.data_ptr(), n, stream()), "pf_axpby")
            if self.on_step is not None:
                self.on_step(step, x)
        return x


class DDIMSampler(DiffusionSampler):
    def __init__(self, model: LatentDiffusion, n_steps: int, ddim_discretize: str = "uniform", ddim_eta: float = 0.0,
                 is_show_image: bool = False, **kw):
        super().__init__(model, **kw)
        self.is_show_image = is_show_image
        self.n_steps = model.n_steps
        if ddim_discretize == "uniform":
            c = self.n_steps // n_steps
            self.time_steps = np.asarray(list(range(0, self.n_steps, c))) + 1
        elif ddim_discretize == "quad":
            self.time_steps = ((np.linspace(0, np.sqrt(self.n_steps * 0.8), n_steps)) ** 2).astype(int) + 1
        else:
            raise NotImplementedError(ddim_discretize)
        ab = model.alpha_bar
        self.ddim_alpha = ab[self.time_steps].clone().to(torch.float32)
        self.ddim_alpha_sqrt = torch.sqrt(self.ddim_alpha)
        self.ddim_alpha_prev = torch.cat([ab[0:1], ab[self.time_steps[:-1]]])
        self.ddim_sigma = (ddim_eta * ((1 - self.ddim_alpha_prev) / (1 - self.ddim_alpha)
                                       * (1 - self.ddim_alpha / self.ddim_alpha_prev)) ** 0.5)
        self.ddim_sqrt_one_minus_alpha = (1.0 - self.ddim_alpha) ** 0.5

    def _coef_table(self, device):
        """([S, 7] fp32 coefficient rows, [S] int32 tau table) on the device - the floats `_coef` passes by value."""
        key = ("ddim_table", str(device))
        if key not in self._dev_cache:
            rows = torch.tensor([[getattr(c, f) for f, _ in _lib.DdimCoef._fields_] for c in map(self._coef, range(len(self.time_steps)))],
                                dtype=torch.float32)
            self._dev_cache[key] = (rows.to(device), torch.from_numpy(np.asarray(self.time_steps, dtype=np.int32)).to(device))
        return self._dev_cache[key]

    def _paint_graph(self, x, cond, t_start, orig, mask, orig_noise, uncond_scale, uncond_cond, cond_concat, noisy: bool):
        """All t_start+1 steps of a DDIM loop as replays of one captured step.  ``noisy``: every step of the range has sigma != 0
        (eta > 0) and draws one noise tensor, in the eager loop's draw order; otherwise (eta = 0) no step draws."""
        lib, dev, B, n = self._lib, x.device, x.shape[0], x.numel()
        (table, taus), st = self._coef_table(dev), self._step_state(dev)
        off = self.sample_offset * (n // B)
        inputs = dict(x=x, cond=cond, orig=orig, mask=mask, orig_noise=orig_noise, uncond_cond=uncond_cond, cond_concat=cond_concat)

        rng = noisy and self._rng_ok(x)

        def make_body(bf):
            xb, tb = bf["x"], torch.empty(B, dtype=torch.long, device=dev)
            nz = torch.empty_like(xb) if (noisy and not rng) else None
            bf["_keep"] = (tb, nz)

            def body():
                stream = _lib.current_stream()
                _lib.check(lib.pf_step_begin(st.data_ptr(), taus.data_ptr(), tb.data_ptr(), B, stream), "pf_step_begin")
                xin = xb if bf["cond_concat"] is None else torch.cat([xb, bf["cond_concat"]], dim=1)
                e_t = self.get_eps(xin, tb, bf["cond"], uncond_scale=uncond_scale, uncond_cond=bf["uncond_cond"], prep=bf["_prep"])
                if rng:
                    _lib.check(lib.pf_ddim_step_rng_dev(xb.data_ptr(), e_t.data_ptr(), _lib.ptr(bf["orig"]), _lib.ptr(bf["orig_noise"]),
                                                        _lib.ptr(bf["mask"]), table.data_ptr(), st.data_ptr(), self.seed, off, xb.data_ptr(), n,
                                                        stream), "pf_ddim_step_rng_dev")
                else:
                    if noisy:
                        _lib.check(lib.pf_randn_dev(nz.data_ptr(), n, self.seed, st.data_ptr(), 0, off, stream), "pf_randn_dev")
                    _lib.check(lib.pf_ddim_step_dev(xb.data_ptr(), e_t.data_ptr(), _lib.ptr(nz), _lib.ptr(bf["orig"]), _lib.ptr(bf["orig_noise"]),
                                                    _lib.ptr(bf["mask"]), table.data_ptr(), st.data_ptr(), xb.data_ptr(), n, stream),
                               "pf_ddim_step_dev")
                _lib.check(lib.pf_step_end(st.data_ptr(), 1 if noisy else 0, stream), "pf_step_end")
            return body

        key = ("ddim", str(dev), float(uncond_scale), self.seed, off, bool(noisy)) + self._shape_key(**inputs)
        ent = self._graph_for(key, inputs, make_body, lambda: self._set_state(st, t_start), uncond_scale)
        g = ent["g"]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(t_start + 1):
            g.replay()
        e1.record()
        self.last_replay = (e0, e1, t_start + 1)
        if noisy:
            self._draws += t_start + 1
        return ent["bufs"]["x"].clone()   # the static buffer is overwritten by the next paint() with these shapes

    def _coef(self, index: int) -> _lib.DdimCoef:
        a, ap, sg = self.ddim_alpha[index], self.ddim_alpha_prev[index], self.ddim_sigma[index]
        return _lib.DdimCoef(float(self.ddim_sqrt_one_minus_alpha[index]), float(a ** 0.5), float(ap ** 0.5),
                             float((1.0 - ap - sg ** 2).sqrt()), float(sg), float(self.ddim_alpha_sqrt[index]),
                             float(self.ddim_sqrt_one_minus_alpha[index]))

    def _step(self, x, e_t, index, orig=None, orig_noise=None, mask=None, temperature=1.0, repeat_noise=False):
        coef = self._coef(index)
        noise = None
        # both kernel paths below read x / e_t / orig / orig_noise / mask as dense buffers
        x, e_t = x.contiguous(), e_t.contiguous()
        orig, orig_noise, mask = (None if v is None else v.contiguous() for v in (orig, orig_noise, mask))
        noisy = float(self.ddim_sigma[index]) != 0.0
        if noisy and not repeat_noise and temperature == 1.0 and self._rng_ok(x, e_t, orig, orig_noise, mask):
            # the step's draw happens inside the update kernel (same values as randn() + pf_ddim_step)
            draw = self._draws
            self._draws += 1
            if orig is not None and orig_noise is None:   # reference: a fresh q_sample noise per step, drawn after p_sample's
                orig_noise = self.randn(orig.shape, x.device)
            out = torch.empty_like(x)
            _lib.check(self._lib.pf_ddim_step_rng(x.data_ptr(), e_t.data_ptr(), _lib.ptr(orig), _lib.ptr(orig_noise), _lib.ptr(mask),
                                                  C.byref(coef), self.seed, draw, self.sample_offset * (x.numel() // x.shape[0]),
                                                  out.data_ptr(), x.numel(), _lib.current_stream()), "pf_ddim_step_rng")
            return out, coef
        if noisy:
            noise = self.randn((1, *x.shape[1:]) if repeat_noise else x.shape, x.device)
            if repeat_noise:
                noise = noise.expand_as(x).contiguous()
            if temperature != 1.0:
                noise = noise * temperature
        if orig is not None and orig_noise is None:
            # reference: q_sample(orig, index, noise=None) draws randn_like per step, after p_sample's draw (sampler_ddim.py:355-359)
            orig_noise = self.randn(orig.shape, x.device)
        out = torch.empty_like(x)
        _lib.check(self._lib.pf_ddim_step(x.data_ptr(), e_t.data_ptr(), _lib.ptr(noise), _lib.ptr(orig), _lib.ptr(orig_noise),
                                          _lib.ptr(mask), C.byref(coef), out.data_ptr(), x.numel(), _lib.current_stream()),
                   "pf_ddim_step")
        return out, coef

    @torch.no_grad()
    def get_x_prev_and_pred_x0(self, e_t, index: int, x, *, temperature: float = 1.0, repeat_noise: bool = False):
        x_prev, coef = self._step(x.contiguous(), e_t.contiguous(), index, temperature=temperature, repeat_noise=repeat_noise)
        pred_x0 = (x - coef.s1m * e_t) / coef.sqrt_a
        return x_prev, pred_x0

    @torch.no_grad()
    def p_sample(self, x, c, t, step: int, index: int, *, repeat_noise=False, temperature=1.0, uncond_scale=1.0,
                 uncond_cond=None, cond_concat=None, prep: Optional[dict] = None):
        e_t = self._eps(x, c, int(step), uncond_scale, uncond_cond, cond_concat, prep)
        x_prev, pred_x0 = self.get_x_prev_and_pred_x0(e_t, index, x, temperature=temperature, repeat_noise=repeat_noise)
        return x_prev, pred_x0, e_t

    @torch.no_grad()
    def q_sample(self, x0, index: int, noise=None):
        if noise is None:
            noise = self.randn(x0.shape, x0.device)
        out = torch.empty_like(x0)
        _lib.check(self._lib.pf_axpby(x0.contiguous().data_ptr(), noise.contiguous().data_ptr(),
                                      float(self.ddim_alpha_sqrt[index]), float(self.ddim_sqrt_one_minus_alpha[index]),
                                      out.data_ptr(), out.numel(), _lib.current_stream()), "pf_axpby")
        return out

    @torch.no_grad()
    def sample(self, shape, cond, repeat_noise=False, temperature=1.0, x_last=None, uncond_scale=1.0, uncond_cond=None,
               t_start: int = 0):
        x = x_last if x_last is not None else self.randn(shape, cond.device)
        time_steps = np.flip(self.time_steps)[t_start:]
        prep = self.prepare(cond, uncond_scale=uncond_scale, uncond_cond=uncond_cond)
        for i, step in enumerate(time_steps):
            index = len(time_steps) - i - 1
            e_t = self._eps(x, cond, int(step), uncond_scale, uncond_cond, None, prep)
            x, _ = self._step(x, e_t, index, temperature=temperature, repeat_noise=repeat_noise)
        return x

    @torch.no_grad()
    def paint(self, x, cond, t_start: int, *, orig=None, mask=None, orig_noise=None, uncond_scale: float = 1.0,
              uncond_cond=None, cond_concat=None, repaint_n: int = 1):
        x = x.contiguous()
        if orig is not None:
            assert mask is not None
            orig, mask = orig.contiguous().float(), mask.contiguous().float()
            orig_noise = None if orig_noise is None else orig_noise.contiguous().float()
        time_steps = np.flip(self.time_steps[: t_start + 1])
        if self.graph and self.noise_fn is None and (orig is None or orig_noise is not None) and self.on_step is None:
            nonzero = self.ddim_sigma[: t_start + 1] != 0
            if not bool(nonzero.any()) or bool(nonzero.all()):    # a mixed range would need a per-step decision on the host
                return self._paint_graph(x, cond, int(t_start), orig, mask, orig_noise, uncond_scale, uncond_cond, cond_concat,
                                         noisy=bool(nonzero.all()))
        prep = self.prepare(cond, uncond_scale=uncond_scale, uncond_cond=uncond_cond)
        for i, step in enumerate(time_steps):
            index = len(time_steps) - i - 1
            e_t = self._eps(x, cond, int(step), uncond_scale, uncond_cond, cond_concat, prep)
            # x_prev and the known-region blend (fixed orig_noise, sampler_ddim.py:355-359) in one kernel
            x, _ = self._step(x, e_t, index, orig=orig, orig_noise=orig_noise, mask=mask)
            if self.on_step is not None:
                self.on_step(int(step), x)
        return x
