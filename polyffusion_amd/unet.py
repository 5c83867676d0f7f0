"""Host mirror of the reference denoiser objects, backed by libpfhip.so.

``UNetModel`` keeps the reference constructor surface (keyword names of
``stable_diffusion/model/unet.py:35-47``), ingests weights by the reference
``state_dict`` key names and evaluates ``forward(x, t, cond)`` with the same
argument meaning (``unet.py:171-196``) - but every FLOP runs in the HIP kernels of
``csrc/``.  ``LatentDiffusion`` mirrors ``stable_diffusion/latent_diffusion.py``
(schedule at :90-103, ``forward`` at :138-147) for the attributes the samplers use.

PyTorch is plumbing here: device buffers, the current stream and (for multi-GPU)
the RCCL broadcast of the packed weight blob.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, List, Mapping, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .arch import UNetConfig


class UNetModel:
    def __init__(self, *, in_channels: int, out_channels: int, channels: int, n_res_blocks: int,
                 attention_levels: Iterable[int], channel_multipliers: Iterable[int], n_heads: int,
                 tf_layers: int = 1, d_cond: int = 768, img_h: int = 128, img_w: int = 128,
                 device: Optional[torch.device] = None, x3: Optional[str] = None):
        """`x3`: which build of the library this model lives in - None: the process default (libpfhip.so, bf16 pieces, unless PF_X3
        says otherwise); "f16": libpfhip_f16.so, whose split mode is "f16x3" (include/pfhip.h pf_x3_element).  Weights are packed by, and
        for, that library; a model never changes library."""
        self.cfg = UNetConfig(in_channels, out_channels, channels, n_res_blocks, tuple(attention_levels),
                              tuple(channel_multipliers), n_heads, tf_layers, d_cond)
        self.img_h, self.img_w = int(img_h), int(img_w)
        self.channels = channels
        self.x3 = x3
        self._lib = _lib.load(x3)
        # name of the split mode: "f16x3" for a model asked to live in the fp16 build; a model in the process default keeps "bf16x3"
        # (also under PF_X3=f16, which exists to run unchanged callers against the other build)
        self._split_name = "f16x3" if x3 == "f16" else "bf16x3"
        c = _lib.UNetCfg()
        c.in_channels, c.out_channels, c.channels, c.n_res_blocks = in_channels, out_channels, channels, n_res_blocks
        c.n_attention_levels = len(self.cfg.attention_levels)
        for i, v in enumerate(self.cfg.attention_levels):
            c.attention_levels[i] = v
        c.n_levels = len(self.cfg.channel_multipliers)
        for i, v in enumerate(self.cfg.channel_multipliers):
            c.channel_multipliers[i] = v
        c.n_heads, c.tf_layers, c.d_cond, c.img_h, c.img_w = n_heads, tf_layers, d_cond, self.img_h, self.img_w
        h = C.c_void_p()
        self._check(self._lib.pf_unet_create(C.byref(c), C.byref(h)), "pf_unet_create")
        self._h = h
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device()) \
            if torch.cuda.is_available() else None
        self._blob_host: Optional[torch.Tensor] = None
        self._blob_dev: Optional[torch.Tensor] = None
        self._ws: Optional[torch.Tensor] = None
        self._amax: Optional[torch.Tensor] = None      # range telemetry word (track_absmax)
        self._ws_key = (0, 0, 0)

    def _check(self, rc: int, what: str = "") -> int:
        return _lib.check(rc, what, self._lib)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.pf_unet_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- weights --------------------------------------------------------------------------------
    def param_shapes(self) -> "Dict[str, Tuple[int, ...]]":
        out = {}
        buf = C.create_string_buffer(256)
        shape = (C.c_int64 * 4)()
        nd = C.c_int()
        for i in range(self._lib.pf_unet_n_params(self._h)):
            self._check(self._lib.pf_unet_param_info(self._h, i, buf, 256, shape, C.byref(nd)))
            out[buf.value.decode()] = tuple(int(shape[d]) for d in range(nd.value))
        return out

    def pack_state_dict(self, state: Mapping[str, object], strict: bool = True) -> torch.Tensor:
        """Repack reference-named tensors into the kernel-friendly host blob (no GPU needed)."""
        nbytes = self._lib.pf_unet_weight_bytes(self._h)
        blob = torch.zeros(nbytes // 4, dtype=torch.float32)
        for key, val in state.items():
            t = torch.as_tensor(np.asarray(val) if not isinstance(val, torch.Tensor) else val).detach().to("cpu", torch.float32).contiguous()
            shape = (C.c_int64 * max(1, t.dim()))(*t.shape)
            rc = self._lib.pf_unet_pack_param(self._h, key.encode(), t.data_ptr(), shape, t.dim(), blob.data_ptr())
            if rc == -2 and not strict:
                continue
            self._check(rc, f"load_state_dict({key})")
        buf = C.create_string_buffer(256)
        missing = self._lib.pf_unet_pack_missing(self._h, buf, 256)
        if missing:
            raise RuntimeError(f"load_state_dict: {missing} missing key(s), first: {buf.value.decode()}")
        self._blob_host = blob
        return blob

    def weight_bytes(self) -> int:
        return int(self._lib.pf_unet_weight_bytes(self._h))

    def bind_packed(self, blob_dev: torch.Tensor):
        """Attach a packed blob that already lives on the GPU (e.g. received by RCCL broadcast)."""
        assert blob_dev.is_cuda and blob_dev.dtype == torch.float32 and blob_dev.numel() * 4 == self.weight_bytes()
        self._blob_dev = blob_dev
        self.device = blob_dev.device
        self._check(self._lib.pf_unet_bind_weights(self._h, blob_dev.data_ptr()), "pf_unet_bind_weights")

    def load_state_dict(self, state: Mapping[str, object], strict: bool = True):
        """Reference-compatible weight ingestion (keys relative to ``eps_model``)."""
        _lib.require_gpu()
        blob = self.pack_state_dict(state, strict)
        self.bind_packed(blob.to(self.device))
        return self

    def eval(self):
        return self

    # ---- forward --------------------------------------------------------------------------------
    def workspace(self, batch: int, n_cond: int, shared_x: bool = False) -> torch.Tensor:
        # tile choice (and buffer sizes) depend on the arithmetic mode and on the plan options; the shared-prefix guidance plan has a layout of its own
        key = (batch, n_cond, shared_x, self._lib.pf_unet_get_precision(self._h)) + self.options_key()
        if self._ws is None or self._ws_key != key:
            size = self._lib.pf_unet_workspace_bytes_cfg if shared_x else self._lib.pf_unet_workspace_bytes
            nbytes = size(self._h, batch, n_cond)
            if self._ws is None or self._ws.numel() < nbytes:
                self._ws = None
                self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._ws_key = key
        return self._ws

    # ---- step-invariant prefix (hoisted out of a sampler's loop; include/pfhip.h pf_unet_prepared) ---------------
    def prepare_time(self, n_rows: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[n_rows, W] table: row r = the additive time biases of every ResBlock for time-step VALUE r - what ``forward`` derives
        from ``t == r`` through ``time_embed`` and the ``emb_layers`` (unet.py:181-182, 286-289).  Depends on the weights only."""
        if self._blob_dev is None:
            raise RuntimeError("UNetModel.prepare_time: weights not loaded")
        w = int(self._lib.pf_unet_time_bias_width(self._h))
        if out is None:
            out = torch.empty(n_rows, w, dtype=torch.float32, device=self.device)
        assert out.shape == (n_rows, w) and out.is_contiguous()
        scratch = torch.empty(n_rows * 4 * self.cfg.channels, dtype=torch.float32, device=self.device)
        self._check(self._lib.pf_unet_prepare_time(self._h, n_rows, out.data_ptr(), scratch.data_ptr(), scratch.numel() * 4,
                                                  _lib.current_stream()), "pf_unet_prepare_time")
        return out

    def prepare_cond(self, cond: torch.Tensor, out: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        """[B, W] collapsed cross-attention biases ``to_out(to_v(cond))`` of every transformer block for ONE context token
        (unet_attention.py:186-212 with softmax over a single key); ``None`` when ``cond`` has several tokens (general path)."""
        if self._blob_dev is None:
            raise RuntimeError("UNetModel.prepare_cond: weights not loaded")
        w = int(self._lib.pf_unet_cross_bias_width(self._h))
        if cond.dim() != 3 or cond.shape[1] != 1 or w == 0:
            return None
        if cond.shape[2] != self.cfg.d_cond:
            raise RuntimeError(f"UNetModel.prepare_cond: cond has shape {tuple(cond.shape)}, expected [B,1,{self.cfg.d_cond}]")
        cond = cond.contiguous().float()
        B = cond.shape[0]
        if out is None:
            out = torch.empty(B, w, dtype=torch.float32, device=cond.device)
        assert out.shape == (B, w) and out.is_contiguous()
        scratch = torch.empty(B * w, dtype=torch.float32, device=cond.device)
        self._check(self._lib.pf_unet_prepare_cond(self._h, cond.data_ptr(), B, out.data_ptr(), scratch.data_ptr(), scratch.numel() * 4,
                                                  _lib.current_stream()), "pf_unet_prepare_cond")
        return out

    def forward(self, x: torch.Tensor, time_steps: torch.Tensor, cond: torch.Tensor, out: Optional[torch.Tensor] = None, *,
                time_table: Optional[torch.Tensor] = None, cross_bias: Optional[torch.Tensor] = None, check_t: bool = False,
                shared_x: bool = False):
        """``time_table`` / ``cross_bias``: results of ``prepare_time`` / ``prepare_cond(cond)`` - the caller vouches that they
        belong to these weights and this ``cond``; either may be ``None`` (computed here).  Bit-identical either way.
        ``shared_x=True`` - classifier-free guidance (``sampler/__init__.py:69-74``): ``x`` holds B samples while ``time_steps`` and
        ``cond`` hold the 2B rows of ``cat([t, t])`` / ``cat([uncond_cond, cond])``; the result equals ``forward(cat([x, x]), ...)`` up to
        tile-choice rounding, but everything in front of the first transformer block - where the two halves cannot differ - is computed
        once (``pf_unet_forward_cfg``).  PRECONDITION: ``time_steps[:B] == time_steps[B:]`` (the shared prefix uses the first half's rows;
        verified under ``check_t``).
        The prepared path is valid for ``0 <= t < time_table.shape[0]`` only; ``check_t=True`` verifies that (one device->host
        sync - the samplers, whose t values are rows of a host-built table, do not ask for it)."""
        if self._blob_dev is None:
            raise RuntimeError("UNetModel.forward: weights not loaded")
        B = x.shape[0] * (2 if shared_x else 1)
        if tuple(x.shape[1:]) != (self.cfg.in_channels, self.img_h, self.img_w):
            raise RuntimeError(f"UNetModel.forward: x has shape {tuple(x.shape)}, expected [B,{self.cfg.in_channels},{self.img_h},{self.img_w}]")
        if cond.dim() != 3 or cond.shape[0] != B or cond.shape[2] != self.cfg.d_cond:
            raise RuntimeError(f"UNetModel.forward: cond has shape {tuple(cond.shape)}, expected [{'2B' if shared_x else 'B'},n_cond,{self.cfg.d_cond}]")
        if time_steps.shape[0] != B:
            raise RuntimeError(f"UNetModel.forward: {time_steps.shape[0]} time steps for {B} condition rows")
        x = x.contiguous().float()
        cond = cond.contiguous().float()
        t = time_steps.to(torch.int64).contiguous()
        if shared_x and check_t and not torch.equal(t[: B // 2], t[B // 2:]):
            # the shared prefix is evaluated once, with the first half's time rows: the two halves of a guidance evaluation carry the same t
            raise RuntimeError("UNetModel.forward(shared_x=True): the two halves of time_steps must be equal "
                               "(the conditional and unconditional evaluations share everything in front of the first transformer block)")
        n_cond = cond.shape[1]
        ws = self.workspace(B, n_cond, shared_x)
        if out is None:
            out = torch.empty(B, self.cfg.out_channels, self.img_h, self.img_w, dtype=torch.float32, device=x.device)
        prep = None
        if time_table is not None or cross_bias is not None:
            prep = _lib.UNetPrepared()
            if time_table is not None:
                assert time_table.dtype == torch.float32 and time_table.is_contiguous() and time_table.device == x.device
                # the kernels clamp a row index into the table for memory safety only: a t outside it would silently read the
                # wrong row.  Callers that hold t on the host say so (check_t); the samplers' loops index a [n_steps+1]-row table
                # with values < n_steps + 1 by construction.
                if check_t:
                    lo, hi = int(t.min()), int(t.max())
                    if lo < 0 or hi >= time_table.shape[0]:
                        raise RuntimeError(f"UNetModel.forward: time step {lo if lo < 0 else hi} outside the prepared table "
                                           f"({time_table.shape[0]} rows)")
                prep.time_table, prep.n_time_rows = time_table.data_ptr(), time_table.shape[0]
            if cross_bias is not None:
                assert n_cond == 1 and cross_bias.shape[0] == B and cross_bias.is_contiguous() and cross_bias.device == x.device
                prep.cross_bias = cross_bias.data_ptr()
        fn = self._lib.pf_unet_forward_cfg if shared_x else self._lib.pf_unet_forward_prepared
        self._check(fn(self._h, x.data_ptr(), t.data_ptr(), cond.data_ptr(), B, n_cond, None if prep is None else C.byref(prep), out.data_ptr(),
                      ws.data_ptr(), ws.numel(), _lib.current_stream()), "pf_unet_forward")
        return out

    __call__ = forward

    # ---- arithmetic mode ------------------------------------------------------------------------
    def set_precision(self, mode: str):
        """"f32" (exact fp32 MFMA) or this model's split mode: "bf16x3" (error-compensated split on the bf16 matrix pipe), or "f16x3"
        (the same on the fp16 pipe) for a model constructed with x3="f16"."""
        if mode not in ("f32", self._split_name):
            raise ValueError(f"precision {mode!r}: this model supports 'f32' and {self._split_name!r} "
                             "(the split element type is chosen at construction: UNetModel(..., x3='f16') for f16x3)")
        code = 0 if mode == "f32" else 1
        self._check(self._lib.pf_unet_set_precision(self._h, code), "pf_unet_set_precision")
        return self

    @property
    def split_mode(self) -> str:
        """Name of this model's split-precision mode: "bf16x3", or "f16x3" for a model constructed with x3="f16"."""
        return self._split_name

    @property
    def precision(self) -> str:
        return ["f32", self._split_name][self._lib.pf_unet_get_precision(self._h)]

    # ---- plan options (which of two equivalent kernel forms the plan launches; include/pfhip.h PF_OPT_*) --------
    _OPTS = {"mlp_fused": _lib.OPT_MLP_FUSED, "attn_wide": _lib.OPT_ATTN_WIDE, "conv_t16": _lib.OPT_CONV_T16, "conv_pp": _lib.OPT_CONV_PP, "conv_wino": _lib.OPT_CONV_WINO}

    def set_option(self, name: str, value: Optional[bool]):
        """``None`` = automatic (the default), ``False`` / ``True`` = never / always (where the form exists)."""
        v = _lib.OPT_AUTO if value is None else int(bool(value))
        self._check(self._lib.pf_unet_set_option(self._h, self._OPTS[name], v), "pf_unet_set_option")
        return self

    def get_option(self, name: str) -> Optional[bool]:
        v = self._lib.pf_unet_get_option(self._h, self._OPTS[name])
        return None if v < 0 else bool(v)

    def options_key(self) -> tuple:
        """Every plan option's current setting: part of the key of anything that bakes the launch sequence in (workspace, graphs)."""
        return tuple(self._lib.pf_unet_get_option(self._h, o) for o in range(_lib.OPT_COUNT)) + (self._amax is not None,)

    # ---- range telemetry (include/pfhip.h pf_unet_track_absmax) -----------------------------------
    def track_absmax(self, on: bool = True):
        """While on, every forward max-combines the largest |value| its layers store into one device word (reset here)."""
        dev = self._blob_dev.device if self._blob_dev is not None else torch.device("cuda")
        self._amax = torch.zeros(1, dtype=torch.int32, device=dev) if on else None
        self._check(self._lib.pf_unet_track_absmax(self._h, None if self._amax is None else self._amax.data_ptr()), "pf_unet_track_absmax")
        self._ws = None          # the plan changes (fused launches run as their chains while tracking)
        return self

    def read_absmax(self) -> float:
        """Largest |stored value| since track_absmax(True) (nan / inf if one was stored); 65504 / it = the fp16 split's headroom."""
        if self._amax is None:
            raise RuntimeError("read_absmax: telemetry is off (track_absmax(True) first)")
        return float(self._amax.view(torch.float32).item())

    # ---- profiling ------------------------------------------------------------------------------
    def set_profiling(self, on: bool):
        self._check(self._lib.pf_unet_set_profiling(self._h, int(on)))

    def read_profile(self) -> List[Tuple[int, float, float]]:
        cap = 4096
        kind = (C.c_int * cap)()
        ms = (C.c_float * cap)()
        fl = (C.c_double * cap)()
        n = self._check(self._lib.pf_unet_profile_read(self._h, kind, ms, fl, cap))
        return [(kind[i], ms[i], fl[i]) for i in range(n)]

    def read_profile_direct(self) -> List[float]:
        """Per launch of the last profiled forward: the operation count of the layer's DIRECT form (= read_profile()'s count except for
        Winograd launches, which execute 16 / 36 of it)."""
        cap = 4096
        fl = (C.c_double * cap)()
        n = self._check(self._lib.pf_unet_profile_read_direct(self._h, fl, cap))
        return [fl[i] for i in range(n)]

    def n_launches(self, batch: int, n_cond: int = 1, prepared: bool = False, shared_x: bool = False) -> int:
        if shared_x:     # batch = the 2B rows of a guidance evaluation
            return int(self._lib.pf_unet_n_launches_cfg(self._h, batch, n_cond, int(prepared), int(prepared)))
        if prepared:
            return int(self._lib.pf_unet_n_launches_prepared(self._h, batch, n_cond, 1, 1))
        return int(self._lib.pf_unet_n_launches(self._h, batch, n_cond))


class LatentDiffusion:
    """Mirror of ``stable_diffusion/latent_diffusion.py:LatentDiffusion`` without an autoencoder
    (``inference_sdf.py:537`` passes ``autoencoder=None``)."""

    def __init__(self, unet_model: UNetModel, autoencoder=None, latent_scaling_factor: float = 0.18215,
                 n_steps: int = 1000, linear_start: float = 0.00085, linear_end: float = 0.012):
        if autoencoder is not None:
            raise NotImplementedError("the Polyffusion path runs without a latent autoencoder")
        self.eps_model = unet_model
        self.first_stage_model = None
        self.latent_scaling_factor = latent_scaling_factor
        self.n_steps = n_steps
        # sqrt-linear schedule computed in float64 then cast (latent_diffusion.py:90-103)
        beta = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_steps, dtype=torch.float64) ** 2
        alpha = 1.0 - beta
        alpha_bar = torch.cumprod(alpha, dim=0)
        self.alpha = alpha.to(torch.float32)
        self.beta = beta.to(torch.float32)
        self.alpha_bar = alpha_bar.to(torch.float32)
        self.sigma2 = self.beta

    @property
    def device(self):
        return self.eps_model.device

    def eval(self):
        return self

    supports_shared_x = True     # forward(..., shared_x=True): the guidance evaluation with its condition-independent prefix computed once

    def forward(self, x: torch.Tensor, t: torch.Tensor, context: torch.Tensor, **prepared):
        return self.eps_model(x, t, context, **prepared)

    __call__ = forward
