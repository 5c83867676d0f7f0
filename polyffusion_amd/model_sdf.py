"""Conditioning wrapper and frozen encoders (host mirrors backed by libpfhip.so).

* ``ChordEncoder`` / ``TextureEncoder``: constructor argument order of the reference modules
  (``dl_modules/chord_enc.py:6``, ``dl_modules/txt_enc.py:6``); ``encode_mean`` returns what the
  reference reads from them, ``forward(x).mean``.
* ``Polyffusion_SDF``: ``models/model_sdf.py`` - ``_encode_chord`` (:92-106), ``_encode_txt``
  (:153-164) and ``load_trained`` (:59-84; legacy ``.pt`` with ``{"model": state_dict}``).
* ``load_pretrained_chd_enc`` / ``load_pretrained_txt_enc``: the key-prefix remaps of
  ``utils.py:48-86`` (``chord_enc.`` / ``rhy_encoder.``).
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace
from typing import Mapping, Optional

import numpy as np
import torch

from . import _lib
from .unet import LatentDiffusion


class _Encoder:
    KIND = -1

    def __init__(self, input_dim, emb_size, hidden_dim, z_dim, num_channel, device=None):
        self._lib = _lib.load()
        self.hidden_dim, self.z_dim = hidden_dim, z_dim
        h = C.c_void_p()
        _lib.check(self._lib.pf_encoder_create(self.KIND, input_dim, emb_size, hidden_dim, z_dim, num_channel, C.byref(h)),
                   "pf_encoder_create")
        self._h = h
        self.device = torch.device(device) if device is not None else (
            torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None)
        self._blob = None
        self._ws = None

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.pf_encoder_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def pack_state_dict(self, state: Mapping[str, object]) -> torch.Tensor:
        blob = torch.zeros(self._lib.pf_encoder_weight_bytes(self._h) // 4, dtype=torch.float32)
        for key, val in state.items():
            t = torch.as_tensor(np.asarray(val) if not isinstance(val, torch.Tensor) else val).detach().to("cpu", torch.float32).contiguous()
            shape = (C.c_int64 * max(1, t.dim()))(*t.shape)
            _lib.check(self._lib.pf_encoder_pack_param(self._h, key.encode(), t.data_ptr(), shape, t.dim(), blob.data_ptr()),
                       f"load_state_dict({key})")
        buf = C.create_string_buffer(256)
        if self._lib.pf_encoder_pack_missing(self._h, buf, 256):
            raise RuntimeError(f"load_state_dict: missing key {buf.value.decode()}")
        return blob

    def bind_packed(self, blob_dev: torch.Tensor):
        self._blob = blob_dev
        self.device = blob_dev.device
        _lib.check(self._lib.pf_encoder_bind_weights(self._h, blob_dev.data_ptr()))

    def load_state_dict(self, state: Mapping[str, object]):
        _lib.require_gpu()
        self.bind_packed(self.pack_state_dict(state).to(self.device))
        return self

    def _run(self, x: torch.Tensor, n_step: int) -> torch.Tensor:
        if self._blob is None:
            raise RuntimeError("encoder weights not loaded")
        x = x.contiguous().float()
        B = x.shape[0]
        nbytes = self._lib.pf_encoder_workspace_bytes(self._h, B)
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        mu = torch.empty(B, self.z_dim, dtype=torch.float32, device=x.device)
        _lib.check(self._lib.pf_encoder_forward(self._h, x.data_ptr(), B, n_step, mu.data_ptr(), self._ws.data_ptr(),
                                                self._ws.numel(), _lib.current_stream()), "pf_encoder_forward")
        return mu

    def forward(self, x):
        """Reference call shape: returns an object whose ``.mean`` is the encoder mean."""
        return SimpleNamespace(mean=self.encode_mean(x))

    __call__ = forward


class ChordEncoder(_Encoder):
    KIND = 0

    def __init__(self, input_dim, hidden_dim, z_dim, device=None):
        super().__init__(input_dim, 0, hidden_dim, z_dim, 0, device)

    def encode_mean(self, chord: torch.Tensor) -> torch.Tensor:  # [B,T,input_dim] -> [B,z]
        return self._run(chord, chord.shape[1])


class TextureEncoder(_Encoder):
    KIND = 1

    def __init__(self, emb_size, hidden_dim, z_dim, num_channel=10, device=None):
        super().__init__(0, emb_size, hidden_dim, z_dim, num_channel, device)

    def encode_mean(self, pr: torch.Tensor) -> torch.Tensor:  # [B,32,128] -> [B,z]
        assert tuple(pr.shape[1:]) == (32, 128), "texture encoder input must be [B,32,128]"
        return self._run(pr, 8)


class PianoTreeEncoder(_Encoder):
    """``dl_modules/pianotree_enc.py:7-59`` (constructor keywords kept); ``forward`` returns ``(dist, None, lengths)`` so that the
    reference's ``self.pnotree_enc(seg)[0].mean`` (models/model_sdf.py:144) reads the same."""
    KIND = 2

    def __init__(self, max_simu_note=20, max_pitch=127, min_pitch=0, pitch_sos=128, pitch_eos=129, pitch_pad=130, dur_pad=2, dur_width=5,
                 num_step=32, note_emb_size=128, enc_notes_hid_size=256, enc_time_hid_size=512, z_size=512, device=None):
        if num_step != 32 or max_simu_note > 32:
            raise ValueError("PianoTreeEncoder: num_step must be 32 and max_simu_note at most 32")
        self.max_simu_note, self.pitch_pad, self.num_step = max_simu_note, pitch_pad, num_step
        pitch_range = max_pitch - min_pitch + 3
        if pitch_pad != pitch_range:
            raise ValueError("PianoTreeEncoder: pitch_pad must be the index right after the pitch classes (max_pitch - min_pitch + 3)")
        super().__init__(pitch_range + dur_width, note_emb_size, enc_time_hid_size, z_size, enc_notes_hid_size, device)

    def encode_mean(self, grid: torch.Tensor) -> torch.Tensor:   # [R,32,max_simu_note,6] integer grid -> [R,z]
        assert tuple(grid.shape[1:]) == (self.num_step, self.max_simu_note, 6), "pianotree grid must be [B,32,max_simu_note,6]"
        return self._run(grid.to(torch.float32), self.max_simu_note)

    def forward(self, grid):
        lengths = self.max_simu_note - (grid[:, :, :, 0] == self.pitch_pad).sum(dim=-1)
        return SimpleNamespace(mean=self.encode_mean(grid)), None, lengths.cpu()

    __call__ = forward


def _strip(state, part: str):
    if isinstance(state, (str, bytes)) or hasattr(state, "__fspath__"):   # a checkpoint path, like the reference's fpath argument
        from .checkpoint import load_checkpoint
        state = load_checkpoint(str(state))[0]
    if "model" in state:
        state = state["model"]
    return {".".join(k.split(".")[1:]): v for k, v in state.items() if k.split(".")[0] == part}


def load_pretrained_chd_enc(state: Mapping[str, object], input_dim, hidden_dim, z_dim, device=None) -> ChordEncoder:
    """utils.py:48-69: keep the ``chord_enc.`` keys of a chd_8bar checkpoint."""
    return ChordEncoder(input_dim, hidden_dim, z_dim, device).load_state_dict(_strip(state, "chord_enc"))


def load_pretrained_txt_enc(state: Mapping[str, object], emb_size, hidden_dim, z_dim, num_channel, device=None) -> TextureEncoder:
    """utils.py:72-86: keep the ``rhy_encoder.`` keys of a Polydis checkpoint."""
    return TextureEncoder(emb_size, hidden_dim, z_dim, num_channel, device).load_state_dict(_strip(state, "rhy_encoder"))


class Polyffusion_SDF:
    def __init__(self, ldm: LatentDiffusion, cond_type, cond_mode="cond", chord_enc: Optional[ChordEncoder] = None,
                 chord_dec=None, pnotree_enc=None, pnotree_dec=None, txt_enc: Optional[TextureEncoder] = None,
                 concat_blurry=False, concat_ratio=1 / 8):
        if pnotree_dec is not None or chord_dec is not None:
            raise NotImplementedError("decoder modules (reconstruction / debugging output) are outside the denoising path (SURVEY.md 2 #12)")
        self.ldm, self.cond_type, self.cond_mode = ldm, cond_type, cond_mode
        self.chord_enc, self.txt_enc, self.pnotree_enc = chord_enc, txt_enc, pnotree_enc
        self.concat_blurry, self.concat_ratio = concat_blurry, concat_ratio

    @classmethod
    def load_trained(cls, ldm, chkpt_fpath, cond_type, cond_mode="cond", chord_enc=None, chord_dec=None,
                     pnotree_enc=None, pnotree_dec=None, txt_enc=None):
        """Legacy ``.pt`` (``{"model": state_dict}``, models/model_sdf.py:59-84) or Lightning ``.ckpt``
        (``model.``-prefixed ``state_dict``, inference_sdf.py:717-732) with ``ldm.eps_model.*``, ``chord_enc.*``,
        ``txt_enc.*`` keys and recomputable schedule vectors ``ldm.{alpha,beta,alpha_bar,sigma2}``."""
        from .checkpoint import load_checkpoint
        model = cls(ldm, cond_type, cond_mode, chord_enc, chord_dec, pnotree_enc, pnotree_dec, txt_enc)
        model.load_state_dict(load_checkpoint(chkpt_fpath)[0])
        return model

    def load_state_dict(self, state: Mapping[str, object]):
        from .checkpoint import split_state, split_state_full
        unet, ce, te = split_state(state)
        self.ldm.eps_model.load_state_dict(unet)
        pe = split_state_full(state)["pnotree_enc"]
        if self.pnotree_enc is not None and pe:
            self.pnotree_enc.load_state_dict(pe)
        if self.chord_enc is not None and ce:
            self.chord_enc.load_state_dict(ce)
        if self.txt_enc is not None and te:
            self.txt_enc.load_state_dict(te)
        return self

    def eval(self):
        return self

    def _encode_chord(self, chord: torch.Tensor) -> torch.Tensor:
        if self.chord_enc is not None:
            return self.chord_enc(chord).mean.unsqueeze(1)  # [B,1,512]
        return torch.reshape(chord, (-1, 1, chord.shape[1] * chord.shape[2]))

    def _encode_pnotree(self, pnotree: torch.Tensor) -> torch.Tensor:
        """models/model_sdf.py:138-151: the four 2-bar segments of every sample through the encoder, means concatenated: [B,1,4z]."""
        assert self.pnotree_enc is not None
        B, S = pnotree.shape[0], pnotree.shape[2]
        segs = pnotree.contiguous().view(B * 4, 32, S, 6)      # [B,128,S,6] -> [B*4,32,S,6]: a pure view, rows (b, segment)
        return self.pnotree_enc(segs)[0].mean.view(B, 1, -1)

    def _encode_txt(self, prmat: torch.Tensor) -> torch.Tensor:
        if self.txt_enc is None:
            return prmat
        B = prmat.shape[0]
        # the four 2-bar segments of every sample go through the encoder as ONE batch of 4B rows;
        # [B,128,128] -> [B*4,32,128] is a pure view and the means come back as [B, 4*z] = cat(dim=-1)
        segs = prmat.contiguous().view(B * 4, 32, prmat.shape[2])
        return self.txt_enc(segs).mean.view(B, 1, -1)
