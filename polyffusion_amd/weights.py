"""Deterministic synthetic weights in the reference checkpoint key namespace.

No trained checkpoint ships with the reference, so parity and
benchmarks run on synthetic weights.  Every tensor is drawn from its own
PCG64 stream keyed by ``seed`` and ``crc32(key)``, so the *same* weights are
regenerated bit-for-bit in the build container (where they are loaded into
the imported reference to make golden vectors) and on the GPU box.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np

from .arch import (
    UNetConfig,
    chord_encoder_param_shapes,
    pianotree_encoder_param_shapes,
    texture_encoder_param_shapes,
    unet_param_shapes,
)


def _draw(key: str, shape: Tuple[int, ...], seed: int) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(key.encode())]))
    if len(shape) == 1:
        if key.endswith(".weight"):  # normalisation gain
            return (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
        return (0.02 * rng.standard_normal(shape)).astype(np.float32)
    fan_in = int(np.prod(shape[1:]))
    std = (1.0 / fan_in) ** 0.5
    return (std * rng.standard_normal(shape)).astype(np.float32)


def synth_tensors(shapes: Dict[str, Tuple[int, ...]], seed: int = 0, prefix: str = "") -> "OrderedDict[str, np.ndarray]":
    return OrderedDict((prefix + k, _draw(prefix + k, s, seed)) for k, s in shapes.items())


def synth_unet_state(cfg: UNetConfig, seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """UNet tensors keyed relative to ``eps_model`` (e.g. ``input_blocks.0.0.weight``)."""
    return synth_tensors(unet_param_shapes(cfg), seed)


def synth_chord_encoder_state(seed: int = 0, input_dim=36, hidden_dim=512, z_dim=512):
    # GRU weights use U(-1/sqrt(H), 1/sqrt(H))-like scale so the recurrence stays bounded
    shapes = chord_encoder_param_shapes(input_dim, hidden_dim, z_dim)
    return _synth_rnn(shapes, seed, "chord_enc.", hidden_dim)


def synth_texture_encoder_state(seed: int = 0, emb_size=256, hidden_dim=1024, z_dim=256, num_channel=10):
    shapes = texture_encoder_param_shapes(emb_size, hidden_dim, z_dim, num_channel)
    return _synth_rnn(shapes, seed, "txt_enc.", hidden_dim)


def _synth_rnn(shapes, seed, prefix, hidden_dim):
    out = OrderedDict()
    for k, s in shapes.items():
        rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32((prefix + k).encode())]))
        if k.startswith("gru."):
            b = 1.0 / hidden_dim ** 0.5
            out[k] = rng.uniform(-b, b, size=s).astype(np.float32)
        else:
            out[k] = _draw(prefix + k, s, seed)
    return out


def synth_pianotree_encoder_state(seed: int = 0, note_size=135, note_emb_size=128, enc_notes_hid_size=256, enc_time_hid_size=512, z_size=512):
    """PianoTreeEncoder tensors (dl_modules/pianotree_enc.py); GRU weights at torch's U(-1/sqrt(H), 1/sqrt(H)) scale."""
    shapes = pianotree_encoder_param_shapes(note_size, note_emb_size, enc_notes_hid_size, enc_time_hid_size, z_size)
    out = OrderedDict()
    for k, s in shapes.items():
        if "_gru." in k:
            hid = enc_notes_hid_size if k.startswith("enc_notes") else enc_time_hid_size
            rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(("pnotree_enc." + k).encode())]))
            out[k] = rng.uniform(-1.0 / hid ** 0.5, 1.0 / hid ** 0.5, size=s).astype(np.float32)
        else:
            out[k] = _draw("pnotree_enc." + k, s, seed)
    return out
