"""ORACLE (test infrastructure - NOT product code).

CPU restatement of the frozen condition encoders and of the conditioning
wrapper ``Polyffusion_SDF._encode_chord/_encode_txt``.  The GRU recurrence is
written out explicitly (torch gate order r, z, n) rather than calling nn.GRU, so
the HIP kernels are checked against the published recurrence, and this file is
in turn pinned against the real ``nn.GRU``-based reference modules through
``tests/golden/encoders.npz``.  Paths relative to ``/root/reference/polyffusion``.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

Tensors = Dict[str, torch.Tensor]


def gru_direction(x: torch.Tensor, w_ih, w_hh, b_ih, b_hh, reverse: bool) -> torch.Tensor:
    """One direction of a single-layer batch_first GRU; returns the final hidden state [B,H]."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    h = x.new_zeros(B, H)
    gi_all = F.linear(x, w_ih, b_ih)  # [B,T,3H]
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        gi = gi_all[:, t]
        gh = F.linear(h, w_hh, b_hh)
        i_r, i_z, i_n = gi.chunk(3, dim=-1)
        h_r, h_z, h_n = gh.chunk(3, dim=-1)
        r = torch.sigmoid(i_r + h_r)
        z = torch.sigmoid(i_z + h_z)
        n = torch.tanh(i_n + r * h_n)
        h = (1.0 - z) * n + z * h
    return h


def bigru_final(w: Tensors, p: str, x: torch.Tensor) -> torch.Tensor:
    """``gru(x)[-1]`` -> [2,B,H] -> transpose -> [B,2H] (chord_enc.py:16-18, txt_enc.py:29-31)."""
    hf = gru_direction(x, w[f"{p}weight_ih_l0"], w[f"{p}weight_hh_l0"], w[f"{p}bias_ih_l0"], w[f"{p}bias_hh_l0"], False)
    hb = gru_direction(x, w[f"{p}weight_ih_l0_reverse"], w[f"{p}weight_hh_l0_reverse"],
                       w[f"{p}bias_ih_l0_reverse"], w[f"{p}bias_hh_l0_reverse"], True)
    return torch.cat([hf, hb], dim=-1)


def chord_encoder_mean(w: Tensors, chord: torch.Tensor) -> torch.Tensor:
    """dl_modules/chord_enc.py:15-22 - only ``Normal(mu, .).mean`` is consumed downstream."""
    h = bigru_final(w, "gru.", chord)
    return F.linear(h, w["linear_mu.weight"], w["linear_mu.bias"])


def texture_encoder_mean(w: Tensors, pr: torch.Tensor) -> torch.Tensor:
    """dl_modules/txt_enc.py:23-35 - conv(4x12,s(4,1))+ReLU+maxpool(1,4); the un-permuted
    ``.view(bs, 8, -1)`` of a [bs,10,8,29] tensor; fc2(fc1(.)) without activation; bi-GRU; mu."""
    bs = pr.shape[0]
    y = F.conv2d(pr.unsqueeze(1), w["cnn.0.weight"], w["cnn.0.bias"], stride=(4, 1))
    y = F.max_pool2d(F.relu(y), kernel_size=(1, 4), stride=(1, 4))
    y = y.reshape(bs, 8, -1)
    y = F.linear(F.linear(y, w["fc1.weight"], w["fc1.bias"]), w["fc2.weight"], w["fc2.bias"])
    h = bigru_final(w, "gru.", y)
    return F.linear(h, w["linear_mu.weight"], w["linear_mu.bias"])


def encode_chord(w: Tensors, chord: torch.Tensor) -> torch.Tensor:
    """models/model_sdf.py:92-106 - [B,32,36] -> [B,1,512]."""
    return chord_encoder_mean(w, chord).unsqueeze(1)


def encode_txt(w: Tensors, prmat: torch.Tensor) -> torch.Tensor:
    """models/model_sdf.py:153-164 - four 2-bar segments, means concatenated -> [B,1,1024]."""
    zs = [texture_encoder_mean(w, seg) for seg in prmat.split(32, 1)]
    return torch.cat(zs, dim=-1).unsqueeze(1)


def gru_direction_packed(x: torch.Tensor, lengths: torch.Tensor, w_ih, w_hh, b_ih, b_hh, reverse: bool) -> torch.Tensor:
    """Final hidden state of one GRU direction over variable-length sequences, i.e. what ``nn.GRU`` returns for a
    ``pack_padded_sequence`` input: sequence b is advanced through its first ``lengths[b]`` elements only (in reverse order for the
    backward direction)."""
    N, S, _ = x.shape
    H = w_hh.shape[1]
    h = x.new_zeros(N, H)
    gi_all = F.linear(x, w_ih, b_ih)
    for step in range(S):
        live = step < lengths
        t = torch.where(live, (lengths - 1 - step) if reverse else torch.full_like(lengths, step), torch.zeros_like(lengths))
        gi = gi_all[torch.arange(N), t]
        gh = F.linear(h, w_hh, b_hh)
        i_r, i_z, i_n = gi.chunk(3, dim=-1)
        h_r, h_z, h_n = gh.chunk(3, dim=-1)
        r = torch.sigmoid(i_r + h_r)
        z = torch.sigmoid(i_z + h_z)
        n = torch.tanh(i_n + r * h_n)
        h = torch.where(live.unsqueeze(1), (1.0 - z) * n + z * h, h)
    return h


def pianotree_encoder_mean(w: Tensors, grid: torch.Tensor, pitch_pad: int = 130) -> torch.Tensor:
    """dl_modules/pianotree_enc.py:69-121 on an index grid [B,32,S,6] (int64): lengths = S - pads; multi-hot rows (one-hot pitch
    over 130 classes - the pad index has no column - + the 5 duration digits as floats, pad digit 2 included); note embedding;
    bi-GRU over each step's notes (packed); bi-GRU over the 32 steps; ``linear_mu``."""
    B, T, S, _ = grid.shape
    P = w["note_embedding.weight"].shape[1] - 5
    lengths = S - (grid[..., 0] == pitch_pad).sum(-1).reshape(-1)
    onehot = F.one_hot(grid[..., 0], P + 1)[..., :P].float()
    x = torch.cat([onehot, grid[..., 1:].float()], dim=-1)
    emb = F.linear(x, w["note_embedding.weight"], w["note_embedding.bias"]).reshape(B * T, S, -1)
    g = "enc_notes_gru."
    hf = gru_direction_packed(emb, lengths, w[g + "weight_ih_l0"], w[g + "weight_hh_l0"], w[g + "bias_ih_l0"], w[g + "bias_hh_l0"], False)
    hb = gru_direction_packed(emb, lengths, w[g + "weight_ih_l0_reverse"], w[g + "weight_hh_l0_reverse"], w[g + "bias_ih_l0_reverse"],
                              w[g + "bias_hh_l0_reverse"], True)
    steps = torch.cat([hf, hb], dim=-1).reshape(B, T, -1)
    h = bigru_final(w, "enc_time_gru.", steps)
    return F.linear(h, w["linear_mu.weight"], w["linear_mu.bias"])


def encode_pnotree(w: Tensors, pnotree: torch.Tensor) -> torch.Tensor:
    """models/model_sdf.py:138-151 - four 2-bar segments [B,32,S,6], means concatenated -> [B,1,2048]."""
    zs = [pianotree_encoder_mean(w, seg) for seg in pnotree.split(32, 1)]
    return torch.cat(zs, dim=-1).unsqueeze(1)
