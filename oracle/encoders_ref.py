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
