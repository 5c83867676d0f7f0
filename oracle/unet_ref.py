"""ORACLE (test infrastructure - NOT product code).

CPU restatement of the reference denoiser forward, written functionally on a
plain ``{key: tensor}`` dict that uses the reference state_dict key names.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; the product path (``polyffusion_amd``) never does.

Pinning: ``tools/make_goldens.py`` imports the real reference from
``/root/reference`` (build container only), loads the same synthetic weights
and writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this
restatement against those vectors (<= 1e-5 abs, fp32 reorder noise).

Every function cites the reference lines it restates (paths relative to
``/root/reference/polyffusion``).  Arithmetic is fp32 via the same ATen ops
the reference dispatches to (conv2d, group_norm, layer_norm, softmax, gelu-erf,
nearest interpolate), so it doubles as the "reference CPU path" timed by
``bench.py``.
"""
from __future__ import annotations

import math
from typing import Dict, List, NamedTuple, Optional, Tuple

import torch
import torch.nn.functional as F

Tensors = Dict[str, torch.Tensor]


class RefConfig(NamedTuple):
    """The reference ``UNetModel`` constructor arguments (unet.py:35-47).  The oracle is self-contained:
    any object with these attributes is accepted (the tests pass the product's ``UNetConfig``), nothing
    is imported from the product package."""
    in_channels: int = 2
    out_channels: int = 2
    channels: int = 64
    n_res_blocks: int = 2
    attention_levels: Tuple[int, ...] = (2, 3)
    channel_multipliers: Tuple[int, ...] = (1, 2, 4, 4)
    n_heads: int = 4
    tf_layers: int = 1
    d_cond: int = 512


def block_lists(cfg):
    """unet.py:70-149 - the constructor's module lists, restated independently of the product's plan
    builder: returns (input_blocks, middle_block, output_blocks), each block a list of layer kinds
    ("conv3" | "res" | "st" | "down" | "up")."""
    n_levels = len(cfg.channel_multipliers)
    inputs = [["conv3"]]
    for lvl in range(n_levels):
        with_attn = lvl in cfg.attention_levels
        inputs += [["res", "st"] if with_attn else ["res"] for _ in range(cfg.n_res_blocks)]
        if lvl < n_levels - 1:
            inputs.append(["down"])
    outputs = []
    for lvl in range(n_levels - 1, -1, -1):
        for j in range(cfg.n_res_blocks + 1):
            blk = ["res"] + (["st"] if lvl in cfg.attention_levels else [])
            if lvl > 0 and j == cfg.n_res_blocks:
                blk.append("up")
            outputs.append(blk)
    return inputs, ["res", "st", "res"], outputs


def time_step_embedding(t: torch.Tensor, channels: int, max_period: float = 10000.0) -> torch.Tensor:
    """stable_diffusion/model/unet.py:151-169 - [cos | sin], half = channels//2."""
    half = channels // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None].to(t.device)
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def time_embed(w: Tensors, t: torch.Tensor, channels: int) -> torch.Tensor:
    """unet.py:63-68,181-182 - Linear, SiLU, Linear on the sinusoid."""
    e = time_step_embedding(t, channels).to(w["time_embed.0.weight"].dtype)   # fp32 like the reference; float64 weights = the truth runs of the stress tests
    e = F.linear(e, w["time_embed.0.weight"], w["time_embed.0.bias"])
    return F.linear(F.silu(e), w["time_embed.2.weight"], w["time_embed.2.bias"])


def res_block(w: Tensors, p: str, x: torch.Tensor, t_emb: torch.Tensor) -> torch.Tensor:
    """unet.py:262-318 - GN32/SiLU/conv3, additive time bias, GN32/SiLU/conv3, skip."""
    h = F.group_norm(x, 32, w[f"{p}.in_layers.0.weight"], w[f"{p}.in_layers.0.bias"], eps=1e-5)
    h = F.conv2d(F.silu(h), w[f"{p}.in_layers.2.weight"], w[f"{p}.in_layers.2.bias"], padding=1)
    tb = F.linear(F.silu(t_emb), w[f"{p}.emb_layers.1.weight"], w[f"{p}.emb_layers.1.bias"])
    h = h + tb[:, :, None, None]
    h = F.group_norm(h, 32, w[f"{p}.out_layers.0.weight"], w[f"{p}.out_layers.0.bias"], eps=1e-5)
    h = F.conv2d(F.silu(h), w[f"{p}.out_layers.3.weight"], w[f"{p}.out_layers.3.bias"], padding=1)
    if f"{p}.skip_connection.weight" in w:
        x = F.conv2d(x, w[f"{p}.skip_connection.weight"], w[f"{p}.skip_connection.bias"])
    return x + h


def attention(w: Tensors, p: str, x: torch.Tensor, cond: Optional[torch.Tensor], n_heads: int) -> torch.Tensor:
    """unet_attention.py:186-212,261-293 - q/k/v without bias, scale d_head^-0.5 after the
    dot product, softmax over keys, to_out with bias.  (The reference's in-place
    half-batch softmax is numerically the plain softmax.)"""
    src = x if cond is None else cond
    q = F.linear(x, w[f"{p}.to_q.weight"])
    k = F.linear(src, w[f"{p}.to_k.weight"])
    v = F.linear(src, w[f"{p}.to_v.weight"])
    b, lq, d = q.shape
    dh = d // n_heads
    q = q.view(b, lq, n_heads, dh)
    k = k.view(b, k.shape[1], n_heads, dh)
    v = v.view(b, v.shape[1], n_heads, dh)
    att = torch.einsum("bihd,bjhd->bhij", q, k) * (dh ** -0.5)
    att = att.softmax(dim=-1)
    out = torch.einsum("bhij,bjhd->bihd", att, v).reshape(b, lq, d)
    return F.linear(out, w[f"{p}.to_out.0.weight"], w[f"{p}.to_out.0.bias"])


def feed_forward(w: Tensors, p: str, x: torch.Tensor) -> torch.Tensor:
    """unet_attention.py:296-333 - GeGLU (value, gate = chunk(2); exact-erf GELU) then Linear."""
    a, gate = F.linear(x, w[f"{p}.net.0.proj.weight"], w[f"{p}.net.0.proj.bias"]).chunk(2, dim=-1)
    return F.linear(a * F.gelu(gate), w[f"{p}.net.2.weight"], w[f"{p}.net.2.bias"])


def transformer_block(w: Tensors, p: str, x: torch.Tensor, cond: torch.Tensor, n_heads: int) -> torch.Tensor:
    """unet_attention.py:89-124 - pre-LN self-attn, cross-attn, FF, each with residual."""
    c = x.shape[-1]
    x = attention(w, f"{p}.attn1", F.layer_norm(x, (c,), w[f"{p}.norm1.weight"], w[f"{p}.norm1.bias"]), None, n_heads) + x
    x = attention(w, f"{p}.attn2", F.layer_norm(x, (c,), w[f"{p}.norm2.weight"], w[f"{p}.norm2.bias"]), cond, n_heads) + x
    x = feed_forward(w, f"{p}.ff", F.layer_norm(x, (c,), w[f"{p}.norm3.weight"], w[f"{p}.norm3.bias"])) + x
    return x


def spatial_transformer(w: Tensors, p: str, x: torch.Tensor, cond: torch.Tensor, cfg) -> torch.Tensor:
    """unet_attention.py:26-86 - GN(eps 1e-6), 1x1, tokens [B,HW,C], blocks, 1x1, residual."""
    b, c, h, wd = x.shape
    y = F.group_norm(x, 32, w[f"{p}.norm.weight"], w[f"{p}.norm.bias"], eps=1e-6)
    y = F.conv2d(y, w[f"{p}.proj_in.weight"], w[f"{p}.proj_in.bias"])
    y = y.permute(0, 2, 3, 1).reshape(b, h * wd, c)
    for i in range(cfg.tf_layers):
        y = transformer_block(w, f"{p}.transformer_blocks.{i}", y, cond, cfg.n_heads)
    y = y.view(b, h, wd, c).permute(0, 3, 1, 2)
    y = F.conv2d(y, w[f"{p}.proj_out.weight"], w[f"{p}.proj_out.bias"])
    return y + x


def _run_layers(w: Tensors, prefix: str, layers, x, t_emb, cond, cfg, trace=None):
    """unet.py:199-215 - dispatch by layer type."""
    for li, kind in enumerate(layers):
        p = f"{prefix}.{li}"
        if kind == "conv3":
            x = F.conv2d(x, w[f"{p}.weight"], w[f"{p}.bias"], padding=1)
        elif kind == "res":
            x = res_block(w, p, x, t_emb)
        elif kind == "st":
            x = spatial_transformer(w, p, x, cond, cfg)
        elif kind == "down":  # unet.py:241-259
            x = F.conv2d(x, w[f"{p}.op.weight"], w[f"{p}.op.bias"], stride=2, padding=1)
        elif kind == "up":  # unet.py:218-238
            x = F.interpolate(x, scale_factor=2, mode="nearest")
            x = F.conv2d(x, w[f"{p}.conv.weight"], w[f"{p}.conv.bias"], padding=1)
        if trace is not None:
            trace[p] = x
    return x


def unet_forward(w: Tensors, cfg, x: torch.Tensor, t: torch.Tensor, cond: torch.Tensor,
                 trace: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """unet.py:171-196 - x [B,Cin,H,W] fp32, t [B] int64, cond [B,n_cond,d_cond] fp32."""
    input_blocks, middle_block, output_blocks = block_lists(cfg)
    t_emb = time_embed(w, t, cfg.channels)
    skips: List[torch.Tensor] = []
    for bi, blk in enumerate(input_blocks):
        x = _run_layers(w, f"input_blocks.{bi}", blk, x, t_emb, cond, cfg, trace)
        skips.append(x)
    x = _run_layers(w, "middle_block", middle_block, x, t_emb, cond, cfg, trace)
    for bi, blk in enumerate(output_blocks):
        x = torch.cat([x, skips.pop()], dim=1)  # x first (unet.py:192)
        x = _run_layers(w, f"output_blocks.{bi}", blk, x, t_emb, cond, cfg, trace)
    x = F.group_norm(x, 32, w["out.0.weight"], w["out.0.bias"], eps=1e-5)
    return F.conv2d(F.silu(x), w["out.2.weight"], w["out.2.bias"], padding=1)


def to_torch(state, dtype=torch.float32) -> Tensors:
    return {k: torch.as_tensor(v).to(dtype) for k, v in state.items()}
