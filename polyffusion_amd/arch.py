"""Architecture walk of the Polyffusion denoiser (host-side description only).

This module holds NO arithmetic.  It reproduces the *constructor logic* of the
reference UNet (reference: polyffusion/stable_diffusion/model/unet.py:30-149)
as a flat, data-only description that two consumers share:

* ``polyffusion_amd.weights``  - names/shapes of every parameter (the reference
  ``state_dict`` key namespace);
* ``tests``                    - to cross-check the C++ plan builder in
  ``csrc/unet.hip`` (which re-derives the same walk natively).

(The CPU oracle does NOT use it: ``oracle/unet_ref.py`` derives its own block lists.)
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Tuple


@dataclass(frozen=True)
class UNetConfig:
    """Constructor surface of the reference ``UNetModel`` (unet.py:35-47)."""

    in_channels: int = 2
    out_channels: int = 2
    channels: int = 64
    n_res_blocks: int = 2
    attention_levels: Tuple[int, ...] = (2, 3)
    channel_multipliers: Tuple[int, ...] = (1, 2, 4, 4)
    n_heads: int = 4
    tf_layers: int = 1
    d_cond: int = 512

    @staticmethod
    def from_params(p) -> "UNetConfig":
        g = (lambda k: p[k]) if isinstance(p, dict) else (lambda k: getattr(p, k))
        return UNetConfig(
            in_channels=int(g("in_channels")),
            out_channels=int(g("out_channels")),
            channels=int(g("channels")),
            n_res_blocks=int(g("n_res_blocks")),
            attention_levels=tuple(int(v) for v in g("attention_levels")),
            channel_multipliers=tuple(int(v) for v in g("channel_multipliers")),
            n_heads=int(g("n_heads")),
            tf_layers=int(g("tf_layers")),
            d_cond=int(g("d_cond")),
        )

    @property
    def d_time_emb(self) -> int:
        return self.channels * 4


# A layer is (kind, cin, cout); kinds: conv3, res, st, down, up
Layer = Tuple[str, int, int]


@dataclass
class UNetLayout:
    input_blocks: List[List[Layer]] = field(default_factory=list)
    middle_block: List[Layer] = field(default_factory=list)
    output_blocks: List[List[Layer]] = field(default_factory=list)
    #: channels of the skip tensor consumed by each output block (pop order)
    skip_channels: List[int] = field(default_factory=list)
    final_channels: int = 0


def unet_layout(cfg: UNetConfig) -> UNetLayout:
    """Block list in execution order (mirrors unet.py:70-149)."""
    lay = UNetLayout()
    levels = len(cfg.channel_multipliers)
    ch = cfg.channels
    lay.input_blocks.append([("conv3", cfg.in_channels, ch)])
    stack = [ch]
    widths = [cfg.channels * m for m in cfg.channel_multipliers]
    for lvl in range(levels):
        for _ in range(cfg.n_res_blocks):
            blk: List[Layer] = [("res", ch, widths[lvl])]
            ch = widths[lvl]
            if lvl in cfg.attention_levels:
                blk.append(("st", ch, ch))
            lay.input_blocks.append(blk)
            stack.append(ch)
        if lvl != levels - 1:
            lay.input_blocks.append([("down", ch, ch)])
            stack.append(ch)
    lay.middle_block = [("res", ch, ch), ("st", ch, ch), ("res", ch, ch)]
    for lvl in reversed(range(levels)):
        for j in range(cfg.n_res_blocks + 1):
            skip = stack.pop()
            lay.skip_channels.append(skip)
            blk = [("res", ch + skip, widths[lvl])]
            ch = widths[lvl]
            if lvl in cfg.attention_levels:
                blk.append(("st", ch, ch))
            if lvl != 0 and j == cfg.n_res_blocks:
                blk.append(("up", ch, ch))
            lay.output_blocks.append(blk)
    lay.final_channels = ch
    return lay


def _res_shapes(prefix: str, cin: int, cout: int, d_t: int, out: Dict):
    out[f"{prefix}.in_layers.0.weight"] = (cin,)
    out[f"{prefix}.in_layers.0.bias"] = (cin,)
    out[f"{prefix}.in_layers.2.weight"] = (cout, cin, 3, 3)
    out[f"{prefix}.in_layers.2.bias"] = (cout,)
    out[f"{prefix}.emb_layers.1.weight"] = (cout, d_t)
    out[f"{prefix}.emb_layers.1.bias"] = (cout,)
    out[f"{prefix}.out_layers.0.weight"] = (cout,)
    out[f"{prefix}.out_layers.0.bias"] = (cout,)
    out[f"{prefix}.out_layers.3.weight"] = (cout, cout, 3, 3)
    out[f"{prefix}.out_layers.3.bias"] = (cout,)
    if cin != cout:
        out[f"{prefix}.skip_connection.weight"] = (cout, cin, 1, 1)
        out[f"{prefix}.skip_connection.bias"] = (cout,)


def _st_shapes(prefix: str, ch: int, cfg: UNetConfig, out: Dict):
    out[f"{prefix}.norm.weight"] = (ch,)
    out[f"{prefix}.norm.bias"] = (ch,)
    out[f"{prefix}.proj_in.weight"] = (ch, ch, 1, 1)
    out[f"{prefix}.proj_in.bias"] = (ch,)
    for i in range(cfg.tf_layers):
        tb = f"{prefix}.transformer_blocks.{i}"
        for a, dc in (("attn1", ch), ("attn2", cfg.d_cond)):
            out[f"{tb}.{a}.to_q.weight"] = (ch, ch)
            out[f"{tb}.{a}.to_k.weight"] = (ch, dc)
            out[f"{tb}.{a}.to_v.weight"] = (ch, dc)
            out[f"{tb}.{a}.to_out.0.weight"] = (ch, ch)
            out[f"{tb}.{a}.to_out.0.bias"] = (ch,)
        for n in ("norm1", "norm2", "norm3"):
            out[f"{tb}.{n}.weight"] = (ch,)
            out[f"{tb}.{n}.bias"] = (ch,)
        out[f"{tb}.ff.net.0.proj.weight"] = (ch * 8, ch)
        out[f"{tb}.ff.net.0.proj.bias"] = (ch * 8,)
        out[f"{tb}.ff.net.2.weight"] = (ch, ch * 4)
        out[f"{tb}.ff.net.2.bias"] = (ch,)
    out[f"{prefix}.proj_out.weight"] = (ch, ch, 1, 1)
    out[f"{prefix}.proj_out.bias"] = (ch,)


def _layer_shapes(prefix: str, layer: Layer, cfg: UNetConfig, out: Dict):
    kind, cin, cout = layer
    if kind == "conv3":
        out[f"{prefix}.weight"] = (cout, cin, 3, 3)
        out[f"{prefix}.bias"] = (cout,)
    elif kind == "res":
        _res_shapes(prefix, cin, cout, cfg.d_time_emb, out)
    elif kind == "st":
        _st_shapes(prefix, cin, cfg, out)
    elif kind == "down":
        out[f"{prefix}.op.weight"] = (cout, cin, 3, 3)
        out[f"{prefix}.op.bias"] = (cout,)
    elif kind == "up":
        out[f"{prefix}.conv.weight"] = (cout, cin, 3, 3)
        out[f"{prefix}.conv.bias"] = (cout,)
    else:  # pragma: no cover
        raise ValueError(kind)


def unet_param_shapes(cfg: UNetConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """name -> shape for the UNet, names relative to ``eps_model`` (Appendix D)."""
    lay = unet_layout(cfg)
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    d_t = cfg.d_time_emb
    out["time_embed.0.weight"] = (d_t, cfg.channels)
    out["time_embed.0.bias"] = (d_t,)
    out["time_embed.2.weight"] = (d_t, d_t)
    out["time_embed.2.bias"] = (d_t,)
    for bi, blk in enumerate(lay.input_blocks):
        for li, layer in enumerate(blk):
            _layer_shapes(f"input_blocks.{bi}.{li}", layer, cfg, out)
    for li, layer in enumerate(lay.middle_block):
        _layer_shapes(f"middle_block.{li}", layer, cfg, out)
    for bi, blk in enumerate(lay.output_blocks):
        for li, layer in enumerate(blk):
            _layer_shapes(f"output_blocks.{bi}.{li}", layer, cfg, out)
    c = lay.final_channels
    out["out.0.weight"] = (c,)
    out["out.0.bias"] = (c,)
    out["out.2.weight"] = (cfg.out_channels, c, 3, 3)
    out["out.2.bias"] = (cfg.out_channels,)
    return out


def chord_encoder_param_shapes(input_dim=36, hidden_dim=512, z_dim=512):
    """reference: dl_modules/chord_enc.py:5-13 (bi-GRU + two heads)."""
    out = OrderedDict()
    for sfx in ("", "_reverse"):
        out[f"gru.weight_ih_l0{sfx}"] = (3 * hidden_dim, input_dim)
        out[f"gru.weight_hh_l0{sfx}"] = (3 * hidden_dim, hidden_dim)
        out[f"gru.bias_ih_l0{sfx}"] = (3 * hidden_dim,)
        out[f"gru.bias_hh_l0{sfx}"] = (3 * hidden_dim,)
    out["linear_mu.weight"] = (z_dim, 2 * hidden_dim)
    out["linear_mu.bias"] = (z_dim,)
    out["linear_var.weight"] = (z_dim, 2 * hidden_dim)
    out["linear_var.bias"] = (z_dim,)
    return out


def texture_encoder_param_shapes(emb_size=256, hidden_dim=1024, z_dim=256, num_channel=10):
    """reference: dl_modules/txt_enc.py:5-21."""
    out = OrderedDict()
    out["cnn.0.weight"] = (num_channel, 1, 4, 12)
    out["cnn.0.bias"] = (num_channel,)
    out["fc1.weight"] = (1000, num_channel * 29)
    out["fc1.bias"] = (1000,)
    out["fc2.weight"] = (emb_size, 1000)
    out["fc2.bias"] = (emb_size,)
    for sfx in ("", "_reverse"):
        out[f"gru.weight_ih_l0{sfx}"] = (3 * hidden_dim, emb_size)
        out[f"gru.weight_hh_l0{sfx}"] = (3 * hidden_dim, hidden_dim)
        out[f"gru.bias_ih_l0{sfx}"] = (3 * hidden_dim,)
        out[f"gru.bias_hh_l0{sfx}"] = (3 * hidden_dim,)
    out["linear_mu.weight"] = (z_dim, 2 * hidden_dim)
    out["linear_mu.bias"] = (z_dim,)
    out["linear_var.weight"] = (z_dim, 2 * hidden_dim)
    out["linear_var.bias"] = (z_dim,)
    return out


def pianotree_encoder_param_shapes(note_size=135, note_emb_size=128, enc_notes_hid_size=256, enc_time_hid_size=512, z_size=512):
    """reference: dl_modules/pianotree_enc.py:43-59 (note embedding, bi-GRU over the notes of a step, bi-GRU over the 32 steps, two heads)."""
    out = OrderedDict()
    out["note_embedding.weight"] = (note_emb_size, note_size)
    out["note_embedding.bias"] = (note_emb_size,)
    for name, hid, inp in (("enc_notes_gru", enc_notes_hid_size, note_emb_size), ("enc_time_gru", enc_time_hid_size, 2 * enc_notes_hid_size)):
        for sfx in ("", "_reverse"):
            out[f"{name}.weight_ih_l0{sfx}"] = (3 * hid, inp)
            out[f"{name}.weight_hh_l0{sfx}"] = (3 * hid, hid)
            out[f"{name}.bias_ih_l0{sfx}"] = (3 * hid,)
            out[f"{name}.bias_hh_l0{sfx}"] = (3 * hid,)
    out["linear_mu.weight"] = (z_size, 2 * enc_time_hid_size)
    out["linear_mu.bias"] = (z_size,)
    out["linear_std.weight"] = (z_size, 2 * enc_time_hid_size)
    out["linear_std.bias"] = (z_size,)
    return out
