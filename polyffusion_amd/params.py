"""The params surface of the reference (``polyffusion/params/*.yaml``, copied by the trainer to
``<run>/params.yaml`` and discovered by ``inference_sdf.py:518-534``).

``load_params(path)`` reads such a YAML/JSON file with PyYAML into an attribute dict (OmegaConf is
not installed here and is not needed: only flat keys and two integer lists are used).  ``PRESETS``
holds the model-defining keys of the shipped configurations so synthetic runs and benchmarks need no
file at all; training-only keys (batch_size, learning_rate, ...) are accepted and ignored.
"""
from __future__ import annotations

import json
import os
from typing import Any, Dict


class Params(dict):
    """dict with attribute access (the subset of OmegaConf behaviour the path relies on)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


_UNET = dict(in_channels=2, out_channels=2, channels=64, attention_levels=[2, 3], n_res_blocks=2,
             channel_multipliers=[1, 2, 4, 4], n_heads=4, tf_layers=1, linear_start=0.00085, linear_end=0.012,
             n_steps=1000, latent_scaling_factor=0.18215, img_h=128, img_w=128, cond_mode="mix", use_enc=True)
_CHD = dict(chd_n_step=32, chd_input_dim=36, chd_z_input_dim=512, chd_hidden_dim=512, chd_z_dim=512)
_TXT = dict(txt_emb_size=256, txt_hidden_dim=1024, txt_z_dim=256, txt_num_channel=10)

PRESETS: Dict[str, Dict[str, Any]] = {
    "sdf_chd8bar": dict(_UNET, model_name="sdf_chd8bar", d_cond=512, cond_type="chord", **_CHD),
    "sdf_txt": dict(_UNET, model_name="sdf_txt", d_cond=1024, cond_type="txt", **_TXT),
    "sdf_chd8bar_txt": dict(_UNET, model_name="sdf_chd8bar_txt", d_cond=1536, cond_type="chord+txt", **_CHD, **_TXT),
    "sdf_txtvnl": dict(_UNET, model_name="sdf_txtvnl", d_cond=128, cond_type="txt", use_enc=False),
    # params/sdf_concat.yaml as shipped (in_channels 3: with a 2-channel blurry image the reference's own cat([x, cond_concat]) does not fit it either)
    "sdf_concat": dict(_UNET, model_name="sdf_concat", in_channels=3, d_cond=1152, cond_type="chord", cond_mode="uncond", use_enc=False,
                       chd_n_step=32, chd_input_dim=36, concat_blurry=True, concat_ratio=0.25),
    "sdf_pnotree": dict(_UNET, model_name="sdf_pnotree", d_cond=2048, cond_type="pnotree"),
    "sdf_chdvnl": dict(_UNET, model_name="sdf_chdvnl", d_cond=1152, cond_type="chord", use_enc=False,
                       chd_n_step=32, chd_input_dim=36),
}


def preset(name: str) -> Params:
    if name not in PRESETS:
        raise KeyError(f"unknown params preset {name!r}; known: {sorted(PRESETS)}")
    return Params(PRESETS[name])


def load_params(path: str) -> Params:
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    with open(path) as f:
        if path.endswith(".json"):
            data = json.load(f)
        else:
            import yaml
            data = yaml.safe_load(f)
    p = Params(data)
    for k in ("in_channels", "out_channels", "channels", "attention_levels", "n_res_blocks", "channel_multipliers",
              "n_heads", "tf_layers", "d_cond", "linear_start", "linear_end", "n_steps", "img_h", "img_w", "cond_type"):
        if k not in p:
            raise KeyError(f"{path}: missing params key {k!r}")
    p.setdefault("model_name", os.path.splitext(os.path.basename(path))[0])
    p.setdefault("cond_mode", "cond")
    p.setdefault("use_enc", True)
    p.setdefault("latent_scaling_factor", 0.18215)
    return p


def find_params(chkpt_path: str, custom_params_path=None) -> str:
    """``<chkpt>/../../params.yaml|json`` unless overridden (inference_sdf.py:518-532)."""
    if custom_params_path is not None:
        return custom_params_path
    model_path = os.path.dirname(os.path.dirname(os.path.abspath(chkpt_path)))
    for ext in ("yaml", "json"):
        cand = os.path.join(model_path, f"params.{ext}")
        if os.path.exists(cand):
            return cand
    raise FileNotFoundError(f"params.yaml or params.json not found in {model_path}, please specify custom_params_path then.")
