"""Builds synthetic checkpoints in the two on-disk formats of the reference (test helper, no reference code).

* ``write_legacy_pt``   - what ``learner.py:70-84`` saves: ``{"step","epoch","model","optimizer","scaler"}``.
* ``write_lightning_ckpt`` - what Lightning writes for ``LightningLearner`` (``lightning_learner.py:5-13``): ``state_dict``
  with ``model.``-prefixed keys and ``hyper_parameters["params"]`` as an OmegaConf ``DictConfig``.  omegaconf is not
  installed, so a throw-away module tree with the same class names and the same pickled attribute layout
  (``_content`` of ``AnyNode``s holding ``_val``, ``_metadata``, ``_parent`` back-references) is registered in
  ``sys.modules`` only while ``torch.save`` runs; the loader under test must cope without it.
"""
import sys
import types
from typing import Any

import numpy as np
import torch



def full_state(unet_state, chord_state=None, txt_state=None, n_steps=1000, lin=(0.00085, 0.012), with_decoder=True):
    st = {f"ldm.eps_model.{k}": torch.as_tensor(np.asarray(v)) for k, v in unet_state.items()}
    beta = (torch.linspace(lin[0] ** 0.5, lin[1] ** 0.5, n_steps, dtype=torch.float64) ** 2)
    alpha = 1.0 - beta
    alpha_bar, alpha, beta = torch.cumprod(alpha, 0).float(), alpha.float(), beta.float()
    st.update({"ldm.alpha": alpha, "ldm.beta": beta, "ldm.alpha_bar": alpha_bar, "ldm.sigma2": beta})
    for k, v in (chord_state or {}).items():
        st[f"chord_enc.{k}"] = torch.as_tensor(np.asarray(v))
    for k, v in (txt_state or {}).items():
        st[f"txt_enc.{k}"] = torch.as_tensor(np.asarray(v))
    if with_decoder:   # frozen decode-only modules ride along in real checkpoints
        st["chord_dec.z2dec_hid.weight"] = torch.zeros(4, 4)
        st["chord_dec.z2dec_hid.bias"] = torch.zeros(4)
    return st


def write_legacy_pt(path, state):
    torch.save({"step": 1234, "epoch": 7, "model": state,
                "optimizer": {"state": {0: {"step": torch.tensor(5.0), "exp_avg": torch.zeros(3)}},
                              "param_groups": [{"lr": 5e-5, "betas": (0.9, 0.999), "params": [0]}]},
                "scaler": {"scale": 65536.0, "growth_factor": 2.0, "_growth_tracker": 0}}, path)


def _fake_omegaconf():
    mods = {}

    def mod(name):
        m = types.ModuleType(name)
        mods[name] = m
        return m

    root, base, nodes, dc, lc = (mod(n) for n in ("omegaconf", "omegaconf.base", "omegaconf.nodes",
                                                  "omegaconf.dictconfig", "omegaconf.listconfig"))

    def cls(m, name):
        c = type(name, (), {"__module__": m.__name__})
        setattr(m, name, c)
        return c

    Metadata, ContainerMetadata = cls(base, "Metadata"), cls(base, "ContainerMetadata")
    AnyNode, DictConfig, ListConfig = cls(nodes, "AnyNode"), cls(dc, "DictConfig"), cls(lc, "ListConfig")
    root.DictConfig, root.ListConfig = DictConfig, ListConfig

    def node(v, parent, key):
        n = AnyNode()
        md = Metadata()
        md.__dict__.update(ref_type=Any, object_type=None, optional=True, key=key, flags=None, flags_root=False, resolver_cache={})
        n.__dict__.update(_metadata=md, _parent=parent, _flags_cache=None, _val=v)
        return n

    def wrap(v, parent=None, key=None):
        if isinstance(v, dict):
            c = DictConfig()
            md = ContainerMetadata()
            md.__dict__.update(ref_type=Any, object_type=dict, optional=True, key=key, flags={}, flags_root=False,
                               resolver_cache={}, key_type=Any, element_type=Any)
            c.__dict__.update(_metadata=md, _parent=parent, _flags_cache=None)
            c.__dict__["_content"] = {k: wrap(x, c, k) for k, x in v.items()}
            return c
        if isinstance(v, (list, tuple)):
            c = ListConfig()
            md = ContainerMetadata()
            md.__dict__.update(ref_type=Any, object_type=list, optional=True, key=key, flags={}, flags_root=False,
                               resolver_cache={}, key_type=int, element_type=Any)
            c.__dict__.update(_metadata=md, _parent=parent, _flags_cache=None)
            c.__dict__["_content"] = [wrap(x, c, i) for i, x in enumerate(v)]
            return c
        return node(v, parent, key)

    return mods, wrap


def write_lightning_ckpt(path, state, params: dict):
    mods, wrap = _fake_omegaconf()
    saved = {k: sys.modules.get(k) for k in mods}
    sys.modules.update(mods)
    try:
        torch.save({"epoch": 3, "global_step": 4321, "pytorch-lightning_version": "2.1.3",
                    "state_dict": {f"model.{k}": v for k, v in state.items()},
                    "loops": {"fit_loop": {"state_dict": {}, "epoch_progress": {"total": {"ready": 4, "completed": 3}}}},
                    "callbacks": {"ModelCheckpoint{'monitor': 'val/loss'}": {"best_model_score": torch.tensor(0.0123), "best_k_models": {}}},
                    "optimizer_states": [], "lr_schedulers": [],
                    "hparams_name": "kwargs",
                    "hyper_parameters": {"params": wrap(params), "param_scheduler": None}}, path)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    assert "omegaconf" not in sys.modules or saved["omegaconf"] is not None
