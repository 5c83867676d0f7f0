"""Row f1 (CPU part): both checkpoint formats decode without lightning / omegaconf, through a restricted unpickler,
and the pretrained-encoder key remaps of ``utils.py:48-86`` keep exactly the reference's keys."""
import pickle

import numpy as np
import pytest
import torch

from ckpt_fixture import full_state, write_legacy_pt, write_lightning_ckpt
from polyffusion_amd import checkpoint
from polyffusion_amd.arch import UNetConfig, unet_param_shapes
from polyffusion_amd.params import PRESETS
from polyffusion_amd.weights import synth_chord_encoder_state, synth_unet_state

SMALL = UNetConfig(in_channels=2, out_channels=2, channels=32, n_res_blocks=1, attention_levels=(1,),
                   channel_multipliers=(1, 2), n_heads=2, tf_layers=1, d_cond=32)


def test_legacy_pt_roundtrip(tmp_path):
    st = full_state(synth_unet_state(SMALL, 0), synth_chord_encoder_state(0))
    write_legacy_pt(str(tmp_path / "weights_best.pt"), st)
    got, params = checkpoint.load_checkpoint(str(tmp_path / "weights_best.pt"))
    assert params is None and set(got) == set(st)
    unet, ce, te = checkpoint.split_state(got)
    assert set(unet) == set(unet_param_shapes(SMALL)) and te == {}
    assert set(ce) == set(synth_chord_encoder_state(0))
    assert all(torch.equal(unet[k], st["ldm.eps_model." + k]) for k in unet)
    with pytest.raises(RuntimeError, match="unexpected key"):
        checkpoint.split_state(dict(got, **{"ldm.first_stage_model.w": torch.zeros(1)}))


def test_lightning_ckpt_without_omegaconf(tmp_path):
    import sys
    assert "omegaconf" not in sys.modules and "lightning" not in sys.modules
    st = full_state(synth_unet_state(SMALL, 0))
    params = dict(PRESETS["sdf_chd8bar"], batch_size=16, learning_rate=5e-5, fp16=True)
    path = str(tmp_path / "epoch=3.ckpt")
    write_lightning_ckpt(path, st, params)
    raw = open(path, "rb").read()
    assert b"omegaconf" in raw                      # the pickle really refers to the absent package
    got, p = checkpoint.load_lightning_ckpt(path)
    assert set(got) == set(st) and all(torch.equal(got[k], st[k]) for k in st)     # "model." removed, tensors intact
    assert p == params                              # DictConfig / ListConfig -> plain dict / list, nested lists included
    assert p["channel_multipliers"] == [1, 2, 4, 4] and isinstance(p["linear_start"], float)
    with pytest.raises(RuntimeError, match="unknown checkpoint type"):
        checkpoint.load_checkpoint(str(tmp_path / "x.bin"))


def test_restricted_unpickler_refuses_code(tmp_path):
    class Evil:
        def __reduce__(self):
            import os
            return (os.system, ("echo pwned > /dev/null",))
    path = str(tmp_path / "evil.ckpt")
    torch.save({"state_dict": {"model.w": torch.zeros(1)}, "hyper_parameters": {"params": Evil()}}, path)
    with pytest.raises(pickle.UnpicklingError, match="does not allow"):
        checkpoint.load_lightning_ckpt(path)
    torch.save({"state_dict": {"w": torch.zeros(1)}}, path)
    with pytest.raises(RuntimeError, match="unexpected key"):
        checkpoint.load_lightning_ckpt(path)


def _raw_ckpt(path, pickle_bytes):
    """A torch zip archive whose data.pkl is `pickle_bytes` (what a hostile download would look like)."""
    import zipfile
    with zipfile.ZipFile(path, "w") as z:
        z.writestr("archive/data.pkl", pickle_bytes)
        z.writestr("archive/version", "3\n")
        z.writestr("archive/byteorder", "little")


@pytest.mark.parametrize("gadget", ["load_from_bytes", "import_dotted_name", "rebuild_from_type", "serialization_load"])
def test_restricted_unpickler_refuses_torch_gadgets(tmp_path, gadget):
    """globals INSIDE torch that reach an unrestricted loader or an arbitrary callable must be refused by name."""
    import io
    inner = io.BytesIO()
    torch.save(torch.zeros(1), inner)
    g = {"load_from_bytes": ("torch.storage", "_load_from_bytes", (inner.getvalue(),)),
         "import_dotted_name": ("torch._utils", "_import_dotted_name", ("os.getcwd",)),
         "rebuild_from_type": ("torch._tensor", "_rebuild_from_type_v2", ()),
         "serialization_load": ("torch.serialization", "load", (inner.getvalue(),))}[gadget]
    # protocol-2 pickle by hand: GLOBAL module name, args tuple, REDUCE - inside {"state_dict": {}, "hyper_parameters": <gadget>}
    body = (b"\x80\x02}(X\n\x00\x00\x00state_dict}X\x10\x00\x00\x00hyper_parametersc"
            + g[0].encode() + b"\n" + g[1].encode() + b"\n" + pickle.dumps(g[2], protocol=2)[2:-1] + b"Ru.")
    path = str(tmp_path / "gadget.ckpt")
    _raw_ckpt(path, body)
    import pickletools
    pickletools.dis(body, out=io.StringIO())   # the hand-made stream is a well-formed pickle
    with pytest.raises(pickle.UnpicklingError, match="does not allow"):
        checkpoint.load_lightning_ckpt(path)


def test_pretrained_encoder_key_remaps():
    from polyffusion_amd.model_sdf import _strip
    ce = synth_chord_encoder_state(0)
    ck = {"model": {**{f"chord_enc.{k}": v for k, v in ce.items()}, "chord_dec.out.weight": np.zeros(2), "step": 3}}
    # utils.py:48-69: keys whose first component is chord_enc, that component removed; everything else dropped
    got = _strip({"model": {k: v for k, v in ck["model"].items() if k != "step"}}, "chord_enc")
    assert set(got) == set(ce) and all(np.array_equal(got[k], ce[k]) for k in ce)
    # utils.py:72-86: a Polydis checkpoint keeps the texture encoder under rhy_encoder.
    pd = {"rhy_encoder.gru.weight_ih_l0": np.ones(3), "rhy_encoder.linear_mu.bias": np.zeros(2), "decoder.x": np.zeros(1),
          "chd_encoder.gru.weight_ih_l0": np.zeros(3)}
    assert set(_strip(pd, "rhy_encoder")) == {"gru.weight_ih_l0", "linear_mu.bias"}
