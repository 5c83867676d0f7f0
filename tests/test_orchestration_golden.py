"""Rows a27/a28: the orchestration code (``Experiments.predict`` incl. the 2B-1 autoregressive schedule,
``get_autoreg_data``, ``get_mask``, ``dummy_cond_input``) pinned to ``tests/golden/orchestration.npz``.

The fixture was recorded by ``tools/make_goldens_orch.py``, which compiles those definitions from the
reference's own ``inference_sdf.py`` (ast-selected, in memory) and drives them with the imported real
samplers and a noise tape.  CPU tests here pin the oracle restatement and the product's host functions;
``tests/test_gpu_orchestration.py`` pins the product's ``Experiments.predict`` on the GPU.
"""
import numpy as np
import pytest
import torch

from oracle import sampler_ref, unet_ref
from polyffusion_amd.arch import UNetConfig
from polyffusion_amd.weights import synth_unet_state

SMALL = UNetConfig(in_channels=2, out_channels=2, channels=32, n_res_blocks=1, attention_levels=(1,),
                   channel_multipliers=(1, 2), n_heads=2, tf_layers=1, d_cond=32)
LIN = (0.00085, 0.012)
MASK_CASES = ("dense", "sparse", "one")
# tag -> (sampler kind, predict kwargs, repaint_n)
PREDICT_CASES = {
    "plain": ("ddpm", dict(), 1),
    "autoreg": ("ddpm", dict(autoreg=True), 1),
    "autoreg_inp": ("ddpm", dict(autoreg=True, uncond_scale=2.0, inpaint=True), 1),
    "autoreg_rp2": ("ddpm", dict(autoreg=True, inpaint=True), 2),
    "autoreg_ddim": ("ddim", dict(autoreg=True, uncond_scale=3.0, inpaint=True), 1),
}


def tape_fn(g, tag):
    draws = iter(g[f"pred_{tag}_tape"])
    return lambda shape: torch.from_numpy(np.ascontiguousarray(next(draws))).reshape(shape)


def test_autoreg_data_oracle_and_product(golden):
    from polyffusion_amd.inference_sdf import get_autoreg_data
    g = golden("orchestration.npz")
    for fn in (sampler_ref.get_autoreg_data, get_autoreg_data):
        assert np.array_equal(fn(torch.from_numpy(g["autoreg_in_dim1"])).numpy(), g["autoreg_out_dim1"])
        assert np.array_equal(fn(torch.from_numpy(g["autoreg_in_dim2"]), split_dim=2).numpy(), g["autoreg_out_dim2"])


@pytest.mark.parametrize("case", MASK_CASES)
@pytest.mark.parametrize("kind", ("remaining", "below", "above"))
def test_get_mask_oracle_and_product(golden, case, kind):
    from polyffusion_amd.inference_sdf import get_mask
    g = golden("orchestration.npz")
    want = g[f"mask_{kind}_{case}"]
    for fn in (sampler_ref.get_mask, get_mask):
        got = fn(torch.from_numpy(g[f"mask_orig_{case}"].copy()), kind)
        assert got.shape == want.shape and np.array_equal(got.numpy(), want), (fn.__module__, kind, case)


def test_get_mask_bars(golden):
    from polyffusion_amd.inference_sdf import get_mask
    g = golden("orchestration.npz")
    for fn in (sampler_ref.get_mask, get_mask):
        got = fn(torch.from_numpy(g["mask_orig_bars"].copy()), "bars", [int(b) for b in g["mask_bars_list"]])
        assert np.array_equal(got.numpy(), g["mask_bars"])
    with pytest.raises(NotImplementedError):
        get_mask(torch.zeros(1, 2, 16, 128), "sideways")


@pytest.mark.parametrize("tag", list(PREDICT_CASES))
def test_oracle_predict_matches_reference(golden, tag):
    g = golden("orchestration.npz")
    kind, kw, repaint_n = PREDICT_CASES[tag]
    kw = dict(kw)
    w = unet_ref.to_torch(synth_unet_state(SMALL, 0))
    model = lambda x, t, c: unet_ref.unet_forward(w, SMALL, x, t, c)
    if kind == "ddpm":
        s, t_idx = sampler_ref.SDFSamplerRef(model, 1000, *LIN, noise_fn=tape_fn(g, tag)), 3
    else:
        s, t_idx = sampler_ref.DDIMSamplerRef(model, 1000, *LIN, n_steps=10, noise_fn=tape_fn(g, tag)), 2
    orig = mask = None
    if kw.pop("inpaint", False):
        orig, mask = torch.from_numpy(g["pred_orig"].copy()), torch.from_numpy(g["pred_mask"].copy())
    cond, cond_mid = torch.from_numpy(g["pred_cond"]), torch.from_numpy(g["pred_cond_mid"])
    with torch.no_grad():
        out = sampler_ref.predict(s, cond, SMALL.d_cond, [3, 2, 16, 16], t_idx, torch.from_numpy(g[f"pred_{tag}_tape0"].copy()),
                                  cond_mid=cond_mid, orig=orig, mask=mask, repaint_n=repaint_n, **kw)
    want = g[f"pred_{tag}_out"]
    assert out.shape == want.shape
    assert float((out - torch.from_numpy(want)).abs().max()) <= 2e-5


def test_get_blurry_image(golden):
    from polyffusion_amd.inference_sdf import get_blurry_image
    g = golden("orchestration.npz")
    img = torch.from_numpy(g["blurry_in"])
    for tag, ratio in (("r8", 1 / 8), ("r4", 0.25)):
        got = get_blurry_image(img.clone(), ratio)
        assert got.shape == g[f"blurry_{tag}"].shape and float((got - torch.from_numpy(g[f"blurry_{tag}"])).abs().max()) <= 1e-6


SMALL4 = UNetConfig(in_channels=4, out_channels=2, channels=32, n_res_blocks=1, attention_levels=(1,),
                    channel_multipliers=(1, 2), n_heads=2, tf_layers=1, d_cond=32)
# tests/golden/orchestration_concat.npz (tools/make_goldens_concat.py): the reference's predict(cond_concat=...) - the concat_blurry variant
CONCAT_CASES = {
    "plain": ("ddpm", dict(), False, 3),
    "inp_cfg": ("ddpm", dict(uncond_scale=2.0), True, 3),
    "autoreg1": ("ddpm", dict(autoreg=True), False, 1),
    "ddim": ("ddim", dict(uncond_scale=3.0), True, 3),
}


@pytest.mark.parametrize("tag", list(CONCAT_CASES))
def test_oracle_predict_with_cond_concat_matches_reference(golden, tag):
    from polyffusion_amd.inference_sdf import get_blurry_image
    g = golden("orchestration_concat.npz")
    kind, kw, inpaint, n = CONCAT_CASES[tag]
    w = unet_ref.to_torch(synth_unet_state(SMALL4, 0))
    model = lambda x, t, c: unet_ref.unet_forward(w, SMALL4, x, t, c)
    draws = iter(g[f"{tag}_tape"])
    noise_fn = lambda shape: torch.from_numpy(np.ascontiguousarray(next(draws))).reshape(shape)
    if kind == "ddpm":
        s, t_idx = sampler_ref.SDFSamplerRef(model, 1000, *LIN, noise_fn=noise_fn), 3
    else:
        s, t_idx = sampler_ref.DDIMSamplerRef(model, 1000, *LIN, n_steps=10, noise_fn=noise_fn), 2
    cc = get_blurry_image(torch.from_numpy(g["image"]), 0.25)          # the product's input preparation, pinned by the same fixture
    assert float((cc - torch.from_numpy(g["cond_concat"])).abs().max()) <= 1e-6
    orig = mask = None
    if inpaint:
        orig, mask = torch.from_numpy(g["orig"].copy()), torch.from_numpy(g["mask"].copy())
    with torch.no_grad():
        out = sampler_ref.predict(s, torch.from_numpy(g["cond"])[:n], SMALL4.d_cond, [n, 2, 16, 16], t_idx, torch.from_numpy(g[f"{tag}_tape0"].copy()),
                                  cond_mid=torch.from_numpy(g["cond_mid"])[:n], orig=orig, mask=mask, cond_concat=cc[:n], **kw)
    want = g[f"{tag}_out"]
    assert out.shape == want.shape and float((out - torch.from_numpy(want)).abs().max()) <= 2e-5
    assert int(g["autoreg3_fails"]) == 1    # the reference itself cannot run cond_concat autoregressively on a multi-segment batch
