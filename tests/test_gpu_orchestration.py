"""Rows a27/a28 on the GPU (-m gpu): the product's ``Experiments.predict`` (plain, autoregressive, inpainting, RePaint n=2,
DDIM) against ``tests/golden/orchestration.npz`` - outputs of the reference's own ``Experiments.predict`` source driven by
the real samplers (``tools/make_goldens_orch.py``) - and ``predict_songs`` (config 5: the autoregressive chain batched across
songs) against independent per-song runs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from polyffusion_amd import _lib  # noqa: E402
from polyffusion_amd.arch import UNetConfig  # noqa: E402
from polyffusion_amd.inference_sdf import Experiments, dummy_cond_input, get_mask  # noqa: E402
from polyffusion_amd.params import Params  # noqa: E402
from polyffusion_amd.sampler import DDIMSampler, SDFSampler  # noqa: E402
from polyffusion_amd.unet import LatentDiffusion, UNetModel  # noqa: E402
from polyffusion_amd.weights import synth_unet_state  # noqa: E402

SMALL = UNetConfig(in_channels=2, out_channels=2, channels=32, n_res_blocks=1, attention_levels=(1,),
                   channel_multipliers=(1, 2), n_heads=2, tf_layers=1, d_cond=32)
LIN = (0.00085, 0.012)
PARAMS = dict(out_channels=2, img_h=16, img_w=16, d_cond=32, n_steps=4)   # t_idx = n_steps-1 = 3, as in the fixture
CASES = {
    "plain": ("ddpm", dict(), 1, False),
    "autoreg": ("ddpm", dict(autoreg=True), 1, False),
    "autoreg_inp": ("ddpm", dict(autoreg=True, uncond_scale=2.0), 1, True),
    "autoreg_rp2": ("ddpm", dict(autoreg=True), 2, True),
    "autoreg_ddim": ("ddim", dict(autoreg=True, uncond_scale=3.0), 1, True),
}


@pytest.fixture(scope="module", params=["f32", "bf16x3"])
def ldm(request):
    _lib.require_gpu()
    m = UNetModel(in_channels=2, out_channels=2, channels=32, n_res_blocks=1, attention_levels=(1,),
                  channel_multipliers=(1, 2), n_heads=2, tf_layers=1, d_cond=32, img_h=16, img_w=16)
    m.load_state_dict(synth_unet_state(SMALL, 0))
    m.set_precision(request.param)
    return LatentDiffusion(m, None, 0.18215, 1000, *LIN)


class Tape:
    def __init__(self, arr):
        self.arr, self.i = arr, 0

    def __call__(self, shape):
        a = torch.from_numpy(np.ascontiguousarray(self.arr[self.i]))
        self.i += 1
        assert tuple(a.shape) == tuple(shape), (a.shape, shape)
        return a


@pytest.mark.parametrize("tag", list(CASES))
def test_predict_vs_reference_golden(ldm, golden, tag):
    g = golden("orchestration.npz")
    kind, kw, repaint_n, inpaint = CASES[tag]
    tape = Tape(g[f"pred_{tag}_tape"])
    if kind == "ddpm":
        ex = Experiments("small", PARAMS, SDFSampler(ldm, noise_fn=tape), repaint_n=repaint_n)
        assert ex.t_idx == 3
    else:
        ex = Experiments("small", PARAMS, DDIMSampler(ldm, 10, "uniform", 0.0, noise_fn=tape), t_idx=2, repaint_n=repaint_n)
    cond, cond_mid = torch.from_numpy(g["pred_cond"]).cuda(), torch.from_numpy(g["pred_cond_mid"]).cuda()
    orig = mask = None
    if inpaint:
        orig, mask = torch.from_numpy(g["pred_orig"]).cuda(), torch.from_numpy(g["pred_mask"]).cuda()
        keep_o, keep_m = orig.clone(), mask.clone()
    out = ex.predict(cond, cond_mid, orig=orig, mask=mask, noise=torch.from_numpy(g[f"pred_{tag}_tape0"]).cuda(), **kw)
    want = g[f"pred_{tag}_out"]
    assert tuple(out.shape) == want.shape
    assert np.abs(out.cpu().numpy() - want).max() < 1e-3
    assert tape.i == len(g[f"pred_{tag}_tape"])          # same number of draws, same shapes, same order as the reference
    if inpaint:   # the caller's tensors are not edited (the product clones where the reference writes through views)
        assert torch.equal(orig, keep_o) and torch.equal(mask, keep_m)


def test_dummy_cond_input_and_masks_on_device(golden):
    g = golden("orchestration.npz")
    for ct in ("chord", "txt"):
        p = Params(dict(img_h=128, img_w=128, cond_type=ct, chd_n_step=32, chd_input_dim=36))
        outs = dummy_cond_input(3, p)
        want = g[f"dummy_{ct}_shapes"]
        for i, o in enumerate(outs):
            if want[i][0] < 0:
                assert o is None
            else:
                assert o.is_cuda and list(o.shape) == [int(v) for v in want[i][: o.dim()]] and float(o.abs().sum()) == 0.0
    for case in ("dense", "sparse", "one"):
        for kind in ("remaining", "below", "above"):
            got = get_mask(torch.from_numpy(g[f"mask_orig_{case}"].copy()).cuda(), kind)
            assert np.array_equal(got.cpu().numpy(), g[f"mask_{kind}_{case}"]), (case, kind)


def test_predict_songs_equals_per_song_predict(ldm):
    """Config 5: S songs x B segments; run r of every song is ONE batch of S.  Must equal S independent
    ``predict(autoreg=True)`` runs fed the same per-song noise."""
    S, B, T = 3, 3, 3
    rng = np.random.Generator(np.random.PCG64(12))
    cond = torch.from_numpy(rng.standard_normal((S, B, 1, 32)).astype(np.float32)).cuda()
    cond_mid = torch.from_numpy(rng.standard_normal((S, B, 1, 32)).astype(np.float32)).cuda()
    noise = torch.from_numpy(rng.standard_normal((S, B, 2, 16, 16)).astype(np.float32)).cuda()
    orig = torch.from_numpy((rng.random((S, B, 2, 16, 16)) < 0.1).astype(np.float32)).cuda()
    mask = torch.zeros(S, B, 2, 16, 16).cuda()
    mask[:, :, :, 12:] = 1
    n_draws = (2 * B - 1) * T * 2
    draws = rng.standard_normal((n_draws, S, 2, 16, 16)).astype(np.float32)
    ex = Experiments("small", PARAMS, SDFSampler(ldm, noise_fn=Tape(draws)), t_idx=T)
    got = ex.predict_songs(cond, cond_mid, uncond_scale=2.0, orig=orig, mask=mask, noise=noise)
    assert got.shape == (S, 2 * B, 2, 8, 16)
    for s in range(S):
        ex1 = Experiments("small", PARAMS, SDFSampler(ldm, noise_fn=Tape(draws[:, s:s + 1])), t_idx=T)
        one = ex1.predict(cond[s], cond_mid[s], uncond_scale=2.0, autoreg=True, orig=orig[s], mask=mask[s], noise=noise[s])
        assert (got[s] - one).abs().max() < 1e-4, s     # batch composition changes the tile choice, not the arithmetic


def test_predict_songs_on_device_noise_is_shard_invariant(ldm):
    """Songs sharded over ranks (sample_offset = first song of the rank) draw what the unsharded run draws."""
    S, B, T = 4, 2, 2
    rng = np.random.Generator(np.random.PCG64(13))
    cond = torch.from_numpy(rng.standard_normal((S, B, 1, 32)).astype(np.float32)).cuda()
    cond_mid = cond.roll(1, 1).contiguous()

    def run(lo, hi):
        ex = Experiments("small", PARAMS, SDFSampler(ldm, seed=5, sample_offset=lo), t_idx=T)
        return ex.predict_songs(cond[lo:hi].contiguous(), cond_mid[lo:hi].contiguous())

    full = run(0, S)
    halves = torch.cat([run(0, 2), run(2, 4)])
    assert torch.equal(halves, torch.cat([run(0, 2), run(2, 4)]))      # bit-reproducible
    assert (halves - full).abs().max() < 1e-4 and full.std() > 0


# ---- the concat_blurry variant (ref:inference_sdf.py:797-803, sampler_sdf.py:108-118): cond_concat through predict / predict_songs ----
SMALL4 = UNetConfig(in_channels=4, out_channels=2, channels=32, n_res_blocks=1, attention_levels=(1,),
                    channel_multipliers=(1, 2), n_heads=2, tf_layers=1, d_cond=32)
CONCAT_CASES = {
    "plain": ("ddpm", dict(), False, 3),
    "inp_cfg": ("ddpm", dict(uncond_scale=2.0), True, 3),
    "autoreg1": ("ddpm", dict(autoreg=True), False, 1),
    "ddim": ("ddim", dict(uncond_scale=3.0), True, 3),
}


@pytest.fixture(scope="module", params=["f32", "bf16x3"])
def ldm4(request):
    _lib.require_gpu()
    m = UNetModel(in_channels=4, out_channels=2, channels=32, n_res_blocks=1, attention_levels=(1,),
                  channel_multipliers=(1, 2), n_heads=2, tf_layers=1, d_cond=32, img_h=16, img_w=16)
    m.load_state_dict(synth_unet_state(SMALL4, 0))
    m.set_precision(request.param)
    return LatentDiffusion(m, None, 0.18215, 1000, *LIN)


@pytest.mark.parametrize("tag", list(CONCAT_CASES))
def test_predict_with_cond_concat_vs_reference_golden(ldm4, golden, tag):
    from polyffusion_amd.inference_sdf import get_blurry_image
    g = golden("orchestration_concat.npz")
    kind, kw, inpaint, n = CONCAT_CASES[tag]
    tape = Tape(g[f"{tag}_tape"])
    if kind == "ddpm":
        ex = Experiments("small", PARAMS, SDFSampler(ldm4, noise_fn=tape))
    else:
        ex = Experiments("small", PARAMS, DDIMSampler(ldm4, 10, "uniform", 0.0, noise_fn=tape), t_idx=2)
    cc = get_blurry_image(torch.from_numpy(g["image"]).cuda(), 0.25)
    orig = mask = None
    if inpaint:
        orig, mask = torch.from_numpy(g["orig"]).cuda(), torch.from_numpy(g["mask"]).cuda()
    out = ex.predict(torch.from_numpy(g["cond"])[:n].cuda(), torch.from_numpy(g["cond_mid"])[:n].cuda(), orig=orig, mask=mask,
                     cond_concat=cc[:n], noise=torch.from_numpy(g[f"{tag}_tape0"]).cuda(), **kw)
    want = g[f"{tag}_out"]
    assert tuple(out.shape) == want.shape and np.abs(out.cpu().numpy() - want).max() < 1e-3
    assert tape.i == len(g[f"{tag}_tape"])


def test_predict_songs_carries_cond_concat(ldm4, golden):
    """The batched multi-song driver with cond_concat: S one-segment songs equal S independent predict(autoreg=True) runs; a multi-segment
    song is refused with torch.cat's message, which is what the reference's own predict raises (fixture: autoreg3_fails)."""
    from polyffusion_amd.inference_sdf import get_blurry_image
    g = golden("orchestration_concat.npz")
    S, T = 3, 3
    rng = np.random.Generator(np.random.PCG64(21))
    cond = torch.from_numpy(g["cond"]).cuda().unsqueeze(1)             # [S, 1, 1, 32]
    cond_mid = torch.from_numpy(g["cond_mid"]).cuda().unsqueeze(1)
    cc = get_blurry_image(torch.from_numpy(g["image"]).cuda(), 0.25).unsqueeze(1)   # [S, 1, 2, 16, 16]
    noise = torch.from_numpy(rng.standard_normal((S, 1, 2, 16, 16)).astype(np.float32)).cuda()
    draws = rng.standard_normal((T * 2, S, 2, 16, 16)).astype(np.float32)
    ex = Experiments("small", PARAMS, SDFSampler(ldm4, noise_fn=Tape(draws)), t_idx=T)
    got = ex.predict_songs(cond, cond_mid, uncond_scale=2.0, cond_concat=cc, noise=noise)
    assert got.shape == (S, 2, 2, 8, 16)
    for s in range(S):
        ex1 = Experiments("small", PARAMS, SDFSampler(ldm4, noise_fn=Tape(draws[:, s:s + 1])), t_idx=T)
        one = ex1.predict(cond[s], cond_mid[s], uncond_scale=2.0, autoreg=True, cond_concat=cc[s], noise=noise[s])
        assert (got[s] - one).abs().max() < 1e-4, s
    assert int(g["autoreg3_fails"]) == 1
    with pytest.raises(RuntimeError, match="Sizes of tensors must match"):
        ex.predict_songs(cond.transpose(0, 1).contiguous(), cond_mid.transpose(0, 1).contiguous(), cond_concat=cc.transpose(0, 1).contiguous())
