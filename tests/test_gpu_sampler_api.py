"""Row a19 and the seldom-used sampler arguments (-m gpu): ``SDFSampler.sample`` / ``DDIMSampler.sample``,
``repeat_noise``, ``temperature``, ``cond_concat`` and DDIM ``paint(orig_noise=None)`` against vectors recorded from the
REAL reference samplers with a noise tape (``tools/make_goldens_sample.py`` -> ``tests/golden/sample.npz``)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from polyffusion_amd import _lib  # noqa: E402
from polyffusion_amd.arch import UNetConfig  # noqa: E402
from polyffusion_amd.sampler import DDIMSampler, SDFSampler  # noqa: E402
from polyffusion_amd.unet import LatentDiffusion, UNetModel  # noqa: E402
from polyffusion_amd.weights import synth_unet_state  # noqa: E402

LIN = (0.00085, 0.012)
TOL = 1e-3   # multi-step trajectory bar; observed ~1e-5


def make_ldm(in_channels, precision):
    _lib.require_gpu()
    cfg = UNetConfig(in_channels=in_channels, out_channels=2, channels=32, n_res_blocks=1, attention_levels=(1,),
                     channel_multipliers=(1, 2), n_heads=2, tf_layers=1, d_cond=32)
    m = UNetModel(in_channels=in_channels, out_channels=2, channels=32, n_res_blocks=1, attention_levels=(1,),
                  channel_multipliers=(1, 2), n_heads=2, tf_layers=1, d_cond=32, img_h=16, img_w=16)
    m.load_state_dict(synth_unet_state(cfg, 0))
    m.set_precision(precision)
    return LatentDiffusion(m, None, 0.18215, 1000, *LIN)


@pytest.fixture(scope="module", params=["f32", "bf16x3"])
def ldm(request):
    return make_ldm(2, request.param)


@pytest.fixture(scope="module", params=["f32", "bf16x3"])
def ldm_cc(request):
    return make_ldm(3, request.param)


class Tape:
    """Serves the recorded draws in order and checks that the product asks for the reference's shapes."""

    def __init__(self, g, tag):
        self.draws = [g[f"{tag}_draw{i}"] for i in range(int(g[f"{tag}_n_draws"]))]
        self.i = 0

    def __call__(self, shape):
        a = self.draws[self.i]
        self.i += 1
        assert tuple(a.shape) == tuple(shape), (a.shape, shape)
        return torch.from_numpy(a)

    def done(self):
        return self.i == len(self.draws)


def check(out, g, tag, tape=None):
    err = np.abs(out.cpu().numpy() - g[f"{tag}_out"]).max()
    assert err < TOL, (tag, err)
    assert tape is None or tape.done(), (tag, tape.i, len(tape.draws))


def test_sdf_sample(ldm, golden):
    g = golden("sample.npz")
    cond, x_last = torch.from_numpy(g["cond"]).cuda(), torch.from_numpy(g["x_last"]).cuda()
    uc = -torch.ones(2, 1, 32).cuda()
    tape = Tape(g, "sdf_a")   # x_T drawn inside; one [1,C,H,W] draw per step (repeat_noise), none at step 0
    out = SDFSampler(ldm, noise_fn=tape).sample([2, 2, 16, 16], cond, repeat_noise=True, temperature=0.7, uncond_scale=2.0,
                                                uncond_cond=uc, t_start=995)
    check(out, g, "sdf_a", tape)
    tape = Tape(g, "sdf_b")
    out = SDFSampler(ldm, noise_fn=tape).sample([2, 2, 16, 16], cond, x_last=x_last.clone(), t_start=996)
    check(out, g, "sdf_b", tape)


def test_ddim_sample(ldm, golden):
    g = golden("sample.npz")
    cond, x_last = torch.from_numpy(g["cond"]).cuda(), torch.from_numpy(g["x_last"]).cuda()
    uc = -torch.ones(2, 1, 32).cuda()
    tape = Tape(g, "ddim_c")
    out = DDIMSampler(ldm, 10, "uniform", 1.0, noise_fn=tape).sample([2, 2, 16, 16], cond, repeat_noise=True, temperature=1.3,
                                                                    uncond_scale=0.0, uncond_cond=uc, t_start=6)
    check(out, g, "ddim_c", tape)
    tape = Tape(g, "ddim_d")
    out = DDIMSampler(ldm, 10, "quad", 0.0, noise_fn=tape).sample([2, 2, 16, 16], cond, x_last=x_last.clone(), uncond_scale=4.0,
                                                                 uncond_cond=uc, t_start=5)
    check(out, g, "ddim_d", tape)


def test_ddim_paint_without_fixed_orig_noise(ldm, golden):
    g = golden("sample.npz")
    cond, x_last, orig, mask = (torch.from_numpy(g[k]).cuda() for k in ("cond", "x_last", "orig", "mask"))
    tape = Tape(g, "ddim_e")   # per step: the sigma draw, then the fresh known-region draw
    out = DDIMSampler(ldm, 10, "uniform", 0.5, noise_fn=tape).paint(x_last.clone(), cond, 3, orig=orig, mask=mask, orig_noise=None,
                                                                   uncond_scale=1.0, uncond_cond=-torch.ones(2, 1, 32).cuda())
    check(out, g, "ddim_e", tape)


def test_cond_concat(ldm_cc, golden):
    g = golden("sample.npz")
    cond, x_last, orig, mask, cc = (torch.from_numpy(g[k]).cuda() for k in ("cond", "x_last", "orig", "mask", "cond_concat"))
    uc = -torch.ones(2, 1, 32).cuda()
    tape = Tape(g, "sdf_f")
    out = SDFSampler(ldm_cc, noise_fn=tape).paint(x_last.clone(), cond, 3, orig=orig, mask=mask, uncond_scale=2.5, uncond_cond=uc,
                                                  cond_concat=cc)
    check(out, g, "sdf_f", tape)
    tape = Tape(g, "sdf_f2")   # orig=None branch
    out = SDFSampler(ldm_cc, noise_fn=tape).paint(x_last.clone(), cond, 2, uncond_scale=1.0, uncond_cond=uc, cond_concat=cc)
    check(out, g, "sdf_f2", tape)
    out = DDIMSampler(ldm_cc, 10, "uniform", 0.0).paint(x_last.clone(), cond, 3, orig=orig, mask=mask, orig_noise=x_last,
                                                       uncond_scale=2.5, uncond_cond=uc, cond_concat=cc)
    check(out, g, "ddim_g")


def test_p_sample_returns_x0_on_request(ldm, golden):
    g = golden("sample.npz")
    cond, x = torch.from_numpy(g["cond"]).cuda(), torch.from_numpy(g["x_last"]).cuda()
    nz = torch.zeros(2, 2, 16, 16)
    s = SDFSampler(ldm, noise_fn=lambda shape: nz)
    x_prev, x0, e_t = s.p_sample(x, cond, None, 500)
    want = s.sqrt_recip_alpha_bar[500] * x - s.sqrt_recip_m1_alpha_bar[500] * e_t
    assert torch.allclose(x0, want, atol=1e-6)
    x_prev2, none, _ = s.p_sample(x, cond, None, 500, return_x0=False)
    assert none is None and torch.equal(x_prev, x_prev2)
