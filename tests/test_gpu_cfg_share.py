"""Classifier-free guidance with a shared prefix (pf_unet_forward_cfg, UNetModel.forward(shared_x=True); -m gpu).

`get_eps` (ref:stable_diffusion/sampler/__init__.py:63-77) evaluates the denoiser on cat([x, x]), cat([t, t]), cat([uncond_cond, cond]).  The
condition reaches the UNet only through the cross-attention of its transformer blocks (ref:unet.py:181-196, unet_attention.py:240-246), so
every layer in front of the first SpatialTransformer computes the same thing for both halves; the plan evaluates those once and shares the skip
tensors.  Checked: equal to the plain evaluation of the concatenated batch (up to the tile choices of the smaller prefix batch), against the CPU
oracle, bit-equal halves for equal conditions, a UNet without attention, the prepared / hipGraph paths of the samplers, both arithmetic modes."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import unet_ref  # noqa: E402
from polyffusion_amd import _lib, synth  # noqa: E402
from polyffusion_amd.arch import UNetConfig  # noqa: E402
from polyffusion_amd.inference_sdf import synthetic_model  # noqa: E402
from polyffusion_amd.params import preset  # noqa: E402
from polyffusion_amd.sampler import DDIMSampler, SDFSampler  # noqa: E402
from polyffusion_amd.unet import LatentDiffusion, UNetModel  # noqa: E402
from polyffusion_amd.weights import synth_unet_state  # noqa: E402


@pytest.fixture(scope="module")
def chd8bar():
    _lib.require_gpu()
    return synthetic_model(preset("sdf_chd8bar"))


def small(levels=(1,), d_cond=32, n_heads=2):
    kw = dict(in_channels=2, out_channels=2, channels=32, n_res_blocks=1, attention_levels=levels, channel_multipliers=(1, 2), n_heads=n_heads,
              tf_layers=1, d_cond=d_cond)
    m = UNetModel(img_h=32, img_w=32, **kw)
    cfg = UNetConfig(**kw)
    m.load_state_dict(synth_unet_state(cfg, 0))
    return m, cfg


@pytest.mark.parametrize("precision,B,tol", [("f32", 3, 2e-5), ("bf16x3", 3, 1e-4), ("bf16x3", 16, 1e-4)])
def test_shared_prefix_equals_the_concatenated_batch_full_size(chd8bar, precision, B, tol):
    u = chd8bar.ldm.eps_model
    u.set_precision(precision)
    try:
        x = torch.from_numpy(synth.gaussian((B, 2, 128, 128), 900 + B)).cuda()
        c = chd8bar._encode_chord(torch.from_numpy(synth.chords(B, 901)).cuda())
        uc = -torch.ones_like(c)
        t = torch.tensor([999, 500, 3, 17] * 4)[:B].cuda()
        t2, c2 = torch.cat([t, t]), torch.cat([uc, c])
        ref = u(torch.cat([x, x]), t2, c2).clone()
        got = u(x, t2, c2, shared_x=True).clone()
        d = (got - ref).abs().max().item()
        print(f"shared prefix [{precision}, B={B}] vs concatenated batch: {d:.2e}")
        assert d <= tol                                             # same kernels; the half-size prefix may pick other tiles (summation order)
        assert not torch.equal(got[:B], got[B:])                    # the halves do differ (conditions differ)
        # with the hoisted prefix as the samplers pass it
        table, cross = u.prepare_time(1001), u.prepare_cond(c2)
        assert torch.equal(u(x, t2, c2, shared_x=True, time_table=table, cross_bias=cross), got)
        assert torch.equal(u(x, t2, c2, shared_x=True), got)        # bit-reproducible
        # equal conditions -> bit-equal halves (both read the same shared skips through the modulo index)
        cc = torch.cat([c, c])
        same = u(x, t2, cc, shared_x=True)
        assert torch.equal(same[:B], same[B:])
        # fewer launches' worth of work: the plan of the shared form has the prefix once (+ the four copies of the hand-over)
        assert u.n_launches(2 * B, shared_x=True) == u.n_launches(2 * B) + 4
    finally:
        u.set_precision("f32")


def test_shared_prefix_vs_oracle_full_size(chd8bar):
    u = chd8bar.ldm.eps_model
    u.set_precision("bf16x3")
    try:
        B = 2
        x = torch.from_numpy(synth.gaussian((B, 2, 128, 128), 77)).cuda()
        c = chd8bar._encode_chord(torch.from_numpy(synth.chords(B, 78)).cuda())
        t = torch.tensor([640, 2]).cuda()
        t2, c2 = torch.cat([t, t]), torch.cat([-torch.ones_like(c), c])
        got = u(x, t2, c2, shared_x=True).cpu()
        cfg = UNetConfig(d_cond=512)
        w = unet_ref.to_torch(synth_unet_state(cfg, 0))
        torch.set_num_threads(min(32, torch.get_num_threads()))
        with torch.no_grad():
            ref = unet_ref.unet_forward(w, cfg, torch.cat([x, x]).cpu(), t2.cpu(), c2.cpu())
        err = (got - ref).abs().max().item()
        print("shared-prefix guidance evaluation vs oracle:", err)
        assert err < 5e-4
    finally:
        u.set_precision("f32")


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
@pytest.mark.parametrize("levels", [(1,), (0, 1), ()])
def test_shared_prefix_small_unets(levels, precision):
    """attention at the last level / at every level (the prefix is the stem + one ResBlock) / at no level (the middle block always has its
    SpatialTransformer, unet.py:121-125: the whole down path is shared)."""
    m, cfg = small(levels, n_heads=1 if 0 in levels else 2)      # d_head must be 32 or 64
    m.set_precision(precision)
    g = torch.Generator().manual_seed(5)
    B = 3
    x, c = torch.randn(B, 2, 32, 32, generator=g).cuda(), torch.randn(B, 1, 32, generator=g).cuda()
    t = torch.tensor([7, 999, 400]).cuda()
    t2, c2 = torch.cat([t, t]), torch.cat([-torch.ones_like(c), c])
    ref = m(torch.cat([x, x]), t2, c2).clone()
    got = m(x, t2, c2, shared_x=True)
    assert (got - ref).abs().max().item() <= (2e-5 if precision == "f32" else 1e-4)
    assert not torch.equal(got[:B], got[B:])
    w = unet_ref.to_torch(synth_unet_state(cfg, 0))
    with torch.no_grad():
        o = unet_ref.unet_forward(w, cfg, torch.cat([x, x]).cpu(), t2.cpu(), c2.cpu())
    assert (got.cpu() - o).abs().max().item() < (1e-4 if precision == "f32" else 5e-4)
    with pytest.raises(RuntimeError):
        m(x, t, c2, shared_x=True)                     # t must carry both halves


@pytest.mark.parametrize("graph", [False, True])
def test_samplers_with_and_without_the_shared_prefix(graph):
    """DDIM paint with guidance 5 (eager and captured step) and one DDPM p_sample: share_cfg_prefix on == off up to rounding."""
    m, _ = small()
    g = torch.Generator().manual_seed(9)
    x, c = torch.randn(2, 2, 32, 32, generator=g).cuda(), torch.randn(2, 1, 32, generator=g).cuda()
    uc = -torch.ones_like(c)
    outs = []
    for share in (True, False):
        d = DDIMSampler(LatentDiffusion(m), 20, "uniform", 0.0, seed=5, graph=graph)
        d.share_cfg_prefix = share
        outs.append(d.paint(x, c, 6, uncond_scale=5.0, uncond_cond=uc))
    assert (outs[0] - outs[1]).abs().max().item() < 1e-4 and torch.isfinite(outs[0]).all()
    s1, s0 = SDFSampler(LatentDiffusion(m), seed=3), SDFSampler(LatentDiffusion(m), seed=3)
    s0.share_cfg_prefix = False
    a, _, ea = s1.p_sample(x, c, None, 500, uncond_scale=2.0, uncond_cond=uc)
    b, _, eb = s0.p_sample(x, c, None, 500, uncond_scale=2.0, uncond_cond=uc)
    assert (ea - eb).abs().max().item() < 2e-5 and (a - b).abs().max().item() < 2e-5


def test_conv_second_source_shared_between_batch_halves():
    """pf_conv_args.x1_bmod at op level: a conv over concat(x0 [2B], x1 [B]) with x1_bmod = B equals the conv over concat(x0, cat([x1, x1])),
    bit for bit, in both modes, with the GroupNorm finalize folded in and with the fused skip projection."""
    from test_gpu_bf16x3 import pack3
    from test_gpu_ops import dev, gn_scale_shift, nhwc, pack_w, rnd, run_conv
    lib = _lib.load()
    B, H, W, c0, c1, cout = 3, 16, 16, 64, 32, 64
    x0, x1 = dev(nhwc(rnd((2 * B, c0, H, W), 1))), dev(nhwc(rnd((B, c1, H, W), 2)))
    x1d = torch.cat([x1, x1]).contiguous()
    w = rnd((cout, c0 + c1, 3, 3), 3, (1.0 / ((c0 + c1) * 9)) ** 0.5)
    gamma, beta = dev(1 + 0.1 * rnd((c0 + c1,), 4)), dev(0.1 * rnd((c0 + c1,), 5))
    sc, sh = gn_scale_shift(lib, x0, x1d, gamma, beta, 1e-5)
    ws, bs = rnd((cout, c0 + c1, 1, 1), 6, (1.0 / (c0 + c1)) ** 0.5), dev(rnd((cout,), 7, 0.1))
    for prec, wp in ((1, pack3(lib, w)), (0, pack_w(lib, w))):
        kw = dict(x0=x0, c0=c0, c1=c1, batch=2 * B, hin=H, win=W, ks=3, stride=1, ups=0, w=wp, n=cout, prologue=1, sc=sc, sh=sh, ld_out=cout,
                  precision=prec)
        if prec == 1:
            kw.update(skip_x0=x0, skip_c0=c0, skip_c1=c1, skip_w=pack3(lib, ws), skip_bias=bs)
        full, shared = torch.empty(2 * B, H, W, cout, device="cuda"), torch.empty(2 * B, H, W, cout, device="cuda")
        run_conv(lib, out=full, x1=x1d, **(dict(kw, skip_x1=x1d) if prec == 1 else kw))
        run_conv(lib, out=shared, x1=x1, x1_bmod=B, **(dict(kw, skip_x1=x1) if prec == 1 else kw))
        assert torch.equal(full, shared), f"precision {prec}: {(full - shared).abs().max().item():.2e}"
