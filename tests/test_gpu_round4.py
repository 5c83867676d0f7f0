"""Round-4 additions (-m gpu): the hoisted step-invariant prefix (pf_unet_prepare_time / pf_unet_prepare_cond /
pf_unet_forward_prepared), the update kernels that draw their noise themselves (pf_ddpm_step_rng / pf_ddim_step_rng and the
device-state forms), the per-handle plan options (pf_unet_set_option), the 16x16-pixel conv tile at op level, and BASELINE
configs[1] / configs[2] checked against the CPU oracle on ALL samples (configs[2]: a first / middle / last subset)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import sampler_ref, unet_ref  # noqa: E402
from polyffusion_amd import _lib, synth  # noqa: E402
from polyffusion_amd.arch import UNetConfig  # noqa: E402
from polyffusion_amd.inference_sdf import synthetic_model  # noqa: E402
from polyffusion_amd.params import preset  # noqa: E402
from polyffusion_amd.sampler import DDIMSampler, SDFSampler  # noqa: E402
from polyffusion_amd.unet import LatentDiffusion, UNetModel  # noqa: E402
from polyffusion_amd.weights import synth_unet_state  # noqa: E402
from test_gpu_bf16x3 import pack3  # noqa: E402
from test_gpu_ops import dev, gn_scale_shift, nhwc, rnd, run_conv  # noqa: E402

LIN = (0.00085, 0.012)


@pytest.fixture(scope="module")
def lib():
    _lib.require_gpu()
    return _lib.load()


@pytest.fixture(scope="module")
def chd8bar():
    _lib.require_gpu()
    m = synthetic_model(preset("sdf_chd8bar"))
    m.ldm.eps_model.set_precision("bf16x3")
    return m


def small_unet(d_cond=32):
    cfg = UNetConfig(in_channels=2, out_channels=2, channels=32, n_res_blocks=1, attention_levels=(1,), channel_multipliers=(1, 2),
                     n_heads=2, tf_layers=1, d_cond=d_cond)
    m = UNetModel(in_channels=2, out_channels=2, channels=32, n_res_blocks=1, attention_levels=(1,), channel_multipliers=(1, 2),
                  n_heads=2, tf_layers=1, d_cond=d_cond, img_h=32, img_w=32)
    m.load_state_dict(synth_unet_state(cfg, 0))
    return m, cfg


# ---------------------------------------------------------------------------------------------- hoisted prefix
@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_prepared_forward_is_bit_identical(chd8bar, precision):
    """pf_unet_forward_prepared with the time table and the collapsed cross-attention biases supplied == pf_unet_forward that
    computes them per call: same kernels on the same values, so the bits must agree (full sdf_chd8bar, per-sample t values)."""
    u = chd8bar.ldm.eps_model
    u.set_precision(precision)
    try:
        B = 5
        g = torch.Generator().manual_seed(11)
        x = torch.randn(B, 2, 128, 128, generator=g).cuda()
        t = torch.tensor([999, 0, 17, 500, 998]).cuda()
        c = torch.randn(B, 1, 512, generator=g).cuda()
        ref = u(x, t, c).clone()
        table = u.prepare_time(1001)
        cross = u.prepare_cond(c)
        assert table.shape == (1001, u._lib.pf_unet_time_bias_width(u._h)) and cross.shape == (B, u._lib.pf_unet_cross_bias_width(u._h))
        assert torch.equal(u(x, t, c, time_table=table, cross_bias=cross), ref)
        assert torch.equal(u(x, t, c, time_table=table), ref)          # either part alone
        assert torch.equal(u(x, t, c, cross_bias=cross), ref)
        # the prepared plan issues four launches less
        assert u.n_launches(B) - u.n_launches(B, prepared=True) == 4
    finally:
        u.set_precision("bf16x3")


def test_prepared_forward_general_cross_attention_keeps_working():
    """n_cond > 1 has no collapsed form: prepare_cond answers None and the time table alone is used."""
    m = synthetic_model(preset("sdf_txtvnl"))
    u = m.ldm.eps_model
    u.set_precision("bf16x3")
    cond = m._encode_txt(torch.from_numpy(synth.prmat(2, 32)).cuda())
    assert cond.shape[1] > 1 and u.prepare_cond(cond) is None
    x = torch.from_numpy(synth.gaussian((2, 2, 128, 128), 55)).cuda()
    t = torch.tensor([0, 640]).cuda()
    assert torch.equal(u(x, t, cond, time_table=u.prepare_time(1001)), u(x, t, cond))
    with pytest.raises(AssertionError):
        u(x, t, cond, cross_bias=torch.zeros(2, 8, device="cuda"))


def test_prepared_rejects_bad_arguments(lib):
    m, _ = small_unet()
    with pytest.raises(RuntimeError):
        buf = torch.empty(8, device="cuda")
        _lib.check(lib.pf_unet_prepare_time(m._h, 10, buf.data_ptr(), buf.data_ptr(), 4, _lib.current_stream()))   # scratch too small


@pytest.mark.parametrize("graph", [False, True])
def test_paint_with_hoisting_equals_reference_trajectory(graph):
    """SDFSampler.paint (which now prepares once and draws inside the update kernel) against the oracle's loop fed the SAME
    noise: the device generator's draws are read back with pf_randn and handed to the oracle as its tape."""
    m, cfg = small_unet()
    w = unet_ref.to_torch(synth_unet_state(cfg, 0))
    B, T = 3, 4
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, 2, 32, 32, generator=g)
    c = torch.randn(B, 1, 32, generator=g)
    orig = torch.randn(B, 2, 32, 32, generator=g)
    mask = (torch.rand(B, 2, 32, 32, generator=g) > 0.5).float()
    s = SDFSampler(LatentDiffusion(m), seed=77, graph=graph)
    got = s.paint(x.cuda(), c.cuda(), T, orig=orig.cuda(), mask=mask.cuda()).cpu()
    assert s._draws == 2 * T
    lib = _lib.load()
    tape = []
    for d in range(2 * T):
        z = torch.empty(B, 2, 32, 32, device="cuda")
        _lib.check(lib.pf_randn(z.data_ptr(), z.numel(), 77, d, 0, _lib.current_stream()))
        tape.append(z.cpu())
    it = iter(tape)
    rs = sampler_ref.SDFSamplerRef(lambda x_, t_, c_: unet_ref.unet_forward(w, cfg, x_, t_, c_), 1000, *LIN, noise_fn=lambda shp: next(it))
    ref = rs.paint(x, c, T, orig=orig, mask=mask)
    assert (got - ref).abs().max().item() < 1e-3


# ---------------------------------------------------------------------------------------------- noise inside the update kernels
@pytest.mark.parametrize("with_orig", [True, False])
def test_ddpm_step_rng_equals_randn_plus_step(lib, with_orig):
    n, seed, off, dq, dp = 16 * 2 * 128 * 128, 1234, 3 * 32768, 40, 41
    g = torch.Generator().manual_seed(1)
    x, eps = torch.randn(n, generator=g).cuda(), torch.randn(n, generator=g).cuda()
    orig = torch.randn(n, generator=g).cuda() if with_orig else None
    mask = (torch.rand(n, generator=g) > 0.3).float().cuda() if with_orig else None
    coef = _lib.DdpmCoef(1.3, 0.7, 0.2, 0.8, 0.05, 0.9, 0.43)
    st = _lib.current_stream()
    nq, npz = torch.empty(n, device="cuda"), torch.empty(n, device="cuda")
    _lib.check(lib.pf_randn(nq.data_ptr(), n, seed, dq, off, st))
    _lib.check(lib.pf_randn(npz.data_ptr(), n, seed, dp, off, st))
    ref = torch.empty(n, device="cuda")
    _lib.check(lib.pf_ddpm_step(x.data_ptr(), eps.data_ptr(), npz.data_ptr(), nq.data_ptr() if with_orig else None, _lib.ptr(orig), _lib.ptr(mask),
                                C.byref(coef), ref.data_ptr(), n, st))
    out = torch.empty(n, device="cuda")
    _lib.check(lib.pf_ddpm_step_rng(x.data_ptr(), eps.data_ptr(), _lib.ptr(orig), _lib.ptr(mask), C.byref(coef), seed, dq, dp, off, out.data_ptr(), n, st))
    assert torch.equal(out, ref)
    # in place (the graph path updates x where it stands)
    xi = x.clone()
    _lib.check(lib.pf_ddpm_step_rng(xi.data_ptr(), eps.data_ptr(), _lib.ptr(orig), _lib.ptr(mask), C.byref(coef), seed, dq, dp, off, xi.data_ptr(), n, st))
    assert torch.equal(xi, ref)
    # device-state form: draws (q, p) = (state.draws, state.draws + 1) with a known region, p = state.draws without
    table = torch.tensor([[0.0] * 7, [1.3, 0.7, 0.2, 0.8, 0.05, 0.9, 0.43]], device="cuda")
    state = torch.zeros(2, dtype=torch.int64, device="cuda")
    _lib.check(lib.pf_step_state_set(state.data_ptr(), 1, dq if with_orig else dp, st))
    out2 = torch.empty(n, device="cuda")
    _lib.check(lib.pf_ddpm_step_rng_dev(x.data_ptr(), eps.data_ptr(), _lib.ptr(orig), _lib.ptr(mask), table.data_ptr(), state.data_ptr(), seed, off,
                                        out2.data_ptr(), n, st))
    assert torch.equal(out2, ref)
    # whole Philox groups only
    with pytest.raises(RuntimeError):
        _lib.check(lib.pf_ddpm_step_rng(x.data_ptr(), eps.data_ptr(), None, None, C.byref(coef), seed, dq, dp, off + 2, out.data_ptr(), n, st))


@pytest.mark.parametrize("with_orig", [True, False])
def test_ddim_step_rng_equals_randn_plus_step(lib, with_orig):
    n, seed, off, d = 4 * 2 * 128 * 128, 99, 32768, 7
    g = torch.Generator().manual_seed(3)
    x, eps = torch.randn(n, generator=g).cuda(), torch.randn(n, generator=g).cuda()
    orig = torch.randn(n, generator=g).cuda() if with_orig else None
    on = torch.randn(n, generator=g).cuda() if with_orig else None
    mask = (torch.rand(n, generator=g) > 0.3).float().cuda() if with_orig else None
    coef = _lib.DdimCoef(0.6, 0.8, 0.85, 0.5, 0.1, 0.8, 0.6)
    st = _lib.current_stream()
    nz = torch.empty(n, device="cuda")
    _lib.check(lib.pf_randn(nz.data_ptr(), n, seed, d, off, st))
    ref = torch.empty(n, device="cuda")
    _lib.check(lib.pf_ddim_step(x.data_ptr(), eps.data_ptr(), nz.data_ptr(), _lib.ptr(orig), _lib.ptr(on), _lib.ptr(mask), C.byref(coef), ref.data_ptr(), n, st))
    out = torch.empty(n, device="cuda")
    _lib.check(lib.pf_ddim_step_rng(x.data_ptr(), eps.data_ptr(), _lib.ptr(orig), _lib.ptr(on), _lib.ptr(mask), C.byref(coef), seed, d, off, out.data_ptr(), n, st))
    assert torch.equal(out, ref)
    table = torch.tensor([[0.6, 0.8, 0.85, 0.5, 0.1, 0.8, 0.6]], device="cuda")
    state = torch.zeros(2, dtype=torch.int64, device="cuda")
    _lib.check(lib.pf_step_state_set(state.data_ptr(), 0, d, st))
    out2 = torch.empty(n, device="cuda")
    _lib.check(lib.pf_ddim_step_rng_dev(x.data_ptr(), eps.data_ptr(), _lib.ptr(orig), _lib.ptr(on), _lib.ptr(mask), table.data_ptr(), state.data_ptr(), seed,
                                        off, out2.data_ptr(), n, st))
    assert torch.equal(out2, ref)


def test_ddim_eta1_paint_graph_equals_eager():
    """eta = 1: every step draws; the eager loop (pf_ddim_step_rng) and the captured step (pf_ddim_step_rng_dev) agree bit for bit."""
    m, _ = small_unet()
    g = torch.Generator().manual_seed(4)
    x, c = torch.randn(2, 2, 32, 32, generator=g).cuda(), torch.randn(2, 1, 32, generator=g).cuda()
    outs = []
    for graph in (False, True):
        d = DDIMSampler(LatentDiffusion(m), 20, "uniform", 1.0, seed=5, graph=graph)
        outs.append(d.paint(x, c, 6))
        assert d._draws == 7
    assert torch.equal(outs[0], outs[1])


# ---------------------------------------------------------------------------------------------- plan options
def test_plan_options_fused_mlp_and_attention_forms(chd8bar):
    """Whole UNet at the bench shape (B = 16, bf16x3): the fused feed-forward launch on == off (up to the statistics tiling),
    the 256-query attention == the 128-query one up to summation order, the 16x16-pixel conv tile == the 8x16 one likewise."""
    u = chd8bar.ldm.eps_model
    B = 16
    x = torch.from_numpy(synth.gaussian((B, 2, 128, 128), 1234)).cuda()
    c = chd8bar._encode_chord(torch.from_numpy(synth.chords(B, 4242)).cuda())
    t = torch.full((B,), 999, dtype=torch.long, device="cuda")
    try:
        base = u(x, t, c).clone()
        n_auto = u.n_launches(B)
        u.set_option("mlp_fused", False)
        assert u.get_option("mlp_fused") is False and u.n_launches(B) > n_auto          # the unfused chain has more launches
        # the launch itself is bit-identical to the chain it replaces (tests/test_gpu_mlp_fused.py); inside the UNet it hands the next
        # GroupNorm 64-row statistics tiles where the chain's last GEMM emits 128-row ones: the fp32 partial sums differ in the last bits
        assert (u(x, t, c) - base).abs().max().item() <= 1e-4     # observed 3e-5 (the bf16x3 path itself sits 5e-5 from the reference)
        u.set_option("mlp_fused", None)
        u.set_option("attn_wide", False)
        narrow = u(x, t, c).clone()
        u.set_option("attn_wide", True)
        wide = u(x, t, c).clone()
        # (auto = 256-query form at the 32x32 level, 128-query form at the 16x16 level: neither forced run equals it bit for bit)
        assert (narrow - wide).abs().max().item() <= 1e-4 and (wide - base).abs().max().item() <= 1e-4
        u.set_option("attn_wide", None)
        u.set_option("conv_t16", False)
        assert (u(x, t, c) - base).abs().max().item() <= 1e-4
        u.set_option("conv_t16", None)
        u.set_option("conv_pp", False)             # the 32x32-level convs as 4-wave workgroups (one K range) instead of the two-group form
        assert u.get_option("conv_pp") is False and u.n_launches(B) == n_auto
        assert (u(x, t, c) - base).abs().max().item() <= 1e-4
    finally:
        for o in ("mlp_fused", "attn_wide", "conv_t16", "conv_pp"):
            u.set_option(o, None)
    assert torch.equal(u(x, t, c), base)


# ---------------------------------------------------------------------------------------------- 16x16-pixel tile, op level
@pytest.mark.parametrize("c0,c1", [(128, 64), (128, 0)])
def test_conv_16x16_pixel_tile_vs_torch(lib, c0, c1):
    """(B, H, W) = (16, 128, 128), >= 128 input channels, 64 output channels: the shape conv_pick_tile gives the 16x16-pixel x 64-channel
    tile (each wave 128 pixels x 32 channels) - against F.conv2d, and against the 8x16 tile (no_t16)."""
    B, H, W, cout = 16, 128, 128, 64
    cin = c0 + c1
    x = rnd((B, cin, H, W), 21) * 1.5 + 0.3
    w, bias = rnd((cout, cin, 3, 3), 22, (1.0 / (cin * 9)) ** 0.5), rnd((cout,), 23, 0.1)
    gamma, beta = 1 + 0.1 * rnd((cin,), 24), 0.1 * rnd((cin,), 25)
    sb, res = rnd((B, cout), 26), rnd((B, cout, H, W), 27)
    ref = F.conv2d(F.silu(F.group_norm(x, 32, gamma, beta, eps=1e-5)), w, bias, padding=1) + sb[:, :, None, None] + res
    x0 = dev(nhwc(x[:, :c0]))
    x1 = dev(nhwc(x[:, c0:])) if c1 else None
    sc, sh = gn_scale_shift(lib, x0, x1, dev(gamma), dev(beta), 1e-5)
    kw = dict(x0=x0, c0=c0, x1=x1, c1=c1, batch=B, hin=H, win=W, ks=3, stride=1, ups=0, w=pack3(lib, w), n=cout, prologue=1, sc=sc, sh=sh,
              bias=dev(bias), sbias=dev(sb), ld_sbias=cout, res=dev(nhwc(res)), ld_res=cout, ld_out=cout, precision=1)
    a = _lib.ConvArgs()
    for k, v in kw.items():
        setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else (0 if v is None else v))
    assert lib.pf_conv_stats_tiles(C.byref(a)) == (H // 16) * (W // 16)      # the 16x16-pixel tile is what this shape gets
    out = torch.empty(B, H, W, cout, device="cuda")
    stats = torch.zeros(B, (H // 16) * (W // 16), cout, 2, device="cuda")
    run_conv(lib, out=out, stats_out=stats, **kw)
    o = out.cpu()
    assert (o - nhwc(ref)).abs().max().item() < 3e-4
    tot = stats.cpu().double().sum(1)
    od = o.double()
    assert (tot[..., 0] - od.sum((1, 2))).abs().max() < 5e-2 and (tot[..., 1] - (od * od).sum((1, 2))).abs().max() < 5e-1
    out8 = torch.empty(B, H, W, cout, device="cuda")
    run_conv(lib, out=out8, no_t16=1, **kw)
    assert (out8 - out).abs().max().item() < 2e-5
    # the per-sample bias through a row index (the hoisted time table's form): row 2 * b + 1 of a table twice as tall
    table = torch.zeros(2 * B, cout)
    table[1::2] = sb
    rows = (2 * torch.arange(B) + 1).cuda()
    out_r = torch.empty(B, H, W, cout, device="cuda")
    kw2 = dict(kw, sbias=dev(table), sbias_rows=rows, sbias_nrows=2 * B)
    run_conv(lib, out=out_r, **kw2)
    assert torch.equal(out_r, out)


# ---------------------------------------------------------------------------------------------- two-group ping-pong form, op level
@pytest.mark.parametrize("c0,c1,skip", [(256, 0, 0), (256, 256, 0), (256, 128, 384)])
def test_conv_pingpong_form_vs_torch(lib, c0, c1, skip):
    """(B, H, W) = (16, 32, 32), 256 output channels: 256 tiles of 8x16 pixels x 128 channels, the shape that runs as 8-wave workgroups
    whose two wave groups split K and work half a tap apart (PF_OPT_CONV_PP) - against F.conv2d (+ the fused 1x1 skip projection of
    the ResBlock input), against the single-group form (no_pp), and twice for bit-reproducibility."""
    B, H, W, cout = 16, 32, 32, 256
    cin = c0 + c1
    x = rnd((B, cin, H, W), 31) * 1.5 + 0.3
    w, bias = rnd((cout, cin, 3, 3), 32, (1.0 / (cin * 9)) ** 0.5), rnd((cout,), 33, 0.1)
    gamma, beta = 1 + 0.1 * rnd((cin,), 34), 0.1 * rnd((cin,), 35)
    sb = rnd((B, cout), 36)
    ref = F.conv2d(F.silu(F.group_norm(x, 32, gamma, beta, eps=1e-5)), w, bias, padding=1) + sb[:, :, None, None]
    x0 = dev(nhwc(x[:, :c0]))
    x1 = dev(nhwc(x[:, c0:])) if c1 else None
    sc, sh = gn_scale_shift(lib, x0, x1, dev(gamma), dev(beta), 1e-5)
    kw = dict(x0=x0, c0=c0, x1=x1, c1=c1, batch=B, hin=H, win=W, ks=3, stride=1, ups=0, w=pack3(lib, w), n=cout, prologue=1, sc=sc, sh=sh,
              bias=dev(bias), sbias=dev(sb), ld_sbias=cout, ld_out=cout, precision=1)
    if skip:
        xs = rnd((B, skip, H, W), 37)
        ws, bs = rnd((cout, skip, 1, 1), 38, (1.0 / skip) ** 0.5), rnd((cout,), 39, 0.1)
        ref = ref + F.conv2d(xs, ws, bs)
        kw.update(skip_x0=dev(nhwc(xs)), skip_c0=skip, skip_w=pack3(lib, ws), skip_bias=dev(bs))
    else:
        res = rnd((B, cout, H, W), 40)
        ref = ref + res
        kw.update(res=dev(nhwc(res)), ld_res=cout)
    out = torch.empty(B, H, W, cout, device="cuda")
    stats = torch.zeros(B, (H // 8) * (W // 16), cout, 2, device="cuda")
    run_conv(lib, out=out, stats_out=stats, **kw)
    o = out.cpu()
    assert (o - nhwc(ref)).abs().max().item() < 3e-4
    tot = stats.cpu().double().sum(1)
    od = o.double()
    assert (tot[..., 0] - od.sum((1, 2))).abs().max() < 5e-2 and (tot[..., 1] - (od * od).sum((1, 2))).abs().max() < 5e-1
    out1 = torch.empty_like(out)
    run_conv(lib, out=out1, no_pp=1, **kw)
    d = (out1 - out).abs().max().item()
    assert 0 < d < 2e-5                              # the two forms differ (K summed in two halves), in the last bits only
    out2 = torch.empty_like(out)
    run_conv(lib, out=out2, **kw)
    assert torch.equal(out2, out)


# ---------------------------------------------------------------------------------------------- all samples vs the oracle
_ORACLE_CFG2 = {}


@pytest.fixture(scope="module")
def chd8bar_f16():
    _lib.require_gpu()
    m = synthetic_model(preset("sdf_chd8bar"), x3="f16")
    m.ldm.eps_model.set_precision("f16x3")
    return m


@pytest.mark.parametrize("precision,tol", [("bf16x3", 5e-4), ("f32", 1e-4), ("f16x3", 1e-4)])
def test_config2_all_16_samples_vs_oracle(chd8bar, chd8bar_f16, precision, tol):
    """BASELINE.json configs[1] (sdf_chd8bar, B = 16): one denoiser evaluation - through the prepared plan paint() uses - checked
    against the CPU oracle on EVERY sample.  The kernels that only exist at B >= 12-16 (fused feed-forward launch, 256-query
    attention, 16x16-pixel conv tile) are all inside it."""
    model = chd8bar_f16 if precision == "f16x3" else chd8bar      # f16x3: the same weights in the fp16-piece build of the library
    u = model.ldm.eps_model
    u.set_precision(precision)
    try:
        B = 16
        x = torch.from_numpy(synth.gaussian((B, 2, 128, 128), 1234)).cuda()
        c = chd8bar._encode_chord(torch.from_numpy(synth.chords(B, 4242)).cuda())
        t = torch.tensor([999, 998, 500, 3] * 4).cuda()
        s = SDFSampler(model.ldm, seed=1)
        eps = s.get_eps(x, t, c, uncond_scale=1.0, uncond_cond=None, prep=s.prepare(c))
        if "ref" not in _ORACLE_CFG2:                              # one oracle evaluation (the slow part) for the three modes
            w = unet_ref.to_torch(synth_unet_state(UNetConfig(d_cond=512), 0))
            torch.set_num_threads(min(32, torch.get_num_threads()))
            with torch.no_grad():
                _ORACLE_CFG2["ref"] = unet_ref.unet_forward(w, UNetConfig(d_cond=512), x.cpu(), t.cpu(), c.cpu())
        ref = _ORACLE_CFG2["ref"]
        err = (eps.cpu() - ref).abs().amax(dim=(1, 2, 3))
        print(f"config 2 [{precision}] per-sample max-abs-diff vs oracle:", [f"{e:.1e}" for e in err.tolist()])
        assert err.max().item() < tol
    finally:
        if precision != "f16x3":
            u.set_precision("bf16x3")


def test_config3_cfg5_batch32_first_middle_last_vs_oracle(chd8bar):
    """BASELINE.json configs[2]: one DDIM step with uncond_scale 5 at B = 32 (a 64-sample evaluation) through the prepared plan,
    samples 0 / 15 / 16 / 31 against the oracle's CFG step."""
    B, sub = 32, [0, 15, 16, 31]
    p = preset("sdf_chd8bar")
    cond = chd8bar._encode_chord(torch.from_numpy(synth.chords(B, 4242)).cuda())
    uc = -torch.ones(B, 1, p.d_cond).cuda()
    x = torch.from_numpy(synth.gaussian((B, 2, 128, 128), 77)).cuda()
    d = DDIMSampler(chd8bar.ldm, 50, "uniform", 0.0)
    step, index = int(d.time_steps[49]), 49
    prep = d.prepare(cond, uncond_scale=5.0, uncond_cond=uc)
    got, _, e_t = d.p_sample(x, cond, None, step, index, uncond_scale=5.0, uncond_cond=uc, prep=prep)
    got_plain, _, e_plain = d.p_sample(x, cond, None, step, index, uncond_scale=5.0, uncond_cond=uc)
    assert torch.equal(got, got_plain) and torch.equal(e_t, e_plain)          # hoisting changes no bit
    w = unet_ref.to_torch(synth_unet_state(UNetConfig(d_cond=512), 0))
    ref = sampler_ref.DDIMSamplerRef(lambda x_, t_, c_: unet_ref.unet_forward(w, UNetConfig(d_cond=512), x_, t_, c_), 1000, *LIN, n_steps=50)
    with torch.no_grad():
        xr, _, _ = ref.p_sample(x[sub].cpu(), cond[sub].cpu(), step, index, 5.0, uc[sub].cpu())
    err = (got[sub].cpu() - xr).abs().max().item()
    print("config 3 (B=32, CFG 5, one DDIM step) max-abs-diff vs oracle on samples 0/15/16/31:", err)
    assert err < 1e-3


# ---------------------------------------------------------------------------------------------- measurement aids
def test_mfma_probe_runs_and_reports_its_flops(lib):
    """pf_mfma_probe (bench.py's `roofline.sustained`): one launch, the flop count it reports, the sink untouched, bad arguments refused."""
    sink = torch.zeros(1, device="cuda")
    fl = C.c_double(0.0)
    _lib.check(lib.pf_mfma_probe(sink.data_ptr(), 50, C.byref(fl), _lib.current_stream()), "pf_mfma_probe")
    torch.cuda.synchronize()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert fl.value == cus * 8 * 24 * 32768 * 50 and sink.item() == 0.0
    assert lib.pf_mfma_probe(0, 50, C.byref(fl), _lib.current_stream()) < 0
    assert lib.pf_mfma_probe(sink.data_ptr(), 0, C.byref(fl), _lib.current_stream()) < 0
