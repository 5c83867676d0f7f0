"""Per-kernel parity (-m gpu): every HIP kernel, called through the C ABI, against the same
torch fp32 op the oracle / reference uses, on seeded inputs.  Tolerances are written per test."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from polyffusion_amd import _lib  # noqa: E402


@pytest.fixture(scope="module")
def lib():
    _lib.require_gpu()
    return _lib.load()


def rnd(shape, seed, scale=1.0):
    return torch.from_numpy(np.random.Generator(np.random.PCG64(seed)).standard_normal(shape).astype(np.float32) * scale)


def dev(t):
    return None if t is None else t.cuda().contiguous()


def pack_w(lib, w):
    """torch conv/linear weight [N,K(,3,3)] -> packed device tensor."""
    n, k = w.shape[0], w.shape[1]
    taps = 9 if w.dim() == 4 and w.shape[2] == 3 else 1
    dst = torch.zeros(lib.pf_packed_gemm_weight_floats(n, k, taps), dtype=torch.float32)
    _lib.check(lib.pf_pack_gemm_weight(w.contiguous().data_ptr(), n, k, taps, dst.data_ptr()))
    return dst.cuda()


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def run_conv(lib, **kw):
    a = _lib.ConvArgs()
    keep = []
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            keep.append(v)
            v = v.data_ptr()
        setattr(a, k, v)
    _lib.check(lib.pf_conv2d(C.byref(a), _lib.current_stream()), "pf_conv2d")
    torch.cuda.synchronize()


def gn_scale_shift(lib, x0, x1, gamma, beta, eps):
    B, H, W, c0 = x0.shape
    c1 = 0 if x1 is None else x1.shape[3]
    C_ = c0 + c1
    sc = torch.empty(B, C_, device="cuda")
    sh = torch.empty(B, C_, device="cuda")
    scratch = torch.empty(B * 64 * C_ * 2 * 8, dtype=torch.uint8, device="cuda")
    _lib.check(lib.pf_gn_scale_shift(x0.data_ptr(), c0, _lib.ptr(x1), c1, B, H * W, 32, eps, gamma.data_ptr(),
                                     beta.data_ptr(), sc.data_ptr(), sh.data_ptr(), scratch.data_ptr(), scratch.numel(),
                                     _lib.current_stream()))
    return sc, sh


@pytest.mark.parametrize("B,H,W,c0,c1,cout", [(2, 32, 32, 64, 0, 64), (1, 16, 16, 256, 128, 256), (2, 8, 8, 32, 32, 32),
                                               (16, 16, 16, 256, 256, 256), (1, 128, 128, 64, 0, 64), (3, 12, 20, 64, 32, 96)])
def test_resblock_conv_gn_silu(lib, B, H, W, c0, c1, cout):
    """GN(32)+SiLU fused into the 3x3 conv load; bias + per-sample bias + residual fused into the store."""
    cin = c0 + c1
    x = rnd((B, cin, H, W), 1) * 1.5 + 0.3
    w, bias = rnd((cout, cin, 3, 3), 2, (1.0 / (cin * 9)) ** 0.5), rnd((cout,), 3, 0.1)
    gamma, beta = 1 + 0.1 * rnd((cin,), 4), 0.1 * rnd((cin,), 5)
    sb, res = rnd((B, cout), 6), rnd((B, cout, H, W), 7)
    ref = F.conv2d(F.silu(F.group_norm(x, 32, gamma, beta, eps=1e-5)), w, bias, padding=1) + sb[:, :, None, None] + res
    x0 = dev(nhwc(x[:, :c0]))
    x1 = dev(nhwc(x[:, c0:])) if c1 else None
    sc, sh = gn_scale_shift(lib, x0, x1, dev(gamma), dev(beta), 1e-5)
    # statistics check: y = x*sc+sh must equal group_norm
    gn_ref = F.group_norm(x, 32, gamma, beta, eps=1e-5)
    gn_hip = x * sc.cpu()[:, :, None, None] + sh.cpu()[:, :, None, None]
    assert (gn_ref - gn_hip).abs().max() < 2e-5
    out = torch.empty(B, H, W, cout, device="cuda")
    run_conv(lib, x0=x0, c0=c0, x1=x1, c1=c1, batch=B, hin=H, win=W, ks=3, stride=1, ups=0, w=pack_w(lib, w), n=cout,
             prologue=1, sc=sc, sh=sh, bias=dev(bias), sbias=dev(sb), ld_sbias=cout, res=dev(nhwc(res)), ld_res=cout,
             out=out, ld_out=cout)
    err = (out.cpu() - nhwc(ref)).abs().max().item()
    assert err < 1e-4, err


@pytest.mark.parametrize("B,H,W,c", [(2, 32, 32, 64), (1, 64, 64, 128), (4, 16, 16, 256), (1, 10, 18, 32)])
def test_downsample_upsample(lib, B, H, W, c):
    x = rnd((B, c, H, W), 11)
    w, bias = rnd((c, c, 3, 3), 12, (1.0 / (c * 9)) ** 0.5), rnd((c,), 13, 0.1)
    xd, wp = dev(nhwc(x)), pack_w(lib, w)
    ref = F.conv2d(x, w, bias, stride=2, padding=1)
    out = torch.empty(B, ref.shape[2], ref.shape[3], c, device="cuda")
    run_conv(lib, x0=xd, c0=c, batch=B, hin=H, win=W, ks=3, stride=2, ups=0, w=wp, n=c, bias=dev(bias), out=out, ld_out=c)
    assert (out.cpu() - nhwc(ref)).abs().max() < 1e-4
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, bias, padding=1)
    out = torch.empty(B, 2 * H, 2 * W, c, device="cuda")
    run_conv(lib, x0=xd, c0=c, batch=B, hin=H, win=W, ks=3, stride=1, ups=1, w=wp, n=c, bias=dev(bias), out=out, ld_out=c)
    assert (out.cpu() - nhwc(ref)).abs().max() < 1e-4


@pytest.mark.parametrize("B,L,k0,k1,n", [(2, 1024, 256, 0, 256), (16, 256, 256, 0, 768), (3, 100, 64, 32, 32),
                                          (1, 16384, 128, 64, 64), (2, 64, 1024, 0, 256), (1, 7, 32, 0, 160)])
def test_linear_plain_and_dual_source(lib, B, L, k0, k1, n):
    """1x1 conv / Linear with bias, residual and per-sample bias; ragged M, N not a multiple of 64."""
    K = k0 + k1
    x = rnd((B, L, K), 21)
    w, bias, res, sb = rnd((n, K), 22, K ** -0.5), rnd((n,), 23, 0.1), rnd((B, L, n), 24), rnd((B, n), 25)
    ref = F.linear(x, w, bias) + res + sb[:, None, :]
    out = torch.empty(B, L, n, device="cuda")
    run_conv(lib, x0=dev(x[..., :k0]), c0=k0, x1=dev(x[..., k0:]) if k1 else None, c1=k1, batch=B, hin=1, win=L, ks=1,
             stride=1, ups=0, w=pack_w(lib, w), n=n, bias=dev(bias), res=dev(res), ld_res=n, sbias=dev(sb), ld_sbias=n,
             out=out, ld_out=n)
    assert (out.cpu() - ref).abs().max() < 1e-4


def test_linear_groupnorm_prologue(lib):
    """SpatialTransformer.norm (eps 1e-6) folded into proj_in."""
    B, H, W, c = 2, 16, 16, 256
    x = rnd((B, c, H, W), 31) * 2 + 0.5
    gamma, beta = 1 + 0.1 * rnd((c,), 32), 0.1 * rnd((c,), 33)
    w, bias = rnd((c, c, 1, 1), 34, c ** -0.5), rnd((c,), 35, 0.1)
    ref = F.conv2d(F.group_norm(x, 32, gamma, beta, eps=1e-6), w, bias)
    xd = dev(nhwc(x))
    sc, sh = gn_scale_shift(lib, xd, None, dev(gamma), dev(beta), 1e-6)
    out = torch.empty(B, H * W, c, device="cuda")
    run_conv(lib, x0=xd, c0=c, batch=B, hin=1, win=H * W, ks=1, stride=1, ups=0, w=pack_w(lib, w), n=c, prologue=2, sc=sc, sh=sh,
             bias=dev(bias), out=out, ld_out=c)
    assert (out.cpu() - nhwc(ref).reshape(B, H * W, c)).abs().max() < 1e-4


@pytest.mark.parametrize("B,L,c", [(2, 1024, 256), (1, 256, 64), (3, 50, 128)])
def test_layernorm_prologue_and_stats(lib, B, L, c):
    x = rnd((B, L, c), 41) * 1.7 - 0.4
    gamma, beta = 1 + 0.1 * rnd((c,), 42), 0.1 * rnd((c,), 43)
    w = rnd((3 * c, c), 44, c ** -0.5)
    ref = F.linear(F.layer_norm(x, (c,), gamma, beta), w)
    xd = dev(x)
    mu, rs = torch.empty(B * L, device="cuda"), torch.empty(B * L, device="cuda")
    _lib.check(lib.pf_ln_stats(xd.data_ptr(), B * L, c, 1e-5, mu.data_ptr(), rs.data_ptr(), _lib.current_stream()))
    assert (mu.cpu() - x.mean(-1).reshape(-1)).abs().max() < 1e-6
    assert (rs.cpu() - (x.var(-1, unbiased=False) + 1e-5).rsqrt().reshape(-1)).abs().max() < 1e-5
    out = torch.empty(B, L, 3 * c, device="cuda")
    run_conv(lib, x0=xd, c0=c, batch=B, hin=1, win=L, ks=1, stride=1, ups=0, w=pack_w(lib, w), n=3 * c, prologue=3,
             sc=dev(gamma), sh=dev(beta), mean=mu, rstd=rs, out=out, ld_out=3 * c)
    assert (out.cpu() - ref).abs().max() < 1e-4


def test_geglu_epilogue(lib):
    """proj -> (value, gate) = chunk(2) -> value * gelu_erf(gate), with the interleaved weight packing."""
    from polyffusion_amd.unet import UNetModel  # packing of ff.net.0.proj is owned by the plan; emulate it here
    B, L, c = 2, 256, 64
    inner = 4 * c
    x = rnd((B, L, c), 51)
    w, bias = rnd((2 * inner, c), 52, c ** -0.5), rnd((2 * inner,), 53, 0.1)
    a, g = F.linear(x, w, bias).chunk(2, dim=-1)
    ref = a * F.gelu(g)
    col = lambda n: 64 * ((n % inner) // 32) + (0 if n < inner else 32) + (n % inner) % 32
    perm = torch.tensor([col(n) for n in range(2 * inner)])
    wp_src = torch.empty_like(w)
    wp_src[perm] = w          # row n of the torch weight lands on packed column col(n)
    bp = torch.empty_like(bias)
    bp[perm] = bias
    out = torch.empty(B, L, inner, device="cuda")
    run_conv(lib, x0=dev(x), c0=c, batch=B, hin=1, win=L, ks=1, stride=1, ups=0, w=pack_w(lib, wp_src), n=2 * inner,
             bias=dev(bp), geglu=1, out=out, ld_out=inner)
    assert (out.cpu() - ref).abs().max() < 1e-4


@pytest.mark.parametrize("B,H,dh,lq,lk", [(2, 4, 64, 1024, 1024), (16, 4, 64, 256, 256), (1, 2, 32, 64, 64),
                                           (2, 2, 32, 256, 4), (1, 4, 64, 1024, 128), (1, 4, 64, 200, 77)])
def test_attention(lib, B, H, dh, lq, lk):
    d = H * dh
    q, k, v = rnd((B, lq, d), 61), rnd((B, lk, d), 62), rnd((B, lk, d), 63)
    qh, kh, vh = (t.view(B, -1, H, dh) for t in (q, k, v))
    att = (torch.einsum("bihd,bjhd->bhij", qh, kh) * dh ** -0.5).softmax(-1)
    ref = torch.einsum("bhij,bjhd->bihd", att, vh).reshape(B, lq, d)
    out = torch.empty(B, lq, d, device="cuda")
    qd, kd, vd = dev(q), dev(k), dev(v)
    _lib.check(lib.pf_attention(qd.data_ptr(), d, kd.data_ptr(), d, vd.data_ptr(), d, out.data_ptr(), d, B, H, dh, lq, lk,
                                _lib.current_stream()))
    torch.cuda.synchronize()
    assert (out.cpu() - ref).abs().max() < 2e-5


def test_attention_peaked_scores(lib):
    """Online-softmax rescale path: one key dominates late in the sequence."""
    B, H, dh, L = 1, 4, 64, 512
    d = H * dh
    q, k, v = rnd((B, L, d), 71), rnd((B, L, d), 72), rnd((B, L, d), 73)
    k[0, 300] = q[0, 5] * 4.0   # spike
    qh, kh, vh = (t.view(B, -1, H, dh) for t in (q, k, v))
    att = (torch.einsum("bihd,bjhd->bhij", qh, kh) * dh ** -0.5).softmax(-1)
    ref = torch.einsum("bhij,bjhd->bihd", att, vh).reshape(B, L, d)
    out = torch.empty(B, L, d, device="cuda")
    qd, kd, vd = dev(q), dev(k), dev(v)
    _lib.check(lib.pf_attention(qd.data_ptr(), d, kd.data_ptr(), d, vd.data_ptr(), d, out.data_ptr(), d, B, H, dh, L, L,
                                _lib.current_stream()))
    assert (out.cpu() - ref).abs().max() < 2e-5


def test_sampler_step_kernels_vs_reference_goldens(lib, golden):
    """pf_ddpm_step / pf_ddim_step / pf_axpby / pf_cfg_combine against known answers from the real reference."""
    from polyffusion_amd.unet import LatentDiffusion
    from polyffusion_amd.sampler import DDIMSampler, SDFSampler
    g = golden("steps.npz")
    x, e, nz = (torch.from_numpy(g[k]).cuda() for k in ("x", "e_t", "noise"))

    class FakeLDM(LatentDiffusion):
        def __init__(self):
            LatentDiffusion.__init__(self, None)

        @property
        def device(self):
            return torch.device("cuda")

    ldm = FakeLDM()
    s = SDFSampler(ldm, noise_fn=lambda shape: nz)
    s.get_eps = lambda *a, **k: e
    for step in (0, 1, 500, 999):
        xp, x0, _ = s.p_sample(x, None, None, step)
        tol = 2e-6 * max(1.0, float(np.abs(g[f"sdf_xprev_{step}"]).max()))
        assert np.abs(xp.cpu().numpy() - g[f"sdf_xprev_{step}"]).max() <= tol
        assert np.abs(x0.cpu().numpy() - g[f"sdf_x0_{step}"]).max() <= tol
        assert np.abs(s.q_sample(x, step, nz).cpu().numpy() - g[f"sdf_q_{step}"]).max() <= 1e-6
    for tag, (S, disc, eta) in dict(u50=(50, "uniform", 0.0), u20e1=(20, "uniform", 1.0)).items():
        d = DDIMSampler(ldm, S, disc, eta, noise_fn=lambda shape: nz)
        for idx in (0, 1, S - 1):
            xp, p0 = d.get_x_prev_and_pred_x0(e, idx, x)
            tol = 2e-6 * max(1.0, float(np.abs(g[f"ddim_{tag}_predx0_{idx}"]).max()))
            assert np.abs(xp.cpu().numpy() - g[f"ddim_{tag}_xprev_{idx}"]).max() <= tol
            assert np.abs(p0.cpu().numpy() - g[f"ddim_{tag}_predx0_{idx}"]).max() <= tol
            assert np.abs(d.q_sample(x, idx, nz).cpu().numpy() - g[f"ddim_{tag}_q_{idx}"]).max() <= 1e-6
    # classifier-free guidance combine through get_eps with the same toy model as the golden
    s2 = SDFSampler(ldm)
    s2.model = lambda x_, t_, c_: x_ * c_.mean(dim=(1, 2))[:, None, None, None] + t_[:, None, None, None].float() * 1e-3
    cc, uc, t7 = torch.from_numpy(g["cfg_c"]).cuda(), -torch.ones(2, 1, 32).cuda(), torch.tensor([7, 7]).cuda()
    for sc in (0.0, 1.0, 5.0):
        got = s2.get_eps(x, t7, cc, uncond_scale=sc, uncond_cond=uc)
        assert np.abs(got.cpu().numpy() - g[f"cfg_eps_{sc}"]).max() <= 2e-6


def test_randn_is_standard_normal_and_shard_invariant(lib):
    n = 1 << 20
    full = torch.empty(n, device="cuda")
    _lib.check(lib.pf_randn(full.data_ptr(), n, 1234, 7, 0, _lib.current_stream()))
    a = full.cpu().double()
    assert abs(a.mean()) < 5e-3 and abs(a.std() - 1) < 5e-3
    assert abs((a ** 3).mean()) < 2e-2 and abs((a ** 4).mean() - 3) < 5e-2
    # a shard [off, off+m) of the global tensor equals the slice of the unsharded draw (any offset/length)
    for off, m in ((0, 1000), (3, 4097), (32768 * 5, 32768), (n - 5, 5)):
        part = torch.empty(m, device="cuda")
        _lib.check(lib.pf_randn(part.data_ptr(), m, 1234, 7, off, _lib.current_stream()))
        assert torch.equal(part, full[off:off + m])
    other = torch.empty(1000, device="cuda")
    _lib.check(lib.pf_randn(other.data_ptr(), 1000, 1234, 8, 0, _lib.current_stream()))
    assert not torch.equal(other, full[:1000])


def test_errors_are_reported_not_ub(lib):
    a = _lib.ConvArgs()
    a.ks = 5
    assert lib.pf_conv2d(C.byref(a), None) < 0 and b"ks" in lib.pf_last_error()
    x = torch.zeros(4, device="cuda")
    assert lib.pf_attention(x.data_ptr(), 4, x.data_ptr(), 4, x.data_ptr(), 4, x.data_ptr(), 4, 1, 1, 48, 1, 1, None) < 0


@pytest.mark.parametrize("precision", [0, 1])
@pytest.mark.parametrize("B,H,W,c0,c1,cout", [(2, 32, 32, 64, 0, 64), (2, 12, 20, 64, 32, 96), (16, 16, 16, 256, 0, 256)])
def test_producer_tile_statistics_feed_groupnorm(lib, precision, B, H, W, c0, c1, cout):
    """A conv launch emits per-tile channel (sum, sumsq) of what it stores; pf_gn_finalize_tiles turns the statistics of
    two producers (channel-concatenated) into the same scale/shift a pass over the data gives."""
    cin = c0 + c1
    x = rnd((B, cin, H, W), 1)
    w, bias = rnd((cout, cin, 3, 3), 2, (1.0 / (cin * 9)) ** 0.5), rnd((cout,), 3, 0.1)
    gamma, beta = 1 + 0.1 * rnd((cin,), 4), 0.1 * rnd((cin,), 5)
    x0 = dev(nhwc(x[:, :c0]))
    x1 = dev(nhwc(x[:, c0:])) if c1 else None
    sc, sh = gn_scale_shift(lib, x0, x1, dev(gamma), dev(beta), 1e-5)
    wp = pack_w(lib, w)
    if precision == 1:
        dst = torch.zeros(lib.pf_packed_gemm_weight_floats(cout, cin, 9), dtype=torch.float32)
        _lib.check(lib.pf_pack_gemm_weight_bf16x3(w.contiguous().data_ptr(), cout, cin, 9, dst.data_ptr()))
        wp = dst.cuda()
    kw = dict(x0=x0, c0=c0, x1=x1, c1=c1, batch=B, hin=H, win=W, ks=3, stride=1, ups=0, w=wp, n=cout, prologue=1, sc=sc, sh=sh,
              bias=dev(bias), precision=precision)
    a = _lib.ConvArgs()
    for k, v in kw.items():
        setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else (0 if v is None else v))
    nt = lib.pf_conv_stats_tiles(C.byref(a))
    assert nt > 0
    out = torch.empty(B, H, W, cout, device="cuda")
    stats = torch.full((B, nt, cout, 2), float("nan"), device="cuda")
    run_conv(lib, out=out, ld_out=cout, stats_out=stats, **kw)
    o = out.cpu().double()
    assert torch.isfinite(stats).all()
    tot = stats.cpu().double().sum(1)  # [B, cout, 2]
    assert (tot[..., 0] - o.sum((1, 2))).abs().max() < 1e-2
    assert (tot[..., 1] - (o * o).sum((1, 2))).abs().max() < 1e-2
    # GroupNorm of concat(out, out2) from statistics only
    g2, b2 = 1 + 0.1 * rnd((2 * cout,), 8), 0.1 * rnd((2 * cout,), 9)
    sc2, sh2 = torch.empty(B, 2 * cout, device="cuda"), torch.empty(B, 2 * cout, device="cuda")
    g2d, b2d = dev(g2), dev(b2)  # keep alive across the launch
    _lib.check(lib.pf_gn_finalize_tiles(stats.data_ptr(), nt, cout, stats.data_ptr(), nt, cout, B, H * W, 32, 1e-5,
                                        g2d.data_ptr(), b2d.data_ptr(), sc2.data_ptr(), sh2.data_ptr(), _lib.current_stream()))
    torch.cuda.synchronize()
    cat = torch.cat([out.cpu(), out.cpu()], dim=-1).permute(0, 3, 1, 2)
    ref = F.group_norm(cat, 32, g2, b2, eps=1e-5)
    got = cat * sc2.cpu()[:, :, None, None] + sh2.cpu()[:, :, None, None]
    assert (ref - got).abs().max() < 5e-5
