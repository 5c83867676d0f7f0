"""The pre-attention half of a SpatialTransformer's first layer as one launch (pf_preattn_fused, csrc/preattn_fused_bf3.hip):
GroupNorm (affine, eps 1e-6) -> proj_in 1x1 -> LayerNorm1 -> to_q | to_k | to_v  (ref:stable_diffusion/model/unet_attention.py:64-72,
:240-243, :150-166).  Checked against a plain torch fp32 statement of those lines, against the three launches it replaces (same plane
layout, values equal up to the summation order inside proj_in), through the attention kernel that consumes its planes, and inside the
whole UNet with the plan option both ways."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from polyffusion_amd import _lib, synth  # noqa: E402
from test_gpu_bf16x3 import TOL_OP, pack3  # noqa: E402
from test_gpu_ops import dev, gn_scale_shift, rnd, run_conv  # noqa: E402

C, H = 256, 4


@pytest.fixture(scope="module")
def lib():
    _lib.require_gpu()
    return _lib.load()


def _setup(B, L, seed):
    x = rnd((B, L, C), seed) * 1.3 + 0.25
    gg, gb = 1 + 0.1 * rnd((C,), seed + 1), 0.1 * rnd((C,), seed + 2)
    w_in, b_in = rnd((C, C), seed + 3, C ** -0.5), rnd((C,), seed + 4, 0.1)
    lg, lb = 1 + 0.1 * rnd((C,), seed + 5), 0.1 * rnd((C,), seed + 6)
    w_qkv = rnd((3 * C, C), seed + 7, C ** -0.5) * 1.5
    # GroupNorm over (group channels x all L tokens) of a sample: NCHW view [B, C, L, 1]
    xn = F.group_norm(x.transpose(1, 2).unsqueeze(-1), 32, gg, gb, eps=1e-6).squeeze(-1).transpose(1, 2)
    y = F.linear(xn, w_in, b_in)
    qkv = F.linear(F.layer_norm(y, (C,), lg, lb, 1e-5), w_qkv)
    return x, gg, gb, w_in, b_in, lg, lb, w_qkv, y, qkv


def _fused(lib, xd, B, L, sc, sh, stats, tiles, gg, gb, p_in, b_in, lg, lb, p_qkv):
    y = torch.empty(B, L, C, device="cuda")
    planes = torch.zeros(B * L * 3 * C, dtype=torch.float32, device="cuda")
    _lib.check(lib.pf_preattn_fused(xd.data_ptr(), B, L, sc.data_ptr(), sh.data_ptr(), _lib.ptr(stats), tiles, _lib.ptr(gg), _lib.ptr(gb), 1e-6,
                                    p_in.data_ptr(), b_in.data_ptr(), y.data_ptr(), lg.data_ptr(), lb.data_ptr(), 1e-5, p_qkv.data_ptr(),
                                    planes.data_ptr(), _lib.current_stream()), "pf_preattn_fused")
    torch.cuda.synchronize()
    return y, planes


def _decode(planes, B, L):
    """6 bf16 planes -> [3][B*L*C] floats (hi + lo) in the buffer's own element order (Q, K row-major; V^T head-major, permuted)."""
    pl = planes.view(_lib.x3_torch_dtype()).float().view(3, 2, B * L * C)
    return pl[:, 0] + pl[:, 1]


@pytest.mark.parametrize("B,L", [(2, 1024), (16, 1024), (3, 256), (1, 128), (5, 64)])
def test_preattn_fused_vs_torch_and_vs_the_three_launches(lib, B, L):
    x, gg, gb, w_in, b_in, lg, lb, w_qkv, y_ref, qkv_ref = _setup(B, L, 500 + B)
    xd = dev(x)
    ggd, gbd, lgd, lbd, b_ind = dev(gg), dev(gb), dev(lg), dev(lb), dev(b_in)
    p_in, p_qkv = pack3(lib, w_in), pack3(lib, w_qkv)
    sc, sh = gn_scale_shift(lib, xd.view(B, 1, L, C), None, ggd, gbd, 1e-6)
    y, planes = _fused(lib, xd, B, L, sc, sh, None, 0, None, None, p_in, b_ind, lgd, lbd, p_qkv)
    assert torch.isfinite(y).all()
    assert (y.cpu() - y_ref).abs().max().item() < TOL_OP
    # Q and K thirds are [token][C] row-major: compare with torch directly
    rec = _decode(planes, B, L).cpu()
    assert (rec[0].view(B, L, C) - qkv_ref[..., :C]).abs().max().item() < 2 * TOL_OP
    assert (rec[1].view(B, L, C) - qkv_ref[..., C:2 * C]).abs().max().item() < 2 * TOL_OP

    # the three launches it replaces: proj_in with the GroupNorm-affine prologue, LayerNorm planes, q|k|v planes GEMM
    y2 = torch.empty(B, L, C, device="cuda")
    run_conv(lib, x0=xd, c0=C, batch=B, hin=1, win=L, ks=1, stride=1, ups=0, w=p_in, n=C, prologue=2, sc=sc, sh=sh, bias=b_ind, out=y2, ld_out=C,
             precision=1)
    lnp = torch.zeros(B * L * C, device="cuda")
    _lib.check(lib.pf_ln_planes(y2.data_ptr(), B * L, C, 1e-5, lgd.data_ptr(), lbd.data_ptr(), lnp.data_ptr(), _lib.current_stream()))
    planes2 = torch.zeros(B * L * 3 * C, dtype=torch.float32, device="cuda")
    dummy = torch.empty(1, device="cuda")
    run_conv(lib, x0=lnp, c0=C, batch=B, hin=1, win=L, ks=1, stride=1, ups=0, w=p_qkv, n=3 * C, out=dummy, ld_out=3 * C, precision=1, a_planes=1,
             qkv_planes=planes2)
    torch.cuda.synchronize()
    assert (y - y2).abs().max().item() < 2e-5 * max(1.0, y2.abs().max().item())     # same products, another summation order inside a 16-deep step
    rec2 = _decode(planes2, B, L).cpu()
    assert (rec - rec2).abs().max().item() < 1e-4 * max(1.0, rec2.abs().max().item())   # all three thirds, V^T in its permuted layout included

    # the attention kernel reads the fused launch's planes like the chain's
    q, k, v = (t.reshape(B, L, H, 64) for t in qkv_ref.chunk(3, dim=-1))
    att = (torch.einsum("bihd,bjhd->bhij", q, k) * 0.125).softmax(-1)
    ref = torch.einsum("bhij,bjhd->bihd", att, v).reshape(B, L, C)
    if L % 128 == 0:
        out = torch.empty(B, L, C, device="cuda")
        _lib.check(lib.pf_attention_bf16x3(planes.data_ptr(), out.data_ptr(), C, None, B, H, L, -1, _lib.current_stream()))
        torch.cuda.synchronize()
        assert (out.cpu() - ref).abs().max().item() < 5e-4

    # bit-reproducible
    for _ in range(4):
        y3, planes3 = _fused(lib, xd, B, L, sc, sh, None, 0, None, None, p_in, b_ind, lgd, lbd, p_qkv)
        assert torch.equal(y3, y) and torch.equal(planes3.view(torch.int32), planes.view(torch.int32))


@pytest.mark.parametrize("B,L,T", [(16, 1024, 16), (2, 256, 4), (3, 1024, 8)])
def test_preattn_fused_with_the_groupnorm_finalize_folded_in(lib, B, L, T):
    """gn_stats given: every workgroup reduces its sample's producer tile statistics itself (fp64, gn_finalize_tiles' arithmetic) and
    the scale / shift rows it writes equal the finalize launch's; the results equal the unfolded call bit for bit."""
    x, gg, gb, w_in, b_in, lg, lb, w_qkv, y_ref, _ = _setup(B, L, 600 + B)
    xd = dev(x)
    ggd, gbd, lgd, lbd, b_ind = dev(gg), dev(gb), dev(lg), dev(lb), dev(b_in)
    p_in, p_qkv = pack3(lib, w_in), pack3(lib, w_qkv)
    xt = xd.view(B, T, L // T, C)
    stats = torch.stack([xt.sum(2), (xt * xt).sum(2)], dim=-1).contiguous()      # [B][T][C][2], what a producer's epilogue emits
    sc0, sh0 = torch.empty(B, C, device="cuda"), torch.empty(B, C, device="cuda")
    _lib.check(lib.pf_gn_finalize_tiles(stats.data_ptr(), T, C, None, 0, 0, B, L, 32, 1e-6, ggd.data_ptr(), gbd.data_ptr(), sc0.data_ptr(), sh0.data_ptr(),
                                        _lib.current_stream()))
    y0, planes0 = _fused(lib, xd, B, L, sc0, sh0, None, 0, None, None, p_in, b_ind, lgd, lbd, p_qkv)
    sc, sh = torch.zeros(B, C, device="cuda"), torch.zeros(B, C, device="cuda")
    y, planes = _fused(lib, xd, B, L, sc, sh, stats, T, ggd, gbd, p_in, b_ind, lgd, lbd, p_qkv)
    assert torch.equal(sc, sc0) and torch.equal(sh, sh0)
    assert torch.equal(y, y0) and torch.equal(planes.view(torch.int32), planes0.view(torch.int32))
    assert (y.cpu() - y_ref).abs().max().item() < TOL_OP


def test_preattn_fused_rejects_bad_arguments(lib):
    z = torch.zeros(1, 96, C, device="cuda")
    with pytest.raises(RuntimeError, match="multiple of 64"):
        _lib.check(lib.pf_preattn_fused(z.data_ptr(), 1, 96, z.data_ptr(), z.data_ptr(), None, 0, None, None, 1e-6, z.data_ptr(), z.data_ptr(), z.data_ptr(),
                                        z.data_ptr(), z.data_ptr(), 1e-5, z.data_ptr(), z.data_ptr(), _lib.current_stream()))
    with pytest.raises(RuntimeError, match="null"):
        _lib.check(lib.pf_preattn_fused(z.data_ptr(), 1, 64, z.data_ptr(), z.data_ptr(), None, 0, None, None, 1e-6, None, z.data_ptr(), z.data_ptr(),
                                        z.data_ptr(), z.data_ptr(), 1e-5, z.data_ptr(), z.data_ptr(), _lib.current_stream()))


def test_unet_with_the_fused_pre_attention_launch_on_and_off():
    """Whole sdf_chd8bar UNet at the bench shape (B = 16, bf16x3): the plan option changes the launch count by 2 per transformer block and
    the result by summation-order rounding only; `auto` keeps the three launches (the fused form measured neutral end to end)."""
    from polyffusion_amd.inference_sdf import synthetic_model
    from polyffusion_amd.params import preset
    m = synthetic_model(preset("sdf_chd8bar"))
    u = m.ldm.eps_model
    u.set_precision("bf16x3")
    B = 16
    x = torch.from_numpy(synth.gaussian((B, 2, 128, 128), 1234)).cuda()
    c = m._encode_chord(torch.from_numpy(synth.chords(B, 4242)).cuda())
    t = torch.full((B,), 999, dtype=torch.long, device="cuda")
    try:
        base = u(x, t, c).clone()
        n_auto = u.n_launches(B)
        u.set_option("pre_fused", False)
        n_off = u.n_launches(B)
        off = u(x, t, c).clone()
        u.set_option("pre_fused", True)
        n_on = u.n_launches(B)
        on = u(x, t, c).clone()
        assert n_off - n_on == 2 * 11 and n_auto == n_off      # forced on: all 11 transformer blocks; auto = off
        assert (on - off).abs().max().item() <= 1e-4 and (base - off).abs().max().item() <= 1e-4
        u.set_option("pre_fused", None)
        assert torch.equal(u(x, t, c), base)
    finally:
        u.set_option("pre_fused", None)
