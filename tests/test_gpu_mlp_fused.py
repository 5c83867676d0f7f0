"""The transformer block's feed-forward half as one launch (pf_mlp_geglu_fused, csrc/mlp_fused_bf3.hip):
`x = ff(norm3(x)) + x` of ref:stable_diffusion/model/unet_attention.py:119-124 with FeedForward / GeGLU of :296-333.
Checked against a plain torch fp32 statement of the same lines and, bit for bit, against the three-launch chain it replaces."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from polyffusion_amd import _lib  # noqa: E402
from test_gpu_bf16x3 import TOL_OP, pack3  # noqa: E402
from test_gpu_ops import dev, rnd, run_conv  # noqa: E402

C, HID = 256, 1024


@pytest.fixture(scope="module")
def lib():
    _lib.require_gpu()
    return _lib.load()


def _weights(seed):
    w1, b1 = rnd((2 * HID, C), seed, C ** -0.5), rnd((2 * HID,), seed + 1, 0.1)
    w2, b2 = rnd((C, HID), seed + 2, HID ** -0.5), rnd((C,), seed + 3, 0.1)
    gamma, beta = 1 + 0.1 * rnd((C,), seed + 4), 0.1 * rnd((C,), seed + 5)
    # value/gate rows interleaved in 32-row blocks: what the plan's D_GEGLU_W packing does (unet.hip geglu_col)
    w1i = torch.stack([w1[:HID].view(HID // 32, 32, C), w1[HID:].view(HID // 32, 32, C)], 1).reshape(2 * HID, C)
    b1i = torch.stack([b1[:HID].view(HID // 32, 32), b1[HID:].view(HID // 32, 32)], 1).reshape(2 * HID)
    return w1, b1, w2, b2, gamma, beta, w1i, b1i


@pytest.mark.parametrize("B,L", [(2, 1024), (16, 1024), (3, 64), (1, 256), (5, 192)])
def test_fused_mlp_vs_torch_and_vs_the_unfused_chain(lib, B, L):
    w1, b1, w2, b2, gamma, beta, w1i, b1i = _weights(300)
    x = rnd((B, L, C), 310 + B) * 1.4 + 0.2
    xn = F.layer_norm(x, (C,), gamma, beta, 1e-5)
    h = F.linear(xn, w1, b1)
    ref = x + F.linear(h[..., :HID] * F.gelu(h[..., HID:]), w2, b2)

    xd, gd, bd = dev(x), dev(gamma), dev(beta)
    p1, p2, b1d, b2d = pack3(lib, w1i), pack3(lib, w2), dev(b1i), dev(b2)
    out = torch.empty(B, L, C, device="cuda")
    st = _lib.current_stream()
    _lib.check(lib.pf_mlp_geglu_fused(xd.data_ptr(), B, L, gd.data_ptr(), bd.data_ptr(), 1e-5, p1.data_ptr(), b1d.data_ptr(),
                                      p2.data_ptr(), b2d.data_ptr(), out.data_ptr(), None, st), "pf_mlp_geglu_fused")
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    assert (out.cpu() - ref).abs().max().item() < TOL_OP

    # the chain it replaces: LayerNorm planes -> GeGLU projection (planes out) -> K = 1024 planes GEMM + residual
    lnp = torch.zeros(B * L * C, device="cuda")
    _lib.check(lib.pf_ln_planes(xd.data_ptr(), B * L, C, 1e-5, gd.data_ptr(), bd.data_ptr(), lnp.data_ptr(), st))
    gp = torch.zeros(B * L * HID, device="cuda")
    run_conv(lib, x0=lnp, c0=C, batch=B, hin=1, win=L, ks=1, stride=1, ups=0, w=p1, n=2 * HID, bias=b1d, geglu=1, out=gp, ld_out=HID,
             precision=1, a_planes=1, out_planes=gp)
    out2 = torch.empty(B, L, C, device="cuda")
    run_conv(lib, x0=gp, c0=HID, batch=B, hin=1, win=L, ks=1, stride=1, ups=0, w=p2, n=C, bias=b2d, res=xd, ld_res=C, out=out2, ld_out=C,
             precision=1, a_planes=1)
    assert torch.equal(out, out2), f"fused and unfused differ by {(out - out2).abs().max().item():.3e}"

    # plane-pair output (what proj_out consumes after the last transformer layer)
    op = torch.zeros(B * L * C, device="cuda")
    _lib.check(lib.pf_mlp_geglu_fused(xd.data_ptr(), B, L, gd.data_ptr(), bd.data_ptr(), 1e-5, p1.data_ptr(), b1d.data_ptr(),
                                      p2.data_ptr(), b2d.data_ptr(), None, op.data_ptr(), st))
    torch.cuda.synchronize()
    pl = op.view(torch.bfloat16).float().view(2, B, L, C)
    assert (pl[0] + pl[1] - out).abs().max().item() < 2e-5 * max(1.0, out.abs().max().item())   # a plane pair carries 16 mantissa bits
    # bit-reproducible
    for _ in range(4):
        o3 = torch.empty_like(out)
        _lib.check(lib.pf_mlp_geglu_fused(xd.data_ptr(), B, L, gd.data_ptr(), bd.data_ptr(), 1e-5, p1.data_ptr(), b1d.data_ptr(),
                                          p2.data_ptr(), b2d.data_ptr(), o3.data_ptr(), None, st))
        assert torch.equal(o3, out)


def test_fused_mlp_rejects_bad_shapes(lib):
    x = torch.zeros(1, 96, C, device="cuda")
    with pytest.raises(RuntimeError, match="multiple of 64"):
        _lib.check(lib.pf_mlp_geglu_fused(x.data_ptr(), 1, 96, x.data_ptr(), x.data_ptr(), 1e-5, x.data_ptr(), x.data_ptr(), x.data_ptr(),
                                          x.data_ptr(), x.data_ptr(), None, _lib.current_stream()))


@pytest.mark.parametrize("B,L", [(2, 1024), (16, 1024), (3, 128)])
def test_fused_mlp_with_chained_proj_out(lib, B, L):
    """pf_mlp_geglu_proj_fused: `proj_out(x + ff(norm3(x))) + x_in` (unet_attention.py:77-79 after :119-124) as one launch, against torch and -
    bit for bit - against the fused MLP with plane output followed by the planes GEMM it replaces; the tile statistics against torch sums."""
    w1, b1, w2, b2, gamma, beta, w1i, b1i = _weights(400)
    w3, b3 = rnd((C, C), 407, C ** -0.5), rnd((C,), 408, 0.1)
    x = rnd((B, L, C), 410 + B) * 1.2 - 0.1
    xin = rnd((B, L, C), 420 + B)
    xn = F.layer_norm(x, (C,), gamma, beta, 1e-5)
    h = F.linear(xn, w1, b1)
    x2 = x + F.linear(h[..., :HID] * F.gelu(h[..., HID:]), w2, b2)
    ref = F.linear(x2, w3, b3) + xin

    xd, xind, gd, bd = dev(x), dev(xin), dev(gamma), dev(beta)
    p1, p2, p3, b1d, b2d, b3d = pack3(lib, w1i), pack3(lib, w2), pack3(lib, w3), dev(b1i), dev(b2), dev(b3)
    out = torch.empty(B, L, C, device="cuda")
    stats = torch.zeros(B, L // 64, C, 2, device="cuda")
    st = _lib.current_stream()
    _lib.check(lib.pf_mlp_geglu_proj_fused(xd.data_ptr(), B, L, gd.data_ptr(), bd.data_ptr(), 1e-5, p1.data_ptr(), b1d.data_ptr(), p2.data_ptr(),
                                           b2d.data_ptr(), p3.data_ptr(), b3d.data_ptr(), xind.data_ptr(), out.data_ptr(), stats.data_ptr(), st))
    torch.cuda.synchronize()
    assert (out.cpu() - ref).abs().max().item() < TOL_OP
    # per-tile channel statistics of what was stored
    oc = out.view(B, L // 64, 64, C)
    assert (stats[..., 0] - oc.sum(2)).abs().max().item() < 2e-3 and (stats[..., 1] - (oc * oc).sum(2)).abs().max().item() < 2e-2
    # the two launches it replaces
    op = torch.zeros(B * L * C, device="cuda")
    _lib.check(lib.pf_mlp_geglu_fused(xd.data_ptr(), B, L, gd.data_ptr(), bd.data_ptr(), 1e-5, p1.data_ptr(), b1d.data_ptr(), p2.data_ptr(),
                                      b2d.data_ptr(), None, op.data_ptr(), st))
    out2 = torch.empty(B, L, C, device="cuda")
    run_conv(lib, x0=op, c0=C, batch=B, hin=1, win=L, ks=1, stride=1, ups=0, w=p3, n=C, bias=b3d, res=xind, ld_res=C, out=out2, ld_out=C,
             precision=1, a_planes=1)
    assert torch.equal(out, out2), f"chained and unchained differ by {(out - out2).abs().max().item():.3e}"
    for _ in range(3):
        o3 = torch.empty_like(out)
        _lib.check(lib.pf_mlp_geglu_proj_fused(xd.data_ptr(), B, L, gd.data_ptr(), bd.data_ptr(), 1e-5, p1.data_ptr(), b1d.data_ptr(), p2.data_ptr(),
                                               b2d.data_ptr(), p3.data_ptr(), b3d.data_ptr(), xind.data_ptr(), o3.data_ptr(), None, st))
        assert torch.equal(o3, out)


@pytest.mark.parametrize("B,L,tail", [(2, 1024, True), (16, 1024, True), (3, 128, False), (16, 1024, False)])
def test_transformer_tail_fused(lib, B, L, tail):
    """pf_transformer_tail_fused: attn1.to_out + residual, norm3, GeGLU feed-forward + residual (and proj_out + block input) in one launch
    (unet_attention.py:115-124, 77-79), against torch and - bit for bit - against the planes GEMM + fused feed-forward pair it replaces."""
    import ctypes as CT
    from test_gpu_bf16x3 import _split_planes
    w1, b1, w2, b2, gamma, beta, w1i, b1i = _weights(500)
    wo, bo = rnd((C, C), 507, C ** -0.5), rnd((C,), 508, 0.1)
    w3, b3 = rnd((C, C), 509, C ** -0.5), rnd((C,), 510, 0.1)
    att = rnd((B, L, C), 511 + B)
    att = (att.to(torch.bfloat16).float() + (att - att.to(torch.bfloat16).float()).to(torch.bfloat16).float())   # exactly a plane pair
    x0, xin, cross = rnd((B, L, C), 520 + B), rnd((B, L, C), 530 + B), rnd((B, 3 * C), 540 + B, 0.3)
    x1 = F.linear(att, wo, bo) + cross[:, None, C:2 * C] + x0
    h = F.linear(F.layer_norm(x1, (C,), gamma, beta, 1e-5), w1, b1)
    x2 = x1 + F.linear(h[..., :HID] * F.gelu(h[..., HID:]), w2, b2)
    ref = F.linear(x2, w3, b3) + xin if tail else x2

    ap, x0d, xind, crossd, gd, bd = _split_planes(att), dev(x0), dev(xin), dev(cross), dev(gamma), dev(beta)
    po, p1, p2, p3 = pack3(lib, wo), pack3(lib, w1i), pack3(lib, w2), pack3(lib, w3)
    bod, b1d, b2d, b3d = dev(bo), dev(b1i), dev(b2), dev(b3)
    x1buf = torch.zeros(B, L, C, device="cuda")
    out = torch.empty(B, L, C, device="cuda")
    stats = torch.zeros(B, L // 64, C, 2, device="cuda")
    a = _lib.TBlockTailArgs()
    a.attn_planes, a.wo, a.bo, a.cross_bias, a.ld_cross_bias, a.x0, a.x1 = ap.data_ptr(), po.data_ptr(), bod.data_ptr(), crossd.data_ptr() + C * 4, 3 * C, \
        x0d.data_ptr(), x1buf.data_ptr()
    a.batch, a.l, a.ln_gamma, a.ln_beta, a.ln_eps = B, L, gd.data_ptr(), bd.data_ptr(), 1e-5
    a.w1, a.b1, a.w2, a.b2 = p1.data_ptr(), b1d.data_ptr(), p2.data_ptr(), b2d.data_ptr()
    if tail:
        a.w3, a.b3, a.res3, a.stats3 = p3.data_ptr(), b3d.data_ptr(), xind.data_ptr(), stats.data_ptr()
    a.out = out.data_ptr()
    st = _lib.current_stream()
    _lib.check(lib.pf_transformer_tail_fused(CT.byref(a), st), "pf_transformer_tail_fused")
    torch.cuda.synchronize()
    assert (x1buf.cpu() - x1).abs().max().item() < TOL_OP and (out.cpu() - ref).abs().max().item() < TOL_OP
    # the pair it replaces
    t1 = torch.empty(B, L, C, device="cuda")
    run_conv(lib, x0=ap, c0=C, batch=B, hin=1, win=L, ks=1, stride=1, ups=0, w=po, n=C, bias=bod, sbias=crossd[:, C:], ld_sbias=3 * C, res=x0d,
             ld_res=C, out=t1, ld_out=C, precision=1, a_planes=1)
    assert torch.equal(t1, x1buf)
    out2 = torch.empty(B, L, C, device="cuda")
    if tail:
        _lib.check(lib.pf_mlp_geglu_proj_fused(t1.data_ptr(), B, L, gd.data_ptr(), bd.data_ptr(), 1e-5, p1.data_ptr(), b1d.data_ptr(), p2.data_ptr(),
                                               b2d.data_ptr(), p3.data_ptr(), b3d.data_ptr(), xind.data_ptr(), out2.data_ptr(), None, st))
    else:
        _lib.check(lib.pf_mlp_geglu_fused(t1.data_ptr(), B, L, gd.data_ptr(), bd.data_ptr(), 1e-5, p1.data_ptr(), b1d.data_ptr(), p2.data_ptr(),
                                          b2d.data_ptr(), out2.data_ptr(), None, st))
    assert torch.equal(out, out2), f"chained and unchained differ by {(out - out2).abs().max().item():.3e}"
    for _ in range(3):
        o3 = torch.empty_like(out)
        a.out = o3.data_ptr()
        _lib.check(lib.pf_transformer_tail_fused(CT.byref(a), st))
        assert torch.equal(o3, out)
