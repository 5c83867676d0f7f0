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
    pl = op.view(_lib.x3_torch_dtype()).float().view(2, B, L, C)
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
