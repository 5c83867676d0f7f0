"""The fused Winograd F(2x2, 3x3) form of the ResBlock convolution (csrc/conv_wino.hip; -m gpu): pf_conv2d with pf_conv_args.wino against
torch's F.conv2d(F.silu(F.group_norm(x))) + bias + per-sample bias + residual (ref: stable_diffusion/model/unet.py:262-318), against the
direct form of the same launch, its GroupNorm tile statistics against sums over the stored output, and bit-reproducibility."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from polyffusion_amd import _lib  # noqa: E402
from test_gpu_ops import dev, gn_scale_shift, nhwc, rnd, run_conv  # noqa: E402
from test_gpu_bf16x3 import pack3  # noqa: E402

TOL_OP = 1e-4   # on O(1) outputs; the direct split form sits at ~3e-5, the transforms amplify operand rounding ~1.5x


@pytest.fixture(scope="module")
def lib():
    _lib.require_gpu()
    return _lib.load()


def pack_wino(lib, w):
    n, k = w.shape[0], w.shape[1]
    dst = torch.zeros(lib.pf_wino_weight_bytes(n, k), dtype=torch.uint8)
    _lib.check(lib.pf_pack_wino_weight_bf16x3(w.contiguous().data_ptr(), n, k, dst.data_ptr()))
    return dst.cuda()


SHAPES = [(2, 32, 32, 64, 0, 64), (1, 128, 128, 64, 0, 64), (2, 64, 64, 128, 64, 128), (1, 16, 16, 32, 32, 64), (3, 16, 48, 64, 0, 128),
          (1, 64, 64, 256, 128, 128)]


@pytest.mark.parametrize("B,H,W,c0,c1,cout", SHAPES)
def test_wino_conv_gn_silu_vs_torch(lib, B, H, W, c0, c1, cout):
    cin = c0 + c1
    x = rnd((B, cin, H, W), 1) * 1.5 + 0.3
    w, bias = rnd((cout, cin, 3, 3), 2, (1.0 / (cin * 9)) ** 0.5), rnd((cout,), 3, 0.1)
    gamma, beta = 1 + 0.1 * rnd((cin,), 4), 0.1 * rnd((cin,), 5)
    sb, res = rnd((B, cout), 6), rnd((B, cout, H, W), 7)
    ref = F.conv2d(F.silu(F.group_norm(x, 32, gamma, beta, eps=1e-5)), w, bias, padding=1) + sb[:, :, None, None] + res
    x0 = dev(nhwc(x[:, :c0]))
    x1 = dev(nhwc(x[:, c0:])) if c1 else None
    sc, sh = gn_scale_shift(lib, x0, x1, dev(gamma), dev(beta), 1e-5)
    kw = dict(x0=x0, c0=c0, x1=x1, c1=c1, batch=B, hin=H, win=W, ks=3, stride=1, ups=0, w=pack3(lib, w), n=cout, prologue=1, sc=sc, sh=sh,
              bias=dev(bias), sbias=dev(sb), ld_sbias=cout, res=dev(nhwc(res)), ld_res=cout, ld_out=cout, precision=1)
    a = _lib.ConvArgs()
    for k, v in kw.items():
        setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else (0 if v is None else v))
    ww = pack_wino(lib, w)
    a.w_wino, a.wino = ww.data_ptr(), 1
    nt = lib.pf_conv_stats_tiles(C.byref(a))
    assert nt == (H // 16) * (W // 16)
    stats = torch.full((B, nt, cout, 2), float("nan"), device="cuda")
    out = torch.full((B, H, W, cout), float("nan"), device="cuda")
    run_conv(lib, out=out, stats_out=stats, w_wino=ww, wino=1, **kw)
    err = (out.cpu() - nhwc(ref)).abs().max().item()
    direct = torch.empty(B, H, W, cout, device="cuda")
    run_conv(lib, out=direct, **kw)
    err_d = (direct.cpu() - nhwc(ref)).abs().max().item()
    print(f"wino {B}x{H}x{W} {c0}+{c1}->{cout}: max-abs-diff vs torch {err:.2e} (direct form {err_d:.2e})")
    assert err < TOL_OP, err
    # tile statistics = per-channel (sum, sum of squares) of the stored outputs over each 16x16-pixel tile
    o = out.view(B, H // 16, 16, W // 16, 16, cout).permute(0, 1, 3, 2, 4, 5).reshape(B, nt, 256, cout).double()
    want = torch.stack([o.sum(2), (o * o).sum(2)], dim=-1)
    rel = ((stats.double() - want).abs() / (want.abs() + 1.0)).max().item()
    assert rel < 1e-5, rel
    # bit-reproducible
    out2 = torch.empty_like(out)
    run_conv(lib, out=out2, w_wino=ww, wino=1, **kw)
    assert torch.equal(out.view(torch.int32), out2.view(torch.int32))


def test_wino_falls_back_when_not_eligible(lib):
    """hin not a multiple of 16: the launch runs the direct form on `w` (and says so through the statistics tile count)."""
    B, H, W, c, cout = 1, 24, 40, 64, 64
    x = rnd((B, c, H, W), 11)
    w = rnd((cout, c, 3, 3), 12, (1.0 / (c * 9)) ** 0.5)
    gamma, beta = 1 + 0.1 * rnd((c,), 13), 0.1 * rnd((c,), 14)
    ref = F.conv2d(F.silu(F.group_norm(x, 32, gamma, beta, eps=1e-5)), w, None, padding=1)
    x0 = dev(nhwc(x))
    sc, sh = gn_scale_shift(lib, x0, None, dev(gamma), dev(beta), 1e-5)
    out = torch.empty(B, H, W, cout, device="cuda")
    ww = pack_wino(lib, w)
    run_conv(lib, x0=x0, c0=c, batch=B, hin=H, win=W, ks=3, stride=1, ups=0, w=pack3(lib, w), n=cout, prologue=1, sc=sc, sh=sh, out=out,
             ld_out=cout, precision=1, w_wino=ww, wino=1)
    assert (out.cpu() - nhwc(ref)).abs().max() < 3e-4


@pytest.mark.parametrize("B,H,W,c0,c1,cout,tiles", [(2, 32, 32, 256, 128, 128, 4), (1, 16, 16, 64, 32, 64, 3)])
def test_wino_groupnorm_finalize_inside_the_consumer(lib, B, H, W, c0, c1, cout, tiles):
    """pf_conv_args.gn_* with the Winograd form: the consumer reduces its producers' per-tile statistics itself (the rows of scale / shift
    it writes equal pf_gn_finalize_tiles'), result against torch."""
    cin = c0 + c1
    x = rnd((B, cin, H, W), 51) * 1.3 + 0.2
    w, bias = rnd((cout, cin, 3, 3), 52, (1.0 / (cin * 9)) ** 0.5), rnd((cout,), 53, 0.1)
    gamma, beta = 1 + 0.1 * rnd((cin,), 54), 0.1 * rnd((cin,), 55)
    ref = F.conv2d(F.silu(F.group_norm(x, 32, gamma, beta, eps=1e-5)), w, bias, padding=1)
    xh = nhwc(x).reshape(B, H * W, cin)

    def tile_stats(part, nt):
        chunks = part.double().tensor_split(nt, dim=1)
        return torch.stack([torch.stack([c.sum(1), (c * c).sum(1)], dim=-1) for c in chunks], dim=1).float().cuda().contiguous()

    s0 = tile_stats(xh[..., :c0], tiles)
    s1 = tile_stats(xh[..., c0:], tiles + 1) if c1 else None
    g, bt = dev(gamma), dev(beta)
    sc_ref, sh_ref = torch.empty(B, cin, device="cuda"), torch.empty(B, cin, device="cuda")
    _lib.check(lib.pf_gn_finalize_tiles(s0.data_ptr(), tiles, c0, _lib.ptr(s1), tiles + 1 if c1 else 0, c1, B, H * W, 32, 1e-5,
                                        g.data_ptr(), bt.data_ptr(), sc_ref.data_ptr(), sh_ref.data_ptr(), _lib.current_stream()))
    sc, sh = torch.full((B, cin), float("nan"), device="cuda"), torch.full((B, cin), float("nan"), device="cuda")
    out = torch.empty(B, H, W, cout, device="cuda")
    ww = pack_wino(lib, w)
    run_conv(lib, x0=dev(nhwc(x[:, :c0])), c0=c0, x1=dev(nhwc(x[:, c0:])) if c1 else None, c1=c1, batch=B, hin=H, win=W, ks=3, stride=1, ups=0,
             w=pack3(lib, w), n=cout, prologue=1, bias=dev(bias), ld_out=cout, precision=1, sc=sc, sh=sh, out=out, gn_stats0=s0, gn_tiles0=tiles,
             gn_stats1=s1 if c1 else 0, gn_tiles1=tiles + 1 if c1 else 0, gn_gamma=g, gn_beta=bt, gn_eps=1e-5, gn_groups=32, w_wino=ww, wino=1)
    assert (sc - sc_ref).abs().max() < 1e-6 and (sh - sh_ref).abs().max() < 1e-6
    assert (out.cpu() - nhwc(ref)).abs().max() < TOL_OP


def _unet(x3=None):
    import numpy as np  # noqa: F401
    from polyffusion_amd.arch import UNetConfig
    from polyffusion_amd.unet import UNetModel
    from polyffusion_amd.weights import synth_unet_state
    cfg = UNetConfig(d_cond=512)
    kw = dict(x3=x3) if x3 else {}
    m = UNetModel(in_channels=cfg.in_channels, out_channels=cfg.out_channels, channels=cfg.channels, n_res_blocks=cfg.n_res_blocks,
                  attention_levels=cfg.attention_levels, channel_multipliers=cfg.channel_multipliers, n_heads=cfg.n_heads,
                  tf_layers=cfg.tf_layers, d_cond=cfg.d_cond, img_h=128, img_w=128, **kw)
    m.load_state_dict(synth_unet_state(cfg, 0))
    return m


@pytest.mark.parametrize("x3", [None, "f16"])
def test_full_unet_with_winograd_convs_vs_reference_golden(golden, x3):
    """Every qualifying ResBlock conv in the Winograd form (PF_OPT_CONV_WINO = on: 14 of the 22 at the 128 / 64 / 32 levels): the full UNet against
    the REAL reference's output (tests/golden/unet_chd8bar_b2.npz), in both split builds, and bit-reproducible."""
    import numpy as np
    from polyffusion_amd import synth
    g = golden("unet_chd8bar_b2.npz")
    m = _unet(x3)
    m.set_precision("f16x3" if x3 else "bf16x3")
    x = torch.from_numpy(synth.gaussian((2, 2, 128, 128), int(g["x_seed"]))).cuda()
    c = torch.from_numpy(synth.gaussian((2, 1, 512), int(g["cond_seed"]))).cuda()
    t = torch.from_numpy(g["t"]).cuda()
    base = m(x, t, c).cpu().numpy()
    n0 = m.n_launches(2)
    m.set_option("conv_wino", True)
    o = m(x, t, c)
    o2 = m(x, t, c)
    assert torch.equal(o.view(torch.int32), o2.view(torch.int32))
    o = o.cpu().numpy()
    err, err0 = np.abs(o - g["out"]).max(), np.abs(base - g["out"]).max()
    print(f"full UNet [{'f16x3' if x3 else 'bf16x3'}] with Winograd convs: max-abs-diff vs the reference {err:.3e} (direct form {err0:.3e}); launches {m.n_launches(2)} (direct {n0})")
    assert np.abs(o - base).max() > 0          # the option changed the arithmetic: the form really ran
    assert err < 1e-4, err


def test_winograd_auto_picks_at_batch_16():
    """AUTO (the default) at the bench's batch: the picker turns the form on for the deep-K convs; same result as the direct plan to the split's rounding."""
    from polyffusion_amd import synth
    m = _unet()
    m.set_precision("bf16x3")
    B = 16
    x = torch.from_numpy(synth.gaussian((B, 2, 128, 128), 3)).cuda()
    c = torch.from_numpy(synth.gaussian((B, 1, 512), 4)).cuda()
    t = torch.full((B,), 500, dtype=torch.long, device="cuda")
    auto = m(x, t, c)
    m.set_option("conv_wino", False)
    direct = m(x, t, c)
    d = (auto - direct).abs().max().item()
    print(f"B = 16: AUTO vs direct plan max-abs-diff {d:.3e}")
    assert 0 < d < 1e-4


@pytest.mark.parametrize("B,shared", [(64, True), (64, False)])
def test_winograd_large_batch_bit_reproducible(B, shared):
    """Batch 64 (config 3's guidance evaluation: 2 x 32 rows, with and without the shared prefix): 2^20 pixels at the 128x128 level - the
    per-piece words hold pixel indices WITHIN the sample (a 20-bit absolute index overflowed here) - every qualifying conv in the Winograd
    form, repeated launches bit-identical, and equal to the direct plan to the split's rounding."""
    from polyffusion_amd import synth
    m = _unet()
    m.set_precision("bf16x3")
    Bx = B // 2 if shared else B
    x = torch.from_numpy(synth.gaussian((Bx, 2, 128, 128), 3)).cuda()
    c = torch.from_numpy(synth.gaussian((B, 1, 512), 4)).cuda()
    t = torch.full((B,), 500, dtype=torch.long, device="cuda")
    m.set_option("conv_wino", False)
    direct = m(x, t, c, shared_x=shared).clone()
    m.set_option("conv_wino", True)
    ref = m(x, t, c, shared_x=shared).clone()
    for _ in range(4):
        assert torch.equal(m(x, t, c, shared_x=shared).view(torch.int32), ref.view(torch.int32))
    assert 0 < (ref - direct).abs().max().item() < 1e-4
