"""bf16x3 split-MFMA mode (-m gpu): same kernels' contract as the fp32 MFMA path, looser per-op
tolerance (products carry ~1e-5 relative error), and the whole-UNet contract of BASELINE.json
(max-abs-diff < 1e-3 vs the reference) checked against the reference goldens."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from polyffusion_amd import _lib, synth  # noqa: E402
from polyffusion_amd.arch import UNetConfig  # noqa: E402
from polyffusion_amd.unet import UNetModel  # noqa: E402
from polyffusion_amd.weights import synth_unet_state  # noqa: E402
from test_gpu_ops import dev, gn_scale_shift, nhwc, rnd, run_conv  # noqa: E402

TOL_OP = 3e-4  # on O(1) outputs; expected ~3e-5


@pytest.fixture(scope="module")
def lib():
    _lib.require_gpu()
    return _lib.load()


def pack3(lib, w):
    n, k = w.shape[0], w.shape[1]
    taps = 9 if w.dim() == 4 and w.shape[2] == 3 else 1
    dst = torch.zeros(lib.pf_packed_gemm_weight_floats(n, k, taps), dtype=torch.float32)
    _lib.check(lib.pf_pack_gemm_weight_bf16x3(w.contiguous().data_ptr(), n, k, taps, dst.data_ptr()))
    return dst.cuda()


@pytest.mark.parametrize("B,H,W,c0,c1,cout", [(2, 32, 32, 64, 0, 64), (1, 16, 16, 256, 128, 256), (2, 8, 8, 32, 32, 32),
                                               (16, 16, 16, 256, 256, 256), (1, 128, 128, 64, 0, 64), (3, 12, 20, 64, 32, 96)])
def test_resblock_conv_gn_silu_bf16x3(lib, B, H, W, c0, c1, cout):
    cin = c0 + c1
    x = rnd((B, cin, H, W), 1) * 1.5 + 0.3
    w, bias = rnd((cout, cin, 3, 3), 2, (1.0 / (cin * 9)) ** 0.5), rnd((cout,), 3, 0.1)
    gamma, beta = 1 + 0.1 * rnd((cin,), 4), 0.1 * rnd((cin,), 5)
    sb, res = rnd((B, cout), 6), rnd((B, cout, H, W), 7)
    ref = F.conv2d(F.silu(F.group_norm(x, 32, gamma, beta, eps=1e-5)), w, bias, padding=1) + sb[:, :, None, None] + res
    x0 = dev(nhwc(x[:, :c0]))
    x1 = dev(nhwc(x[:, c0:])) if c1 else None
    sc, sh = gn_scale_shift(lib, x0, x1, dev(gamma), dev(beta), 1e-5)
    out = torch.empty(B, H, W, cout, device="cuda")
    run_conv(lib, x0=x0, c0=c0, x1=x1, c1=c1, batch=B, hin=H, win=W, ks=3, stride=1, ups=0, w=pack3(lib, w), n=cout,
             prologue=1, sc=sc, sh=sh, bias=dev(bias), sbias=dev(sb), ld_sbias=cout, res=dev(nhwc(res)), ld_res=cout,
             out=out, ld_out=cout, precision=1)
    err = (out.cpu() - nhwc(ref)).abs().max().item()
    assert err < TOL_OP, err


@pytest.mark.parametrize("B,H,W,c", [(2, 32, 32, 64), (1, 64, 64, 128), (4, 16, 16, 256), (1, 10, 18, 32)])
def test_downsample_upsample_bf16x3(lib, B, H, W, c):
    x = rnd((B, c, H, W), 11)
    w, bias = rnd((c, c, 3, 3), 12, (1.0 / (c * 9)) ** 0.5), rnd((c,), 13, 0.1)
    xd, wp = dev(nhwc(x)), pack3(lib, w)
    ref = F.conv2d(x, w, bias, stride=2, padding=1)
    out = torch.empty(B, ref.shape[2], ref.shape[3], c, device="cuda")
    run_conv(lib, x0=xd, c0=c, batch=B, hin=H, win=W, ks=3, stride=2, ups=0, w=wp, n=c, bias=dev(bias), out=out, ld_out=c, precision=1)
    assert (out.cpu() - nhwc(ref)).abs().max() < TOL_OP
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, bias, padding=1)
    out = torch.empty(B, 2 * H, 2 * W, c, device="cuda")
    run_conv(lib, x0=xd, c0=c, batch=B, hin=H, win=W, ks=3, stride=1, ups=1, w=wp, n=c, bias=dev(bias), out=out, ld_out=c, precision=1)
    assert (out.cpu() - nhwc(ref)).abs().max() < TOL_OP


@pytest.mark.parametrize("B,L,k0,k1,n", [(2, 1024, 256, 0, 256), (16, 256, 256, 0, 768), (3, 100, 64, 32, 32),
                                          (1, 16384, 128, 64, 64), (2, 64, 1024, 0, 256), (1, 7, 32, 0, 160)])
def test_linear_bf16x3(lib, B, L, k0, k1, n):
    K = k0 + k1
    x = rnd((B, L, K), 21)
    w, bias, res, sb = rnd((n, K), 22, K ** -0.5), rnd((n,), 23, 0.1), rnd((B, L, n), 24), rnd((B, n), 25)
    ref = F.linear(x, w, bias) + res + sb[:, None, :]
    out = torch.empty(B, L, n, device="cuda")
    run_conv(lib, x0=dev(x[..., :k0]), c0=k0, x1=dev(x[..., k0:]) if k1 else None, c1=k1, batch=B, hin=1, win=L, ks=1,
             stride=1, ups=0, w=pack3(lib, w), n=n, bias=dev(bias), res=dev(res), ld_res=n, sbias=dev(sb), ld_sbias=n,
             out=out, ld_out=n, precision=1)
    assert (out.cpu() - ref).abs().max() < TOL_OP


def test_layernorm_prologue_bf16x3(lib):
    B, L, c = 2, 1024, 256
    x = rnd((B, L, c), 41) * 1.7 - 0.4
    gamma, beta = 1 + 0.1 * rnd((c,), 42), 0.1 * rnd((c,), 43)
    w = rnd((3 * c, c), 44, c ** -0.5)
    ref = F.linear(F.layer_norm(x, (c,), gamma, beta), w)
    xd = dev(x)
    mu, rs = torch.empty(B * L, device="cuda"), torch.empty(B * L, device="cuda")
    _lib.check(lib.pf_ln_stats(xd.data_ptr(), B * L, c, 1e-5, mu.data_ptr(), rs.data_ptr(), _lib.current_stream()))
    out = torch.empty(B, L, 3 * c, device="cuda")
    run_conv(lib, x0=xd, c0=c, batch=B, hin=1, win=L, ks=1, stride=1, ups=0, w=pack3(lib, w), n=3 * c, prologue=3,
             sc=dev(gamma), sh=dev(beta), mean=mu, rstd=rs, out=out, ld_out=3 * c, precision=1)
    assert (out.cpu() - ref).abs().max() < TOL_OP


def _make(cfg, h, w, x3=None):
    """x3 = None: the default split build (bf16 pieces); "f16": the fp16-piece build (libpfhip_f16.so), mode name f16x3."""
    m = UNetModel(in_channels=cfg.in_channels, out_channels=cfg.out_channels, channels=cfg.channels,
                  n_res_blocks=cfg.n_res_blocks, attention_levels=cfg.attention_levels,
                  channel_multipliers=cfg.channel_multipliers, n_heads=cfg.n_heads, tf_layers=cfg.tf_layers,
                  d_cond=cfg.d_cond, img_h=h, img_w=w, **(dict(x3=x3) if x3 else {}))
    m.load_state_dict(synth_unet_state(cfg, 0))
    return m.set_precision(m.split_mode)


X3 = [None, "f16"]       # both split builds are covered by the driver's default run, not by an env-var re-run


@pytest.mark.parametrize("x3", X3)
def test_full_unet_bf16x3_meets_the_contract(golden, x3):
    """BASELINE.json: UNet output max-abs-diff < 1e-3 vs the reference on identical (x_t, t, cond)."""
    g = golden("unet_chd8bar_b2.npz")
    m = _make(UNetConfig(d_cond=512), 128, 128, x3)
    assert m.precision == m.split_mode
    x = torch.from_numpy(synth.gaussian((2, 2, 128, 128), int(g["x_seed"]))).cuda()
    c = torch.from_numpy(synth.gaussian((2, 1, 512), int(g["cond_seed"]))).cuda()
    o = m(x, torch.from_numpy(g["t"]).cuda(), c).cpu().numpy()
    err = np.abs(o - g["out"]).max()
    rms = float(np.sqrt(((o - g["out"]) ** 2).mean()))
    print(f"{m.split_mode} chd8bar B=2: max-abs-diff {err:.3e}, rms {rms:.3e}")
    assert err < 5e-4, err   # half the contract bar
    m.set_precision("f32")
    o32 = m(x, torch.from_numpy(g["t"]).cuda(), c).cpu().numpy()
    assert np.abs(o32 - g["out"]).max() < 1e-4


@pytest.mark.parametrize("x3", X3)
def test_small_unet_bf16x3_vs_reference_golden(golden, x3):
    SMALL = UNetConfig(in_channels=2, out_channels=2, channels=32, n_res_blocks=1, attention_levels=(1,),
                       channel_multipliers=(1, 2), n_heads=2, tf_layers=1, d_cond=32)
    g = golden("unet_small.npz")
    m = _make(SMALL, 32, 32, x3)
    x, t = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["t"]).cuda()
    assert np.abs(m(x, t, torch.from_numpy(g["cond1"]).cuda()).cpu().numpy() - g["out1"]).max() < 5e-4
    assert np.abs(m(x, t, torch.from_numpy(g["cond4"]).cuda()).cpu().numpy() - g["out4"]).max() < 5e-4


@pytest.mark.parametrize("B,H,W,c0,c1,cout,fk", [(2, 16, 16, 256, 256, 256, 0), (1, 8, 8, 64, 0, 64, 2), (4, 16, 16, 256, 0, 256, 4), (1, 32, 32, 256, 128, 256, 0)])
def test_split_k_small_m_layers(lib, B, H, W, c0, c1, cout, fk):
    """Small-M 3x3 layers split K over workgroups; the reduce kernel applies bias/sbias/residual and emits the statistics.  fk == 0: the
    library's own choice (deep K on few workgroups - the only case where the split earns its reduce pass, tools/sweep_conv.py); fk > 0: the
    split forced (pf_conv_args.force_ksplit) on shapes the picker now leaves whole."""
    cin = c0 + c1
    x = rnd((B, cin, H, W), 1) * 1.5 + 0.3
    w, bias = rnd((cout, cin, 3, 3), 2, (1.0 / (cin * 9)) ** 0.5), rnd((cout,), 3, 0.1)
    gamma, beta = 1 + 0.1 * rnd((cin,), 4), 0.1 * rnd((cin,), 5)
    sb, res = rnd((B, cout), 6), rnd((B, cout, H, W), 7)
    ref = F.conv2d(F.silu(F.group_norm(x, 32, gamma, beta, eps=1e-5)), w, bias, padding=1) + sb[:, :, None, None] + res
    x0 = dev(nhwc(x[:, :c0]))
    x1 = dev(nhwc(x[:, c0:])) if c1 else None
    sc, sh = gn_scale_shift(lib, x0, x1, dev(gamma), dev(beta), 1e-5)
    kw = dict(x0=x0, c0=c0, x1=x1, c1=c1, batch=B, hin=H, win=W, ks=3, stride=1, ups=0, w=pack3(lib, w), n=cout,
              prologue=1, sc=sc, sh=sh, bias=dev(bias), sbias=dev(sb), ld_sbias=cout, res=dev(nhwc(res)), ld_res=cout, precision=1, force_ksplit=fk)
    a = _lib.ConvArgs()
    for k, v in kw.items():
        setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else (0 if v is None else v))
    need = lib.pf_conv_splitk_ws_bytes(C.byref(a))
    assert need > 0, "shape was expected to split"
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    a.splitk_ws, a.splitk_ws_bytes = ws.data_ptr(), need
    nt = lib.pf_conv_stats_tiles(C.byref(a))
    assert nt == H * W // 64
    out = torch.empty(B, H, W, cout, device="cuda")
    stats = torch.full((B, nt, cout, 2), float("nan"), device="cuda")
    run_conv(lib, out=out, ld_out=cout, stats_out=stats, splitk_ws=ws, splitk_ws_bytes=need, **kw)
    assert (out.cpu() - nhwc(ref)).abs().max() < TOL_OP
    o = out.cpu().double()
    tot = stats.cpu().double().sum(1)
    assert (tot[..., 0] - o.sum((1, 2))).abs().max() < 1e-2 and (tot[..., 1] - (o * o).sum((1, 2))).abs().max() < 1e-2


@pytest.mark.parametrize("B,L,H,wide", [(2, 1024, 4, "0"), (3, 256, 4, "0"), (1, 128, 2, "0"), (2, 1024, 4, "1"), (3, 256, 4, "1"),
                                        (16, 1024, 4, None)])
def test_qkv_planes_and_bf16x3_attention(lib, B, L, H, wide, monkeypatch):
    """q|k|v projection (LayerNorm prologue) written as pre-split bf16 hi/lo planes, consumed by the bf16x3 attention - in its
    128-query form, in the 256-query form (two query fragments per wave), and at the bench shape with the launcher's own choice."""
    form = -1 if wide is None else int(wide)   # pf_attention_bf16x3's `form`: auto / 128-query / 256-query workgroups
    c = H * 64
    x = rnd((B, L, c), 81) * 1.3 + 0.2
    gamma, beta = 1 + 0.1 * rnd((c,), 82), 0.1 * rnd((c,), 83)
    w = rnd((3 * c, c), 84, c ** -0.5) * 1.5
    qkv = F.linear(F.layer_norm(x, (c,), gamma, beta), w)
    q, k, v = (t.reshape(B, L, H, 64) for t in qkv.chunk(3, dim=-1))
    att = (torch.einsum("bihd,bjhd->bhij", q, k) * 0.125).softmax(-1)
    ref = torch.einsum("bhij,bjhd->bihd", att, v).reshape(B, L, c)
    xd = dev(x)
    mu, rs = torch.empty(B * L, device="cuda"), torch.empty(B * L, device="cuda")
    _lib.check(lib.pf_ln_stats(xd.data_ptr(), B * L, c, 1e-5, mu.data_ptr(), rs.data_ptr(), _lib.current_stream()))
    planes = torch.zeros(B * L * 3 * c, dtype=torch.float32, device="cuda")   # 6 bf16 planes = the bytes of fp32 [M][3C]
    dummy = torch.empty(1, device="cuda")
    run_conv(lib, x0=xd, c0=c, batch=B, hin=1, win=L, ks=1, stride=1, ups=0, w=pack3(lib, w), n=3 * c, prologue=3,
             sc=dev(gamma), sh=dev(beta), mean=mu, rstd=rs, out=dummy, ld_out=3 * c, precision=1, qkv_planes=planes)
    # the planes reconstruct the projection: hi + lo == qkv to ~2^-17
    pl = planes.view(_lib.x3_torch_dtype()).float().cpu().view(6, B, L * c)
    qrec = (pl[0] + pl[1]).view(B, L, c)
    assert (qrec - qkv[..., :c]).abs().max() < 3e-4
    out = torch.empty(B, L, c, device="cuda")
    _lib.check(lib.pf_attention_bf16x3(planes.data_ptr(), out.data_ptr(), c, None, B, H, L, form, _lib.current_stream()))
    torch.cuda.synchronize()
    err = (out.cpu() - ref).abs().max().item()
    assert err < 3e-4, err


def _split_planes(x):
    """fp32 [M][K] -> the bf16 hi|lo plane pair a producer epilogue writes (as a float32-typed byte buffer)."""
    hi = x.to(_lib.x3_torch_dtype())
    lo = (x - hi.float()).to(_lib.x3_torch_dtype())
    return torch.cat([hi.reshape(-1), lo.reshape(-1)]).cuda().view(torch.float32)


@pytest.mark.parametrize("B,L,k,n", [(2, 1024, 256, 256), (16, 256, 1024, 256), (3, 200, 64, 96), (1, 128, 32, 64)])
def test_planes_gemm_chain(lib, B, L, k, n):
    """Linear whose A operand arrives as pre-split planes (global->LDS direct), writing fp32 or planes again;
    and a GeGLU projection that emits its product as planes (the ff1 -> ff2 -> proj_out chain of the transformer)."""
    x = rnd((B, L, k), 91)
    w, bias, res = rnd((n, k), 92, k ** -0.5), rnd((n,), 93, 0.1), rnd((B, L, n), 94)
    ref = F.linear(x, w, bias) + res
    ap = _split_planes(x)
    out = torch.empty(B, L, n, device="cuda")
    run_conv(lib, x0=ap, c0=k, batch=B, hin=1, win=L, ks=1, stride=1, ups=0, w=pack3(lib, w), n=n, bias=dev(bias),
             res=dev(res), ld_res=n, out=out, ld_out=n, precision=1, a_planes=1)
    assert (out.cpu() - ref).abs().max().item() < TOL_OP
    # same GEMM, result as planes
    op = torch.zeros(B * L * n, device="cuda")
    run_conv(lib, x0=ap, c0=k, batch=B, hin=1, win=L, ks=1, stride=1, ups=0, w=pack3(lib, w), n=n, bias=dev(bias),
             res=dev(res), ld_res=n, out=op, ld_out=n, precision=1, a_planes=1, out_planes=op)
    pl = op.view(_lib.x3_torch_dtype()).float().cpu().view(2, B, L, n)
    assert (pl[0] + pl[1] - ref).abs().max().item() < TOL_OP
    # GeGLU projection from fp32 input -> planes, then consumed by a planes GEMM
    if n % 64 == 0:
        wg, bg = rnd((2 * n, k), 95, k ** -0.5), rnd((2 * n,), 96, 0.1)
        h = F.linear(x, wg, bg)
        gref = h[..., :n] * F.gelu(h[..., n:])
        # interleaved GeGLU packing is the plan's job (unet.hip D_GEGLU_W): value/gate rows alternate in 32-column blocks
        wi = torch.stack([wg[:n].view(n // 32, 32, k), wg[n:].view(n // 32, 32, k)], 1).reshape(2 * n, k)
        bi = torch.stack([bg[:n].view(n // 32, 32), bg[n:].view(n // 32, 32)], 1).reshape(2 * n)
        gp = torch.zeros(B * L * n, device="cuda")
        run_conv(lib, x0=dev(x), c0=k, batch=B, hin=1, win=L, ks=1, stride=1, ups=0, w=pack3(lib, wi), n=2 * n,
                 bias=dev(bi), geglu=1, out=gp, ld_out=n, precision=1, out_planes=gp)
        pl = gp.view(_lib.x3_torch_dtype()).float().cpu().view(2, B, L, n)
        assert (pl[0] + pl[1] - gref).abs().max().item() < TOL_OP
        w2 = rnd((k, n), 97, n ** -0.5)
        out2 = torch.empty(B, L, k, device="cuda")
        run_conv(lib, x0=gp, c0=n, batch=B, hin=1, win=L, ks=1, stride=1, ups=0, w=pack3(lib, w2), n=k, out=out2, ld_out=k,
                 precision=1, a_planes=1)
        assert (out2.cpu() - F.linear(gref, w2)).abs().max().item() < TOL_OP


@pytest.mark.parametrize("L,wide", [(256, "0"), (256, "1"), (512, "1")])
def test_attention_planes_output(lib, L, wide, monkeypatch):
    """hi/lo plane output (the operand of the to_out planes GEMM) from both forms of the kernel."""
    form = int(wide)
    B, H = 2, 4
    c = H * 64
    qkv = rnd((B, L, 3 * c), 101)
    q, k, v = (t.reshape(B, L, H, 64) for t in qkv.chunk(3, dim=-1))
    att = (torch.einsum("bihd,bjhd->bhij", q, k) * 0.125).softmax(-1)
    ref = torch.einsum("bhij,bjhd->bihd", att, v).reshape(B, L, c)
    planes = torch.zeros(B * L * 3 * c, dtype=torch.float32, device="cuda")
    eye = torch.eye(3 * c)
    dummy = torch.empty(1, device="cuda")
    run_conv(lib, x0=dev(qkv), c0=3 * c, batch=B, hin=1, win=L, ks=1, stride=1, ups=0, w=pack3(lib, eye), n=3 * c,
             out=dummy, ld_out=3 * c, precision=1, qkv_planes=planes)
    op = torch.zeros(B * L * c, device="cuda")
    _lib.check(lib.pf_attention_bf16x3(planes.data_ptr(), None, c, op.data_ptr(), B, H, L, form, _lib.current_stream()))
    torch.cuda.synchronize()
    pl = op.view(_lib.x3_torch_dtype()).float().cpu().view(2, B, L, c)
    assert (pl[0] + pl[1] - ref).abs().max().item() < 3e-4


@pytest.mark.parametrize("wide", ["0", "1"])
def test_attention_running_maximum_extremes(lib, wide, monkeypatch):
    """Scores whose row maxima keep growing from tile to tile (key norms ramp up along the sequence: the reference exponent of the
    256-query form moves several times per row, the 128-query form rescales on every tile), rows dominated by one late key, and rows
    whose scores are all far below the first tile's - against the fp32 softmax."""
    form = int(wide)
    B, L, H = 1, 1024, 4
    c = H * 64
    g = torch.Generator().manual_seed(5)
    q = torch.randn(B, L, H, 64, generator=g) * 2.0
    k = torch.randn(B, L, H, 64, generator=g) * torch.linspace(0.05, 3.0, L).view(1, L, 1, 1)
    v = torch.randn(B, L, H, 64, generator=g)
    k[:, 900, 1] = q[:, 17, 1] * 1.5            # query 17 of head 1 is dominated by key 900
    k[:, :64, 2] *= 40.0                         # head 2: the first tile holds the extremes of both signs
    qkv = torch.cat([q.reshape(B, L, c), k.reshape(B, L, c), v.reshape(B, L, c)], dim=-1).contiguous()
    att = (torch.einsum("bihd,bjhd->bhij", q.double(), k.double()) * 0.125).softmax(-1)
    ref = torch.einsum("bhij,bjhd->bihd", att, v.double()).reshape(B, L, c).float()
    planes = torch.zeros(B * L * 3 * c, dtype=torch.float32, device="cuda")
    dummy = torch.empty(1, device="cuda")
    run_conv(lib, x0=dev(qkv), c0=3 * c, batch=B, hin=1, win=L, ks=1, stride=1, ups=0, w=pack3(lib, torch.eye(3 * c)), n=3 * c,
             out=dummy, ld_out=3 * c, precision=1, qkv_planes=planes)
    out = torch.empty(B, L, c, device="cuda")
    _lib.check(lib.pf_attention_bf16x3(planes.data_ptr(), out.data_ptr(), c, None, B, H, L, form, _lib.current_stream()))
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    err = (out.cpu() - ref).abs().max().item()
    assert err < 1e-3, err                       # operands are bf16 hi+lo (2^-17 relative) and the scores reach +-400 here


@pytest.mark.parametrize("B,L,H", [(2, 1024, 4), (3, 256, 4)])
def test_ln_planes_feeds_the_projection(lib, B, L, H):
    """LayerNorm applied once into hi/lo planes (pf_ln_planes), consumed by the q|k|v planes GEMM and the bf16x3 attention:
    the all-planes form of the transformer block's first half."""
    c = H * 64
    x = rnd((B, L, c), 111) * 1.3 + 0.2
    gamma, beta = 1 + 0.1 * rnd((c,), 112), 0.1 * rnd((c,), 113)
    w = rnd((3 * c, c), 114, c ** -0.5) * 1.5
    xn = F.layer_norm(x, (c,), gamma, beta)
    qkv = F.linear(xn, w)
    q, k, v = (t.reshape(B, L, H, 64) for t in qkv.chunk(3, dim=-1))
    att = (torch.einsum("bihd,bjhd->bhij", q, k) * 0.125).softmax(-1)
    ref = torch.einsum("bhij,bjhd->bihd", att, v).reshape(B, L, c)
    xd = dev(x)
    lnp = torch.zeros(B * L * c, device="cuda")
    gd, bd = dev(gamma), dev(beta)
    _lib.check(lib.pf_ln_planes(xd.data_ptr(), B * L, c, 1e-5, gd.data_ptr(), bd.data_ptr(), lnp.data_ptr(),
                                _lib.current_stream()))
    pl = lnp.view(_lib.x3_torch_dtype()).float().cpu().view(2, B, L, c)
    assert (pl[0] + pl[1] - xn).abs().max().item() < 1e-4
    planes = torch.zeros(B * L * 3 * c, dtype=torch.float32, device="cuda")
    dummy = torch.empty(1, device="cuda")
    run_conv(lib, x0=lnp, c0=c, batch=B, hin=1, win=L, ks=1, stride=1, ups=0, w=pack3(lib, w), n=3 * c, out=dummy, ld_out=3 * c,
             precision=1, a_planes=1, qkv_planes=planes)
    out = torch.empty(B, L, c, device="cuda")
    _lib.check(lib.pf_attention_bf16x3(planes.data_ptr(), out.data_ptr(), c, None, B, H, L, -1, _lib.current_stream()))
    torch.cuda.synchronize()
    assert (out.cpu() - ref).abs().max().item() < 3e-4


@pytest.mark.parametrize("B,H,W,ch,c0,c1,cout", [(2, 32, 32, 128, 64, 64, 128), (1, 16, 16, 256, 256, 256, 256), (16, 16, 16, 256, 256, 256, 256),
                                                  (3, 12, 20, 64, 96, 0, 64), (2, 64, 64, 64, 128, 64, 64)])
def test_fused_skip_projection(lib, B, H, W, ch, c0, c1, cout):
    """ResBlock tail in one launch: conv3x3(SiLU(GN(h))) + b2 + per-sample bias + skip_w . concat(x0, x1) + skip_b."""
    cx = c0 + c1
    hsrc = rnd((B, ch, H, W), 121) * 1.2 + 0.1
    x = rnd((B, cx, H, W), 122)
    w, bias = rnd((cout, ch, 3, 3), 123, (1.0 / (ch * 9)) ** 0.5), rnd((cout,), 124, 0.1)
    ws, bs = rnd((cout, cx), 125, cx ** -0.5), rnd((cout,), 126, 0.1)
    gamma, beta = 1 + 0.1 * rnd((ch,), 127), 0.1 * rnd((ch,), 128)
    ref = F.conv2d(F.silu(F.group_norm(hsrc, 32, gamma, beta, eps=1e-5)), w, bias, padding=1) + F.conv2d(x, ws[:, :, None, None], bs)
    hd = dev(nhwc(hsrc))
    x0 = dev(nhwc(x[:, :c0]))
    x1 = dev(nhwc(x[:, c0:])) if c1 else None
    sc, sh = gn_scale_shift(lib, hd, None, dev(gamma), dev(beta), 1e-5)
    out = torch.empty(B, H, W, cout, device="cuda")
    ws_bytes = 0
    kw = dict(x0=hd, c0=ch, batch=B, hin=H, win=W, ks=3, stride=1, ups=0, w=pack3(lib, w), n=cout, prologue=1, sc=sc, sh=sh,
              bias=dev(bias), out=out, ld_out=cout, precision=1, skip_x0=x0, skip_c0=c0, skip_x1=x1, skip_c1=c1,
              skip_w=pack3(lib, ws), skip_bias=dev(bs))
    a = _lib.ConvArgs()
    for k, v in kw.items():
        setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else (0 if v is None else v))
    ws_bytes = lib.pf_conv_splitk_ws_bytes(C.byref(a))
    scratch = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device="cuda")
    run_conv(lib, splitk_ws=scratch, splitk_ws_bytes=ws_bytes, **{k: v for k, v in kw.items() if v is not None})
    err = (out.cpu() - nhwc(ref)).abs().max().item()
    assert err < TOL_OP, err


def pack_upfold(lib, w):
    n, k = w.shape[0], w.shape[1]
    dst = torch.zeros(lib.pf_packed_gemm_weight_floats(n, k, 16), dtype=torch.float32)
    _lib.check(lib.pf_pack_upfold_weight_bf16x3(w.contiguous().data_ptr(), n, k, dst.data_ptr()))
    return dst.cuda()


@pytest.mark.parametrize("B,H,W,c,cout", [(2, 32, 32, 64, 64), (1, 64, 64, 128, 128), (16, 16, 16, 256, 256), (1, 10, 18, 32, 96), (3, 8, 16, 64, 32)])
def test_upsample_conv_parity_folded(lib, B, H, W, c, cout):
    """UpSample (nearest x2 + conv3x3, ref:unet.py:218-238) as four 2x2 convs on the source grid with row/column-summed
    weights: same result as the 9-tap form, 4/9 of the MACs; the per-tile GroupNorm statistics cover every output pixel."""
    x = rnd((B, c, H, W), 131)
    w, bias = rnd((cout, c, 3, 3), 132, (1.0 / (c * 9)) ** 0.5), rnd((cout,), 133, 0.1)
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, bias, padding=1)
    out = torch.empty(B, 2 * H, 2 * W, cout, device="cuda")
    a = _lib.ConvArgs()
    kw = dict(x0=dev(nhwc(x)), c0=c, batch=B, hin=H, win=W, ks=3, stride=1, ups=1, ups_fold=1, w=pack_upfold(lib, w), n=cout,
              bias=dev(bias), out=out, ld_out=cout, precision=1)
    for k, v in kw.items():
        setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    nt = lib.pf_conv_stats_tiles(C.byref(a))
    stats = torch.zeros(B, nt, cout, 2, device="cuda")
    run_conv(lib, stats_out=stats, **kw)
    o = out.cpu()
    assert (o - nhwc(ref)).abs().max().item() < TOL_OP
    tot = stats.cpu().sum(1)
    assert (tot[..., 0] - o.sum((1, 2))).abs().max() < 2e-2 and (tot[..., 1] - (o * o).sum((1, 2))).abs().max() < 2e-2


def test_full_size_batch16_properties_bf16x3():
    """BASELINE.json configs[1] size (B = 16, 128x128, sdf_chd8bar) in the product mode, through size-independent
    properties: run-to-run bit reproducibility (no atomics anywhere on the path), independence of a sample from the rest of
    the batch (GroupNorm / attention are per sample), and the n_cond == 1 collapse - the output depends on the condition
    only through per-sample biases, so equal conditions give equal outputs for equal inputs."""
    cfg = UNetConfig(d_cond=512)
    m = UNetModel(in_channels=2, out_channels=2, channels=64, n_res_blocks=2, attention_levels=(2, 3),
                  channel_multipliers=(1, 2, 4, 4), n_heads=4, tf_layers=1, d_cond=512)
    m.load_state_dict(synth_unet_state(cfg, 0))
    m.set_precision("bf16x3")
    g = torch.Generator().manual_seed(5)
    x = torch.randn(16, 2, 128, 128, generator=g).cuda()
    t = torch.randint(0, 1000, (16,), generator=g).cuda()
    c = torch.randn(16, 1, 512, generator=g).cuda()
    x[9], t[9], c[9] = x[2], t[2], c[2]                      # two identical samples at different batch positions
    a = m(x, t, c).clone()
    b = m(x, t, c).clone()
    assert torch.isfinite(a).all() and torch.equal(a, b)    # deterministic
    assert torch.equal(a[2], a[9])                           # position in the batch does not matter
    perm = torch.randperm(16, generator=g).cuda()
    p = m(x[perm].contiguous(), t[perm].contiguous(), c[perm].contiguous())
    assert torch.equal(p, a[perm])                           # neither does the batch order


@pytest.mark.parametrize("B", [1, 5])
def test_full_unet_small_and_odd_batches_vs_oracle(B):
    """Full-size sdf_chd8bar at batch sizes where other code paths run than at B = 2 / 16: split-K across workgroups
    (few tiles), the intra-workgroup K split, ragged last mat-vec row group - against the CPU oracle (contract 1e-3)."""
    from oracle import unet_ref
    cfg = UNetConfig(d_cond=512)
    m = UNetModel(in_channels=2, out_channels=2, channels=64, n_res_blocks=2, attention_levels=(2, 3),
                  channel_multipliers=(1, 2, 4, 4), n_heads=4, tf_layers=1, d_cond=512)
    st = synth_unet_state(cfg, 0)
    m.load_state_dict(st)
    m.set_precision("bf16x3")
    g = torch.Generator().manual_seed(3 + B)
    x = torch.randn(B, 2, 128, 128, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    c = torch.randn(B, 1, 512, generator=g)
    ref = unet_ref.unet_forward(unet_ref.to_torch(st), cfg, x, t, c)
    err = (m(x.cuda(), t.cuda(), c.cuda()).cpu() - ref).abs().max().item()
    assert err < 5e-4, err


@pytest.mark.parametrize("x3", X3)
def test_full_unet_txt_bf16x3_vs_reference_golden(golden, x3):
    """BASELINE.json configs[3] model (sdf_txt, d_cond 1024) in the product arithmetic mode, against the reference vector."""
    g = golden("unet_txt_b1.npz")
    m = _make(UNetConfig(d_cond=1024), 128, 128, x3)
    x = torch.from_numpy(synth.gaussian((1, 2, 128, 128), int(g["x_seed"]))).cuda()
    c = torch.from_numpy(synth.gaussian((1, 1, 1024), int(g["cond_seed"]))).cuda()
    o = m(x, torch.from_numpy(g["t"]).cuda(), c).cpu().numpy()
    assert np.abs(o - g["out"]).max() < 5e-4


@pytest.mark.parametrize("B,H,W,c0,c1,cout,tiles", [(2, 16, 16, 256, 0, 256, 4), (3, 32, 32, 256, 128, 128, 8), (1, 16, 16, 64, 32, 64, 3)])
def test_groupnorm_finalize_inside_the_consumer(lib, B, H, W, c0, c1, cout, tiles):
    """pf_conv_args.gn_*: the consuming conv reduces the producers' per-tile (sum, sumsq) itself (no pf_gn_finalize_tiles launch).
    The tile statistics are built here from the data (any split of the pixels into `tiles` parts is a valid producer layout); the
    result must equal the two-launch path, and the scale/shift rows the kernel wrote must equal the separate finalize's."""
    cin = c0 + c1
    x = rnd((B, cin, H, W), 51) * 1.3 + 0.2
    w, bias = rnd((cout, cin, 3, 3), 52, (1.0 / (cin * 9)) ** 0.5), rnd((cout,), 53, 0.1)
    gamma, beta = 1 + 0.1 * rnd((cin,), 54), 0.1 * rnd((cin,), 55)
    ref = F.conv2d(F.silu(F.group_norm(x, 32, gamma, beta, eps=1e-5)), w, bias, padding=1)
    xh = nhwc(x).reshape(B, H * W, cin)

    def tile_stats(part, nt):     # [B, nt, C, 2] fp32: per-part channel sums, as a producing conv's epilogue would emit them
        chunks = part.double().tensor_split(nt, dim=1)
        return torch.stack([torch.stack([c.sum(1), (c * c).sum(1)], dim=-1) for c in chunks], dim=1).float().cuda().contiguous()

    s0 = tile_stats(xh[..., :c0], tiles)
    s1 = tile_stats(xh[..., c0:], tiles + 1) if c1 else None
    x0 = dev(nhwc(x[:, :c0]))
    x1 = dev(nhwc(x[:, c0:])) if c1 else None
    g, bt = dev(gamma), dev(beta)
    # reference path: separate finalize launch
    sc_ref, sh_ref = torch.empty(B, cin, device="cuda"), torch.empty(B, cin, device="cuda")
    _lib.check(lib.pf_gn_finalize_tiles(s0.data_ptr(), tiles, c0, _lib.ptr(s1), tiles + 1 if c1 else 0, c1, B, H * W, 32, 1e-5,
                                        g.data_ptr(), bt.data_ptr(), sc_ref.data_ptr(), sh_ref.data_ptr(), _lib.current_stream()))
    kw = dict(x0=x0, c0=c0, x1=x1, c1=c1, batch=B, hin=H, win=W, ks=3, stride=1, ups=0, w=pack3(lib, w), n=cout, prologue=1,
              bias=dev(bias), ld_out=cout, precision=1)
    out_ref = torch.empty(B, H, W, cout, device="cuda")
    run_conv(lib, sc=sc_ref, sh=sh_ref, out=out_ref, **kw)
    # fused path: sc / sh start as NaN and are written by the kernel
    sc, sh = torch.full((B, cin), float("nan"), device="cuda"), torch.full((B, cin), float("nan"), device="cuda")
    out = torch.empty(B, H, W, cout, device="cuda")
    run_conv(lib, sc=sc, sh=sh, out=out, gn_stats0=s0, gn_tiles0=tiles, gn_stats1=s1 if c1 else 0, gn_tiles1=tiles + 1 if c1 else 0,
             gn_gamma=g, gn_beta=bt, gn_eps=1e-5, gn_groups=32, **kw)
    assert (sc - sc_ref).abs().max() < 1e-6 and (sh - sh_ref).abs().max() < 1e-6
    assert (out - out_ref).abs().max() < 1e-5
    assert (out.cpu() - nhwc(ref)).abs().max() < TOL_OP
    # refused where the kernel cannot do it
    with pytest.raises(RuntimeError, match="fused GroupNorm"):
        run_conv(lib, sc=sc, sh=sh, out=out, gn_stats0=s0, gn_tiles0=tiles, gn_gamma=g, gn_beta=bt, gn_eps=1e-5, gn_groups=32,
                 **dict(kw, precision=0))


@pytest.mark.parametrize("B,L", [(1, 1024), (2, 1024), (1, 512), (3, 1024)])
def test_attention_key_split_form_small_batches(lib, B, L):
    """pf_attention_bf16x3_split: at batch 1 / 2 the keys of every 128-query tile are walked by four workgroups and merged by a second launch
    (online-softmax combination) - against torch, against the one-launch form, fp32 and plane outputs, bit-reproducible; a shape that fills
    the chip (B = 3 at L = 1024 does not qualify) reports zero scratch and runs the one launch."""
    H, c = 4, 256
    x = rnd((B, L, c), 181) * 1.3 + 0.2
    gamma, beta = 1 + 0.1 * rnd((c,), 182), 0.1 * rnd((c,), 183)
    w = rnd((3 * c, c), 184, c ** -0.5) * 1.5
    qkv = F.linear(F.layer_norm(x, (c,), gamma, beta), w)
    q, k, v = (t.reshape(B, L, H, 64) for t in qkv.chunk(3, dim=-1))
    att = (torch.einsum("bihd,bjhd->bhij", q, k) * 0.125).softmax(-1)
    ref = torch.einsum("bhij,bjhd->bihd", att, v).reshape(B, L, c)
    xd = dev(x)
    mu, rs = torch.empty(B * L, device="cuda"), torch.empty(B * L, device="cuda")
    st = _lib.current_stream()
    _lib.check(lib.pf_ln_stats(xd.data_ptr(), B * L, c, 1e-5, mu.data_ptr(), rs.data_ptr(), st))
    planes = torch.zeros(B * L * 3 * c, dtype=torch.float32, device="cuda")
    run_conv(lib, x0=xd, c0=c, batch=B, hin=1, win=L, ks=1, stride=1, ups=0, w=pack3(lib, w), n=3 * c, prologue=3, sc=dev(gamma), sh=dev(beta), mean=mu, rstd=rs,
             out=torch.empty(1, device="cuda"), ld_out=3 * c, precision=1, qkv_planes=planes)
    need = int(lib.pf_attention_split_scratch_bytes(B, H, L))
    assert (need > 0) == (B <= 2)
    scratch = torch.empty(max(need, 16), dtype=torch.uint8, device="cuda")
    one = torch.empty(B, L, c, device="cuda")
    _lib.check(lib.pf_attention_bf16x3(planes.data_ptr(), one.data_ptr(), c, None, B, H, L, 0, st))
    out = torch.empty(B, L, c, device="cuda")
    _lib.check(lib.pf_attention_bf16x3_split(planes.data_ptr(), out.data_ptr(), c, None, B, H, L, scratch.data_ptr(), need, st))
    torch.cuda.synchronize()
    assert (out.cpu() - ref).abs().max().item() < 3e-4
    d = (out - one).abs().max().item()
    assert d < 2e-5 and (need == 0) == (d == 0.0)          # merged partial sums: another summation order; no split: the same launch
    op = torch.zeros(B * L * c, device="cuda")
    _lib.check(lib.pf_attention_bf16x3_split(planes.data_ptr(), None, c, op.data_ptr(), B, H, L, scratch.data_ptr(), need, st))
    pl = op.view(_lib.x3_torch_dtype()).float().view(2, B, L, c)
    assert (pl[0] + pl[1] - out).abs().max().item() < 2e-5 * max(1.0, out.abs().max().item())
    for _ in range(4):
        o2 = torch.empty_like(out)
        _lib.check(lib.pf_attention_bf16x3_split(planes.data_ptr(), o2.data_ptr(), c, None, B, H, L, scratch.data_ptr(), need, st))
        assert torch.equal(o2, out)
