"""CPU-side checks (-m "not gpu"): the C-ABI library loads and exports every symbol declared in
include/pfhip.h, host-only entry points work without a GPU (plan / parameter table / weight
packing), the params surface, masks and the batch-sharding helper (world_size-2 gloo)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from polyffusion_amd import _lib
from polyffusion_amd.arch import UNetConfig, unet_layout, unet_param_shapes
from polyffusion_amd.params import PRESETS, find_params, load_params, preset

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        from polyffusion_amd.build import build
        build(verbose=False)
    return _lib.load()


def test_header_symbols_exported_and_bound(lib):
    hdr = open(os.path.join(REPO, "include", "pfhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(pf_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in pfhip.h but not exported by libpfhip.so"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.pf_version() >= 100


def test_both_builds_of_the_library_load_side_by_side_and_say_which_they_are(lib):
    """libpfhip.so (bf16 pieces) and libpfhip_f16.so (fp16 pieces, -DPF_X3_F16) are the same sources and export the same C ABI;
    pf_x3_element() tells them apart; one process can hold both (UNetModel(..., x3="f16"))."""
    if not os.path.exists(_lib.lib_path("f16")):
        from polyffusion_amd.build import build
        build(verbose=False, variant="f16")
    l16 = _lib.load("f16")
    for name in _lib.SIGNATURES:
        assert hasattr(l16, name), f"{name} not exported by libpfhip_f16.so"
    assert l16.pf_x3_element() == 1 and _lib.load("").pf_x3_element() == 0
    assert l16 is not _lib.load("") and l16 is _lib.load("f16")
    assert l16.pf_version() == _lib.load("").pf_version()
    # host-side packing differs (fp16 pieces of 256 w against bf16 pieces of w): same byte count, other bytes
    import numpy as np
    w = np.random.default_rng(0).standard_normal((64, 16)).astype(np.float32) * 0.05
    outs = []
    for L in (_lib.load(""), l16):
        dst = np.zeros(2 * 64 * 16, dtype=np.uint16)
        assert L.pf_pack_gemm_weight_bf16x3(w.ctypes.data, 64, 16, 1, dst.ctypes.data) == 0
        outs.append(dst)
    hi16 = outs[1].view(np.float16).reshape(2, 2, 64, 8)       # [K/8][plane][N][8]
    rec = (hi16[:, 0].astype(np.float64) + hi16[:, 1].astype(np.float64)) / 256.0
    assert np.abs(rec.transpose(1, 0, 2).reshape(64, 16) - w).max() < 2.0 ** -21 * 0.3       # 22 mantissa bits of |w| < 0.3
    b = (outs[0].astype(np.uint32) << 16).view(np.float32).reshape(2, 2, 64, 8)
    rec_b = b[:, 0].astype(np.float64) + b[:, 1]
    assert np.abs(rec_b.transpose(1, 0, 2).reshape(64, 16) - w).max() < 2.0 ** -16 * 0.3


def _unet(lib, cfg: UNetConfig, h=128, w=128):
    from polyffusion_amd.unet import UNetModel
    return UNetModel(in_channels=cfg.in_channels, out_channels=cfg.out_channels, channels=cfg.channels,
                     n_res_blocks=cfg.n_res_blocks, attention_levels=cfg.attention_levels,
                     channel_multipliers=cfg.channel_multipliers, n_heads=cfg.n_heads, tf_layers=cfg.tf_layers,
                     d_cond=cfg.d_cond, img_h=h, img_w=w)


def test_plan_param_table_is_the_reference_key_namespace(lib):
    for cfg in (UNetConfig(d_cond=512), UNetConfig(d_cond=1024),
                UNetConfig(channels=32, n_res_blocks=1, attention_levels=(1,), channel_multipliers=(1, 2), n_heads=2, d_cond=32)):
        m = _unet(lib, cfg)
        assert m.param_shapes() == dict(unet_param_shapes(cfg))
    assert len(unet_param_shapes(UNetConfig(d_cond=512))) == 556
    assert sum(int(np.prod(s)) for s in unet_param_shapes(UNetConfig(d_cond=512)).values()) == 41_082_370


def test_layout_block_table():
    lay = unet_layout(UNetConfig(d_cond=512))
    assert len(lay.input_blocks) == 12 and len(lay.output_blocks) == 12
    assert [b[0][1] for b in lay.output_blocks] == [512, 512, 512, 512, 512, 384, 384, 256, 192, 192, 128, 128]
    assert lay.skip_channels == [256, 256, 256, 256, 256, 128, 128, 128, 64, 64, 64, 64]


def test_workspace_and_launch_count_without_gpu(lib):
    m = _unet(lib, UNetConfig(d_cond=512))
    ws16 = lib.pf_unet_workspace_bytes(m._h, 16, 1)
    ws1 = lib.pf_unet_workspace_bytes(m._h, 1, 1)
    assert 0 < ws1 < ws16 < 8 << 30
    assert 200 < m.n_launches(16) < 500
    assert m.weight_bytes() > 41_082_370 * 4


def test_gemm_weight_packing_layout(lib):
    n, k = 96, 32
    w = np.arange(n * k * 9, dtype=np.float32).reshape(n, k, 3, 3)
    dst = np.zeros(lib.pf_packed_gemm_weight_floats(n, k, 9), np.float32)
    assert dst.size == 9 * k * 128
    _lib.check(lib.pf_pack_gemm_weight(w.ctypes.data, n, k, 9, dst.ctypes.data))
    p = dst.reshape(9, k // 4, 128, 4)
    for tap, kk, nn in ((0, 0, 0), (4, 17, 95), (8, 31, 3)):
        assert p[tap, kk // 4, nn, kk % 4] == w[nn, kk, tap // 3, tap % 3]
    assert (p[:, :, 96:, :] == 0).all()


def test_pack_state_dict_errors_without_gpu(lib):
    from polyffusion_amd.weights import synth_unet_state
    cfg = UNetConfig(channels=32, n_res_blocks=1, attention_levels=(1,), channel_multipliers=(1, 2), n_heads=2, d_cond=32)
    m = _unet(lib, cfg, 32, 32)
    st = synth_unet_state(cfg, 0)
    blob = m.pack_state_dict(st)
    assert blob.numel() * 4 == m.weight_bytes() and float(blob.abs().sum()) > 0
    bad = dict(st); bad["bogus"] = np.zeros(1, np.float32)
    with pytest.raises(RuntimeError, match="unexpected key"):
        _unet(lib, cfg, 32, 32).pack_state_dict(bad)
    bad = dict(st); del bad["time_embed.0.bias"]
    with pytest.raises(RuntimeError, match="missing"):
        _unet(lib, cfg, 32, 32).pack_state_dict(bad)
    with pytest.raises(RuntimeError, match="channels must be a multiple of 32"):
        _unet(lib, UNetConfig(channels=48))


def test_product_refuses_to_run_without_gpu(lib):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from polyffusion_amd.weights import synth_unet_state
    cfg = UNetConfig(channels=32, n_res_blocks=1, attention_levels=(1,), channel_multipliers=(1, 2), n_heads=2, d_cond=32)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _unet(lib, cfg, 32, 32).load_state_dict(synth_unet_state(cfg, 0))


def test_params_surface(tmp_path):
    p = preset("sdf_chd8bar")
    assert p.d_cond == 512 and p.cond_type == "chord" and p.channel_multipliers == [1, 2, 4, 4]
    assert preset("sdf_txt").d_cond == 1024 and set(PRESETS) >= {"sdf_chd8bar", "sdf_txt", "sdf_txtvnl"}
    run = tmp_path / "run"
    (run / "chkpts").mkdir(parents=True)
    import yaml
    (run / "params.yaml").write_text(yaml.safe_dump(dict(PRESETS["sdf_txt"], batch_size=16, learning_rate=5e-5)))
    found = find_params(str(run / "chkpts" / "weights_best.pt"))
    assert found.endswith("params.yaml")
    q = load_params(found)
    assert q.d_cond == 1024 and UNetConfig.from_params(q) == UNetConfig(d_cond=1024)
    with pytest.raises(FileNotFoundError):
        find_params(str(tmp_path / "a" / "b" / "c.pt"))


def test_masks_and_autoreg_data_match_oracle():
    from oracle import sampler_ref
    from polyffusion_amd.inference_sdf import get_autoreg_data, get_mask
    a = torch.arange(3 * 4 * 5, dtype=torch.float32).view(3, 4, 5)
    assert torch.equal(get_autoreg_data(a, 1), sampler_ref.get_autoreg_data(a, 1))
    rng = np.random.Generator(np.random.PCG64(0))
    orig = torch.from_numpy((rng.random((2, 2, 128, 128)) < 0.01).astype(np.float32))
    orig[:, :, :, 0] = 0
    orig[0, 0, 0:3] = 0  # leading empty steps
    assert torch.equal(get_mask(orig, "remaining"), orig)
    m = get_mask(orig, "bars", [1, 5])
    assert m[:, :, 16:32].sum() == 0 and m[:, :, 80:96].sum() == 0 and m[:, :, 0:16].min() == 1
    # row-loop statement of the reference's "below"/"above" (inference_sdf.py:138-181) as the check
    on = orig[:, 0].reshape(-1, 128)
    lo = on.argmax(1).clone()
    first = int(lo.nonzero()[0])
    lo[:first] = lo[first]
    for i in range(len(lo)):
        if lo[i] == 0:
            lo[i] = lo[i - 1]
    want = torch.zeros_like(on)
    for i in range(len(lo)):
        want[i, lo[i]:] = 1
    assert torch.equal(get_mask(orig, "below")[:, 0].reshape(-1, 128), want)
    hi = 127 - on.flip(1).argmax(1)
    first = int(hi.nonzero()[0])
    hi[:first] = hi[first]
    for i in range(len(hi)):
        if hi[i] == 127:
            hi[i] = hi[i - 1]
    want = torch.zeros_like(on)
    for i in range(len(hi)):
        want[i, 0:hi[i] + 1] = 1
    assert torch.equal(get_mask(orig, "above")[:, 0].reshape(-1, 128), want)


def test_shard_plan_world_size_2_gloo(tmp_path):
    """N>1 path on CPU: two gloo ranks agree on a disjoint cover of the batch and on the broadcast blob."""
    script = tmp_path / "w.py"
    script.write_text(f'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {REPO!r})
from polyffusion_amd.dist import shard_range, broadcast_blob, gather_rows, gather_floats, broadcast_int, ranks_seen
dist.init_process_group("gloo")
r, n = dist.get_rank(), dist.get_world_size()
lo, hi = shard_range(37, r, n)
got = [None] * n
dist.all_gather_object(got, (lo, hi))
assert got[0][0] == 0 and got[-1][1] == 37 and all(got[i][1] == got[i + 1][0] for i in range(n - 1)), got
blob = torch.arange(1000, dtype=torch.float32) if r == 0 else torch.zeros(1000)
broadcast_blob(blob, src=0)
assert blob.sum().item() == 499500.0
# ragged end-of-run gather: 5 rows over 2 ranks (3 + 2), every rank ends with rows 0..4 in order
a, b = shard_range(5, r, n)
rows = torch.arange(a, b, dtype=torch.float32)[:, None, None].expand(b - a, 2, 3).contiguous()
allrows = gather_rows(rows, 5, r, n)
assert allrows.shape == (5, 2, 3) and torch.equal(allrows[:, 0, 0], torch.arange(5.)), allrows
assert gather_rows(torch.zeros(0, 4) if r == 1 else torch.ones(1, 4), 1, r, n).shape == (1, 4)   # a rank with no rows
assert broadcast_int(1234 + r) == 1234
assert gather_floats(1.5 + r) == [1.5 + i for i in range(n)]     # bench.py's per-rank step times
assert ranks_seen()[0] == n
dist.barrier()
if r == 0: print("OK", got)
''')
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29617", str(script)],
                         capture_output=True, text=True, timeout=240, env=env)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr


def test_winograd_weight_packing_reproduces_the_convolution(lib):
    """pf_pack_wino_weight_bf16x3 on the host: U = G g G^T per (n, k), split into hi | lo pieces, laid out [i][K/16][N/64][j][nb][plane][lane][8]
    (csrc/conv_wino.hip).  Unpacked again here and used in a numpy Winograd F(2x2, 3x3) evaluation, the packing must reproduce a plain 3x3
    correlation (ref: stable_diffusion/model/unet.py:262-318 convs) to the split's 16-bit-mantissa accuracy - layout AND transform."""
    import numpy as np
    rng = np.random.default_rng(5)
    N, K, H, W = 64, 32, 8, 8
    w = (rng.standard_normal((N, K, 3, 3)) * 0.1).astype(np.float32)
    x = rng.standard_normal((K, H + 2, W + 2)).astype(np.float32)           # already padded input
    assert lib.pf_wino_weight_bytes(N, K) == 16 * K * N * 4
    dst = np.zeros(16 * K * N * 2, dtype=np.uint16)
    assert lib.pf_pack_wino_weight_bf16x3(w.ctypes.data, N, K, dst.ctypes.data) == 0
    assert lib.pf_pack_wino_weight_bf16x3(w.ctypes.data, 48, K, dst.ctypes.data) != 0     # n must be a multiple of 64
    d = dst.reshape(4, K // 16, N // 64, 4, 2, 2, 64, 8)                     # [i][kk][ntile][j][nb][plane][lane][e]
    if lib.pf_x3_element() == 0:
        piece = lambda u: (u.astype(np.uint32) << 16).view(np.float32)      # bf16 -> fp32
        scale = 1.0
    else:
        piece = lambda u: u.view(np.float16).astype(np.float32)
        scale = 1.0 / 256.0
    U = np.zeros((4, 4, K, N), np.float64)
    for kk in range(K // 16):
        for nb in range(2):
            for lane in range(64):
                for e in range(8):
                    k, n = kk * 16 + (lane >> 5) * 8 + e, nb * 32 + (lane & 31)
                    U[:, :, k, n] = (piece(d[:, kk, 0, :, nb, 0, lane, e]).astype(np.float64) + piece(d[:, kk, 0, :, nb, 1, lane, e])) * scale
    Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
    At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)
    out = np.zeros((N, H, W))
    for ty in range(H // 2):
        for tx in range(W // 2):
            dd = x[:, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4].astype(np.float64)          # [K,4,4]
            V = np.einsum("ia,kab,jb->ijk", Bt, dd, Bt)
            M = np.einsum("ijk,ijkn->ijn", V, U)
            out[:, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = np.einsum("pi,ijn,qj->npq", At, M, At)
    ref = np.zeros((N, H, W))
    for ky in range(3):
        for kx in range(3):
            ref += np.einsum("nk,khw->nhw", w[:, :, ky, kx].astype(np.float64), x[:, ky:ky + H, kx:kx + W].astype(np.float64))
    err = np.abs(out - ref).max() / np.abs(ref).max()
    assert err < 2e-5, err
