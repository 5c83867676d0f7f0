"""Every convolution / linear launch of the UNet's layer shapes must reproduce its output bit for bit (no atomics, no
timing-dependent reads anywhere on the path).  Per launch this is far more sensitive than the end-to-end check in
test_gpu_bf16x3.py: a hazard that corrupts one accumulator in one of eight runs shows up here.

The shapes are tests/layer_launch.py's own list: no tool edit can change what runs here."""
import os
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import layer_launch  # noqa: E402

pytestmark = pytest.mark.gpu

PREC = {"f32": 0, "bf16x3": 1}


def _ids(prec):
    return [s for s in layer_launch.SHAPES if layer_launch.supported(s, PREC[prec])]


@pytest.mark.parametrize("precision", ["bf16x3", "f32"])
def test_every_layer_shape_is_bit_reproducible(precision):
    shapes = _ids(precision)
    assert len(shapes) >= (28 if precision == "bf16x3" else 15)
    failures = []
    for s in shapes:
        bad, first = layer_launch.Launch(s, PREC[precision]).differing_runs(8)
        if bad:
            failures.append(f"{s[0]}: {bad}/8 runs differ; {first}")
    assert not failures, "\n".join(failures)


def test_unsupported_modes_are_refused_in_f32():
    """Plane operands and the fused skip projection are bf16x3-only: fp32 mode answers PF_EINVAL instead of computing something else."""
    for s in layer_launch.SHAPES:
        if not layer_launch.supported(s, 0):
            with pytest.raises(Exception):
                layer_launch.Launch(s, 0).run()


@pytest.mark.parametrize("precision", ["bf16x3"])
def test_layer_shapes_stress(precision):
    """The bounded form of tools/stress_determinism.py: the hand-scheduled kernels (3x3 conv incl. fused skip and K split,
    planes GEMMs) 60 more times each - a 1-in-8 hazard survives this with probability 3e-4."""
    names = ("r16_256_256", "r16_256+256_256", "r32_256_256", "r64_128_128", "r128_64_64", "rs64_128_128", "rs32_256_256", "up64_128",
             "pff1_1024_256_2048", "pqkv_1024_256_768", "pff2_1024_1024_256", "p256_256_256")
    failures = []
    for s in layer_launch.SHAPES:
        if s[0] in names:
            bad, first = layer_launch.Launch(s, PREC[precision]).differing_runs(60)
            if bad:
                failures.append(f"{s[0]}: {bad}/60 runs differ; {first}")
    assert not failures, "\n".join(failures)


@pytest.mark.parametrize("batch", [1, 3, 8])
@pytest.mark.parametrize("precision", ["bf16x3", "f32"])
def test_small_batch_unet_is_bit_reproducible(batch, precision):
    """The small-batch dispatch (split-K slices + reduce kernel, intra-workgroup K split, narrow tiles) on the full-size UNet:
    five evaluations of the same input, identical bit patterns."""
    from polyffusion_amd.arch import UNetConfig
    from polyffusion_amd.unet import UNetModel
    from polyffusion_amd.weights import synth_unet_state
    cfg = UNetConfig(d_cond=512)
    m = UNetModel(in_channels=2, out_channels=2, channels=64, n_res_blocks=2, attention_levels=(2, 3),
                  channel_multipliers=(1, 2, 4, 4), n_heads=4, tf_layers=1, d_cond=512)
    m.load_state_dict(synth_unet_state(cfg, 0))
    m.set_precision(precision)
    g = torch.Generator().manual_seed(50 + batch)
    x = torch.randn(batch, 2, 128, 128, generator=g).cuda()
    t = torch.randint(0, 1000, (batch,), generator=g).cuda()
    c = torch.randn(batch, 1, 512, generator=g).cuda()
    ref = m(x, t, c).clone()
    assert torch.isfinite(ref).all()
    for _ in range(4):
        assert torch.equal(m(x, t, c), ref)


@pytest.mark.parametrize("batch,reps", [(16, 40), (5, 40)])
def test_full_unet_stress_bf16x3(batch, reps):
    """tools/stress_determinism.py with a bounded repeat count: the whole B=16 plan (and a ragged batch) evaluated `reps` times."""
    from polyffusion_amd.arch import UNetConfig
    from polyffusion_amd.unet import UNetModel
    from polyffusion_amd.weights import synth_unet_state
    cfg = UNetConfig(d_cond=512)
    m = UNetModel(in_channels=2, out_channels=2, channels=64, n_res_blocks=2, attention_levels=(2, 3),
                  channel_multipliers=(1, 2, 4, 4), n_heads=4, tf_layers=1, d_cond=512)
    m.load_state_dict(synth_unet_state(cfg, 0))
    m.set_precision("bf16x3")
    g = torch.Generator().manual_seed(batch)
    x = torch.randn(batch, 2, 128, 128, generator=g).cuda()
    t = torch.randint(0, 1000, (batch,), generator=g).cuda()
    c = torch.randn(batch, 1, 512, generator=g).cuda()
    ref = m(x, t, c).clone()
    bad = sum(int(not torch.equal(m(x, t, c), ref)) for _ in range(reps))
    assert bad == 0, f"{bad}/{reps} evaluations differ"


@pytest.mark.parametrize("L,B,wide", [(1024, 16, "1"), (1024, 16, "0"), (256, 16, "0"), (512, 5, "1")])
def test_attention_is_bit_reproducible(L, B, wide, monkeypatch):
    """Both forms of the bf16x3 self-attention (hand-counted LDS / direct-to-LDS waits, hidden fragment loads, a static filler
    schedule) repeated on one input: every repeat must give the same bits, in both output formats."""
    from polyffusion_amd import _lib
    form = int(wide)
    lib = _lib.load()
    H, c = 4, 256
    g = torch.Generator().manual_seed(L + B)
    planes = (torch.randn(B * L * 3 * c * 2, generator=g) * 0.7).to(_lib.x3_torch_dtype()).cuda()
    st = _lib.current_stream()
    for as_planes in (False, True):
        outs = []
        for _ in range(24):
            out = torch.zeros(B, L, c, device="cuda")
            args = (None, c, out.data_ptr()) if as_planes else (out.data_ptr(), c, None)
            _lib.check(lib.pf_attention_bf16x3(planes.data_ptr(), args[0], args[1], args[2], B, H, L, form, st))
            outs.append(out)
        torch.cuda.synchronize()
        bad = sum(not torch.equal(outs[0], o) for o in outs[1:])
        assert bad == 0, f"{bad}/23 repeats differ (planes output: {as_planes})"
        assert torch.isfinite(outs[0]).all() if not as_planes else True
