"""Every convolution / linear launch of the UNet's layer shapes must reproduce its output bit for bit (no atomics, no
timing-dependent reads anywhere on the path).  Per launch this is far more sensitive than the end-to-end check in
test_gpu_bf16x3.py: a hazard that corrupts one accumulator in one of eight runs shows up here."""
import os
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision", ["bf16x3", "f32"])
def test_every_layer_shape_is_bit_reproducible(precision, capsys, monkeypatch):
    from tools import bench_conv
    monkeypatch.setenv("PF_DET", "1")
    monkeypatch.setattr(sys, "argv", ["bench_conv.py", precision])
    bench_conv.main()
    out = capsys.readouterr().out
    lines = [l for l in out.splitlines() if "deterministic:" in l]
    assert len(lines) >= 10, out
    assert all(l.rstrip().endswith("yes") for l in lines), out


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
