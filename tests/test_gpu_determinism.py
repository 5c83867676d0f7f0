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
