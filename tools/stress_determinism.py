"""Stress form of the bit-reproducibility tests: hundreds of evaluations of the full-size UNet per batch size and precision."""
import sys, torch
sys.path.insert(0, '.')
from polyffusion_amd.arch import UNetConfig
from polyffusion_amd.unet import UNetModel
from polyffusion_amd.weights import synth_unet_state
cfg = UNetConfig(d_cond=512)
for prec in ("bf16x3", "f32"):
    m = UNetModel(in_channels=2, out_channels=2, channels=64, n_res_blocks=2, attention_levels=(2, 3), channel_multipliers=(1, 2, 4, 4), n_heads=4, tf_layers=1, d_cond=512)
    m.load_state_dict(synth_unet_state(cfg, 0)); m.set_precision(prec)
    for B in (16, 32, 5):
        g = torch.Generator().manual_seed(B)
        x = torch.randn(B, 2, 128, 128, generator=g).cuda(); t = torch.randint(0, 1000, (B,), generator=g).cuda(); c = torch.randn(B, 1, 512, generator=g).cuda()
        ref = m(x, t, c).clone(); bad = 0
        n = 200 if prec == "bf16x3" else 60
        for _ in range(n):
            bad += int(not torch.equal(m(x, t, c), ref))
        print(prec, "B", B, "runs", n, "differing", bad, flush=True)
