"""Where does the bf16x3 error of a badly-scaled UNet come from?  (CPU, float64; no GPU.)

The oracle is run in float64 as the truth, then again with the operands of ONE class of contraction rounded to what the
bf16x3 split represents (x -> bf16(x) + bf16(x - bf16(x)): 16-17 significant bits) - convs, linear layers, the attention score
product, the attention value product - and with fp32-rounded operands for comparison.  The eps error of each variant, relative
to max|eps|, says which contraction the stressed net is sensitive to.

    python tools/stress_diag.py [seed]
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))

from oracle import unet_ref  # noqa: E402
from polyffusion_amd.arch import UNetConfig  # noqa: E402


def split16(x):
    x32 = x.float()
    hi = x32.bfloat16().float()
    lo = (x32 - hi).bfloat16().float()
    return (hi + lo).double()


def r32(x):
    return x.float().double()


def run(w, cfg, x, t, c, classes, rnd):
    oc, ol, oe = F.conv2d, F.linear, torch.einsum

    def conv2d(a, wt, b=None, **kw):
        if ("conv3" in classes and wt.shape[-1] == 3) or ("conv1" in classes and wt.shape[-1] == 1):
            return oc(rnd(a), rnd(wt), b, **kw)
        return oc(a, wt, b, **kw)

    def linear(a, wt, b=None):
        a = a.to(wt.dtype)          # the sinusoidal time embedding is built in fp32 by the oracle
        if "linear" in classes:
            return ol(rnd(a), rnd(wt), b)
        return ol(a, wt, b)

    def einsum(eq, a, b):
        if eq.startswith("bihd,bjhd") and "qk" in classes:
            return oe(eq, rnd(a), rnd(b))
        if eq.startswith("bhij,bjhd") and "pv" in classes:
            return oe(eq, rnd(a), rnd(b))
        return oe(eq, a, b)
    F.conv2d, F.linear, torch.einsum = conv2d, linear, einsum
    try:
        with torch.no_grad():
            return unet_ref.unet_forward(w, cfg, x, t, c)
    finally:
        F.conv2d, F.linear, torch.einsum = oc, ol, oe


def main():
    from test_gpu_long_parity import stressed_state
    from polyffusion_amd.weights import synth_unet_state
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    stress = (sys.argv[2] if len(sys.argv) > 2 else "stress") == "stress"
    cfg = UNetConfig(d_cond=512)
    st = stressed_state(cfg, seed) if stress else synth_unet_state(cfg, seed)
    rng = np.random.Generator(np.random.PCG64(99 + seed))
    x = rng.standard_normal((2, 2, 128, 128)).astype(np.float32)
    hot = rng.random(x.shape) < 0.01
    if stress:
        x[hot] = 50.0 * np.sign(x[hot])
    c = ((3.0 if stress else 1.0) * rng.standard_normal((2, 1, 512))).astype(np.float32)
    t = torch.tensor([987, 12])
    x, c = torch.from_numpy(x).double(), torch.from_numpy(c).double()
    w = unet_ref.to_torch(st, dtype=torch.float64)
    torch.set_num_threads(32)
    truth = run(w, cfg, x, t, c, (), None)
    scale = truth.abs().max().item()
    print(f"seed {seed} stress={stress}: max|eps| {scale:.4g}  rms {truth.pow(2).mean().sqrt().item():.4g}")
    w32 = unet_ref.to_torch(st)
    with torch.no_grad():
        e32 = unet_ref.unet_forward(w32, cfg, x.float(), t, c.float()).double()
    print(f"  fp32 oracle vs fp64 truth: {(e32 - truth).abs().max().item() / scale:.3e} rel")
    for classes in (("conv3",), ("conv1",), ("linear",), ("qk",), ("pv",), ("conv3", "conv1", "linear", "qk", "pv")):
        e = run(w, cfg, x, t, c, classes, split16)
        print(f"  split16 operands in {'+'.join(classes):32s}: {(e - truth).abs().max().item() / scale:.3e} rel")
    e = run(w, cfg, x, t, c, ("conv3", "conv1", "linear", "qk", "pv"), r32)
    print(f"  fp32-rounded operands everywhere          : {(e - truth).abs().max().item() / scale:.3e} rel")


if __name__ == "__main__":
    main()
