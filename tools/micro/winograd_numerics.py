#!/usr/bin/env python
"""What would Winograd F(2x2, 3x3) cost in accuracy under the split arithmetic?  (CPU, no GPU.)

A 3x3 convolution as 16 channel-contractions per 2x2 output tile instead of 36 (2.25x fewer MFMA flops); the input transform B^T d B and
the output transform A^T m A are additions, the filter transform G g G^T is done once on the host.  The contraction operands are what gets
split into hi + lo pieces, so the question is how much the transforms amplify the split's operand rounding.  Emulation: operands rounded to
hi + lo (bf16 or fp16 pieces, the fp16 filter pieces of 256 U as in libpfhip_f16.so), products and sums in float64 - i.e. only the
operand-representation error the split adds; the fp32 accumulation is common to all variants and left out.

    python tools/micro/winograd_numerics.py
"""
import numpy as np
import torch
import torch.nn.functional as F

torch.manual_seed(0)
torch.set_num_threads(16)


def split(x, dt, scale=1.0):
    x32 = (x * scale).float()
    hi = x32.to(dt).float()
    lo = (x32 - hi).to(dt).float()
    return ((hi.double() + lo.double()) / scale)


def run(cin, cout, hw, act_scale=1.0, w_scale=1.0):
    x = F.silu(torch.randn(2, cin, hw, hw, dtype=torch.float64) * act_scale)
    w = torch.randn(cout, cin, 3, 3, dtype=torch.float64) * w_scale / np.sqrt(9 * cin)
    ref = F.conv2d(x, w, padding=1)
    sc = ref.abs().max().item()
    Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
    At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
    xp = F.pad(x, (1, 1, 1, 1))
    # tiles: 4x4 input windows at stride 2
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                      # [B, C, th, tw, 4, 4]
    V = torch.einsum("ij,bcyxjk,lk->bcyxil", Bt, d.float().double(), Bt)      # input transform (fp32 inputs; adds are exact enough in f64 here)
    V32 = V.float().double()                                    # the transformed tile is held in fp32 before the split
    U = torch.einsum("ij,ocjk,lk->ocil", G, w, G)               # filter transform on the host in f64
    out = {}
    for name, dt, ws in (("bf16x3", torch.bfloat16, 1.0), ("f16x3", torch.float16, 256.0)):
        # direct
        xd, wd = split(x, dt), split(w, dt, ws)
        direct = F.conv2d(xd, wd, padding=1)
        # winograd
        Vs, Us = split(V32, dt), split(U, dt, ws)
        M = torch.einsum("bcyxil,ocil->boyxil", Vs, Us)
        Y = torch.einsum("ij,boyxjk,lk->boyxil", At, M, At)     # [B, O, th, tw, 2, 2]
        wino = Y.permute(0, 1, 2, 4, 3, 5).reshape(ref.shape)
        out[name] = ((direct - ref).abs().max().item() / sc, (wino - ref).abs().max().item() / sc)
    # winograd with exact operands, transformed tile rounded to fp32 only: the transforms' own contribution
    M = torch.einsum("bcyxil,ocil->boyxil", V32, U)
    Y = torch.einsum("ij,boyxjk,lk->boyxil", At, M, At)
    w32 = ((Y.permute(0, 1, 2, 4, 3, 5).reshape(ref.shape) - ref).abs().max().item() / sc)
    d32 = ((F.conv2d(x.float().double(), w.float().double(), padding=1) - ref).abs().max().item() / sc)
    print(f"Cin {cin:4d} Cout {cout:4d} {hw}x{hw} act x{act_scale:g} w x{w_scale:g}: relative to max|out|  "
          f"fp32 operands direct {d32:.1e} / winograd {w32:.1e};  bf16x3 direct {out['bf16x3'][0]:.1e} / winograd {out['bf16x3'][1]:.1e};  "
          f"f16x3 direct {out['f16x3'][0]:.1e} / winograd {out['f16x3'][1]:.1e}")


for args in ((64, 64, 32), (128, 128, 32), (256, 256, 16), (128, 128, 32, 10.0, 1.0), (128, 128, 32, 1.0, 0.1)):
    run(*args)
