"""The UNet's layer shapes (B=16) as single pf_conv2d launches: the inputs of tests/test_gpu_determinism.py.

The shape list lives HERE, with the test that depends on it; tools/bench_conv.py times the same launches (and may add its own
experimental shapes on top) but nothing a tool does can change what the test runs."""
import ctypes as C

import torch

from polyffusion_amd import _lib

# name, B, H, W, c0, c1, cout, ks, stride, ups, prologue[, mode[, skip_c0, skip_c1]]
#   mode 1: A operand as pre-split bf16 hi/lo planes   2: GeGLU product out as planes   3: q|k|v planes out
#        4: planes in, planes out (+ residual)          5: planes out, no residual       6: fp32 out, no residual
#        7: second conv of a channel-changing ResBlock, the 1x1 skip projection of concat(x, skip) fused in
#        8: the fused Winograd F(2x2, 3x3) form of the ResBlock conv (pf_conv_args.wino, csrc/conv_wino.hip)
SHAPES = [
    ("r128_64_64", 16, 128, 128, 64, 0, 64, 3, 1, 0, 1),
    ("r128_128+64_64", 16, 128, 128, 128, 64, 64, 3, 1, 0, 1),
    ("r64_128_128", 16, 64, 64, 128, 0, 128, 3, 1, 0, 1),
    ("r64_256+128_128", 16, 64, 64, 256, 128, 128, 3, 1, 0, 1),
    ("r32_256_256", 16, 32, 32, 256, 0, 256, 3, 1, 0, 1),
    ("r32_256+256_256", 16, 32, 32, 256, 256, 256, 3, 1, 0, 1),
    ("r16_256_256", 16, 16, 16, 256, 0, 256, 3, 1, 0, 1),
    ("r16_256+256_256", 16, 16, 16, 256, 256, 256, 3, 1, 0, 1),
    ("up64_128", 16, 64, 64, 128, 0, 128, 3, 1, 1, 0),
    ("down128_64", 16, 128, 128, 64, 0, 64, 3, 2, 0, 0),
    ("rs128_64_64", 16, 128, 128, 64, 0, 64, 3, 1, 0, 1, 7, 128, 64),
    ("rs64_128_128", 16, 64, 64, 128, 0, 128, 3, 1, 0, 1, 7, 256, 128),
    ("rs32_256_256", 16, 32, 32, 256, 0, 256, 3, 1, 0, 1, 7, 256, 256),
    ("w128_64_64", 16, 128, 128, 64, 0, 64, 3, 1, 0, 1, 8),
    ("w128_128+64_64", 16, 128, 128, 128, 64, 64, 3, 1, 0, 1, 8),
    ("w64_256+128_128", 16, 64, 64, 256, 128, 128, 3, 1, 0, 1, 8),
    ("w32_256_256", 16, 32, 32, 256, 0, 256, 3, 1, 0, 1, 8),
    ("w32_256+256_256", 16, 32, 32, 256, 256, 256, 3, 1, 0, 1, 8),
    ("g1024_256_256", 16, 1, 1024, 256, 0, 256, 1, 1, 0, 0),
    ("g1024_256_768ln", 16, 1, 1024, 256, 0, 768, 1, 1, 0, 3),
    ("g1024_1024_256", 16, 1, 1024, 1024, 0, 256, 1, 1, 0, 0),
    ("g256_256_256", 16, 1, 256, 256, 0, 256, 1, 1, 0, 0),
    ("gn1024_256_256", 16, 1, 1024, 256, 0, 256, 1, 1, 0, 2),   # SpatialTransformer.proj_in (GroupNorm affine prologue), 32x32 level
    ("gn256_256_256", 16, 1, 256, 256, 0, 256, 1, 1, 0, 2),      # ... 16x16 level
    ("skip128_192_64", 16, 1, 16384, 128, 64, 64, 1, 1, 0, 0),
    ("p1024_256_256", 16, 1, 1024, 256, 0, 256, 1, 1, 0, 0, 1),
    ("p1024_1024_256", 16, 1, 1024, 1024, 0, 256, 1, 1, 0, 0, 1),
    ("p256_256_256", 16, 1, 256, 256, 0, 256, 1, 1, 0, 0, 1),
    ("p256_1024_256", 16, 1, 256, 1024, 0, 256, 1, 1, 0, 0, 1),
    ("pff1_1024_256_2048", 16, 1, 1024, 256, 0, 2048, 1, 1, 0, 0, 2),
    ("pwide_1024_256_2048", 16, 1, 1024, 256, 0, 2048, 1, 1, 0, 0, 1),
    ("pqkv_1024_256_768", 16, 1, 1024, 256, 0, 768, 1, 1, 0, 0, 3),
    ("pff2_1024_1024_256", 16, 1, 1024, 1024, 0, 256, 1, 1, 0, 0, 4),
    ("pff2nores_1024_1024_256", 16, 1, 1024, 1024, 0, 256, 1, 1, 0, 0, 5),
    ("pnores_1024_1024_256", 16, 1, 1024, 1024, 0, 256, 1, 1, 0, 0, 6),
]


def shape_mode(shape) -> int:
    return shape[11] if len(shape) > 11 else 0


def supported(shape, prec: int) -> bool:
    """Plane operands / outputs and the fused skip projection exist only in the bf16x3 mode (pf_conv2d answers PF_EINVAL
    for them in f32 mode, as the header says)."""
    return prec == 1 or shape_mode(shape) in (0, 8)     # (8: pf_conv_args.wino in f32 mode runs the direct form - nothing to refuse)


class Launch:
    """One pf_conv2d launch on random operands.  Keeps every tensor alive; `run()` enqueues it on the current stream."""

    def __init__(self, shape, prec: int, seed: int = 0):
        name, B, H, W, c0, c1, n, ks, stride, ups, pro, *rest = shape
        self.name, self.n = name, n
        mode = shape_mode(shape)
        lib = self.lib = _lib.load()
        g = torch.Generator(device="cuda").manual_seed(seed)
        rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
        cin, taps = c0 + c1, ks * ks
        x0 = rn(B, H, W, c0)
        x1 = rn(B, H, W, c1) if c1 else None
        w = rn(lib.pf_packed_gemm_weight_floats(n, cin, taps)) * 0.01
        ho, wo = (H * 2, W * 2) if ups else ((H // 2, W // 2) if stride == 2 else (H, W))
        self.out = out = torch.empty(B, ho, wo, n, device="cuda")
        sc = torch.ones(B, cin, device="cuda"); sh = torch.zeros(B, cin, device="cuda")
        mean = torch.zeros(B * H * W, device="cuda"); rstd = torch.ones(B * H * W, device="cuda")
        bias = torch.zeros(n, device="cuda")
        res = rn(B, ho, wo, n)
        a = self.args = _lib.ConvArgs()
        a.x0, a.c0, a.x1, a.c1 = x0.data_ptr(), c0, (x1.data_ptr() if c1 else 0), c1
        a.batch, a.hin, a.win, a.ks, a.stride, a.ups = B, H, W, ks, stride, ups
        a.w, a.n, a.prologue = w.data_ptr(), n, pro
        a.sc, a.sh, a.mean, a.rstd = sc.data_ptr(), sh.data_ptr(), mean.data_ptr(), rstd.data_ptr()
        a.bias, a.res, a.ld_res = bias.data_ptr(), res.data_ptr(), n
        a.out, a.ld_out, a.precision = out.data_ptr(), n, prec
        a.a_planes = int(mode not in (0, 7, 8))   # same bytes as fp32 [M][K]: random bits are fine for timing / reproducibility
        self.keep = [x0, x1, w, sc, sh, mean, rstd, bias, res]
        if mode == 2:
            a.geglu, a.ld_out, a.out_planes, a.res = 1, n // 2, out.data_ptr(), 0
        if mode in (4, 5):
            a.out_planes = out.data_ptr()
        if mode in (5, 6):
            a.res = 0
        if mode == 3:
            a.qkv_planes, a.res = out.data_ptr(), 0
        skip_k = 0
        if mode == 8:
            ww = torch.randint(0, 2 ** 15, (lib.pf_wino_weight_bytes(n, cin) // 2,), device="cuda", dtype=torch.int16, generator=g)   # finite pieces
            a.w_wino, a.wino = ww.data_ptr(), 1
            self.keep.append(ww)
        if mode == 7:
            sc0, sc1 = rest[1], rest[2]
            sx0 = rn(B, H, W, sc0); sx1 = rn(B, H, W, sc1)
            sw = rn(lib.pf_packed_gemm_weight_floats(n, sc0 + sc1, 1)) * 0.01
            a.skip_x0, a.skip_c0, a.skip_x1, a.skip_c1, a.skip_w, a.res = sx0.data_ptr(), sc0, sx1.data_ptr(), sc1, sw.data_ptr(), 0
            self.keep += [sx0, sx1, sw]
            skip_k = sc0 + sc1
        self.gflop = 2.0 * B * ho * wo * n * (cin * taps + skip_k) / 1e9
        wsb = int(lib.pf_conv_splitk_ws_bytes(C.byref(a)))   # a layer the library would split over K gets the scratch the plan gives it
        if wsb:
            ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
            a.splitk_ws, a.splitk_ws_bytes = ws.data_ptr(), wsb
            self.keep.append(ws)

    def run(self, check: bool = True):
        rc = self.lib.pf_conv2d(C.byref(self.args), torch.cuda.current_stream().cuda_stream)
        if check:
            _lib.check(rc)

    def differing_runs(self, reps: int = 8):
        """Bit patterns of `reps` launches against a first one (the random packed weights hold NaNs: compare as int32).
        Returns (number of runs that differ, description of the first difference or '')."""
        for _ in range(3):
            self.run()
        self.out.zero_()   # plane outputs cover only part of the buffer
        self.run()
        torch.cuda.synchronize()
        ref = self.out.clone()
        bad, first = 0, ""
        for _ in range(reps):
            self.out.zero_()
            self.run()
            torch.cuda.synchronize()
            ne = self.out.view(torch.int32) != ref.view(torch.int32)
            if ne.any():
                bad += 1
                if not first:
                    idx = ne.flatten().nonzero().flatten()
                    first = (f"{int(ne.sum())} elements differ, first flat indices {idx[:6].tolist()} (row {int(idx[0]) // self.n}, "
                             f"col {int(idx[0]) % self.n}); values {self.out.flatten()[idx[:3]].tolist()} vs {ref.flatten()[idx[:3]].tolist()}")
        return bad, first
