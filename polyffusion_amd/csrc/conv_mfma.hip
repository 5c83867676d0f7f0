// conv_mfma.hip - implicit-GEMM convolution / linear on NHWC activations for gfx950 (CDNA4).
//
// One kernel family covers every dense contraction of the denoiser that is not attention:
//   * ResBlock 3x3 convs with the GroupNorm+SiLU of their input fused into the tile load and
//     bias + time-embedding bias + residual fused into the store (unet.py:262-318),
//   * DownSample (stride 2) and UpSample (nearest x2 folded into the halo read) convs
//     (unet.py:218-259), skip 1x1 convs, SpatialTransformer proj_in/proj_out,
//   * every Linear of the transformer block with LayerNorm applied on load and
//     bias / residual / cross-attention bias / GeGLU applied on store (unet_attention.py:89-333).
//
// Mapping to the hardware (see DESIGN.md "conv_mfma"):
//   M = output pixels (rows), N = output channels, K = taps x input channels.
//   A workgroup (256 threads = 4 waves, 2x2) owns a TH x TW pixel tile x BN channels.  For each
//   BK-channel chunk the (TH-1)*S+KS by (TW-1)*S+KS input halo is fetched ONCE (coalesced 16-B
//   loads along channels), normalised/activated in registers and written to LDS; the KS*KS taps
//   then read shifted windows of that LDS image, so the input is read from HBM/L2 once instead
//   of nine times and GroupNorm/SiLU is evaluated once per element.  Weight tiles (BK x BN per tap)
//   are pre-packed as [tap][K/4][N][4] so a lane's MFMA B operands are one 16-B LDS read.
//   Arithmetic is v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain; the 1e-3 parity bar rules out
//   bf16 inputs - SURVEY.md 0).  Both LDS images are double buffered and the next tile's global
//   loads are in flight (registers) while the current tile's MFMAs run; one barrier per tile.
#include <stdlib.h>
#include <cstdlib>
#include "conv_common.h"

namespace pf {

template <int KS, int STRIDE, bool UPS, int TH, int TW, int BN, int BK, int PRO>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(ConvP p) {
  constexpr int BM = TH * TW;
  constexpr int THIN = (TH - 1) * STRIDE + KS;
  constexpr int TWIN = (TW - 1) * STRIDE + KS;
  constexpr int NPIX = THIN * TWIN;
  constexpr int BKP = BK + 4;
  constexpr int KQ = BK / 4;              // float4 per pixel per chunk
  constexpr int TOTA = NPIX * KQ;         // float4 per A image
  constexpr int NA = (TOTA + 255) / 256;
  constexpr int PSTEP = 256 / KQ;         // pixel stride between a thread's successive float4
  constexpr int TOTW = KQ * BN;           // float4 per W image
  constexpr int NW = TOTW / 256;
  constexpr int TAPS = KS * KS;
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int FM = WM / 32, FN = WN / 32;
  constexpr int PAD = (KS == 3) ? 1 : 0;
  static_assert(TOTW % 256 == 0, "weight tile must divide over the block");
  static_assert(FM >= 1 && FN >= 1, "tile too small");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;                         // [2][NPIX][BKP]
  float* sW = smem + 2 * NPIX * BKP;        // [2][KQ][BN][4]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- XCD-aware (bijective) block remap: consecutive logical tiles share an XCD's L2 ----
  const int nwg = gridDim.x;
  int lid;
  {
    const int orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  const int nti = lid % p.nt;
  int mt = lid / p.nt;
  const int tx = mt % p.tiles_x; mt /= p.tiles_x;
  const int ty = mt % p.tiles_y;
  const int b = mt / p.tiles_y;
  conv_shared_x1(p, b);
  const int n0 = nti * BN;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int iy0 = oy0 * STRIDE - PAD, ix0 = ox0 * STRIDE - PAD;
  const int Hlog = UPS ? 2 * p.Hin : p.Hin, Wlog = UPS ? 2 * p.Win : p.Win;
  const int cin = p.c0 + p.c1;

  // ---- per-thread staging geometry (constant across chunks) ----
  const int c4 = tid % KQ;
  int poff[NA];            // source pixel index or -1 (zero padding / out of tile)
  float pmu[NA], prs[NA];  // LayerNorm row statistics (PRO == 3)
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int pix = tid / KQ + i * PSTEP;
    poff[i] = -1;
    pmu[i] = 0.f; prs[i] = 0.f;
    if (pix < NPIX) {
      const int hy = pix / TWIN, hx = pix % TWIN;
      const int iy = iy0 + hy, ix = ix0 + hx;
      if (iy >= 0 && iy < Hlog && ix >= 0 && ix < Wlog) {
        const int sy = UPS ? (iy >> 1) : iy, sx = UPS ? (ix >> 1) : ix;
        poff[i] = (b * p.Hin + sy) * p.Win + sx;
        if (PRO == 3) { pmu[i] = p.mean[poff[i]]; prs[i] = p.rstd[poff[i]]; }
      }
    }
  }

  f32x4 ra[NA], rw[NW], vsc, vsh;
  vsc = f32x4{1.f, 1.f, 1.f, 1.f}; vsh = f32x4{0.f, 0.f, 0.f, 0.f};

  auto loadA = [&](int chunk) {
    const int cg = chunk * BK;
    const float* src; int cs, co;
    if (cg < p.c0) { src = p.x0; cs = p.c0; co = cg; } else { src = p.x1; cs = p.c1; co = cg - p.c0; }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      if (poff[i] >= 0) ra[i] = *reinterpret_cast<const f32x4*>(src + (size_t)poff[i] * cs + co + c4 * 4);
      else ra[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (PRO == 1 || PRO == 2) {
      vsc = *reinterpret_cast<const f32x4*>(p.sc + (size_t)b * cin + cg + c4 * 4);
      vsh = *reinterpret_cast<const f32x4*>(p.sh + (size_t)b * cin + cg + c4 * 4);
    } else if (PRO == 3) {
      vsc = *reinterpret_cast<const f32x4*>(p.sc + cg + c4 * 4);
      vsh = *reinterpret_cast<const f32x4*>(p.sh + cg + c4 * 4);
    }
  };
  auto storeA = [&](int buf) {
    float* dst = sA + buf * (NPIX * BKP) + (tid / KQ) * BKP + c4 * 4;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      if (tid / KQ + i * PSTEP < NPIX) {
        f32x4 v = ra[i];
        if (PRO != 0 && poff[i] >= 0) {
          if (PRO == 3) {
            v = (v - pmu[i]) * prs[i] * vsc + vsh;
          } else {
            v = v * vsc + vsh;
            if (PRO == 1) { v[0] = silu_f(v[0]); v[1] = silu_f(v[1]); v[2] = silu_f(v[2]); v[3] = silu_f(v[3]); }
          }
        }
        *reinterpret_cast<f32x4*>(dst + i * PSTEP * BKP) = v;
      }
    }
  };
  auto loadW = [&](int chunk, int tap) {
    const size_t krow0 = (size_t)tap * (cin / 4) + (size_t)chunk * KQ;
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int u = tid + j * 256;
      const int row = u / BN, col = u % BN;
      rw[j] = *reinterpret_cast<const f32x4*>(static_cast<const float*>(p.w) + ((krow0 + row) * p.Npad + n0 + col) * 4);
    }
  };
  auto storeW = [&](int buf) {
    float* dst = sW + buf * (TOTW * 4);
#pragma unroll
    for (int j = 0; j < NW; ++j) *reinterpret_cast<f32x4*>(dst + (tid + j * 256) * 4) = rw[j];
  };

  // ---- MFMA operand addressing ----
  int hbase[FM];
#pragma unroll
  for (int fm = 0; fm < FM; ++fm) {
    const int pp = wm * WM + fm * 32 + (lane & 31);
    const int py = pp / TW, px = pp % TW;
    hbase[fm] = ((py * STRIDE) * TWIN + px * STRIDE) * BKP + 4 * (lane >> 5);
  }
  const int wbase = ((lane >> 5) * BN + wn * WN + (lane & 31)) * 4;

  f32x16 acc[FM][FN];
#pragma unroll
  for (int fm = 0; fm < FM; ++fm)
#pragma unroll
    for (int fn = 0; fn < FN; ++fn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[fm][fn][r] = 0.f;

  const int nchunk = cin / BK;

  loadA(0); loadW(0, 0);
  storeA(0); storeW(0);
  __syncthreads();

  // One barrier per (chunk, tap) tile.  The prefetch of the next tile is unconditional: past the end
  // it re-reads the last tile into the idle LDS buffer, which nobody reads (keeps the body branch-free).
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    const int nchunk1 = min(chunk + 1, nchunk - 1);
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int wpar = (chunk * TAPS + tap) & 1;
      if (tap + 1 < TAPS) loadW(chunk, tap + 1); else loadW(nchunk1, 0);
      if (tap == 0) loadA(nchunk1);

      const float* cA = sA + (chunk & 1) * (NPIX * BKP) + ((tap / KS) * TWIN + (tap % KS)) * BKP;
      const float* cW = sW + wpar * (TOTW * 4) + wbase;
#pragma unroll
      for (int k8 = 0; k8 < BK / 8; ++k8) {
        f32x4 a[FM], bb[FN];
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) a[fm] = *reinterpret_cast<const f32x4*>(cA + hbase[fm] + k8 * 8);
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) bb[fn] = *reinterpret_cast<const f32x4*>(cW + (k8 * 2 * BN + fn * 32) * 4);
#pragma unroll
        for (int fm = 0; fm < FM; ++fm)
#pragma unroll
          for (int fn = 0; fn < FN; ++fn) {
            acc[fm][fn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[fm][0], bb[fn][0], acc[fm][fn], 0, 0, 0);
            acc[fm][fn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[fm][1], bb[fn][1], acc[fm][fn], 0, 0, 0);
            acc[fm][fn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[fm][2], bb[fn][2], acc[fm][fn], 0, 0, 0);
            acc[fm][fn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[fm][3], bb[fn][3], acc[fm][fn], 0, 0, 0);
          }
      }

      storeW(wpar ^ 1);
      if (tap == 0) storeA((chunk + 1) & 1);
      __syncthreads();
    }
  }

  // ---- epilogue: bias / per-sample bias / residual / GeGLU, NHWC store (+ optional GroupNorm partial statistics) ----
  conv_epilogue<TH, TW, BN, FM, FN>(p, acc, b, oy0, ox0, n0, wm, wn, lane, tid, smem);
}

template <int KS, int STRIDE, bool UPS, int TH, int TW, int BN, int BK, int PRO>
static int launch_cfg(ConvP& p, hipStream_t stream) {
  constexpr int THIN = (TH - 1) * STRIDE + KS, TWIN = (TW - 1) * STRIDE + KS;
  constexpr size_t lds = (size_t)(2 * THIN * TWIN * (BK + 4) + 2 * BK * BN) * sizeof(float);
  static_assert(lds <= 160 * 1024, "LDS budget");
  p.tiles_x = cdiv(p.Wout, TW);
  p.tiles_y = cdiv(p.Hout, TH);
  p.nt = cdiv(p.Npad, BN);
  auto kern = conv_mfma_kernel<KS, STRIDE, UPS, TH, TW, BN, BK, PRO>;
  static std::atomic<uint64_t> attr_done{0};  // raise the dynamic-LDS cap once per instantiation and device
  if (int rc = set_max_lds_once(reinterpret_cast<const void*>(kern), (int)lds, attr_done)) return rc;
  const int grid = p.B * p.tiles_y * p.tiles_x * p.nt;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, p);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}

// tile choice shared by both arithmetic modes: 0 = 128 px x 128 ch, 1 = 128 px x 64 ch, 2 = 64 px x 64 ch
static void conv_out_dims(const pf_conv_args& a, int* hout, int* wout) {
  int h = a.hin, w = a.win;
  if (a.ups) { h *= 2; w *= 2; }
  if (a.stride == 2) { h = (h - 1) / 2 + 1; w = (w - 1) / 2 + 1; }
  *hout = h; *wout = w;
}
int conv_pick_tile(const pf_conv_args& a) {
  if (a.geglu) return 0;
  int hout, wout;
  conv_out_dims(a, &hout, &wout);
  const int npad = (a.n + 63) / 64 * 64;
  const int mt128 = a.ks == 1 ? a.batch * hout * cdiv(wout, 128) : a.batch * cdiv(hout, 8) * cdiv(wout, 16);
  if (a.ks == 3 && a.stride == 2) return 2;
  if (a.force_tile >= 1 && a.force_tile <= 3 && a.precision == PF_PREC_BF16X3 && a.stride == 1 && !a.ups && !a.ups_fold &&
      (a.ks == 3 || a.a_planes) && (a.force_tile != 1 || npad % 128 == 0)) return a.force_tile - 1;   // measurement aid (pf_conv_args.force_tile)
  // bf16x3 3x3: the wide tile + split-K beats twice as many narrow tiles; planes GEMMs (both operands direct-to-LDS): one 128x128 workgroup
  // per CU beats two 128x64 ones as soon as every CU gets one (measured at M = 16384, N = 256: K = 256 18.1 -> 16.1 us, K = 1024 37.5 -> 33.9 us)
  // bf16x3 3x3 with 64 output channels in all (the 128x128 level): a 16x16-pixel tile when that still gives every CU two rounds of
  // two workgroups - each wave then owns 128 pixels x 32 channels (four A fragments per weight fragment instead of two)
  if (a.precision == PF_PREC_BF16X3 && a.ks == 3 && a.stride == 1 && !a.ups && !a.ups_fold && npad == 64 && hout % 16 == 0 && wout % 16 == 0 &&
      a.batch * (hout / 16) * (wout / 16) >= 4 * num_cus() && !a.skip_w && a.c0 + a.c1 >= 128 && !a.no_t16) return 3;
  // (the same 16x16-pixel footprint for the 128-channel tile - 128 x 64 per wave, one workgroup per CU - measured worse at the 64x64
  // level: r64_128_128 55.3 -> 57 us, the fused-skip form 79 -> 82, only the K = 3456 conv gained 2.5 %)
  const bool wide_at_256 = a.precision == PF_PREC_BF16X3 && (a.ks == 3 || a.a_planes);
  const int cus = num_cus();   // (the thresholds were measured on 256 CUs; they are rounds of the chip, not literals)
  if (npad % 128 == 0 && mt128 * (npad / 128) >= (wide_at_256 ? cus : 2 * cus)) return 0;
  // bf16x3 3x3: the 128 px x 64 ch tile as soon as it gives every CU a workgroup - at B = 8 the 32x32 level has exactly 256 of them and they
  // beat 512 narrow tiles by 6-10 % (tools/sweep_conv.py, profiles/r05_sweep_conv_before.log); below that the 64 px tile fills more CUs
  const bool bf3x3 = a.precision == PF_PREC_BF16X3 && a.ks == 3 && a.stride == 1 && !a.ups;
  if (mt128 * (npad / 64) >= (bf3x3 ? cus : 2 * cus)) return 1;
  return 2;
}
void conv_tile_shape(const pf_conv_args& a, int tile, int* th, int* tw) {
  if (a.ks == 1) { *th = 1; *tw = tile == 2 ? 64 : 128; }
  else if (a.stride == 2) { *th = 4; *tw = 16; }
  else { *th = tile == 3 ? 16 : tile == 2 ? 4 : 8; *tw = 16; }
}
// split-K (bf16x3 3x3 only): layers whose tile grid cannot fill the chip (the 16x16 level at batch 16, most levels at
// small batch) run ksplit K-slices per tile and a reduce kernel that also applies the epilogue
static int ksplit_wanted(const pf_conv_args& a) {
  if (a.precision != PF_PREC_BF16X3 || a.ks != 3 || a.stride != 1 || a.geglu || a.ups_fold || conv_wino_eligible(a)) return 1;
  int hout, wout, th, tw;
  conv_out_dims(a, &hout, &wout);
  if ((hout * wout) % 64 != 0) return 1;
  conv_tile_shape(a, conv_pick_tile(a), &th, &tw);
  const int blocks = a.batch * cdiv(hout, th) * cdiv(wout, tw) * cdiv((a.n + 63) / 64 * 64, 64);
  const int nchunk = (a.c0 + a.c1) / 32;
  if (a.force_ksplit >= 1 && nchunk % a.force_ksplit == 0) return a.force_ksplit;   // measurement aid (pf_conv_args.force_ksplit)
  // Round 5, from a sweep of every layer shape at B = 1 / 8 / 16 (tools/sweep_conv.py, profiles/r05_sweep_conv_before.log): the split pays
  // only for DEEP K on FEW workgroups - its fp32 partial sums and the reduce launch cost ~8 us, which a K = 2304 loop (25 us unsplit on
  // any number of workgroups, 17 us with the intra-workgroup split) never earns back: B = 8, 16x16 level 22.9 -> 17.4 us, B = 1, 64x64
  // level 19.5 -> 11.5 us without it.  K >= 4608 on at most half the CUs, or K >= 3456 on at most a quarter: four slices (+10 ... +30 %).
  const int cus = num_cus();
  if (nchunk % 4 == 0 && ((nchunk >= 16 && blocks <= cus / 2) || (nchunk >= 12 && blocks <= cus / 4))) return 4;
  return 1;
}
size_t conv_splitk_ws_bytes(const pf_conv_args& a) {
  const int s = ksplit_wanted(a);
  if (s <= 1) return 0;
  int hout, wout;
  conv_out_dims(a, &hout, &wout);
  return (size_t)s * a.batch * hout * wout * a.n * sizeof(float);
}
int conv_ksplit(const pf_conv_args& a) {
  const size_t need = conv_splitk_ws_bytes(a);
  return (need && a.splitk_ws && a.splitk_ws_bytes >= need) ? ksplit_wanted(a) : 1;
}
int conv_stats_tiles(const pf_conv_args& a) {
  int hout, wout, th, tw;
  conv_out_dims(a, &hout, &wout);
  if (conv_wino_eligible(a)) return (hout / 16) * (wout / 16);   // the Winograd form: one statistics tile per 16x16-pixel workgroup
  if (conv_ksplit(a) > 1) return hout * wout / 64;   // the reduce kernel emits one statistics tile per 64 rows
  conv_tile_shape(a, conv_pick_tile(a), &th, &tw);
  if (a.ups_fold) return cdiv(a.hin, th) * cdiv(a.win, tw) * 4;   // tiles walk the source grid, one statistics tile per parity
  return cdiv(hout, th) * cdiv(wout, tw);
}

template <int KS, int STRIDE, bool UPS, int PRO>
static int dispatch_tile(ConvP& p, int tile, hipStream_t s) {
  if constexpr (KS == 1) {
    if (tile == 0) return launch_cfg<1, 1, false, 1, 128, 128, 16, PRO>(p, s);
    if (tile == 1) return launch_cfg<1, 1, false, 1, 128, 64, 16, PRO>(p, s);
    return launch_cfg<1, 1, false, 1, 64, 64, 32, PRO>(p, s);
  } else if constexpr (STRIDE == 2) {
    return launch_cfg<3, 2, false, 4, 16, 64, 16, PRO>(p, s);
  } else {
    if (tile == 0) return launch_cfg<3, 1, UPS, 8, 16, 128, 16, PRO>(p, s);
    if (tile == 1) return launch_cfg<3, 1, UPS, 8, 16, 64, 16, PRO>(p, s);
    return launch_cfg<3, 1, UPS, 4, 16, 64, 32, PRO>(p, s);
  }
}

double conv_flops(const pf_conv_args& a) {
  const int cin = a.c0 + a.c1;
  int hout = a.hin, wout = a.win;
  if (a.ks == 3) {
    if (a.ups) { hout *= 2; wout *= 2; }
    if (a.stride == 2) { hout = (hout - 1) / 2 + 1; wout = (wout - 1) / 2 + 1; }
  }
  const double skip = a.skip_w ? (double)(a.skip_c0 + a.skip_c1) : 0.0;   // fused 1x1 projection of a second tensor
  // folded upsampling conv: 2x2 taps per output pixel; Winograd F(2x2, 3x3): 16 products per 2x2 output pixels (work actually done)
  const double taps = (a.ups_fold || conv_wino_eligible(a)) ? 4.0 : (double)(a.ks * a.ks);
  return 2.0 * a.batch * hout * wout * (double)a.n * (cin * taps + skip);
}

int launch_conv(const pf_conv_args& a, hipStream_t stream) {
  PF_REQUIRE(a.ks == 1 || a.ks == 3, "conv: ks must be 1 or 3 (got %d)", a.ks);
  PF_REQUIRE(a.stride == 1 || a.stride == 2, "conv: stride must be 1 or 2");
  PF_REQUIRE(!(a.ks == 1 && (a.stride != 1 || a.ups)), "conv: 1x1 supports stride 1 without upsampling only");
  PF_REQUIRE(!(a.ups && a.stride != 1), "conv: upsample fold needs stride 1");
  PF_REQUIRE(a.c0 > 0 && a.c0 % 32 == 0 && a.c1 >= 0 && a.c1 % 32 == 0, "conv: channel counts must be multiples of 32 (c0=%d c1=%d)", a.c0, a.c1);
  PF_REQUIRE(a.x0 && (a.c1 == 0 || a.x1), "conv: null input");
  PF_REQUIRE(a.n > 0 && a.w && a.out, "conv: null weight/output");
  PF_REQUIRE(a.prologue >= 0 && a.prologue <= 3, "conv: bad prologue %d", a.prologue);
  PF_REQUIRE(a.prologue == 0 || (a.sc && a.sh), "conv: prologue needs sc/sh");
  PF_REQUIRE(!a.gn_stats0 || (a.precision == PF_PREC_BF16X3 && (a.prologue == 1 || a.prologue == 2) && a.gn_gamma && a.gn_beta && a.gn_groups > 0 &&
                              (a.c0 + a.c1) % a.gn_groups == 0 && a.c0 + a.c1 <= 1024 && a.gn_tiles0 > 0 && (a.c1 == 0 || (a.gn_stats1 && a.gn_tiles1 > 0))),
             "conv: fused GroupNorm finalize needs the bf16x3 path, prologue 1/2, gamma/beta, statistics of every source and <= 1024 channels");
  PF_REQUIRE(a.prologue != 3 || (a.mean && a.rstd && a.ks == 1), "conv: LayerNorm prologue needs mean/rstd and ks=1");
  PF_REQUIRE(!a.geglu || (a.n % 64 == 0 && !a.sbias && !a.res), "conv: geglu needs N %% 64 == 0 and no residual");
  PF_REQUIRE((a.ks == 3 && (a.prologue == 0 || a.prologue == 1)) || a.ks == 1, "conv: 3x3 supports prologue 0/1 only");
  PF_REQUIRE(!(a.ks == 3 && a.prologue == 1 && (a.ups || a.stride == 2)), "conv: GN prologue only on plain 3x3");
  PF_REQUIRE(!(a.ks == 3 && a.prologue == 0 && !a.ups && a.stride == 1), "conv: plain 3x3 without prologue is not instantiated");
  PF_REQUIRE(!(a.ks == 1 && a.prologue == 1), "conv: 1x1 with SiLU prologue is not instantiated");
  PF_REQUIRE(!(a.stats_out && a.geglu), "conv: statistics are not available with the GeGLU epilogue");
  PF_REQUIRE(!a.qkv_planes || (a.ks == 1 && a.n % 192 == 0 && a.win % 16 == 0 && !a.geglu && !a.res && !a.sbias && !a.stats_out),
             "conv: qkv planes need ks=1, N = 3*heads*64 and L %% 16 == 0");

  PF_REQUIRE(a.precision == PF_PREC_F32 || a.precision == PF_PREC_BF16X3, "conv: bad precision %d", a.precision);
  PF_REQUIRE(!a.ups_fold || (a.ups && a.precision == PF_PREC_BF16X3 && a.ks == 3 && a.stride == 1 && a.prologue == 0 && !a.res && !a.skip_w),
             "conv: ups_fold needs ups=1, bf16x3, ks=3, no prologue / residual");
  PF_REQUIRE(!a.skip_w || (a.precision == PF_PREC_BF16X3 && a.ks == 3 && a.stride == 1 && !a.ups && a.skip_x0 && a.skip_c0 > 0 &&
                           a.skip_c0 % 32 == 0 && a.skip_c1 % 32 == 0 && (a.skip_c1 == 0 || a.skip_x1) && !a.geglu),
             "conv: fused skip projection needs bf16x3, ks=3, stride 1, channel counts multiples of 32");
  PF_REQUIRE(!a.out_planes || (a.precision == PF_PREC_BF16X3 && a.ks == 1 && !a.stats_out && a.ld_out % 8 == 0 &&
                               (a.geglu ? a.n / 2 : a.n) % 8 == 0),
             "conv: out_planes needs bf16x3, ks=1, no statistics, ld_out and n multiples of 8");
  PF_REQUIRE(!a.a_planes || (a.precision == PF_PREC_BF16X3 && a.ks == 1 && a.prologue == 0 && a.c1 == 0),
             "conv: a_planes needs bf16x3, ks=1, no prologue, single source");
  if (a.a_planes) return launch_gemm_planes(a, stream);
  if (a.precision == PF_PREC_BF16X3) return launch_conv_bf3(a, stream);

  ConvP p;
  memset(&p, 0, sizeof p);
  p.x0 = a.x0; p.x1 = a.x1; p.c0 = a.c0; p.c1 = a.c1; p.x1_bmod = a.x1_bmod;
  p.B = a.batch; p.Hin = a.hin; p.Win = a.win;
  p.Hout = a.hin; p.Wout = a.win;
  if (a.ups) { p.Hout *= 2; p.Wout *= 2; }
  if (a.stride == 2) { p.Hout = (p.Hout - 1) / 2 + 1; p.Wout = (p.Wout - 1) / 2 + 1; }
  p.w = a.w; p.N = a.n; p.Npad = (a.n + 63) / 64 * 64;
  p.sc = a.sc; p.sh = a.sh; p.mean = a.mean; p.rstd = a.rstd;
  p.bias = a.bias; p.sbias = a.sbias; p.ld_sbias = a.ld_sbias; p.res = a.res; p.ld_res = a.ld_res;
  p.sb_rows = reinterpret_cast<const long long*>(a.sbias_rows); p.sb_nrows = a.sbias_nrows;
  p.geglu = a.geglu; p.out = a.out; p.ld_out = a.ld_out; p.stats = a.stats_out;
  p.ksplit = 1; p.partial = nullptr; p.qkv = a.qkv_planes; p.out_planes = a.out_planes;
  p.amax = static_cast<unsigned*>(a.absmax_slot);

  const int tile = conv_pick_tile(a);

  if (a.ks == 1) {
    switch (a.prologue) {
      case 0: return dispatch_tile<1, 1, false, 0>(p, tile, stream);
      case 2: return dispatch_tile<1, 1, false, 2>(p, tile, stream);
      default: return dispatch_tile<1, 1, false, 3>(p, tile, stream);
    }
  }
  if (a.stride == 2) return dispatch_tile<3, 2, false, 0>(p, tile, stream);
  if (a.ups) return dispatch_tile<3, 1, true, 0>(p, tile, stream);
  return dispatch_tile<3, 1, false, 1>(p, tile, stream);
}

}  // namespace pf
