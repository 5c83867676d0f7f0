// conv_bf16x3.hip - the same implicit-GEMM convolution / linear as conv_mfma.hip, computed on the
// bf16 matrix pipe with an error-compensated split ("bf16x3"):
//
//      a = a_hi + a_lo,  w = w_hi + w_lo   (hi = bf16_rne(x), lo = bf16_rne(x - hi))
//      a.w ~= a_hi.w_hi + a_hi.w_lo + a_lo.w_hi          (fp32 accumulate inside the MFMA)
//
// Every bf16 x bf16 product is exact in fp32, the dropped a_lo.w_lo term and the two representation
// errors are each <= 2^-18 relative, so a product carries ~1e-5 relative error (vs 2^-9 for plain
// bf16, which misses the 1e-3 parity bar by 20x - SURVEY.md 0) while running on
// v_mfma_f32_32x32x16_bf16 (2.5 PFLOP/s dense) instead of the 157 TFLOP/s fp32 MFMA: 3 bf16 MFMAs
// replace 8 fp32 MFMAs of twice the latency, a 5.3x higher matrix-pipe ceiling.
//
// Structure = conv_mfma.hip (halo tile in LDS, weights per (chunk, tap), fused GN+SiLU / LayerNorm
// prologue and bias / residual / GeGLU epilogue, XCD-aware block remap).  Differences:
//   * activations are split into hi/lo bf16 planes when the tile is staged (after the prologue
//     transform, 3.5 VALU ops per element, once per element per block);
//   * weights are pre-split on the host and packed [tap][K/8][plane][Npad][8] so that a lane's eight
//     consecutive-k B operands are one 16-byte LDS read per plane;
//   * BK = 32; the 3x3 halo image is single-buffered (it changes every 9 taps; its global loads and its
//     normalise/activate/split arithmetic are spread over the taps before), weight tiles stream
//     global->LDS directly through a 3-slot ring, fragments are software-pipelined with hand-counted
//     LDS waits (see the loop); 1x1 mode is register-staged and double-buffers both operands;
//   * optional phases / variants around the same loop: the ResBlock's 1x1 skip projection as an
//     extra K range (SKIP), parity-folded UpSample convs as 2x2 taps on the source grid (KS == 2),
//     split-K across workgroups (ksplit) and across two wave groups of one workgroup (KG == 2).
#include <vector>

#include "conv_common.h"

namespace pf {

#ifdef PF_TRACE
__device__ unsigned long long g_trace[8192];
#define TR() do { if (trace_on && tslot < 2040) { g_trace[tbase + tslot++] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define TR() do {} while (0)
#endif

typedef x3_t x3x4 __attribute__((ext_vector_type(4)));

// bf16 elements between two halo rows of one plane.  A pixel is 40 elements (80 B: consecutive pixels of a row are
// conflict-free for ds_read_b128); a fragment's 32 pixels span TWO tile rows, and with the natural row pitch TWIN*40 the second
// row's 16-byte slots land on the banks of the first (2-way conflict on every A read: SQ_LDS_BANK_CONFLICT was 35 % of the LDS
// cycles).  Padding the row pitch to a multiple of 256 B makes the two rows tile the 64 banks exactly.  Only for the 128-wide
// tile: on the 64-wide one the padding costs the third workgroup per CU (55,296 B vs 53,376 B of 160 KiB / 3), and that tile -
// K = 576 at the 128x128 level, half of a workgroup's life waiting for HBM - loses more to the missing wave than to the conflicts
// (r128_64_64: 78 -> 70 us per launch).
template <int KS, int STRIDE, int TWIN, int BN>
constexpr int conv_bf3_row_pitch() {
  return (KS != 1 && STRIDE == 1 && BN >= 128) ? (TWIN * 40 + 127) / 128 * 128 : TWIN * 40;
}

// Weight-ring depth.  3x3: three slots, two tiles in flight across every barrier - enough where a tap carries 12-24 MFMAs per wave
// (1.4-2.3 k cycles per tap against ~1 us from issue to landing of a direct-to-LDS load).  The K-split tile of the 16x16 level (KG == 2:
// 6 MFMAs per wave and tap) runs its taps in ~0.6 k cycles when nothing waits, so two tiles of lead are ~1.2 k cycles and every tap
// waited ~0.4 k for its weights: it gets a deeper ring (its single workgroup per CU has the LDS for it).  A ring that does not divide
// the taps addresses its slots through a register instead of instruction immediates (see DYN in the kernel).
#ifndef PF_KG2_RING
#define PF_KG2_RING 5
#endif
template <int KS, int BN, int KG>
constexpr bool conv_bf3_pingpong() { return KS == 3 && BN == 128 && KG == 2; }
template <int KS, int KG, int BN = 64>
constexpr int conv_bf3_wring() { return KS == 3 ? (KG == 2 && !conv_bf3_pingpong<KS, BN, KG>() ? PF_KG2_RING : 3) : 2; }

// LDS bytes of one wave group: the main loop's halo image + weight ring, or the fused 1x1 phase's double buffers
template <int KS, int STRIDE, int TH, int TW, int BN, bool SKIP, int KG = 1>
constexpr size_t conv_bf3_group_lds() {
  constexpr int THIN = (TH - 1) * STRIDE + KS, TWIN = (TW - 1) * STRIDE + KS;
  constexpr int NABUF = (KS == 1) ? 2 : 1, WRING = conv_bf3_wring<KS, KG, BN>();
  constexpr size_t main_b = (size_t)(2 * NABUF * THIN * conv_bf3_row_pitch<KS, STRIDE, TWIN, BN>() + WRING * 8 * BN * 8) * 2;
  constexpr int NB1 = BN >= 128 ? 2 : 1;   // buffers of the fused 1x1 phase (see conv_bf3_kernel: the 64-wide tile keeps its third workgroup per CU)
  constexpr size_t skip_b = SKIP ? (size_t)(2 * NB1 * TH * TW * 40 + NB1 * 8 * BN * 8) * 2 : 0;
  return main_b > skip_b ? main_b : skip_b;
}

// KG = 2: intra-workgroup K split.  When a layer has no more tiles than the chip has CUs (B = 16: the 32x32 and 16x16 levels),
// one 4-wave workgroup per CU leaves every SIMD with a single wave and nothing to hide latencies behind (used for the 4x16x64
// tile, i.e. the 16x16 level at B = 16: +2.5 % end to end).  The workgroup then
// carries TWO 4-wave groups, each running the complete pipeline below on its own half of the K range and its own LDS
// region (identical instruction streams, so the workgroup barriers line up); group 1 hands its accumulators to group 0
// through LDS at the end and only group 0 runs the epilogue.  No global reduce pass, no extra HBM traffic.
template <int KS, int STRIDE, bool UPS, int TH, int TW, int BN, int PRO, int NWM, bool SKIP, int KG = 1>
__global__ __launch_bounds__(KG * NWM * 128, 2) void conv_bf3_kernel(ConvP p) {
  constexpr int NT = NWM * 128;            // threads of a wave group: NWM x 2 waves
  constexpr int BK = 32;
  constexpr int BM = TH * TW;
  constexpr int THIN = (TH - 1) * STRIDE + KS;
  constexpr int TWIN = (TW - 1) * STRIDE + KS;
  constexpr int NPIX = THIN * TWIN;
  constexpr int PITCH = BK + 8;            // bf16 elements per pixel row (80 B: conflict-free b128 for consecutive pixels)
  constexpr int KQ = BK / 4;               // float4 per pixel per chunk (global side)
  constexpr int TOTA = NPIX * KQ;
  constexpr int NA = (TOTA + NT - 1) / NT;
  constexpr int PSTEP = NT / KQ;           // pixels between a thread's successive float4
  constexpr int TOTW = 8 * BN;             // 16-byte units per W tile (2 planes x 4 k8 x BN)
  constexpr int NW = TOTW / NT;
  constexpr int TAPS = KS * KS;
  constexpr int NABUF = (KS == 1) ? 2 : 1;
  constexpr int WRING = conv_bf3_wring<KS, KG, BN>();   // W tile buffers (3x3: direct-to-LDS ring of 3, deeper for KG == 2; the 4-tap folded conv: ring of 2)
  constexpr int WM = BM / NWM, WN = BN / 2;
  constexpr int FM = WM / 32, FN = WN / 32;
  constexpr int PAD = (KS == 3) ? 1 : 0;   // KS == 2 (parity-folded upsampling conv): the pad depends on the parity, see iy0
  constexpr int RP = conv_bf3_row_pitch<KS, STRIDE, TWIN, BN>();   // bf16 elements per halo row (padded, see conv_bf3_row_pitch)
  constexpr int ABUF = THIN * RP;               // bf16 elements per halo image
  constexpr int APLANE = NABUF * ABUF;          // bf16 elements per A plane
  static_assert(TOTW % NT == 0 && FM >= 1 && FN >= 1, "bad tile");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  const int kg = KG == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x / NT);   // wave group (K half), wave-uniform
  unsigned char* smem_raw = smem_all + kg * conv_bf3_group_lds<KS, STRIDE, TH, TW, BN, SKIP, KG>();
  x3_t* sAh = reinterpret_cast<x3_t*>(smem_raw);
  x3_t* sAl = sAh + APLANE;
  x3_t* sW = sAl + APLANE;               // [WRING bufs][4 k8][2 planes][BN][8]

  const int tid = KG == 1 ? threadIdx.x : threadIdx.x % NT, lane = tid & 63;   // group-local
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: LDS-DMA destinations (M0) and tile roles stay in SGPRs
#ifdef PF_TRACE
  const bool trace_on = (threadIdx.x == 0) && (blockIdx.x == 0 || blockIdx.x == 301 || blockIdx.x == gridDim.x - 1);
  const int tbase = blockIdx.x == 0 ? 0 : (blockIdx.x == 301 ? 2048 : 4096);
  int tslot = 0;
  TR();
#endif
  const int wm = wave >> 1, wn = wave & 1;

  const int nwg = gridDim.x;
  int lid;
  {
    const int orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  int q = 0;                           // KS == 2: output parity (py, px) = (q >> 1, q & 1); the four parities of a source
  if constexpr (KS == 2) { q = lid & 3; lid >>= 2; p.fold_py = q >> 1; p.fold_px = q & 1; }   // tile run together (shared halo in L2)
  int t = fdiv(lid, p.d_ks);
  const int sidx = lid - t * p.ksplit;   // K-slice (fastest varying: the slices of a tile run together and share its halo in L2)
  lid = t;
  int mt = fdiv(lid, p.d_nt);
  const int nti = lid - mt * p.nt;
  t = fdiv(mt, p.d_tx);
  const int tx = mt - t * p.tiles_x; mt = t;
  const int b = fdiv(mt, p.d_ty);
  const int ty = mt - b * p.tiles_y;
  const int n0 = nti * BN;
  const int oy0 = ty * TH, ox0 = tx * TW;
  conv_shared_x1(p, b);
  // pixel the unconditional loads of padding / out-of-tile pieces read (their values are zeroed by the transform): pixel 0 of the first
  // sample - or, with a rebased second source, of the sample the rebase points back to, so that the address stays inside both tensors
  const int ppad = p.x1_bmod > 0 ? (b - b % p.x1_bmod) * p.Hin * p.Win : 0;
  // KS == 2: output row 2y+py reads source rows {y-1, y} (py = 0) or {y, y+1} (py = 1); same for columns
  const int iy0 = KS == 2 ? oy0 - 1 + (q >> 1) : oy0 * STRIDE - PAD, ix0 = KS == 2 ? ox0 - 1 + (q & 1) : ox0 * STRIDE - PAD;
  const int Hlog = UPS ? 2 * p.Hin : p.Hin, Wlog = UPS ? 2 * p.Win : p.Win;
  const int cin = p.c0 + p.c1;
  const int K8 = cin / 8;
  const int nsl = p.ksplit * KG, sl = sidx * KG + kg;       // K slices: across workgroups (split-K) x across wave groups
  // this wave group's K-slice [cbeg, cbeg + nchunk) in BK-channel chunks (nsl is 1 or a power of two in every dispatch: shifts)
  const bool pow2 = (nsl & (nsl - 1)) == 0;
  const int nsh = 31 - __builtin_clz(nsl);
  const int cbeg = pow2 ? ((cin / BK) * sl) >> nsh : (cin / BK) * sl / nsl;
  const int nchunk = (pow2 ? ((cin / BK) * (sl + 1)) >> nsh : (cin / BK) * (sl + 1) / nsl) - cbeg;

  const int c4 = tid % KQ;
  int poff[NA];
  float pmu[NA], prs[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int pix = tid / KQ + i * PSTEP;
    poff[i] = -1; pmu[i] = 0.f; prs[i] = 0.f;
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

  f32x4 ra[NA], vsc, vsh;
  u32x4 rw[NW];
  vsc = f32x4{1.f, 1.f, 1.f, 1.f}; vsh = f32x4{0.f, 0.f, 0.f, 0.f};

  // With the GroupNorm finalize folded into this launch (p.gn_s0) the sample's scale / shift rows do not exist yet when loadA(0)
  // runs: its two loads then fetch gamma / beta instead (same count of vector loads, values discarded) - nothing reads the rows,
  // or pulls their cache lines into this CU's L1, before gn_fused_prologue has written them.
  bool sc_ready = !((PRO == 1 || PRO == 2) && p.gn_s0);
  auto loadA = [&](int chunk) {
    const int cg = (cbeg + chunk) * BK;
    const float* src; int cs, co;
    if (cg < p.c0) { src = p.x0; cs = p.c0; co = cg; } else { src = p.x1; cs = p.c1; co = cg - p.c0; }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      // unconditional (padding / out-of-tile pieces read pixel 0 and are zeroed by the transform): the number of
      // vector loads in flight is then a compile-time constant, which the counted vmcnt waits of the 3x3 loop rely on
      if (KS != 1 || poff[i] >= 0) ra[i] = *reinterpret_cast<const f32x4*>(src + co + (unsigned)(max(poff[i], ppad) * cs + c4 * 4));
      else ra[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (PRO == 1 || PRO == 2) {
      const float* scp = sc_ready ? p.sc + (size_t)b * cin : p.gn_gamma;
      const float* shp = sc_ready ? p.sh + (size_t)b * cin : p.gn_beta;
      vsc = *reinterpret_cast<const f32x4*>(scp + cg + c4 * 4);
      vsh = *reinterpret_cast<const f32x4*>(shp + cg + c4 * 4);
    } else if (PRO == 3) {
      vsc = *reinterpret_cast<const f32x4*>(p.sc + cg + c4 * 4);
      vsh = *reinterpret_cast<const f32x4*>(p.sh + cg + c4 * 4);
    }
  };
  // staged halo values after the prologue transform, split into bf16 hi / lo quads (kept in registers between the
  // transform, which is scheduled in the shadow of a mid-chunk tap's MFMAs, and the LDS write at the chunk boundary)
  x3x4 qh[NA], ql[NA];
  auto transformPiece = [&](int i) {
    {
      f32x4 v = ra[i];
      if (poff[i] < 0) v = f32x4{0.f, 0.f, 0.f, 0.f};
      else if (PRO != 0) {
        if (PRO == 3) {
          v = (v - pmu[i]) * prs[i] * vsc + vsh;
        } else {
          v = v * vsc + vsh;
          if (PRO == 1) { v[0] = silu_f(v[0]); v[1] = silu_f(v[1]); v[2] = silu_f(v[2]); v[3] = silu_f(v[3]); }
        }
      }
      qh[i] = __builtin_convertvector(v, x3x4);
      const f32x4 hf = __builtin_convertvector(qh[i], f32x4);
      ql[i] = __builtin_convertvector(v - hf, x3x4);
    }
  };
  auto transformA = [&]() {
#pragma unroll
    for (int i = 0; i < NA; ++i) transformPiece(i);
  };
  auto writeA = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int pix = tid / KQ + i * PSTEP;
      if (pix < NPIX) {
        const int o = buf * ABUF + (pix / TWIN) * RP + (pix % TWIN) * PITCH + c4 * 4;
        *reinterpret_cast<x3x4*>(sAh + o) = qh[i];
        *reinterpret_cast<x3x4*>(sAl + o) = ql[i];
      }
    }
  };
  auto storeA = [&](int buf) { transformA(); writeA(buf); };
  auto loadW = [&](int chunk, int tap) {
    const size_t toff = ((size_t)tap * K8 + (size_t)(cbeg + chunk) * 4) * ((size_t)2 * p.Npad * 8);
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int u = tid + j * NT;
      const int k8l = u / (2 * BN), plane = (u / BN) & 1, n = u % BN;
      rw[j] = *reinterpret_cast<const u32x4*>(static_cast<const x3_t*>(p.w) + ((size_t)(k8l * 2 + plane) * p.Npad + n0 + n) * 8 + toff);
    }
  };
  auto storeW = [&](int buf) {
    x3_t* dst = sW + buf * (TOTW * 8);
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      *reinterpret_cast<u32x4*>(dst + (tid + j * NT) * 8) = rw[j];
    }
  };

  // direct global->LDS copy of one weight tile (no registers, no transform): LDS image = tile order, lane-linear
  // per-thread source pointers of the NW pieces inside a tile are fixed; a tile only adds a wave-uniform offset
  static_assert(NT % (2 * BN) == 0, "pieces of a thread must be whole k8 rows apart");
  // (buffer form, dma16 of conv_common.h: SGPR resource at this column tile's first weight, one offset VGPR per lane, the tile as a
  // wave-uniform SGPR offset - next to MFMAs it issues several times faster than the global form with its address VGPR pair)
  const int wrow_b = 2 * p.Npad * 8 * 2;        // BYTES per k8 row pair (hi|lo planes); a layer's packing is a few MB: int offsets
  const __amdgpu_buffer_rsrc_t rsW = dma_resource(static_cast<const x3_t*>(p.w) + (size_t)n0 * 8);
  const int vw0 = (((tid / (2 * BN)) * 2 + ((tid / BN) & 1)) * p.Npad + tid % BN) * 16;
  auto gldsW = [&](int chunk, int tap, int buf) {
    const int toff = ((q * TAPS + tap) * K8 + (cbeg + chunk) * 4) * wrow_b;   // q != 0 only for the folded conv
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      x3_t* l = sW + buf * (TOTW * 8) + (wave * 64 + j * NT) * 8;   // wave-uniform base; the hardware adds lane*16 B
      dma16(rsW, vw0, toff + j * (NT / (2 * BN)) * wrow_b, l);
    }
  };

  int hbase[FM];
#pragma unroll
  for (int fm = 0; fm < FM; ++fm) {
    const int pp = wm * WM + fm * 32 + (lane & 31);
    const int py = pp / TW, px = pp % TW;
    hbase[fm] = (py * STRIDE) * RP + (px * STRIDE) * PITCH + 8 * (lane >> 5);
  }
  const int wbase = ((lane >> 5) * 2 * BN + wn * WN + (lane & 31)) * 8;

  f32x16 acc[FM][FN];
#pragma unroll
  for (int fm = 0; fm < FM; ++fm)
#pragma unroll
    for (int fn = 0; fn < FN; ++fn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[fm][fn][r] = 0.f;


  // GroupNorm finalize folded into this launch (pf_conv_args.gn_*): runs while the first tile's loads are in flight - the halo /
  // A-tile region of the LDS serves as its scratch and is only written by storeA afterwards - then reads chunk 0's scale/shift
  // (loadA(0) could not: the rows did not exist yet, see sc_ready).
  auto gnFused = [&]() {
    if constexpr (PRO == 1 || PRO == 2) {
      if (p.gn_s0) {
        gn_fused_prologue<KG * NT>(p, b, threadIdx.x, p.Hin * p.Win, reinterpret_cast<double*>(smem_all));
        const int cg = cbeg * BK;
        vsc = *reinterpret_cast<const f32x4*>(p.sc + (size_t)b * cin + cg + c4 * 4);
        vsh = *reinterpret_cast<const f32x4*>(p.sh + (size_t)b * cin + cg + c4 * 4);
        sc_ready = true;
      }
    }
  };
  if constexpr (KS == 1) {
    // 1x1 / linear: A and W tiles both change every chunk; register-staged, double-buffered, one barrier per chunk
    loadA(0); loadW(0, 0);
    gnFused();
    storeA(0); storeW(0);
    __syncthreads();
    for (int chunk = 0; chunk < nchunk; ++chunk) {
      const int nchunk1 = min(chunk + 1, nchunk - 1);
      loadW(nchunk1, 0);
      loadA(nchunk1);
      const int aoff = (chunk & 1) * ABUF;
      const x3_t* cW = sW + (chunk & 1) * (TOTW * 8) + wbase;
#pragma unroll
      for (int s = 0; s < BK / 16; ++s) {
        x3x8 ah[FM], al[FM], bh[FN], bl[FN];
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
          ah[fm] = *reinterpret_cast<const x3x8*>(sAh + aoff + hbase[fm] + s * 16);
          al[fm] = *reinterpret_cast<const x3x8*>(sAl + aoff + hbase[fm] + s * 16);
        }
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) {
          bh[fn] = *reinterpret_cast<const x3x8*>(cW + ((4 * s) * BN + fn * 32) * 8);
          bl[fn] = *reinterpret_cast<const x3x8*>(cW + ((4 * s + 1) * BN + fn * 32) * 8);
        }
#pragma unroll
        for (int fm = 0; fm < FM; ++fm)
#pragma unroll
          for (int fn = 0; fn < FN; ++fn) acc[fm][fn] = x3_mfma_32x32x16(al[fm], bh[fn], acc[fm][fn], 0, 0, 0);
#pragma unroll
        for (int fm = 0; fm < FM; ++fm)
#pragma unroll
          for (int fn = 0; fn < FN; ++fn) acc[fm][fn] = x3_mfma_32x32x16(ah[fm], bl[fn], acc[fm][fn], 0, 0, 0);
#pragma unroll
        for (int fm = 0; fm < FM; ++fm)
#pragma unroll
          for (int fn = 0; fn < FN; ++fn) acc[fm][fn] = x3_mfma_32x32x16(ah[fm], bh[fn], acc[fm][fn], 0, 0, 0);
      }
      storeW((chunk + 1) & 1);
      storeA((chunk + 1) & 1);
      __syncthreads();
    }
  } else {
    // 3x3: the halo image changes every 9 taps (register-staged, transformed, single buffer); weight tiles stream
    // through a 3-slot LDS ring by direct global->LDS loads.  A 16-deep K step is three MFMA groups on one fragment set
    //     X = a_lo x w_hi      Z = a_hi x w_hi      Y = a_hi x w_lo
    // and every fragment is re-loaded for the NEXT step as soon as its last reader has issued (a_lo after X, w_hi after
    // Z, a_hi / w_lo after Y), so each ds_read has one or two MFMA groups of cover and no second register set is needed:
    //
    //   tap it:   X0 | ld a_lo(1)     Z0 | ld w_hi(1)     Y0 | ld a_hi(1), w_lo(1)     X1
    //             wait: tile it+1 landed, own LDS reads done; s_barrier
    //             -> every read of ring slot it%3 has completed: refill it with tile it+3 (two tiles stay in flight)
    //             ld a_lo(0')      Z1 | ld w_hi(0')     Y1 | ld a_hi(0'), w_lo(0')          (0' = step 0 of tap it+1)
    //
    // One barrier per tap.  At a chunk boundary the halo buffer is rewritten right after the barrier of tap 8 (every wave
    // holds its last A fragments by then) and a second barrier publishes it before the next chunk's A fragments are read.
    constexpr int NS = BK / 16;
    static_assert(NS == 2 && WRING >= 2 && WRING <= TAPS, "pipeline is written for two K steps per tap and a ring no deeper than a chunk's taps");
    // DYN: the ring does not divide the taps, so a tile's slot is not a function of its tap alone: the current / next slot's byte
    // offsets live in SGPRs (rs_cur / rs_nxt, advanced once per tap) and are added to the weight base register (one VALU op per tap)
    constexpr bool DYN = TAPS % WRING != 0;
    constexpr int NAL = NA + (PRO != 0 ? 2 : 0);   // vector loads issued by loadA (all unconditional)
    // PP: the two wave groups of a KG = 2 workgroup run half a tap apart (see the PP loop below); its first segment issues tile 2
    constexpr bool PP = conv_bf3_pingpong<KS, BN, KG>();
#pragma unroll
    for (int d = 0; d < (PP ? 2 : WRING); ++d) gldsW(0, d, d);
    TR();
    loadA(0);
    TR();
    gnFused();
    storeA(0);
    TR();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    x3x8 ah[FM], al[FM], bh[FN], bl[FN];
    // LDS byte addresses: one base VGPR per A fragment row set and one for the weights; everything else is an immediate
    unsigned abase[FM];
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) abase[fm] = (unsigned)(size_t)(__attribute__((address_space(3))) x3_t*)(sAh + hbase[fm]);
    const unsigned wb = (unsigned)(size_t)(__attribute__((address_space(3))) x3_t*)(sW + wbase);
    constexpr int ALO = APLANE * 2;          // byte offset of the lo plane
    constexpr int WSLOT = TOTW * 16;         // bytes per ring slot
    static_assert(ALO + (2 * RP + 2 * PITCH) * 2 + 64 < 65536 && (DYN ? 1 : WRING) * WSLOT < 65536, "LDS immediates must fit 16 bits");
    // AOFF = halo pixel offset of a tap (elements), S = K step inside the tap, SLOT = ring slot
    // single-fragment reads (I = fragment index); the *_ALL forms read a whole operand set
#define LD_AL1(I, AOFF, S) al[I] = lds_read128<ALO + ((AOFF) + (S) * 16) * 2>(abase[I])
#define LD_AH1(I, AOFF, S) ah[I] = lds_read128<((AOFF) + (S) * 16) * 2>(abase[I])
    // SLOT: `cur` / `nxt` - the ring slot of this tap's / the next tap's tile (immediate offset, or - DYN - a base register of its own)
#define LD_BH1(I, SLOT, S) bh[I] = lds_read128<(DYN ? 0 : slot_##SLOT * WSLOT) + ((4 * (S)) * BN + (I) * 32) * 16>(DYN ? wb_##SLOT : wb)
#define LD_BL1(I, SLOT, S) bl[I] = lds_read128<(DYN ? 0 : slot_##SLOT * WSLOT) + ((4 * (S) + 1) * BN + (I) * 32) * 16>(DYN ? wb_##SLOT : wb)
#define LD_AL(AOFF, S) static_for<0, FM>([&](auto i) { LD_AL1(i.value, AOFF, S); })
#define LD_AH(AOFF, S) static_for<0, FM>([&](auto i) { LD_AH1(i.value, AOFF, S); })
#define LD_BH(SLOT, S) static_for<0, FN>([&](auto i) { LD_BH1(i.value, SLOT, S); })
#define LD_BL(SLOT, S) static_for<0, FN>([&](auto i) { LD_BL1(i.value, SLOT, S); })
#define SB() __builtin_amdgcn_sched_barrier(0)
#define FENCE() asm volatile("" ::: "memory")
    // An in-order wave hides at most ~5 single-issue instructions behind one 32-cycle MFMA: everything that is not an MFMA
    // (fragment reads, the next halo's loads and its normalise/activate/split arithmetic, the weight-ring refill) is therefore
    // cut into small pieces and dealt out BETWEEN the individual MFMAs instead of sitting in blocks between MFMA groups
    // (measured on the block form: the halo arithmetic alone, although placed "in the shadow" of a group, cost 9-14 %).
    // An MFMA group walks its G = FM*FN accumulators; `f(fm, fn)` runs right after the MFMA on acc[fm][fn] has issued.
    constexpr int G = FM * FN;
    auto GX = [&](auto&& f) {   // a_lo x w_hi, fm outer: al[fm] is free after its last fn
      static_for<0, FM>([&](auto fm) { static_for<0, FN>([&](auto fn) {
        acc[fm.value][fn.value] = x3_mfma_32x32x16(al[fm.value], bh[fn.value], acc[fm.value][fn.value], 0, 0, 0);
        SB(); f(fm, fn); SB(); }); });
    };
    auto GZ = [&](auto&& f) {   // a_hi x w_hi, fn outer: bh[fn] is free after its last fm
      static_for<0, FN>([&](auto fn) { static_for<0, FM>([&](auto fm) {
        acc[fm.value][fn.value] = x3_mfma_32x32x16(ah[fm.value], bh[fn.value], acc[fm.value][fn.value], 0, 0, 0);
        SB(); f(fm, fn); SB(); }); });
    };
    auto GY = [&](auto&& f) {   // a_hi x w_lo, fm outer: ah[fm] is free after its last fn, the w_lo set after the last MFMA
      static_for<0, FM>([&](auto fm) { static_for<0, FN>([&](auto fn) {
        acc[fm.value][fn.value] = x3_mfma_32x32x16(ah[fm.value], bl[fn.value], acc[fm.value][fn.value], 0, 0, 0);
        SB(); f(fm, fn); SB(); }); });
    };
    // ---- the non-MFMA work of a chunk, in pieces
    auto loadApiece = [&](int chunk, int i) {   // piece i < NA: one halo float4; i == NA: the chunk's GroupNorm scale/shift
      const int cg = (cbeg + chunk) * BK;
      const float* src; int cs, co;
      if (cg < p.c0) { src = p.x0; cs = p.c0; co = cg; } else { src = p.x1; cs = p.c1; co = cg - p.c0; }
      // Issued as inline asm, hidden from hipcc's wait-count pass: with direct-to-LDS loads in flight the compiler waits
      // vmcnt(0) before the first use of an ordinary load's result, which drained the whole weight ring once per chunk.
      // The destination registers are read by transformSub only after the hand-placed vmcnt wait below.
      // (the buffer form that pays for the weight tiles was tried here too: no gain at 0.7 halo loads per tap - kept global)
      if (i < NA) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ra[i]) : "v"(src + co + (unsigned)(max(poff[i], ppad) * cs + c4 * 4)));
      else if (PRO == 1 || PRO == 2) {
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(vsc) : "v"(p.sc + (size_t)b * cin + cg + c4 * 4));
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(vsh) : "v"(p.sh + (size_t)b * cin + cg + c4 * 4));
      }
    };
    // transform of piece i in four sub-steps: elements {0,1}, {2,3} in place, then the hi plane, then the lo plane
    auto transformSub = [&](int i, int sub) {
      if (sub < 2) {
#pragma unroll
        for (int e = 2 * sub; e < 2 * sub + 2; ++e) {
          float v = ra[i][e];
          if (poff[i] < 0) v = 0.f;
          else if (PRO != 0) { v = v * vsc[e] + vsh[e]; if (PRO == 1) v = silu_f(v); }
          ra[i][e] = v;
        }
      } else if (sub == 2) {
        qh[i] = __builtin_convertvector(ra[i], x3x4);
      } else {
        ql[i] = __builtin_convertvector(ra[i] - __builtin_convertvector(qh[i], f32x4), x3x4);
      }
    };
    auto gldsWpiece = [&](int chunk, int tap, int buf, int j) {
      const int toff = ((q * TAPS + tap) * K8 + (cbeg + chunk) * 4) * wrow_b;
      x3_t* l = sW + buf * (TOTW * 8) + (wave * 64 + j * NT) * 8;
      dma16(rsW, vw0, toff + j * (NT / (2 * BN)) * wrow_b, l);
    };
    if constexpr (PP) {
      // ---- ping-pong form (KG == 2, 128-wide tile): a tap is a LOAD segment - every fragment of the tap (both K steps) into registers,
      // the refill of the ring slot freed one tap ago, the next chunk's halo loads - and a COMPUTE segment - the tap's 6G MFMAs back to
      // back, the halo arithmetic and (tap 8) the halo image rewrite as fillers - each closed by the workgroup barrier.  Group 1 enters
      // the loop one barrier late, so on every SIMD one wave computes while its partner loads: the matrix pipe sees an MFMA stream
      // that no fragment wait interrupts (tools/micro/tap_pingpong.hip: 100 % of the pipe on this mix, against 88-92 % dealt out between
      // the MFMAs of a single stream).  A group's barriers are those of the lockstep form (all reads of a slot before its refill, all
      // pieces of a tile landed before its first read); pairing them with the other group's barriers of the other kind changes nothing.
      static_assert(WRING == 3 && TAPS == 9 && NS == 2, "ping-pong loop: ring of 3 over 9 taps");
      x3x8 ah2[NS][FM], al2[NS][FM], bh2[NS][FN], bl2[NS][FN];
      int woff[NA];   // element offsets of this thread's halo pieces (fixed: the rewrite in tap 8 must not need VALU work)
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int pix = min(tid / KQ + i * PSTEP, NPIX - 1);
        woff[i] = (pix / TWIN) * RP + (pix % TWIN) * PITCH + c4 * 4;
      }
      if (kg) __builtin_amdgcn_s_barrier();
      for (int chunk = 0; chunk < nchunk; ++chunk) {
        const bool has_next = chunk + 1 < nchunk;
        const int chunkn = min(chunk + 1, nchunk - 1);
        static_for<0, TAPS>([&](auto tapc) {
          constexpr int tap = decltype(tapc)::value;
          constexpr int aoff = (tap / KS) * RP + (tap % KS) * PITCH;
          constexpr int slot_cur = tap % WRING, slot_ref = (tap + 2) % WRING;
          constexpr int T0 = 2, TSPAN = TAPS - 1 - T0;
          int c2 = chunk + (tap + 2) / TAPS, t2 = (tap + 2) % TAPS;     // the tile issued by this tap: two taps ahead
          if (c2 >= nchunk) { c2 = nchunk - 1; t2 = TAPS - 1; }
          // ---- LOAD: fragment reads first (their latency covers everything else), then the copies, then this tap's halo work.
          // VALU work belongs HERE: a computing wave queues its MFMAs and runs ahead to the closing barrier - that is what lets the
          // partner queue its own MFMAs before the pipe runs dry - and any VALU instruction between or after the MFMAs would hold
          // the wave back until the queue has drained (in-order VALU): measured +190 cycles per segment with the fillers there.
          TR();
          SB();
          static_for<0, NS>([&](auto sc_) {
            constexpr int st = decltype(sc_)::value;
            static_for<0, FM>([&](auto i) { al2[st][i.value] = lds_read128<ALO + (aoff + st * 16) * 2>(abase[i.value]); });
            static_for<0, FN>([&](auto i) { bh2[st][i.value] = lds_read128<slot_cur * WSLOT + ((4 * st) * BN + i.value * 32) * 16>(wb); });
            static_for<0, FM>([&](auto i) { ah2[st][i.value] = lds_read128<(aoff + st * 16) * 2>(abase[i.value]); });
            static_for<0, FN>([&](auto i) { bl2[st][i.value] = lds_read128<slot_cur * WSLOT + ((4 * st + 1) * BN + i.value * 32) * 16>(wb); });
          });
          static_for<0, NW>([&](auto jc) { gldsWpiece(c2, t2, slot_ref, decltype(jc)::value); });
          if constexpr (tap == 0) {
            static_for<0, NA + 1>([&](auto ic) { if (ic.value < NA || PRO != 0) loadApiece(chunkn, ic.value); });
          }
          if constexpr (tap == T0) {   // first use of the halo registers: their loads (tap 0) are older than the two tiles issued since
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NW) : "memory");
            SB();
          }
          if (has_next) {
            static_for<0, NA>([&](auto ic) {
              constexpr int i = decltype(ic)::value;
              if constexpr (tap == T0 + (i * TSPAN) / NA) { static_for<0, 4>([&](auto sc_) { transformSub(i, decltype(sc_)::value); }); }
            });
          }
          SB();
          TR();
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          TR();
          __builtin_amdgcn_s_barrier();
          FENCE();
          SB();
          TR();
          // ---- COMPUTE: MFMAs only (tap 8: plus the LDS stores of the next halo image - every wave holds its tap-8 fragments)
          static_for<0, NS>([&](auto sc_) {
            constexpr int st = decltype(sc_)::value;
            static_for<0, FM>([&](auto fm) { static_for<0, FN>([&](auto fn) {
              acc[fm.value][fn.value] = x3_mfma_32x32x16(al2[st][fm.value], bh2[st][fn.value], acc[fm.value][fn.value], 0, 0, 0);
              SB(); }); });
            static_for<0, FN>([&](auto fn) { static_for<0, FM>([&](auto fm) {
              acc[fm.value][fn.value] = x3_mfma_32x32x16(ah2[st][fm.value], bh2[st][fn.value], acc[fm.value][fn.value], 0, 0, 0);
              SB(); }); });
            static_for<0, FM>([&](auto fm) { static_for<0, FN>([&](auto fn) {
              acc[fm.value][fn.value] = x3_mfma_32x32x16(ah2[st][fm.value], bl2[st][fn.value], acc[fm.value][fn.value], 0, 0, 0);
              SB(); }); });
          });
          if constexpr (tap == TAPS - 1) {
            if (has_next) {
#pragma unroll
              for (int i = 0; i < NA; ++i) {
                if (tid / KQ + i * PSTEP < NPIX) {
                  *reinterpret_cast<x3x4*>(sAh + woff[i]) = qh[i];
                  *reinterpret_cast<x3x4*>(sAl + woff[i]) = ql[i];
                }
              }
            }
          }
          SB();
          TR();
          // the next tap's tile has landed (newer: the tile issued by this tap and, until tap 2 has consumed them, the halo loads);
          // lgkmcnt(0): the halo image rewrite of tap 8
          asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NW + (tap < T0 ? NAL : 0)) : "memory");
          TR();
          __builtin_amdgcn_s_barrier();
          FENCE();
          SB();
        });
      }
      if (!kg) __builtin_amdgcn_s_barrier();
    } else {
    int rs_cur = 0;                     // DYN: ring slot of the current tap's tile (wave-uniform)
    {
      constexpr int slot_cur = 0;
      const unsigned wb_cur = wb;
      (void)slot_cur; (void)wb_cur;
      LD_AL(0, 0); LD_BH(cur, 0); LD_AH(0, 0); LD_BL(cur, 0);
    }
    // The loop body is branch-free around the MFMAs (tail iterations prefetch clamped tiles and read fragments that are
    // never used).  LDS reads complete in issue order and are issued in the fixed order a_lo, w_hi, a_hi, w_lo (each set right
    // after its last reader), so "at most FM+FN reads outstanding" is the wait before every group in the steady state.
    for (int chunk = 0; chunk < nchunk; ++chunk) {
      const bool has_next = chunk + 1 < nchunk;
      const int chunkn = min(chunk + 1, nchunk - 1);
      static_for<0, TAPS>([&](auto tapc) {
        constexpr int tap = decltype(tapc)::value;
        constexpr int aoff = (tap / KS) * RP + (tap % KS) * PITCH;
        constexpr int aoff1 = ((tap + 1) / KS) * RP + ((tap + 1) % KS) * PITCH;
        constexpr int slot_cur = DYN ? 0 : tap % WRING, slot_nxt = DYN ? 0 : (tap + 1) % WRING;   // TAPS % WRING == 0: a tile's ring slot is tap % WRING
        const int rs_nxt = rs_cur + 1 == WRING ? 0 : rs_cur + 1;
        const int slot = DYN ? rs_cur : slot_cur;                        // ring slot refilled by this tap (runtime in DYN mode)
        const unsigned wb_cur = wb + (unsigned)rs_cur * WSLOT, wb_nxt = wb + (unsigned)rs_nxt * WSLOT;
        (void)wb_cur; (void)wb_nxt; (void)slot_nxt;
        // which halo piece this tap transforms (spread over the middle taps: 2..7 of 9, 1..2 of 4), -1 = none
        constexpr int T0 = TAPS >= 9 ? 2 : 1, TSPAN = TAPS - 1 - T0;
        // filler after MFMA number k of the tap (k = 0..6G-1: groups X0 Z0 Y0 X1 | barrier | Z1 Y1)
        int c2 = chunk + (tap + WRING) / TAPS, t2 = (tap + WRING) % TAPS;     // the tile that refills this tap's ring slot
        if (c2 >= nchunk) { c2 = nchunk - 1; t2 = TAPS - 1; }
        auto filler = [&](auto kc) {
          constexpr int k = decltype(kc)::value;
          if constexpr (tap == 0 && k < 3 * G) {       // next chunk's halo loads over X0 Z0 Y0 (all before this tap's barrier)
            static_for<0, NA + 1>([&](auto ic) {
              constexpr int i = decltype(ic)::value;
              if constexpr (k == (i * 3 * G) / (NA + 1)) { if (i < NA || PRO != 0) loadApiece(chunkn, i); }
            });
          }
          if constexpr (k >= 3 * G && k < 4 * G) {     // X1: this tap's share of the halo arithmetic
            if constexpr (tap == T0 && k == 3 * G) {
              // first use of the halo registers in this chunk: the loads issued in tap 0 are older than the WRING-1 weight tiles
              // issued since (tap T0's own refill comes after this group).  Unconditional - also in the last chunk, whose halo
              // loads are never consumed - so that every path to a transform step passes the wait (tools/lint_asm.py checks it).
              // (one refill per tap since then: after the barriers of taps 0 .. T0-1)
              asm volatile("s_waitcnt vmcnt(%0)" ::"n"(T0 * NW) : "memory");
              SB();
            }
            if (has_next) {
              static_for<0, NA>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (tap == T0 + (i * TSPAN) / NA) {
                  static_for<0, 4>([&](auto sc_) {
                    constexpr int sub = decltype(sc_)::value;
                    if constexpr (k - 3 * G == (sub * G) / 4) transformSub(i, sub);
                  });
                }
              });
            }
          }
          if constexpr (k >= 4 * G && k < 5 * G) {     // Z1 (after the barrier): refill the ring slot this tap has finished with
            static_for<0, NW>([&](auto jc) {
              constexpr int j = decltype(jc)::value;
              if constexpr (k - 4 * G == (j * G) / NW) gldsWpiece(c2, t2, slot, j);
            });
          }
        };
        TR();
        SB();
        // outstanding on entry: a_lo, w_hi, a_hi, w_lo of (tap, step 0) in that order - after a chunk boundary a_lo, a_hi
        lgkm_wait<(tap == 0) ? FM : FM + FN>(); SB();
        GX([&](auto fm, auto fn) {
          if constexpr (fn.value == FN - 1) LD_AL1(fm.value, aoff, 1);
          filler(std::integral_constant<int, 0 * G + fm.value * FN + fn.value>{});
        });
        lgkm_wait<(tap == 0) ? FM : FM + FN>(); SB();
        GZ([&](auto fm, auto fn) {
          if constexpr (fm.value == FM - 1) LD_BH1(fn.value, cur, 1);
          filler(std::integral_constant<int, 1 * G + fn.value * FM + fm.value>{});
        });
        lgkm_wait<FM + FN>(); SB();
        GY([&](auto fm, auto fn) {
          if constexpr (fn.value == FN - 1) LD_AH1(fm.value, aoff, 1);
          if constexpr (fm.value == FM - 1 && fn.value == FN - 1) LD_BL(cur, 1);
          filler(std::integral_constant<int, 2 * G + fm.value * FN + fn.value>{});
        });
        lgkm_wait<FM + FN>(); SB();
        GX([&](auto fm, auto fn) { filler(std::integral_constant<int, 3 * G + fm.value * FN + fn.value>{}); });
        SB();
        TR();
        // the next tile must be complete.  This thread's vector loads issued after it are the WRING-2 newer weight tiles and -
        // while tile it+1 was issued before this chunk's tap 0 - the NAL halo loads.  lgkmcnt(0): all reads of this slot done.
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((WRING - 2) * NW + (tap < WRING - 1 ? NAL : 0)) : "memory");
        TR();
        __builtin_amdgcn_s_barrier();
        FENCE();
        SB();   // nothing of Z1 may be hoisted above the wait: its a_hi / w_lo fragments are only now guaranteed to have landed
        TR();
        if constexpr (tap == TAPS - 1) {
          if (has_next) writeA(0);
          SB();
          GZ([&](auto fm, auto fn) {
            if constexpr (fm.value == FM - 1) LD_BH1(fn.value, nxt, 0);
            filler(std::integral_constant<int, 4 * G + fn.value * FM + fm.value>{});
          });
          GY([&](auto fm, auto fn) {
            if constexpr (fm.value == FM - 1 && fn.value == FN - 1) LD_BL(nxt, 0);
            filler(std::integral_constant<int, 5 * G + fm.value * FN + fn.value>{});
          });
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          FENCE();
          LD_AL(0, 0); LD_AH(0, 0); SB();
        } else {
          // the A reads of the next tap do not depend on the barrier but keep the fixed issue order a_lo, w_hi, a_hi, w_lo
          GZ([&](auto fm, auto fn) {
            if constexpr (fn.value == 0) LD_AL1(fm.value, aoff1, 0);
            if constexpr (fm.value == FM - 1) LD_BH1(fn.value, nxt, 0);
            filler(std::integral_constant<int, 4 * G + fn.value * FM + fm.value>{});
          });
          GY([&](auto fm, auto fn) {
            if constexpr (fn.value == FN - 1) LD_AH1(fm.value, aoff1, 0);
            if constexpr (fm.value == FM - 1 && fn.value == FN - 1) LD_BL(nxt, 0);
            filler(std::integral_constant<int, 5 * G + fm.value * FN + fn.value>{});
          });
        }
        rs_cur = rs_nxt;
      });
    }
    }
#undef LD_AL1
#undef LD_AH1
#undef LD_BH1
#undef LD_BL1
#undef LD_AL
#undef LD_AH
#undef LD_BH
#undef LD_BL
#undef FENCE
#undef SB
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // drain the clamped tail prefetches before the workgroup retires
    // ... and before the registers of the hidden halo loads (last issued in the final chunk's tap 0, never consumed) can be reused
#pragma unroll
    for (int i = 0; i < NA; ++i) asm volatile("" : "+v"(ra[i]));
    asm volatile("" : "+v"(vsc), "+v"(vsh));
    // Same for the fragment registers: the last tap's reads for a tap that never runs are still in flight when the loop exits,
    // and to the compiler those values are dead - it used their registers for the accumulator copies of the loop-exit edge, i.e.
    // BEFORE the wait above, and a late LDS return then overwrote an accumulator (seen as run-to-run differences once a change
    // elsewhere altered the register assignment).  A use after the wait keeps the registers allocated until the data has landed.
    if constexpr (!PP) {   // (the ping-pong form leaves no fragment read in flight)
    #pragma unroll
    for (int i = 0; i < FM; ++i) asm volatile("" ::"v"(al[i]), "v"(ah[i]));
    #pragma unroll
    for (int i = 0; i < FN; ++i) asm volatile("" ::"v"(bh[i]), "v"(bl[i]));
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  if constexpr (SKIP) {
    // ---- fused 1x1 projection of a second tensor into the same accumulators (the ResBlock skip conv): one extra K range
    // over concat(sx0, sx1), no prologue, weights [K/8][plane][Npad][8].  Register-staged and double-buffered like the
    // KS == 1 path, on the LDS the 3x3 loop has finished with; done by K-slice 0 only when the conv is split.  The 64-wide tile
    // single-buffers (one more barrier per 32-channel chunk): double buffers would be the tile's largest LDS user and cost it
    // the third workgroup per CU.
    if (sidx == 0) {
      constexpr int NB1 = BN >= 128 ? 2 : 1;
      constexpr int NA1 = BM * KQ / NT;          // float4 pieces per thread of a 32-channel slab of the tile's own pixels
      static_assert(BM * KQ % NT == 0, "tile pixels must divide over the block");
      x3_t* a1h = reinterpret_cast<x3_t*>(smem_raw);
      x3_t* a1l = a1h + NB1 * BM * PITCH;
      x3_t* w1 = a1l + NB1 * BM * PITCH;        // [NB1 bufs][4 k8][2 planes][BN][8]
      int goff[NA1];
#pragma unroll
      for (int i = 0; i < NA1; ++i) {
        const int pix = tid / KQ + i * PSTEP;
        const int oy = oy0 + pix / TW, ox = ox0 + pix % TW;
        goff[i] = (oy < p.Hout && ox < p.Wout) ? (b * p.Hout + oy) * p.Wout + ox : -1;
      }
      f32x4 r1[NA1];
      u32x4 rw1[NW];
      auto load1 = [&](int chunk) {
        const int cg = chunk * BK;
        const float* src; int cs, co;
        if (cg < p.sc0) { src = p.sx0; cs = p.sc0; co = cg; } else { src = p.sx1; cs = p.sc1; co = cg - p.sc0; }
#pragma unroll
        for (int i = 0; i < NA1; ++i) {
          if (goff[i] >= 0) r1[i] = *reinterpret_cast<const f32x4*>(src + co + (unsigned)(goff[i] * cs + c4 * 4));
          else r1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < NW; ++j) {
          const int u = tid + j * NT;
          const int k8l = u / (2 * BN), plane = (u / BN) & 1, n = u % BN;
          rw1[j] = *reinterpret_cast<const u32x4*>(static_cast<const x3_t*>(p.sw) + ((size_t)((chunk * 4 + k8l) * 2 + plane) * p.Npad + n0 + n) * 8);
        }
      };
      auto store1 = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA1; ++i) {
          const x3x4 hi = __builtin_convertvector(r1[i], x3x4);
          const x3x4 lo = __builtin_convertvector(r1[i] - __builtin_convertvector(hi, f32x4), x3x4);
          const int o = (buf * BM + tid / KQ + i * PSTEP) * PITCH + c4 * 4;
          *reinterpret_cast<x3x4*>(a1h + o) = hi;
          *reinterpret_cast<x3x4*>(a1l + o) = lo;
        }
#pragma unroll
        for (int j = 0; j < NW; ++j) *reinterpret_cast<u32x4*>(w1 + buf * (TOTW * 8) + (tid + j * NT) * 8) = rw1[j];
      };
      int arow[FM];
#pragma unroll
      for (int fm = 0; fm < FM; ++fm) arow[fm] = (wm * WM + fm * 32 + (lane & 31)) * PITCH + 8 * (lane >> 5);
      const int nchx = (p.sc0 + p.sc1) / BK;
      const int ch0 = nchx * kg / KG, nch1 = nchx * (kg + 1) / KG;      // this wave group's share (equal shares: dispatch)
      load1(ch0);
      __syncthreads();                       // every wave is out of the 3x3 loop: its LDS images are dead
      store1(0);
      __syncthreads();
      for (int chunk = ch0; chunk < nch1; ++chunk) {
        if (chunk + 1 < nch1) load1(chunk + 1);
        const int cur = NB1 == 2 ? ((chunk - ch0) & 1) : 0;
        const int ao = cur * BM * PITCH;
        const x3_t* cW = w1 + cur * (TOTW * 8) + wbase;
#pragma unroll
        for (int s2 = 0; s2 < BK / 16; ++s2) {
          x3x8 ah[FM], al[FM], bh[FN], bl[FN];
#pragma unroll
          for (int fm = 0; fm < FM; ++fm) {
            ah[fm] = *reinterpret_cast<const x3x8*>(a1h + ao + arow[fm] + s2 * 16);
            al[fm] = *reinterpret_cast<const x3x8*>(a1l + ao + arow[fm] + s2 * 16);
          }
#pragma unroll
          for (int fn = 0; fn < FN; ++fn) {
            bh[fn] = *reinterpret_cast<const x3x8*>(cW + ((4 * s2) * BN + fn * 32) * 8);
            bl[fn] = *reinterpret_cast<const x3x8*>(cW + ((4 * s2 + 1) * BN + fn * 32) * 8);
          }
#pragma unroll
          for (int fm = 0; fm < FM; ++fm)
#pragma unroll
            for (int fn = 0; fn < FN; ++fn) acc[fm][fn] = x3_mfma_32x32x16(al[fm], bh[fn], acc[fm][fn], 0, 0, 0);
#pragma unroll
          for (int fm = 0; fm < FM; ++fm)
#pragma unroll
            for (int fn = 0; fn < FN; ++fn) acc[fm][fn] = x3_mfma_32x32x16(ah[fm], bl[fn], acc[fm][fn], 0, 0, 0);
#pragma unroll
          for (int fm = 0; fm < FM; ++fm)
#pragma unroll
            for (int fn = 0; fn < FN; ++fn) acc[fm][fn] = x3_mfma_32x32x16(ah[fm], bh[fn], acc[fm][fn], 0, 0, 0);
        }
        if constexpr (NB1 == 1) __syncthreads();   // every wave has read the single buffer
        if (chunk + 1 < nch1) store1(NB1 == 2 ? ((chunk + 1 - ch0) & 1) : 0);
        __syncthreads();
      }
    }
  }

  TR();
  if (p.partial) p.partial += (size_t)sidx * ((size_t)p.B * p.Hout * p.Wout) * p.N;
  if constexpr (KG == 2) {   // group 1 -> LDS -> group 0 (register-order dump: conflict-free, no index math)
    float* xch = reinterpret_cast<float*>(smem_all);
    __syncthreads();            // both groups are done with their LDS regions
    if (kg == 1) {
#pragma unroll
      for (int fm = 0; fm < FM; ++fm)
#pragma unroll
        for (int fn = 0; fn < FN; ++fn)
#pragma unroll
          for (int r = 0; r < 16; ++r) xch[((fm * FN + fn) * 16 + r) * NT + tid] = acc[fm][fn][r];
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
      for (int fm = 0; fm < FM; ++fm)
#pragma unroll
        for (int fn = 0; fn < FN; ++fn)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[fm][fn][r] += xch[((fm * FN + fn) * 16 + r) * NT + tid];
    }
    __syncthreads();            // the epilogue reuses this LDS for the statistics
  }
  x3_unscale(acc);
  conv_epilogue<TH, TW, BN, FM, FN, NWM>(p, acc, b, oy0, ox0, n0, wm, wn, lane, tid, reinterpret_cast<float*>(smem_all), kg == 0);
  TR();
}

// out = sum_s partial[s] + bias + sbias[b] + res over a 64-row slab per workgroup; also emits the slab's per-channel
// (sum, sumsq) as one statistics tile.  Deterministic (fixed summation order).
__global__ __launch_bounds__(1024) void splitk_reduce_kernel(const float* __restrict__ part, int S, int M, int N, int hw,
                                                            const float* __restrict__ bias, const float* __restrict__ bias2,
                                                            const float* __restrict__ sbias, int ld_sb,
                                                            const long long* __restrict__ sb_rows, int sb_nrows,
                                                            const float* __restrict__ res, int ld_res, float* __restrict__ out,
                                                            int ld_out, float* __restrict__ stats, unsigned* amax) {
  __shared__ float red[16][64][2];
  const int tid = threadIdx.x, c = tid & 63, rg = tid >> 6;   // 16 row groups x 64 columns
  const int row0 = blockIdx.x * 64;
  const int b = row0 / hw, tile = (row0 % hw) / 64, ntiles = hw / 64;
  {
    const int n = blockIdx.y * 64 + c;
    float s1 = 0.f, s2 = 0.f;
    unsigned am = 0;
    if (n < N) {
      long long sr = b;
      if (sb_rows) { sr = sb_rows[b]; sr = sr < 0 ? 0 : (sr >= sb_nrows ? sb_nrows - 1 : sr); }
      const float cb = (bias ? bias[n] : 0.f) + (bias2 ? bias2[n] : 0.f) + (sbias ? sbias[(size_t)sr * ld_sb + n] : 0.f);
      for (int i = 0; i < 4; ++i) {
        const size_t m = row0 + rg + 16 * i;
        float v = cb;
        for (int s = 0; s < S; ++s) v += part[((size_t)s * M + m) * N + n];
        if (res) v += res[m * ld_res + n];
        out[m * ld_out + n] = v;
        amax_acc(am, v);
        s1 += v; s2 += v * v;
      }
    }
    amax_flush(amax, am);
    if (stats) {
      red[rg][c][0] = s1; red[rg][c][1] = s2;
      __syncthreads();
      if (rg == 0 && n < N) {
        float* dst = stats + (((size_t)b * ntiles + tile) * N + n) * 2;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) { a0 += red[j][c][0]; a1 += red[j][c][1]; }
        dst[0] = a0; dst[1] = a1;
      }
      __syncthreads();
    }
  }
}

template <int KS, int STRIDE, bool UPS, int TH, int TW, int BN, int PRO, int NWM = 2, bool SKIP = false, int KG = 1>
static int launch3_cfg(ConvP& p, hipStream_t stream) {
  constexpr int FMFN = (TH * TW / NWM / 32) * (BN / 2 / 32);
  constexpr size_t lds_groups = KG * conv_bf3_group_lds<KS, STRIDE, TH, TW, BN, SKIP, KG>();
  constexpr size_t lds_xch = KG == 2 ? (size_t)FMFN * 16 * NWM * 128 * 4 : 0;   // accumulator hand-over between the wave groups
  constexpr size_t lds = lds_groups > lds_xch ? lds_groups : lds_xch;
  static_assert(lds <= 160 * 1024, "LDS budget");
  p.tiles_x = cdiv(p.Wout, TW);
  p.tiles_y = cdiv(p.Hout, TH);
  p.nt = cdiv(p.Npad, BN);
  conv_fill_divs(p);
  auto kern = conv_bf3_kernel<KS, STRIDE, UPS, TH, TW, BN, PRO, NWM, SKIP, KG>;
  static std::atomic<uint64_t> attr_done{0};
  if (int rc = set_max_lds_once(reinterpret_cast<const void*>(kern), (int)lds, attr_done)) return rc;
  const int grid = p.B * p.tiles_y * p.tiles_x * p.nt * p.ksplit * (KS == 2 ? 4 : 1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(KG * NWM * 128), lds, stream, p);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}

template <int KS, int STRIDE, bool UPS, int PRO>
static int dispatch_tile3(ConvP& p, int tile, hipStream_t s) {
  if constexpr (KS == 1) {
    if (tile == 0) return launch3_cfg<1, 1, false, 1, 128, 128, PRO>(p, s);
    if (tile == 1) return launch3_cfg<1, 1, false, 1, 128, 64, PRO>(p, s);
    return launch3_cfg<1, 1, false, 1, 64, 64, PRO>(p, s);
  } else if constexpr (KS == 2) {
    if (tile == 0) return launch3_cfg<2, 1, false, 8, 16, 128, PRO>(p, s);
    if (tile == 1) return launch3_cfg<2, 1, false, 8, 16, 64, PRO>(p, s);
    return launch3_cfg<2, 1, false, 4, 16, 64, PRO>(p, s);
  } else if constexpr (STRIDE == 2) {
    return launch3_cfg<3, 2, false, 4, 16, 64, PRO>(p, s);
  } else {
    if constexpr (!UPS && PRO == 1) {
      // no more tiles than CUs (and an even number of K chunks): two wave groups per workgroup split K (see the kernel)
      const int blocks = p.B * cdiv(p.Hout, tile == 3 ? 16 : tile == 2 ? 4 : 8) * cdiv(p.Wout, 16) * cdiv(p.Npad, tile == 0 ? 128 : 64);
      // (the 128x128 tile split the same way measured neutral - its two wave groups run in lockstep behind the shared barrier - DESIGN.md 3)
      const int cus = num_cus();
      const bool kg2 = tile == 2 && p.ksplit == 1 && blocks <= cus && ((p.c0 + p.c1) / 32) % 2 == 0 && (!p.sw || ((p.sc0 + p.sc1) / 32) % 2 == 0);
      if (kg2) return p.sw ? launch3_cfg<3, 1, false, 4, 16, 64, 1, 2, true, 2>(p, s) : launch3_cfg<3, 1, false, 4, 16, 64, 1, 2, false, 2>(p, s);
      // ... but run as two groups half a tap apart (conv_bf3_pingpong) it gains: one group's fragment reads / copies / halo arithmetic
      // hide behind the other's MFMAs (B = 16: the 32x32 level; B = 8: the 64x64 level)
      // (from K = 2304 up: at K = 576 ... 1728 the two-group form measured 3-6 % behind the four-wave one - r32_128_256, r64_128_128 at B = 8)
      if (p.pp && tile == 0 && p.ksplit == 1 && blocks <= cus && (p.c0 + p.c1) >= 256 && ((p.c0 + p.c1) / 32) % 2 == 0 && (!p.sw || ((p.sc0 + p.sc1) / 32) % 2 == 0))
        return p.sw ? launch3_cfg<3, 1, false, 8, 16, 128, 1, 2, true, 2>(p, s) : launch3_cfg<3, 1, false, 8, 16, 128, 1, 2, false, 2>(p, s);
      if (tile == 3) return p.sw ? launch3_cfg<3, 1, false, 16, 16, 64, 1, 2, true>(p, s) : launch3_cfg<3, 1, false, 16, 16, 64, 1, 2, false>(p, s);
      if (p.sw) {   // fused skip projection: only the ResBlock second-conv configurations are instantiated
        if (tile == 0) return launch3_cfg<3, 1, false, 8, 16, 128, 1, 2, true>(p, s);
        if (tile == 1) return launch3_cfg<3, 1, false, 8, 16, 64, 1, 2, true>(p, s);
        return launch3_cfg<3, 1, false, 4, 16, 64, 1, 2, true>(p, s);
      }
    }
    if (tile == 0) return launch3_cfg<3, 1, UPS, 8, 16, 128, PRO>(p, s);
    if (tile == 1) return launch3_cfg<3, 1, UPS, 8, 16, 64, PRO>(p, s);
    return launch3_cfg<3, 1, UPS, 4, 16, 64, PRO>(p, s);
  }
}

// same argument validation as launch_conv (done by the caller); w points to the bf16x3 packing
int launch_conv_bf3(const pf_conv_args& a, hipStream_t stream) {
  if (conv_wino_eligible(a)) return launch_conv_wino(a, stream);
  ConvP p;
  memset(&p, 0, sizeof p);
  p.x0 = a.x0; p.x1 = a.x1; p.c0 = a.c0; p.c1 = a.c1;
  p.B = a.batch; p.Hin = a.hin; p.Win = a.win;
  p.Hout = a.hin; p.Wout = a.win;
  if (a.ups) { p.Hout *= 2; p.Wout *= 2; }
  if (a.stride == 2) { p.Hout = (p.Hout - 1) / 2 + 1; p.Wout = (p.Wout - 1) / 2 + 1; }
  p.w = a.w; p.N = a.n; p.Npad = (a.n + 63) / 64 * 64;
  p.sc = a.sc; p.sh = a.sh; p.mean = a.mean; p.rstd = a.rstd;
  p.bias = a.bias; p.sbias = a.sbias; p.ld_sbias = a.ld_sbias; p.res = a.res; p.ld_res = a.ld_res;
  p.sb_rows = reinterpret_cast<const long long*>(a.sbias_rows); p.sb_nrows = a.sbias_nrows;
  p.geglu = a.geglu; p.out = a.out; p.ld_out = a.ld_out; p.stats = a.stats_out;
  p.ksplit = conv_ksplit(a);
  p.pp = !a.no_pp;
  p.partial = p.ksplit > 1 ? static_cast<float*>(a.splitk_ws) : nullptr;
  p.qkv = a.qkv_planes; p.out_planes = a.out_planes;
  if (a.gn_stats0 && (a.prologue == 1 || a.prologue == 2)) {
    p.gn_s0 = a.gn_stats0; p.gn_t0 = a.gn_tiles0; p.gn_s1 = a.gn_stats1; p.gn_t1 = a.gn_tiles1;
    p.gn_gamma = a.gn_gamma; p.gn_beta = a.gn_beta; p.gn_eps = a.gn_eps; p.gn_groups = a.gn_groups;
  }
  p.sx0 = a.skip_x0; p.sc0 = a.skip_c0; p.sx1 = a.skip_x1; p.sc1 = a.skip_c1; p.sw = a.skip_w; p.bias2 = a.skip_w ? a.skip_bias : nullptr;
  p.x1_bmod = a.x1_bmod;
  p.amax = static_cast<unsigned*>(a.absmax_slot);
  const int tile = conv_pick_tile(a);
  if (a.ks == 1) {
    switch (a.prologue) {
      case 0: return dispatch_tile3<1, 1, false, 0>(p, tile, stream);
      case 2: return dispatch_tile3<1, 1, false, 2>(p, tile, stream);
      default: return dispatch_tile3<1, 1, false, 3>(p, tile, stream);
    }
  }
  if (a.ups_fold) {   // tiles walk the source grid; every workgroup stores one parity of its pixels
    p.Hout = a.hin; p.Wout = a.win; p.fold = 1;
    return dispatch_tile3<2, 1, false, 0>(p, tile, stream);
  }
  if (a.stride == 2) return dispatch_tile3<3, 2, false, 0>(p, tile, stream);
  int rc = a.ups ? dispatch_tile3<3, 1, true, 0>(p, tile, stream) : dispatch_tile3<3, 1, false, 1>(p, tile, stream);
  if (rc != PF_OK || p.ksplit == 1) return rc;
  const int M = p.B * p.Hout * p.Wout;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(M / 64, cdiv(p.N, 64)), dim3(1024), 0, stream, static_cast<const float*>(a.splitk_ws), p.ksplit, M, p.N,
                     p.Hout * p.Wout, p.bias, p.bias2, p.sbias, p.ld_sbias, p.sb_rows, p.sb_nrows, p.res, p.ld_res, p.out, p.ld_out, p.stats, p.amax);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}

#ifdef PF_TRACE
extern "C" int pf_debug_trace_read(unsigned long long* dst, int n) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_trace), (size_t)n * 8) == hipSuccess ? 0 : -1;
}
extern "C" int pf_debug_trace_clear() {
  static unsigned long long z[8192];
  return hipMemcpyToSymbol(HIP_SYMBOL(g_trace), z, sizeof z) == hipSuccess ? 0 : -1;
}
#endif

// host: fp32 torch weight [N][K][taps] -> bf16x3 packing [tap][K/8][plane][Npad][8] at column col(n)
static inline unsigned short f2bf_rne(float f) {
  unsigned int u; memcpy(&u, &f, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
static inline float bf2f(unsigned short h) { unsigned int u = (unsigned int)h << 16; float f; memcpy(&f, &u, 4); return f; }

bool pack_gemm_bf3(void* dst_, const float* src, int n_src, int K, int taps, int Npad, int n_off, const int* colmap) {
  unsigned short* dst = (unsigned short*)dst_;
  bool fits = true;
  const int K8 = K / 8;
  for (int n = 0; n < n_src; ++n) {
    const int col = colmap ? colmap[n] : n_off + n;
    for (int k = 0; k < K; ++k)
      for (int t = 0; t < taps; ++t) {
        const float v = src[((size_t)n * K + k) * taps + t];
#ifdef PF_X3_F16
        const float vs = fminf(fmaxf(v * PF_X3_WS, -65504.f), 65504.f);
        fits = fits && vs == v * PF_X3_WS;
        const _Float16 fh = (_Float16)vs, fl = (_Float16)(vs - (float)fh);
        unsigned short hi, lo;
        memcpy(&hi, &fh, 2); memcpy(&lo, &fl, 2);
#else
        const unsigned short hi = f2bf_rne(v);
        const unsigned short lo = f2bf_rne(v - bf2f(hi));
#endif
        const size_t base = ((size_t)t * K8 + k / 8) * 2;
        dst[((base + 0) * Npad + col) * 8 + (k & 7)] = hi;
        dst[((base + 1) * Npad + col) * 8 + (k & 7)] = lo;
      }
  }
  return fits;
}


// UpSample = nearest x2 then conv3x3 (ref:unet.py:236-238).  Output pixel (2y+py, 2x+px) only ever sees a 2x2 block of SOURCE
// pixels: rows {y-1, y} with weights {w[0], w[1]+w[2]} for py = 0, rows {y, y+1} with {w[0]+w[1], w[2]} for py = 1 (columns
// alike), so the layer is four 2x2 convolutions on the source grid - 16 taps in total instead of 4 x 9.
bool pack_upfold_bf3(void* dst, const float* src, int N, int K, int Npad) {
  std::vector<float> f((size_t)N * K * 16);
  static const int lo[2][2] = {{0, 1}, {0, 2}}, hi[2][2] = {{0, 2}, {1, 2}};   // [parity][tap] -> kernel index range [lo, hi]
  for (size_t nk = 0; nk < (size_t)N * K; ++nk) {
    const float* w = src + nk * 9;
    for (int py = 0; py < 2; ++py)
      for (int px = 0; px < 2; ++px)
        for (int dy = 0; dy < 2; ++dy)
          for (int dx = 0; dx < 2; ++dx) {
            double acc = 0.0;
            for (int ky = lo[py][dy]; ky <= hi[py][dy]; ++ky)
              for (int kx = lo[px][dx]; kx <= hi[px][dx]; ++kx) acc += w[ky * 3 + kx];
            f[nk * 16 + (py * 2 + px) * 4 + dy * 2 + dx] = (float)acc;
          }
  }
  return pack_gemm_bf3(dst, f.data(), N, K, 16, Npad, 0, nullptr);
}

}  // namespace pf
