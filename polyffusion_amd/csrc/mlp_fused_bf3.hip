// mlp_fused_bf3.hip - the transformer block's feed-forward half as ONE launch (bf16x3 split MFMA):
//
//      out = x + ff2( GeGLU( ff1( LayerNorm3(x) ) ) )            ref:stable_diffusion/model/unet_attention.py:119-124, 296-333
//
// Before: ln_planes (x -> LN planes, 33 MB), the GeGLU projection (writes 67 MB of hidden planes at L = 1024, B = 16) and the
// K = 1024 output GEMM (streams them back).  Here a workgroup owns 64 rows end to end: the LayerNorm output stays in LDS as
// hi/lo planes (64 KB, the A operand of every ff1 slice), the 1024-wide hidden dimension is walked in 16 slices of 64 units -
// ff1 slice (64 x 128, value|gate) -> GeGLU in registers -> the slice's 64 x 64 product as planes in LDS (16 KB) -> 64 x 256
// ff2 accumulators in registers - and the hidden tensor never exists in HBM.
//
// Weights stream global->LDS (global_load_lds_dwordx4) through ONE ring of 16 KB slots shared by both matrices: a slice is
// 8 W1 slots (32 k x 128 n) followed by 4 W2 slots (16 k x 256 n), 12 MFMAs per wave per slot either way.  The MFMA pipeline is
// gemm_planes_bf3.hip's (skewed single fragment set, X = a_lo.w_hi, Z = a_hi.w_hi, Y = a_hi.w_lo, hand-counted lgkmcnt, the
// barrier of a slot sits between its last X and Z so the next slot's fragments load under Z / Y).  The accumulation order of
// every output element equals the unfused pair's (ascending k, X Z Y per 16-deep step), so the result is bit-identical to it.
#include "conv_common.h"

namespace pf {

struct MlpX {
  const float* gamma; const float* beta; float eps;
  const x3_t* w1; const float* b1;    // GeGLU projection: bf16x3 packing [K/8][plane][2048][8], columns value|gate interleaved by 32
  // TAIL (the SpatialTransformer's proj_out chained on, unet_attention.py:77-79): ff output + residual stay on chip as the A operand of
  // one more 256 x 256 projection; ConvP then describes THAT projection's epilogue (bias, block input as residual, statistics)
  const float* b2;                      // ff.net.2 bias (the ff residual is p.x0, the LayerNorm input)
  const x3_t* w3;                     // proj_out: bf16x3 packing [32][plane][256][8]
};

typedef x3_t x3x4_m __attribute__((ext_vector_type(4)));

template <int RING, bool TAIL>
__global__ __launch_bounds__(256, 1) void mlp_bf3_kernel(ConvP p, MlpX e) {
  constexpr int BM = 64, C = 256, HID = 1024, HS = 64, NSL = HID / HS;
  constexpr int A_B = BM * C * 4;           // 64 KB: LayerNorm planes, [chunk 8][plane 2][row 64][64 B]
  constexpr int SLOT_B = 16384;
  constexpr int H_B = BM * HS * 4;          // 16 KB: the slice's GeGLU product, [chunk 2][plane 2][row 64][64 B]
  constexpr int CH_B = 2 * BM * 64;         // bytes of one 32-deep A chunk (both planes)
  constexpr int LO_B = BM * 64;             // offset of the lo plane inside a chunk
  static_assert(12 % RING == 0, "a slice's 12 slots must map to fixed ring positions");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;
  unsigned char* sR = smem + A_B;
  unsigned char* sH = sR + RING * SLOT_B;
  float* sB1 = reinterpret_cast<float*>(sH + H_B);   // 2048 floats: the projection's bias in packed column order

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int lid;
  {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  const int L = p.Wout;
  const int b = fdiv(lid, p.d_tx);
  const int ox0 = (lid - b * p.tiles_x) * BM;
  const size_t row0 = (size_t)b * L + ox0;

  // ---- weight stream -------------------------------------------------------------------------------------------------
  // unit u = tid + 256 q of a slot (16 B each).  W1 slot (chunk r of slice j): k8 = 4 r + q, plane = tid / 128, n = 128 j + tid % 128.
  // W2 slot (16-deep step c of slice j): k8 = 8 j + 2 c + (q >> 1), plane = q & 1, n = tid.
  // (buffer-form direct-to-LDS loads, dma16 of conv_common.h: one SGPR resource per matrix, one offset VGPR per lane, the slot as a
  // wave-uniform SGPR offset)
  const __amdgpu_buffer_rsrc_t g1 = dma_resource(e.w1), g2 = dma_resource(p.w), g3 = dma_resource(TAIL ? static_cast<const void*>(e.w3) : p.w);
  const int v1 = (int)(((tid >> 7) * 2048 + (tid & 127)) * 16), vn = tid * 16;
  // one 1 KiB piece (q = 0..3) of a slot into ring position pos
  auto issue_w1 = [&](int js, int c, int pos, int q) {   // chunk c (32 k) of ff1 slice js
    dma16(g1, v1, (((c * 4 + q) * 2) * 2048 + js * 128) * 16, sR + pos * SLOT_B + wave * 1024 + q * 4096);
  };
  auto issue_wn = [&](const __amdgpu_buffer_rsrc_t& gbase, int kstep, int pos, int q) {   // 16-deep step `kstep` of a [K][256] matrix
    dma16(gbase, vn, (((kstep * 2 + (q >> 1)) * 2 + (q & 1)) * 256) * 16, sR + pos * SLOT_B + wave * 1024 + q * 4096);
  };
  auto issue_w2 = [&](int js, int c, int pos, int q) { issue_wn(g2, js * 4 + c, pos, q); };   // 16-deep step c of ff2 slice js
  // Slot order of the whole kernel (the GeGLU of slice j runs inside the MFMA gaps of ff1 slice j+1, so ff2 lags one slice):
  //   P0..P7 = ff1 slice 0 | for j = 0..14: M(j,0..7) = ff1 slice j+1, M(j,8..11) = ff2 slice j | F0..F3 = ff2 slice 15
  // every group is a multiple of RING slots, so a slot's ring position is its index within the group mod RING.
  for (int i = tid; i < 2 * HID; i += 256) sB1[i] = e.b1[i];

  // ---- LayerNorm of the tile into the resident planes (one wave per row, arithmetic of ln_planes_kernel) ---------------
  auto layer_norm_tile = [&]() {
    const f32x4 g = *reinterpret_cast<const f32x4*>(e.gamma + lane * 4), be = *reinterpret_cast<const f32x4*>(e.beta + lane * 4);
    constexpr int RPW = BM / 4;   // rows per wave
    f32x4 v[RPW];
    float red[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      v[i] = *reinterpret_cast<const f32x4*>(p.x0 + (row0 + wave * RPW + i) * C + lane * 4);
    }
    // same arithmetic as ln_planes_kernel (xor butterfly 32, 16, .., 1), but the 16 rows' reductions advance together: a
    // cross-lane move has ~60 cycles of latency and a row needs twelve of them in sequence
#pragma unroll
    for (int i = 0; i < RPW; ++i) red[i] = (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
      for (int i = 0; i < RPW; ++i) red[i] += __shfl_xor(red[i], off);
    float mu[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      mu[i] = red[i] / (float)C;
      const float a0 = v[i][0] - mu[i], a1 = v[i][1] - mu[i], a2 = v[i][2] - mu[i], a3 = v[i][3] - mu[i];
      red[i] = (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
      for (int i = 0; i < RPW; ++i) red[i] += __shfl_xor(red[i], off);
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int row = wave * RPW + i;
      const float rs = 1.0f / sqrtf(red[i] / (float)C + e.eps);
      const f32x4 y = (v[i] - mu[i]) * rs * g + be;
      const x3x4_m hi = __builtin_convertvector(y, x3x4_m);
      const x3x4_m lo = __builtin_convertvector(y - __builtin_convertvector(hi, f32x4), x3x4_m);
      // k = 4 lane: chunk = lane / 8, 16-byte slot = (lane % 8) / 2 (XOR-swizzled with the row), half = lane & 1
      unsigned char* d = sA + (lane >> 3) * CH_B + row * 64 + ((((lane & 7) >> 1) ^ ((row >> 2) & 3)) * 16) + (lane & 1) * 8;
      *reinterpret_cast<x3x4_m*>(d) = hi;
      *reinterpret_cast<x3x4_m*>(d + LO_B) = lo;
    }
  };

  // ---- fragment addresses ----------------------------------------------------------------------------------------------
  const unsigned ldsA = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)sA;
  const unsigned ldsR = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)sR;
  const unsigned ldsH = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)sH;
  const int arow = wm * 32 + (lane & 31);
  const int sw = ((lane & 31) >> 2) & 3;
  const int x0s = (sw & 2) | (((lane >> 5) ^ sw) & 1);
  const unsigned ab0 = ldsA + (arow * 4 + x0s) * 16, ab1 = ldsA + (arow * 4 + (x0s ^ 2)) * 16;      // K step 0 / 1 of a chunk
  const unsigned hb0 = ldsH + (arow * 4 + x0s) * 16, hb1 = ldsH + (arow * 4 + (x0s ^ 2)) * 16;
  const unsigned wb1 = ldsR + (((lane >> 5) * 2 * 128 + wn * 64 + (lane & 31)) * 16);                 // W1 slot: [k8 4][plane][128]
  const unsigned wb2 = ldsR + (((lane >> 5) * 2 * 256 + wn * 128 + (lane & 31)) * 16);                // W2 slot: [k8 2][plane][256]

  f32x16 acc1[2], acc2[1][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[0][i][r] = 0.f;

  // Fragment registers, double-buffered per SLOT: while the MFMAs of slot i run on one set, every fragment of slot i+1 is read into
  // the other (a 16-byte LDS read needs 100-200 cycles to come back; with the per-K-step skew of gemm_planes_bf3.hip this
  // tile's two-MFMA groups gave a read 64-128 cycles of cover and the wave spent more time waiting than multiplying).
  x3x8 fbh[2][2][2], fbl[2][2][2];                          // ff1 slot, weights: [set][K step][fn]
  // The A operand of ff1 - this wave's 32 rows of the LayerNorm planes, all 256 channels - is RESIDENT in registers (8 chunks x 2 K steps x
  // hi | lo = 128 registers; one wave per SIMD has 512): read from the LDS once instead of once per slice.  With them in the LDS an ff1
  // slot read 12 fragments for its 12 MFMAs - 48 KB per slot for the four waves, the LDS port busy for as long as the matrix pipe; a timing
  // variant without those reads ran 90 -> 78 us.
  x3x8 Al[8][2], Ah[8][2];
  x3x8 gal[2], gah[2], gbh[2][4], gbl[2][4];                // ff2 slot: [set]([fn])
#define SB() __builtin_amdgcn_sched_barrier(0)
#define IC(N) std::integral_constant<int, (N)>{}
#define FROM_H std::false_type{}
#define FROM_A std::true_type{}
#define OPEN_B std::true_type{}
#define OPEN_A std::false_type{}
  // n-th weight fragment read (in consumption order: bh0, bh1, bl0, bl1 per K step) of an ff1 slot at ring position RP into set S
  auto ld1 = [&](auto S_, auto RP_, auto N_) {
    constexpr int S = S_.value, RP = RP_.value, ks = N_.value / 4, m = N_.value % 4;
    if constexpr (m < 2) fbh[S][ks][m] = lds_read128<RP * SLOT_B + ((4 * ks) * 128 + m * 32) * 16>(wb1);
    else fbl[S][ks][m - 2] = lds_read128<RP * SLOT_B + ((4 * ks + 1) * 128 + (m - 2) * 32) * 16>(wb1);
  };
  // ff2 slot (16-deep step CC of the slice, ring position RP): n = 0..3 bh, 4..7 bl (weights), 8 al, 9 ah (the GeGLU product in sH)
  auto ld2 = [&](auto S_, auto CC_, auto RP_, auto N_, auto FROMA_) {
    constexpr int S = S_.value, CC = CC_.value, RP = RP_.value, n = N_.value;
    constexpr bool fa = decltype(FROMA_)::value;    // A operand from the resident 64 x 256 planes (sA) instead of the slice product (sH)
    if constexpr (n < 4) gbh[S][n] = lds_read128<RP * SLOT_B + (n * 32) * 16>(wb2);
    else if constexpr (n < 8) gbl[S][n - 4] = lds_read128<RP * SLOT_B + (256 + (n - 4) * 32) * 16>(wb2);
    else if constexpr (n == 8) gal[S] = lds_read128<(CC >> 1) * CH_B + LO_B>(fa ? ((CC & 1) ? ab1 : ab0) : ((CC & 1) ? hb1 : hb0));
    else gah[S] = lds_read128<(CC >> 1) * CH_B>(fa ? ((CC & 1) ? ab1 : ab0) : ((CC & 1) ? hb1 : hb0));
  };
  // n-th MFMA of a slot: per K step X X Z Z Y Y (ff1, two column fragments) / X X X X Z Z Z Z Y Y Y Y (ff2, four)
  auto mf1 = [&](auto S_, auto CC_, auto N_) {
    constexpr int S = S_.value, CC = CC_.value, ks = N_.value / 6, m = N_.value % 6, fn = m & 1;
    if constexpr (m < 2) acc1[fn] = x3_mfma_32x32x16(Al[CC][ks], fbh[S][ks][fn], acc1[fn], 0, 0, 0);
    else if constexpr (m < 4) acc1[fn] = x3_mfma_32x32x16(Ah[CC][ks], fbh[S][ks][fn], acc1[fn], 0, 0, 0);
    else acc1[fn] = x3_mfma_32x32x16(Ah[CC][ks], fbl[S][ks][fn], acc1[fn], 0, 0, 0);
  };
  auto mf2 = [&](auto S_, auto N_) {
    constexpr int S = S_.value, g = N_.value / 4, fn = N_.value % 4;
    if constexpr (g == 0) acc2[0][fn] = x3_mfma_32x32x16(gal[S], gbh[S][fn], acc2[0][fn], 0, 0, 0);
    else if constexpr (g == 1) acc2[0][fn] = x3_mfma_32x32x16(gah[S], gbh[S][fn], acc2[0][fn], 0, 0, 0);
    else acc2[0][fn] = x3_mfma_32x32x16(gah[S], gbl[S][fn], acc2[0][fn], 0, 0, 0);
  };
  // Slot hand-over of slot i, in two halves.  FRAGS_READY (before MFMA 0): every fragment of slot i has arrived in its registers.
  // SLOT_SYNC (after MFMA 1, so that the matrix pipe has work while the barrier resolves): this thread's pieces of slot i+1 have
  // landed (slot i+2 stays in flight); after the barrier slot i+1 is readable by everybody - its fragments are read behind MFMAs
  // 2..5 - and the ring position of slot i-1 is free: it is refilled with slot i+3, one piece after each of MFMAs 2..5
  // (as early as the barrier allows: under the contention of 256 workgroups streaming the same weights the two-slot lead is not generous).
  // (sched_barriers on both sides: an MFMA is a pure register operation to the compiler and would otherwise move across the waits)
#define FRAGS_READY() do { SB(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); SB(); } while (0)
#define SLOT_SYNC() do { SB(); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); SB(); } while (0)

  // ---- GeGLU of the PREVIOUS ff1 slice as 64 filler pieces for the MFMA gaps of the current one --------------------------------
  // A lone wave per SIMD hides about five single-issue instructions behind a 32-cycle MFMA and nothing else runs on the SIMD, so the
  // ~600 instructions of a slice's GeGLU (16 values per lane: bias, exact-erf GELU, product; then the lane-quad transposes, the
  // hi/lo split and the LDS stores) would cost their full 2.8 k cycles per slice (measured: 26 % of the kernel) anywhere but here.
  // gv / gg: the slice's raw value / gate accumulators, copied out of acc1 before the next slice overwrites it; bv / bg: its biases.
  f32x16 gv, gg;
  float bv = 0.f, bg = 0.f, bvn, bgn;
  float tz[16], tt[16], tq[16];
  f32x4 hq[4];
  x3x4_m hhi[4], hlo[4];
  f32x4 spl = {0.f, 0.f, 0.f, 0.f};   // scratch of the hi/lo split with registers of its own for the whole kernel (kept live below): as an ordinary temporary it
                                      // landed in the A-fragment registers of the MFMA just issued, and a VALU write to a register the matrix pipe is still reading stalls
  unsigned hdst[4];   // LDS byte address of this lane's 8-byte piece of the product, one per 8-row group q (hi plane; lo = + LO_B)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = wm * 32 + 8 * q + 4 * (lane >> 5) + (lane & 3);
    hdst[q] = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)sH + wn * CH_B + row * 64 +
              ((((lane & 31) >> 3) ^ ((row >> 2) & 3)) * 16) + (((lane & 31) >> 2) & 1) * 8;
  }
  const bool lp1 = lane & 1, lp2 = lane & 2;
  constexpr int NPIECE = 76;   // 19 per 8-row group: 4 values x 3 GELU steps, then transpose (4), split (2), store (1)
  auto geglu_piece = [&](auto K_) {
    constexpr int k = K_.value, q = k / 19, i = k % 19;
    if constexpr (i < 12) {
      constexpr int e = 4 * q + i / 3, st = i % 3;
      if constexpr (st == 0) {          // gelu_erf_f / erf_as_f of conv_common.h, cut in three
        tz[e] = PF_X3_UNSCALE(gg[e]) + bg;
        const float ax = fabsf(tz[e] * 0.70710678118654752440f);
        tt[e] = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
        tq[e] = ax;
      } else if constexpr (st == 1) {
        float pq = fmaf(1.061405429f, tt[e], -1.453152027f);
        pq = fmaf(pq, tt[e], 1.421413741f);
        pq = fmaf(pq, tt[e], -0.284496736f);
        pq = fmaf(pq, tt[e], 0.254829592f);
        tt[e] = pq * tt[e];
      } else {
        const float y = 1.0f - tt[e] * __expf(-tq[e] * tq[e]);
        const float er = copysignf(y, tz[e] * 0.70710678118654752440f);
        hq[q][e & 3] = (PF_X3_UNSCALE(gv[e]) + bv) * (0.5f * tz[e] * (1.0f + er));
      }
    } else if constexpr (i == 12 || i == 13) {   // quad_transpose stage 1 (conv_common.h), one register pair per piece
      constexpr int lo = (i - 12) * 2;            // pair (0,1) then (2,3)
      const float sv = lp1 ? hq[q][lo] : hq[q][lo + 1];
      const float rv = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, sv), 0xB1, 0xF, 0xF, true));
      const float k0 = hq[q][lo], k1 = hq[q][lo + 1];   // selects, not branches
      hq[q][lo] = lp1 ? rv : k0; hq[q][lo + 1] = lp1 ? k1 : rv;
    } else if constexpr (i == 14 || i == 15) {   // stage 2: pairs (0,2) then (1,3)
      constexpr int lo = i - 14;
      const float sv = lp2 ? hq[q][lo] : hq[q][lo + 2];
      const float rv = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, sv), 0x4E, 0xF, 0xF, true));
      const float k0 = hq[q][lo], k1 = hq[q][lo + 2];
      hq[q][lo] = lp2 ? rv : k0; hq[q][lo + 2] = lp2 ? k1 : rv;
      // now lane j of a quad holds row 8q + 4(lane>>5) + j, hidden units 4*((lane&31)>>2) .. +3 of this wave's 32
    } else if constexpr (i == 16) hhi[q] = __builtin_convertvector(hq[q], x3x4_m);
    else if constexpr (i == 17) { spl = __builtin_convertvector(hhi[q], f32x4); spl = hq[q] - spl; hlo[q] = __builtin_convertvector(spl, x3x4_m); asm volatile("" : "+v"(spl)); }
    else {
      typedef unsigned u32x2_m __attribute__((ext_vector_type(2)));
      const unsigned dd = hdst[q];   // (locals: clang does not capture a variable that a lambda names only in an asm operand)
      const u32x2_m vh = __builtin_bit_cast(u32x2_m, hhi[q]), vl = __builtin_bit_cast(u32x2_m, hlo[q]);
      // no "memory" clobber: with one, hipcc orders the store behind every direct-to-LDS load in flight (s_waitcnt vmcnt(0): the whole ring)
      asm volatile("ds_write_b64 %0, %1" ::"v"(dd), "v"(vh));
      asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(dd), "v"(vl), "n"(LO_B));
    }
  };
  // biases of ff1 slice js (read one slice ahead, as asm: a compiler-visible LDS read would make hipcc drain every fragment read in flight)
  auto load_bias = [&](int js) {
    const unsigned ba = (unsigned)(size_t)(__attribute__((address_space(3))) float*)sB1 + (js * 128 + wn * 64 + (lane & 31)) * 4;
    asm volatile("ds_read_b32 %0, %1" : "=v"(bvn) : "v"(ba));
    asm volatile("ds_read_b32 %0, %1 offset:128" : "=v"(bgn) : "v"(ba));
  };

  // one ff1 slot: chunk c, fragments from set c & 1.  NEXT (whose fragments are read meanwhile): 0 = chunk c+1, 1 = the first ff2 step of
  // a slice (weights only: its A operand does not exist yet), 2 = chunk 0 of the next ff1 slice.  FILL: GeGLU piece 12c + n after MFMA n
  auto ff1_slot = [&](auto cc, auto next, auto fill, auto&& dma) {
    constexpr int c = cc.value, S = c & 1, NS = S ^ 1, rpn = (c + 1) % RING;
    FRAGS_READY();
    static_for<0, 12>([&](auto nn) {
      constexpr int n = nn.value, m = n - 2;   // m: index among the gaps that carry the next slot's fragment reads
      mf1(IC(S), cc, nn); SB();
      if constexpr (n == 1) SLOT_SYNC();
      if constexpr (m >= 0 && m < 4) {
        if constexpr (next.value == 0) { ld1(IC(NS), IC(rpn), IC(2 * m)); ld1(IC(NS), IC(rpn), IC(2 * m + 1)); }
        else if constexpr (next.value == 1) { ld2(IC(0), IC(0), IC(rpn), IC(2 * m), FROM_H); ld2(IC(0), IC(0), IC(rpn), IC(2 * m + 1), FROM_H); }
        else { ld1(IC(NS), IC(rpn), IC(2 * m)); ld1(IC(NS), IC(rpn), IC(2 * m + 1)); }
      }
      if constexpr (n >= 2 && n < 6) dma(cc, IC(n - 2));
      if constexpr (fill.value && c * 12 + n < NPIECE) geglu_piece(IC(c * 12 + n));
      SB();
    });
  };
  // one ff2 slot (16-deep step c).  NEXT: 0 = step c+1, 1 = chunk 0 of an ff1 slice, 2 = step 0 of the next ff2 slice (weights only), 3 = nothing
  auto ff2_slot = [&](auto cc, auto next, auto&& dma, auto&& extra, auto froma, auto openb) {
    constexpr int c = cc.value, S = c & 1, NS = S ^ 1, rpn = (c + 1) % RING;
    FRAGS_READY();
    if constexpr (c == 0) {   // the barrier also publishes the A operand just written (GeGLU product / chained input), readable only now
      SLOT_SYNC();
      // OPEN_B: the pass starts behind compiler-scheduled code (or a loop exit): nothing may be in flight across that, so the weight
      // fragments of its first step are read here too instead of during the previous slot
      if constexpr (decltype(openb)::value) static_for<0, 8>([&](auto n) { ld2(IC(0), IC(0), IC(0), n, froma); });
      ld2(IC(0), IC(0), IC(0), IC(8), froma); ld2(IC(0), IC(0), IC(0), IC(9), froma);
      FRAGS_READY();
    }
    static_for<0, 12>([&](auto nn) {
      constexpr int n = nn.value, m = n - 2;
      mf2(IC(S), nn); SB();
      if constexpr (n == 1 && c != 0) SLOT_SYNC();
      if constexpr (m >= 0) {
        if constexpr (next.value == 0) { if constexpr (m < 5) { ld2(IC(NS), IC(c + 1), IC(rpn), IC(2 * m), froma); ld2(IC(NS), IC(c + 1), IC(rpn), IC(2 * m + 1), froma); } }
        else if constexpr (next.value == 1) { if constexpr (m < 4) { ld1(IC(NS), IC(rpn), IC(2 * m)); ld1(IC(NS), IC(rpn), IC(2 * m + 1)); } }
        else if constexpr (next.value == 2) { if constexpr (m < 4) { ld2(IC(NS), IC(0), IC(rpn), IC(2 * m), FROM_H); ld2(IC(NS), IC(0), IC(rpn), IC(2 * m + 1), FROM_H); } }
      }
      if constexpr (n >= 2 && n < 6) dma(cc, IC(n - 2));
      extra(nn);
      SB();
    });
  };
  auto no_extra = [](auto) {};
  // the finished ff1 accumulators move to gv / gg (eight registers after each of MFMAs 4..7 of the slice's last ff2 slot) and acc1 restarts
  auto take_acc = [&](auto nn) {
    constexpr int n = nn.value;
    if constexpr (n >= 4 && n < 8) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { gv[(n - 4) * 4 + r] = acc1[0][(n - 4) * 4 + r]; gg[(n - 4) * 4 + r] = acc1[1][(n - 4) * 4 + r]; }
    }
    if constexpr (n == 8) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc1[0][r] = 0.f; acc1[1][r] = 0.f; }
    }
  };

#pragma unroll
  for (int d = 0; d < RING - 1; ++d)
#pragma unroll
    for (int q = 0; q < 4; ++q) issue_w1(0, d, d, q);
  layer_norm_tile();
  SB();
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((RING - 2) * 4) : "memory");   // slot P0 landed, LN planes and bias written
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  static_for<0, 8>([&](auto n) { ld1(IC(0), IC(0), n); });
  static_for<0, 8>([&](auto c) {          // the resident A operand
    Al[c.value][0] = lds_read128<c.value * CH_B + LO_B>(ab0); Al[c.value][1] = lds_read128<c.value * CH_B + LO_B>(ab1);
    Ah[c.value][0] = lds_read128<c.value * CH_B>(ab0); Ah[c.value][1] = lds_read128<c.value * CH_B>(ab1);
  });
  load_bias(0);
  SB();
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc1[0][r] = 0.f; acc1[1][r] = 0.f; }

  // ---- P: ff1 slice 0 (nothing to fill the gaps with yet) ----
  static_for<0, 8>([&](auto cc) {
    auto dma = [&](auto c_, auto q_) {
      constexpr int t = c_.value + 3, q = q_.value;
      if constexpr (t < 8) issue_w1(0, t, t % RING, q); else issue_w1(1, t - 8, t % RING, q);
    };
    if constexpr (cc.value < 7) ff1_slot(cc, IC(0), IC(0), dma); else ff1_slot(cc, IC(2), IC(0), dma);
  });
  // no ff2 slot yet to hide the accumulator hand-over in: exposed once
  SB();
  gv = acc1[0]; gg = acc1[1];
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc1[0][r] = 0.f; acc1[1][r] = 0.f; }
  SB();

  // ---- M(j): ff1 slice j+1 with the GeGLU of slice j in its gaps, then ff2 slice j ----
  for (int j = 0; j < NSL - 1; ++j) {
    static_for<0, 8>([&](auto cc) {
      auto dma = [&](auto c_, auto q_) {
        constexpr int t = c_.value + 3, q = q_.value;
        if constexpr (t < 8) issue_w1(j + 1, t, t % RING, q); else issue_w2(j, t - 8, t % RING, q);
      };
      if constexpr (cc.value == 0) {   // once every LDS read has returned, the biases fetched one slice ago are in their registers
        FRAGS_READY();
        bv = bvn; bg = bgn;
        load_bias(j + 1);
      }
      if constexpr (cc.value < 7) ff1_slot(cc, IC(0), IC(1), dma); else ff1_slot(cc, IC(1), IC(1), dma);
    });
    static_for<0, 3>([&](auto cc) {
      auto dma = [&](auto c_, auto q_) {
        constexpr int t = 8 + c_.value + 3, q = q_.value;   // 11, 12, 13
        if constexpr (t < 12) issue_w2(j, t - 8, t % RING, q);
        else if (j < NSL - 2) issue_w1(j + 2, t - 12, t % RING, q);
        else issue_w2(NSL - 1, t - 12, t % RING, q);
      };
      ff2_slot(cc, IC(0), dma, no_extra, FROM_H, OPEN_A);
    });
    {
      auto dma = [&](auto, auto q_) {
        constexpr int q = q_.value;   // slot 14 of the iteration = slot 2 of the next group
        if (j < NSL - 2) issue_w1(j + 2, 2, 2, q); else issue_w2(NSL - 1, 2, 2, q);
      };
      // always "next = chunk 0 of an ff1 slice": after the last iteration those fragments are never used (ring position 0 then holds an
      // ff2 step) - one instruction stream for every j keeps hidden loads out of data-dependent control flow; F0 reads its own weights
      ff2_slot(IC(3), IC(1), dma, take_acc, FROM_H, OPEN_A);
    }
  }
  // ---- F: GeGLU of the last slice (nothing left to hide it behind), then ff2 slice 15 ----
  FRAGS_READY();
  // the fragment reads of an ff1 slice that never runs are dead values to the compiler: keep their registers until the wait above
#pragma unroll
  for (int k = 0; k < 2; ++k) {
#pragma unroll
    for (int i = 0; i < 2; ++i) asm volatile("" ::"v"(fbh[0][k][i]), "v"(fbl[0][k][i]));
  }
  bv = bvn; bg = bgn;
  static_for<0, NPIECE>([&](auto k) { geglu_piece(k); });
  SB();
  static_for<0, 4>([&](auto cc) {
    auto dma = [&](auto, auto q_) {   // F3, then (TAIL) the first proj_out steps - otherwise re-fetches nobody reads (keeps vmcnt(4) meaningful)
      constexpr int c = decltype(cc)::value, q = decltype(q_)::value;
      if constexpr (c == 0 || !TAIL) issue_w2(NSL - 1, 3, (c + 3) % RING, q); else issue_wn(g3, c - 1, (c + 3) % RING, q);
    };
    if constexpr (cc.value < 3) ff2_slot(cc, IC(0), dma, no_extra, FROM_H, OPEN_B);
    else ff2_slot(cc, IC(3), dma, no_extra, FROM_H, OPEN_B);
  });
  if constexpr (TAIL) {
    // ---- x2 = ff output + bias + x, straight from the accumulators into the resident A planes (same lane-quad transpose / split / 8-byte
    // stores as the GeGLU product); the ff1 slices are done with sA, and x2 itself is needed nowhere else (the layer's only consumer is
    // proj_out).  Then 16 more 16-deep steps against proj_out's weights.
    {
      const int cq = (lane & 31) & ~3;
      f32x4 b4[4];
#pragma unroll
      for (int fn = 0; fn < 4; ++fn) b4[fn] = *reinterpret_cast<const f32x4*>(e.b2 + wn * 128 + fn * 32 + cq);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = wm * 32 + 8 * q + 4 * (lane >> 5) + (lane & 3);
        f32x4 r4[4];
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) r4[fn] = *reinterpret_cast<const f32x4*>(p.x0 + (row0 + row) * C + wn * 128 + fn * 32 + cq);
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) {
          f32x4 v = {acc2[0][fn][4 * q], acc2[0][fn][4 * q + 1], acc2[0][fn][4 * q + 2], acc2[0][fn][4 * q + 3]};
          quad_transpose(v, lane);
          v = PF_X3_UNSCALE(v) + (b4[fn] + r4[fn]);        // acc + (bias + residual): the order of conv_epilogue's plane-pair path, which this replaces
          const x3x4_m hi = __builtin_convertvector(v, x3x4_m);
          const x3x4_m lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), x3x4_m);
          unsigned char* d = sA + (wn * 4 + fn) * CH_B + row * 64 + (((cq >> 3) ^ ((row >> 2) & 3)) * 16) + ((cq >> 2) & 1) * 8;
          *reinterpret_cast<x3x4_m*>(d) = hi;
          *reinterpret_cast<x3x4_m*>(d + LO_B) = lo;
        }
      }
#pragma unroll
      for (int fn = 0; fn < 4; ++fn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[0][fn][r] = 0.f;
    }
    SB();
    static_for<0, 16>([&](auto cc) {
      auto dma = [&](auto, auto q_) { issue_wn(g3, cc.value + 3 < 16 ? cc.value + 3 : 15, (cc.value + 3) % RING, q_.value); };
      if constexpr (cc.value < 15) ff2_slot(cc, IC(0), dma, no_extra, FROM_A, OPEN_B); else ff2_slot(cc, IC(3), dma, no_extra, FROM_A, OPEN_B);
    });
  }
#undef FRAGS_READY
#undef SLOT_SYNC
#undef IC
#undef FROM_H
#undef FROM_A
#undef OPEN_B
#undef OPEN_A
#undef SB
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  x3_unscale(acc2);
  conv_epilogue<1, BM, C, 1, 4, 2>(p, acc2, b, 0, ox0, 0, wm, wn, lane, tid, reinterpret_cast<float*>(sR));
}

// x: fp32 [B*L][256] (LayerNorm input AND residual); w1 / w2: bf16x3 packings of ff.net.0.proj (GeGLU-interleaved) and ff.net.2.
// w3 != nullptr chains the SpatialTransformer's proj_out (bf16x3 packing of the [256][256] 1x1 conv, bias b3) and its residual `res3`
// (the block input) onto the tile: out = res3 + b3 + W3 . (x + ff(LN(x))), with the per-64-row-tile channel statistics in `stats3`.
int launch_mlp_fused(const float* x, int batch, int l, const float* gamma, const float* beta, float eps, const void* w1, const float* b1,
                     const void* w2, const float* b2, float* out, void* out_planes, hipStream_t stream, const void* w3, const float* b3,
                     const float* res3, float* stats3) {
  PF_REQUIRE(x && gamma && beta && w1 && b1 && w2 && b2 && (out || out_planes), "mlp_fused: null argument");
  PF_REQUIRE(batch > 0 && l > 0 && l % 64 == 0, "mlp_fused: rows per sample must be a multiple of 64 (got %d)", l);
  PF_REQUIRE(!w3 || (b3 && res3 && out && !out_planes), "mlp_fused: the chained projection needs its bias, its residual and an fp32 output");
  constexpr int RING = 4;
  ConvP p;
  memset(&p, 0, sizeof p);
  p.x0 = x; p.c0 = 256; p.B = batch; p.Hin = 1; p.Win = l; p.Hout = 1; p.Wout = l;
  p.w = w2; p.N = 256; p.Npad = 256;
  p.bias = w3 ? b3 : b2; p.res = w3 ? res3 : x; p.ld_res = 256;
  p.out = out; p.ld_out = 256; p.out_planes = out_planes; p.stats = w3 ? stats3 : nullptr;
  p.ksplit = 1;
  p.tiles_x = l / 64; p.tiles_y = 1; p.nt = 1;
  conv_fill_divs(p);
  MlpX e{gamma, beta, eps, static_cast<const x3_t*>(w1), b1, b2, static_cast<const x3_t*>(w3)};
  constexpr size_t main_b = 65536 + RING * 16384 + 16384 + 8192;
  constexpr size_t epi_b = 65536 + (size_t)64 * (256 + 8) * 4;   // planes output: the fp32 tile is transposed through the ring region
  constexpr size_t lds = main_b > epi_b ? main_b : epi_b;
  static_assert(lds <= 160 * 1024, "LDS budget");
  static std::atomic<uint64_t> done_plain{0}, done_tail{0};
  if (int rc = set_max_lds_once(reinterpret_cast<const void*>(mlp_bf3_kernel<RING, false>), (int)lds, done_plain)) return rc;
  if (int rc = set_max_lds_once(reinterpret_cast<const void*>(mlp_bf3_kernel<RING, true>), (int)lds, done_tail)) return rc;
  const dim3 grid(batch * (l / 64)), block(256);
  if (w3) hipLaunchKernelGGL((mlp_bf3_kernel<RING, true>), grid, block, lds, stream, p, e);
  else hipLaunchKernelGGL((mlp_bf3_kernel<RING, false>), grid, block, lds, stream, p, e);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}

}  // namespace pf
