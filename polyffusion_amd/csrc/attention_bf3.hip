// attention_bf3.hip - self-attention softmax(q k^T / sqrt(d)) v on the bf16 matrix pipe with the bf16x3
// error-compensated split (see conv_bf16x3.hip), d_head = 64, L a multiple of 128.
//
// Operands arrive PRE-SPLIT: the q/k/v projection GEMM's epilogue (conv_common.h, "qkv planes") writes
//   Q, K  as hi/lo bf16 planes [B*L][C]            (row = token),
//   V^T   as hi/lo bf16 planes [B][H][64][L]        (row = head channel; inside every 16-token block the two middle
//                                                    token quads are swapped so that a lane's eight B-operand keys of
//                                                    the P^T fragment are 16 contiguous bytes),
// so this kernel does no conversion work on K/V at all: 64-key K and V^T tiles stream global->LDS directly
// (global_load_lds_dwordx4, double-buffered, counted vmcnt), Q fragments sit in registers.
// Both products are computed transposed exactly like attention.hip (S^T = K.Q^T, O^T += V^T.P^T), so the softmax
// statistics are lane-local and the fp32 accumulators of S^T, after exp and the hi/lo split, ARE the B operands
// of the second product.  LDS rows are 128 B; the 16-byte slot index is XOR-swizzled with ((row >> 1) & 7) on the source
// address of the direct load and on the fragment read, so a ds_read_b128 over 32 consecutive rows is conflict-free (see faddr).
#include <cstdlib>
#include <type_traits>
#include "conv_common.h"   // lds_read128 / lgkm_wait / static_for

namespace pf {

#ifdef PF_TRACE
__device__ unsigned long long g_trace_attn[4096];
#define TRA() do { if (trace_on && tslot < 2040) { g_trace_attn[tslot++] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define TRA() do {} while (0)
#endif

typedef x3_t x3x4 __attribute__((ext_vector_type(4)));

struct AttnP3 {
  const x3_t* planes;   // [qh | ql | kh | kl | vth | vtl], each B*L*C elements
  float* o; int ldo; x3_t* o_planes;
  int B, H, L;
  float scale;
  FastDiv d_nqt, d_h;     // reciprocals of the query tiles per (batch, head) and of H (tile decode without emulated divisions)
  // key split (small batches: 128-query form only): nsplit > 1 workgroups share a query tile, each walks L / nsplit keys and leaves its
  // un-normalised O^T with the running maximum and row sum in part_o / part_ml; attn_merge_kernel combines them
  int nsplit; float* part_o; float* part_ml; FastDiv d_ns;
};

// NWAVES waves x 32 queries per workgroup share every K / V^T tile; RING tile stages in LDS (32 KB each).
// <4, 2>: 128 queries, two workgroups per CU.  <8, 4>: 256 queries, one workgroup per CU - half the tile traffic and half
// the direct-to-LDS issue work per wave, three tiles of prefetch distance.
template <int NWAVES, int RING>
__global__ __launch_bounds__(NWAVES * 64, NWAVES == 4 ? 2 : 1) void attn_bf3_kernel(AttnP3 p) {
  constexpr int DH = 64, KT = 64, NT = NWAVES * 64, NP = 2048 / NT;   // NP direct-to-LDS pieces per thread and tile
  constexpr int STAGE = 4 * KT * DH;          // bf16 elements per stage: K hi, K lo, V^T hi, V^T lo (8 KB each)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  x3_t* sm = reinterpret_cast<x3_t*>(smem_raw);   // [RING stages][4 arrays][64 rows][64]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: LDS-DMA destinations (M0) stay in SGPRs
  // XCD-aware block order: workgroup ids go round-robin over the 8 XCDs, and every query tile of one (batch, head) streams the SAME
  // K / V^T planes (512 KB at L = 1024) - with the natural (qt, h, b) order its 8 query tiles sat on 8 different XCDs and each L2
  // fetched its own copy (FETCH_SIZE 141 MB for 50 MB of planes).  The remap gives XCD x the contiguous id range
  // [x*n/8, (x+1)*n/8), i.e. all query tiles of a (batch, head) share one L2.
  const int nqt = p.L / (NWAVES * 32);
  int lid;
  {
    const int orig = blockIdx.x, nwg = gridDim.x;
    const int xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  }
  int sp = 0;
  if (p.nsplit > 1) { const int t = fdiv(lid, p.d_ns); sp = lid - t * p.nsplit; lid = t; }   // key slice (fastest: the slices of a query tile run together)
  const int bh = fdiv(lid, p.d_nqt), qt = lid - bh * nqt;
  const int b = fdiv(bh, p.d_h), h = bh - b * p.H;
#ifdef PF_TRACE
  const bool trace_on = tid == 0 && qt == 1 && h == 1 && b == 1;
  int tslot = 0;
#endif
  const int g = lane >> 5;
  const int C = p.H * DH, L = p.L;
  const size_t MC = (size_t)p.B * L * C;
  const int qi = qt * (NWAVES * 32) + wave * 32 + (lane & 31);

  // Q fragments (B operand of S^T): lane (q, g) holds d = 16 s + 8 g .. +7 of its query, hi and lo
  x3x8 qh[4], ql[4];
  {
    const x3_t* qp = p.planes + ((size_t)b * L + qi) * C + h * DH + 8 * g;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qh[s] = *reinterpret_cast<const x3x8*>(qp + 16 * s);
      ql[s] = *reinterpret_cast<const x3x8*>(qp + MC + 16 * s);
    }
  }

  // direct-to-LDS tile loads: 2048 16-byte units per stage, NP per thread; unit u = tid + j*NT -> array u/512 (K hi, K lo,
  // V^T hi, V^T lo), row (u%512)/8, slot u%8
  const x3_t* kbase = p.planes + 2 * MC + (size_t)b * L * C + h * DH;           // + plane*MC + key*C + d
  const x3_t* vbase = p.planes + 4 * MC + ((size_t)b * p.H + h) * DH * L;       // + plane*MC + d*L + key
  // (buffer-form direct-to-LDS loads, dma16 of conv_common.h; the lo plane lies MC elements behind the hi plane: with 2 * MC bytes >= 2 GiB
  // the launcher refuses, so every offset fits the 32-bit offset registers)
  const __amdgpu_buffer_rsrc_t rsK = dma_resource(kbase), rsV = dma_resource(vbase);
  auto issue_piece = [&](int t, int stage, int j) {
    const int u = tid + j * NT;
    const int arr = u >> 9, row = (u & 511) >> 3;
    const int src_slot = (u & 7) ^ ((row >> 1) & 7);     // swizzle on the SOURCE side; the LDS image stays lane-linear
    x3_t* lp = sm + stage * STAGE + (j * NT + wave * 64) * 8;   // wave-uniform; the hardware adds lane*16 B
    if (arr < 2) dma16(rsK, (int)(((size_t)(arr & 1) * MC + (size_t)row * C + src_slot * 8) * 2), t * KT * C * 2, lp);
    else dma16(rsV, (int)(((size_t)(arr & 1) * MC + (size_t)row * L + src_slot * 8) * 2), t * KT * 2, lp);
  };
  auto issue_tile = [&](int t, int stage) {
#pragma unroll
    for (int j = 0; j < NP; ++j) issue_piece(t, stage, j);
  };

  f32x16 oacc[2];
#pragma unroll
  for (int df = 0; df < 2; ++df)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[df][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int ntile_all = L / KT;        // even: L is a multiple of 128
  const int ntile = ntile_all / p.nsplit, tile0 = sp * ntile;   // this workgroup's keys: tiles [tile0, tile0 + ntile), ntile a multiple of RING (launcher)
  const int r31 = lane & 31;
  // LDS byte address of this lane's fragment row inside an 8 KB array, one per K-step pair: row = r31 (+32 per fragment,
  // an immediate), 16-byte slot = (2 sp + g) ^ ((row >> 1) & 7).  K and V^T fragments share the formula (row = key or channel).
  // The key is row >> 1, not row: the LDS of this part has 64 banks (256 B), a ds_read_b128 is served sixteen lanes at a time, and
  // sixteen consecutive 128-byte rows must spread over both halves of the 256 bytes AND all eight slots of each half - with (row & 7)
  // rows r and r + 8 met in the same banks (SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE; tools/micro/lds_b128.hip: 32 against
  // 24.8 cycles per wave-instruction with four waves reading).
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) x3_t*)sm;
  unsigned faddr[4];
#pragma unroll
  for (int sp = 0; sp < 4; ++sp) faddr[sp] = lds0 + r31 * 128 + (((2 * sp + g) ^ ((r31 >> 1) & 7)) * 16);
  constexpr int STAGE_B = STAGE * 2, ARR_B = KT * DH * 2;
#define SB() __builtin_amdgcn_sched_barrier(0)

  // One tile = 4 K-step pairs of S^T = K.Q^T, the softmax, and 4 key blocks of O^T += V^T.P^T.  Each step issues the four
  // ds_read_b128 (hi/lo of two fragments) of the NEXT step before its own six MFMAs, which alternate between two
  // independent accumulators so the wave's in-order issue never stalls on an accumulator dependency; the next tile's eight
  // direct-to-LDS loads are spread over the S steps and the hi/lo split of the next key block's probabilities is
  // interleaved with the MFMAs of the current one.  LDS waits are counted by hand (inline-asm reads, conv_common.h).
  // One barrier per tile: it publishes tile t and frees the other stage for tile t+1.
  typedef float f32x8 __attribute__((ext_vector_type(8)));
  auto psplit = [&](const f32x16& sv, int half, x3x8& ph, x3x8& pl) {   // whole-vector conversions (packed cvt path)
    f32x8 v;
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = sv[half * 8 + q];
    ph = __builtin_convertvector(v, x3x8);
    pl = __builtin_convertvector(v - __builtin_convertvector(ph, f32x8), x3x8);
  };
  auto tile_body = [&](auto stc, int t) {
    constexpr int ST = decltype(stc)::value;
    constexpr int SB0 = ST * STAGE_B;
    TRA();
    // tile t has landed once only the RING-2 newer tiles of this thread are outstanding; the barrier publishes it and
    // frees the stage of tile t-1, which is refilled with tile t+RING-1 during the S phase
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((RING - 2) * NP) : "memory");
    TRA();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    TRA();
    const int tn = tile0 + min(t + RING - 1, ntile - 1);
    constexpr int STN = (ST + RING - 1) % RING;

    // ---- S^T = K . Q^T ----
    f32x16 s[2];
#pragma unroll
    for (int kf = 0; kf < 2; ++kf)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kf][r] = 0.f;
    x3x8 fh[2][2], fl[2][2];   // [buffer][fragment]
    fh[0][0] = lds_read128<SB0>(faddr[0]);          fl[0][0] = lds_read128<SB0 + ARR_B>(faddr[0]);
    fh[0][1] = lds_read128<SB0 + 4096>(faddr[0]);   fl[0][1] = lds_read128<SB0 + ARR_B + 4096>(faddr[0]);
    static_for<0, 4>([&](auto ic) {
      constexpr int sp = decltype(ic)::value, b = sp & 1;
      SB();
      if constexpr (sp < 3) {
        fh[b ^ 1][0] = lds_read128<SB0>(faddr[sp + 1]);          fl[b ^ 1][0] = lds_read128<SB0 + ARR_B>(faddr[sp + 1]);
        fh[b ^ 1][1] = lds_read128<SB0 + 4096>(faddr[sp + 1]);   fl[b ^ 1][1] = lds_read128<SB0 + ARR_B + 4096>(faddr[sp + 1]);
      } else {   // first V^T fragments, in flight across the softmax
        fh[b ^ 1][0] = lds_read128<SB0 + 2 * ARR_B>(faddr[0]);          fl[b ^ 1][0] = lds_read128<SB0 + 3 * ARR_B>(faddr[0]);
        fh[b ^ 1][1] = lds_read128<SB0 + 2 * ARR_B + 4096>(faddr[0]);   fl[b ^ 1][1] = lds_read128<SB0 + 3 * ARR_B + 4096>(faddr[0]);
      }
      if constexpr (NP == 8) { issue_piece(tn, STN, 2 * sp); issue_piece(tn, STN, 2 * sp + 1); }
      else issue_piece(tn, STN, sp);
      lgkm_wait<4>(); SB();
      s[0] = x3_mfma_32x32x16(fl[b][0], qh[sp], s[0], 0, 0, 0);
      s[1] = x3_mfma_32x32x16(fl[b][1], qh[sp], s[1], 0, 0, 0);
      s[0] = x3_mfma_32x32x16(fh[b][0], ql[sp], s[0], 0, 0, 0);
      s[1] = x3_mfma_32x32x16(fh[b][1], ql[sp], s[1], 0, 0, 0);
      s[0] = x3_mfma_32x32x16(fh[b][0], qh[sp], s[0], 0, 0, 0);
      s[1] = x3_mfma_32x32x16(fh[b][1], qh[sp], s[1], 0, 0, 0);
      SB();
    });
    TRA();
    // ---- online softmax in the exp2 domain (lane holds 32 keys of its query, the partner lane^32 the other 32).
    // Whole-vector arithmetic lets the compiler use the packed fp32 ALU ops; the running maximum only moves - and the
    // accumulators are only rescaled - when some query of the wave actually found a larger score in this tile.
    const float c2 = p.scale * 1.44269504088896340736f;
    float mx = -INFINITY;
#pragma unroll
    for (int kf = 0; kf < 2; ++kf) {
      s[kf] = s[kf] * c2;
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kf][r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    if (__builtin_amdgcn_ballot_w64(mx > m_run) != 0) {   // wave-uniform
      const float m_new = fmaxf(m_run, mx);
      const float alpha = exp2f(m_run - m_new);
      l_run *= alpha;
      m_run = m_new;
#pragma unroll
      for (int df = 0; df < 2; ++df) oacc[df] = oacc[df] * alpha;
    }
    float psum = 0.f;
#pragma unroll
    for (int kf = 0; kf < 2; ++kf) {
      s[kf] = s[kf] - m_run;
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float pv = __builtin_amdgcn_exp2f(s[kf][r]); s[kf][r] = pv; psum += pv; }
    }
    l_run += psum;
    TRA();
    // ---- O^T += V^T . P^T : key block kb (16 keys); registers 8kb'..8kb'+7 of a score fragment are this lane's 8 keys of
    // block kb.  Block 0's V^T fragments sit in buffer 0 (loaded by S step 3: b ^ 1 == 0). ----
    x3x8 ph[2], pl[2];
    psplit(s[0], 0, ph[0], pl[0]);
    static_for<0, 4>([&](auto jc) {
      constexpr int kb = decltype(jc)::value, b = kb & 1;
      SB();
      if constexpr (kb < 3) {
        fh[b ^ 1][0] = lds_read128<SB0 + 2 * ARR_B>(faddr[kb + 1]);          fl[b ^ 1][0] = lds_read128<SB0 + 3 * ARR_B>(faddr[kb + 1]);
        fh[b ^ 1][1] = lds_read128<SB0 + 2 * ARR_B + 4096>(faddr[kb + 1]);   fl[b ^ 1][1] = lds_read128<SB0 + 3 * ARR_B + 4096>(faddr[kb + 1]);
      }
      lgkm_wait<(kb < 3) ? 4 : 0>(); SB();
      oacc[0] = x3_mfma_32x32x16(fl[b][0], ph[b], oacc[0], 0, 0, 0);
      oacc[1] = x3_mfma_32x32x16(fl[b][1], ph[b], oacc[1], 0, 0, 0);
      if constexpr (kb < 3) psplit(s[(kb + 1) >> 1], (kb + 1) & 1, ph[b ^ 1], pl[b ^ 1]);   // next block's split, in the MFMA shadow
      oacc[0] = x3_mfma_32x32x16(fh[b][0], pl[b], oacc[0], 0, 0, 0);
      oacc[1] = x3_mfma_32x32x16(fh[b][1], pl[b], oacc[1], 0, 0, 0);
      oacc[0] = x3_mfma_32x32x16(fh[b][0], ph[b], oacc[0], 0, 0, 0);
      oacc[1] = x3_mfma_32x32x16(fh[b][1], ph[b], oacc[1], 0, 0, 0);
      SB();
    });
  };
  TRA();
#pragma unroll
  for (int d = 0; d < RING - 1; ++d) issue_tile(tile0 + min(d, ntile - 1), d);
  for (int t = 0; t < ntile; t += RING)   // ntile is a multiple of RING (launcher)
    static_for<0, RING>([&](auto st) { tile_body(st, t + decltype(st)::value); });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped tail prefetch
#undef SB

  if (p.nsplit > 1) {   // partial result of this key slice: O^T as accumulated (relative to m_run), m_run, the row sum
    const size_t row = (((size_t)sp * p.B + b) * p.H + h) * L + qi;
    float* po = p.part_o + row * DH;
#pragma unroll
    for (int df = 0; df < 2; ++df)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        *reinterpret_cast<f32x4*>(po + df * 32 + 8 * c + 4 * g) = f32x4{oacc[df][4 * c + 0], oacc[df][4 * c + 1], oacc[df][4 * c + 2], oacc[df][4 * c + 3]};
    const float lsum = l_run + __shfl_xor(l_run, 32);
    if (g == 0) { p.part_ml[row * 2] = m_run; p.part_ml[row * 2 + 1] = lsum; }
    return;
  }
  const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32));
  float* op = p.o + ((size_t)b * L + qi) * p.ldo + h * DH;
#pragma unroll
  for (int df = 0; df < 2; ++df)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f32x4 v;
      v[0] = oacc[df][4 * c + 0] * inv; v[1] = oacc[df][4 * c + 1] * inv;
      v[2] = oacc[df][4 * c + 2] * inv; v[3] = oacc[df][4 * c + 3] * inv;
      if (p.o_planes) {   // hi/lo planes [M][C] for the to_out planes GEMM
        x3x4 h4 = __builtin_convertvector(v, x3x4);
        const f32x4 hf = __builtin_convertvector(h4, f32x4);
        x3x4 l4 = __builtin_convertvector(v - hf, x3x4);
        x3_t* pp = p.o_planes + ((size_t)b * L + qi) * C + h * DH + df * 32 + 8 * c + 4 * g;
        *reinterpret_cast<x3x4*>(pp) = h4;
        *reinterpret_cast<x3x4*>(pp + MC) = l4;
      } else {
        *reinterpret_cast<f32x4*>(op + df * 32 + 8 * c + 4 * g) = v;
      }
    }
}

// ---- 256-query form: two query fragments per wave, softmax software-pipelined across tiles --------------------------------------
// With one 32-query fragment per wave every wave reads the whole K / V^T tile for 48 MFMAs; here a fragment read feeds two query
// fragments (half the LDS bytes per MFMA, half the workgroups streaming tiles).  That costs the second wave per SIMD (one wave owns
// the whole 512-register file: O^T and S^T accumulators in the accumulator file, scores / probabilities in arch VGPRs), so nothing
// covers the softmax any more - it is pipelined by hand instead: while the matrix pipe computes S(t+1) the vector ALU turns the
// scores of tile t into probabilities (fma, exp2, row sums) and splits the first key block, while it computes O += V.P(t) the ALU
// splits the other key blocks, moves the raw scores of S(t+1) out of the accumulator file and reduces their row maxima.  The
// reference exponent of a row moves lazily (see TAU), so the accumulators are rescaled a few times per row, in a wave-uniform
// branch at the head of an iteration where no register load is in flight.
// Measured (B = 16, 4 heads, L = 1024: 256 workgroups): 57 -> 51 us per launch in a back-to-back loop, 70 -> 64 us inside the
// step (kernel trace); the whole step does not move (+-0.3 %), the part runs power-managed (sclk ~2.0 GHz in the step, ~1.65 GHz
// in a loop of this kernel alone) and a kernel that idles less clocks lower.  Cycle stamps: an S region (12 MFMAs) takes ~500 cycles,
// a PV region 850-1200 against 384 at pipe speed - the open problem of this kernel.
template <int RING>
__global__ __launch_bounds__(256, 1) void attn_bf3_wide_kernel(AttnP3 p) {
  constexpr int NWAVES = 4, QF = 2, DH = 64, KT = 64, NT = NWAVES * 64, NP = 2048 / NT;
  constexpr int STAGE = 4 * KT * DH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  x3_t* sm = reinterpret_cast<x3_t*>(smem_raw);   // [RING stages][K hi, K lo, V^T hi, V^T lo][64 rows][64]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nqt = p.L / (NWAVES * 32 * QF);
  int lid;
  {   // XCD-aware order: all query tiles of a (batch, head) on one XCD's L2 (see attn_bf3_kernel)
    const int orig = blockIdx.x, nwg = gridDim.x;
    const int xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  }
  const int bh = fdiv(lid, p.d_nqt), qt = lid - bh * nqt;
  const int b = fdiv(bh, p.d_h), h = bh - b * p.H;
#ifdef PF_TRACE
  const bool trace_on = tid == 0 && qt == 1 && h == 1 && b == 1;
  int tslot = 0;
#endif
  const int g = lane >> 5;
  const int C = p.H * DH, L = p.L;
  const size_t MC = (size_t)p.B * L * C;
  const int qi = qt * (NWAVES * 32 * QF) + wave * (32 * QF) + (lane & 31);   // + 32 qf

  x3x8 qh[QF][4], ql[QF][4];
#pragma unroll
  for (int qf = 0; qf < QF; ++qf) {
    const x3_t* qp = p.planes + ((size_t)b * L + qi + 32 * qf) * C + h * DH + 8 * g;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qh[qf][s] = *reinterpret_cast<const x3x8*>(qp + 16 * s);
      ql[qf][s] = *reinterpret_cast<const x3x8*>(qp + MC + 16 * s);
    }
  }

  const x3_t* kbase = p.planes + 2 * MC + (size_t)b * L * C + h * DH;
  const x3_t* vbase = p.planes + 4 * MC + ((size_t)b * p.H + h) * DH * L;
  // piece j of a tile (unit u = tid + 256 j): array j/2 (K hi, K lo, V^T hi, V^T lo), row tid/8 + 32 (j & 1), 16-byte slot
  // (tid & 7) ^ ((row >> 1) & 7) - the same slot for every j.  The source address is a wave-uniform base (tile, array, row half: scalar ALU)
  // plus one 32-bit per-thread offset per operand, so an issue costs no vector instructions and no address registers.
  const unsigned rsl = (unsigned)(tid >> 3), ssl = (unsigned)((tid & 7) ^ ((tid >> 4) & 7)) * 8u;
  const unsigned koff = (rsl * (unsigned)C + ssl) * 2u, voff = (rsl * (unsigned)L + ssl) * 2u;   // bytes
  // Issued in the buffer form (dma16, conv_common.h): SGPR resource + that one offset VGPR + the wave-uniform part as an SGPR offset.
  const __amdgpu_buffer_rsrc_t rsK = dma_resource(kbase), rsV = dma_resource(vbase);
  auto issue_piece = [&](int t, int stage, int j) {
    const int arr = j >> 1, half = j & 1;
    x3_t* lp = sm + stage * STAGE + (j * NT + wave * 64) * 8;
    if (arr < 2) dma16(rsK, (int)koff, (int)(((size_t)(arr & 1) * MC + (size_t)(t * KT + 32 * half) * C) * 2), lp);
    else dma16(rsV, (int)voff, (int)(((size_t)(arr & 1) * MC + (size_t)(32 * half) * L + t * KT) * 2), lp);
  };
  auto issue_tile = [&](int t, int stage) {
#pragma unroll
    for (int j = 0; j < NP; ++j) issue_piece(t, stage, j);
  };

  f32x16 oacc[QF][2];       // O^T accumulators (accumulator file)
  f32x16 sacc[QF][2];       // S^T accumulators of the tile being multiplied (accumulator file)
  f32x16 pv[QF][2];         // arch VGPRs: scores of the tile in the softmax - raw, then probabilities; a key block's sixteen registers
                            // are refilled with the next tile's raw scores as soon as its probabilities have been split
  float m_run[QF], l_run[QF], alpha[QF], mx[QF];
  constexpr float TAU = 6.f;   // the reference exponent of a row only moves when its maximum grew by more than 2^TAU (see below)
#pragma unroll
  for (int qf = 0; qf < QF; ++qf) {
    m_run[qf] = -INFINITY; l_run[qf] = 0.f; alpha[qf] = 0.f; mx[qf] = -INFINITY;
#pragma unroll
    for (int df = 0; df < 2; ++df)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[qf][df][r] = 0.f;
  }

  const int ntile = L / KT;
  const int r31 = lane & 31;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) x3_t*)sm;
  unsigned faddr[4];
#pragma unroll
  for (int sp = 0; sp < 4; ++sp) faddr[sp] = lds0 + r31 * 128 + (((2 * sp + g) ^ ((r31 >> 1) & 7)) * 16);
  constexpr int STAGE_B = STAGE * 2, ARR_B = KT * DH * 2;
  const float c2 = p.scale * 1.44269504088896340736f;
#define SB() __builtin_amdgcn_sched_barrier(0)
#define IC(N) std::integral_constant<int, (N)>{}
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef x3_t x3x2 __attribute__((ext_vector_type(2)));
  x3x8 fh[2][2], fl[2][2];      // [buffer][fragment]
  u32x4 phw[2][QF], plw[2][QF];   // probabilities of one key block as packed bf16 pairs, hi and lo [buffer][query fragment]
  float th0[8], th1[8];           // hi/lo split in flight (one slot per item of a key block)

  // ---- the static schedule of one iteration ------------------------------------------------------------------------------------
  // 96 gaps = 96 MFMAs: gaps 0..47 are S(t+1) (region r = gap / 12 is K step r), gaps 48..95 are PV(t) (region 4 + kb).  Inside a
  // region MFMA m is product m / 4 of the split (a_lo.b_hi, a_hi.b_lo, a_hi.b_hi) on accumulator (qf, kf) = ((m >> 1) & 1, m & 1).
  // Every gap carries one MFMA and its share of the ALU items below, fenced from its neighbours; a measured lone wave hides about
  // four single-issue instructions per MFMA and pays ~3 cycles for each further one (tools/micro/mfma_war.hip), a v_accvgpr_read of
  // an idle accumulator costs what a v_mov costs (tools/micro/mfma_acc.hip), four ds_read_b128 or two LDS-DMA issues back to back
  // stall the wave for 50-150 cycles each - hence one LDS read per gap and an even ~5.8 instructions per gap:
  //  * exp item i (0..63: key block i / 16, query fragment (i >> 3) & 1, key i & 7), three dependent instructions in three
  //    consecutive gaps from gap 3 i / 4: fma (s c2 - m), exp2, row-sum add;
  //  * split item (block b, j = 4 qf + pair): cvt_pk | shift + mask | two subtractions | cvt_pk in four consecutive gaps.  Block 0
  //    from gap 16 + 4 j (its exponentials are done by gap 14), block b >= 1 from gap 48 + 12 (b - 1) + j - inside the PV region
  //    before the one that multiplies it;
  //  * copy pair c (0..31: block c / 8): two raw scores of S(t+1) from the accumulator file into the registers of a key block whose
  //    probabilities have been split, one v_max3 - pairs 0-4 in PV region 0, 5-9 in region 1, 10-14 in region 2, the rest in region 3;
  //  * the four fragment reads for the next region in gaps 4..7 of a region, the tile's eight direct-to-LDS pieces in gap 9.
  auto exp_stage = [&](auto ic, auto stc) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value, ST = decltype(stc)::value, blk = i >> 4, qf = (i >> 3) & 1, e = 8 * (blk & 1) + (i & 7);
    const float x = pv[qf][blk >> 1][e];
    if constexpr (ST == 0) pv[qf][blk >> 1][e] = __builtin_fmaf(x, c2, -m_run[qf]);
    else if constexpr (ST == 1) pv[qf][blk >> 1][e] = __builtin_amdgcn_exp2f(x);
    else l_run[qf] += x;
  };
  auto split_stage = [&](auto blkc, auto jc, auto stc) __attribute__((always_inline)) {
    constexpr int BLK = decltype(blkc)::value, j8 = decltype(jc)::value, ST = decltype(stc)::value, BUF = BLK & 1;
    constexpr int qf = j8 >> 2, j = j8 & 3, e = 8 * (BLK & 1) + 2 * j;
    const float v0 = pv[qf][BLK >> 1][e], v1 = pv[qf][BLK >> 1][e + 1];
    if constexpr (ST == 0) {
      const f32x2 v = {v0, v1};
      phw[BUF][qf][j] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, x3x2));   // hi = RNE bf16 pair (key 2j in the low half)
    } else if constexpr (ST == 1) {   // the float values of hi, rebuilt from the packed word
#ifdef PF_X3_F16
      // (spelled as instructions: hipcc 7.2 folds `(float)bit_cast<half2>(phw[..][j])[i]` to element j = 0's conversion for every j)
      const unsigned w = phw[BUF][qf][j];
      asm("v_cvt_f32_f16_e32 %0, %1" : "=v"(th0[j8]) : "v"(w));
      asm("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(th1[j8]) : "v"(w));
#else
      th0[j8] = __uint_as_float(phw[BUF][qf][j] << 16);
      th1[j8] = __uint_as_float(phw[BUF][qf][j] & 0xffff0000u);
#endif
    } else if constexpr (ST == 2) {
      th0[j8] = v0 - th0[j8];
      th1[j8] = v1 - th1[j8];
    } else {
      const f32x2 r = {th0[j8], th1[j8]};
      plw[BUF][qf][j] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, x3x2));
    }
  };
  auto copy_pair = [&](auto cc) __attribute__((always_inline)) {
    constexpr int c = decltype(cc)::value, blk = c >> 3, qf = (c >> 2) & 1, e = 8 * (blk & 1) + 2 * (c & 3);
    const float r0 = sacc[qf][blk >> 1][e], r1 = sacc[qf][blk >> 1][e + 1];
    pv[qf][blk >> 1][e] = r0;
    pv[qf][blk >> 1][e + 1] = r1;
    mx[qf] = fmaxf(fmaxf(mx[qf], r0), r1);
  };
  // ALU items of gap G
  auto gap_items = [&](auto gc, auto dosc, auto dopvc) __attribute__((always_inline)) {
    constexpr int G = decltype(gc)::value;
    constexpr bool DO_S = decltype(dosc)::value, DO_PV = decltype(dopvc)::value;
    if constexpr (DO_PV) {
      static_for<0, 64>([&](auto ic) {
        constexpr int i = decltype(ic)::value, g0 = (3 * i) >> 2;
        if constexpr (G >= g0 && G <= g0 + 2) exp_stage(ic, IC(G - g0));
      });
      static_for<0, 8>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (G >= 16 + 4 * j && G <= 19 + 4 * j) split_stage(IC(0), jc, IC(G - 16 - 4 * j));
        static_for<1, 4>([&](auto bc) {
          constexpr int b = decltype(bc)::value, g0 = 48 + 12 * (b - 1) + j;
          if constexpr (G >= g0 && G <= g0 + 3) split_stage(bc, jc, IC(G - g0));
        });
      });
    }
    if constexpr (DO_S) {
      static_for<0, 32>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        // pairs 0..14: five per PV region 0..2 at gaps 1, 3, 5, 8, 10 of the region; pairs 15..31: PV region 3, gap (c - 15) * 12 / 17
        constexpr int five[5] = {1, 3, 5, 8, 10};
        constexpr int g = c < 15 ? 48 + 12 * (c / 5) + five[c % 5] : 84 + ((c - 15) * 12) / 17;
        if constexpr (G == g) copy_pair(cc);
      });
    }
  };

  // One iteration: S(t+1) [DO_S] beside the exponentials of tile t, then PV(t) [DO_PV] beside the hi/lo splits of tile t and the
  // raw scores + row maxima of S(t+1).  ST1 = LDS stage of tile t+1, ST0 = stage of tile t.
  auto iteration = [&](auto st1c, auto dosc, auto dopvc, int t) __attribute__((always_inline)) {
    constexpr int ST1 = decltype(st1c)::value, ST0 = (ST1 + RING - 1) % RING;
    constexpr bool DO_S = decltype(dosc)::value, DO_PV = decltype(dopvc)::value;
    constexpr int SB1 = ST1 * STAGE_B, SB0 = ST0 * STAGE_B;
    TRA();
    if constexpr (DO_S) {
      // tile t+1 has landed once only tile t+2 of this thread is outstanding; the barrier publishes it and frees the stage of tile t-1
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((RING - 3) * NP) : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    TRA();
    if constexpr (DO_PV) {
      // lazy rescale: rows whose reference exponent moved (alpha != 1) - rare after the first tiles; no register load is in flight here
      bool moved = false;
#pragma unroll
      for (int qf = 0; qf < QF; ++qf) moved |= alpha[qf] != 1.0f;
      if (__builtin_amdgcn_ballot_w64(moved) != 0) {
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
          l_run[qf] *= alpha[qf];
#pragma unroll
          for (int df = 0; df < 2; ++df) oacc[qf][df] = oacc[qf][df] * alpha[qf];
        }
      }
    }
    if constexpr (DO_S) {
#pragma unroll
      for (int qf = 0; qf < QF; ++qf) mx[qf] = -INFINITY;
    }
    const int tn = min(t + RING - 1, ntile - 1);
    constexpr int STN = (ST1 + RING - 2) % RING;   // stage of tile t-1 == stage of tile t+RING-1

    // fragments of region 0 (K step 0 of tile t+1, or - last iteration - nothing: region 3 reads V^T block 0 as always)
    if constexpr (DO_S) {
      fh[0][0] = lds_read128<SB1>(faddr[0]);          fl[0][0] = lds_read128<SB1 + ARR_B>(faddr[0]);
      fh[0][1] = lds_read128<SB1 + 4096>(faddr[0]);   fl[0][1] = lds_read128<SB1 + ARR_B + 4096>(faddr[0]);
    }
    static_for<0, 96>([&](auto gc) {
      constexpr int G = decltype(gc)::value, R = G / 12, m = G % 12, bb = R & 1, prod = m >> 2, qf = (m >> 1) & 1, kf = m & 1;
      constexpr bool S_REG = R < 4;
      if constexpr (m == 0) {   // region head: this region's fragments (read in gaps 4..7 of the previous region) have landed
        if constexpr (G == 48) TRA();
        SB(); lgkm_wait<0>(); SB();
      }
      // ---- the MFMA ----
      if constexpr (S_REG && DO_S) {
        constexpr int sp = R;
        const x3x8 a = prod == 0 ? fl[bb][kf] : fh[bb][kf];
        const x3x8 bq = prod == 1 ? ql[qf][sp] : qh[qf][sp];
        if constexpr (sp == 0 && prod == 0) {
          const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          sacc[qf][kf] = x3_mfma_32x32x16(a, bq, z, 0, 0, 0);
        } else {
          sacc[qf][kf] = x3_mfma_32x32x16(a, bq, sacc[qf][kf], 0, 0, 0);
        }
      }
      if constexpr (!S_REG && DO_PV) {
        const x3x8 a = prod == 0 ? fl[bb][kf] : fh[bb][kf];   // kf = V^T channel fragment here
        const x3x8 bp = __builtin_bit_cast(x3x8, prod == 1 ? plw[bb][qf] : phw[bb][qf]);
        oacc[qf][kf] = x3_mfma_32x32x16(a, bp, oacc[qf][kf], 0, 0, 0);
      }
      // ---- one fragment read for the next region (gaps 4..7), into the buffer the previous region used ----
      if constexpr (m >= 4 && m < 8) {
        constexpr int f = m - 4, NR = R + 1;   // f: 0 hi frag 0, 1 lo frag 0, 2 hi frag 1, 3 lo frag 1
        constexpr int FO = (f >> 1) * 4096, PL = f & 1;
        if constexpr (NR < 4) {   // K step NR of tile t+1
          if constexpr (DO_S) {
            const x3x8 v = lds_read128<SB1 + PL * ARR_B + FO>(faddr[NR]);
            if constexpr (PL == 0) fh[bb ^ 1][f >> 1] = v; else fl[bb ^ 1][f >> 1] = v;
          }
        } else if constexpr (NR < 8) {   // V^T key block NR - 4 of tile t
          if constexpr (DO_PV) {
            const x3x8 v = lds_read128<SB0 + (2 + PL) * ARR_B + FO>(faddr[NR - 4]);
            if constexpr (PL == 0) fh[bb ^ 1][f >> 1] = v; else fl[bb ^ 1][f >> 1] = v;
          }
        }
      }
      if constexpr (m == 9 && DO_S) issue_piece(tn, STN, R);
      gap_items(gc, dosc, dopvc);
      SB();
    });
    if constexpr (DO_S) {
#pragma unroll
      for (int qf = 0; qf < QF; ++qf) {
        const unsigned u = __float_as_uint(mx[qf] * c2);   // c2 > 0: the maximum of the scaled scores
        const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        // p = exp2(s - m) only needs SOME common m per row that keeps p in range: m follows the row maximum when that grew by more
        // than 2^TAU (p <= 64 otherwise - harmless in fp32 and in the hi/lo split), so the accumulators are rescaled a few times per
        // row instead of once per tile
        const float mt = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        const float m_new = mt > m_run[qf] + TAU ? mt : m_run[qf];
        alpha[qf] = __builtin_amdgcn_exp2f(m_run[qf] - m_new);   // first tile: exp2(-inf) = 0 on zero accumulators
        m_run[qf] = m_new;
      }
    }
  };
  using T = std::true_type; using Fz = std::false_type;
  TRA();
  // iteration t issues tile t+RING-1 and needs tile t+1: RING-3 tiles stay in flight across its barrier
#pragma unroll
  for (int d = 0; d < RING - 2; ++d) issue_tile(min(d, ntile - 1), d);
  iteration(std::integral_constant<int, 0>{}, T{}, Fz{}, -1);
  for (int t = 0; t + RING < ntile; t += RING)   // ntile is a multiple of RING (launcher)
    static_for<0, RING>([&](auto st) {
      constexpr int S0 = decltype(st)::value;
      iteration(std::integral_constant<int, (S0 + 1) % RING>{}, T{}, T{}, t + S0);
    });
  static_for<0, RING>([&](auto st) {   // last RING tiles: the final one has no S(t+1)
    constexpr int S0 = decltype(st)::value;
    if constexpr (S0 < RING - 1) iteration(std::integral_constant<int, (S0 + 1) % RING>{}, T{}, T{}, ntile - RING + S0);
    else iteration(std::integral_constant<int, (S0 + 1) % RING>{}, Fz{}, T{}, ntile - 1);
  });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped tail prefetch
#undef IC
#undef SB

#pragma unroll
  for (int qf = 0; qf < QF; ++qf) {
    const float inv = 1.0f / (l_run[qf] + __shfl_xor(l_run[qf], 32));
    const int qq = qi + 32 * qf;
    float* op = p.o + ((size_t)b * L + qq) * p.ldo + h * DH;
#pragma unroll
    for (int df = 0; df < 2; ++df)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        f32x4 v;
        v[0] = oacc[qf][df][4 * c + 0] * inv; v[1] = oacc[qf][df][4 * c + 1] * inv;
        v[2] = oacc[qf][df][4 * c + 2] * inv; v[3] = oacc[qf][df][4 * c + 3] * inv;
        if (p.o_planes) {
          x3x4 h4 = __builtin_convertvector(v, x3x4);
          const f32x4 hf = __builtin_convertvector(h4, f32x4);
          x3x4 l4 = __builtin_convertvector(v - hf, x3x4);
          x3_t* pp = p.o_planes + ((size_t)b * L + qq) * C + h * DH + df * 32 + 8 * c + 4 * g;
          *reinterpret_cast<x3x4*>(pp) = h4;
          *reinterpret_cast<x3x4*>(pp + MC) = l4;
        } else {
          *reinterpret_cast<f32x4*>(op + df * 32 + 8 * c + 4 * g) = v;
        }
      }
  }
}

// Combine the key slices of the split form: out[q] = sum_s 2^(m_s - m) O_s[q] / sum_s 2^(m_s - m) l_s with m = max_s m_s (the online-softmax
// merge, in the exp2 domain the kernel works in).  One thread per (query, four channels); fp32 rows or hi/lo planes like the kernel's own store.
__global__ __launch_bounds__(256) void attn_merge_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml, int nsplit, int B, int H, int L,
                                                        float* __restrict__ o, int ldo, x3_t* __restrict__ o_planes) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;     // over B*H*L*16
  const size_t nrow = (size_t)B * H * L;
  if (i >= nrow * 16) return;
  const size_t row = i >> 4; const int c4 = (int)(i & 15) * 4;
  float m = -INFINITY;
  for (int s = 0; s < nsplit; ++s) m = fmaxf(m, part_ml[(s * nrow + row) * 2]);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}; float lsum = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const float w = exp2f(part_ml[(s * nrow + row) * 2] - m);
    lsum += w * part_ml[(s * nrow + row) * 2 + 1];
    acc += *reinterpret_cast<const f32x4*>(part_o + (s * nrow + row) * 64 + c4) * w;
  }
  const f32x4 v = acc * (1.0f / lsum);
  const int q = (int)(row % L); const size_t bh = row / L; const int h = (int)(bh % H); const size_t b = bh / H;
  const int C = H * 64;
  if (o_planes) {
    const size_t MC = (size_t)B * L * C;
    const x3x4 h4 = __builtin_convertvector(v, x3x4);
    const x3x4 l4 = __builtin_convertvector(v - __builtin_convertvector(h4, f32x4), x3x4);
    x3_t* pp = o_planes + (b * L + q) * C + h * 64 + c4;
    *reinterpret_cast<x3x4*>(pp) = h4;
    *reinterpret_cast<x3x4*>(pp + MC) = l4;
  } else {
    *reinterpret_cast<f32x4*>(o + (b * L + q) * ldo + h * 64 + c4) = v;
  }
}

// scratch floats the key-split form wants for (batch, n_heads, l): 0 = it would not split
size_t attention_bf3_split_floats(int batch, int n_heads, int l, int* nsplit_out) {
  int ns = 1;
  // few workgroups walking many key tiles (batch 1 / 2 at L = 1024: 32 / 64 workgroups x 16 tiles): four key slices per query tile
  if (l % 512 == 0 && (l / 128) * n_heads * batch * 4 <= num_cus()) ns = 4;
  if (nsplit_out) *nsplit_out = ns;
  return ns > 1 ? (size_t)ns * batch * n_heads * l * (64 + 2) : 0;
}

int launch_attention_bf3(const void* planes, float* o, int ldo, void* o_planes, int batch, int n_heads, int l, int form, hipStream_t stream,
                         float* scratch, size_t scratch_floats) {
  PF_REQUIRE(planes && (o || o_planes) && batch > 0 && n_heads > 0 && l > 0 && l % 128 == 0, "attention_bf3: L must be a positive multiple of 128");
  PF_REQUIRE(form == PF_OPT_AUTO || form == 0 || (form == 1 && l % 256 == 0), "attention_bf3: form must be auto, 0 (128-query) or 1 (256-query, L %% 256 == 0)");
  PF_REQUIRE((size_t)batch * l * n_heads * 64 * 2 * 2 < ((size_t)1 << 31), "attention_bf3: a plane pair must stay below 2 GiB (32-bit offsets of the direct-to-LDS loads)");
  // auto: the 256-query form where it still gives three quarters of the CUs a workgroup
  const bool wide = l % 256 == 0 && (form == PF_OPT_AUTO ? (l / 256) * n_heads * batch >= num_cus() * 3 / 4 : form == 1);
  AttnP3 p{static_cast<const x3_t*>(planes), o, ldo, static_cast<x3_t*>(o_planes), batch, n_heads, l, 0.125f,
           make_fastdiv(wide ? l / 256 : l / 128), make_fastdiv(n_heads), 1, nullptr, nullptr, make_fastdiv(1)};
  int ns = 1;
  const size_t want = wide ? 0 : attention_bf3_split_floats(batch, n_heads, l, &ns);
  const bool split = !wide && ns > 1 && scratch && scratch_floats >= want;
  // (an 8-wave / 256-query form - half the K/V^T tile traffic per query - was measured and lost: its waves run S / softmax / PV in
  // lockstep behind one barrier, so the matrix pipe idles during every softmax, while two independent 4-wave workgroups per CU drift
  // apart and fill each other's gaps)
  static std::atomic<uint64_t> done_n{0}, done_w{0};
  if (int rc = set_max_lds_once(reinterpret_cast<const void*>(attn_bf3_kernel<4, 2>), 2 * 32768, done_n)) return rc;
  if (int rc = set_max_lds_once(reinterpret_cast<const void*>(attn_bf3_wide_kernel<4>), 4 * 32768, done_w)) return rc;
  if (wide) hipLaunchKernelGGL((attn_bf3_wide_kernel<4>), dim3((l / 256) * n_heads * batch), dim3(256), 4 * 32768, stream, p);
  else if (split) {
    p.nsplit = ns; p.d_ns = make_fastdiv(ns);
    p.part_o = scratch; p.part_ml = scratch + (size_t)ns * batch * n_heads * l * 64;
    hipLaunchKernelGGL((attn_bf3_kernel<4, 2>), dim3((l / 128) * n_heads * batch * ns), dim3(256), 2 * 32768, stream, p);
    PF_CHECK_HIP(hipGetLastError());
    const size_t n16 = (size_t)batch * n_heads * l * 16;
    hipLaunchKernelGGL(attn_merge_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, stream, p.part_o, p.part_ml, ns, batch, n_heads, l, o, ldo,
                       static_cast<x3_t*>(o_planes));
  } else hipLaunchKernelGGL((attn_bf3_kernel<4, 2>), dim3((l / 128) * n_heads * batch), dim3(256), 2 * 32768, stream, p);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}

#ifdef PF_TRACE
extern "C" int pf_debug_trace_read_attn(unsigned long long* dst, int n) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_trace_attn), (size_t)n * 8) == hipSuccess ? 0 : -1;
}
#endif

}  // namespace pf
