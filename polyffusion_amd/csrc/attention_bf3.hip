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
// of the second product.  LDS rows are 128 B; the 16-byte slot index is XOR-swizzled with (row & 7) on the source
// address of the direct load and on the fragment read, so ds_read_b128 over 32 consecutive rows is <= 2-way.
#include "conv_common.h"   // lds_read128 / lgkm_wait / static_for

namespace pf {

#ifdef PF_TRACE
__device__ unsigned long long g_trace_attn[4096];
#define TRA() do { if (trace_on && tslot < 2040) { g_trace_attn[tslot++] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define TRA() do {} while (0)
#endif

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct AttnP3 {
  const __bf16* planes;   // [qh | ql | kh | kl | vth | vtl], each B*L*C elements
  float* o; int ldo; __bf16* o_planes;
  int B, H, L;
  float scale;
  FastDiv d_nqt, d_h;     // reciprocals of the query tiles per (batch, head) and of H (tile decode without emulated divisions)
};

// NWAVES waves x 32 queries per workgroup share every K / V^T tile; RING tile stages in LDS (32 KB each).
// <4, 2>: 128 queries, two workgroups per CU.  <8, 4>: 256 queries, one workgroup per CU - half the tile traffic and half
// the direct-to-LDS issue work per wave, three tiles of prefetch distance.
template <int NWAVES, int RING>
__global__ __launch_bounds__(NWAVES * 64, NWAVES == 4 ? 2 : 1) void attn_bf3_kernel(AttnP3 p) {
  constexpr int DH = 64, KT = 64, NT = NWAVES * 64, NP = 2048 / NT;   // NP direct-to-LDS pieces per thread and tile
  constexpr int STAGE = 4 * KT * DH;          // bf16 elements per stage: K hi, K lo, V^T hi, V^T lo (8 KB each)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* sm = reinterpret_cast<__bf16*>(smem_raw);   // [RING stages][4 arrays][64 rows][64]

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
  bf16x8 qh[4], ql[4];
  {
    const __bf16* qp = p.planes + ((size_t)b * L + qi) * C + h * DH + 8 * g;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qh[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
      ql[s] = *reinterpret_cast<const bf16x8*>(qp + MC + 16 * s);
    }
  }

  // direct-to-LDS tile loads: 2048 16-byte units per stage, NP per thread; unit u = tid + j*NT -> array u/512 (K hi, K lo,
  // V^T hi, V^T lo), row (u%512)/8, slot u%8
  const __bf16* kbase = p.planes + 2 * MC + (size_t)b * L * C + h * DH;           // + plane*MC + key*C + d
  const __bf16* vbase = p.planes + 4 * MC + ((size_t)b * p.H + h) * DH * L;       // + plane*MC + d*L + key
  auto issue_piece = [&](int t, int stage, int j) {
    const int u = tid + j * NT;
    const int arr = u >> 9, row = (u & 511) >> 3;
    const int src_slot = (u & 7) ^ (row & 7);     // swizzle on the SOURCE side; the LDS image stays lane-linear
    const __bf16* gp = (arr < 2) ? kbase + (size_t)(arr & 1) * MC + (size_t)(t * KT + row) * C + src_slot * 8
                                 : vbase + (size_t)(arr & 1) * MC + (size_t)row * L + t * KT + src_slot * 8;
    __bf16* lp = sm + stage * STAGE + (j * NT + wave * 64) * 8;   // wave-uniform; the hardware adds lane*16 B
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                     (__attribute__((address_space(3))) void*)lp, 16, 0, 0);
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

  const int ntile = L / KT;            // even: L is a multiple of 128
  const int r31 = lane & 31;
  // LDS byte address of this lane's fragment row inside an 8 KB array, one per K-step pair: row = r31 (+32 per fragment,
  // an immediate), 16-byte slot = (2 sp + g) ^ (row & 7).  K and V^T fragments share the formula (row = key or channel).
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) __bf16*)sm;
  unsigned faddr[4];
#pragma unroll
  for (int sp = 0; sp < 4; ++sp) faddr[sp] = lds0 + r31 * 128 + (((2 * sp + g) ^ (r31 & 7)) * 16);
  constexpr int STAGE_B = STAGE * 2, ARR_B = KT * DH * 2;
#define SB() __builtin_amdgcn_sched_barrier(0)

  // One tile = 4 K-step pairs of S^T = K.Q^T, the softmax, and 4 key blocks of O^T += V^T.P^T.  Each step issues the four
  // ds_read_b128 (hi/lo of two fragments) of the NEXT step before its own six MFMAs, which alternate between two
  // independent accumulators so the wave's in-order issue never stalls on an accumulator dependency; the next tile's eight
  // direct-to-LDS loads are spread over the S steps and the hi/lo split of the next key block's probabilities is
  // interleaved with the MFMAs of the current one.  LDS waits are counted by hand (inline-asm reads, conv_common.h).
  // One barrier per tile: it publishes tile t and frees the other stage for tile t+1.
  typedef float f32x8 __attribute__((ext_vector_type(8)));
  auto psplit = [&](const f32x16& sv, int half, bf16x8& ph, bf16x8& pl) {   // whole-vector conversions (packed cvt path)
    f32x8 v;
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = sv[half * 8 + q];
    ph = __builtin_convertvector(v, bf16x8);
    pl = __builtin_convertvector(v - __builtin_convertvector(ph, f32x8), bf16x8);
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
    const int tn = min(t + RING - 1, ntile - 1);
    constexpr int STN = (ST + RING - 1) % RING;

    // ---- S^T = K . Q^T ----
    f32x16 s[2];
#pragma unroll
    for (int kf = 0; kf < 2; ++kf)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kf][r] = 0.f;
    bf16x8 fh[2][2], fl[2][2];   // [buffer][fragment]
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
      s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl[b][0], qh[sp], s[0], 0, 0, 0);
      s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl[b][1], qh[sp], s[1], 0, 0, 0);
      s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[b][0], ql[sp], s[0], 0, 0, 0);
      s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[b][1], ql[sp], s[1], 0, 0, 0);
      s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[b][0], qh[sp], s[0], 0, 0, 0);
      s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[b][1], qh[sp], s[1], 0, 0, 0);
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
    bf16x8 ph[2], pl[2];
    psplit(s[0], 0, ph[0], pl[0]);
    static_for<0, 4>([&](auto jc) {
      constexpr int kb = decltype(jc)::value, b = kb & 1;
      SB();
      if constexpr (kb < 3) {
        fh[b ^ 1][0] = lds_read128<SB0 + 2 * ARR_B>(faddr[kb + 1]);          fl[b ^ 1][0] = lds_read128<SB0 + 3 * ARR_B>(faddr[kb + 1]);
        fh[b ^ 1][1] = lds_read128<SB0 + 2 * ARR_B + 4096>(faddr[kb + 1]);   fl[b ^ 1][1] = lds_read128<SB0 + 3 * ARR_B + 4096>(faddr[kb + 1]);
      }
      lgkm_wait<(kb < 3) ? 4 : 0>(); SB();
      oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl[b][0], ph[b], oacc[0], 0, 0, 0);
      oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl[b][1], ph[b], oacc[1], 0, 0, 0);
      if constexpr (kb < 3) psplit(s[(kb + 1) >> 1], (kb + 1) & 1, ph[b ^ 1], pl[b ^ 1]);   // next block's split, in the MFMA shadow
      oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[b][0], pl[b], oacc[0], 0, 0, 0);
      oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[b][1], pl[b], oacc[1], 0, 0, 0);
      oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[b][0], ph[b], oacc[0], 0, 0, 0);
      oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[b][1], ph[b], oacc[1], 0, 0, 0);
      SB();
    });
  };
  TRA();
#pragma unroll
  for (int d = 0; d < RING - 1; ++d) issue_tile(min(d, ntile - 1), d);
  for (int t = 0; t < ntile; t += RING)   // ntile is a multiple of RING (launcher)
    static_for<0, RING>([&](auto st) { tile_body(st, t + decltype(st)::value); });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped tail prefetch
#undef SB

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
        bf16x4 h4 = __builtin_convertvector(v, bf16x4);
        const f32x4 hf = __builtin_convertvector(h4, f32x4);
        bf16x4 l4 = __builtin_convertvector(v - hf, bf16x4);
        __bf16* pp = p.o_planes + ((size_t)b * L + qi) * C + h * DH + df * 32 + 8 * c + 4 * g;
        *reinterpret_cast<bf16x4*>(pp) = h4;
        *reinterpret_cast<bf16x4*>(pp + MC) = l4;
      } else {
        *reinterpret_cast<f32x4*>(op + df * 32 + 8 * c + 4 * g) = v;
      }
    }
}

int launch_attention_bf3(const void* planes, float* o, int ldo, void* o_planes, int batch, int n_heads, int l, hipStream_t stream) {
  PF_REQUIRE(planes && (o || o_planes) && batch > 0 && n_heads > 0 && l > 0 && l % 128 == 0, "attention_bf3: L must be a positive multiple of 128");
  AttnP3 p{static_cast<const __bf16*>(planes), o, ldo, static_cast<__bf16*>(o_planes), batch, n_heads, l, 0.125f,
           make_fastdiv(l / 128), make_fastdiv(n_heads)};
  // (an 8-wave / 256-query form - half the K/V^T tile traffic per query - was measured and lost: its waves run S / softmax / PV in
  // lockstep behind one barrier, so the matrix pipe idles during every softmax, while two independent 4-wave workgroups per CU drift
  // apart and fill each other's gaps)
  static bool done = false;
  if (!done) {
    PF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bf3_kernel<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 32768));
    done = true;
  }
  hipLaunchKernelGGL((attn_bf3_kernel<4, 2>), dim3((l / 128) * n_heads * batch), dim3(256), 2 * 32768, stream, p);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}

#ifdef PF_TRACE
extern "C" int pf_debug_trace_read_attn(unsigned long long* dst, int n) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_trace_attn), (size_t)n * 8) == hipSuccess ? 0 : -1;
}
#endif

}  // namespace pf
