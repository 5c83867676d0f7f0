// preattn_fused_bf3.hip - the pre-attention half of a SpatialTransformer's first transformer layer as ONE launch (bf16x3 split MFMA):
//
//      y   = proj_in( GroupNorm(x) )                      ref:stable_diffusion/model/unet_attention.py:64-72  (norm, 1x1 proj_in, tokens = NHWC rows)
//      qkv = [to_q | to_k | to_v]( LayerNorm1(y) )        ref:unet_attention.py:240-243 attn1(norm1(x)), :150-166 the three projections
//
// Before: three launches per block - the GroupNorm-prologue 1x1 conv (24 us at L = 1024, B = 16: one 32-channel slab per barrier), ln_planes
// (y -> LayerNorm planes: 6 us) and the q|k|v planes GEMM (30 us) - with y making a round trip through HBM in between.  Here a four-wave
// workgroup owns 64 tokens end to end, in the fused feed-forward launch's (mlp_fused_bf3.hip) manner:
//   * the GroupNorm-normalised tile is written once into LDS as hi/lo planes (64 KB) and is the resident A operand of proj_in;
//   * proj_in's 64 x 256 result (+ bias) goes accumulators -> LDS (fp32, row-major) -> one wave per row: the fp32 row to HBM (the block's
//     residual stream, read again by attn1.to_out's epilogue) and its LayerNorm1 - ln_planes_kernel's arithmetic - back into the resident
//     planes, now the A operand of the three 256-column passes of the q|k|v projection;
//   * every weight matrix streams global -> LDS through one ring of 16 KB slots (a 16-deep K step x 256 columns, hi|lo planes:
//     12 MFMAs per wave per slot), fragments double-buffered per slot, hand-counted waits - the feed-forward launch's slot machinery;
//   * each q|k|v pass ends in conv_epilogue's plane-pair writers (Q, K [token][C]; V^T [head][d][token]) - the very code the planes GEMM
//     runs, staged through the ring region, so attention_bf3.hip reads the same layout.
// 1 MB of weights per workgroup, L2-resident (every workgroup of an XCD streams the same bytes).  One workgroup per CU (136 KB of LDS).
#include "conv_common.h"

namespace pf {

struct PreX {
  const x3_t* w_in; const float* b_in;   // proj_in: bf16x3 packing [32][plane][256][8], bias [256]
  float* y;                                 // proj_in output, fp32 [B*L][256]
  const float* gamma; const float* beta; float eps;   // LayerNorm1
  const x3_t* w_qkv;                      // to_q | to_k | to_v: bf16x3 packing [32][plane][768][8]
};

typedef x3_t x3x4_p __attribute__((ext_vector_type(4)));

// cycle stamps of workgroup 0 (tools/trace_pre.py builds a -DPF_PRE_TRACE copy; the buffer pointer travels in ConvP::mean, unused here)
#ifdef PF_PRE_TRACE
#define TRP() do { if (tr) { *tr++ = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define TRP() do {} while (0)
#endif

template <int RING>
__global__ __launch_bounds__(256, 1) void preattn_bf3_kernel(ConvP p, PreX e) {
  constexpr int BM = 64, C = 256, NSTEP = C / 16;
  constexpr int A_B = BM * C * 4;           // 64 KB: resident planes, [chunk 8][plane 2][row 64][64 B], 16-byte slots XOR-swizzled by row
  constexpr int SLOT_B = 16384;             // ring slot: [k8 2][plane 2][256 n][16 B]
  constexpr int CH_B = 2 * BM * 64;         // bytes of one 32-deep A chunk (both planes)
  constexpr int LO_B = BM * 64;             // offset of the lo plane inside a chunk
  static_assert(RING == 4, "slot schedule below: three slots of lead, positions = step % 4");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;
  unsigned char* sR = smem + A_B;                 // ring; together with the pad behind it also the 72 KB staging region of the exchanges
  unsigned char* sP = sR + RING * SLOT_B;         // 8 KB pad (GroupNorm finalize scratch; tail of the epilogue staging)

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

#ifdef PF_PRE_TRACE
  unsigned long long* tr = (blockIdx.x == 0 && threadIdx.x == 0) ? reinterpret_cast<unsigned long long*>(const_cast<float*>(p.mean)) : nullptr;
#endif
  TRP();
  // ---- weight stream: 16-deep step `kstep` of a [256][npad] matrix, columns ncol0 .. ncol0 + 255, piece q (1 KiB per wave) into ring position pos
  const __amdgpu_buffer_rsrc_t g_in = dma_resource(e.w_in), g_qkv = dma_resource(e.w_qkv);
  const int vn = tid * 16;
  auto issue = [&](const __amdgpu_buffer_rsrc_t& g, int npad, int ncol0, int kstep, int pos, int q) {
    dma16(g, vn, (((kstep * 2 + (q >> 1)) * 2 + (q & 1)) * npad + ncol0) * 16, sR + pos * SLOT_B + wave * 1024 + q * 4096);
  };
#pragma unroll
  for (int d = 0; d < RING - 1; ++d)
#pragma unroll
    for (int q = 0; q < 4; ++q) issue(g_in, C, 0, d, d, q);

  // ---- GroupNorm of the tile into the resident planes: y = x * scale[b][c] + shift[b][c] (prologue 2 of pf_conv2d), one wave per row ----
  constexpr int RPW = BM / 4;   // rows per wave
  f32x4 v[RPW];
#pragma unroll
  for (int i = 0; i < RPW; ++i) v[i] = *reinterpret_cast<const f32x4*>(p.x0 + (row0 + wave * RPW + i) * C + lane * 4);
  TRP();
  if (p.gn_s0) gn_fused_prologue<256>(p, b, tid, p.Hin * p.Win, reinterpret_cast<double*>(sP));   // finalize folded into this launch
  TRP();
  {
    const f32x4 sc = *reinterpret_cast<const f32x4*>(p.sc + (size_t)b * C + lane * 4);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(p.sh + (size_t)b * C + lane * 4);
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int row = wave * RPW + i;
      const f32x4 y = v[i] * sc + sh;
      const x3x4_p hi = __builtin_convertvector(y, x3x4_p);
      const x3x4_p lo = __builtin_convertvector(y - __builtin_convertvector(hi, f32x4), x3x4_p);
      // k = 4 lane: chunk = lane / 8, 16-byte slot = (lane % 8) / 2 (XOR-swizzled with the row), half = lane & 1
      unsigned char* d = sA + (lane >> 3) * CH_B + row * 64 + ((((lane & 7) >> 1) ^ ((row >> 2) & 3)) * 16) + (lane & 1) * 8;
      *reinterpret_cast<x3x4_p*>(d) = hi;
      *reinterpret_cast<x3x4_p*>(d + LO_B) = lo;
    }
  }

  TRP();
  // ---- fragment addresses (mlp_fused_bf3.hip's) ----
  const unsigned ldsA = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)sA;
  const unsigned ldsR = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)sR;
  const int arow = wm * 32 + (lane & 31);
  const int sw = ((lane & 31) >> 2) & 3;
  const int x0s = (sw & 2) | (((lane >> 5) ^ sw) & 1);
  const unsigned ab0 = ldsA + (arow * 4 + x0s) * 16, ab1 = ldsA + (arow * 4 + (x0s ^ 2)) * 16;      // K step 0 / 1 of a chunk
  const unsigned wb = ldsR + (((lane >> 5) * 2 * 256 + wn * 128 + (lane & 31)) * 16);               // slot: [k8 2][plane][256]

  f32x16 acc[1][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][i][r] = 0.f;
  x3x8 gal[2], gah[2], gbh[2][4], gbl[2][4];   // fragments, double-buffered per slot
#define SB() __builtin_amdgcn_sched_barrier(0)
#define IC(N) std::integral_constant<int, (N)>{}
  // n-th fragment read of step CC (ring position RP) into set S: n = 0..3 w_hi, 4..7 w_lo, 8 a_lo, 9 a_hi
  auto ld = [&](auto S_, auto CC_, auto RP_, auto N_) {
    constexpr int S = S_.value, CC = CC_.value, RP = RP_.value, n = N_.value;
    if constexpr (n < 4) gbh[S][n] = lds_read128<RP * SLOT_B + (n * 32) * 16>(wb);
    else if constexpr (n < 8) gbl[S][n - 4] = lds_read128<RP * SLOT_B + (256 + (n - 4) * 32) * 16>(wb);
    else if constexpr (n == 8) gal[S] = lds_read128<(CC >> 1) * CH_B + LO_B>((CC & 1) ? ab1 : ab0);
    else gah[S] = lds_read128<(CC >> 1) * CH_B>((CC & 1) ? ab1 : ab0);
  };
  // n-th MFMA of a slot: X X X X (a_lo.w_hi)  Z Z Z Z (a_hi.w_hi)  Y Y Y Y (a_hi.w_lo) over the wave's four column fragments
  auto mf = [&](auto S_, auto N_) {
    constexpr int S = S_.value, g = N_.value / 4, fn = N_.value % 4;
    if constexpr (g == 0) acc[0][fn] = x3_mfma_32x32x16(gal[S], gbh[S][fn], acc[0][fn], 0, 0, 0);
    else if constexpr (g == 1) acc[0][fn] = x3_mfma_32x32x16(gah[S], gbh[S][fn], acc[0][fn], 0, 0, 0);
    else acc[0][fn] = x3_mfma_32x32x16(gah[S], gbl[S][fn], acc[0][fn], 0, 0, 0);
  };
#define FRAGS_READY() do { SB(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); SB(); } while (0)
#define SLOT_SYNC() do { SB(); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); SB(); } while (0)

  // One GEMM pass: acc += A(resident planes) x W[:, ncol0 .. ncol0 + 255] in NSTEP slots.  On entry the first RING-1 slots of the pass
  // have been issued and nothing else is in flight; on exit nothing is in flight (the tail re-fetches of slot NSTEP-1 are drained).
  auto gemm_pass = [&](const __amdgpu_buffer_rsrc_t& g, int npad, int ncol0) {
    SB();
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((RING - 2) * 4) : "memory");   // slot 0 landed; this thread's plane stores are done
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    TRP();
    static_for<0, 10>([&](auto n) { ld(IC(0), IC(0), IC(0), n); });
    SB();
    static_for<0, NSTEP>([&](auto cc) {
      constexpr int c = cc.value, S = c & 1, NS = S ^ 1, rpn = (c + 1) % RING;
      FRAGS_READY();
      static_for<0, 12>([&](auto nn) {
        constexpr int n = nn.value, m = n - 2;
        mf(IC(S), nn); SB();
        if constexpr (n == 1) SLOT_SYNC();       // slot c+1 landed for everybody; the ring position of slot c-1 is free
        if constexpr (c + 1 < NSTEP && m >= 0 && m < 5) { ld(IC(NS), IC(c + 1), IC(rpn), IC(2 * m)); ld(IC(NS), IC(c + 1), IC(rpn), IC(2 * m + 1)); }
        // refill: slot c+3 (the last slot again once the pass has none left: keeps vmcnt(4) meaningful, nobody reads it)
        if constexpr (n >= 2 && n < 6) issue(g, npad, ncol0, c + 3 < NSTEP ? c + 3 : NSTEP - 1, (c + 3) % RING, n - 2);
        SB();
      });
    });
    SB();
    TRP();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    SB();
    TRP();
  };

  // ================= proj_in =================
  gemm_pass(g_in, C, 0);
  // small operands of the exchange, fetched while nothing else is in flight (a compiler-visible load behind the weight copies issued
  // below would make hipcc drain them before its first use)
  const int cq = (lane & 31) & ~3;
  f32x4 b4[4];
#pragma unroll
  for (int fn = 0; fn < 4; ++fn) b4[fn] = *reinterpret_cast<const f32x4*>(e.b_in + wn * 128 + fn * 32 + cq);
  f32x4 g = *reinterpret_cast<const f32x4*>(e.gamma + lane * 4), be = *reinterpret_cast<const f32x4*>(e.beta + lane * 4);
#pragma unroll
  for (int fn = 0; fn < 4; ++fn) asm volatile("" : "+v"(b4[fn]));
  asm volatile("" : "+v"(g), "+v"(be));
  __builtin_amdgcn_s_barrier();     // every wave is done with the planes and the ring
  asm volatile("" ::: "memory");
  // the first slots of q|k|v pass 0 start now: their latency (a first touch of w_qkv: HBM) hides behind the exchange below, which
  // therefore stages through the planes region - the ring is busy
#pragma unroll
  for (int d = 0; d < RING - 1; ++d)
#pragma unroll
    for (int q = 0; q < 4; ++q) issue(g_qkv, 3 * C, 0, d, d, q);
  {
    // y = acc + bias: accumulators -> LDS as fp32 rows (lane-quad transpose: a lane stores four consecutive channels of one token; the 16-byte
    // slot of a row is XOR-ed with (row & 3) << 2 so the 16 lanes of a store pass hit 16 different slots), then one wave per row.
    // Raw barriers with LDS-only waits: a __syncthreads() would also wait for the weight copies just issued.
    float* sY = reinterpret_cast<float*>(sA);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = wm * 32 + 8 * q + 4 * (lane >> 5) + (lane & 3);
#pragma unroll
      for (int fn = 0; fn < 4; ++fn) {
        f32x4 t = {acc[0][fn][4 * q], acc[0][fn][4 * q + 1], acc[0][fn][4 * q + 2], acc[0][fn][4 * q + 3]};
        quad_transpose(t, lane);
        t = PF_X3_UNSCALE(t) + b4[fn];
        const int slot = (wn * 128 + fn * 32 + cq) >> 2;
        *reinterpret_cast<f32x4*>(sY + row * C + ((slot ^ ((row & 3) << 2)) << 2)) = t;
      }
    }
#pragma unroll
    for (int fn = 0; fn < 4; ++fn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][fn][r] = 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    TRP();
    float red[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int row = wave * RPW + i;
      v[i] = *reinterpret_cast<const f32x4*>(sY + row * C + ((lane ^ ((row & 3) << 2)) << 2));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();     // every wave holds its rows: the region may become planes again
    asm volatile("" ::: "memory");
    TRP();
    // LayerNorm1: ln_planes_kernel's arithmetic (xor butterfly 32, 16, .., 1), the 16 rows' reductions advancing together
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
      const x3x4_p hi = __builtin_convertvector(y, x3x4_p);
      const x3x4_p lo = __builtin_convertvector(y - __builtin_convertvector(hi, f32x4), x3x4_p);
      unsigned char* d = sA + (lane >> 3) * CH_B + row * 64 + ((((lane & 7) >> 1) ^ ((row >> 2) & 3)) * 16) + (lane & 1) * 8;
      *reinterpret_cast<x3x4_p*>(d) = hi;
      *reinterpret_cast<x3x4_p*>(d + LO_B) = lo;
    }
    // (the pass's own opening wait + barrier publishes the planes; the fp32 rows v[] go to HBM at the very end of the kernel: stores
    // issued here would sit in the same counter as the weight copies and every slot hand-over would wait for the write burst)
    TRP();
  }

  // ================= q | k | v: three 256-column passes over the LayerNorm planes =================
  // Q and K thirds ([token][C] plane pairs) leave straight from the accumulators: lane-quad transpose, hi/lo split, 8-byte stores (eight
  // lanes cover 64 contiguous bytes of a row).  With one wave per SIMD every vector instruction is exposed, and conv_epilogue's staged
  // writer (64 LDS writes, two barriers, 8 x 60 instructions per lane) cost 6-7.6 k cycles per pass in the cycle stamps against 7.2 k for
  // the pass's MFMAs; it also needs the ring as staging, which kept the next pass's first slots from being fetched meanwhile.  Same values,
  // same conversions: the planes are bit-identical to the staged writer's.
  auto store_qk = [&](int which) {
    const size_t MC = (size_t)p.B * L * C;
    x3_t* ph = static_cast<x3_t*>(p.qkv) + (size_t)(2 * which) * MC;
    const int cq = (lane & 31) & ~3;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = wm * 32 + 8 * q + 4 * (lane >> 5) + (lane & 3);
      x3_t* pr = ph + (row0 + row) * C + wn * 128 + cq;
#pragma unroll
      for (int fn = 0; fn < 4; ++fn) {
        f32x4 t = {acc[0][fn][4 * q], acc[0][fn][4 * q + 1], acc[0][fn][4 * q + 2], acc[0][fn][4 * q + 3]};
        quad_transpose(t, lane);
        t = PF_X3_UNSCALE(t);
        const x3x4_p hi = __builtin_convertvector(t, x3x4_p);
        const x3x4_p lo = __builtin_convertvector(t - __builtin_convertvector(hi, f32x4), x3x4_p);
        *reinterpret_cast<x3x4_p*>(pr + fn * 32) = hi;
        *reinterpret_cast<x3x4_p*>(pr + MC + fn * 32) = lo;
      }
    }
#pragma unroll
    for (int fn = 0; fn < 4; ++fn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][fn][r] = 0.f;
  };
  auto issue_first = [&](int pass) {   // every wave is past the last slot hand-over of the pass before: no ring position is being read
#pragma unroll
    for (int d = 0; d < RING - 1; ++d)
#pragma unroll
      for (int q = 0; q < 4; ++q) issue(g_qkv, 3 * C, pass * C, d, d, q);
  };
  // The stores issued between a pass's first slots and its hand-overs sit in the same counter as the weight copies.  The waits stay
  // correct - loads complete in order among themselves, so "at most N operations outstanding" still implies "at most the last N copies
  // outstanding" - they are only stricter until the stores have drained, which the epilogue's own instructions mostly cover.
  gemm_pass(g_qkv, 3 * C, 0);                           // Q (its first slots were issued before the exchange)
  issue_first(1);
  // proj_in's output rows (the block's residual stream): here rather than in the exchange, whose stores the hand-overs of this pass
  // would have waited for (the write burst of 256 workgroups in lockstep), and not later - they cost the V^T writer its registers
#pragma unroll
  for (int i = 0; i < RPW; ++i) *reinterpret_cast<f32x4*>(e.y + (row0 + wave * RPW + i) * C + lane * 4) = v[i];
  store_qk(0);
  TRP();
  gemm_pass(g_qkv, 3 * C, C);                           // K
  issue_first(2);
  store_qk(1);
  TRP();
  gemm_pass(g_qkv, 3 * C, 2 * C);                       // V: V^T [head][d][token] runs along the tokens - conv_epilogue's transposing
  // writer, staged through the planes + ring regions (both idle now: 128 KB for its 69.6 KB)
  x3_unscale(acc);
  conv_epilogue<1, BM, C, 1, 4, 2>(p, acc, b, 0, ox0, 2 * C, wm, wn, lane, tid, reinterpret_cast<float*>(sA));
  TRP();
#undef FRAGS_READY
#undef SLOT_SYNC
#undef IC
#undef SB
}

// x: block input fp32 [batch*l][256]; sc / sh: GroupNorm scale / shift rows [batch][256] (written by this launch when the finalize is
// folded in: gn_stats != nullptr, the producer's per-tile statistics [batch][gn_tiles][256][2]); w_in / w_qkv: bf16x3 packings of proj_in
// ([256][256]) and of the concatenated to_q | to_k | to_v ([768][256]); y: proj_in output fp32 [batch*l][256]; qkv_planes: as pf_conv2d's.
int launch_preattn_fused(const float* x, int batch, int l, float* sc, float* sh, const float* gn_stats, int gn_tiles, const float* gn_gamma,
                         const float* gn_beta, float gn_eps, const void* w_in, const float* b_in, float* y, const float* ln_gamma,
                         const float* ln_beta, float ln_eps, const void* w_qkv, void* qkv_planes, hipStream_t stream) {
  PF_REQUIRE(x && sc && sh && w_in && b_in && y && ln_gamma && ln_beta && w_qkv && qkv_planes, "preattn_fused: null argument");
  PF_REQUIRE(batch > 0 && l > 0 && l % 64 == 0, "preattn_fused: rows per sample must be a multiple of 64 (got %d)", l);
  PF_REQUIRE(!gn_stats || (gn_tiles > 0 && gn_gamma && gn_beta), "preattn_fused: folded GroupNorm finalize needs tiles, gamma and beta");
  // 32-bit offsets of the plane-pair writers and the row stores
  PF_REQUIRE((size_t)batch * l * 256 * 4 < ((size_t)1 << 31), "preattn_fused: tensor too large for 32-bit offsets (batch %d, l %d)", batch, l);
  constexpr int RING = 4;
  ConvP p;
  memset(&p, 0, sizeof p);
  p.x0 = x; p.c0 = 256; p.B = batch; p.Hin = 1; p.Win = l; p.Hout = 1; p.Wout = l;
  p.w = w_qkv; p.N = 768; p.Npad = 768; p.ld_out = 768;
  p.sc = sc; p.sh = sh;
  if (gn_stats) { p.gn_s0 = gn_stats; p.gn_t0 = gn_tiles; p.gn_gamma = gn_gamma; p.gn_beta = gn_beta; p.gn_eps = gn_eps; p.gn_groups = 32; }
  p.qkv = qkv_planes;
  p.ksplit = 1;
  p.tiles_x = l / 64; p.tiles_y = 1; p.nt = 1;
  conv_fill_divs(p);
#ifdef PF_PRE_TRACE
  if (const char* tp = getenv("PF_TRACE_PTR")) p.mean = reinterpret_cast<const float*>(strtoull(tp, nullptr, 16));
#endif
  PreX e{static_cast<const x3_t*>(w_in), b_in, y, ln_gamma, ln_beta, ln_eps, static_cast<const x3_t*>(w_qkv)};
  constexpr size_t lds = 65536 + RING * 16384 + 8192;
  static_assert(lds <= 160 * 1024 && RING * 16384 + 8192 >= 256 * (64 + 4) * 4 && RING * 16384 + 8192 >= 64 * (256 + 8) * 4, "LDS budget / epilogue staging");
  static std::atomic<uint64_t> done{0};
  if (int rc = set_max_lds_once(reinterpret_cast<const void*>(preattn_bf3_kernel<RING>), (int)lds, done)) return rc;
  hipLaunchKernelGGL((preattn_bf3_kernel<RING>), dim3(batch * (l / 64)), dim3(256), lds, stream, p, e);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}

}  // namespace pf
