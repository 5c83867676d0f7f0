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

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct AttnP3 {
  const __bf16* planes;   // [qh | ql | kh | kl | vth | vtl], each B*L*C elements
  float* o; int ldo; __bf16* o_planes;
  int B, H, L;
  float scale;
};

__global__ __launch_bounds__(256, 2) void attn_bf3_kernel(AttnP3 p) {
  constexpr int DH = 64, KT = 64;
  constexpr int STAGE = 4 * KT * DH;          // bf16 elements per stage: K hi, K lo, V^T hi, V^T lo (8 KB each)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* sm = reinterpret_cast<__bf16*>(smem_raw);   // [2 stages][4 arrays][64 rows][64]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int g = lane >> 5;
  const int C = p.H * DH, L = p.L;
  const size_t MC = (size_t)p.B * L * C;
  const int qi = qt * 128 + wave * 32 + (lane & 31);

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

  // direct-to-LDS tile loads: 2048 16-byte units per stage, 8 per thread; unit u -> array u/512, row (u%512)/8, slot u%8
  const int urow = (tid >> 3), uslot = tid & 7;
  const __bf16* kbase = p.planes + 2 * MC + (size_t)b * L * C + h * DH;           // + plane*MC + key*C + d
  const __bf16* vbase = p.planes + 4 * MC + ((size_t)b * p.H + h) * DH * L;       // + plane*MC + d*L + key
  auto issue_tile = [&](int t, int stage) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int arr = j >> 1;                     // 0 K hi, 1 K lo, 2 V^T hi, 3 V^T lo
      const int row = urow + 32 * (j & 1);
      const int src_slot = uslot ^ (row & 7);     // swizzle on the SOURCE side; the LDS image stays lane-linear
      const __bf16* gp = (arr < 2) ? kbase + (size_t)(arr & 1) * MC + (size_t)(t * KT + row) * C + src_slot * 8
                                   : vbase + (size_t)(arr & 1) * MC + (size_t)row * L + t * KT + src_slot * 8;
      __bf16* lp = sm + stage * STAGE + arr * (KT * DH) + (wave * 64 + (j & 1) * 256) * 8;   // wave-uniform; HW adds lane*16 B
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                       (__attribute__((address_space(3))) void*)lp, 16, 0, 0);
    }
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

  // One tile = 16 "steps" of {two ds_read_b128 (hi, lo), three MFMAs}; the reads of step i+1 are issued before the MFMAs of
  // step i and waited for with a counted lgkmcnt (inline asm reads - see conv_common.h), so the matrix pipe never waits
  // for an LDS round trip.  One barrier per tile: it publishes tile t and frees the other stage for tile t+1.
  auto tile_body = [&](auto stc, int t) {
    constexpr int ST = decltype(stc)::value;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue_tile(min(t + 1, ntile - 1), ST ^ 1);

    // ---- S^T = K . Q^T : step i -> K-step pair sp = i >> 1, key fragment kf = i & 1 ----
    f32x16 s[2];
#pragma unroll
    for (int kf = 0; kf < 2; ++kf)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kf][r] = 0.f;
    bf16x8 fh[2], fl[2];
    fh[0] = lds_read128<ST * STAGE_B>(faddr[0]);
    fl[0] = lds_read128<ST * STAGE_B + ARR_B>(faddr[0]);
    static_for<0, 8>([&](auto ic) {
      constexpr int i = decltype(ic)::value, sp = i >> 1, kf = i & 1, b = i & 1;
      SB();
      if constexpr (i < 7) {
        constexpr int sp1 = (i + 1) >> 1, kf1 = (i + 1) & 1;
        fh[b ^ 1] = lds_read128<ST * STAGE_B + kf1 * 4096>(faddr[sp1]);
        fl[b ^ 1] = lds_read128<ST * STAGE_B + ARR_B + kf1 * 4096>(faddr[sp1]);
      } else {   // first V^T fragments, in flight across the softmax
        fh[b ^ 1] = lds_read128<ST * STAGE_B + 2 * ARR_B>(faddr[0]);
        fl[b ^ 1] = lds_read128<ST * STAGE_B + 3 * ARR_B>(faddr[0]);
      }
      lgkm_wait<2>(); SB();
      s[kf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl[b], qh[sp], s[kf], 0, 0, 0);
      s[kf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[b], ql[sp], s[kf], 0, 0, 0);
      s[kf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[b], qh[sp], s[kf], 0, 0, 0);
      SB();
    });
    // ---- online softmax (lane holds 32 keys of its query, the partner lane^32 the other 32) ----
    float mx = -INFINITY;
#pragma unroll
    for (int kf = 0; kf < 2; ++kf)
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[kf][r] *= p.scale; mx = fmaxf(mx, s[kf][r]); }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int kf = 0; kf < 2; ++kf)
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float pv = __expf(s[kf][r] - m_new); s[kf][r] = pv; psum += pv; }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int df = 0; df < 2; ++df)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[df][r] *= alpha;

    // ---- O^T += V^T . P^T : step j -> 16-key block kb = j >> 1, channel fragment df = j & 1; registers 8kb'..8kb'+7 of a
    // score fragment are this lane's 8 keys of block kb.  Step 0's fragments sit in buffer 0 (loaded by S step 7). ----
    bf16x8 ph, pl;
    static_for<0, 8>([&](auto jc) {
      constexpr int j = decltype(jc)::value, kb = j >> 1, df = j & 1, b = j & 1;
      SB();
      if constexpr (j < 7) {
        constexpr int kb1 = (j + 1) >> 1, df1 = (j + 1) & 1;
        fh[b ^ 1] = lds_read128<ST * STAGE_B + 2 * ARR_B + df1 * 4096>(faddr[kb1]);
        fl[b ^ 1] = lds_read128<ST * STAGE_B + 3 * ARR_B + df1 * 4096>(faddr[kb1]);
      }
      if constexpr (df == 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float v = s[kb >> 1][(kb & 1) * 8 + q];
          const __bf16 hi = (__bf16)v;
          ph[q] = hi;
          pl[q] = (__bf16)(v - (float)hi);
        }
      }
      lgkm_wait<(j < 7) ? 2 : 0>(); SB();
      oacc[df] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl[b], ph, oacc[df], 0, 0, 0);
      oacc[df] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[b], pl, oacc[df], 0, 0, 0);
      oacc[df] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[b], ph, oacc[df], 0, 0, 0);
      SB();
    });
  };
  issue_tile(0, 0);
  for (int t = 0; t < ntile; t += 2) {
    tile_body(std::integral_constant<int, 0>{}, t);
    tile_body(std::integral_constant<int, 1>{}, t + 1);
  }
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
  AttnP3 p{static_cast<const __bf16*>(planes), o, ldo, static_cast<__bf16*>(o_planes), batch, n_heads, l, 0.125f};
  constexpr size_t lds = (size_t)2 * 4 * 64 * 64 * 2;
  static bool done = false;
  if (!done) { PF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bf3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); done = true; }
  hipLaunchKernelGGL(attn_bf3_kernel, dim3(l / 128, n_heads, batch), dim3(256), lds, stream, p);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}

}  // namespace pf
