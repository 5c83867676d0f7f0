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
#include "pf_internal.h"

namespace pf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
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

  const int ntile = L / KT;
  issue_tile(0, 0);
  const int r31 = lane & 31;
  for (int t = 0; t < ntile; ++t) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // every wave is done with the stage refilled next
    if (t + 1 < ntile) {
      issue_tile(t + 1, (t + 1) & 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // this thread's part of tile t has landed (8 newer loads in flight)
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                       // tile t visible to all waves
    const __bf16* st = sm + (t & 1) * STAGE;

    // ---- S^T = K . Q^T ----
    f32x16 s[2];
#pragma unroll
    for (int kf = 0; kf < 2; ++kf) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kf][r] = 0.f;
      const int key = kf * 32 + r31;
#pragma unroll
      for (int sp = 0; sp < 4; ++sp) {
        const int unit = key * 8 + ((2 * sp + g) ^ (key & 7));
        const bf16x8 kh = *reinterpret_cast<const bf16x8*>(st + unit * 8);
        const bf16x8 kl = *reinterpret_cast<const bf16x8*>(st + KT * DH + unit * 8);
        s[kf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh[sp], s[kf], 0, 0, 0);
        s[kf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[sp], s[kf], 0, 0, 0);
        s[kf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[sp], s[kf], 0, 0, 0);
      }
    }
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

    // ---- O^T += V^T . P^T : registers 8kb'..8kb'+7 of a fragment are this lane's 8 keys of 16-key block kb ----
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      bf16x8 ph, pl;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = s[kb >> 1][(kb & 1) * 8 + j];
        const __bf16 hi = (__bf16)v;
        ph[j] = hi;
        pl[j] = (__bf16)(v - (float)hi);
      }
#pragma unroll
      for (int df = 0; df < 2; ++df) {
        const int d = df * 32 + r31;
        const int unit = d * 8 + ((2 * kb + g) ^ (d & 7));
        const bf16x8 vh = *reinterpret_cast<const bf16x8*>(st + 2 * KT * DH + unit * 8);
        const bf16x8 vl = *reinterpret_cast<const bf16x8*>(st + 3 * KT * DH + unit * 8);
        oacc[df] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph, oacc[df], 0, 0, 0);
        oacc[df] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl, oacc[df], 0, 0, 0);
        oacc[df] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph, oacc[df], 0, 0, 0);
      }
    }
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
