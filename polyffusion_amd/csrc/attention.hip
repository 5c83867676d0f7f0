// attention.hip - flash-style softmax(q k^T / sqrt(d)) v with fp32 MFMA for gfx950.
//
// Replaces CrossAttention.normal_attention (unet_attention.py:261-293) without materialising
// the [B,heads,Lq,Lk] score tensor (16.8 MB/sample at L=1024 in the reference).
//
// Per workgroup: 128 queries of one (batch, head); 4 waves x 32 queries.  K/V are streamed in
// 64-key tiles through double-buffered LDS.  Both contractions are computed TRANSPOSED so that
// no operand ever changes layout:
//   S^T[key][q]  = K . Q^T      -> accumulator lane (q = lane&31) holds 16 keys of ITS query per 32-key
//                                  fragment; the row max/sum are in-lane + one lane^32 exchange;
//   O^T[d][q]   += V^T . P^T    -> P^T's accumulator registers ARE the MFMA B operands, V^T's A operand is
//                                  a conflict-free 4-byte LDS read, and the online-softmax rescale
//                                  factor (per query) is lane-local.
// Scores are scaled after the dot product exactly like the reference (attn = qk * scale).
#include "pf_internal.h"

namespace pf {

struct AttnP {
  const float* q; const float* k; const float* v; float* o;
  int ldq, ldk, ldv, ldo;
  int H, Lq, Lk;
  float scale;
};

template <int DH>
__global__ __launch_bounds__(256, 2) void attn_kernel(AttnP p) {
  constexpr int KT = 64;          // keys per tile
  constexpr int KP = DH + 4;      // K row pitch (floats)
  constexpr int VP = DH;          // V row pitch
  constexpr int DQ = DH / 8;      // 8-wide k-steps over d
  constexpr int DF = DH / 32;     // output d fragments
  constexpr int NK4 = KT * DH / 4 / 256;  // float4 per thread per K (or V) tile
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sK = smem;                    // [2][KT][KP]
  float* sV = smem + 2 * KT * KP;      // [2][KT][VP]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int g = lane >> 5;
  const int qi = qt * 128 + wave * 32 + (lane & 31);   // this lane's query
  const bool qvalid = qi < p.Lq;

  // Q fragment: lane (q, g) holds Q[q][8j+4g .. +3], j < DQ
  f32x4 qf[DQ];
  {
    const float* qp = p.q + ((size_t)b * p.Lq + (qvalid ? qi : 0)) * p.ldq + h * DH + 4 * g;
#pragma unroll
    for (int j = 0; j < DQ; ++j) qf[j] = qvalid ? *reinterpret_cast<const f32x4*>(qp + 8 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
  }

  f32x16 oacc[DF];
#pragma unroll
  for (int df = 0; df < DF; ++df)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[df][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const float* kbase = p.k + (size_t)b * p.Lk * p.ldk + h * DH;
  const float* vbase = p.v + (size_t)b * p.Lk * p.ldv + h * DH;
  f32x4 rk[NK4], rv[NK4];
  constexpr int RQ = DH / 4;  // float4 per row
  auto loadKV = [&](int t) {
#pragma unroll
    for (int i = 0; i < NK4; ++i) {
      const int u = tid + i * 256;
      const int row = u / RQ, c = u % RQ;
      const int key = t * KT + row;
      if (key < p.Lk) {
        rk[i] = *reinterpret_cast<const f32x4*>(kbase + (size_t)key * p.ldk + c * 4);
        rv[i] = *reinterpret_cast<const f32x4*>(vbase + (size_t)key * p.ldv + c * 4);
      } else {
        rk[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        rv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  auto storeKV = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NK4; ++i) {
      const int u = tid + i * 256;
      const int row = u / RQ, c = u % RQ;
      *reinterpret_cast<f32x4*>(sK + buf * (KT * KP) + row * KP + c * 4) = rk[i];
      *reinterpret_cast<f32x4*>(sV + buf * (KT * VP) + row * VP + c * 4) = rv[i];
    }
  };

  const int ntile = (p.Lk + KT - 1) / KT;
  loadKV(0);
  storeKV(0);
  __syncthreads();

  for (int t = 0; t < ntile; ++t) {
    if (t + 1 < ntile) loadKV(t + 1);
    const float* cK = sK + (t & 1) * (KT * KP);
    const float* cV = sV + (t & 1) * (KT * VP);

    // ---- S^T = K . Q^T : two 32-key fragments ----
    f32x16 s[2];
#pragma unroll
    for (int kf = 0; kf < 2; ++kf) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kf][r] = 0.f;
      const float* kr = cK + (kf * 32 + (lane & 31)) * KP + 4 * g;
#pragma unroll
      for (int j = 0; j < DQ; ++j) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(kr + 8 * j);
        s[kf] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], qf[j][0], s[kf], 0, 0, 0);
        s[kf] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], qf[j][1], s[kf], 0, 0, 0);
        s[kf] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], qf[j][2], s[kf], 0, 0, 0);
        s[kf] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], qf[j][3], s[kf], 0, 0, 0);
      }
    }
    // ---- online softmax over this tile's 64 keys (lane holds 32 of them, partner lane^32 the rest) ----
    float mx = -INFINITY;
#pragma unroll
    for (int kf = 0; kf < 2; ++kf)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = t * KT + kf * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
        float v = s[kf][r] * p.scale;
        v = (key < p.Lk) ? v : -INFINITY;
        s[kf][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);          // finite: every tile has >= 1 valid key
    const float alpha = __expf(m_run - m_new);     // exp(-inf) = 0 on the first tile
    float psum = 0.f;
#pragma unroll
    for (int kf = 0; kf < 2; ++kf)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __expf(s[kf][r] - m_new);
        s[kf][r] = pv;
        psum += pv;
      }
    l_run = l_run * alpha + psum;   // per-lane partial; partner halves are added at the end
    m_run = m_new;
#pragma unroll
    for (int df = 0; df < DF; ++df)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[df][r] *= alpha;

    // ---- O^T += V^T . P^T ----
#pragma unroll
    for (int kf = 0; kf < 2; ++kf)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int krow = kf * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
#pragma unroll
        for (int df = 0; df < DF; ++df) {
          const float a = cV[krow * VP + df * 32 + (lane & 31)];
          oacc[df] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s[kf][r], oacc[df], 0, 0, 0);
        }
      }

    if (t + 1 < ntile) storeKV((t + 1) & 1);
    __syncthreads();
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  if (qvalid) {
    float* op = p.o + ((size_t)b * p.Lq + qi) * p.ldo + h * DH;
#pragma unroll
    for (int df = 0; df < DF; ++df)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float4 v;
        v.x = oacc[df][4 * c + 0] * inv; v.y = oacc[df][4 * c + 1] * inv;
        v.z = oacc[df][4 * c + 2] * inv; v.w = oacc[df][4 * c + 3] * inv;
        *reinterpret_cast<float4*>(op + df * 32 + 8 * c + 4 * g) = v;
      }
  }
}

int launch_attention(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo,
                     int batch, int n_heads, int d_head, int lq, int lk, hipStream_t stream) {
  PF_REQUIRE(d_head == 32 || d_head == 64, "attention: d_head must be 32 or 64 (got %d)", d_head);
  PF_REQUIRE(lq > 0 && lk > 0 && batch > 0 && n_heads > 0, "attention: empty problem");
  PF_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0, "attention: row strides must be multiples of 4 floats");
  AttnP p{q, k, v, o, ldq, ldk, ldv, ldo, n_heads, lq, lk, 1.0f / sqrtf((float)d_head)};
  dim3 grid((lq + 127) / 128, n_heads, batch);
  if (d_head == 64) {
    constexpr size_t lds = (size_t)(2 * 64 * 68 + 2 * 64 * 64) * sizeof(float);
    static std::atomic<uint64_t> attr_done{0};
    if (int rc = set_max_lds_once(reinterpret_cast<const void*>(attn_kernel<64>), (int)lds, attr_done)) return rc;
    hipLaunchKernelGGL(attn_kernel<64>, grid, dim3(256), lds, stream, p);
  } else {
    constexpr size_t lds = (size_t)(2 * 64 * 36 + 2 * 64 * 32) * sizeof(float);
    hipLaunchKernelGGL(attn_kernel<32>, grid, dim3(256), lds, stream, p);
  }
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}

}  // namespace pf
