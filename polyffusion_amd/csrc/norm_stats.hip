// norm_stats.hip - GroupNorm / LayerNorm statistics for gfx950.
//
// The normalisations themselves are never run as separate passes: the consumer GEMM/conv
// applies y = x*scale + shift (GroupNorm) or (x-mean)*rstd*gamma+beta (LayerNorm) while it
// stages its input tile (conv_mfma.hip).  These kernels only produce the statistics:
//   GroupNorm32 / SpatialTransformer.norm (unet.py:321-336, unet_attention.py:40-42):
//       per-(sample, channel) scale/shift from per-(sample, group) mean/var, on NHWC input
//       that may be the channel-concat of two tensors (the U-Net skip concat is never
//       materialised; a group may straddle the two sources, e.g. 256+128 channels).
//   LayerNorm (unet_attention.py:104-110): per-row mean / rstd.
// Both are HBM-bound single reads; per-thread partial sums are short fp32 runs, all
// cross-thread / cross-block accumulation is done in fp64 and is deterministic (no atomics).
#include "pf_internal.h"

namespace pf {

int gn_nsplit(int hw) {
  int n = hw / 256;
  if (n < 1) n = 1;
  if (n > 64) n = 64;
  return n;
}

size_t gn_scratch_bytes(int batch, int c, int hw) { return (size_t)batch * gn_nsplit(hw) * c * 2 * sizeof(float); }

// Statistics of a tensor whose producer did not emit them (the stem conv): same [B][splits][C][2] layout as the
// per-tile statistics written by the conv epilogues, so one finalize kernel serves both.
__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x0, int c0, const float* __restrict__ x1,
                                                         int c1, int hw, int nsplit, float* __restrict__ part) {
  __shared__ double red[256 * 8];
  const int C = c0 + c1, CQ = C / 4, PL = 256 / CQ;
  const int s = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int chunk = (hw + nsplit - 1) / nsplit;
  const int p0 = s * chunk, p1 = min(hw, p0 + chunk);
  const int pl = tid / CQ, cq = tid % CQ;
  float sm[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
  if (pl < PL) {
    const int ch = cq * 4;
    const float* src; int cs, co;
    if (ch < c0) { src = x0; cs = c0; co = ch; } else { src = x1; cs = c1; co = ch - c0; }
    for (int pix = p0 + pl; pix < p1; pix += PL) {
      const float4 v = *reinterpret_cast<const float4*>(src + ((size_t)b * hw + pix) * cs + co);
      sm[0] += v.x; sm[1] += v.y; sm[2] += v.z; sm[3] += v.w;
      sq[0] += v.x * v.x; sq[1] += v.y * v.y; sq[2] += v.z * v.z; sq[3] += v.w * v.w;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) { red[tid * 8 + i] = sm[i]; red[tid * 8 + 4 + i] = sq[i]; }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    double a = 0.0, q = 0.0;
    const int cqq = c >> 2, ci = c & 3;
    for (int l = 0; l < PL; ++l) { a += red[(l * CQ + cqq) * 8 + ci]; q += red[(l * CQ + cqq) * 8 + 4 + ci]; }
    float* dst = part + (((size_t)b * nsplit + s) * C + c) * 2;
    dst[0] = (float)a; dst[1] = (float)q;
  }
}

// scale/shift per (sample, channel) from per-tile (sum, sumsq) of one or two channel-concatenated producers.
// One 64-thread workgroup per (sample, group): the tile x channel partials of the group are summed in fp64
// (lanes stride the (tile, channel) pairs, then a shuffle tree) - deterministic, no pass over the tensor.
__global__ __launch_bounds__(64) void gn_finalize_tiles_kernel(const float* __restrict__ s0, int t0, int c0,
                                                               const float* __restrict__ s1, int t1, int c1, int hw, int groups,
                                                               float eps, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ scale,
                                                               float* __restrict__ shift, int bmod1) {
  const int g = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  const int b1 = bmod1 > 0 ? b % bmod1 : b;   // source 1 shared between the halves of a guidance batch (ConvP::x1_bmod)
  const int C = c0 + c1;
  const int gs = C / groups;
  double a = 0.0, q = 0.0;
  // the group's channels that live in source 0 / source 1 (a group may straddle the concat boundary); lanes stride the
  // flattened (tile, channel) pairs of each part, so the loads of one lane are few and independent even when a tensor has
  // only a handful of tiles (16x16 level: 4 tiles x 8 channels = one load per lane instead of a chain of eight)
  const int cb = g * gs, ce = cb + gs;
  const int n0 = max(0, min(ce, c0) - cb);            // channels [cb, cb + n0) from source 0
  for (int i = lane; i < n0 * t0; i += 64) {
    const int t = i / n0, cl = cb + i % n0;
    const float2 v = *reinterpret_cast<const float2*>(s0 + (((size_t)b * t0 + t) * c0 + cl) * 2);
    a += v.x; q += v.y;
  }
  const int n1 = gs - n0, cb1 = max(cb, c0) - c0;     // channels [cb1, cb1 + n1) of source 1
  for (int i = lane; i < n1 * t1; i += 64) {
    const int t = i / n1, cl = cb1 + i % n1;
    const float2 v = *reinterpret_cast<const float2*>(s1 + (((size_t)b1 * t1 + t) * c1 + cl) * 2);
    a += v.x; q += v.y;
  }
  for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); q += __shfl_xor(q, off); }
  const double cnt = (double)gs * hw;
  const double mean = a / cnt;
  double var = q / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const double rstd = 1.0 / sqrt(var + (double)eps);
  for (int i = lane; i < gs; i += 64) {
    const int c = g * gs + i;
    const double ga = gamma[c];
    scale[(size_t)b * C + c] = (float)(rstd * ga);
    shift[(size_t)b * C + c] = (float)((double)beta[c] - mean * rstd * ga);
  }
}

int launch_gn_finalize_tiles(const float* s0, int t0, int c0, const float* s1, int t1, int c1, int batch, int hw, int groups,
                             float eps, const float* gamma, const float* beta, float* scale, float* shift, hipStream_t stream, int bmod1) {
  const int C = c0 + c1;
  PF_REQUIRE(s0 && t0 > 0 && c0 > 0 && (c1 == 0 || (s1 && t1 > 0)), "gn_finalize: bad statistics inputs");
  PF_REQUIRE(groups > 0 && C % groups == 0, "gn: groups=%d unsupported for C=%d", groups, C);
  hipLaunchKernelGGL(gn_finalize_tiles_kernel, dim3(groups, batch), dim3(64), 0, stream, s0, t0, c0, s1, t1, c1, hw, groups, eps,
                     gamma, beta, scale, shift, bmod1);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}

int launch_gn_partial(const float* x0, int c0, const float* x1, int c1, int batch, int hw, float* stats, hipStream_t stream) {
  const int C = c0 + c1;
  PF_REQUIRE(x0 && c0 > 0 && c0 % 4 == 0 && c1 % 4 == 0 && (c1 == 0 || x1), "gn: bad inputs (c0=%d c1=%d)", c0, c1);
  PF_REQUIRE(C <= 1024, "gn: at most 1024 channels (got %d)", C);
  const int ns = gn_nsplit(hw);
  hipLaunchKernelGGL(gn_partial_kernel, dim3(ns, batch), dim3(256), 0, stream, x0, c0, x1, c1, hw, ns, stats);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}

int launch_gn_scale_shift(const float* x0, int c0, const float* x1, int c1, int batch, int hw, int groups, float eps,
                          const float* gamma, const float* beta, float* scale, float* shift, void* scratch,
                          size_t scratch_bytes, hipStream_t stream) {
  const int C = c0 + c1;
  PF_REQUIRE(scratch && scratch_bytes >= gn_scratch_bytes(batch, C, hw), "gn: scratch too small");
  int rc = launch_gn_partial(x0, c0, x1, c1, batch, hw, (float*)scratch, stream);
  if (rc) return rc;
  return launch_gn_finalize_tiles((const float*)scratch, gn_nsplit(hw), C, nullptr, 0, 0, batch, hw, groups, eps, gamma, beta, scale,
                                  shift, stream);
}

// One wave per row; two passes over the (L1/L2-resident) row: mean, then centred variance.
__global__ __launch_bounds__(256) void ln_stats_kernel(const float* __restrict__ x, int rows, int C, float eps,
                                                       float* __restrict__ mean, float* __restrict__ rstd) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (size_t)row * C;
  float s = 0.f;
  for (int k = lane * 4; k < C; k += 256) {
    const float4 v = *reinterpret_cast<const float4*>(xr + k);
    s += (v.x + v.y) + (v.z + v.w);
  }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  const float mu = s / (float)C;
  float q = 0.f;
  for (int k = lane * 4; k < C; k += 256) {
    const float4 v = *reinterpret_cast<const float4*>(xr + k);
    const float a = v.x - mu, b = v.y - mu, c = v.z - mu, d = v.w - mu;
    q += (a * a + b * b) + (c * c + d * d);
  }
  for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
  if (lane == 0) {
    mean[row] = mu;
    rstd[row] = 1.0f / sqrtf(q / (float)C + eps);
  }
}

// LayerNorm applied and split into the bf16 hi/lo planes a planes GEMM (gemm_planes_bf3.hip) streams straight into LDS:
// y = (x - mean) * rstd * gamma + beta, same operation order as the GEMM-prologue form.  One wave per row, the row stays
// in registers between the statistics and the output (C <= 1024), so x is read once: 4 B in, 4 B out per element.
typedef x3_t x3x4_n __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void ln_planes_kernel(const float* __restrict__ x, int rows, int C, float eps,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        x3_t* __restrict__ planes) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (size_t)row * C;
  f32x4 v[4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = lane * 4 + i * 256;
    if (k < C) { v[i] = *reinterpret_cast<const f32x4*>(xr + k); s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]); }
  }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  const float mu = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = lane * 4 + i * 256;
    if (k < C) {
      const float a = v[i][0] - mu, b = v[i][1] - mu, c = v[i][2] - mu, d = v[i][3] - mu;
      q += (a * a + b * b) + (c * c + d * d);
    }
  }
  for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
  const float rs = 1.0f / sqrtf(q / (float)C + eps);
  const size_t MC = (size_t)rows * C;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = lane * 4 + i * 256;
    if (k < C) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + k), be = *reinterpret_cast<const f32x4*>(beta + k);
      const f32x4 y = (v[i] - mu) * rs * g + be;
      const x3x4_n hi = __builtin_convertvector(y, x3x4_n);
      const x3x4_n lo = __builtin_convertvector(y - __builtin_convertvector(hi, f32x4), x3x4_n);
      *reinterpret_cast<x3x4_n*>(planes + (size_t)row * C + k) = hi;
      *reinterpret_cast<x3x4_n*>(planes + MC + (size_t)row * C + k) = lo;
    }
  }
}

int launch_ln_planes(const float* x, int rows, int c, float eps, const float* gamma, const float* beta, void* planes,
                     hipStream_t stream) {
  PF_REQUIRE(x && gamma && beta && planes && rows > 0 && c > 0 && c % 4 == 0 && c <= 1024, "ln_planes: bad arguments (c=%d)", c);
  hipLaunchKernelGGL(ln_planes_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, stream, x, rows, c, eps, gamma, beta,
                     static_cast<x3_t*>(planes));
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}

int launch_ln_stats(const float* x, int rows, int c, float eps, float* mean, float* rstd, hipStream_t stream) {
  PF_REQUIRE(x && mean && rstd && rows > 0 && c > 0 && c % 4 == 0, "ln_stats: bad arguments");
  hipLaunchKernelGGL(ln_stats_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, stream, x, rows, c, eps, mean, rstd);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}

}  // namespace pf
