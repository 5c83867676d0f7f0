// small_kernels.hip - the memory-bound / tiny kernels of the path (gfx950):
//   stem and head convolutions (NCHW <-> NHWC at the ABI edge, unet.py:78-80,145-149),
//   timestep embedding MLP (unet.py:63-68,151-169), batched mat-vec (ResBlock emb_layers
//   unet.py:286-289; n_cond==1 cross-attention collapse unet_attention.py:186-212),
//   fused sampler updates (sampler_sdf.py:80-171,307-341; sampler_ddim.py:220-272,355-359),
//   counter-based Gaussian noise, GRU gate update and the texture-encoder front end
//   (dl_modules/chord_enc.py:15-22, txt_enc.py:23-27).
#include "pf_internal.h"

namespace pf {

int num_cus() {
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  int v = cache[dev & 63].load(std::memory_order_relaxed);
  if (v == 0) {
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    cache[dev & 63].store(v, std::memory_order_relaxed);
  }
  return v;
}

__device__ __forceinline__ float silu_s(float v) { return v / (1.0f + __expf(-v)); }

// ------------------------------------------------------------------ stem conv (Cin tiny)
// thread = (pixel, 4 output channels); x NCHW, out NHWC. Weights [Cout][Cin][3][3] read through LDS.
__global__ __launch_bounds__(256) void conv_in_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ out, int B, int Cin,
                                                      int Cout, int H, int W) {
  extern __shared__ float sw[];  // [Cout][Cin*9]
  const int K = Cin * 9;
  for (int i = threadIdx.x; i < Cout * K; i += 256) sw[i] = w[i];
  __syncthreads();
  const int CQ = Cout / 4;
  const size_t total = (size_t)B * H * W * CQ;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int cq = (int)(idx % CQ);
    const size_t pix = idx / CQ;
    const int xw = (int)(pix % W), yh = (int)((pix / W) % H), b = (int)(pix / ((size_t)W * H));
    float acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = bias[cq * 4 + j];
    for (int ci = 0; ci < Cin; ++ci) {
      const float* xp = x + ((size_t)b * Cin + ci) * H * W;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int iy = yh + r - 1;
        if (iy < 0 || iy >= H) continue;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const int ix = xw + s - 1;
          if (ix < 0 || ix >= W) continue;
          const float v = xp[(size_t)iy * W + ix];
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = fmaf(v, sw[(cq * 4 + j) * K + ci * 9 + r * 3 + s], acc[j]);
        }
      }
    }
    *reinterpret_cast<float4*>(out + pix * Cout + cq * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  }
}

// Register-weight variant: a workgroup owns a 16x16 pixel tile, stages its 18x18xCIN input halo in LDS once, and every
// thread keeps the 4 x CIN x 9 weights of ONE output-channel quad in registers while it walks the 16 pixels of a tile
// column.  A pixel's Cout floats are written by 16 adjacent lanes as one contiguous NHWC run (16 bytes per lane).
template <int CIN>
__global__ __launch_bounds__(256) void conv_in_reg_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ out, int B,
                                                          int H, int W, float* __restrict__ stats) {
  constexpr int K = CIN * 9, T = 16, TI = T + 2, COUT = 64;
  __shared__ float sx[CIN][TI][TI + 1];
  __shared__ float sred[16][16][8];   // [pixel column][channel quad][4 sums | 4 sums of squares] (statistics only)
  const int tid = threadIdx.x;
  int bid = blockIdx.x;
  const int tilesx = (W + T - 1) / T, tilesy = (H + T - 1) / T;
  const int tx = bid % tilesx; bid /= tilesx;
  const int ty = bid % tilesy;
  const int b = bid / tilesy;
  for (int u = tid; u < CIN * TI * TI; u += 256) {
    const int ci = u / (TI * TI), r = (u / TI) % TI, c = u % TI;
    const int iy = ty * T + r - 1, ix = tx * T + c - 1;
    sx[ci][r][c] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? x[((size_t)b * CIN + ci) * H * W + (size_t)iy * W + ix] : 0.f;
  }
  const int cq = tid & 15, px = tid >> 4;
  float wr[4][K], bj[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    bj[j] = bias[cq * 4 + j];
#pragma unroll
    for (int k = 0; k < K; ++k) wr[j][k] = w[(size_t)(cq * 4 + j) * K + k];
  }
  __syncthreads();
  const int ox = tx * T + px;
  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
  if (ox < W) {
#pragma unroll 4
  for (int py = 0; py < T; ++py) {
    const int oy = ty * T + py;
    if (oy >= H) break;
    float acc[4] = {bj[0], bj[1], bj[2], bj[3]};
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s2 = 0; s2 < 3; ++s2) {
          const float v = sx[ci][py + r][px + s2];
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = fmaf(v, wr[j][ci * 9 + r * 3 + s2], acc[j]);   // order: ci, then tap
        }
    *reinterpret_cast<float4*>(out + (((size_t)b * H + oy) * W + ox) * COUT + cq * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) { ssum[j] += acc[j]; ssq[j] += acc[j] * acc[j]; }
  }
  }
  // per-tile channel (sum, sumsq) of what was stored, in the layout of the conv epilogues' statistics ([B][tiles][64][2]): the
  // GroupNorm of the first ResBlock then needs no pass over the stem output (deterministic: fixed reduction order)
  if (stats) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { sred[px][cq][j] = ssum[j]; sred[px][cq][4 + j] = ssq[j]; }
    __syncthreads();
    if (tid < COUT) {
      const int q = tid >> 2, j = tid & 3;
      float a = 0.f, qq = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) { a += sred[c][q][j]; qq += sred[c][q][4 + j]; }
      float* dst = stats + (((size_t)b * (tilesx * tilesy) + ty * tilesx + tx) * COUT + tid) * 2;
      dst[0] = a; dst[1] = qq;
    }
  }
}

int launch_conv_in_stats_tiles(int cin, int cout, int h, int w_) { return (cin <= 2 && cout == 64) ? cdiv(h, 16) * cdiv(w_, 16) : 0; }

int launch_conv_in(const float* x, const float* w, const float* bias, float* out, int batch, int cin, int cout, int h, int w_,
                   hipStream_t stream, float* stats) {
  PF_REQUIRE(!stats || launch_conv_in_stats_tiles(cin, cout, h, w_) > 0, "conv_in: statistics only from the register-weight kernel (cin <= 2, cout == 64)");
  PF_REQUIRE(cout % 4 == 0 && (size_t)cout * cin * 9 * 4 <= 64 * 1024, "conv_in: unsupported channel counts %d->%d", cin, cout);
  const size_t total = (size_t)batch * h * w_ * (cout / 4);
  if (cin <= 2 && cout == 64) {
    const int grid = batch * cdiv(h, 16) * cdiv(w_, 16);
    if (cin == 1) hipLaunchKernelGGL(conv_in_reg_kernel<1>, dim3(grid), dim3(256), 0, stream, x, w, bias, out, batch, h, w_, stats);
    else hipLaunchKernelGGL(conv_in_reg_kernel<2>, dim3(grid), dim3(256), 0, stream, x, w, bias, out, batch, h, w_, stats);
    PF_CHECK_HIP(hipGetLastError());
    return PF_OK;
  }
  const int grid = (int)min((size_t)4096, (total + 255) / 256);
  hipLaunchKernelGGL(conv_in_kernel, dim3(grid), dim3(256), (size_t)cout * cin * 9 * sizeof(float), stream, x, w, bias, out, batch,
                     cin, cout, h, w_);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}

// ------------------------------------------------------------------ head: GN+SiLU -> conv3x3 -> few channels, NCHW out
// HBM-bound by construction (B x H x W x Cin floats in, a few channels out: 67 MB at the bench shape, 0.6 GFLOP), so the job is to keep the
// memory pipe busy: 16x16 output pixels per block, the normalised + activated 18x18 halo staged in LDS 16 channels at a time, and
//   * the NEXT chunk's global loads are issued (into registers) before the current chunk's arithmetic, so every block overlaps its own
//     memory latency instead of leaving that to its neighbours on the CU;
//   * SiLU with the hardware reciprocal (silu_f of conv_common.h) - an IEEE division per element costs as much as the rest of the transform;
//   * weights packed [9][Cin][Cout]: a tap's Cout weights of one input channel are adjacent, wave-uniform addresses -> scalar loads, and for
//     two / four output channels one v_pk_fma_f32 feeds two accumulators (half the VALU instructions of the dot products).
typedef float f32x2_s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float silu_fast(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

template <int COUT>
__global__ __launch_bounds__(256) void conv_out_kernel(const float* __restrict__ x, const float* __restrict__ sc,
                                                       const float* __restrict__ sh, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ out, int B, int Cin,
                                                       int H, int W) {
  constexpr int T = 16, TI = T + 2, CK = 16, CP = CK + 4, NPIECE = TI * TI * (CK / 4), NP = (NPIECE + 255) / 256;
  constexpr int NPAIR = COUT / 2, ODD = COUT & 1;
  __shared__ __attribute__((aligned(16))) float sx[TI * TI * CP];
  const int tid = threadIdx.x;
  const int tilesx = (W + T - 1) / T, tilesy = (H + T - 1) / T;
  int bid = blockIdx.x;
  const int tx = bid % tilesx; bid /= tilesx;
  const int ty = bid % tilesy;
  const int b = bid / tilesy;
  const int py = tid / T, px = tid % T;
  // this thread's pieces of a chunk (a piece = four channels of one halo pixel): fixed across the chunks.  256 % 4 == 0, so the
  // channel quad (tid & 3) - and with it the thread's scale / shift vectors - is the same for all of them.
  const int c4 = tid & 3;
  // Loads are unconditional (padding / surplus pieces read the tile's first valid element and are zeroed by a select when staged):
  // no exec-mask branches between them, all NP loads of a chunk in flight together.
  unsigned goff[NP];
  int soff[NP];
  float keep[NP];                          // 1: a pixel inside the image, 0: conv padding (zeros of the ACTIVATED tensor)
  const unsigned gsafe = (unsigned)(((b * H + min(ty * T, H - 1)) * W + min(tx * T, W - 1)) * Cin) + c4 * 4;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int u = tid + i * 256, pix = u >> 2;
    const int iy = ty * T + pix / TI - 1, ix = tx * T + pix % TI - 1;
    const bool in = u < NPIECE && iy >= 0 && iy < H && ix >= 0 && ix < W;
    soff[i] = u < NPIECE ? pix * CP + c4 * 4 : -1;
    goff[i] = in ? (unsigned)(((b * H + iy) * W + ix) * Cin) + c4 * 4 : gsafe;
    keep[i] = in ? 1.f : 0.f;
  }
  f32x4 r[NP], a4, d4;
  auto load = [&](int c0) {
#pragma unroll
    for (int i = 0; i < NP; ++i) r[i] = *reinterpret_cast<const f32x4*>(x + (size_t)goff[i] + c0);
    a4 = *reinterpret_cast<const f32x4*>(sc + (size_t)b * Cin + c0 + c4 * 4);
    d4 = *reinterpret_cast<const f32x4*>(sh + (size_t)b * Cin + c0 + c4 * 4);
  };
  auto stage = [&]() {   // normalise + SiLU
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      f32x4 v = r[i] * a4 + d4;
      v[0] = silu_fast(v[0]) * keep[i]; v[1] = silu_fast(v[1]) * keep[i]; v[2] = silu_fast(v[2]) * keep[i]; v[3] = silu_fast(v[3]) * keep[i];
      if (soff[i] >= 0) *reinterpret_cast<f32x4*>(sx + soff[i]) = v;
    }
  };
  f32x2_s acc2[NPAIR > 0 ? NPAIR : 1];
  float acc1 = 0.f;
#pragma unroll
  for (int q = 0; q < NPAIR; ++q) acc2[q] = f32x2_s{0.f, 0.f};
  load(0);
  stage();
  __syncthreads();
  for (int c0 = 0; c0 < Cin; c0 += CK) {
    const bool more = c0 + CK < Cin;
    if (more) load(c0 + CK);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float* xp = sx + ((py + t / 3) * TI + px + t % 3) * CP;
      const float* wt = w + (size_t)__builtin_amdgcn_readfirstlane((t * Cin + c0) * COUT);   // [tap][Cin][COUT]: uniform address -> scalar loads
#pragma unroll
      for (int k4 = 0; k4 < CK / 4; ++k4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xp + k4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float* we = wt + (k4 * 4 + e) * COUT;
#pragma unroll
          for (int q = 0; q < NPAIR; ++q)
            acc2[q] = __builtin_elementwise_fma(f32x2_s{v[e], v[e]}, f32x2_s{we[2 * q], we[2 * q + 1]}, acc2[q]);
          if (ODD) acc1 = fmaf(v[e], we[COUT - 1], acc1);
        }
      }
    }
    __syncthreads();            // every thread has finished reading this chunk's image
    if (more) {
      stage();
      __syncthreads();
    }
  }
  const int oy = ty * T + py, ox = tx * T + px;
  if (oy < H && ox < W) {
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
      const float a = (ODD && co == COUT - 1) ? acc1 : acc2[co / 2][co & 1];
      out[(((size_t)b * COUT + co) * H + oy) * W + ox] = a + bias[co];
    }
  }
}

int launch_conv_out(const float* x, const float* sc, const float* sh, const float* w, const float* bias, float* out, int batch,
                    int cin, int cout, int h, int w_, hipStream_t stream) {
  PF_REQUIRE(cout >= 1 && cout <= 4 && cin % 16 == 0, "conv_out: unsupported channel counts %d->%d", cin, cout);
  PF_REQUIRE((size_t)batch * h * w_ * cin < ((size_t)1 << 31), "conv_out: tensor too large for 32-bit element offsets");
  const int grid = batch * cdiv(h, 16) * cdiv(w_, 16);
  switch (cout) {
    case 1: hipLaunchKernelGGL(conv_out_kernel<1>, dim3(grid), dim3(256), 0, stream, x, sc, sh, w, bias, out, batch, cin, h, w_); break;
    case 2: hipLaunchKernelGGL(conv_out_kernel<2>, dim3(grid), dim3(256), 0, stream, x, sc, sh, w, bias, out, batch, cin, h, w_); break;
    case 3: hipLaunchKernelGGL(conv_out_kernel<3>, dim3(grid), dim3(256), 0, stream, x, sc, sh, w, bias, out, batch, cin, h, w_); break;
    default: hipLaunchKernelGGL(conv_out_kernel<4>, dim3(grid), dim3(256), 0, stream, x, sc, sh, w, bias, out, batch, cin, h, w_); break;
  }
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}

// ------------------------------------------------------------------ output step: onset/sustain image -> note durations
// utils.py:240-269 / 433-476 of the reference: a note starts where round(onset) > 0 and lasts while round(sustain) > 0 on
// the following steps of the same key.  Python's round is half-to-even on these float32 values, so the predicate is
// exactly v > 0.5 (custom_round: 0.95 < v < 1.05, onset only).  One thread per (image, key) column walks its S steps
// backwards keeping the length of the sustain run that starts at the next step: a single coalesced pass over the image
// (lanes = consecutive keys), 4 B read per input element, 4 B written per output element.
__global__ __launch_bounds__(128) void prmat2c_durations_kernel(const float* __restrict__ x, int S, int custom_round,
                                                                int32_t* __restrict__ dur) {
  const int n = blockIdx.x, key = threadIdx.x;
  const float* on = x + ((size_t)n * 2 + 0) * S * 128 + key;
  const float* su = x + ((size_t)n * 2 + 1) * S * 128 + key;
  int32_t* d = dur + (size_t)n * S * 128 + key;
  int run = 0;                                   // sustain run starting at step t+1
  for (int t = S - 1; t >= 0; --t) {
    const float o = on[(size_t)t * 128], sv = su[(size_t)t * 128];
    const bool is_on = custom_round ? (o > 0.95f && o < 1.05f) : (o > 0.5f);
    d[(size_t)t * 128] = is_on ? 1 + run : 0;
    run = (sv > 0.5f) ? run + 1 : 0;
  }
}

int launch_prmat2c_durations(const float* x, int n, int steps, int custom_round, int32_t* dur, hipStream_t stream) {
  PF_REQUIRE(x && dur && n > 0 && steps > 0, "prmat2c_durations: bad arguments");
  hipLaunchKernelGGL(prmat2c_durations_kernel, dim3(n), dim3(128), 0, stream, x, steps, custom_round, dur);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}

// ------------------------------------------------------------------ timestep embedding MLP
// out[b][:] = silu( W2 . silu(W0 . [cos(t f) | sin(t f)] + b0) + b2 )   (the SiLU of emb_layers is folded in)
__global__ __launch_bounds__(256) void time_embed_kernel(const int64_t* __restrict__ t, const float* __restrict__ w0,
                                                         const float* __restrict__ b0, const float* __restrict__ w2,
                                                         const float* __restrict__ b2, float* __restrict__ out, int channels,
                                                         int d_t) {
  extern __shared__ float sm[];  // e[channels] | h[d_t]
  float* e = sm;
  float* hbuf = sm + channels;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int half = channels / 2;
  const float tv = t ? (float)t[b] : (float)b;   // t == nullptr: the table form, row b = time-step value b
  for (int i = tid; i < 2 * half; i += 256) {
    const int j = i % half;
    float a = -9.210340371976184f * (float)j;  // -ln(10000) * j / half, evaluated in fp32 like the reference
    a = a / (float)half;
    const float arg = tv * expf(a);
    e[i] = (i < half) ? cosf(arg) : sinf(arg);
  }
  __syncthreads();
  for (int n = tid; n < d_t; n += 256) {
    float acc = b0[n];
    const float* wr = w0 + (size_t)n * channels;
    for (int k = 0; k < channels; ++k) acc = fmaf(wr[k], e[k], acc);
    hbuf[n] = silu_s(acc);
  }
  __syncthreads();
  for (int n = tid; n < d_t; n += 256) {
    float acc = b2[n];
    const float* wr = w2 + (size_t)n * d_t;
    for (int k = 0; k < d_t; ++k) acc = fmaf(wr[k], hbuf[k], acc);
    out[(size_t)b * d_t + n] = silu_s(acc);
  }
}

int launch_time_embed(const int64_t* t, const float* w0, const float* b0, const float* w2, const float* b2, float* out_silu,
                      int batch, int channels, int d_t, hipStream_t stream) {
  PF_REQUIRE(channels % 2 == 0 && channels + d_t <= 8192, "time_embed: unsupported sizes");
  hipLaunchKernelGGL(time_embed_kernel, dim3(batch), dim3(256), (size_t)(channels + d_t) * sizeof(float), stream, t, w0, b0, w2,
                     b2, out_silu, channels, d_t);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}

// ------------------------------------------------------------------ batched mat-vec: y[b][n] = W[n][:] . x[b][:] + bias[n]
// A wave owns 8 outputs x 8 K-slices (lane = 8*o + j): the 8 lanes of an output stride its weight row in 128-byte
// pieces, every lane keeps 8 batch-row accumulators, and the cross-lane reduction is 3 shuffles per accumulator.
// 8 batch rows per block, so a weight row is read once per 8 samples.  The block first stages its 8 input rows in LDS
// (the only data every output needs), so the global loads of the main loop are the weight pieces alone - independent,
// unrolled 8 deep, all in flight together - instead of nine dependent-latency loads per K step.
// Each row's accumulation is an explicit fmaf chain in a fixed order (identical for every row: batch-position invariant).
__global__ __launch_bounds__(256) void matvec_kernel(const float* __restrict__ x0, int ldx, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ y, int ldy, int B, int N,
                                                     int K, int n_per_group, int x_group_stride) {
  constexpr int RB = 8;
  extern __shared__ __attribute__((aligned(16))) float sx[];   // [RB][K] (vector path only)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int o = lane >> 3, j = lane & 7;
  const int b0 = blockIdx.y * RB;
  const int nb = min(RB, B - b0);
  const int n = (blockIdx.x * 4 + wave) * 8 + o;
  const int nc = min(n, N - 1);                                       // out-of-range outputs compute a duplicate, never store
  const float* wr = w + (size_t)nc * K;
  float acc[RB];
#pragma unroll
  for (int r = 0; r < RB; ++r) acc[r] = 0.f;
  const bool vec = (K % 32 == 0) && (ldx % 4 == 0) && (x_group_stride % 4 == 0);
  // the grouped (block-diagonal) form gives every output group its own input slice: stage only when the block shares one
  const int g_first = (blockIdx.x * 32) / n_per_group, g_last = (min(blockIdx.x * 32 + 31, N - 1)) / n_per_group;
  if (vec && g_first == g_last) {
    const float* xg = x0 + (size_t)g_first * x_group_stride;
    for (int i = threadIdx.x; i < RB * (K / 4); i += 256) {
      const int r = i / (K / 4), k4 = i % (K / 4);
      *reinterpret_cast<float4*>(sx + r * K + k4 * 4) = *reinterpret_cast<const float4*>(xg + (size_t)(b0 + min(r, nb - 1)) * ldx + k4 * 4);
    }
    __syncthreads();
    for (int kb = 0; kb < K; kb += 256) {
      float4 wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = kb + (j + 8 * u) * 4;
        wv[u] = k < K ? *reinterpret_cast<const float4*>(wr + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = kb + (j + 8 * u) * 4;
        if (k < K) {
#pragma unroll
          for (int r = 0; r < RB; ++r) {
            const float4 xv = *reinterpret_cast<const float4*>(sx + r * K + k);
            acc[r] = fmaf(wv[u].w, xv.w, fmaf(wv[u].z, xv.z, fmaf(wv[u].y, xv.y, fmaf(wv[u].x, xv.x, acc[r]))));
          }
        }
      }
    }
  } else {
    const float* x = x0 + (size_t)(nc / n_per_group) * x_group_stride;
    for (int k = j; k < K; k += 8) {
      const float wv = wr[k];
#pragma unroll
      for (int r = 0; r < RB; ++r) acc[r] = fmaf(wv, x[(size_t)(b0 + min(r, nb - 1)) * ldx + k], acc[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    float v = acc[r];
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
    if (j == 0 && n < N && r < nb) y[(size_t)(b0 + r) * ldy + n] = v + (bias ? bias[n] : 0.f);
  }
}

int launch_matvec(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int batch, int n, int k,
                  hipStream_t stream, int n_per_group, int x_group_stride) {
  PF_REQUIRE(x && w && y && batch > 0 && n > 0 && k > 0, "matvec: bad arguments");
  if (n_per_group <= 0) { n_per_group = n; x_group_stride = 0; }
  PF_REQUIRE((size_t)8 * k * sizeof(float) <= 64 * 1024, "matvec: K=%d too large for the staged input rows", k);
  hipLaunchKernelGGL(matvec_kernel, dim3(cdiv(n, 32), cdiv(batch, 8)), dim3(256), (size_t)8 * k * sizeof(float), stream, x, ldx, w, bias, y, ldy, batch, n, k,
                     n_per_group, x_group_stride);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}

// ------------------------------------------------------------------ sampler elementwise kernels
// Individually rounded fp32 operations the optimiser cannot merge.  __fmul_rn / __fadd_rn are plain * and + in this toolchain (no
// OCML_BASIC_ROUNDED_OPERATIONS) and `#pragma clang fp contract(off)` does not survive inlining + SLP vectorisation here (the *_step_rng
// kernels came out with v_pk_fma_f32 where the scalar kernels had separate multiplies and adds: last-bit differences between two
// kernels that must agree).  One opaque VALU instruction per operation: every kernel that inlines the update computes the same bits,
// in the reference's operation order (each product rounded, then the sum).
__device__ __forceinline__ float mul_rn(float a, float b) { float r; asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float add_rn(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float sub_rn(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

static inline dim3 ew_grid(size_t n) { return dim3((unsigned)min((size_t)2048, (n + 255) / 256)); }

__global__ void cfg_combine_kernel(const float* __restrict__ e2, float s, float* __restrict__ e, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float u = e2[i], c = e2[n + i];
    e[i] = add_rn(u, mul_rn(s, sub_rn(c, u)));   // e_u + s * (e_c - e_u) as three rounded operations, like the reference's tensor ops
  }
}
int launch_cfg_combine(const float* eps2, float scale, float* eps, size_t n, hipStream_t s) {
  PF_REQUIRE(eps2 && eps && n > 0, "cfg_combine: bad arguments");
  hipLaunchKernelGGL(cfg_combine_kernel, ew_grid(n), dim3(256), 0, s, eps2, scale, eps, n);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}

// one element of the DDPM / RePaint update on VALUES (has_*: which optional operands take part)
__device__ __forceinline__ float ddpm_update_v(float xv, float ev, bool has_p, float np, bool has_orig, bool has_q, float nq, float ov, float m,
                                               const pf_ddpm_coef& c) {
  // same operation order as the reference (each product rounded, then the sum)
  const float x0 = sub_rn(mul_rn(c.c_recip, xv), mul_rn(c.c_recipm1, ev));
  const float mean = add_rn(mul_rn(c.c_x0, x0), mul_rn(c.c_xt, xv));
  float xu = mean;
  if (has_p) xu = add_rn(mean, mul_rn(c.sigma, np));
  if (has_orig) {
    float xk = mul_rn(c.sqrt_ab, ov);
    if (has_q) xk = add_rn(xk, mul_rn(c.sqrt_1mab, nq));
    xu = add_rn(mul_rn(xk, m), mul_rn(xu, sub_rn(1.0f, m)));
  }
  return xu;
}
__device__ __forceinline__ float ddpm_update(float xv, float ev, const float* np, const float* nq, const float* orig, const float* mask,
                                             const pf_ddpm_coef& c, size_t i) {
  return ddpm_update_v(xv, ev, np != nullptr, np ? np[i] : 0.f, orig != nullptr, nq != nullptr, nq ? nq[i] : 0.f, orig ? orig[i] : 0.f,
                       orig ? mask[i] : 0.f, c);
}
__global__ void ddpm_step_kernel(const float* __restrict__ x, const float* __restrict__ eps, const float* __restrict__ np,
                                 const float* __restrict__ nq, const float* __restrict__ orig, const float* __restrict__ mask,
                                 pf_ddpm_coef c, float* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = ddpm_update(x[i], eps[i], np, nq, orig, mask, c, i);
}
// the same update with the coefficients read from a device table row named by the device-resident step state (graph replay)
__global__ void ddpm_step_dev_kernel(const float* x, const float* __restrict__ eps, const float* __restrict__ np,
                                     const float* __restrict__ nq, const float* __restrict__ orig, const float* __restrict__ mask,
                                     const pf_ddpm_coef* __restrict__ table, const pf_step_state* __restrict__ st, float* out, size_t n) {
  const pf_ddpm_coef c = table[st->index];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = ddpm_update(x[i], eps[i], np, nq, orig, mask, c, i);
}
int launch_ddpm_step_dev(const float* x, const float* eps, const float* noise_p, const float* noise_q, const float* orig,
                         const float* mask, const pf_ddpm_coef* table, const pf_step_state* st, float* out, size_t n, hipStream_t s) {
  PF_REQUIRE(x && eps && out && table && st && n > 0 && (!orig || mask), "ddpm_step_dev: bad arguments");
  hipLaunchKernelGGL(ddpm_step_dev_kernel, ew_grid(n), dim3(256), 0, s, x, eps, noise_p, noise_q, orig, mask, table, st, out, n);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}
int launch_ddpm_step(const float* x, const float* eps, const float* noise_p, const float* noise_q, const float* orig,
                     const float* mask, const pf_ddpm_coef& c, float* out, size_t n, hipStream_t s) {
  PF_REQUIRE(x && eps && out && n > 0 && (!orig || mask), "ddpm_step: bad arguments");
  hipLaunchKernelGGL(ddpm_step_kernel, ew_grid(n), dim3(256), 0, s, x, eps, noise_p, noise_q, orig, mask, c, out, n);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}

__global__ void axpby_kernel(const float* __restrict__ x, const float* __restrict__ y, float a, float b, float* __restrict__ out,
                             size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = add_rn(mul_rn(a, x[i]), mul_rn(b, y[i]));
}
int launch_axpby(const float* x, const float* y, float a, float b, float* out, size_t n, hipStream_t s) {
  PF_REQUIRE(x && y && out && n > 0, "axpby: bad arguments");
  hipLaunchKernelGGL(axpby_kernel, ew_grid(n), dim3(256), 0, s, x, y, a, b, out, n);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}

__device__ __forceinline__ float ddim_update_v(float xv, float e, bool has_noise, float nz, bool has_orig, float ov, float onv, float m,
                                               const pf_ddim_coef& c) {
  const float p0 = __fdiv_rn(sub_rn(xv, mul_rn(c.s1m, e)), c.sqrt_a);   // (a correctly rounded division has nothing to be merged with)
  float xp = add_rn(mul_rn(c.sqrt_aprev, p0), mul_rn(c.dir_coef, e));
  if (has_noise) xp = add_rn(xp, mul_rn(c.sigma, nz));
  if (has_orig) {
    const float ot = add_rn(mul_rn(c.q_sqrt_a, ov), mul_rn(c.q_s1m, onv));
    xp = add_rn(mul_rn(ot, m), mul_rn(xp, sub_rn(1.0f, m)));
  }
  return xp;
}
__device__ __forceinline__ float ddim_update(float xv, float e, const float* noise, const float* orig, const float* on, const float* mask,
                                             const pf_ddim_coef& c, size_t i) {
  return ddim_update_v(xv, e, noise != nullptr, noise ? noise[i] : 0.f, orig != nullptr, orig ? orig[i] : 0.f, orig ? on[i] : 0.f,
                       orig ? mask[i] : 0.f, c);
}
__global__ void ddim_step_kernel(const float* __restrict__ x, const float* __restrict__ eps, const float* __restrict__ noise,
                                 const float* __restrict__ orig, const float* __restrict__ on, const float* __restrict__ mask,
                                 pf_ddim_coef c, float* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = ddim_update(x[i], eps[i], noise, orig, on, mask, c, i);
}
__global__ void ddim_step_dev_kernel(const float* x, const float* __restrict__ eps, const float* __restrict__ noise,
                                     const float* __restrict__ orig, const float* __restrict__ on, const float* __restrict__ mask,
                                     const pf_ddim_coef* __restrict__ table, const pf_step_state* __restrict__ st, float* out, size_t n) {
  const pf_ddim_coef c = table[st->index];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = ddim_update(x[i], eps[i], noise, orig, on, mask, c, i);
}
int launch_ddim_step_dev(const float* x, const float* eps, const float* noise, const float* orig, const float* orig_noise,
                         const float* mask, const pf_ddim_coef* table, const pf_step_state* st, float* out, size_t n, hipStream_t s) {
  PF_REQUIRE(x && eps && out && table && st && n > 0 && (!orig || (mask && orig_noise)), "ddim_step_dev: bad arguments");
  hipLaunchKernelGGL(ddim_step_dev_kernel, ew_grid(n), dim3(256), 0, s, x, eps, noise, orig, orig_noise, mask, table, st, out, n);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}
// ---- device-resident step state: what changes from one reverse step to the next lives in device memory, so a captured
// step (hipGraph) can be replayed unchanged
__global__ void step_state_set_kernel(pf_step_state* st, long long index, unsigned long long draws) { st->index = index; st->draws = draws; }
__global__ void step_begin_kernel(const pf_step_state* __restrict__ st, const int* __restrict__ time_steps, long long* __restrict__ t_out, int batch) {
  const long long t = time_steps ? (long long)time_steps[st->index] : (long long)st->index;
  for (int i = threadIdx.x; i < batch; i += blockDim.x) t_out[i] = t;
}
__global__ void step_end_kernel(pf_step_state* st, int draws_used) { st->index -= 1; st->draws += (unsigned long long)draws_used; }
int launch_step_state_set(pf_step_state* st, int64_t index, uint64_t draws, hipStream_t s) {
  PF_REQUIRE(st, "step_state_set: null state");
  hipLaunchKernelGGL(step_state_set_kernel, dim3(1), dim3(1), 0, s, st, (long long)index, (unsigned long long)draws);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}
int launch_step_begin(const pf_step_state* st, const int* time_steps, int64_t* t_out, int batch, hipStream_t s) {
  PF_REQUIRE(st && t_out && batch > 0, "step_begin: bad arguments");
  hipLaunchKernelGGL(step_begin_kernel, dim3(1), dim3(64), 0, s, st, time_steps, reinterpret_cast<long long*>(t_out), batch);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}
int launch_step_end(pf_step_state* st, int draws_used, hipStream_t s) {
  PF_REQUIRE(st && draws_used >= 0, "step_end: bad arguments");
  hipLaunchKernelGGL(step_end_kernel, dim3(1), dim3(1), 0, s, st, draws_used);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}
int launch_ddim_step(const float* x, const float* eps, const float* noise, const float* orig, const float* orig_noise,
                     const float* mask, const pf_ddim_coef& c, float* out, size_t n, hipStream_t s) {
  PF_REQUIRE(x && eps && out && n > 0 && (!orig || (mask && orig_noise)), "ddim_step: bad arguments");
  hipLaunchKernelGGL(ddim_step_kernel, ew_grid(n), dim3(256), 0, s, x, eps, noise, orig, orig_noise, mask, c, out, n);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}

// ------------------------------------------------------------------ Philox4x32-10 + Box-Muller
__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
// the four standard normals of group g (elements 4g .. 4g+3 of the global tensor) of draw `sid`.
// NOT inlined: randn_kernel and the *_step_rng kernels must produce the same bits for the same (g, sid, seed), and the generated code
// of logf / sincosf / the products around them depends on its surroundings when inlined (measured: two calls in one kernel differ from
// the stand-alone kernel in the last bit of ~20 % of the draws) - one shared body is the same arithmetic by construction.
__device__ __noinline__ f32x4 philox_normal4v(uint64_t g, uint64_t sid, uint64_t seed) {
#pragma clang fp contract(off)
  float z[4];
  uint32_t c0 = (uint32_t)g, c1 = (uint32_t)(g >> 32), c2 = (uint32_t)sid, c3 = (uint32_t)(sid >> 32);
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) { philox_round(c0, c1, c2, c3, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
  const float u0 = ((float)(c0 >> 8) + 0.5f) * (1.0f / 16777216.0f), u1 = ((float)(c1 >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u2 = ((float)(c2 >> 8) + 0.5f) * (1.0f / 16777216.0f), u3 = ((float)(c3 >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float ra = sqrtf(-2.0f * logf(u0)), rb = sqrtf(-2.0f * logf(u2));
  float sa, ca, sb, cb;
  sincosf(6.283185307179586f * u1, &sa, &ca);
  sincosf(6.283185307179586f * u3, &sb, &cb);
  z[0] = ra * ca; z[1] = ra * sa; z[2] = rb * cb; z[3] = rb * sb;
  return f32x4{z[0], z[1], z[2], z[3]};
}
__device__ __forceinline__ void philox_normal4(uint64_t g, uint64_t sid, uint64_t seed, float (&z)[4]) {
  const f32x4 v = philox_normal4v(g, sid, seed);
  z[0] = v[0]; z[1] = v[1]; z[2] = v[2]; z[3] = v[3];
}
__device__ __forceinline__ void randn_body(float* __restrict__ out, size_t n, uint64_t seed, uint64_t sid, uint64_t off) {
  const uint64_t g0 = off >> 2, g1 = (off + n + 3) >> 2;
  for (uint64_t g = g0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < g1; g += (uint64_t)gridDim.x * blockDim.x) {
    float z[4];
    philox_normal4(g, sid, seed, z);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint64_t e = g * 4 + j;
      if (e >= off && e < off + n) out[e - off] = z[j];
    }
  }
}
__global__ void randn_kernel(float* __restrict__ out, size_t n, uint64_t seed, uint64_t sid, uint64_t off) { randn_body(out, n, seed, sid, off); }
// draw index = device-resident counter + slot (the position of this draw inside the step)
__global__ void randn_dev_kernel(float* __restrict__ out, size_t n, uint64_t seed, const pf_step_state* __restrict__ st, int slot, uint64_t off) {
  randn_body(out, n, seed, st->draws + (uint64_t)slot, off);
}
int launch_randn_dev(float* out, size_t n, uint64_t seed, const pf_step_state* st, int slot, uint64_t elem_offset, hipStream_t s) {
  PF_REQUIRE(out && st && n > 0 && slot >= 0, "randn_dev: bad arguments");
  hipLaunchKernelGGL(randn_dev_kernel, ew_grid(n / 4 + 1), dim3(256), 0, s, out, n, seed, st, slot, elem_offset);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}
int launch_randn(float* out, size_t n, uint64_t seed, uint64_t stream_id, uint64_t elem_offset, hipStream_t s) {
  PF_REQUIRE(out && n > 0, "randn: bad arguments");
  hipLaunchKernelGGL(randn_kernel, ew_grid(n / 4 + 1), dim3(256), 0, s, out, n, seed, stream_id, elem_offset);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}

// ---- the sampler updates with the noise drawn in the kernel: a thread owns the four elements of one Philox group, the draws are
// philox_normal4 of (group, draw index, seed) - exactly what randn_kernel stores for that draw - and the update is ddpm_update /
// ddim_update on those values: bit-identical to randn + step, without the noise tensors' round trip through HBM
template <class Coef>
__device__ __forceinline__ Coef coef_of(const Coef& by_value, const Coef* table, const pf_step_state* st) { return table ? table[st->index] : by_value; }
__global__ void ddpm_step_rng_kernel(const float* x, const float* __restrict__ eps, const float* __restrict__ orig, const float* __restrict__ mask,
                                     pf_ddpm_coef cv, const pf_ddpm_coef* __restrict__ table, const pf_step_state* __restrict__ st, uint64_t seed,
                                     uint64_t draw_q, uint64_t draw_p, uint64_t off, float* out, size_t n) {
  const pf_ddpm_coef c = coef_of(cv, table, st);
  if (st) { draw_q = st->draws; draw_p = st->draws + (orig ? 1u : 0u); }
  const size_t ng = n >> 2;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < ng; g += (size_t)gridDim.x * blockDim.x) {
    float zp[4], zq[4] = {0.f, 0.f, 0.f, 0.f};
    philox_normal4((off >> 2) + g, draw_p, seed, zp);
    if (orig) philox_normal4((off >> 2) + g, draw_q, seed, zq);
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + 4 * g), ev = *reinterpret_cast<const f32x4*>(eps + 4 * g);
    f32x4 ov = {0.f, 0.f, 0.f, 0.f}, mv = ov, o;
    if (orig) { ov = *reinterpret_cast<const f32x4*>(orig + 4 * g); mv = *reinterpret_cast<const f32x4*>(mask + 4 * g); }
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = ddpm_update_v(xv[j], ev[j], true, zp[j], orig != nullptr, orig != nullptr, zq[j], ov[j], mv[j], c);
    *reinterpret_cast<f32x4*>(out + 4 * g) = o;
  }
}
__global__ void ddim_step_rng_kernel(const float* x, const float* __restrict__ eps, const float* __restrict__ orig, const float* __restrict__ on,
                                     const float* __restrict__ mask, pf_ddim_coef cv, const pf_ddim_coef* __restrict__ table,
                                     const pf_step_state* __restrict__ st, uint64_t seed, uint64_t draw, uint64_t off, float* out, size_t n) {
  const pf_ddim_coef c = coef_of(cv, table, st);
  if (st) draw = st->draws;
  const size_t ng = n >> 2;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < ng; g += (size_t)gridDim.x * blockDim.x) {
    float z[4];
    philox_normal4((off >> 2) + g, draw, seed, z);
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + 4 * g), ev = *reinterpret_cast<const f32x4*>(eps + 4 * g);
    f32x4 ov = {0.f, 0.f, 0.f, 0.f}, nv = ov, mv = ov, o;
    if (orig) { ov = *reinterpret_cast<const f32x4*>(orig + 4 * g); nv = *reinterpret_cast<const f32x4*>(on + 4 * g); mv = *reinterpret_cast<const f32x4*>(mask + 4 * g); }
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = ddim_update_v(xv[j], ev[j], true, z[j], orig != nullptr, ov[j], nv[j], mv[j], c);
    *reinterpret_cast<f32x4*>(out + 4 * g) = o;
  }
}
int launch_ddpm_step_rng(const float* x, const float* eps, const float* orig, const float* mask, const pf_ddpm_coef* c_host,
                         const pf_ddpm_coef* table, const pf_step_state* st, uint64_t seed, uint64_t draw_q, uint64_t draw_p, uint64_t off,
                         float* out, size_t n, hipStream_t s) {
  PF_REQUIRE(x && eps && out && n > 0 && (!orig || mask) && (c_host || (table && st)), "ddpm_step_rng: bad arguments");
  PF_REQUIRE(n % 4 == 0 && off % 4 == 0, "ddpm_step_rng: n and elem_offset must be multiples of 4 (one Philox call yields four normals)");
  PF_REQUIRE((((uintptr_t)x | (uintptr_t)eps | (uintptr_t)out | (uintptr_t)orig | (uintptr_t)mask) & 15) == 0, "ddpm_step_rng: tensors must be 16-byte aligned");
  hipLaunchKernelGGL(ddpm_step_rng_kernel, ew_grid(n / 4), dim3(256), 0, s, x, eps, orig, mask, c_host ? *c_host : pf_ddpm_coef{}, c_host ? nullptr : table,
                     c_host ? nullptr : st, seed, draw_q, draw_p, off, out, n);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}
int launch_ddim_step_rng(const float* x, const float* eps, const float* orig, const float* orig_noise, const float* mask,
                         const pf_ddim_coef* c_host, const pf_ddim_coef* table, const pf_step_state* st, uint64_t seed, uint64_t draw,
                         uint64_t off, float* out, size_t n, hipStream_t s) {
  PF_REQUIRE(x && eps && out && n > 0 && (!orig || (mask && orig_noise)) && (c_host || (table && st)), "ddim_step_rng: bad arguments");
  PF_REQUIRE(n % 4 == 0 && off % 4 == 0, "ddim_step_rng: n and elem_offset must be multiples of 4 (one Philox call yields four normals)");
  PF_REQUIRE((((uintptr_t)x | (uintptr_t)eps | (uintptr_t)out | (uintptr_t)orig | (uintptr_t)orig_noise | (uintptr_t)mask) & 15) == 0, "ddim_step_rng: tensors must be 16-byte aligned");
  hipLaunchKernelGGL(ddim_step_rng_kernel, ew_grid(n / 4), dim3(256), 0, s, x, eps, orig, orig_noise, mask, c_host ? *c_host : pf_ddim_coef{},
                     c_host ? nullptr : table, c_host ? nullptr : st, seed, draw, off, out, n);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}

// ------------------------------------------------------------------ clock probe
// {shader-clock counter, constant-rate reference counter} per XCD at the moment the probe runs: two probes bracketing a stretch of
// stream work give the AVERAGE shader clock of every XCD over it (the part is power-managed: the same binary clocks differently from
// box to box and under different loads, and the eight XCDs have counters - and clocks - of their own, so a begin / end pair must
// come from the same XCD).  s_memtime counts shader cycles, s_memrealtime the fixed reference clock (100 MHz on gfx950; bench.py
// calibrates it against the host clock instead of assuming it).  64 single-wave workgroups are dealt round-robin over the XCDs;
// each writes row XCC_ID of out[8][2] (several per XCD: the last writer wins, they differ by nanoseconds).
__global__ void clock_probe_kernel(unsigned long long* out) {
  if (threadIdx.x != 0) return;
  const unsigned xcc = __builtin_amdgcn_s_getreg((20 /* HW_REG_XCC_ID */) | (0 << 6) | ((4 - 1) << 11)) & 7u;   // bits 3:0 = XCC id
  out[2 * xcc + 0] = __builtin_amdgcn_s_memtime();
  out[2 * xcc + 1] = __builtin_amdgcn_s_memrealtime();
}
int launch_clock_probe(unsigned long long* out2, hipStream_t s) {
  PF_REQUIRE(out2, "clock_probe: null output");
  hipLaunchKernelGGL(clock_probe_kernel, dim3(64), dim3(64), 0, s, out2);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}

// ------------------------------------------------------------------ matrix-pipe probe
// What the bf16 matrix pipe SUSTAINS on this box, as a reference for roofline fractions: one 8-wave workgroup per CU (two waves per SIMD),
// every wave a dependency-free stream of v_mfma_f32_32x32x16_bf16 over four accumulators - 100 % pipe duty - on operands with random signs
// and mantissas.  The part is power-managed: with such operands a full pipe clocks ~1.6 GHz (all-ones operands: ~2.3 GHz), so the
// sustained rate sits well below the nominal 2.5 PFLOP/s (tools/micro/tap_pingpong.hip, profiles/r04_ab_pingpong.md).
// bench.py times a few launches with events; flops per launch = grid x 8 waves x 24 x 32768 x iters.
__global__ __launch_bounds__(512) void mfma_probe_kernel(float* sink, int iters) {
  typedef float f32x16_t __attribute__((ext_vector_type(16)));
  typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  f32x16_t acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  unsigned h = (threadIdx.x + 1) * 2654435761u + blockIdx.x * 40503u;
  auto nx = [&]() { h ^= h << 13; h ^= h >> 17; h ^= h << 5; return (h & 0x807f807fu) | 0x3f003f00u; };   // bf16 pairs in +-[0.5, 1)
  bf16x8_t a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const u32x4_t ua = {nx(), nx(), nx(), nx()}, ub = {nx(), nx(), nx(), nx()};
    a[i] = __builtin_bit_cast(bf16x8_t, ua); b[i] = __builtin_bit_cast(bf16x8_t, ub);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 24; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i >> 2) & 3], b[(i + (i >> 3)) & 3], acc[i & 3], 0, 0, 0);
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][7];
  if (r == 12345.678f) sink[0] = r;   // keeps the stream alive; never true in practice
}
int launch_mfma_probe(float* sink, int iters, double* flops, hipStream_t s) {
  PF_REQUIRE(sink && iters > 0 && iters <= (1 << 20), "mfma_probe: bad arguments");
  const int grid = num_cus();
  hipLaunchKernelGGL(mfma_probe_kernel, dim3(grid), dim3(512), 0, s, sink, iters);
  PF_CHECK_HIP(hipGetLastError());
  if (flops) *flops = (double)grid * 8.0 * 24.0 * 32768.0 * iters;
  return PF_OK;
}

// ------------------------------------------------------------------ GRU cell update (torch gate order r, z, n)
// gi = W_ih x_t + b_ih  [B][3H] (row stride ld_gi), gh = W_hh h + b_hh [B][3H]
__global__ void gru_gates_kernel(const float* __restrict__ gi, int ld_gi, const float* __restrict__ gh, float* __restrict__ h,
                                 int ld_h, int B, int H) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * H) return;
  const int b = i / H, j = i % H;
  const float* a = gi + (size_t)b * ld_gi;
  const float* c = gh + (size_t)b * 3 * H;
  const float r = 1.0f / (1.0f + expf(-(a[j] + c[j])));
  const float z = 1.0f / (1.0f + expf(-(a[H + j] + c[H + j])));
  const float nn = tanhf(a[2 * H + j] + r * c[2 * H + j]);
  float* hp = h + (size_t)b * ld_h + j;
  *hp = (1.0f - z) * nn + z * *hp;
}
int launch_gru_gates(const float* gi, int ld_gi, const float* gh, float* h, int ld_h, int batch, int hidden, hipStream_t s) {
  hipLaunchKernelGGL(gru_gates_kernel, dim3(cdiv(batch * hidden, 256)), dim3(256), 0, s, gi, ld_gi, gh, h, ld_h, batch, hidden);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}

// ------------------------------------------------------------------ PianoTree encoder pieces (dl_modules/pianotree_enc.py)
// grid row = (pitch index, 5 duration digits) as floats; its multi-hot form (one-hot pitch over P classes - the pad index P has no
// column - followed by the five digits, :78-95) times note_embedding (:43): out[row][j] = b[j] + W[j][pitch] + sum_k d_k W[j][P+k]
__global__ void pnotree_embed_kernel(const float* __restrict__ grid, const float* __restrict__ w, const float* __restrict__ bias,
                                     float* __restrict__ out, int rows, int E, int P) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * E) return;
  const int row = i / E, j = i % E;
  const float* g = grid + (size_t)row * 6;
  const float* wj = w + (size_t)j * (P + 5);
  const int pitch = (int)g[0];
  float acc = bias[j];
  if (pitch >= 0 && pitch < P) acc += wj[pitch];
  for (int k = 0; k < 5; ++k) acc = fmaf(g[1 + k], wj[P + k], acc);
  out[i] = acc;
}
// notes sounding at a time step = S minus the pad entries (get_len_index_tensor :69-76)
__global__ void pnotree_lengths_kernel(const float* __restrict__ grid, int* __restrict__ lens, int nseq, int S, int pad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nseq) return;
  int n = S;
  for (int k = 0; k < S; ++k) n -= ((int)grid[((size_t)i * S + k) * 6] == pad);
  lens[i] = n;
}
// GRU cell update over packed variable-length sequences (pack_padded_sequence, :104-107): sequence b takes part in `step` only while
// step < lens[b]; its input is element `step` (forward) or lens[b]-1-step (reverse) of gi [N][S][3H]
__global__ void gru_gates_masked_kernel(const float* __restrict__ gi, const float* __restrict__ gh, float* __restrict__ h, int ld_h,
                                        int N, int H, int S, const int* __restrict__ lens, int step, int reverse) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * H) return;
  const int b = i / H, j = i % H;
  const int len = lens[b];
  if (step >= len) return;
  const int t = reverse ? len - 1 - step : step;
  const float* a = gi + ((size_t)b * S + t) * 3 * H;
  const float* c = gh + (size_t)b * 3 * H;
  const float r = 1.0f / (1.0f + expf(-(a[j] + c[j])));
  const float z = 1.0f / (1.0f + expf(-(a[H + j] + c[H + j])));
  const float nn = tanhf(a[2 * H + j] + r * c[2 * H + j]);
  float* hp = h + (size_t)b * ld_h + j;
  *hp = (1.0f - z) * nn + z * *hp;
}
int launch_pnotree_embed(const float* grid, const float* w, const float* bias, float* out, int rows, int emb, int pitch_range, hipStream_t s) {
  hipLaunchKernelGGL(pnotree_embed_kernel, dim3(cdiv(rows * emb, 256)), dim3(256), 0, s, grid, w, bias, out, rows, emb, pitch_range);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}
int launch_pnotree_lengths(const float* grid, int* lens, int nseq, int max_simu_note, int pad, hipStream_t s) {
  hipLaunchKernelGGL(pnotree_lengths_kernel, dim3(cdiv(nseq, 256)), dim3(256), 0, s, grid, lens, nseq, max_simu_note, pad);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}
int launch_gru_gates_masked(const float* gi, const float* gh, float* h, int ld_h, int nseq, int hidden, int seq_len, const int* lens,
                            int step, int reverse, hipStream_t s) {
  hipLaunchKernelGGL(gru_gates_masked_kernel, dim3(cdiv(nseq * hidden, 256)), dim3(256), 0, s, gi, gh, h, ld_h, nseq, hidden, seq_len, lens, step,
                     reverse);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}

// ------------------------------------------------------------------ texture encoder front end
// pr [B,32,128] -> conv(1->C,(4,12),stride(4,1)) -> ReLU -> maxpool(1,4) -> [B,C,8,29] (contiguous; the
// reference then reinterprets this buffer as [B,8,C*29] without a permute - txt_enc.py:27).
__global__ void txt_frontend_kernel(const float* __restrict__ pr, const float* __restrict__ w, const float* __restrict__ bias,
                                    float* __restrict__ out, int B, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = B * C * 8 * 29;
  if (i >= total) return;
  const int pw = i % 29, row = (i / 29) % 8, ch = (i / (29 * 8)) % C, b = i / (29 * 8 * C);
  const float* src = pr + (size_t)b * 32 * 128 + (size_t)row * 4 * 128;
  const float* wk = w + (size_t)ch * 48;
  float best = 0.f;  // ReLU floor
  for (int q = 0; q < 4; ++q) {
    const int col = pw * 4 + q;
    float acc = bias[ch];
    for (int r = 0; r < 4; ++r)
      for (int s = 0; s < 12; ++s) acc = fmaf(src[r * 128 + col + s], wk[r * 12 + s], acc);
    best = fmaxf(best, acc);
  }
  out[i] = best;
}
int launch_txt_frontend(const float* pr, const float* w, const float* bias, float* out, int batch, int num_channel, hipStream_t s) {
  hipLaunchKernelGGL(txt_frontend_kernel, dim3(cdiv(batch * num_channel * 8 * 29, 256)), dim3(256), 0, s, pr, w, bias, out, batch,
                     num_channel);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}

}  // namespace pf
