// conv_common.h - kernel parameter block and the fused epilogue shared by conv_mfma.hip (fp32 MFMA)
// and conv_bf16x3.hip (split bf16 MFMA).  Both kernels finish with the same 32x32 accumulator
// fragments (C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)).
#pragma once
#include <type_traits>

#include "pf_internal.h"

namespace pf {

typedef x3_t x3x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// LDS fragment read issued as inline asm, NOT as a C++ load: with direct-to-LDS loads in flight the compiler's wait-count
// pass degrades every LDS dependency to lgkmcnt(0) (a pending global_load_lds counts as a "flat" access), which serialises
// each ds_read with its MFMA.  The direct-to-LDS pipelines (3x3 loop, planes GEMM, bf16x3 attention) issue their reads here and place counted
// s_waitcnt lgkmcnt(N) themselves.
template <int OFF>
__device__ __forceinline__ x3x8 lds_read128(unsigned addr) {
  u32x4 v;
  constexpr int HI = OFF & ~0xFFFF, LO = OFF & 0xFFFF;   // the instruction's offset field is 16 bits
  if constexpr (HI != 0) addr += HI;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(LO));
  return __builtin_bit_cast(x3x8, v);
}
// Direct-to-LDS copy of 16 bytes per lane in its BUFFER form: SGPR resource (base) + one 32-bit offset VGPR per lane + a wave-uniform
// SGPR offset, destination = M0-based LDS address + lane * 16.  Next to MFMAs the global form (a 64-bit address VGPR pair per lane)
// costs its wave 100-300 cycles to issue at two waves per SIMD, this one 20-110 (tools/micro/glds_issue.hip): every streaming loop of
// the library issues its tile pieces this way.  `base` must cover [voff + soff, +16) for every lane; offsets are bytes, < 2 GiB.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dma_resource(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, int voff, int soff, void* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}

template <int N>
__device__ __forceinline__ void lgkm_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N)); }
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}


// Division of a non-negative int (< 2^31) by a launch-time constant as multiply-high + shift: the tile decode at the top of every
// workgroup otherwise runs four to six emulated integer divisions (no divide instruction: ~40 instructions each, ~2 k cycles
// before the first load is issued).  mul == 0 encodes a divisor of 1.
struct FastDiv { unsigned mul, shr; };
static inline FastDiv make_fastdiv(int d) {
  if (d <= 1) return FastDiv{0u, 0u};
  unsigned l = 31u - (unsigned)__builtin_clz((unsigned)d);
  if (d & (d - 1)) ++l;                                   // ceil(log2 d)
  const unsigned p = 31u + l;
  return FastDiv{(unsigned)(((1ull << p) + (unsigned)d - 1u) / (unsigned)d), p - 32u};
}
__device__ __forceinline__ int fdiv(int x, FastDiv f) { return f.mul ? (int)(__umulhi((unsigned)x, f.mul) >> f.shr) : x; }

struct ConvP {
  const float* x0; const float* x1; int c0, c1;
  int B, Hin, Win, Hout, Wout;
  const void* w; int N, Npad;
  const float* sc; const float* sh; const float* mean; const float* rstd;
  const float* bias; const float* sbias; int ld_sbias; const float* res; int ld_res;
  const long long* sb_rows; int sb_nrows;   // optional: sample b reads row clamp(sb_rows[b]) of sbias (hoisted time-bias table indexed by t[b])
  int geglu;
  float* out; int ld_out;
  float* stats;   // optional [B][tiles][N][2]: per-tile, per-channel (sum, sum of squares) of the stored outputs
  int tiles_x, tiles_y, nt;
  FastDiv d_tx, d_ty, d_nt, d_ks;   // reciprocals of tiles_x, tiles_y, nt, ksplit (conv_fill_divs)
  int fold, fold_py, fold_px;   // parity-folded upsampling conv: tiles walk the SOURCE grid, pixel (y, x) is stored at (2y+py, 2x+px)
  const float* sx0; const float* sx1; int sc0, sc1;   // fused 1x1 projection of a second tensor (ResBlock skip conv)
  const void* sw; const float* bias2;
  void* out_planes;             // result as bf16 hi/lo planes [M][ld_out] | [M][ld_out] instead of fp32 (consumer: gemm_planes_bf3.hip)
  void* qkv;                    // fused q|k|v projection written as bf16 hi/lo planes for attention_bf3.hip (d_head 64)
  int pp;                       // host side only: the two-group ping-pong form is allowed (pf_conv_args.no_pp)
  int ksplit; float* partial;   // split-K: raw accumulators to partial[split][M][N]; bias/residual/statistics happen in the reduce kernel
  // GroupNorm finalize inside the consumer (bf16x3 kernels): per-tile statistics of the one or two producers, see gn_fused_prologue
  const float* gn_s0; const float* gn_s1; int gn_t0, gn_t1;
  const float* gn_gamma; const float* gn_beta; float gn_eps; int gn_groups;
  // x1_bmod > 0: the SECOND source (x1, sx1, gn_s1) holds only x1_bmod samples and sample b reads sample b % x1_bmod of it - the skip
  // tensors of the classifier-free-guidance forward, whose conditional and unconditional halves share everything computed before the first
  // transformer block (pf_unet_forward_cfg).  The kernels fold it into the base pointers once per workgroup (conv_shared_x1).
  int x1_bmod;
  // optional range telemetry (pf_unet_track_absmax): the largest |value| this launch stores, as fp32 bits, max-combined into one device word
  unsigned* amax;
};

// rebase the second-source pointers of this workgroup's sample (see ConvP::x1_bmod); hw_in / hw_out: pixels per sample of x1 / sx1
__device__ __forceinline__ void conv_shared_x1(ConvP& p, int b) {
  if (p.x1_bmod > 0) {
    const size_t d = (size_t)(b - b % p.x1_bmod);
    if (p.x1) p.x1 -= d * ((size_t)p.Hin * p.Win * p.c1);
    if (p.sx1) p.sx1 -= d * ((size_t)p.Hout * p.Wout * p.sc1);
    if (p.gn_s1) p.gn_s1 -= d * ((size_t)p.gn_t1 * p.c1 * 2);
  }
}

static inline void conv_fill_divs(ConvP& p) {
  p.d_tx = make_fastdiv(p.tiles_x); p.d_ty = make_fastdiv(p.tiles_y); p.d_nt = make_fastdiv(p.nt); p.d_ks = make_fastdiv(p.ksplit);
}

// GroupNorm scale/shift of THIS workgroup's sample, computed by the consuming kernel itself instead of a separate finalize launch
// (a 5 us launch between every two convolutions).  Same arithmetic as gn_finalize_tiles_kernel: the per-tile
// (sum, sumsq) of every channel are added in fp64, then the group's channels, mean / rstd in fp64, scale = gamma*rstd and
// shift = beta - mean*rstd*gamma rounded to fp32.  Every workgroup of a sample writes the same values to the sample's rows of
// p.sc / p.sh; the block-level fence + barrier make them visible to this workgroup's own later loads.  Deterministic.
// `scratch`: LDS, 2*cin doubles, not in use by anything else yet.  Must be called by all NTHREADS threads of the workgroup.
template <int NTHREADS>
__device__ __forceinline__ void gn_fused_prologue(const ConvP& p, int b, int tid_wg, int hw, double* scratch) {
  const int cin = p.c0 + p.c1, gs = cin / p.gn_groups;
  for (int c = tid_wg; c < cin; c += NTHREADS) {
    const float* s; int T, C, cl;
    if (c < p.c0) { s = p.gn_s0; T = p.gn_t0; C = p.c0; cl = c; } else { s = p.gn_s1; T = p.gn_t1; C = p.c1; cl = c - p.c0; }
    double a = 0.0, q = 0.0;
    // eight tiles' loads in flight at a time, added in tile order (the finalize kernel's order): one load per iteration is one L2
    // round trip per tile - 14 k cycles for 16 tiles in the cycle stamps of preattn_fused_bf3.hip, most of a small kernel's prologue
    for (int t0 = 0; t0 < T; t0 += 8) {
      float2 vv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        vv[j] = *reinterpret_cast<const float2*>(s + (((size_t)b * T + min(t0 + j, T - 1)) * C + cl) * 2);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (t0 + j < T) { a += vv[j].x; q += vv[j].y; }
    }
    scratch[2 * c] = a; scratch[2 * c + 1] = q;
  }
  __syncthreads();
  float* scw = const_cast<float*>(p.sc) + (size_t)b * cin;
  float* shw = const_cast<float*>(p.sh) + (size_t)b * cin;
  for (int c = tid_wg; c < cin; c += NTHREADS) {
    const int g0 = (c / gs) * gs;
    double a = 0.0, q = 0.0;
    for (int j = 0; j < gs; ++j) { a += scratch[2 * (g0 + j)]; q += scratch[2 * (g0 + j) + 1]; }
    const double cnt = (double)gs * hw;
    const double mean = a / cnt;
    double var = q / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)p.gn_eps);
    const double ga = p.gn_gamma[c];
    scw[c] = (float)(rstd * ga);
    shw[c] = (float)((double)p.gn_beta[c] - mean * rstd * ga);
  }
  __threadfence_block();
  __syncthreads();
}

// ---- range telemetry.  Every tensor a layer stores is (after at most a normalisation, which only shrinks it) the split A operand of the next
// layer; in the fp16-piece build such an operand overflows beyond 65504.  With a slot bound (ConvP::amax / the launchers' `amax` argument) the
// epilogues fold |v| of everything they store into a per-thread maximum - compared as unsigned bit patterns, so that a NaN (0x7fc00000) or an
// inf is kept, not dropped - and a_max_flush combines it across the wave and into the slot with ONE atomic per wave, no return value.
__device__ __forceinline__ void amax_acc(unsigned& m, float v) { m = max(m, __float_as_uint(v) & 0x7fffffffu); }
__device__ __forceinline__ void amax_acc4(unsigned& m, f32x4 v) { amax_acc(m, v[0]); amax_acc(m, v[1]); amax_acc(m, v[2]); amax_acc(m, v[3]); }
__device__ __forceinline__ void amax_flush(unsigned* slot, unsigned m) {
  if (!slot) return;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0 && m) atomicMax(slot, m);
}

// x * sigmoid(x) with the hardware reciprocal (1 ulp) instead of an IEEE division sequence (10 instructions shorter)
__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
// exact-erf GELU (F.gelu default, unet_attention.py:333).  erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, the size
// of an fp32 ulp of erf near 1): one v_exp + one v_rcp + a degree-5 Horner chain instead of the ~50-instruction libm erff,
// which dominated the GeGLU epilogue (64 evaluations per lane per tile).
__device__ __forceinline__ float erf_as_f(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));   // hardware reciprocal (1 ulp): an IEEE division is ten instructions
  float pq = fmaf(1.061405429f, t, -1.453152027f);
  pq = fmaf(pq, t, 1.421413741f);
  pq = fmaf(pq, t, -0.284496736f);
  pq = fmaf(pq, t, 0.254829592f);
  const float y = 1.0f - pq * t * __expf(-ax * ax);
  return copysignf(y, x);
}
__device__ __forceinline__ float gelu_erf_f(float g) { return 0.5f * g * (1.0f + erf_as_f(g * 0.70710678118654752440f)); }

__device__ __forceinline__ void store_out(const ConvP& p, size_t idx, float v) { p.out[idx] = v; }
// this sample's row of the per-sample bias (nullptr: none)
__device__ __forceinline__ const float* sbias_row(const ConvP& p, int b) {
  if (!p.sbias) return nullptr;
  long long r = b;
  if (p.sb_rows) { r = p.sb_rows[b]; r = r < 0 ? 0 : (r >= p.sb_nrows ? p.sb_nrows - 1 : r); }
  return p.sbias + (size_t)r * p.ld_sbias;
}
// flat output pixel index of tile pixel (oy, ox) of sample b
__device__ __forceinline__ size_t out_pixel(const ConvP& p, int b, int oy, int ox) {
  if (p.fold) return ((size_t)b * (2 * p.Hout) + 2 * oy + p.fold_py) * (size_t)(2 * p.Wout) + 2 * ox + p.fold_px;
  return ((size_t)b * p.Hout + oy) * p.Wout + ox;
}

// 4x4 transpose across a lane quad: on entry lane i of the quad holds a[j] = M[i][j], on exit a[j] = M[j][i] (two butterfly
// stages of quad-permute DPP moves; no LDS).  Used by the epilogue: a lane owns ONE channel of four consecutive pixels, and after
// the transpose it owns FOUR consecutive channels of one pixel - a 16-byte store / residual load instead of four 4-byte ones.
__device__ __forceinline__ void quad_transpose_s1(f32x4& a, int lane) {   // first butterfly stage (lane ^ 1)
  const bool p1 = lane & 1;
  float s0 = p1 ? a[0] : a[1], s1 = p1 ? a[2] : a[3];
  float r0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s0), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  float r1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s1), 0xB1, 0xF, 0xF, true));
  if (p1) { a[0] = r0; a[2] = r1; } else { a[1] = r0; a[3] = r1; }
}
__device__ __forceinline__ void quad_transpose_s2(f32x4& a, int lane) {   // second stage (lane ^ 2)
  const bool p2 = lane & 2;
  float s0 = p2 ? a[0] : a[2], s1 = p2 ? a[1] : a[3];
  float r0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s0), 0x4E, 0xF, 0xF, true));         // quad_perm [2,3,0,1]
  float r1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s1), 0x4E, 0xF, 0xF, true));
  if (p2) { a[0] = r0; a[1] = r1; } else { a[2] = r0; a[3] = r1; }
}
__device__ __forceinline__ void quad_transpose(f32x4& a, int lane) { quad_transpose_s1(a, lane); quad_transpose_s2(a, lane); }
__device__ __forceinline__ float quad_sum(float v) {   // sum over the four lanes of a quad, every lane gets the total
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  return v;
}

// out[m][n] = acc + bias[n] + sbias[b][n] + res[m][n]   (or the GeGLU product), NHWC store.
// When p.stats is set, the workgroup also emits the per-channel sum / sum-of-squares of what it stored, so that the
// GroupNorm that consumes this tensor needs no pass over it (deterministic: lane pair -> LDS -> one writer per channel).
// `red` is LDS scratch of at least 2*NWM*BN floats that no wave is still reading (callers barrier before reuse).
template <int TH, int TW, int BN, int FM, int FN, int NWM = 2>
__device__ __forceinline__ void conv_epilogue(const ConvP& p, f32x16 (&acc)[FM][FN], int b, int oy0, int ox0, int n0,
                                              int wm, int wn, int lane, int tid, float* red, bool active = true) {
  constexpr int WM = TH * TW / NWM, WN = BN / 2;
  unsigned am = 0;            // range telemetry (amax_acc): largest |stored value| of this thread, as bits
  // `active` == false: a wave group of the workgroup that holds no results (intra-workgroup K split, conv_bf16x3.hip) - it only
  // keeps the workgroup's barrier count in step with the active group
  if (!active) {
    if (p.stats && !p.partial) { __syncthreads(); __syncthreads(); }
    return;
  }
  if (p.partial) {   // split-K slice: raw accumulators only
#pragma unroll
    for (int fm = 0; fm < FM; ++fm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int pp = wm * WM + fm * 32 + row;
        const int oy = oy0 + pp / TW, ox = ox0 + pp % TW;
        if (oy >= p.Hout || ox >= p.Wout) continue;
        const size_t m = out_pixel(p, b, oy, ox);
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) {
          const int n = n0 + wn * WN + fn * 32 + (lane & 31);
          if (n < p.N) p.partial[m * p.N + n] = acc[fm][fn][r];
        }
      }
    return;
  }
  // Destination of a plane-pair tile: the generic out_planes tensor, or - for a q|k|v projection tile that lies inside the
  // Q or the K third - that third's [token][C] plane pair (the V third keeps its transposed layout below).
  x3_t* pq = static_cast<x3_t*>(p.out_planes);
  size_t pq_plane = (size_t)p.B * p.Hout * p.Wout * p.ld_out;
  int pq_ld = p.ld_out, pq_n0 = p.geglu ? n0 / 2 : n0, pq_n = p.geglu ? p.N / 2 : p.N;
  if (p.qkv) {
    const int C = p.N / 3, which = n0 / C;
    pq = nullptr;
    if (C % BN == 0 && which < 2) {
      pq_plane = (size_t)p.B * p.Wout * C;
      pq = static_cast<x3_t*>(p.qkv) + (size_t)(2 * which) * pq_plane;
      pq_ld = C; pq_n0 = n0 - which * C; pq_n = C;
    }
  }
  if constexpr (TH == 1) if (pq) {   // linear layers only: keeps this path's registers out of the 3x3 kernels' budget
    const float* sb = p.qkv ? nullptr : sbias_row(p, b);
    const float* resp = p.qkv ? nullptr : p.res;
    // hi/lo plane output (linear layers only, TH == 1): the finished tile is transposed through LDS as fp32 so that the
    // split and the global stores run row-wise - 16 bytes per lane per plane, whole 128-byte lines - instead of one 2-byte
    // store per element from the column-per-lane accumulator layout.
    typedef x3_t x3x8_t __attribute__((ext_vector_type(8)));
    const int L = p.Wout;
    const int bno = p.geglu ? BN / 2 : BN;         // output columns of this tile
    const int pitch = bno + 8;                     // floats; +8 keeps the two half-waves (rows r, r+4) on different banks
    __syncthreads();                               // nobody is still reading the main loop's LDS images
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
      if (p.geglu) {
        if constexpr (FN == 2) {
          const int nv = n0 + wn * WN + (lane & 31);  // packed column of the value half; gate = nv + 32
          const float bv = p.bias ? p.bias[nv] : 0.f, bg = p.bias ? p.bias[nv + 32] : 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rl = wm * WM + fm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            red[rl * pitch + wn * (WN / 2) + (lane & 31)] = (acc[fm][0][r] + bv) * gelu_erf_f(acc[fm][1][r] + bg);
          }
        }
      } else {   // raw accumulators; bias / per-sample bias / residual are added row-wise below
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) {
          const int cl = wn * WN + fn * 32 + (lane & 31);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rl = wm * WM + fm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            red[rl * pitch + cl] = acc[fm][fn][r];
          }
        }
      }
    }
    __syncthreads();
    // row-wise pass: a thread keeps one 8-column chunk (NT is a multiple of the chunks per row) and walks down the rows.
    // All global loads (bias, per-sample bias, residual rows) are issued before the first LDS read so they overlap.
    constexpr int NT = NWM * 128, ITER = TH * TW * (BN / 8) / NT;
    const int cpr = bno / 8;
    const int col = (tid % cpr) * 8, row0 = tid / cpr, rstep = NT / cpr;
    const bool cok = pq_n0 + col < pq_n;
    const int n = n0 + col;
    f32x4 ba = {0.f, 0.f, 0.f, 0.f}, bc = ba;
    if (!p.geglu && cok) {
      if (p.bias) { ba += *reinterpret_cast<const f32x4*>(p.bias + n); bc += *reinterpret_cast<const f32x4*>(p.bias + n + 4); }
      if (sb) { ba += *reinterpret_cast<const f32x4*>(sb + n); bc += *reinterpret_cast<const f32x4*>(sb + n + 4); }
    }
    f32x4 ra[ITER], rc[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int rl = row0 + it * rstep;
      ra[it] = ba; rc[it] = bc;
      if (resp && cok && rl < TH * TW && ox0 + rl < L) {
        const float* rp = resp + ((size_t)b * L + ox0 + rl) * p.ld_res + n;
        ra[it] += *reinterpret_cast<const f32x4*>(rp); rc[it] += *reinterpret_cast<const f32x4*>(rp + 4);
      }
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int rl = row0 + it * rstep;
      if (!cok || rl >= TH * TW || ox0 + rl >= L) continue;
      const f32x4 a = *reinterpret_cast<const f32x4*>(red + rl * pitch + col) + ra[it];
      const f32x4 c = *reinterpret_cast<const f32x4*>(red + rl * pitch + col + 4) + rc[it];
      if (p.amax) { amax_acc4(am, a); amax_acc4(am, c); }
      // whole-vector conversions: the packed v_cvt_pk_bf16_f32 path (element-wise casts fall back to integer rounding code)
      typedef x3_t x3x4_t __attribute__((ext_vector_type(4)));
      const x3x4_t ha = __builtin_convertvector(a, x3x4_t), hc = __builtin_convertvector(c, x3x4_t);
      const x3x4_t la = __builtin_convertvector(a - __builtin_convertvector(ha, f32x4), x3x4_t);
      const x3x4_t lc = __builtin_convertvector(c - __builtin_convertvector(hc, f32x4), x3x4_t);
      const x3x8_t hi = __builtin_shufflevector(ha, hc, 0, 1, 2, 3, 4, 5, 6, 7);
      const x3x8_t lo = __builtin_shufflevector(la, lc, 0, 1, 2, 3, 4, 5, 6, 7);
      const size_t o = ((size_t)b * L + ox0 + rl) * pq_ld + pq_n0 + col;
      *reinterpret_cast<x3x8_t*>(pq + o) = hi;
      *reinterpret_cast<x3x8_t*>(pq + pq_plane + o) = lo;
    }
    amax_flush(p.amax, am);
    return;
  }
  if constexpr (TH == 1) if (p.qkv && (p.N / 3) % BN == 0 && n0 >= 2 * (p.N / 3) && ox0 + TW <= p.Wout) {
    // A tile inside the V third: V^T [head][d][token] runs along the tokens, which the accumulator layout holds four at a
    // time per lane - 8-byte stores scattered over 64 d-rows per instruction.  Transpose the tile through LDS instead
    // ([column][row], so a lane's four consecutive rows are one ds_write_b128) and store token-wise: 16 bytes per lane per
    // plane, 256 contiguous bytes of one d-row per 16 lanes.
    typedef x3_t x3x4_t __attribute__((ext_vector_type(4)));
    typedef x3_t x3x8_t __attribute__((ext_vector_type(8)));
    constexpr int BM = TW, pitchT = BM + 4;   // floats; +4: the 8 lanes of a b128 write group land on distinct banks
    const int C = p.N / 3, L = p.Wout, H = C / 64;
    const size_t MC = (size_t)p.B * L * C;
    x3_t* base = static_cast<x3_t*>(p.qkv) + 4 * MC;
    __syncthreads();
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) {
      const int cl = wn * WN + fn * 32 + (lane & 31);
      const float bn = p.bias ? p.bias[n0 + cl] : 0.f;
#pragma unroll
      for (int fm = 0; fm < FM; ++fm)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          f32x4 v4;
#pragma unroll
          for (int j = 0; j < 4; ++j) v4[j] = acc[fm][fn][4 * rq + j] + bn;
          *reinterpret_cast<f32x4*>(red + cl * pitchT + wm * WM + fm * 32 + 8 * rq + 4 * (lane >> 5)) = v4;
        }
    }
    __syncthreads();
    constexpr int NT = NWM * 128, CPR = BM / 8, ITER = BN * CPR / NT;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int item = tid + it * NT;
      const int cl = item / CPR, c = item % CPR;
      // stored positions 8c..8c+7 of this d-row: the first half of a 16-token block holds source quads 0 and 2, the second 1 and 3
      const float* src = red + cl * pitchT + (c >> 1) * 16 + (c & 1) * 4;
      const f32x4 a = *reinterpret_cast<const f32x4*>(src), q = *reinterpret_cast<const f32x4*>(src + 8);
      if (p.amax) { amax_acc4(am, a); amax_acc4(am, q); }
      const x3x4_t ha = __builtin_convertvector(a, x3x4_t), hq = __builtin_convertvector(q, x3x4_t);
      const x3x4_t la = __builtin_convertvector(a - __builtin_convertvector(ha, f32x4), x3x4_t);
      const x3x4_t lq = __builtin_convertvector(q - __builtin_convertvector(hq, f32x4), x3x4_t);
      const int cc = n0 - 2 * C + cl;
      const size_t o = (((size_t)b * H + cc / 64) * 64 + cc % 64) * L + ox0 + 8 * c;
      *reinterpret_cast<x3x8_t*>(base + o) = __builtin_shufflevector(ha, hq, 0, 1, 2, 3, 4, 5, 6, 7);
      *reinterpret_cast<x3x8_t*>(base + MC + o) = __builtin_shufflevector(la, lq, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    amax_flush(p.amax, am);
    return;
  }
  if constexpr (TH == 1) if (p.qkv) {
    // q|k|v planes for the bf16x3 attention: Q,K [token][C] and V^T [head][d][token] (middle token quads of every
    // 16-token block swapped, see attention_bf3.hip), each as a bf16 hi plane and a bf16 lo = bf16(x - hi) plane.
    typedef x3_t x3x4_t __attribute__((ext_vector_type(4)));
    const int C = p.N / 3, L = p.Wout, H = C / 64;
    const size_t MC = (size_t)p.B * L * C;
    x3_t* base = static_cast<x3_t*>(p.qkv);
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) {
      const int n = n0 + wn * WN + fn * 32 + (lane & 31);
      if (n >= p.N) continue;
      const int which = n / C, cc = n - which * C;
      const float bn = p.bias ? p.bias[n] : 0.f;
#pragma unroll
      for (int fm = 0; fm < FM; ++fm) {
        const int tok0 = ox0 + wm * WM + fm * 32;   // first token of this 32-row fragment (dense mode: TH == 1)
        if (which < 2) {
          x3_t* ph = base + (size_t)(2 * which) * MC + ((size_t)b * L + tok0) * C + cc;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (tok0 + row >= L) continue;
            const float v = acc[fm][fn][r] + bn;
            if (p.amax) amax_acc(am, v);
            const x3_t hi = (x3_t)v;
            ph[(size_t)row * C] = hi;
            ph[MC + (size_t)row * C] = (x3_t)(v - (float)hi);
          }
        } else {
          const int hh = cc / 64, d = cc % 64;
          x3_t* pv = base + 4 * MC + (((size_t)b * H + hh) * 64 + d) * L;
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            const int t0 = tok0 + 8 * rq + 4 * (lane >> 5);   // four consecutive tokens held in registers 4rq..4rq+3
            if (t0 >= L) continue;
            const int qd = (t0 >> 2) & 3;
            const int qp = (qd == 1) ? 2 : (qd == 2 ? 1 : qd);
            f32x4 v4;
#pragma unroll
            for (int j = 0; j < 4; ++j) v4[j] = acc[fm][fn][4 * rq + j] + bn;
            if (p.amax) amax_acc4(am, v4);
            const x3x4_t hi4 = __builtin_convertvector(v4, x3x4_t);
            const x3x4_t lo4 = __builtin_convertvector(v4 - __builtin_convertvector(hi4, f32x4), x3x4_t);
            const size_t off = (size_t)(t0 & ~15) + qp * 4;
            *reinterpret_cast<x3x4_t*>(pv + off) = hi4;
            *reinterpret_cast<x3x4_t*>(pv + MC + off) = lo4;
          }
        }
      }
    }
    amax_flush(p.amax, am);
    return;
  }
  const float* sb = sbias_row(p, b);
  const bool full = (oy0 + TH <= p.Hout) && (ox0 + TW <= p.Wout) && (n0 + BN <= p.N) && !p.geglu;
  float ssum[FN], ssq[FN];
#pragma unroll
  for (int fn = 0; fn < FN; ++fn) { ssum[fn] = 0.f; ssq[fn] = 0.f; }

  const bool wide = full && (p.ld_out & 3) == 0 && (!p.res || (p.ld_res & 3) == 0);   // workgroup-uniform
  if (wide) {
    // Interior tile, 16-byte accesses.  In the accumulator layout a lane owns one channel of 16 pixels (registers 4q..4q+3 = four
    // consecutive pixels): 4-byte stores and residual loads, 16 of each per fragment, and it was their address path - not HBM -
    // that bounded the epilogue (a timing variant moving the same bytes in 16-byte pieces: 72 -> 63 us on the 128x128-level
    // conv).  A 4x4 transpose across each lane quad turns that into four consecutive channels of one pixel per lane: lane j of
    // a quad ends up with pixel 8q + 4h + j, channels 4*(lane/4)..+3 of the fragment.  No LDS, no barrier.
    const int j4 = lane & 3, cq = (lane & 31) & ~3;
    f32x4 s1[FN], s2[FN];
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) { s1[fn] = f32x4{0.f, 0.f, 0.f, 0.f}; s2[fn] = s1[fn]; }
    f32x4 cb4[FN];
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) {
      const int n = n0 + wn * WN + fn * 32 + cq;
      cb4[fn] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.bias) cb4[fn] += *reinterpret_cast<const f32x4*>(p.bias + n);
      if (sb) cb4[fn] += *reinterpret_cast<const f32x4*>(sb + n);
      if (p.bias2) cb4[fn] += *reinterpret_cast<const f32x4*>(p.bias2 + n);
    }
    // one fragment row (FN fragments) at a time: its residual loads back to back, then transposes, adds and stores - bounds the
    // in-flight temporaries to 16*FN registers (the epilogue must not need more VGPRs than the main loop)
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
      size_t m[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int pp = wm * WM + fm * 32 + 8 * q + 4 * (lane >> 5) + j4;
        m[q] = out_pixel(p, b, oy0 + pp / TW, ox0 + pp % TW);
      }
      f32x4 rr[4][FN];
      if (p.res) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int fn = 0; fn < FN; ++fn)
            rr[q][fn] = *reinterpret_cast<const f32x4*>(p.res + m[q] * p.ld_res + n0 + wn * WN + fn * 32 + cq);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) {
          f32x4 v = {acc[fm][fn][4 * q], acc[fm][fn][4 * q + 1], acc[fm][fn][4 * q + 2], acc[fm][fn][4 * q + 3]};
          quad_transpose(v, lane);
          v += cb4[fn];
          if (p.res) v += rr[q][fn];
          *reinterpret_cast<f32x4*>(p.out + m[q] * p.ld_out + n0 + wn * WN + fn * 32 + cq) = v;
          if (p.amax) amax_acc4(am, v);
          s1[fn] += v; s2[fn] += v * v;
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    amax_flush(p.amax, am);
    if (p.stats) {   // workgroup-uniform
      // a lane summed its four channels over pixels j (mod 4) of one half (h): quad + half-wave reduce, then waves through LDS
#pragma unroll
      for (int fn = 0; fn < FN; ++fn)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a = quad_sum(s1[fn][e]), c = quad_sum(s2[fn][e]);
          a += __shfl_xor(a, 32); c += __shfl_xor(c, 32);
          s1[fn][e] = a; s2[fn][e] = c;
        }
      __syncthreads();   // nobody is still reading the main loop's LDS images
      if (lane < 32 && j4 == 0) {
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) {
          float* pr = red + (wm * BN + wn * WN + fn * 32 + cq) * 2;
          *reinterpret_cast<f32x4*>(pr) = f32x4{s1[fn][0], s2[fn][0], s1[fn][1], s2[fn][1]};
          *reinterpret_cast<f32x4*>(pr + 4) = f32x4{s1[fn][2], s2[fn][2], s1[fn][3], s2[fn][3]};
        }
      }
      __syncthreads();
      if (tid < BN && n0 + tid < p.N) {
        const int par = p.fold ? 4 : 1;      // a folded upsampling conv emits one statistics tile per (source tile, parity)
        const int tile = ((oy0 / TH) * p.tiles_x + ox0 / TW) * par + (p.fold ? p.fold_py * 2 + p.fold_px : 0);
        float* dst = p.stats + (((size_t)b * (p.tiles_x * p.tiles_y * par) + tile) * p.N + n0 + tid) * 2;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int i = 0; i < NWM; ++i) { a0 += red[(i * BN + tid) * 2 + 0]; a1 += red[(i * BN + tid) * 2 + 1]; }
        dst[0] = a0; dst[1] = a1;
      }
    }
    return;
  }
  {   // edge tiles, GeGLU to fp32, row strides that are not multiples of four floats: per-element form with bounds checks
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int pp = wm * WM + fm * 32 + row;
        const int oy = oy0 + pp / TW, ox = ox0 + pp % TW;
        if (oy >= p.Hout || ox >= p.Wout) continue;
        const size_t m = out_pixel(p, b, oy, ox);
        if (p.geglu) {
          if (FN == 2) {
            const int nv = n0 + wn * WN + (lane & 31);  // packed column of the value half; gate = nv + 32
            const int j = (n0 + wn * WN) / 2 + (lane & 31);
            if (j < p.N / 2) {
              float v = acc[fm][0][r], g = acc[fm][FN - 1][r];
              if (p.bias) { v += p.bias[nv]; g += p.bias[nv + 32]; }
              const float o_ = v * gelu_erf_f(g);
              if (p.amax) amax_acc(am, o_);
              store_out(p, m * p.ld_out + j, o_);
            }
          }
        } else {
#pragma unroll
          for (int fn = 0; fn < FN; ++fn) {
            const int n = n0 + wn * WN + fn * 32 + (lane & 31);
            if (n < p.N) {
              float v = acc[fm][fn][r];
              if (p.bias) v += p.bias[n];
              if (sb) v += sb[n];
              if (p.bias2) v += p.bias2[n];
              if (p.res) v += p.res[m * p.ld_res + n];
              store_out(p, m * p.ld_out + n, v);
              if (p.amax) amax_acc(am, v);
              ssum[fn] += v; ssq[fn] += v * v;
            }
          }
        }
      }
    }
  }

  amax_flush(p.amax, am);
  if (p.stats) {   // workgroup-uniform
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) { ssum[fn] += __shfl_xor(ssum[fn], 32); ssq[fn] += __shfl_xor(ssq[fn], 32); }
    __syncthreads();   // nobody is still reading the main loop's LDS images
    if (lane < 32) {
#pragma unroll
      for (int fn = 0; fn < FN; ++fn) {
        const int col = wn * WN + fn * 32 + lane;
        red[(wm * BN + col) * 2 + 0] = ssum[fn];
        red[(wm * BN + col) * 2 + 1] = ssq[fn];
      }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < p.N) {
      const int par = p.fold ? 4 : 1;      // a folded upsampling conv emits one statistics tile per (source tile, parity)
      const int tile = ((oy0 / TH) * p.tiles_x + ox0 / TW) * par + (p.fold ? p.fold_py * 2 + p.fold_px : 0);
      float* dst = p.stats + (((size_t)b * (p.tiles_x * p.tiles_y * par) + tile) * p.N + n0 + tid) * 2;
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int i = 0; i < NWM; ++i) { a0 += red[(i * BN + tid) * 2 + 0]; a1 += red[(i * BN + tid) * 2 + 1]; }
      dst[0] = a0; dst[1] = a1;
    }
  }
}

}  // namespace pf
