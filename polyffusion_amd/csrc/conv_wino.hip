// conv_wino.hip - the stride-1 3x3 convolution of a ResBlock (GroupNorm + SiLU prologue, bias / time-bias / residual epilogue,
// GroupNorm tile statistics; ref: stable_diffusion/model/unet.py:262-318) as a FUSED Winograd F(2x2, 3x3) on the split bf16 / fp16
// matrix pipe: 16 channel contractions per 2x2 output pixels instead of 36 - 2.25x fewer MFMAs than the implicit GEMM of
// conv_bf16x3.hip - for the levels where activations, not weights, dominate the traffic (128x128 and 64x64 at B = 16).
//
//      Y = A^T [ (G g G^T) . (B^T d B) ] A         d: 4x4 input patch (stride 2), g: 3x3 filter, Y: 2x2 outputs, "." per (i, j) of
//                                                  the 4x4 transform domain: M[i][j] = sum_c V[i][j][tile][c] * U[i][j][c][n]
//
// One workgroup (4 waves, one per SIMD, 512 registers each) owns 8x8 Winograd tiles = 16x16 output pixels x 64 output channels:
// 16 x 64 x 64 fp32 accumulators = all 256 accumulator registers of its four waves.  Wave i owns ROW i of the transform domain
// (M[i][0..3]) for all 64 tiles x 64 channels - so that
//   * every weight fragment (U, pre-transformed and pre-split on the host, packed in MFMA B-operand order) is read by exactly one wave:
//     it goes global -> registers, never through the LDS;
//   * every input fragment is computed by the wave that consumes it, IN the A-operand layout: the halo image sits in the LDS once, as
//     fp32 after GroupNorm + SiLU (one transcendental pair per input element, as in the direct kernel); a lane reads the two input rows
//     its transform row needs (B^T row i = x +- y), forms t[0..3] = x[b] + sigma y[b], V[i][j] = {t0 - t2, t1 + t2, t2 - t1, t1 - t3}
//     for its 8 channels, splits into hi / lo pieces and feeds the MFMAs - no V planes in the LDS, one ds_read_b128 per 0.75 MFMA;
//   * 12 MFMAs per (j, 16-channel step) reuse each fragment: 2 tile halves x 2 channel halves x {lo.hi, hi.lo, hi.hi}.
// The output transform is separable: each wave applies the column half (S[i][q] = M[i][.] A) in registers, the four row partners
// exchange S through the LDS (128 KB, once per workgroup) and each wave finishes one 32-tile x 32-channel block:
// Y[p][q] = A^T[p][.] S[.][q], + bias + time bias + residual, 16-byte stores after a lane-quad transpose, tile statistics.
#include "conv_common.h"

namespace pf {

typedef x3_t x3x4 __attribute__((ext_vector_type(4)));

#ifdef PF_TRACE
__device__ unsigned long long g_wtrace[3 * 256];
#define WTR() do { if (trace_on && tslot < 255) { g_wtrace[tbase + tslot++] = __builtin_amdgcn_s_memtime(); } } while (0)
#define WTR_DECL() const bool trace_on = (threadIdx.x == 0) && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2 || blockIdx.x == gridDim.x - 1); \
                   const int tbase = blockIdx.x == 0 ? 0 : (blockIdx.x == gridDim.x - 1 ? 512 : 256); int tslot = 0; (void)tbase
#else
#define WTR() do {} while (0)
#define WTR_DECL() do {} while (0)
#endif

// halo image in the LDS: [18 rows][18 pixels][32 channels + 4 pad] fp32, row pitch padded by 8 floats; the 16-byte channel piece of a pixel
// sits at position (piece ^ ((row >> 1) & 1)).  With these three choices the b128 reads of the A-operand layout (lane = tile, stride two
// pixels / two rows) are bank-conflict free in every hardware lane group (tools/micro/wino_banks.py).
constexpr int WPP = 36, WRP = 18 * WPP + 8, WHALO = 18 * WRP;
constexpr int WINO_LDS = 4 * 2 * 4 * 16 * 64 * 4;   // the S exchange (131072 B) is the largest user; halo images: 47232 B each

// workgroup barrier that orders LDS traffic only: __syncthreads() also drains the vector-memory counter, i.e. waits for the residual rows
// requested just before it
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// a pointer every lane holds the same value of, moved to scalar registers (a buffer resource built from a pointer the compiler cannot PROVE
// uniform is legalised with a read-first-lane loop around every load that uses it)
template <class T>
__device__ __forceinline__ const T* uniform_ptr(const T* q) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(q);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const T*>(((unsigned long long)hi << 32) | lo);
}

// buffer resource over a tensor that may be absent: a null pointer gives a resource of size zero - every load through it returns zeros,
// no branch and no wait at the use
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_or_empty(const void* q) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q), 0, q ? 0x7fffffff : 0, 0x00020000);
}
// this sample's row of the per-sample bias, resolved with SCALAR loads (the row index table is read through the constant address space)
__device__ __forceinline__ const float* sbias_row_scalar(const ConvP& p, int b) {
  if (!p.sbias) return nullptr;
  long long r = b;
  if (p.sb_rows) {
    typedef const long long __attribute__((address_space(4))) * cptr;
    r = *((cptr)(unsigned long long)(p.sb_rows + b));
    r = r < 0 ? 0 : (r >= p.sb_nrows ? p.sb_nrows - 1 : r);
  }
  return p.sbias + (size_t)r * p.ld_sbias;
}

// ---- output transform + epilogue.  acc[j][h][nb]: this wave's row of the transform domain.
__device__ __forceinline__ void wino_epilogue(const ConvP& p, f32x16 (&acc)[4][2][2], unsigned char* smem_all, int b, int oy0, int ox0, int n0,
                                              int ty_t, int tx_t, int wave, int lane, int tid, const float* sb, bool trace_on = false, int tbase = 0, int tslot = 0) {
  (void)trace_on; (void)tbase; (void)tslot;
  // Wave w finishes block (h = w >> 1, nb = w & 1).  The exchange image is [slot = (i, q, block)][register r][lane] in single floats: the row of
  // one (slot, r) holds the 32 channels of the tiles m = (r & 3) + 8 (r >> 2) + 4 g at lanes 32 g + channel, so a 16-byte READ at
  // [r][32 g + 4 c] delivers four consecutive channels of one tile - the transposition the 16-byte global stores need happens in the LDS
  // addressing instead of in 21 cross-lane instructions per quad.  Reader lane: c = lane & 7 (channel quad), tl = lane >> 3 (tile column);
  // pass u = tile row: tile m = 8 u + tl  ->  r = (tl & 3) + 4 u, g = (tl >> 2) & 1.
  const int fh = wave >> 1, fnb = wave & 1;
  const int c8 = lane & 7, tl = lane >> 3;
  const int n = n0 + fnb * 32 + c8 * 4;
  const int pix0 = (b * p.Hout + oy0 + 8 * fh) * p.Wout + ox0 + 2 * tl;
  auto pix = [&](int u, int pp, int q) { return pix0 + (2 * u + pp) * p.Wout + q; };
  // the residual rows and biases are requested before anything else (buffer loads: no 64-bit address arithmetic, absent tensors read as
  // zero): their latency hides behind the exchange
  const __amdgpu_buffer_rsrc_t rsRes = rsrc_or_empty(uniform_ptr(p.res)), rsOut = rsrc_or_empty(uniform_ptr(p.out));
  f32x4 rr[4][2][2];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
      for (int q = 0; q < 2; ++q)
        rr[u][pp][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsRes, (pix(u, pp, q) * p.ld_res + n) * 4, 0, 0));
  const f32x4 cba = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_or_empty(uniform_ptr(p.bias)), n * 4, 0, 0));
  const f32x4 cbb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_or_empty(uniform_ptr(sb)), n * 4, 0, 0));
  const f32x4 cbc = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_or_empty(uniform_ptr(p.bias2)), n * 4, 0, 0));
  // ---- output transform, column half in registers: S[q] = M[i][.] A  (A^T = [1 1 1 0; 0 1 -1 -1]), then the exchange
  lds_barrier();              // the halo image is dead
  float* X = reinterpret_cast<float*>(smem_all);
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int blk = h * 2 + nb;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float s0 = acc[0][h][nb][r] + acc[1][h][nb][r] + acc[2][h][nb][r];
        const float s1 = acc[1][h][nb][r] - acc[2][h][nb][r] - acc[3][h][nb][r];
        X[((((wave * 2 + 0) * 4 + blk) * 16 + r) * 64) + lane] = s0;
        X[((((wave * 2 + 1) * 4 + blk) * 16 + r) * 64) + lane] = s1;
      }
      __builtin_amdgcn_sched_barrier(0);   // one block's 64 accumulators in vector registers at a time (all 256 at once spill)
    }
  WTR();
  lds_barrier();
  WTR();
  // Y[0][q] = S0 + S1 + S2, Y[1][q] = S1 - S2 - S3
  const f32x4 cb4 = cba + cbb + cbc;
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
  unsigned am = 0;
  const float* Xr = X + (tl & 3) * 64 + ((tl >> 2) & 1) * 32 + c8 * 4;
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      f32x4 sv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) sv[i] = *reinterpret_cast<const f32x4*>(Xr + ((((i * 2 + q) * 4 + wave) * 16 + 4 * u) * 64));
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {
        f32x4 v = pp == 0 ? sv[0] + sv[1] + sv[2] : sv[1] - sv[2] - sv[3];
        v = PF_X3_UNSCALE(v) + cb4 + rr[u][pp][q];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsOut, (pix(u, pp, q) * p.ld_out + n) * 4, 0, 0);
        if (p.amax) amax_acc4(am, v);
        s1 += v; s2 += v * v;
      }
    }
  WTR();
  amax_flush(p.amax, am);
  if (p.stats) {   // workgroup-uniform
    // a lane summed its four channels over the 16 pixels of its tile column; the 16 partial sums of a channel (8 tile columns x 2 tile
    // halves) meet in the LDS and are added in a fixed order by the channel's thread (cross-lane shuffles cost a round trip each here)
    lds_barrier();            // the exchange area is dead
    float* red = reinterpret_cast<float*>(smem_all);   // [wave][lane][8]
    *reinterpret_cast<f32x4*>(red + (wave * 64 + lane) * 8) = s1;
    *reinterpret_cast<f32x4*>(red + (wave * 64 + lane) * 8 + 4) = s2;
    lds_barrier();
    if (tid < 64) {
      const int nb = tid >> 5, cc = (tid & 31) >> 2, e = tid & 3;
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int t8 = 0; t8 < 8; ++t8) {
          const float* q = red + ((h * 2 + nb) * 64 + t8 * 8 + cc) * 8 + e;
          a0 += q[0]; a1 += q[4];
        }
      const int tile = ty_t * p.tiles_x + tx_t;
      float* dst = p.stats + (((size_t)b * (p.tiles_x * p.tiles_y) + tile) * p.N + n0 + tid) * 2;
      dst[0] = a0; dst[1] = a1;
    }
  }
  WTR();
}

// x * sigmoid(x) on a pair: the multiplies and the add as packed operations (two elements per instruction), one v_exp + one v_rcp per element
typedef float f32x2_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_ silu2(f32x2_ v) {
  f32x2_ t = v * -1.44269504088896340736f;
  t[0] = __builtin_amdgcn_exp2f(t[0]); t[1] = __builtin_amdgcn_exp2f(t[1]);
  t = t + 1.0f;
  t[0] = __builtin_amdgcn_rcpf(t[0]); t[1] = __builtin_amdgcn_rcpf(t[1]);
  return v * t;
}

// packed fp32 arithmetic the compiler does not form by itself here (it emits two scalar operations): a - b and a * b + c on pairs
__device__ __forceinline__ f32x2_ pk_sub(f32x2_ a, f32x2_ b) { f32x2_ d; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ f32x2_ pk_fma(f32x2_ a, f32x2_ b, f32x2_ c) { f32x2_ d; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ f32x4 sub4(f32x4 a, f32x4 b) {
  const f32x2_ l = pk_sub(f32x2_{a[0], a[1]}, f32x2_{b[0], b[1]}), h = pk_sub(f32x2_{a[2], a[3]}, f32x2_{b[2], b[3]});
  return f32x4{l[0], l[1], h[0], h[1]};
}
__device__ __forceinline__ f32x4 fma4(f32x4 a, f32x2_ s2, f32x4 c) {
  const f32x2_ l = pk_fma(f32x2_{a[0], a[1]}, s2, f32x2_{c[0], c[1]}), h = pk_fma(f32x2_{a[2], a[3]}, s2, f32x2_{c[2], c[3]});
  return f32x4{l[0], l[1], h[0], h[1]};
}

// ---- split of four fp32 values into hi | lo pieces.  lo = v - float(hi) is formed by a two-element dot product on the packed hi pair
// (v_dot2c_f32_bf16 / _f16 with the constant (-1, 0) or (0, -1) and v as the accumulator): no unpacking of hi - 8 instructions per four
// elements instead of 10.  v - hi is exactly representable, so the result is the one of the subtract form.
typedef x3_t x3x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// (the two constants come from registers the compiler cannot see through: folded into the instruction they would become an INLINE
// constant, and what "-1.0" means for a packed 16-bit pair operand of this instruction is not something to rely on)
struct SplitK { unsigned c0, c1; };
__device__ __forceinline__ SplitK split_consts() {
#ifdef PF_X3_F16
  SplitK k{0x0000BC00u, 0xBC000000u};      // (-1, 0), (0, -1) as fp16 pairs
#else
  SplitK k{0x0000BF80u, 0xBF800000u};      // ... as bf16 pairs
#endif
  asm volatile("" : "+s"(k.c0), "+s"(k.c1));
  return k;
}
__device__ __forceinline__ float x3_sub_hi(x3x2 hp, unsigned c, float v) {
#ifdef PF_X3_F16
  return __builtin_amdgcn_fdot2(hp, __builtin_bit_cast(x3x2, c), v, false);
#else
  return __builtin_amdgcn_fdot2_f32_bf16(hp, __builtin_bit_cast(x3x2, c), v, false);
#endif
}
__device__ __forceinline__ void split4(f32x4 v, x3x4& hi, x3x4& lo, SplitK k) {
  const x3x2 h01 = __builtin_convertvector(f32x2{v[0], v[1]}, x3x2), h23 = __builtin_convertvector(f32x2{v[2], v[3]}, x3x2);
  const float l0 = x3_sub_hi(h01, k.c0, v[0]), l1 = x3_sub_hi(h01, k.c1, v[1]), l2 = x3_sub_hi(h23, k.c0, v[2]), l3 = x3_sub_hi(h23, k.c1, v[3]);
  const x3x2 q01 = __builtin_convertvector(f32x2{l0, l1}, x3x2), q23 = __builtin_convertvector(f32x2{l2, l3}, x3x2);
  hi = __builtin_shufflevector(h01, h23, 0, 1, 2, 3);
  lo = __builtin_shufflevector(q01, q23, 0, 1, 2, 3);
}

// ---- the pipelined form.  A "step" is one (16-channel half s, tile half h) of a 32-channel chunk: 24 MFMAs (4 columns j of the transform
// domain x 2 channel blocks x {lo.hi, hi.lo, hi.hi}), in column order so that a column's weight fragments die after its 6 MFMAs of the h = 1
// step and are re-loaded right there for the next 16 channels - one whole step before their next use, 64 registers of weights in all.
// Everything else runs in the shadow of those MFMAs, dealt out in small items between them (an in-order wave hides what it issues while
// the matrix pipe is busy): the LDS reads, transform and split of the NEXT step's input fragments (two register sets), the normalise +
// SiLU + LDS write of the next chunk's halo image (two images; three pixels per step, their global loads issued two steps earlier), the
// weight re-loads.  One workgroup barrier per chunk.  All loads are ordinary loads: the compiler's wait counts are exact.
template <int DUMMY>
__global__ __launch_bounds__(256, 1) void conv_wino_pipe_kernel(ConvP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  WTR_DECL();
  WTR();
  // LDS: [4 KB dump | halo image 0 | 4 KB dump | halo image 1 | piece words | scale / shift rows]: a piece outside the image is written to its
  // thread's slot of the dump region in front of the image it belongs to (its place in the image is zeroed once, below)
  constexpr int WIMG = WHALO + 1024;
  float* H0 = reinterpret_cast<float*>(smem_all) + 1024;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = row i of the transform domain

  // tile decode on the scalar unit (every quantity is a function of blockIdx and launch constants; said explicitly - read-first-lane -
  // because the compiler otherwise runs it on the vector unit and then cannot use scalar loads / scalar offsets for what depends on it)
  int lid;
  {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    lid = __builtin_amdgcn_readfirstlane((xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3));
  }
  int mt = __builtin_amdgcn_readfirstlane(fdiv(lid, p.d_nt));
  const int nti = lid - mt * p.nt;
  int t = __builtin_amdgcn_readfirstlane(fdiv(mt, p.d_tx));
  const int tx_t = mt - t * p.tiles_x; mt = t;
  const int b = __builtin_amdgcn_readfirstlane(fdiv(mt, p.d_ty));
  const int ty_t = mt - b * p.tiles_y;
  const int n0 = nti * 64, oy0 = ty_t * 16, ox0 = tx_t * 16;
  conv_shared_x1(p, b);
  const int cin = p.c0 + p.c1, nchunk = cin / 32, KK = cin / 16;
  const float* sbp = sbias_row_scalar(p, b);     // (scalar registers: nothing to keep in a vector register across the loop)

  if (p.gn_s0) gn_fused_prologue<256>(p, b, tid, p.Hin * p.Win, reinterpret_cast<double*>(smem_all));

  // ---- halo staging: piece = (pixel, 4-channel group); a thread keeps one channel group (256 % 8 == 0) and 11 pixels.
  // gp[i] = (pixel index within the sample + 1, 0 = outside the image) | (LDS offset in 16-byte units from 4 KB in front of the image) << 20, kept in the LDS behind the two halo images
  // ([piece][thread]: the registers are needed elsewhere) and read back one item ahead of its use
  const int sub = tid & 7;
  int* gpL = reinterpret_cast<int*>(H0 + WIMG + WHALO) + tid;
  int gpr[11];                                   // register copies for the prologue's loads
  const int bpix = b * p.Hin * p.Win;            // first pixel of this sample
#pragma unroll
  for (int it = 0; it < 11; ++it) {
    const int pix = min((tid >> 3) + 32 * it, 323);           // (the last round's surplus threads repeat pixel 323: same value, same place)
    // (24-bit multiplies throughout: a 32-bit v_mul_lo is a quarter-rate instruction, and this loop is on the critical path to the first load)
    const int row = (int)__umul24(pix, 57) >> 10, col = pix - (int)__umul24(row, 18);          // pix / 18 for pix < 324
    const int iy = oy0 - 1 + row, ix = ox0 - 1 + col;
    const bool inside = (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
    const int gpix = inside ? (int)__umul24(iy, p.Win) + ix + 1 : 0;          // within the sample (the sample's base goes into the scalar offset)
    const int real4 = (int)__umul24(row, WRP / 4) + (int)__umul24(col, WPP / 4) + (sub ^ ((row >> 1) & 1));
    const int lo4 = inside ? 256 + real4 : tid;           // in units of 16 bytes from 4 KB in front of the image
    if (!inside) {                                        // zero padding applies to the ACTIVATED tensor: written once, never overwritten
      *reinterpret_cast<f32x4*>(H0 + real4 * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(H0 + WIMG + real4 * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    gpr[it] = gpix | (lo4 << 20);
    gpL[it * 256] = gpr[it];
  }
  const __amdgpu_buffer_rsrc_t rsSc = dma_resource(uniform_ptr(p.sc + (size_t)b * cin)), rsSh = dma_resource(uniform_ptr(p.sh + (size_t)b * cin));
  // the sample's GroupNorm scale / shift rows, in the LDS behind the piece words: a step reads its eight values right before its staging
  // items and lets them go afterwards (eight registers that the fragment items need in the first two thirds of a step)
  float* scL = reinterpret_cast<float*>(gpL - tid + 11 * 256);
  float* shL = scL + cin;
  const float* px0 = uniform_ptr(p.x0);
  const float* px1 = uniform_ptr(p.x1);
  f32x4 ra[11], vsc, vsh;
  auto scRead = [&](int ct) {
    vsc = *reinterpret_cast<const f32x4*>(scL + ct * 32 + sub * 4);
    vsh = *reinterpret_cast<const f32x4*>(shL + ct * 32 + sub * 4);
  };
  int gq = 0, gst = 0;                            // piece words read one item ahead of their use (an LDS read right before its use stalls the wave)
  auto gpRead = [&](int pi) { gq = gpL[pi * 256]; };
  auto haloFetch = [&](int ct, int gpv, int skip = 0) -> f32x4 {          // global load of the piece with word gpv of chunk ct (skip = 1 << 31: no access)
    const int cg = ct * 32;
    const bool first = cg < p.c0;                // wave-uniform: the source tensor of this chunk
    const __amdgpu_buffer_rsrc_t rs = dma_resource(first ? px0 : px1);
    const int cs4 = (first ? p.c0 : p.c1) * 4, co4 = (first ? cg : cg - p.c0) * 4 + bpix * cs4;
    const int gpix = (gpv & 0xFFFFF) - 1;
    // outside the image (gpix = -1): bit 31 set - an offset >= the resource's size reads as zero, no access (branch-free)
    const int vo = ((int)__umul24(gpix, cs4) + sub * 16) | (gpix & (int)0x80000000u) | skip;
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, co4, 0));
  };
  auto haloLoad = [&](int pi, int ct, int gpv, int skip = 0) { ra[pi] = haloFetch(ct, gpv, skip); };
  auto stageWhole = [&](f32x4 r, f32x4 sc4, f32x4 sh4, int gpv, float* Ht) {     // prologue: a whole piece at once
    r = r * sc4 + sh4;
    const f32x2_ a = silu2(f32x2_{r[0], r[1]}), c2 = silu2(f32x2_{r[2], r[3]});
    *reinterpret_cast<f32x4*>(Ht - 1024 + (((unsigned)gpv >> 20) << 2)) = f32x4{a[0], a[1], c2[0], c2[1]};
  };
  auto scLoad = [&](int ct) {
    vsc = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsSc, sub * 16, ct * 128, 0));
    vsh = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsSh, sub * 16, ct * 128, 0));
  };
  auto stageA = [&](int pi) {                    // first half of a piece: normalise, SiLU of two elements
    gst = gpL[pi * 256];
    const f32x2_ a = silu2(f32x2_{ra[pi][0], ra[pi][1]} * f32x2_{vsc[0], vsc[1]} + f32x2_{vsh[0], vsh[1]});
    ra[pi][0] = a[0]; ra[pi][1] = a[1];
  };
  auto stageB = [&](int pi, float* Ht) {         // second half + the LDS write
    const f32x2_ a = silu2(f32x2_{ra[pi][2], ra[pi][3]} * f32x2_{vsc[2], vsc[3]} + f32x2_{vsh[2], vsh[3]});
    ra[pi][2] = a[0]; ra[pi][3] = a[1];
    *reinterpret_cast<f32x4*>(Ht - 1024 + (((unsigned)gst >> 20) << 2)) = ra[pi];
  };

  // ---- A-operand side: lane = (tile m = lane & 31, channel group g = lane >> 5); B^T row of this wave: t = x + sigma * y
  const int m = lane & 31, g = lane >> 5, tyl = m >> 3, txl = m & 7;
  const int ax = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
  const int ay = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
  const float sigma = wave == 1 ? 1.f : -1.f;
  // float offsets of this lane's first pixel in the x / y row of tile half 0 for channel half 0, with the piece swizzle folded in (it does
  // not depend on the tile half: 4 h is even).  The pixel base is a multiple of 8 floats, so channel half 1 - the neighbouring 16-byte
  // piece - is the same offset ^ 4 (formed where it is used: two registers instead of four); everything else of a fragment read's
  // address - pixel, 16-channel step, tile half - is a compile-time constant: the instruction's offset field.
  int ox_, oy_;
  {
    const int rx = 2 * tyl + ax, ry = 2 * tyl + ay;
    ox_ = rx * WRP + (2 * txl) * WPP + (((2 * g) ^ ((rx >> 1) & 1)) << 2);
    oy_ = ry * WRP + (2 * txl) * WPP + (((2 * g) ^ ((ry >> 1) & 1)) << 2);
    asm volatile("" : "+v"(ox_), "+v"(oy_));   // opaque (the compiler otherwise keeps - and spills - their many parts)
  }
  const SplitK spk = split_consts();
  f32x2_ sig2 = {sigma, sigma};
  asm volatile("" : "+v"(sig2));
  f32x4 rwx[2], rwy[2], tt[4];
  x3x4 fh_[2][4][2], fl_[2][4][2];              // input fragments [set][j][channel half]: hi | lo
  auto fragRead = [&](const float* Hn, int s, int h, int e, int b2) {     // pixels 2 * b2, 2 * b2 + 1 of the two rows
    int ex = ox_, ey = oy_;
    if (e) { asm volatile("v_xor_b32 %0, 4, %1" : "=v"(ex) : "v"(ox_)); asm volatile("v_xor_b32 %0, 4, %1" : "=v"(ey) : "v"(oy_)); }   // (volatile: not hoisted out of the loop)
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
      rwx[bb] = *reinterpret_cast<const f32x4*>(Hn + ex + ((2 * b2 + bb) * WPP + s * 16 + h * 8 * WRP));
      rwy[bb] = *reinterpret_cast<const f32x4*>(Hn + ey + ((2 * b2 + bb) * WPP + s * 16 + h * 8 * WRP));
    }
  };
  auto fragT = [&](int b2) {
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) tt[2 * b2 + bb] = fma4(rwy[bb], sig2, rwx[bb]);
  };
  auto fragV = [&](int set, int j, int e) {
    const f32x4 v = j == 0 ? sub4(tt[0], tt[2]) : (j == 1 ? tt[1] + tt[2] : (j == 2 ? sub4(tt[2], tt[1]) : sub4(tt[1], tt[3])));
    split4(v, fh_[set][j][e], fl_[set][j][e], spk);
  };

  // weights: [i][kk][ntile][j][nb][plane][lane][8]: 16 KB per (i, kk, ntile)
  const __amdgpu_buffer_rsrc_t rsW = dma_resource(uniform_ptr(static_cast<const x3_t*>(p.w) + ((size_t)wave * KK * p.nt + nti) * 8192));
  const int wstep = p.nt * 16384;                // bytes between consecutive 16-channel steps
  x3x8 bh[4][2], bl[4][2];
  auto wLoad = [&](int j, int kk) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      bh[j][nb] = __builtin_bit_cast(x3x8, __builtin_amdgcn_raw_buffer_load_b128(rsW, lane * 16, kk * wstep + ((j * 2 + nb) * 2 + 0) * 1024, 0));
      bl[j][nb] = __builtin_bit_cast(x3x8, __builtin_amdgcn_raw_buffer_load_b128(rsW, lane * 16, kk * wstep + ((j * 2 + nb) * 2 + 1) * 1024, 0));
    }
  };

  auto wLoad1 = [&](int j, int nb, bool lo, int kk, int skip) {
    if (lo) bl[j][nb] = __builtin_bit_cast(x3x8, __builtin_amdgcn_raw_buffer_load_b128(rsW, (lane * 16) | skip, kk * wstep + ((j * 2 + nb) * 2 + 1) * 1024, 0));
    else bh[j][nb] = __builtin_bit_cast(x3x8, __builtin_amdgcn_raw_buffer_load_b128(rsW, (lane * 16) | skip, kk * wstep + ((j * 2 + nb) * 2 + 0) * 1024, 0));
  };
  f32x16 acc[4][2][2];

#define SB() __builtin_amdgcn_sched_barrier(0)
  // ---- prologue: every global load the first steps depend on goes out first (weights of the first 16 channels, the rows of scale /
  // shift, the halo of chunk 0 and the first three pixels of chunk 1); the accumulators are cleared while they fly; then the image of
  // chunk 0, the three pixels, the loads of the next six, the barrier, and the fragments of step 0
  {
    const int c1 = min(1, nchunk - 1);
#pragma unroll
    for (int pi = 0; pi < 11; ++pi) haloLoad(pi, 0, gpr[pi]);   // (the longest latency first)
    scLoad(0);
    const f32x4 vsc0 = vsc, vsh0 = vsh;
    f32x4 rb[3];
#pragma unroll
    for (int pi = 0; pi < 3; ++pi) rb[pi] = haloFetch(c1, gpr[pi]);
    scLoad(c1);
#pragma unroll
    for (int j = 0; j < 4; ++j) wLoad(j, 0);
    float tsc[4], tsh[4];                        // this thread's share of the rows (cin <= 1024)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = min(tid + 256 * i, cin - 1);
      tsc[i] = p.sc[(size_t)b * cin + c]; tsh[i] = p.sh[(size_t)b * cin + c];
    }
    WTR();
    SB();
    // the accumulators are cleared by the matrix pipe itself (0 x 0 + 0: 16 instructions that run beside the vector work, instead of 256
    // register writes in front of it)
    {
      x3x8 z8;
#pragma unroll
      for (int e = 0; e < 8; ++e) z8[e] = (x3_t)0.f;
      asm volatile("" : "+v"(z8));
      const f32x16 z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            asm volatile("" : "+v"(z8));       // (sixteen distinct instructions: identical ones are merged into one plus 240 register copies)
            acc[j][h][nb] = x3_mfma_32x32x16(z8, z8, z16, 0, 0, 0);
          }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (tid + 256 * i < cin) { scL[tid + 256 * i] = tsc[i]; shL[tid + 256 * i] = tsh[i]; }
    SB();
#pragma unroll
    for (int pi = 0; pi < 11; ++pi) stageWhole(ra[pi], vsc0, vsh0, gpr[pi], H0);
    WTR();
#pragma unroll
    for (int pi = 0; pi < 3; ++pi) stageWhole(rb[pi], vsc, vsh, gpr[pi], H0 + WIMG);
#pragma unroll
    for (int pi = 3; pi < 9; ++pi) haloLoad(pi, c1, gpr[pi]);
    WTR();
    __syncthreads();
    WTR();
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      fragRead(H0, 0, 0, e, 0); fragT(0); fragRead(H0, 0, 0, e, 1); fragT(1);
#pragma unroll
      for (int j = 0; j < 4; ++j) fragV(0, j, e);
    }
  }

  for (int chunk = 0; chunk < nchunk; ++chunk) {
    const int c1 = min(chunk + 1, nchunk - 1), c2 = min(chunk + 2, nchunk - 1);
    // loads for chunks past the last one: issued all the same (branch-free steps), with an offset beyond the resource - no memory access
    const int skip1 = chunk + 1 < nchunk ? 0 : (int)0x80000000u, skip2 = chunk + 2 < nchunk ? 0 : (int)0x80000000u;
    float* Hcur = H0 + (chunk & 1) * WIMG;
    float* Hnxt = H0 + ((chunk + 1) & 1) * WIMG;
    static_for<0, 4>([&](auto qc) {
      constexpr int QS = decltype(qc)::value, s = QS >> 1, h = QS & 1;
      constexpr int QN = (QS + 1) & 3, sn = QN >> 1, hn = QN & 1;       // the step whose fragments are produced now, into set hn
      const float* Hn = QS == 3 ? Hnxt : Hcur;
      float* Ht = QS == 3 ? Hcur : Hnxt;                                // image written by this step's staging items
      // staging pieces of this step (first, count) and the pieces whose global loads it issues (first, count, chunk)
      constexpr int sp0 = QS == 3 ? 0 : 3 * QS + 3, spn = QS == 2 ? 2 : 3;
      constexpr int lp0 = QS == 0 ? 9 : 3 * (QS - 1), lpn = QS == 0 ? 2 : 3;
      const int lct = QS == 0 ? c1 : c2;
      WTR();
      if constexpr (QS == 3) {
        // all staging writes of the next image are done and visible, every wave has read the last fragments of the current one
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      SB();
      static_for<0, 24>([&](auto kc) {
        constexpr int k = decltype(kc)::value, j = k / 6, kind = (k % 6) >> 1, nb = k & 1;
        const x3x8 ah = __builtin_shufflevector(fh_[h][j][0], fh_[h][j][1], 0, 1, 2, 3, 4, 5, 6, 7);
        const x3x8 al = __builtin_shufflevector(fl_[h][j][0], fl_[h][j][1], 0, 1, 2, 3, 4, 5, 6, 7);
        if constexpr (kind == 0) acc[j][h][nb] = x3_mfma_32x32x16(al, bh[j][nb], acc[j][h][nb], 0, 0, 0);
        else if constexpr (kind == 1) acc[j][h][nb] = x3_mfma_32x32x16(ah, bl[j][nb], acc[j][h][nb], 0, 0, 0);
        else acc[j][h][nb] = x3_mfma_32x32x16(ah, bh[j][nb], acc[j][h][nb], 0, 0, 0);
        SB();
        // ---- the item of this slot
        // next step's fragments, channel half e = 0 in slots 0-7, e = 1 in slots 8-15: read two pixels of both rows, t, read the other two, t, V x 4
        if constexpr (k == 0 || k == 8) fragRead(Hn, sn, hn, k / 8, 0);
        if constexpr (k == 2 || k == 10) { fragT(0); fragRead(Hn, sn, hn, k / 8, 1); }
        if constexpr (k == 4 || k == 12) { fragT(1); fragV(hn, 0, k / 8); }
        if constexpr (k == 5 || k == 6 || k == 7) fragV(hn, k - 4, 0);
        if constexpr (k == 13 || k == 14 || k == 15) fragV(hn, k - 12, 1);
        if constexpr (k >= 16 && k < 16 + 2 * spn) {
          constexpr int pi = sp0 + (k - 16) / 2;
          if constexpr (((k - 16) & 1) == 0) stageA(pi); else stageB(pi, Ht);
        }
        if constexpr (k == 3 || k == 9 || k == 11) {
          constexpr int li = k == 3 ? 0 : (k == 9 ? 1 : 2);
          if constexpr (li < lpn) haloLoad(lp0 + li, lct, gq, QS == 0 ? skip1 : skip2);
        }
        if constexpr (k == 1 || k == 7 || k == 9) {            // piece word, one slot ahead of the load that needs it
          constexpr int li = k == 1 ? 0 : (k == 7 ? 1 : 2);
          if constexpr (li < lpn) gpRead(lp0 + li);
        }
        if constexpr (k == 15) scRead(QS == 3 ? c2 : c1);
        // a weight fragment is re-loaded for the next 16 channels right after its last MFMA (lo plane: the hi.lo product, hi plane: hi.hi):
        // one load per slot - four in a row stall the wave on the address unit the four waves share
        if constexpr (h == 1 && (k % 6) >= 2) wLoad1(j, k & 1, (k % 6) < 4, s == 0 ? chunk * 2 + 1 : c1 * 2, s == 0 ? 0 : skip1);
        SB();
      });
    });
  }
#undef SB
  WTR();
#ifdef PF_TRACE
  wino_epilogue(p, acc, smem_all, b, oy0, ox0, n0, ty_t, tx_t, wave, lane, tid, sbp, trace_on, tbase, tslot);
#else
  wino_epilogue(p, acc, smem_all, b, oy0, ox0, n0, ty_t, tx_t, wave, lane, tid, sbp);
#endif
}

bool conv_wino_eligible(const pf_conv_args& a) {
  return a.wino > 0 && a.w_wino && a.precision == PF_PREC_BF16X3 && a.ks == 3 && a.stride == 1 && !a.ups && !a.ups_fold && a.prologue == 1 &&
         a.hin % 16 == 0 && a.win % 16 == 0 && a.n % 64 == 0 && a.c0 % 32 == 0 && a.c1 % 32 == 0 && !a.geglu && !a.out_planes && !a.qkv_planes &&
         a.c0 + a.c1 <= 1024 && a.hin * a.win < (1 << 20) &&
         (long long)a.batch * a.hin * a.win * (a.c0 > a.c1 ? a.c0 : a.c1) * 4 < (1ll << 31) &&      // 32-bit buffer offsets
         (long long)a.batch * a.hin * a.win * (a.ld_out > a.ld_res ? a.ld_out : a.ld_res) * 4 < (1ll << 31) &&
         !a.skip_w && (a.ld_out & 3) == 0 && (!a.res || (a.ld_res & 3) == 0);
}

int launch_conv_wino(const pf_conv_args& a, hipStream_t stream) {
  ConvP p;
  memset(&p, 0, sizeof p);
  p.x0 = a.x0; p.x1 = a.x1; p.c0 = a.c0; p.c1 = a.c1;
  p.B = a.batch; p.Hin = a.hin; p.Win = a.win; p.Hout = a.hin; p.Wout = a.win;
  p.w = a.w_wino; p.N = a.n; p.Npad = a.n;
  p.sc = a.sc; p.sh = a.sh;
  p.bias = a.bias; p.sbias = a.sbias; p.ld_sbias = a.ld_sbias; p.res = a.res; p.ld_res = a.ld_res;
  p.sb_rows = reinterpret_cast<const long long*>(a.sbias_rows); p.sb_nrows = a.sbias_nrows;
  p.out = a.out; p.ld_out = a.ld_out; p.stats = a.stats_out;
  p.ksplit = 1;
  if (a.gn_stats0) {
    p.gn_s0 = a.gn_stats0; p.gn_t0 = a.gn_tiles0; p.gn_s1 = a.gn_stats1; p.gn_t1 = a.gn_tiles1;
    p.gn_gamma = a.gn_gamma; p.gn_beta = a.gn_beta; p.gn_eps = a.gn_eps; p.gn_groups = a.gn_groups;
  }
  p.x1_bmod = a.x1_bmod;
  p.amax = static_cast<unsigned*>(a.absmax_slot);
  p.tiles_x = a.win / 16; p.tiles_y = a.hin / 16; p.nt = a.n / 64;
  conv_fill_divs(p);
  const int grid = p.B * p.tiles_y * p.tiles_x * p.nt;
  auto kern = conv_wino_pipe_kernel<0>;
  static std::atomic<uint64_t> attr_done{0};
  if (int rc = set_max_lds_once(reinterpret_cast<const void*>(kern), WINO_LDS, attr_done)) return rc;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), WINO_LDS, stream, p);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}

#ifdef PF_TRACE
extern "C" int pf_debug_wino_trace_read(unsigned long long* dst, int n) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wtrace), (size_t)n * 8) == hipSuccess ? 0 : -1;
}
extern "C" int pf_debug_wino_trace_clear() {
  static unsigned long long z[3 * 256];
  return hipMemcpyToSymbol(HIP_SYMBOL(g_wtrace), z, sizeof z) == hipSuccess ? 0 : -1;
}
#endif

// host: fp32 torch weight [N][K][3][3] -> U = G g G^T per (n, k), split into hi | lo pieces, in MFMA B-operand order
// [i][K/16][N/64][j][nb][plane][lane = (k % 16 / 8) * 32 + n % 32][k % 8]   (N % 64 == 0, K % 16 == 0)
static inline unsigned short wf2bf_rne(float f) {
  unsigned int u; memcpy(&u, &f, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
static inline float wbf2f(unsigned short h) { unsigned int u = (unsigned int)h << 16; float f; memcpy(&f, &u, 4); return f; }

bool pack_wino_bf3(void* dst_, const float* src, int N, int K) {
  unsigned short* dst = (unsigned short*)dst_;
  bool fits = true;
  const int KK = K / 16, NT = N / 64;
  static const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k) {
      const float* g = src + ((size_t)n * K + k) * 9;
      double Gg[4][3];
      for (int i = 0; i < 4; ++i)
        for (int c = 0; c < 3; ++c) Gg[i][c] = G[i][0] * g[0 * 3 + c] + G[i][1] * g[1 * 3 + c] + G[i][2] * g[2 * 3 + c];
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
          const double ud = Gg[i][0] * G[j][0] + Gg[i][1] * G[j][1] + Gg[i][2] * G[j][2];
          const float v = (float)ud;
#ifdef PF_X3_F16
          const float vs = fminf(fmaxf(v * PF_X3_WS, -65504.f), 65504.f);
          fits = fits && vs == v * PF_X3_WS;
          const _Float16 fh = (_Float16)vs, fl = (_Float16)(vs - (float)fh);
          unsigned short hi, lo;
          memcpy(&hi, &fh, 2); memcpy(&lo, &fl, 2);
#else
          const unsigned short hi = wf2bf_rne(v);
          const unsigned short lo = wf2bf_rne(v - wbf2f(hi));
#endif
          const int kk = k / 16, ln = ((k % 16) / 8) * 32 + (n % 32), e = k % 8, ntile = n / 64, nb = (n % 64) / 32;
          const size_t frag = ((((size_t)i * KK + kk) * NT + ntile) * 4 + j) * 2 + nb;
          dst[((frag * 2 + 0) * 64 + ln) * 8 + e] = hi;
          dst[((frag * 2 + 1) * 64 + ln) * 8 + e] = lo;
        }
    }
  return fits;
}

}  // namespace pf
