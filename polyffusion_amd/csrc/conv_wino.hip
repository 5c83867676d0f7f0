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

// halo image in the LDS: [18 rows][18 pixels][32 channels + 4 pad] fp32, row pitch padded by 8 floats; the 16-byte channel piece of a pixel
// sits at position (piece ^ ((row >> 1) & 1)).  With these three choices the b128 reads of the A-operand layout (lane = tile, stride two
// pixels / two rows) are bank-conflict free in every hardware lane group (tools/micro/wino_banks.py).
constexpr int WPP = 36, WRP = 18 * WPP + 8, WHALO = 18 * WRP;
constexpr int WINO_LDS = 4 * 2 * 4 * 16 * 64 * 4;   // the S exchange (131072 B) is the largest user; halo images: 47232 B each

template <int DUMMY>
__global__ __launch_bounds__(256, 1) void conv_wino_kernel(ConvP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  float* H0 = reinterpret_cast<float*>(smem_all);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = row i of the transform domain

  int lid;
  {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  int mt = fdiv(lid, p.d_nt);
  const int nti = lid - mt * p.nt;
  int t = fdiv(mt, p.d_tx);
  const int tx_t = mt - t * p.tiles_x; mt = t;
  const int b = fdiv(mt, p.d_ty);
  const int ty_t = mt - b * p.tiles_y;
  const int n0 = nti * 64, oy0 = ty_t * 16, ox0 = tx_t * 16;
  conv_shared_x1(p, b);
  const int cin = p.c0 + p.c1, nchunk = cin / 32, KK = cin / 16;

  if (p.gn_s0) gn_fused_prologue<256>(p, b, tid, p.Hin * p.Win, reinterpret_cast<double*>(smem_all));

  // ---- halo staging: piece = (pixel, 4-channel group); a thread keeps one channel group (256 % 8 == 0) and 11 pixels
  const int sub = tid & 7;
  int goff[11], loff[11];
#pragma unroll
  for (int it = 0; it < 11; ++it) {
    const int pix = (tid >> 3) + 32 * it;
    const int row = pix / 18, col = pix - row * 18;
    const int iy = oy0 - 1 + row, ix = ox0 - 1 + col;
    const bool valid = pix < 324;
    goff[it] = (valid && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win) ? (b * p.Hin + iy) * p.Win + ix : -1;
    loff[it] = valid ? row * WRP + col * WPP + ((sub ^ ((row >> 1) & 1)) << 2) : -1;
  }
  auto stage = [&](int chunk) {
    const int cg = chunk * 32;
    const float* src; int cs, co;
    if (cg < p.c0) { src = p.x0; cs = p.c0; co = cg; } else { src = p.x1; cs = p.c1; co = cg - p.c0; }
    const f32x4 vsc = *reinterpret_cast<const f32x4*>(p.sc + (size_t)b * cin + cg + sub * 4);
    const f32x4 vsh = *reinterpret_cast<const f32x4*>(p.sh + (size_t)b * cin + cg + sub * 4);
    f32x4 ra[11];
#pragma unroll
    for (int it = 0; it < 11; ++it)
      ra[it] = goff[it] >= 0 ? *reinterpret_cast<const f32x4*>(src + co + (size_t)goff[it] * cs + sub * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < 11; ++it) {
      f32x4 v = ra[it] * vsc + vsh;
      v[0] = silu_f(v[0]); v[1] = silu_f(v[1]); v[2] = silu_f(v[2]); v[3] = silu_f(v[3]);
      if (goff[it] < 0) v = f32x4{0.f, 0.f, 0.f, 0.f};     // zero padding applies to the ACTIVATED tensor
      if (loff[it] >= 0) *reinterpret_cast<f32x4*>(H0 + loff[it]) = v;
    }
  };

  // ---- A-operand side: lane = (tile m = lane & 31, channel group g = lane >> 5); B^T row of this wave: t = x + sigma * y
  const int m = lane & 31, g = lane >> 5, tyl = m >> 3, txl = m & 7;
  const int ax = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
  const int ay = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
  const float sigma = wave == 1 ? 1.f : -1.f;
  // weights: [i][kk][ntile][j][nb][plane][lane][8]
  const x3_t* wI = static_cast<const x3_t*>(p.w) + (size_t)wave * KK * p.nt * 8192 + (size_t)nti * 8192 + lane * 8;

  f32x16 acc[4][2][2];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][h][nb][r] = 0.f;

  for (int chunk = 0; chunk < nchunk; ++chunk) {
    __syncthreads();          // every wave has finished reading the previous image (and the GroupNorm scratch)
    stage(chunk);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const x3_t* wq = wI + (size_t)(chunk * 2 + s) * p.nt * 8192;
      x3x8 bh[4][2], bl[4][2];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          bh[j][nb] = *reinterpret_cast<const x3x8*>(wq + ((j * 2 + nb) * 2 + 0) * 512);
          bl[j][nb] = *reinterpret_cast<const x3x8*>(wq + ((j * 2 + nb) * 2 + 1) * 512);
        }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int rx = 2 * (4 * h + tyl) + ax, ry = 2 * (4 * h + tyl) + ay;
        const float* px = H0 + rx * WRP + (2 * txl) * WPP + s * 16;
        const float* py = H0 + ry * WRP + (2 * txl) * WPP + s * 16;
        const int sx = (rx >> 1) & 1, sy = (ry >> 1) & 1;
        f32x4 tt[4][2];
#pragma unroll
        for (int bb = 0; bb < 4; ++bb)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const f32x4 X = *reinterpret_cast<const f32x4*>(px + bb * WPP + (((2 * g + e) ^ sx) << 2));
            const f32x4 Y = *reinterpret_cast<const f32x4*>(py + bb * WPP + (((2 * g + e) ^ sy) << 2));
            tt[bb][e] = X + sigma * Y;
          }
        x3x8 ah[4], al[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          x3x4 hq[2], lq[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const f32x4 v = j == 0 ? tt[0][e] - tt[2][e] : (j == 1 ? tt[1][e] + tt[2][e] : (j == 2 ? tt[2][e] - tt[1][e] : tt[1][e] - tt[3][e]));
            hq[e] = __builtin_convertvector(v, x3x4);
            lq[e] = __builtin_convertvector(v - __builtin_convertvector(hq[e], f32x4), x3x4);
          }
          ah[j] = __builtin_shufflevector(hq[0], hq[1], 0, 1, 2, 3, 4, 5, 6, 7);
          al[j] = __builtin_shufflevector(lq[0], lq[1], 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            acc[j][h][nb] = x3_mfma_32x32x16(al[j], bh[j][nb], acc[j][h][nb], 0, 0, 0);
            acc[j][h][nb] = x3_mfma_32x32x16(ah[j], bl[j][nb], acc[j][h][nb], 0, 0, 0);
            acc[j][h][nb] = x3_mfma_32x32x16(ah[j], bh[j][nb], acc[j][h][nb], 0, 0, 0);
          }
      }
    }
  }

  // ---- output transform, column half in registers: S[q] = M[i][.] A  (A^T = [1 1 1 0; 0 1 -1 -1]), then the exchange
  __syncthreads();            // the halo image is dead
  float* X = reinterpret_cast<float*>(smem_all);   // [(i*2+q)*4 + blk][r/4][lane][4]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int blk = h * 2 + nb;
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        f32x4 s0, s1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = rq * 4 + e;
          s0[e] = acc[0][h][nb][r] + acc[1][h][nb][r] + acc[2][h][nb][r];
          s1[e] = acc[1][h][nb][r] - acc[2][h][nb][r] - acc[3][h][nb][r];
        }
        *reinterpret_cast<f32x4*>(X + ((((wave * 2 + 0) * 4 + blk) * 4 + rq) * 64 + lane) * 4) = s0;
        *reinterpret_cast<f32x4*>(X + ((((wave * 2 + 1) * 4 + blk) * 4 + rq) * 64 + lane) * 4) = s1;
      }
    }
  __syncthreads();
  // wave w finishes block (h = w >> 1, nb = w & 1): Y[0][q] = S0 + S1 + S2, Y[1][q] = S1 - S2 - S3
  const int fh = wave >> 1, fnb = wave & 1;
  f32x4 yv[2][2][4];          // [p][q][register quad]
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      f32x4 sv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) sv[i] = *reinterpret_cast<const f32x4*>(X + ((((i * 2 + q) * 4 + wave) * 4 + rq) * 64 + lane) * 4);
      yv[0][q][rq] = sv[0] + sv[1] + sv[2];
      yv[1][q][rq] = sv[1] - sv[2] - sv[3];
    }
  // registers 4u..4u+3 of a block = tiles (tyl = u, txl = 4 * (lane >> 5) + 0..3), channel lane & 31; after the lane-quad transpose a
  // lane owns tile txl = 4 * (lane >> 5) + (lane & 3) and the four channels cq..cq+3
  const int j4 = lane & 3, cq = (lane & 31) & ~3;
  const int n = n0 + fnb * 32 + cq;
  const float* sb = sbias_row(p, b);
  f32x4 cb4 = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) cb4 += *reinterpret_cast<const f32x4*>(p.bias + n);
  if (sb) cb4 += *reinterpret_cast<const f32x4*>(sb + n);
  if (p.bias2) cb4 += *reinterpret_cast<const f32x4*>(p.bias2 + n);
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    size_t mo[2][2];
    f32x4 rr[2][2];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        mo[pp][q] = out_pixel(p, b, oy0 + 2 * (4 * fh + u) + pp, ox0 + 2 * (4 * (lane >> 5) + j4) + q);
        if (p.res) rr[pp][q] = *reinterpret_cast<const f32x4*>(p.res + mo[pp][q] * p.ld_res + n);
      }
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        f32x4 v = yv[pp][q][u];
        quad_transpose(v, lane);
        v = PF_X3_UNSCALE(v) + cb4;
        if (p.res) v += rr[pp][q];
        *reinterpret_cast<f32x4*>(p.out + mo[pp][q] * p.ld_out + n) = v;
        s1 += v; s2 += v * v;
      }
  }
  if (p.stats) {   // workgroup-uniform
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a = quad_sum(s1[e]), c = quad_sum(s2[e]);
      a += __shfl_xor(a, 32); c += __shfl_xor(c, 32);
      s1[e] = a; s2[e] = c;
    }
    __syncthreads();          // the exchange area is dead
    float* red = reinterpret_cast<float*>(smem_all);   // [h][64 channels][2]
    if (lane < 32 && j4 == 0) {
      float* pr = red + (fh * 64 + fnb * 32 + cq) * 2;
      *reinterpret_cast<f32x4*>(pr) = f32x4{s1[0], s2[0], s1[1], s2[1]};
      *reinterpret_cast<f32x4*>(pr + 4) = f32x4{s1[2], s2[2], s1[3], s2[3]};
    }
    __syncthreads();
    if (tid < 64) {
      const int tile = ty_t * p.tiles_x + tx_t;
      float* dst = p.stats + (((size_t)b * (p.tiles_x * p.tiles_y) + tile) * p.N + n0 + tid) * 2;
      dst[0] = red[tid * 2] + red[(64 + tid) * 2];
      dst[1] = red[tid * 2 + 1] + red[(64 + tid) * 2 + 1];
    }
  }
}

bool conv_wino_eligible(const pf_conv_args& a) {
  return a.wino > 0 && a.w_wino && a.precision == PF_PREC_BF16X3 && a.ks == 3 && a.stride == 1 && !a.ups && !a.ups_fold && a.prologue == 1 &&
         a.hin % 16 == 0 && a.win % 16 == 0 && a.n % 64 == 0 && a.c0 % 32 == 0 && a.c1 % 32 == 0 && !a.geglu && !a.out_planes && !a.qkv_planes &&
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
  p.tiles_x = a.win / 16; p.tiles_y = a.hin / 16; p.nt = a.n / 64;
  conv_fill_divs(p);
  auto kern = conv_wino_kernel<0>;
  static std::atomic<uint64_t> attr_done{0};
  if (int rc = set_max_lds_once(reinterpret_cast<const void*>(kern), WINO_LDS, attr_done)) return rc;
  const int grid = p.B * p.tiles_y * p.tiles_x * p.nt;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), WINO_LDS, stream, p);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}

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
