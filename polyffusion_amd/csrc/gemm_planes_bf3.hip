// gemm_planes_bf3.hip - Linear / 1x1 convolution whose A operand is ALREADY split into bf16 hi/lo planes
// (written by the producing kernel's epilogue: attention output, GeGLU product, FF output), bf16x3 split MFMA.
//
// With no prologue arithmetic left, both operands go global->LDS directly (global_load_lds_dwordx4): the main loop is
// nothing but a 3-stage ring of {A tile 128x32 hi|lo, W tile 32xBN hi|lo}, one barrier per 32-deep K step with
// counted vmcnt (one stage stays in flight across the barrier), and 6*FM*FN MFMAs per step.  A rows are 64 B in LDS;
// the 16-byte slot is XOR-swizzled with (row>>2)&3 on the source address and on the fragment read, which makes
// ds_read_b128 over 32 consecutive rows conflict-free.  Epilogue = conv_common.h (bias / residual / per-sample bias /
// GroupNorm statistics / fp32 or split-plane output).
#include "conv_common.h"

namespace pf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void gemm_planes_kernel(ConvP p) {
  constexpr int BK = 32, RING = 3, D = RING - 1;
  constexpr int WM = BM / 2, WN = BN / 2, FM = WM / 32, FN = WN / 32;
  constexpr int AU = 2 * BM * 4;            // 16-byte units of one A stage (2 planes x BM rows x 4 slots)
  constexpr int WU = 8 * BN;                // 16-byte units of one W stage (4 k8 x 2 planes x BN)
  constexpr int NAu = AU / 256, NWu = WU / 256;
  constexpr int STAGE = (AU + WU) * 8;      // bf16 elements per stage
  static_assert(AU % 256 == 0 && WU % 256 == 0, "tile must divide over the block");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* sm = reinterpret_cast<__bf16*>(smem_raw);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nwg = gridDim.x;
  int lid;
  {
    const int orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  const int nti = lid % p.nt;
  int mt = lid / p.nt;
  const int tx = mt % p.tiles_x;
  const int b = mt / p.tiles_x;
  const int n0 = nti * BN, ox0 = tx * BM;
  const int K = p.c0, L = p.Wout;
  const size_t MK = (size_t)p.B * L * K;
  const __bf16* A = reinterpret_cast<const __bf16*>(p.x0);

  // per-thread source pointers of the stage pieces (a chunk adds a uniform offset)
  const __bf16* ga[NAu];
#pragma unroll
  for (int j = 0; j < NAu; ++j) {
    const int u = tid + j * 256;
    const int plane = u / (BM * 4), w = u % (BM * 4), row = w >> 2, slot = w & 3;
    const int srow = min(ox0 + row, L - 1);                       // rows past the sample are never stored
    ga[j] = A + (size_t)plane * MK + ((size_t)b * L + srow) * K + ((slot ^ ((row >> 2) & 3)) * 8);
  }
  const __bf16* gw[NWu];
#pragma unroll
  for (int j = 0; j < NWu; ++j) {
    const int u = tid + j * 256;
    const int k8l = u / (2 * BN), plane = (u / BN) & 1, n = u % BN;
    gw[j] = static_cast<const __bf16*>(p.w) + ((size_t)(k8l * 2 + plane) * p.Npad + n0 + n) * 8;
  }
  const size_t wrow = (size_t)2 * p.Npad * 8;
  auto issue = [&](int chunk, int stage) {
    __bf16* sb = sm + stage * STAGE;
#pragma unroll
    for (int j = 0; j < NAu; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga[j] + chunk * BK),
                                       (__attribute__((address_space(3))) void*)(sb + (wave * 64 + j * 256) * 8), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < NWu; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gw[j] + (size_t)chunk * 4 * wrow),
                                       (__attribute__((address_space(3))) void*)(sb + AU * 8 + (wave * 64 + j * 256) * 8), 16, 0, 0);
  };

  int aunit[FM];
#pragma unroll
  for (int fm = 0; fm < FM; ++fm) aunit[fm] = (wm * WM + fm * 32 + (lane & 31)) * 4;
  const int asw = ((lane & 31) >> 2) & 3;   // (row>>2)&3 of this lane's row (tile rows are multiples of 32 apart)
  const int wbase = ((lane >> 5) * 2 * BN + wn * WN + (lane & 31)) * 8;

  f32x16 acc[FM][FN];
#pragma unroll
  for (int fm = 0; fm < FM; ++fm)
#pragma unroll
    for (int fn = 0; fn < FN; ++fn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[fm][fn][r] = 0.f;

  const int nchunk = K / BK;
#pragma unroll
  for (int d = 0; d < D; ++d) issue(min(d, nchunk - 1), d);
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((D - 1) * (NAu + NWu)) : "memory");
    __builtin_amdgcn_s_barrier();
    const __bf16* st = sm + (chunk % RING) * STAGE;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8 ah[FM], al[FM], bh[FN], bl[FN];
#pragma unroll
      for (int fm = 0; fm < FM; ++fm) {
        const int unit = aunit[fm] + ((2 * s2 + (lane >> 5)) ^ asw);
        ah[fm] = *reinterpret_cast<const bf16x8*>(st + unit * 8);
        al[fm] = *reinterpret_cast<const bf16x8*>(st + (BM * 4 + unit) * 8);
      }
#pragma unroll
      for (int fn = 0; fn < FN; ++fn) {
        bh[fn] = *reinterpret_cast<const bf16x8*>(st + AU * 8 + wbase + ((4 * s2) * BN + fn * 32) * 8);
        bl[fn] = *reinterpret_cast<const bf16x8*>(st + AU * 8 + wbase + ((4 * s2 + 1) * BN + fn * 32) * 8);
      }
#pragma unroll
      for (int fm = 0; fm < FM; ++fm)
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) acc[fm][fn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[fm], bh[fn], acc[fm][fn], 0, 0, 0);
      if (s2 == 0) issue(min(chunk + D, nchunk - 1), (chunk + D) % RING);   // refill the slot read one step ago, in the MFMA shadow
#pragma unroll
      for (int fm = 0; fm < FM; ++fm)
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) acc[fm][fn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[fm], bl[fn], acc[fm][fn], 0, 0, 0);
#pragma unroll
      for (int fm = 0; fm < FM; ++fm)
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) acc[fm][fn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[fm], bh[fn], acc[fm][fn], 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  conv_epilogue<1, BM, BN, FM, FN, 2>(p, acc, b, 0, ox0, n0, wm, wn, lane, tid, reinterpret_cast<float*>(smem_raw));
}

template <int BM, int BN>
static int launch_gp(ConvP& p, hipStream_t stream) {
  constexpr size_t lds = (size_t)3 * (2 * BM * 4 + 8 * BN) * 16;
  p.tiles_x = cdiv(p.Wout, BM); p.tiles_y = 1; p.nt = cdiv(p.Npad, BN);
  auto kern = gemm_planes_kernel<BM, BN>;
  static bool done = false;
  if (!done) { PF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); done = true; }
  hipLaunchKernelGGL(kern, dim3(p.B * p.tiles_x * p.nt), dim3(256), lds, stream, p);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}

// a.x0 = A planes (bf16 hi [M][K] then lo [M][K]); everything else as pf_conv2d with ks = 1, prologue 0, bf16x3 weights
int launch_gemm_planes(const pf_conv_args& a, hipStream_t stream) {
  ConvP p;
  memset(&p, 0, sizeof p);
  p.x0 = a.x0; p.c0 = a.c0; p.B = a.batch; p.Hin = 1; p.Win = a.win; p.Hout = 1; p.Wout = a.win;
  p.w = a.w; p.N = a.n; p.Npad = (a.n + 63) / 64 * 64;
  p.bias = a.bias; p.sbias = a.sbias; p.ld_sbias = a.ld_sbias; p.res = a.res; p.ld_res = a.ld_res;
  p.geglu = a.geglu; p.out = a.out; p.ld_out = a.ld_out; p.stats = a.stats_out; p.out_planes = a.out_planes;
  p.ksplit = 1;
  const int tile = conv_pick_tile(a);
  if (tile == 0) return launch_gp<128, 128>(p, stream);
  if (tile == 1) return launch_gp<128, 64>(p, stream);
  return launch_gp<64, 64>(p, stream);   // same row tiling as the register path: the GroupNorm statistics tiles must agree
}

}  // namespace pf
