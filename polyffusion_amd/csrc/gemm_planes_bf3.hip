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

template <int BM, int BN, int RING>
__global__ __launch_bounds__(256, 2) void gemm_planes_kernel(ConvP p) {
  constexpr int BK = 32;
  constexpr int WM = BM / 2, WN = BN / 2, FM = WM / 32, FN = WN / 32;
  constexpr int AU = 2 * BM * 4;            // 16-byte units of one A stage (2 planes x BM rows x 4 slots)
  constexpr int WU = 8 * BN;                // 16-byte units of one W stage (4 k8 x 2 planes x BN)
  constexpr int NAu = AU / 256, NWu = WU / 256;
  constexpr int STAGE = (AU + WU) * 8;      // bf16 elements per stage
  static_assert(AU % 256 == 0 && WU % 256 == 0, "tile must divide over the block");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  x3_t* sm = reinterpret_cast<x3_t*>(smem_raw);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: LDS-DMA destinations (M0) stay in SGPRs
  const int wm = wave >> 1, wn = wave & 1;
  const int nwg = gridDim.x;
  int lid;
  {
    const int orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  const int mt = fdiv(lid, p.d_nt);
  const int nti = lid - mt * p.nt;
  const int b = fdiv(mt, p.d_tx);
  const int tx = mt - b * p.tiles_x;
  const int n0 = nti * BN, ox0 = tx * BM;
  const int K = p.c0, L = p.Wout;
  const size_t MK = (size_t)p.B * L * K;
  const x3_t* A = reinterpret_cast<const x3_t*>(p.x0);

  // per-thread byte offsets of the stage pieces inside the workgroup's A rows / W column tile (a chunk adds a wave-uniform offset):
  // buffer-form direct-to-LDS loads (dma16, conv_common.h).  The A resource starts at this sample tile's first row of the hi plane,
  // so every offset stays far below 2 GiB whatever the tensor size.
  const x3_t* Abase = A + ((size_t)b * L + min(ox0, L - 1)) * K;
  const __amdgpu_buffer_rsrc_t rsA = dma_resource(Abase), rsW = dma_resource(static_cast<const x3_t*>(p.w) + (size_t)n0 * 8);
  int va[NAu];
#pragma unroll
  for (int j = 0; j < NAu; ++j) {
    const int u = tid + j * 256;
    const int plane = u / (BM * 4), w = u % (BM * 4), row = w >> 2, slot = w & 3;
    const int srow = min(ox0 + row, L - 1) - min(ox0, L - 1);     // rows past the sample are never stored
    va[j] = (int)(((size_t)plane * MK + (size_t)srow * K + ((slot ^ ((row >> 2) & 3)) * 8)) * 2);
  }
  int vw[NWu];
#pragma unroll
  for (int j = 0; j < NWu; ++j) {
    const int u = tid + j * 256;
    const int k8l = u / (2 * BN), plane = (u / BN) & 1, n = u % BN;
    vw[j] = (int)((((size_t)(k8l * 2 + plane) * p.Npad + n) * 8) * 2);
  }
  const int wrow_b = 2 * p.Npad * 8 * 2;      // bytes per k8 row pair (hi|lo planes)
  auto issue = [&](int chunk, int stage) {
    x3_t* sb = sm + stage * STAGE;
    const int soa = chunk * BK * 2, sow = chunk * 4 * wrow_b;
#pragma unroll
    for (int j = 0; j < NAu; ++j) dma16(rsA, va[j], soa, sb + (wave * 64 + j * 256) * 8);
#pragma unroll
    for (int j = 0; j < NWu; ++j) dma16(rsW, vw[j], sow, sb + AU * 8 + (wave * 64 + j * 256) * 8);
  };

  // LDS byte addresses of this lane's fragments inside a stage.  The swizzled 16-byte slot of K step s is
  // (2s + g) ^ sw = ((g ^ sw) & 1 | sw & 2) ^ 2s, so step 1 is step 0 with bit 1 flipped: two base registers per row set.
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) x3_t*)sm;
  const int sw = ((lane & 31) >> 2) & 3;
  const int x0 = (sw & 2) | (((lane >> 5) ^ sw) & 1);
  unsigned abase[FM][2];
#pragma unroll
  for (int fm = 0; fm < FM; ++fm) {
    const int row = wm * WM + fm * 32 + (lane & 31);
    abase[fm][0] = lds0 + (row * 4 + x0) * 16;
    abase[fm][1] = lds0 + (row * 4 + (x0 ^ 2)) * 16;
  }
  const unsigned wb = lds0 + AU * 16 + (((lane >> 5) * 2 * BN + wn * WN + (lane & 31)) * 16);
  constexpr int ALO = BM * 64;              // byte offset of the lo plane inside the A region
  constexpr int STAGE_B = STAGE * 2;        // bytes per stage

  f32x16 acc[FM][FN];
#pragma unroll
  for (int fm = 0; fm < FM; ++fm)
#pragma unroll
    for (int fn = 0; fn < FN; ++fn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[fm][fn][r] = 0.f;

  x3x8 ah[FM], al[FM], bh[FN], bl[FN];
  // S = K step (0/1), OFF = byte offset of the ring stage (runtime: added to the base register)
#define LD_AL(S, OFF) static_for<0, FM>([&](auto i) { al[i.value] = lds_read128<ALO>(abase[i.value][S] + (OFF)); })
#define LD_AH(S, OFF) static_for<0, FM>([&](auto i) { ah[i.value] = lds_read128<0>(abase[i.value][S] + (OFF)); })
#define LD_BH(S, OFF) static_for<0, FN>([&](auto i) { bh[i.value] = lds_read128<((4 * (S)) * BN + i.value * 32) * 16>(wb + (OFF)); })
#define LD_BL(S, OFF) static_for<0, FN>([&](auto i) { bl[i.value] = lds_read128<((4 * (S) + 1) * BN + i.value * 32) * 16>(wb + (OFF)); })
#define SB() __builtin_amdgcn_sched_barrier(0)
  auto X = [&]() {
#pragma unroll
    for (int fm = 0; fm < FM; ++fm)
#pragma unroll
      for (int fn = 0; fn < FN; ++fn) acc[fm][fn] = x3_mfma_32x32x16(al[fm], bh[fn], acc[fm][fn], 0, 0, 0);
  };
  auto Z = [&]() {
#pragma unroll
    for (int fm = 0; fm < FM; ++fm)
#pragma unroll
      for (int fn = 0; fn < FN; ++fn) acc[fm][fn] = x3_mfma_32x32x16(ah[fm], bh[fn], acc[fm][fn], 0, 0, 0);
  };
  auto Y = [&]() {
#pragma unroll
    for (int fm = 0; fm < FM; ++fm)
#pragma unroll
      for (int fn = 0; fn < FN; ++fn) acc[fm][fn] = x3_mfma_32x32x16(ah[fm], bl[fn], acc[fm][fn], 0, 0, 0);
  };

  // Same skewed single-fragment-set pipeline as the 3x3 loop of conv_bf16x3.hip: X = a_lo.w_hi, Z = a_hi.w_hi, Y = a_hi.w_lo;
  // each fragment is re-loaded for the next K step right after its last reader, waits are counted by hand (FM+FN reads may
  // stay outstanding before every group), one barrier per 32-deep chunk, and the stage read last is refilled three ahead.
  const int nchunk = K / BK;
#pragma unroll
  for (int d = 0; d < RING; ++d) issue(min(d, nchunk - 1), d);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 1) * (NAu + NWu)) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  LD_AL(0, 0u); LD_BH(0, 0u); LD_AH(0, 0u); LD_BL(0, 0u);
  int slot = 0;
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    const unsigned so = slot * STAGE_B;
    const int slotn = slot == RING - 1 ? 0 : slot + 1;
    const unsigned son = slotn * STAGE_B;
    SB();
    lgkm_wait<FM + FN>(); SB();
    X(); SB(); LD_AL(1, so); SB();
    lgkm_wait<FM + FN>(); SB();
    Z(); SB(); LD_BH(1, so); SB();
    lgkm_wait<FM + FN>(); SB();
    Y(); SB(); LD_AH(1, so); LD_BL(1, so); SB();
    lgkm_wait<FM + FN>(); SB();
    X(); SB();
    // stage chunk+1 complete (this thread's newer stage may stay in flight), every read of stage `slot` done
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((RING - 2) * (NAu + NWu)) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(min(chunk + RING, nchunk - 1), slot);
    LD_AL(0, son); SB();
    Z(); SB(); LD_BH(0, son); SB();
    Y(); SB(); LD_AH(0, son); LD_BL(0, son); SB();
    slot = slotn;
  }
#undef LD_AL
#undef LD_AH
#undef LD_BH
#undef LD_BL
#undef SB
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  // the last iteration's fragment reads (for a chunk that never runs) are dead values to the compiler: keep their registers
  // allocated until the wait above has passed, or a late LDS return lands in whatever was placed there (see conv_bf16x3.hip)
  #pragma unroll
  for (int i = 0; i < FM; ++i) asm volatile("" ::"v"(al[i]), "v"(ah[i]));
  #pragma unroll
  for (int i = 0; i < FN; ++i) asm volatile("" ::"v"(bh[i]), "v"(bl[i]));
  __syncthreads();
  x3_unscale(acc);
  conv_epilogue<1, BM, BN, FM, FN, 2>(p, acc, b, 0, ox0, n0, wm, wn, lane, tid, reinterpret_cast<float*>(smem_raw));
}

template <int BM, int BN, int RING>
static int launch_gp(ConvP& p, hipStream_t stream) {
  constexpr size_t ring = (size_t)RING * (2 * BM * 4 + 8 * BN) * 16, epi = (size_t)BM * (BN + 8) * 4;   // epi: planes-output transpose
  constexpr size_t lds = ring > epi ? ring : epi;
  p.tiles_x = cdiv(p.Wout, BM); p.tiles_y = 1; p.nt = cdiv(p.Npad, BN);
  conv_fill_divs(p);
  auto kern = gemm_planes_kernel<BM, BN, RING>;
  static std::atomic<uint64_t> attr_done{0};
  if (int rc = set_max_lds_once(reinterpret_cast<const void*>(kern), (int)lds, attr_done)) return rc;
  hipLaunchKernelGGL(kern, dim3(p.B * p.tiles_x * p.nt), dim3(256), lds, stream, p);
  PF_CHECK_HIP(hipGetLastError());
  return PF_OK;
}

// a.x0 = A planes (bf16 hi [M][K] then lo [M][K]); everything else as pf_conv2d with ks = 1, prologue 0, bf16x3 weights
int launch_gemm_planes(const pf_conv_args& a, hipStream_t stream) {
  PF_REQUIRE((size_t)a.batch * a.win * a.c0 * 2 * 2 < ((size_t)1 << 31), "gemm_planes: the A plane pair must stay below 2 GiB (32-bit offsets of the direct-to-LDS loads)");
  ConvP p;
  memset(&p, 0, sizeof p);
  p.x0 = a.x0; p.c0 = a.c0; p.B = a.batch; p.Hin = 1; p.Win = a.win; p.Hout = 1; p.Wout = a.win;
  p.w = a.w; p.N = a.n; p.Npad = (a.n + 63) / 64 * 64;
  p.bias = a.bias; p.sbias = a.sbias; p.ld_sbias = a.ld_sbias; p.res = a.res; p.ld_res = a.ld_res;
  p.sb_rows = reinterpret_cast<const long long*>(a.sbias_rows); p.sb_nrows = a.sbias_nrows;
  p.geglu = a.geglu; p.out = a.out; p.ld_out = a.ld_out; p.stats = a.stats_out; p.out_planes = a.out_planes; p.qkv = a.qkv_planes;
  p.ksplit = 1;
  p.amax = static_cast<unsigned*>(a.absmax_slot);
  const int tile = conv_pick_tile(a);
  // ring depth: the deepest that still lets two workgroups share a CU's 160 KB (a 128x128 stage is 32 KB)
  if (tile == 0) return launch_gp<128, 128, 2>(p, stream);
  if (tile == 1) return launch_gp<128, 64, 3>(p, stream);
  return launch_gp<64, 64, 3>(p, stream);   // same row tiling as the register path: the GroupNorm statistics tiles must agree
}

}  // namespace pf
