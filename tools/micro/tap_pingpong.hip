// How much of the matrix pipe can one tap of the 3x3 kernel's mix keep busy, as a function of how the mix is arranged?
// One "tap" of the 8x16x128 tile per wave: 24 x v_mfma_f32_32x32x16_bf16 (2 K steps x 2x2 fragments x 3 split terms), 16 x ds_read_b128
// (A and W fragments, hi and lo planes), 4 direct-to-LDS pieces of 1 KiB (its share of the next weight tile), NV VALU instructions
// (GroupNorm + SiLU + split of the next halo, a third of them quarter-rate), barriers.
//   MODE 0: 4-wave workgroups, everything dealt out between the MFMAs (what csrc/conv_bf16x3.hip does), WPS workgroups per CU
//   MODE 1: 8-wave workgroups = two groups of 4 half a tap apart: a group's tap is a LOAD segment (all 16 fragment reads, the copies)
//           and a COMPUTE segment (24 MFMAs back to back, the VALU work as fillers), separated by workgroup barriers, so that on
//           every SIMD one wave computes while its partner loads
//   MODE 2: as 1, the VALU work in the LOAD segment          MODE 3: 4-wave workgroups with the segmented tap, no partner
// Output: cycles per tap of one wave (768 = matrix pipe alone for one wave per SIMD, 1536 for two) and chip-wide TFLOP/s.  With several
// workgroups per CU the cycle column is workgroup 0's and depends on which workgroups the dispatcher co-locates: read the TFLOP/s there.
// Build: hipcc --offload-arch=gfx950 -O3 [-DRANDDATA] [-DACCV] [-DZORDER] -o build/tap_pingpong tools/micro/tap_pingpong.hip
//   -DRANDDATA  operands with random signs / mantissas instead of 1.0 everywhere: same cycles, but the chip then clocks ~1.6 GHz instead of
//               ~2.3 GHz under a full matrix pipe (power) - profiles/r04_ab_pingpong.md
//   -DACCV      accumulators in VGPRs instead of AGPRs (no difference)      -DZORDER  the kernel's accumulator order in the middle MFMA group (none)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifdef ACCV
#define MFMA(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))
#else
#define MFMA(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(B))
#endif
#define DSR(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#define LGKM(N) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory")
#define VMC(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define BAR() asm volatile("s_barrier" ::: "memory")

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, int voff, int soff, void* lds) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}
// the halo arithmetic: NV units of fma, exp, add, rcp, mul, and, sub (7 issue slots, two of them quarter rate).  Op J is step J / NV of unit
// J % NV, so neighbouring ops belong to different units (independent), as in a compiler-scheduled conversion of several elements
template <int NV>
struct Halo { float x[4], y[NV ? NV : 1], e[NV ? NV : 1]; unsigned hi[NV ? NV : 1]; float sc, sh; };
template <int NV>
__device__ __forceinline__ void valu_op(Halo<NV>& h, int J) {
  if (NV == 0) return;
  const int u = J % (NV ? NV : 1), st = J / (NV ? NV : 1);
  switch (st) {
    case 0: asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(h.y[u]) : "v"(h.x[u & 3]), "v"(h.sc), "v"(h.sh)); break;
    case 1: asm volatile("v_exp_f32 %0, %1" : "=v"(h.e[u]) : "v"(h.y[u])); break;
    case 2: asm volatile("v_add_f32 %0, 1.0, %1" : "=v"(h.e[u]) : "v"(h.e[u])); break;
    case 3: asm volatile("v_rcp_f32 %0, %1" : "=v"(h.e[u]) : "v"(h.e[u])); break;
    case 4: asm volatile("v_mul_f32 %0, %1, %2" : "=v"(h.y[u]) : "v"(h.y[u]), "v"(h.e[u])); break;
    case 5: asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(h.hi[u]) : "v"(h.y[u])); break;
    default: asm volatile("v_sub_f32 %0, %1, %2" : "=v"(h.x[u & 3]) : "v"(h.y[u]), "v"(h.hi[u])); break;
  }
}
// ops [m * TOT / 24, (m + 1) * TOT / 24) after MFMA number m of the tap
template <int NV>
__device__ __forceinline__ void valu_after(Halo<NV>& h, int m) {
  constexpr int TOT = 7 * NV;
#pragma unroll
  for (int j = m * TOT / 24; j < (m + 1) * TOT / 24; ++j) valu_op<NV>(h, j);
}

template <int MODE, int NV7>
__global__ __launch_bounds__(MODE == 1 || MODE == 2 ? 512 : 256) void k(const char* __restrict__ src, float* out, int iters, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  constexpr int RING = 3, WT = 16384, AREG = 30720;                  // per group: A region 30 KiB, weight ring 3 x 16 KiB (79872 B, as the kernel)
  constexpr int GROUP = AREG + RING * WT;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) & 3;
  constexpr bool TWO = MODE == 1 || MODE == 2;
  const int grp = TWO ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) : 0;
  unsigned char* gsm = sm + grp * GROUP;
  const unsigned abase = (unsigned)(size_t)(gsm) + lane * 16 + (wave & 1) * 4096;             // A fragments: conflict-free 16 B per lane
  const unsigned wbase = (unsigned)(size_t)(gsm + AREG) + lane * 16 + (wave >> 1) * 4096;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, 0x7fffffff, 0x00020000);
  const int voff = (int)((blockIdx.x & 7) * 262144 + wave * 1024 + lane * 16);
  Halo<NV7> hl;
  for (int i = 0; i < 4; ++i) hl.x[i] = (float)lane * 1e-3f * (i + 1);
  hl.sc = 1.0001f; hl.sh = 0.01f;
  u32x4 ah[2][2], al[2][2], bh[2][2], bl[2][2];   // [k step][fragment]
  // fill the LDS and the ring once
  for (int i = threadIdx.x; i < (TWO ? 2 : 1) * GROUP / 16; i += blockDim.x) {
#ifdef RANDDATA
    unsigned hsh = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
    auto nx = [&]() { hsh ^= hsh << 13; hsh ^= hsh >> 17; hsh ^= hsh << 5; return (hsh & 0x807f807fu) | 0x3f003f00u; };   // bf16 pairs in [0.5, 1) with random signs / mantissas
    reinterpret_cast<u32x4*>(sm)[i] = u32x4{nx(), nx(), nx(), nx()};
#else
    reinterpret_cast<u32x4*>(sm)[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
#endif
  }
  __syncthreads();
  for (int d = 0; d < RING - 1; ++d)
    for (int pc = 0; pc < 4; ++pc) dma16(rs, voff, d * WT + pc * 4096, gsm + AREG + d * WT + pc * 4096 + wave * 1024);
  VMC(0);
  __syncthreads();
  auto load_frags = [&](int ks, unsigned wslot) {
    DSR(ah[ks][0], abase, 0); DSR(al[ks][0], abase, 8192); DSR(ah[ks][1], abase, 16384); DSR(al[ks][1], abase, 24576);
    const unsigned wb = wbase + wslot;
    DSR(bh[ks][0], wb, 0); DSR(bl[ks][0], wb, 1024); DSR(bh[ks][1], wb, 2048); DSR(bl[ks][1], wb, 3072);
  };
  int slot = 0;
  unsigned long long t0 = 0;
  if (MODE == 0) {
    load_frags(0, 0);
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
      const int nslot = slot == RING - 1 ? 0 : slot + 1, fslot = nslot == RING - 1 ? 0 : nslot + 1;
      const int toff = (it & 7) * WT;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        LGKM(0);
        // next K step's fragments into the other register set while this one computes; group order X (lo x hi), Z (hi x hi), Y (hi x lo)
#pragma unroll
        for (int g = 0; g < 3; ++g) {
#pragma unroll
          for (int f = 0; f < 4; ++f) {
            const int fm = f >> 1, fn = f & 1;
            if (g == 0) MFMA(acc[f], al[ks][fm], bh[ks][fn]);
            else if (g == 1) MFMA(acc[f], ah[ks][fm], bh[ks][fn]);
            else MFMA(acc[f], ah[ks][fm], bl[ks][fn]);
            const int idx = g * 4 + f;
            if (ks == 0 && idx == 1) {   // K step 1 of this tap
              DSR(ah[1][0], abase, 512); DSR(al[1][0], abase, 8192 + 512);
            } else if (ks == 0 && idx == 2) {
              DSR(ah[1][1], abase, 16384 + 512); DSR(al[1][1], abase, 24576 + 512);
            } else if (ks == 0 && idx == 3) {
              DSR(bh[1][0], wbase + slot * WT, 512); DSR(bl[1][0], wbase + slot * WT, 1024 + 512);
            } else if (ks == 0 && idx == 4) {
              DSR(bh[1][1], wbase + slot * WT, 2048 + 512); DSR(bl[1][1], wbase + slot * WT, 3072 + 512);
            } else if (ks == 0 && idx >= 6 && idx < 10) {
              dma16(rs, voff, toff + (idx - 6) * 4096, gsm + AREG + fslot * WT + (idx - 6) * 4096 + wave * 1024);
            }
            valu_after<NV7>(hl, ks * 12 + idx);
            if (ks == 1 && idx == 6) { VMC(4); BAR(); }
            if (ks == 1 && idx == 7) { DSR(ah[0][0], abase, 0); DSR(al[0][0], abase, 8192); }
            if (ks == 1 && idx == 8) { DSR(ah[0][1], abase, 16384); DSR(al[0][1], abase, 24576); }
            if (ks == 1 && idx == 9) { DSR(bh[0][0], wbase + nslot * WT, 0); DSR(bl[0][0], wbase + nslot * WT, 1024); }
            if (ks == 1 && idx == 10) { DSR(bh[0][1], wbase + nslot * WT, 2048); DSR(bl[0][1], wbase + nslot * WT, 3072); }
          }
        }
      }
      slot = nslot;
    }
  } else {
    if (TWO && grp) BAR();   // group 1 runs one segment behind
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
      const int nslot = slot == RING - 1 ? 0 : slot + 1, fslot = nslot == RING - 1 ? 0 : nslot + 1;
      const int toff = (it & 7) * WT;
      // LOAD segment: every fragment of the tap, then the refill of the slot freed one tap ago
      load_frags(0, slot * WT);
      DSR(ah[1][0], abase, 512); DSR(al[1][0], abase, 8192 + 512); DSR(ah[1][1], abase, 16384 + 512); DSR(al[1][1], abase, 24576 + 512);
      DSR(bh[1][0], wbase + slot * WT, 512); DSR(bl[1][0], wbase + slot * WT, 1024 + 512);
      DSR(bh[1][1], wbase + slot * WT, 2048 + 512); DSR(bl[1][1], wbase + slot * WT, 3072 + 512);
#pragma unroll
      for (int pc = 0; pc < 4; ++pc) dma16(rs, voff, toff + pc * 4096, gsm + AREG + fslot * WT + pc * 4096 + wave * 1024);
      if (MODE >= 2)
#pragma unroll
        for (int m = 0; m < 24; ++m) valu_after<NV7>(hl, m);
      LGKM(0);
      if (TWO) BAR();
      // COMPUTE segment
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
          for (int f = 0; f < 4; ++f) {
#ifdef ZORDER
            const int ff = g == 1 ? ((f & 1) << 1 | (f >> 1)) : f;   // middle group walks the accumulators column-major, as the kernel's GZ
#else
            const int ff = f;
#endif
            const int fm = ff >> 1, fn = ff & 1;
            if (g == 0) MFMA(acc[ff], al[ks][fm], bh[ks][fn]);
            else if (g == 1) MFMA(acc[ff], ah[ks][fm], bh[ks][fn]);
            else MFMA(acc[ff], ah[ks][fm], bl[ks][fn]);
            if (MODE == 1) valu_after<NV7>(hl, ks * 12 + g * 4 + f);
          }
      VMC(4);   // the next tap's weight tile has landed (this tap's refill may still be in flight)
      BAR();
      slot = nslot;
    }
    if (TWO && !grp) BAR();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  VMC(0);
  float r = hl.x[0] + hl.x[1] + hl.x[2] + hl.x[3];
  for (int i = 0; i < 4; ++i) r += acc[i][0];
  if (r == 12345.678f) out[0] = r;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE, int NV7>
void run(const char* src, int wps, int iters) {
  float* d; unsigned long long* c; hipMalloc(&d, 4); hipMalloc(&c, 8);
  const bool two = MODE == 1 || MODE == 2;
  const size_t lds = (two ? 2 : 1) * (30720 + 3 * 16384);
  auto kern = k<MODE, NV7>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int threads = two ? 512 : 256;
  hipLaunchKernelGGL(kern, dim3(256 * wps), dim3(threads), lds, 0, src, d, 10, c);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(kern, dim3(256 * wps), dim3(threads), lds, 0, src, d, iters, c);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  const double flop = (double)256 * wps * (threads / 64) * 24.0 * 32768.0 * iters;
  const int wsimd = wps * (threads / 256);
  printf("%s  %d wave(s)/SIMD  VALU units/tap %2d: %7.1f cycles per tap of one wave (pipe alone %4d)  %6.1f TFLOP/s (products: /3)  %.3f ms\n",
         MODE == 3 ? "4-wave workgroups, segments       " : MODE == 2 ? "segments, VALU in the load segment" : MODE ? "segments, VALU between the MFMAs  " : "interleaved                       ", wsimd, NV7, (double)h / iters, 768 * wsimd, flop / ms * 1e-9, ms);
  hipFree(d); hipFree(c);
}

int main() {
  char* src; hipMalloc(&src, (size_t)8 * 262144 + (1 << 20)); {
#ifdef RANDDATA
    size_t nb = (size_t)8 * 262144 + (1 << 20); unsigned* h = (unsigned*)malloc(nb); unsigned x = 12345u;
    for (size_t i = 0; i < nb / 4; ++i) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; h[i] = (x & 0x807f807fu) | 0x3f003f00u; }
    hipMemcpy(src, h, nb, hipMemcpyHostToDevice); free(h);
#else
    hipMemset(src, 1, (size_t)8 * 262144 + (1 << 20));
#endif
  }
  const int iters = 4000;
  run<0, 0>(src, 1, iters); run<0, 3>(src, 1, iters); run<0, 8>(src, 1, iters);
  run<0, 0>(src, 2, iters); run<0, 3>(src, 2, iters); run<0, 8>(src, 2, iters);
  run<1, 0>(src, 1, iters); run<1, 3>(src, 1, iters); run<1, 8>(src, 1, iters);
  run<2, 3>(src, 1, iters); run<2, 8>(src, 1, iters);
  run<3, 3>(src, 1, iters); run<3, 3>(src, 2, iters); run<3, 8>(src, 2, iters);
#ifdef BIGVALU   // how much vector-ALU work fits beside the partner's MFMA stream (the attention kernel's softmax is ~40 units per 48 MFMAs)
  run<2, 16>(src, 1, iters); run<2, 24>(src, 1, iters); run<2, 40>(src, 1, iters);
  run<1, 16>(src, 1, iters); run<1, 24>(src, 1, iters); run<1, 40>(src, 1, iters);
#endif
  return 0;
}
