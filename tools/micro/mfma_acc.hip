// What does moving scores out of the accumulator file cost beside MFMAs?  One wave per SIMD, four accumulators in rotation, every
// MFMA followed by NV fillers: (mode 0) v_mov_b32 between arch VGPRs, (mode 1) v_accvgpr_read_b32 from accumulator registers NO
// MFMA of the loop touches, (mode 2) v_accvgpr_read_b32 from the accumulator the MFMA issued three slots earlier wrote,
// (mode 3) accumulators in arch VGPRs (vgpr-cd form) and v_mov_b32 from the accumulator written three slots earlier,
// (mode 4) v_accvgpr_read_b32 from accumulators that MFMAs wrote BEFORE the timed loop and nothing touches inside it.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/mfma_acc tools/micro/mfma_acc.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* cyc) {
  f32x16 acc[4], other;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int r = 0; r < 16; ++r) other[r] = (float)threadIdx.x;
  if (MODE == 1) asm volatile("" : "+a"(other));
  f32x16 idle[2];
  if (MODE == 4) {
    for (int r = 0; r < 16; ++r) { idle[0][r] = 0.f; idle[1][r] = 0.f; }
    u32x4 aa = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, threadIdx.x};
    for (int q = 0; q < 4; ++q) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(idle[0]) : "v"(aa), "v"(aa));
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(idle[1]) : "v"(aa), "v"(aa));
    }
    asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15");
  }
  u32x4 a0 = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, threadIdx.x}, b = a0;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (MODE == 3) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a0), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a0), "v"(b));
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if (MODE == 0) asm volatile("v_mov_b32 %0, %1" : "=v"(s[v & 3]) : "v"(b[v & 3]));
        else if (MODE == 1) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(s[v & 3]) : "a"(other[(4 * i + v) & 15]));
        else if (MODE == 4) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(s[v & 3]) : "a"(idle[i & 1][(4 * (i >> 1) + v) & 15]));
        else if (MODE == 2) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(s[v & 3]) : "a"(acc[(i + 1) & 3][v & 15]));
        else asm volatile("v_mov_b32 %0, %1" : "=v"(s[v & 3]) : "v"(acc[(i + 1) & 3][v & 15]));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0.f;
  for (int i = 0; i < 4; ++i) r += acc[i][0];
  if (MODE == 4) r += idle[0][0] + idle[1][0];
  if (r == 12345.678f) out[0] = r + s[0] + s[1] + s[2] + s[3] + other[0];
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE, int NV>
void run(int iters) {
  float* d; unsigned long long* c; hipMalloc(&d, 4); hipMalloc(&c, 8);
  hipLaunchKernelGGL((k<MODE, NV>), dim3(256), dim3(256), 0, 0, d, 10, c);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((k<MODE, NV>), dim3(256), dim3(256), 0, 0, d, iters, c);
  hipDeviceSynchronize();
  unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  static const char* names[] = {"v_mov between arch VGPRs", "v_accvgpr_read of registers no MFMA touches", "v_accvgpr_read of the accumulator written 3 MFMAs ago",
                                "arch-VGPR accumulators, v_mov of the accumulator written 3 MFMAs ago",
                                "v_accvgpr_read of accumulators MFMAs wrote before the loop"};
  printf("mode %d (%s)  %d fillers per MFMA: %.1f cycles per MFMA\n", MODE, names[MODE], NV, (double)h / (4.0 * iters));
  hipFree(d); hipFree(c);
}

int main() {
  run<0, 0>(20000);
  run<0, 2>(20000); run<1, 2>(20000); run<2, 2>(20000); run<3, 2>(20000);
  run<4, 2>(20000);
  run<0, 4>(20000); run<1, 4>(20000); run<2, 4>(20000); run<3, 4>(20000); run<4, 4>(20000);
  return 0;
}
