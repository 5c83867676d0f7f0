// What does a VALU write into a register that the MFMA just issued reads as its A / B operand cost?  One wave per SIMD, four
// accumulators, every MFMA followed by NV single-issue VALU instructions whose destination is (mode 0) a scratch register,
// (mode 1) a register of the A operand of the MFMA just issued, (mode 2) of the MFMA before it.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/mfma_war tools/micro/mfma_war.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* cyc) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  u32x4 a0 = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, threadIdx.x}, a1 = a0, b = a0, s = a0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // MFMA i reads a0 (even i) or a1 (odd i)
      if (i & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a1), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a0), "v"(b));
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if (MODE == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(s[v & 3]) : "v"(b[0]));
        else if ((MODE == 1) == ((i & 1) != 0)) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a1[v & 3]) : "v"(b[0]));   // mode 1: the operand just issued
        else asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0[v & 3]) : "v"(b[0]));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0.f;
  for (int i = 0; i < 4; ++i) r += acc[i][0];
  if (r == 12345.678f) out[0] = r + s[0] + a0[0] + a1[0];
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
  printf("mode %d (%s)  %d VALU per MFMA: %.1f cycles per MFMA\n", MODE,
         MODE == 0 ? "scratch destination" : MODE == 1 ? "writes the A operand of the MFMA just issued" : "writes the A operand of the MFMA before",
         NV, (double)h / (4.0 * iters));
  hipFree(d); hipFree(c);
}

int main() {
  run<0, 0>(20000);
  run<0, 2>(20000); run<1, 2>(20000); run<2, 2>(20000);
  run<0, 4>(20000); run<1, 4>(20000); run<2, 4>(20000);
  run<0, 6>(20000); run<1, 6>(20000);
  return 0;
}
