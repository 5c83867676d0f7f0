// Practical matrix-pipe ceiling on this part: every SIMD issues back-to-back v_mfma_f32_32x32x16_bf16 on independent
// accumulators, no memory traffic.  Build: hipcc --offload-arch=gfx950 -O3 -o build/mfma_peak tools/micro/mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x & 7); b[j] = (__bf16)(float)(j); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
  if (s == 12345.678f) out[0] = s;
}

template <int NACC>
void run(int blocks_per_cu, int iters) {
  float* d; hipMalloc(&d, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * blocks_per_cu;
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, d, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = 2.0 * 32 * 32 * 16 * (double)NACC * iters * 4.0 * grid;
  printf("acc/wave %d  waves/SIMD %d  iters %d: %.3f ms  %.1f TFLOP/s bf16 dense  (%.1f fp32-equivalent at 3 MFMAs per product)\n", NACC,
         blocks_per_cu, iters, ms, flops / ms / 1e9, flops / ms / 1e9 / 3);
  hipFree(d);
}

int main() {
  run<4>(1, 20000); run<4>(2, 20000); run<8>(2, 10000); run<4>(3, 20000);
  run<4>(2, 200000);   // ~0.1 s: sustained (power-managed) clock
  return 0;
}
