// Issue cost of LDS fragment reads and barriers beside MFMAs.  4 waves per workgroup, WPS waves per SIMD (WPS workgroups per CU).
// Per MFMA: NR ds_read_b128 (conflict-free, consumed by nobody; waited for with lgkmcnt(8) so they never stall on latency),
// NV VALU fillers; every PER MFMAs one s_barrier (0 = none).
// Build: hipcc --offload-arch=gfx950 -O3 -o build/mfma_lds tools/micro/mfma_lds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NR, int NV, int PER>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* cyc) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[65536];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, threadIdx.x}, b = a, s = a;
  u32x4 f[4];
  for (int i = 0; i < 4; ++i) f[i] = a;
  const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)sm + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 8192;
  reinterpret_cast<u32x4*>(sm)[threadIdx.x] = a;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i & 3]) : "v"(a), "v"(b));
#pragma unroll
      for (int r = 0; r < NR; ++r) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[(i * NR + r) & 3]) : "v"(addr), "n"(((0 * 2 + 0) & 3) * 1024));
#pragma unroll
      for (int v = 0; v < NV; ++v) asm volatile("v_add_u32 %0, %0, %1" : "+v"(s[v & 3]) : "v"(b[0]));
      if (NR) asm volatile("s_waitcnt lgkmcnt(8)");
      if (PER && (i % PER) == PER - 1) asm volatile("s_barrier");
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)");
  float r = 0.f;
  for (int i = 0; i < 4; ++i) r += acc[i][0] + __builtin_bit_cast(float, f[i][0]);
  if (r == 12345.678f) out[0] = r + s[0];
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NR, int NV, int PER>
void run(int wps, int iters) {
  float* d; unsigned long long* c; hipMalloc(&d, 4); hipMalloc(&c, 8);
  hipLaunchKernelGGL((k<NR, NV, PER>), dim3(256 * wps), dim3(256), 0, 0, d, 10, c);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((k<NR, NV, PER>), dim3(256 * wps), dim3(256), 0, 0, d, iters, c);
  hipDeviceSynchronize();
  unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("waves/SIMD %d  ds_read_b128/MFMA %d  VALU/MFMA %d  barrier every %d MFMAs: %.1f cycles per MFMA of one wave  (pipe share %.0f %%)\n", wps, NR, NV, PER,
         (double)h / (8.0 * iters), 100.0 * 32.0 * wps / ((double)h / (8.0 * iters)));
  hipFree(d); hipFree(c);
}

int main() {
  run<0, 0, 0>(1, 20000);
  run<1, 0, 0>(1, 20000); run<2, 0, 0>(1, 20000);
  run<1, 3, 0>(1, 20000); run<1, 4, 0>(1, 20000);
  run<0, 0, 8>(1, 20000); run<0, 0, 4>(1, 20000);
  run<1, 3, 8>(1, 20000);
  run<0, 0, 0>(2, 20000); run<1, 3, 0>(2, 20000); run<1, 3, 8>(2, 20000); run<1, 4, 8>(2, 20000); run<1, 6, 8>(2, 20000); run<2, 6, 8>(2, 20000);
  return 0;
}
