// What does a direct-to-LDS load cost the wave that issues it, next to MFMAs - and is the buffer form cheaper than the global one?
// 4 waves per workgroup, WPS workgroups per CU.  Per 12 MFMAs (one "slot" of the kernels in csrc/): ND loads of 1 KiB (dwordx4 per lane)
// from an L2-resident window into an LDS ring nobody reads, vmcnt keeps 2 slots in flight.  MODE 0: no loads, 1: global_load_lds
// (64-bit address VGPR pair per lane), 2: buffer_load ... lds (SGPR resource + 32-bit offset VGPR + SGPR slot offset).
// Build: hipcc --offload-arch=gfx950 -O3 -o build/glds_issue tools/micro/glds_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int ND>
__global__ __launch_bounds__(256) void k(const char* __restrict__ src, float* out, int iters, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, threadIdx.x}, b = a;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const char* gp = src + (size_t)(blockIdx.x & 7) * 262144 + wave * 1024 + lane * 16;   // 8 windows of 256 KiB: L2 resident
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, 0x7fffffff, 0x00020000);
  const int voff = (int)((blockIdx.x & 7) * 262144 + wave * 1024 + lane * 16);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int slot = 0;
  for (int it = 0; it < iters; ++it) {
    const int toff = (it & 15) * (4096 * ND);
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i & 3]) : "v"(a), "v"(b));
      if (MODE && i >= 2 && i < 2 + ND) {
        auto l = (__attribute__((address_space(3))) void*)(sm + slot * (4096 * ND) + (i - 2) * 4096 + wave * 1024);
        if (MODE == 1) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + toff + (i - 2) * 4096), l, 16, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, l, 16, voff, toff + (i - 2) * 4096, 0, 0);
      }
    }
    if (MODE) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * ND) : "memory");
    slot = slot == 3 ? 0 : slot + 1;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float r = 0.f;
  for (int i = 0; i < 4; ++i) r += acc[i][0];
  if (r == 12345.678f) out[0] = r;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE, int ND>
void run(const char* src, int wps, int iters) {
  float* d; unsigned long long* c; hipMalloc(&d, 4); hipMalloc(&c, 8);
  const size_t lds = 4 * 4096 * (ND ? ND : 1);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, ND>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((k<MODE, ND>), dim3(256 * wps), dim3(256), lds, 0, src, d, 10, c);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((k<MODE, ND>), dim3(256 * wps), dim3(256), lds, 0, src, d, iters, c);
  hipDeviceSynchronize();
  unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  const char* nm[3] = {"no loads        ", "global_load_lds ", "buffer_load lds "};
  printf("waves/SIMD %d  %s x%d per 12 MFMAs: %.1f cycles per slot of one wave (384 = matrix pipe alone)  -> %.0f cycles per load\n", wps, nm[MODE], ND,
         (double)h / iters, ND ? ((double)h / iters - 384.0 * (wps > 1 ? 0 : 1)) / ND : 0.0);
  hipFree(d); hipFree(c);
}

int main() {
  char* src; hipMalloc(&src, (size_t)8 * 262144 + (1 << 20)); hipMemset(src, 1, (size_t)8 * 262144 + (1 << 20));
  for (int wps = 1; wps <= 2; ++wps) {
    run<0, 0>(src, wps, 20000);
    run<1, 1>(src, wps, 20000); run<2, 1>(src, wps, 20000);
    run<1, 2>(src, wps, 20000); run<2, 2>(src, wps, 20000);
    run<1, 4>(src, wps, 20000); run<2, 4>(src, wps, 20000);
  }
  return 0;
}
