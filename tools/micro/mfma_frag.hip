// The attention kernel's skeleton in isolation: 12-MFMA regions whose A operands come from four ds_read_b128 issued during the
// previous region (one per gap, gaps 4..7), s_waitcnt lgkmcnt(0) at the region head.  One wave per SIMD (4 waves per workgroup,
// one workgroup per CU).  MODE 0: the MFMAs consume the fragments; MODE 1: same reads, MFMAs use loop-invariant registers;
// MODE 2: no reads.  Build: hipcc --offload-arch=gfx950 -O3 -o build/mfma_frag tools/micro/mfma_frag.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int ACCV>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* cyc) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[65536];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, threadIdx.x}, b = a;
  u32x4 f[2][4];
  for (int i = 0; i < 4; ++i) { f[0][i] = a; f[1][i] = a; }
  const int lane = threadIdx.x & 63, r31 = lane & 31, g = lane >> 5;
  const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)sm + r31 * 128 + ((g ^ ((r31 >> 1) & 7)) * 16);
  for (int i = threadIdx.x; i < 4096; i += 256) reinterpret_cast<u32x4*>(sm)[i] = a;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int R = 0; R < 2; ++R) {
      asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
      for (int m = 0; m < 12; ++m) {
        const u32x4 av = MODE == 0 ? f[R][m & 3] : a;
        if (ACCV) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(av), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[m & 3]) : "v"(av), "v"(b));
        if (MODE != 2 && m >= 4 && m < 8) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[R ^ 1][m - 4]) : "v"(addr), "n"(4096));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0.f;
  for (int i = 0; i < 4; ++i) r += acc[i][0];
  if (r == 12345.678f) out[0] = r + f[0][0][0] + f[1][1][1];
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE, int ACCV>
void run(int iters) {
  float* d; unsigned long long* c; (void)hipMalloc(&d, 4); (void)hipMalloc(&c, 8);
  hipLaunchKernelGGL((k<MODE, ACCV>), dim3(256), dim3(256), 0, 0, d, 10, c);
  hipLaunchKernelGGL((k<MODE, ACCV>), dim3(256), dim3(256), 0, 0, d, iters, c);
  (void)hipDeviceSynchronize();
  unsigned long long h; (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  static const char* names[] = {"MFMAs consume the fragments", "same reads, MFMA operands loop-invariant", "no reads"};
  printf("mode %d (%s), accumulators in %s: %.1f cycles per MFMA\n", MODE, names[MODE], ACCV ? "arch VGPRs" : "the accumulator file", (double)h / (24.0 * iters));
  (void)hipFree(d); (void)hipFree(c);
}

int main() {
  run<0, 0>(5000); run<1, 0>(5000); run<2, 0>(5000);
  run<0, 1>(5000); run<1, 1>(5000); run<2, 1>(5000);
  return 0;
}
