// glds_rate.hip - how fast can ONE CU stream an L2-resident buffer into LDS with global_load_lds_dwordx4?
// Every workgroup (256 threads) sweeps the same `span` bytes (L2/MALL resident after the first pass) `iters` times, keeping
// DEPTH 16-KiB tiles in flight.  Prints GB/s per CU and aggregate for 1..4 workgroups per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o build/glds_rate tools/micro/glds_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int DEPTH>
__global__ __launch_bounds__(256) void stream(const char* __restrict__ src, size_t span, int iters, int distinct) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // distinct != 0: every workgroup reads its own window (HBM / MALL stream); 0: all read the same bytes (L2 hot)
  const char* base = src + (distinct ? (size_t)blockIdx.x * span : 0);
  const size_t tiles = span / 16384;
  size_t t = 0;
  for (int it = 0; it < iters; ++it) {
    for (size_t k = 0; k < tiles; ++k, ++t) {
      const int slot = t % DEPTH;
#pragma unroll
      for (int j = 0; j < 4; ++j)   // 4 waves x 4 instrs x 1 KiB = 16 KiB
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + k * 16384 + (size_t)(j * 4 + wave) * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(lds + slot * 16384 + (j * 4 + wave) * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * 4) : "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int DEPTH>
void run(const char* d, size_t span, int wg_per_cu, int distinct) {
  const int cus = 256, grid = cus * wg_per_cu, iters = distinct ? 8 : 64;
  const size_t lds = DEPTH * 16384;
  hipFuncSetAttribute(reinterpret_cast<const void*>(stream<DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(stream<DEPTH>, dim3(grid), dim3(256), lds, 0, d, span, 2, distinct);
  hipEventRecord(e0);
  hipLaunchKernelGGL(stream<DEPTH>, dim3(grid), dim3(256), lds, 0, d, span, iters, distinct);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)grid * span * iters;
  printf("depth %d  wg/cu %d  %s  span %6zu KiB: %7.1f GB/s per CU, %6.2f TB/s aggregate\n", DEPTH, wg_per_cu, distinct ? "distinct" : "shared  ",
         span >> 10, bytes / ms / 1e6 / cus, bytes / ms / 1e9);
}

int main() {
  const size_t total = (size_t)1 << 30;
  char* d;
  hipMalloc(&d, total);
  hipMemset(d, 1, total);
  for (int wg = 1; wg <= 4; wg *= 2) {
    run<2>(d, 256 << 10, wg, 0);
    run<4>(d, 256 << 10, wg, 0);
    if (wg <= 2) run<8>(d, 256 << 10, wg, 0);
    run<4>(d, 2 << 20, wg, 0);
    run<4>(d, 1 << 20, wg, 1);
  }
  return 0;
}
