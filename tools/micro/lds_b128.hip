// Which lane -> address patterns does ds_read_b128 serve at full rate?  One workgroup per CU; every lane reads 16 bytes at a
// per-lane offset given by the host, 8 independent reads per loop trip, timed with s_memtime.  Patterns are printed with the cycles
// per wave-instruction (1 wave) and per wave-instruction with 4 waves sharing the LDS pipe.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/lds_b128 tools/micro/lds_b128.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <functional>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void k(const unsigned* offs, int iters, unsigned long long* cyc, unsigned* sink) {
  extern __shared__ unsigned char sm[];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<unsigned*>(sm)[i] = i;
  __syncthreads();
  const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)sm + offs[threadIdx.x & 63];
  u32x4 v[8];
  unsigned acc = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(v[j]) : "v"(a));
    asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += v[j][0];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (acc == 0x12345) sink[0] = acc;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

static double run(const std::vector<unsigned>& offs, int threads) {
  unsigned* d; unsigned long long* c; unsigned* s;
  (void)hipMalloc(&d, 256); (void)hipMalloc(&c, 8); (void)hipMalloc(&s, 4);
  (void)hipMemcpy(d, offs.data(), 256, hipMemcpyHostToDevice);
  const int iters = 2000;
  hipLaunchKernelGGL(k, dim3(256), dim3(threads), 65536, 0, d, 10, c, s);
  hipLaunchKernelGGL(k, dim3(256), dim3(threads), 65536, 0, d, iters, c, s);
  (void)hipDeviceSynchronize();
  unsigned long long h; (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  (void)hipFree(d); (void)hipFree(c); (void)hipFree(s);
  return (double)h / (8.0 * iters);
}

int main() {
  struct P { const char* name; std::function<unsigned(int)> f; };
  std::vector<P> ps = {
    {"linear: lane*16", [](int l) { return (unsigned)l * 16; }},
    {"all lanes same address", [](int) { return 0u; }},
    {"row = lane&31 (128 B pitch), slot = g                 (no swizzle)", [](int l) { int r = l & 31, g = l >> 5; return (unsigned)(r * 128 + g * 16); }},
    {"row = lane&31, slot = g ^ (r&7)                        (attention today)", [](int l) { int r = l & 31, g = l >> 5; return (unsigned)(r * 128 + ((g ^ (r & 7)) * 16)); }},
    {"row = lane&31, slot = g ^ 2(r&3) ^ ((r>>2)&1)", [](int l) { int r = l & 31, g = l >> 5; return (unsigned)(r * 128 + ((g ^ (2 * (r & 3)) ^ ((r >> 2) & 1)) * 16)); }},
    {"row = lane&31, slot = (g + 2r) & 7", [](int l) { int r = l & 31, g = l >> 5; return (unsigned)(r * 128 + (((g + 2 * r) & 7) * 16)); }},
    {"row = lane&31, slot = (2g + r) & 7 ... g picks 32-byte half", [](int l) { int r = l & 31, g = l >> 5; return (unsigned)(r * 128 + (((4 * g) ^ (r & 7)) * 16)); }},
    {"row = lane&31, slot = g ^ ((r>>1)&7)                   (64 banks: 16 lanes cover 256 B)", [](int l) { int r = l & 31, g = l >> 5; return (unsigned)(r * 128 + ((g ^ ((r >> 1) & 7)) * 16)); }},
    {"row = lane&31, slot = (2+g) ^ ((r>>1)&7)", [](int l) { int r = l & 31, g = l >> 5; return (unsigned)(r * 128 + (((2 + g) ^ ((r >> 1) & 7)) * 16)); }},
    {"row = lane&31, 144 B pitch, slot = g", [](int l) { int r = l & 31, g = l >> 5; return (unsigned)(r * 144 + g * 16); }},
    {"row = lane&31, 160 B pitch, slot = g", [](int l) { int r = l & 31, g = l >> 5; return (unsigned)(r * 160 + g * 16); }},
    {"row = lane&31, 80 B pitch, slot = g                    (conv fragment)", [](int l) { int r = l & 31, g = l >> 5; return (unsigned)(r * 80 + g * 16); }},
    {"conv fragment: 16 pixels x 80 B per row, rows 1440 B apart (natural 18-pixel halo row)", [](int l) { int p = l & 31, g = l >> 5; return (unsigned)((p & 15) * 80 + (p >> 4) * 1440 + g * 16); }},
    {"conv fragment: 16 pixels x 80 B per row, rows 1536 B apart (padded to 256 B)", [](int l) { int p = l & 31, g = l >> 5; return (unsigned)((p & 15) * 80 + (p >> 4) * 1536 + g * 16); }},
    {"conv fragment: rows 1472 B apart (1440 + 32)", [](int l) { int p = l & 31, g = l >> 5; return (unsigned)((p & 15) * 80 + (p >> 4) * 1472 + g * 16); }},
    {"conv fragment 4x16x64 / stride 1: rows 1440 B apart, 8 pixels x 4 rows?  (p&15, p>>4)", [](int l) { int p = l & 31, g = l >> 5; return (unsigned)((p & 15) * 80 + (p >> 4) * 1440 + g * 16); }},
    {"stride-2 conv fragment: 16 pixels x 160 B, rows 2 x 2640 B apart", [](int l) { int p = l & 31, g = l >> 5; return (unsigned)((p & 15) * 160 + (p >> 4) * 2 * 33 * 80 + g * 16); }},
    {"dense 1x1 tile: pixel pitch 80 B, 32 consecutive pixels", [](int l) { int p = l & 31, g = l >> 5; return (unsigned)(p * 80 + g * 16); }},
    {"row = lane&15 (128 B), slot = (lane>>4) ^ (r&7)", [](int l) { int r = l & 15, q = l >> 4; return (unsigned)(r * 128 + ((q ^ (r & 7)) * 16)); }},
    {"row = lane&31, slot = g ^ (r&7), second half rows +2048", [](int l) { int r = l & 31, g = l >> 5; return (unsigned)(r * 128 + g * 2048 + (((r & 7)) * 16)); }},
  };
  for (auto& p : ps) {
    std::vector<unsigned> o(64);
    for (int l = 0; l < 64; ++l) o[l] = p.f(l);
    printf("%-90s  1 wave: %6.1f   4 waves: %6.1f cycles per ds_read_b128\n", p.name, run(o, 64), run(o, 256));
  }
  return 0;
}
