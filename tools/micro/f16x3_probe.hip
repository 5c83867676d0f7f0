// Is an fp16 hi/lo split ("f16x3": 22 mantissa bits, same three MFMAs per product as bf16x3) viable on gfx950?
//   (1) does v_mfma_f32_32x32x16_f16 honour fp16 SUBNORMAL inputs (the lo piece of |x| < 0.125 is subnormal)?
//   (2) error of a K = 2304 dot product with operands like the UNet's (activations O(1), weights O(0.03)) for bf16x3 / f16x3 vs double.
// build: hipcc --offload-arch=gfx950 -O3 -o f16x3_probe tools/micro/f16x3_probe.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

// one wave: C[32x32] = A[32xK] . B[Kx32], A row-major [32][K], B as [32 n][K]; lane (r = lane & 31, g = lane >> 5) holds k = 16 s + 8 g .. +7
template <bool F16>
__global__ void gemm32(const float* A, const float* B, int K, float* C) {
  const int lane = threadIdx.x, r = lane & 31, g = lane >> 5;
  f16v acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int s = 0; s < K / 16; ++s) {
    float a[8], b[8];
    for (int j = 0; j < 8; ++j) { a[j] = A[r * K + 16 * s + 8 * g + j]; b[j] = B[r * K + 16 * s + 8 * g + j]; }
    if constexpr (F16) {
      h8 ah, al, bh, bl;
      for (int j = 0; j < 8; ++j) { ah[j] = (_Float16)a[j]; al[j] = (_Float16)(a[j] - (float)ah[j]); bh[j] = (_Float16)b[j]; bl[j] = (_Float16)(b[j] - (float)bh[j]); }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
    } else {
      b8 ah, al, bh, bl;
      for (int j = 0; j < 8; ++j) { ah[j] = (__bf16)a[j]; al[j] = (__bf16)(a[j] - (float)ah[j]); bh[j] = (__bf16)b[j]; bl[j] = (__bf16)(b[j] - (float)bh[j]); }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    }
  }
  // C/D layout: col = lane & 31 (the B row n), row = (i & 3) + 8 (i >> 2) + 4 g (the A row m)
  for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * g) * 32 + r] = acc[i];
}
__global__ void denorm_probe(float* out) {
  h8 a, b; for (int j = 0; j < 8; ++j) { a[j] = (_Float16)0; b[j] = (_Float16)0; }
  a[0] = (_Float16)3.0e-6f;   // subnormal in fp16 (min normal 6.1e-5)
  b[0] = (_Float16)1024.f;
  f16v acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)a[0]; }
}
int main() {
  float* d; hipMalloc(&d, 64); denorm_probe<<<1, 64>>>(d); float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
  printf("subnormal probe: fp16(3.0e-6) = %.6e as stored; MFMA(3.0e-6 x 1024) = %.6e (honoured if ~3.07e-3, flushed if 0)\n", h[1], h[0]);
  for (int which = 0; which < 3; ++which) {
    const int K = 2304; std::mt19937 rng(7 + which); std::normal_distribution<float> nd(0.f, 1.f);
    const float as = which == 2 ? 30.f : 1.f, ws = which == 1 ? 0.002f : 0.03f;
    std::vector<float> A(32 * K), B(32 * K); for (auto& v : A) v = as * nd(rng); for (auto& v : B) v = ws * nd(rng);
    float *dA, *dB, *dC; hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    std::vector<double> ref(1024); double scale = 0;
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { double s = 0; for (int k = 0; k < K; ++k) s += (double)A[m * K + k] * B[n * K + k]; ref[m * 32 + n] = s; scale = fmax(scale, fabs(s)); }
    std::vector<float> C(1024);
    double e[2];
    for (int f = 0; f < 2; ++f) {
      if (f) gemm32<true><<<1, 64>>>(dA, dB, K, dC); else gemm32<false><<<1, 64>>>(dA, dB, K, dC);
      hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
      double m = 0; for (int i = 0; i < 1024; ++i) m = fmax(m, fabs(C[i] - ref[i])); e[f] = m / scale;
    }
    double ef = 0; for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { float s = 0; for (int k = 0; k < K; ++k) s += A[m * K + k] * B[n * K + k]; ef = fmax(ef, fabs(s - ref[m * 32 + n])); }
    printf("K = %d, |a| ~ %g, |w| ~ %g: max error / max|c|: bf16x3 %.2e   f16x3 %.2e   (plain fp32 loop %.2e)\n", K, as, ws, e[0], e[1], ef / scale);
  }
  return 0;
}
