// Internal declarations shared by the translation units of libpfhip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <atomic>
#include <string>
#include "../../include/pfhip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace pf {

int set_error(int code, const char* fmt, ...);

#define PF_CHECK_HIP(expr)                                                                   \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) return pf::set_error(PF_EHIP, "%s: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

#define PF_REQUIRE(cond, ...)                                   \
  do {                                                          \
    if (!(cond)) return pf::set_error(PF_EINVAL, __VA_ARGS__);  \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: set it once per (kernel, device) - a process may
// drive several GPUs - with one bit per device in a mask owned by the launcher (two racing threads both set it: harmless)
static inline int set_max_lds_once(const void* kern, int bytes, std::atomic<uint64_t>& done_mask) {
  int dev = 0;
  PF_CHECK_HIP(hipGetDevice(&dev));
  const uint64_t bit = 1ull << (dev & 63);
  if (done_mask.load(std::memory_order_acquire) & bit) return PF_OK;
  PF_CHECK_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  done_mask.fetch_or(bit, std::memory_order_release);
  return PF_OK;
}
// compute units of the current device (cached per device index; 256 on MI355X)
int num_cus();
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- launchers implemented in the .hip files (all enqueue on `stream`, never sync) ----
int launch_conv(const pf_conv_args& a, hipStream_t stream);
double conv_flops(const pf_conv_args& a);
int launch_conv_bf3(const pf_conv_args& a, hipStream_t stream);  // called by launch_conv after validation
int conv_pick_tile(const pf_conv_args& a);                        // 0: 128px x 128ch, 1: 128px x 64ch, 2: 64px x 64ch
void conv_tile_shape(const pf_conv_args& a, int tile, int* th, int* tw);
int conv_stats_tiles(const pf_conv_args& a);
int conv_ksplit(const pf_conv_args& a);                           // K-split factor this launch uses (1 = none); needs a.splitk_ws
size_t conv_splitk_ws_bytes(const pf_conv_args& a);              // scratch wanted for the split (0 = would not split)                      // per-sample tiles emitted into stats_out
// both return false when a weight does not fit the split's element type (fp16 build: |w| * 2^8 > 65504; the packing then holds the clamped value)
bool pack_gemm_bf3(void* dst, const float* src, int n_src, int K, int taps, int Npad, int n_off, const int* colmap);
bool pack_upfold_bf3(void* dst, const float* src, int N, int K, int Npad);   // UpSample conv weight -> 4 parities x 4 taps
// fused Winograd F(2x2, 3x3) form of the stride-1 3x3 conv (conv_wino.hip): eligibility of a launch, the launch, the weight transform + packing
bool conv_wino_eligible(const pf_conv_args& a);
int launch_conv_wino(const pf_conv_args& a, hipStream_t stream);
bool pack_wino_bf3(void* dst, const float* src, int N, int K);            // [N][K][3][3] -> U = G g G^T, hi | lo, MFMA B-operand order

int launch_attention(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo,
                     int batch, int n_heads, int d_head, int lq, int lk, hipStream_t stream);

// form: PF_OPT_AUTO | 0 = 128-query workgroups | 1 = 256-query workgroups (needs l % 256 == 0)
// scratch (optional): attention_bf3_split_floats(...) floats enable the key-split form for small batches (a second, merging launch)
int launch_attention_bf3(const void* planes, float* o, int ldo, void* o_planes, int batch, int n_heads, int l, int form, hipStream_t stream,
                         float* scratch = nullptr, size_t scratch_floats = 0);
size_t attention_bf3_split_floats(int batch, int n_heads, int l, int* nsplit_out);
int launch_gemm_planes(const pf_conv_args& a, hipStream_t stream);   // called by launch_conv when a.a_planes
// out = x + ff2(GeGLU(ff1(LayerNorm(x)))) for C = 256, hidden 1024, as one launch (mlp_fused_bf3.hip); w1 / w2 = bf16x3 packings
int launch_mlp_fused(const float* x, int batch, int l, const float* gamma, const float* beta, float eps, const void* w1, const float* b1,
                     const void* w2, const float* b2, float* out, void* out_planes, hipStream_t stream, const void* w3 = nullptr,
                     const float* b3 = nullptr, const float* res3 = nullptr, float* stats3 = nullptr);

size_t gn_scratch_bytes(int batch, int c, int hw);
int launch_gn_scale_shift(const float* x0, int c0, const float* x1, int c1, int batch, int hw, int groups, float eps,
                          const float* gamma, const float* beta, float* scale, float* shift, void* scratch,
                          size_t scratch_bytes, hipStream_t stream);
int launch_gn_finalize_tiles(const float* s0, int t0, int c0, const float* s1, int t1, int c1, int batch, int hw, int groups,
                             float eps, const float* gamma, const float* beta, float* scale, float* shift, hipStream_t stream, int bmod1 = 0);
// per-split (sum, sumsq) of an NHWC tensor in the tile-statistics layout [B][gn_nsplit(hw)][C][2] (for tensors whose producer emits none)
int gn_nsplit(int hw);
int launch_gn_partial(const float* x0, int c0, const float* x1, int c1, int batch, int hw, float* stats, hipStream_t stream);
int launch_ln_stats(const float* x, int rows, int c, float eps, float* mean, float* rstd, hipStream_t stream);
int launch_ln_planes(const float* x, int rows, int c, float eps, const float* gamma, const float* beta, void* planes,
                     hipStream_t stream);

// stem / head convolutions (NCHW <-> NHWC at the ABI edge)
int launch_conv_in(const float* x_nchw, const float* w /*[Cout][Cin][3][3]*/, const float* bias, float* out_nhwc,
                   int batch, int cin, int cout, int h, int w_, hipStream_t stream, float* stats = nullptr);
int launch_conv_in_stats_tiles(int cin, int cout, int h, int w_);   // per-sample statistics tiles launch_conv_in can emit (0: none)
int launch_conv_out(const float* x_nhwc, const float* sc, const float* sh, const float* w /*[9][Cin][Cout]*/,
                    const float* bias, float* out_nchw, int batch, int cin, int cout, int h, int w_, hipStream_t stream);

// time embedding: t[B] -> silu(time_embed(sinusoid(t))) [B][d_t]
// t == nullptr: row b is the embedding of time-step value b (the hoisted table of pf_unet_prepare_time)
int launch_time_embed(const int64_t* t, const float* w0, const float* b0, const float* w2, const float* b2,
                      float* out_silu, int batch, int channels, int d_t, hipStream_t stream);
// y[b][n] = sum_k W[n][k] * x[b][k] + bias[n]   (row-major W [N][K]; one wave per output)
// grouped form: output n reads x + (n / n_per_group) * x_group_stride (block-diagonal W); n_per_group <= 0 disables
int launch_prmat2c_durations(const float* x, int n, int steps, int custom_round, int32_t* dur, hipStream_t stream);
int launch_matvec(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int batch, int n, int k,
                  hipStream_t stream, int n_per_group = 0, int x_group_stride = 0);

// sampler elementwise kernels
int launch_cfg_combine(const float* eps2, float scale, float* eps, size_t n, hipStream_t s);
int launch_ddpm_step(const float* x, const float* eps, const float* noise_p, const float* noise_q, const float* orig,
                     const float* mask, const pf_ddpm_coef& c, float* out, size_t n, hipStream_t s);
int launch_axpby(const float* x, const float* y, float a, float b, float* out, size_t n, hipStream_t s);
int launch_ddim_step(const float* x, const float* eps, const float* noise, const float* orig, const float* orig_noise,
                     const float* mask, const pf_ddim_coef& c, float* out, size_t n, hipStream_t s);
int launch_randn(float* out, size_t n, uint64_t seed, uint64_t stream_id, uint64_t elem_offset, hipStream_t s);
int launch_randn_dev(float* out, size_t n, uint64_t seed, const pf_step_state* st, int slot, uint64_t elem_offset, hipStream_t s);
int launch_ddpm_step_dev(const float* x, const float* eps, const float* noise_p, const float* noise_q, const float* orig,
                         const float* mask, const pf_ddpm_coef* table, const pf_step_state* st, float* out, size_t n, hipStream_t s);
int launch_ddim_step_dev(const float* x, const float* eps, const float* noise, const float* orig, const float* orig_noise,
                         const float* mask, const pf_ddim_coef* table, const pf_step_state* st, float* out, size_t n, hipStream_t s);
// in-kernel Philox noise (draw indices by value, or - st != nullptr - from the device step state)
int launch_ddpm_step_rng(const float* x, const float* eps, const float* orig, const float* mask, const pf_ddpm_coef* c_host,
                         const pf_ddpm_coef* table, const pf_step_state* st, uint64_t seed, uint64_t draw_q, uint64_t draw_p, uint64_t off,
                         float* out, size_t n, hipStream_t s);
int launch_ddim_step_rng(const float* x, const float* eps, const float* orig, const float* orig_noise, const float* mask,
                         const pf_ddim_coef* c_host, const pf_ddim_coef* table, const pf_step_state* st, uint64_t seed, uint64_t draw,
                         uint64_t off, float* out, size_t n, hipStream_t s);
int launch_clock_probe(unsigned long long* out2, hipStream_t s);
int launch_mfma_probe(float* sink, int iters, double* flops, hipStream_t s);
int launch_step_state_set(pf_step_state* st, int64_t index, uint64_t draws, hipStream_t s);
int launch_step_begin(const pf_step_state* st, const int* time_steps, int64_t* t_out, int batch, hipStream_t s);
int launch_step_end(pf_step_state* st, int draws_used, hipStream_t s);

// encoder kernels
int launch_gru_gates(const float* gi, int ld_gi, const float* gh, float* h, int ld_h, int batch, int hidden, hipStream_t s);
int launch_pnotree_embed(const float* grid, const float* w, const float* bias, float* out, int rows, int emb, int pitch_range, hipStream_t s);
int launch_pnotree_lengths(const float* grid, int* lens, int nseq, int max_simu_note, int pad, hipStream_t s);
int launch_gru_gates_masked(const float* gi, const float* gh, float* h, int ld_h, int nseq, int hidden, int seq_len, const int* lens,
                            int step, int reverse, hipStream_t s);
int launch_txt_frontend(const float* pr, const float* w, const float* bias, float* out, int batch, int num_channel,
                        hipStream_t s);

}  // namespace pf

// ---- element type of the split-precision ("x3") kernels ---------------------------------------------------------------------------------
// The default build splits every operand into two bf16 pieces (8 + 8 mantissa bits, fp32's exponent range).  -DPF_X3_F16 compiles the very
// same kernels with fp16 pieces (11 + 11 bits, three v_mfma_f32_32x32x16_f16 per product at the same rate): ~16x less rounding error per
// product, paid for with fp16's range - |activation| must stay below 65504 and pieces below 6e-5 lose bits (fp16 subnormals ARE honoured by
// the matrix pipe, tools/micro/f16x3_probe.hip).  Weights are small numbers (|w| ~ 0.03 puts the lo piece deep in the subnormals), so that
// build packs them times PF_X3_WS = 2^8 and every epilogue takes its accumulators times 2^-8 - exact, and folded into the bias add as an fma.
// The kernels spell the element type x3_t, its vectors x3x2 / x3x4 / x3x8 (typedef'd where they are used) and the MFMA x3_mfma_32x32x16.
#ifdef PF_X3_F16
typedef _Float16 x3_t;
#define x3_mfma_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define PF_X3_WS 256.0f
#define PF_X3_WS_INV 0.00390625f
#define PF_X3_UNSCALE(a) ((a) * PF_X3_WS_INV)
#else
typedef __bf16 x3_t;
#define x3_mfma_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define PF_X3_WS 1.0f
#define PF_X3_WS_INV 1.0f
#define PF_X3_UNSCALE(a) (a)
#endif
// accumulator tile -> true units (a no-op in the bf16 build)
template <int FM, int FN>
__device__ __forceinline__ void x3_unscale(f32x16 (&acc)[FM][FN]) {
#ifdef PF_X3_F16
#pragma unroll
  for (int fm = 0; fm < FM; ++fm)
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) acc[fm][fn] = acc[fm][fn] * PF_X3_WS_INV;
#endif
}
