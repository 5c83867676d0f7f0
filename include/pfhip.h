/* pfhip.h - C ABI of libpfhip.so: the MI355X (gfx950) denoising hot path of Polyffusion.
 *
 * The reference is pure Python/PyTorch and has no FFI for this path; its interface is the
 * call signature of the denoiser and of the sampler step.  Each entry point
 * below cites the reference callable it replaces (paths relative to
 * /root/reference/polyffusion).  INTEGRATION.md shows the ctypes stub a reference
 * maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative PF_E* code on failure; the message
 *     is available from pf_last_error() (thread-local).  No exceptions cross the ABI.
 *   - all device buffers are owned by the caller (allocated e.g. through torch); the
 *     library never allocates or frees device memory and never synchronises the device.
 *   - kernels are enqueued on the caller's hipStream_t (passed as void*), so they order with
 *     the caller's other work and can be captured into a hipGraph.
 *   - activations at the ABI are fp32.  x / eps are NCHW [B,C,H,W] exactly as the reference
 *     passes them; internal activations are NHWC and never leave the workspace.
 */
#ifndef PFHIP_H
#define PFHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PF_OK 0
#define PF_EINVAL (-1)   /* bad argument / unsupported shape */
#define PF_ENOTFOUND (-2) /* unknown parameter key */
#define PF_ESTATE (-3)   /* call order violated (e.g. forward before bind) */
#define PF_EHIP (-4)     /* HIP runtime error */

int pf_version(void);
const char* pf_last_error(void);

/* ---- denoiser (replaces UNetModel.__init__/forward, stable_diffusion/model/unet.py:30-196,
 *      incl. SpatialTransformer, unet_attention.py:26-333) ---------------------------------- */
typedef struct pf_unet pf_unet;

typedef struct pf_unet_cfg {
  int32_t in_channels, out_channels, channels, n_res_blocks;
  int32_t n_attention_levels;
  int32_t attention_levels[8];
  int32_t n_levels;
  int32_t channel_multipliers[8];
  int32_t n_heads, tf_layers, d_cond;
  int32_t img_h, img_w; /* spatial size of x (params.img_h/img_w) */
} pf_unet_cfg;

int pf_unet_create(const pf_unet_cfg* cfg, pf_unet** out);
void pf_unet_destroy(pf_unet* u);

/* Weight ingestion in the reference state_dict namespace (keys relative to ldm.eps_model.,
 * SURVEY.md Appendix D; replaces nn.Module.load_state_dict for unet.py).  The caller owns a
 * host staging blob of pf_unet_weight_bytes(); pf_unet_pack_param repacks ONE named tensor
 * (fp32, contiguous, torch layout) into its kernel-friendly place inside that blob.
 * pf_unet_pack_missing returns the number of keys not yet packed (0 = complete) and writes
 * the first missing key into buf.  The packed blob is then copied to the device by the
 * caller (or received by RCCL broadcast) and attached with pf_unet_bind_weights. */
size_t pf_unet_weight_bytes(const pf_unet* u);
int pf_unet_n_params(const pf_unet* u);
int pf_unet_param_info(const pf_unet* u, int i, char* key_buf, size_t key_buf_len, int64_t shape[4], int* ndim);
int pf_unet_pack_param(pf_unet* u, const char* key, const float* src, const int64_t* shape, int ndim, void* host_blob);
int pf_unet_pack_missing(const pf_unet* u, char* buf, size_t buf_len);
int pf_unet_bind_weights(pf_unet* u, const void* dev_blob);

/* Workspace (activations, skips, statistics) for a given batch / number of context tokens. */
size_t pf_unet_workspace_bytes(const pf_unet* u, int batch, int n_cond);

/* eps = UNet(x, t, cond): x [B,Cin,H,W] f32, t [B] i64, cond [B,n_cond,d_cond] f32,
 * eps [B,Cout,H,W] f32 (unet.py:171-196; LatentDiffusion.forward latent_diffusion.py:138-147). */
int pf_unet_forward(pf_unet* u, const float* x, const int64_t* t, const float* cond, int batch, int n_cond,
                    float* eps, void* workspace, size_t workspace_bytes, void* stream);

/* ---- step-invariant prefix, hoisted out of the reverse loop -----------------------------------------------------------
 * Two parts of UNetModel.forward do not depend on x: the time MLP + every ResBlock's emb_layers (unet.py:181-182, 286-289:
 * a function of t only) and, with ONE context token, the whole cross-attention (unet_attention.py:186-212, 261-293: softmax over
 * one key is 1, so attn2(x, c) = to_out(to_v(c)) for every position: a function of cond only).  A sampler knows every t of its
 * loop in advance and holds cond fixed, so it computes both ONCE and hands them to every forward of the loop:
 *   pf_unet_prepare_time - rows r = 0 .. n_rows-1 of `table` [n_rows][pf_unet_time_bias_width] = the additive time biases of all
 *                          ResBlocks for time-step VALUE r (what forward computes from t == r); scratch: n_rows * 4 * channels floats
 *   pf_unet_prepare_cond - `cross` [batch][pf_unet_cross_bias_width] = to_out(to_v(cond[b])) + bias of every transformer block
 *                          (n_cond == 1 only); scratch: batch * width floats
 * Both buffers are caller-owned device memory; nothing is cached inside the library.  pf_unet_forward_prepared with prep == NULL
 * (or with NULL members) computes the missing part itself, exactly like pf_unet_forward; results are bit-identical either way.
 * With a time table every t[b] must lie in [0, n_time_rows) (values outside are clamped for memory safety). */
typedef struct pf_unet_prepared {
  const float* time_table; int32_t n_time_rows;   /* from pf_unet_prepare_time, or NULL */
  const float* cross_bias;                         /* from pf_unet_prepare_cond for THIS cond / batch, or NULL */
} pf_unet_prepared;
int pf_unet_time_bias_width(const pf_unet* u);
int pf_unet_cross_bias_width(const pf_unet* u);
int pf_unet_prepare_time(pf_unet* u, int n_rows, float* table, void* scratch, size_t scratch_bytes, void* stream);
int pf_unet_prepare_cond(pf_unet* u, const float* cond, int batch, float* cross, void* scratch, size_t scratch_bytes, void* stream);
int pf_unet_forward_prepared(pf_unet* u, const float* x, const int64_t* t, const float* cond, int batch, int n_cond,
                             const pf_unet_prepared* prep, float* eps, void* workspace, size_t workspace_bytes, void* stream);
/* Classifier-free guidance (stable_diffusion/sampler/__init__.py:63-77 `get_eps`: the model is evaluated on cat([x, x]), cat([t, t]),
 * cat([uncond_cond, cond])).  The condition enters the UNet only through the cross-attention of its transformer blocks (unet.py:181-196,
 * unet_attention.py:240-246), so everything in front of the first SpatialTransformer - the stem, the ResBlocks and DownSamples of the levels
 * without attention, the first ResBlock of the first attention level - is IDENTICAL for the two halves.  This entry computes that prefix once on
 * the `batch2 / 2` samples of `x`, shares the skip tensors it produced between the halves (a sample index taken modulo batch2 / 2 inside the
 * consuming kernels: no copies) and runs the rest on all batch2 samples: the same eps2 [batch2, ...] as pf_unet_forward_prepared on the
 * concatenated inputs up to tile-choice rounding, for 13 % fewer FLOPs at sdf_chd8bar.  x: [batch2 / 2, ...]; t, cond (and prep->cross_bias): batch2
 * rows, second half = first half for t.  Workspace: pf_unet_workspace_bytes_cfg(u, batch2, n_cond) bytes. */
size_t pf_unet_workspace_bytes_cfg(const pf_unet* u, int batch2, int n_cond);
int pf_unet_forward_cfg(pf_unet* u, const float* x, const int64_t* t, const float* cond, int batch2, int n_cond,
                        const pf_unet_prepared* prep, float* eps2, void* workspace, size_t workspace_bytes, void* stream);
int pf_unet_n_launches_cfg(const pf_unet* u, int batch2, int n_cond, int has_time, int has_cross);
/* kernel launches of one forward; has_time / has_cross: with that member of pf_unet_prepared supplied */
int pf_unet_n_launches_prepared(const pf_unet* u, int batch, int n_cond, int has_time, int has_cross);

/* Plan options, per handle (default PF_OPT_AUTO for all): which of two equivalent kernel forms the plan launches.
 *   PF_OPT_MLP_FUSED  - transformer feed-forward (+ proj_out) as ONE launch per 64-token tile vs LayerNorm planes + GeGLU GEMM + FF-out
 *                       GEMM + proj_out (bit-identical results); auto: fused when its last round of workgroups is >= 85 % full
 *   PF_OPT_ATTN_WIDE  - self-attention with 256-query vs 128-query workgroups (equal up to summation order); auto: 256 when L % 256 == 0
 *                       and that still gives 3/4 of the CUs a workgroup
 *   PF_OPT_CONV_T16   - 16x16-pixel tile for the 64-output-channel 3x3 convs with >= 128 input channels vs the 8x16 tile
 *   PF_OPT_CONV_PP    - 3x3 convs whose 128-wide tiles give every CU at most one workgroup (B = 16: the 32x32 level) as 8-wave workgroups:
 *                       two wave groups split K and run half a tap apart, one loading while the other computes (equal up to summation order)
 *   PF_OPT_CONV_WINO  - ResBlock 3x3 convs in the fused Winograd F(2x2, 3x3) form (pf_conv_args.wino; bf16x3 / f16x3 modes; equal to the direct
 *                       form up to rounding, 2.25x fewer matrix operations) vs the direct implicit GEMM; auto: where it measured faster
 *                       (profiles/r06_ab_winograd.md: input channels >= 192, or >= 128 at 32x32 and below); on: every conv that qualifies */
enum { PF_OPT_MLP_FUSED = 0, PF_OPT_ATTN_WIDE = 1, PF_OPT_CONV_T16 = 2, PF_OPT_CONV_PP = 3, PF_OPT_CONV_WINO = 4, PF_OPT_COUNT = 5 };
enum { PF_OPT_AUTO = -1, PF_OPT_OFF = 0, PF_OPT_ON = 1 };
int pf_unet_set_option(pf_unet* u, int option, int value);
/* Range telemetry.  `device_word` (4 bytes of caller-owned device memory, zeroed by the caller; NULL switches it off): while bound, every forward
 * max-combines into it - as an fp32 bit pattern, NaN / inf kept - the largest |value| of every tensor its layers store: block outputs, the residual
 * stream, q | k | v, GeGLU products (the fused feed-forward / pre-attention launches are replaced by their bit-identical chains while it is bound).
 * Each of those is, after at most a normalisation, the split A operand of a following layer; the fp16-piece build (f16x3) overflows beyond 65504,
 * so 65504 / *device_word is the run's headroom - measured in ANY arithmetic mode (in f32 nothing overflows while measuring).  inference_sdf's
 * `--precision auto` reads it from its f32 probe before it considers f16x3; bench.py prints it (f16x3_mode.max_abs_activation). */
int pf_unet_track_absmax(pf_unet* u, void* device_word);
int pf_unet_get_option(const pf_unet* u, int option);

/* Per-launch profiling: when enabled, forward brackets every kernel launch with hipEvents
 * on the launch stream.  pf_unet_profile_read synchronises those events and returns, per
 * launch: a kernel-family id (PF_K_*), elapsed ms and the algorithmic FLOPs of the launch. */
enum { PF_K_CONV3 = 0, PF_K_GEMM = 1, PF_K_ATTN = 2, PF_K_GNSTAT = 3, PF_K_LNSTAT = 4, PF_K_SMALL = 5, PF_K_COUNT = 6 };
int pf_unet_set_profiling(pf_unet* u, int enabled);
int pf_unet_profile_read(pf_unet* u, int* kind, float* ms, double* flops, int capacity);
/* the same launches' operation counts in the DIRECT form of each layer: equal to `flops` except for Winograd launches (PF_OPT_CONV_WINO), whose
 * `flops` are the 16 / 36 actually executed - keeps achieved rates comparable between plans (bench.py: direct_equivalent_tflops) */
int pf_unet_profile_read_direct(pf_unet* u, double* direct_flops, int capacity);
int pf_unet_n_launches(const pf_unet* u, int batch, int n_cond);

/* ---- sampler steps (elementwise, NCHW fp32, n = B*C*H*W elements) -------------------------
 * Classifier-free guidance combine (DiffusionSampler.get_eps, stable_diffusion/sampler/__init__.py:69-77):
 *   eps = e_uncond + scale * (e_cond - e_uncond), where eps2 = [e_uncond ; e_cond] (2n elements). */
int pf_cfg_combine(const float* eps2, float scale, float* eps, size_t n, void* stream);

/* One DDPM RePaint iteration (SDFSampler.paint body + p_sample, sampler_sdf.py:80-171,307-336):
 *   x0    = c_recip*x - c_recipm1*eps ; mean = c_x0*x0 + c_xt*x ; x_unkn = mean + sigma*noise_p
 *   x_kn  = sqrt_ab*orig + sqrt_1mab*noise_q          (skipped when orig == NULL)
 *   x_out = x_kn*mask + x_unkn*(1-mask)
 * noise_p / noise_q may be NULL (treated as zero: step 0).  x_out may alias x. */
typedef struct pf_ddpm_coef { float c_recip, c_recipm1, c_x0, c_xt, sigma, sqrt_ab, sqrt_1mab; } pf_ddpm_coef;
int pf_ddpm_step(const float* x, const float* eps, const float* noise_p, const float* noise_q,
                 const float* orig, const float* mask, const pf_ddpm_coef* c, float* x_out, size_t n, void* stream);

/* RePaint re-noise between inner repeats (sampler_sdf.py:337-341; keeps the reference's beta-not-sqrt quirk):
 *   x_t = a*x + b*noise */
int pf_axpby(const float* x, const float* noise, float a, float b, float* out, size_t n, void* stream);

/* One DDIM iteration (DDIMSampler.get_x_prev_and_pred_x0 + paint blend, sampler_ddim.py:220-272,355-359):
 *   pred_x0 = (x - s1m*eps)/sqrt_a ; x_prev = sqrt_aprev*pred_x0 + dir_coef*eps + sigma*noise
 *   if orig: x_prev = (q_sqrt_a*orig + q_s1m*orig_noise)*mask + x_prev*(1-mask) */
typedef struct pf_ddim_coef { float s1m, sqrt_a, sqrt_aprev, dir_coef, sigma, q_sqrt_a, q_s1m; } pf_ddim_coef;
int pf_ddim_step(const float* x, const float* eps, const float* noise, const float* orig, const float* orig_noise,
                 const float* mask, const pf_ddim_coef* c, float* x_out, size_t n, void* stream);

/* Standard-normal noise, counter-based (Philox4x32-10 + Box-Muller), keyed by (seed, stream_id,
 * global element index) so results do not depend on how a batch is sharded over GPUs.
 * elem_offset = index of out[0] in the global (unsharded) tensor. */
int pf_randn(float* out, size_t n, uint64_t seed, uint64_t stream_id, uint64_t elem_offset, void* stream);

/* The same updates with the noise drawn INSIDE the kernel (no noise tensor in HBM, two launches per step less): element i of draw d is
 * exactly what pf_randn(seed, d, elem_offset) writes at i, so results are bit-identical to pf_randn + pf_ddpm_step / pf_ddim_step.
 * DDPM: noise_q = draw `draw_q` (only when orig != NULL), noise_p = draw `draw_p` - the reference's order is q then p
 * (sampler_sdf.py:317-321, 153-160).  DDIM: noise = draw `draw` (sigma != 0 steps only; orig_noise stays a tensor: the reference's DDIM
 * paint re-uses one fixed tensor, sampler_ddim.py:355-359).  n and elem_offset must be multiples of 4 (one Philox call = 4 normals). */
int pf_ddpm_step_rng(const float* x, const float* eps, const float* orig, const float* mask, const pf_ddpm_coef* c,
                     uint64_t seed, uint64_t draw_q, uint64_t draw_p, uint64_t elem_offset, float* x_out, size_t n, void* stream);
int pf_ddim_step_rng(const float* x, const float* eps, const float* orig, const float* orig_noise, const float* mask, const pf_ddim_coef* c,
                     uint64_t seed, uint64_t draw, uint64_t elem_offset, float* x_out, size_t n, void* stream);

/* Measurement aid (bench.py): writes {shader-cycle counter, constant-rate reference counter} of the moment the probe kernel runs into
 * row x of out[8][2] for every XCD x (XCDs have counters and clocks of their own; rows of XCDs the part does not have stay
 * untouched); two probes around a stretch of stream work give the average shader clock each XCD sustained over it. */
int pf_clock_probe(uint64_t* out8x2, void* stream);
/* Measurement aid (bench.py): one launch that keeps the bf16 matrix pipe of every CU 100 % busy (8 waves per CU, dependency-free
 * v_mfma_f32_32x32x16_bf16 streams on operands with random signs / mantissas) for `iters` x 24 MFMAs per wave; *flops_out = the launch's
 * flops.  Timed with events it gives the rate the pipe SUSTAINS on this box under its power management - the reference
 * `roofline.sustained` of the bench line; `sink` is one float of device memory (never written in practice). */
int pf_mfma_probe(float* sink, int iters, double* flops_out, void* stream);

/* ---- replayable reverse step: everything that changes from one step to the next - the table row, the
 * time-step value fed to the denoiser, the noise draw counter - lives in a small device-resident state, so ONE captured
 * hipGraph of {begin, randn, pf_unet_forward, randn, step, end} is replayed for every step of the loop.  `table` is the
 * per-step coefficient table on the device (one pf_ddpm_coef / pf_ddim_coef per row), `time_steps` the DDIM tau table
 * (int32 per row; NULL: the row index itself is the time step, as in the DDPM sampler). */
typedef struct pf_step_state { int64_t index; uint64_t draws; } pf_step_state;
int pf_step_state_set(pf_step_state* dev_state, int64_t index, uint64_t draws, void* stream);
int pf_step_begin(const pf_step_state* dev_state, const int32_t* time_steps, int64_t* t_out, int batch, void* stream);
int pf_step_end(pf_step_state* dev_state, int draws_used, void* stream);          /* index -= 1; draws += draws_used */
int pf_randn_dev(float* out, size_t n, uint64_t seed, const pf_step_state* dev_state, int slot, uint64_t elem_offset, void* stream);
int pf_ddpm_step_dev(const float* x, const float* eps, const float* noise_p, const float* noise_q, const float* orig,
                     const float* mask, const pf_ddpm_coef* table, const pf_step_state* dev_state, float* x_out, size_t n, void* stream);
int pf_ddim_step_dev(const float* x, const float* eps, const float* noise, const float* orig, const float* orig_noise,
                     const float* mask, const pf_ddim_coef* table, const pf_step_state* dev_state, float* x_out, size_t n, void* stream);
/* in-kernel noise, draw indices from the device state: DDPM q = state.draws (orig != NULL only), p = the next one; DDIM = state.draws */
int pf_ddpm_step_rng_dev(const float* x, const float* eps, const float* orig, const float* mask, const pf_ddpm_coef* table,
                         const pf_step_state* dev_state, uint64_t seed, uint64_t elem_offset, float* x_out, size_t n, void* stream);
int pf_ddim_step_rng_dev(const float* x, const float* eps, const float* orig, const float* orig_noise, const float* mask,
                         const pf_ddim_coef* table, const pf_step_state* dev_state, uint64_t seed, uint64_t elem_offset, float* x_out,
                         size_t n, void* stream);

/* ---- weight broadcast over RCCL / xGMI.  The path has ONE exchange: rank 0 ships the packed weight
 * blob at start-up; the step loop has no collective.  librccl.so is opened on first use (dlopen), so a single-GPU process
 * never loads it.  `unique_id` is the 128-byte ncclUniqueId: rank 0 obtains it with pf_comm_unique_id and hands it to the
 * other ranks out of band (file / TCP store), exactly like an ncclUniqueId. */
typedef struct pf_comm pf_comm;
int pf_comm_unique_id(void* unique_id_out128);
int pf_comm_init(const void* unique_id128, int rank, int nranks, pf_comm** out);
int pf_comm_bcast(pf_comm* comm, void* dev_buf, size_t bytes, int root, void* stream);
int pf_comm_destroy(pf_comm* comm);

/* ---- output step: generated onset/sustain image -> note durations ----
 * Replaces the triple Python loop of utils.py:240-269 (prmat2c_to_prmat) and the note extraction of
 * utils.py:433-476 (prmat2c_to_midi_file): dur[n][t][key] = length in steps of the note that starts at (t, key), 0 where no
 * onset.  prmat2c is [n][2][steps][128] fp32 on the device (channel 0 onset, 1 sustain), dur is [n][steps][128] int32.
 * Rounding is the reference's int(round(v)) > 0 (== v > 0.5); custom_round != 0 selects utils.py:395-399 for the onset. */
int pf_prmat2c_durations(const float* prmat2c, int n, int steps, int custom_round, int32_t* dur, void* stream);

/* ---- condition encoders (replace RnnEncoder.forward dl_modules/chord_enc.py:15-22, TextureEncoder.forward
 *      dl_modules/txt_enc.py:23-35 and PianoTreeEncoder.forward dl_modules/pianotree_enc.py:97-121; only the Normal's mean is produced)
 *      PF_ENC_PNOTREE: input_dim = pitch classes + 5 duration digits (135), emb_size = note embedding (128), num_channel = hidden size
 *      of the per-step note GRU (256), hidden_dim = hidden size of the GRU over the 32 steps (512), z_dim 512 */
typedef struct pf_encoder pf_encoder;
enum { PF_ENC_CHORD = 0, PF_ENC_TEXTURE = 1, PF_ENC_PNOTREE = 2 };
int pf_encoder_create(int kind, int input_dim, int emb_size, int hidden_dim, int z_dim, int num_channel, pf_encoder** out);
void pf_encoder_destroy(pf_encoder* e);
size_t pf_encoder_weight_bytes(const pf_encoder* e);
int pf_encoder_pack_param(pf_encoder* e, const char* key, const float* src, const int64_t* shape, int ndim, void* host_blob);
int pf_encoder_pack_missing(const pf_encoder* e, char* buf, size_t buf_len);
int pf_encoder_bind_weights(pf_encoder* e, const void* dev_blob);
size_t pf_encoder_workspace_bytes(const pf_encoder* e, int batch);
/* chord:   x [B,T,input_dim] -> mu [B,z_dim];   texture: x [B,32,128] -> mu [B,z_dim];
 * pnotree: x [B,32,n_step,6] - the index grid (pitch index, 5 duration digits) of B two-bar segments with n_step = max_simu_note
 *          entries per time step, stored as floats - -> mu [B,z_dim] */
int pf_encoder_forward(pf_encoder* e, const float* x, int batch, int n_step, float* mu,
                       void* workspace, size_t workspace_bytes, void* stream);

/* Arithmetic of the dense contractions (convs / linears):
 *   PF_PREC_F32    - v_mfma_f32_32x32x2_f32, exact fp32 FMA chains (157 TFLOP/s pipe);
 *   PF_PREC_BF16X3 - error-compensated split on the bf16 pipe: a.w ~= a_hi.w_hi + a_hi.w_lo + a_lo.w_hi with fp32
 *                    accumulation (~1e-5 relative per product, three v_mfma_f32_32x32x16_bf16 instead of eight fp32 MFMAs).
 * Both packings of every weight live in the blob; the mode can be switched between calls. */
enum { PF_PREC_F32 = 0, PF_PREC_BF16X3 = 1 };
int pf_unet_set_precision(pf_unet* u, int precision);
int pf_unet_get_precision(const pf_unet* u);
/* Element type of the split: the library is built twice from the same sources.  libpfhip.so splits into bf16 pieces (8 + 8 mantissa
 * bits, fp32's exponent range: "bf16x3", unit roundoff ~2^-18).  libpfhip_f16.so (-DPF_X3_F16) splits into fp16 pieces (11 + 11 bits,
 * three v_mfma_f32_32x32x16_f16 per product: "f16x3", ~2^-22 - the error of the fp32 mode - at ~0.97 of bf16x3's speed), with fp16's
 * range: activations must stay below 65504 in magnitude (weights below 255; they are stored times 2^8 and every epilogue undoes it).
 * In that library every entry point and constant named bf16x3 means f16x3; weight packings and plane buffers of the two libraries are
 * not interchangeable.  Returns 0 (bf16 pieces) or 1 (fp16 pieces). */
int pf_x3_element(void);

/* ---- per-op entry points (unit-testable kernels; same kernels the plan launches) ---------- */
/* Host helper: torch weight [N, K, kh, kw] (kh=kw=1 or 3) -> packed [kh*kw][K/4][Npad][4], Npad = roundup(N,64). */
size_t pf_packed_gemm_weight_floats(int n, int k, int taps);
int pf_pack_gemm_weight(const float* w, int n, int k, int taps, float* dst);
/* same bytes, bf16x3 packing [kh*kw][K/8][hi|lo][Npad][8] (bf16); needs k % 8 == 0 */
int pf_pack_gemm_weight_bf16x3(const float* w, int n, int k, int taps, void* dst);
/* [n][k][3][3] conv weight of an UpSample layer -> folded packing [4 parities x 4 taps][k/8][plane][Npad][8] (host; k % 8 == 0) */
int pf_pack_upfold_weight_bf16x3(const float* w, int n, int k, void* dst);
/* Winograd F(2x2, 3x3) packing of a 3x3 conv weight [n][k][3][3] (n % 64 == 0, k % 16 == 0) for pf_conv_args.w_wino: U = G g G^T per (n, k),
 * split into hi | lo pieces, laid out in matrix-operand order; pf_wino_weight_bytes(n, k) = 16 * k * n * 4 bytes */
size_t pf_wino_weight_bytes(int n, int k);
int pf_pack_wino_weight_bf16x3(const float* w, int n, int k, void* dst);

/* GroupNorm statistics on NHWC (optionally the channel-concat of two tensors) -> per-(b,c)
 * scale/shift so that y = x*scale + shift equals GroupNorm(x) (unet.py:321-336; eps 1e-5 / 1e-6). */
int pf_gn_scale_shift(const float* x0, int c0, const float* x1, int c1, int batch, int hw, int groups, float eps,
                      const float* gamma, const float* beta, float* scale, float* shift,
                      void* scratch, size_t scratch_bytes, void* stream);
/* LayerNorm statistics over the last dim: mean[m], rstd[m] (unet_attention.py:104-110, eps 1e-5). */
int pf_ln_stats(const float* x, int rows, int c, float eps, float* mean, float* rstd, void* stream);
/* LayerNorm applied and split into bf16 hi/lo planes [rows][c] | [rows][c] (the A operand of pf_conv2d with a_planes=1);
 * replaces pf_ln_stats + the LayerNorm GEMM prologue for the transformer block's q/k/v and GeGLU projections */
int pf_ln_planes(const float* x, int rows, int c, float eps, const float* gamma, const float* beta, void* planes, void* stream);
/* The feed-forward half of BasicTransformerBlock as ONE launch (unet_attention.py:119-124 `x = ff(norm3(x)) + x`, :296-333 FeedForward /
 * GeGLU) for d_model 256, hidden 1024, bf16x3 arithmetic:  out = x + W2 . (a * gelu(g)) + b2 with [a|g] = W1 . LayerNorm(x) + b1.
 * x fp32 [batch*l][256], l % 64 == 0.  w1_bf16x3: the bf16x3 packing (pf_pack_gemm_weight_bf16x3) of ff.net.0.proj with its 2048 rows
 * reordered value/gate-interleaved in blocks of 32 (row 64 i + j <- value 32 i + j, row 64 i + 32 + j <- gate 32 i + j), b1 in the same
 * order; w2_bf16x3: packing of ff.net.2 ([256][1024]).  The result goes to fp32 `out` or, when out_planes != NULL, to bf16 hi/lo planes
 * [M][256] | [M][256].  Bit-identical to pf_ln_planes + pf_conv2d(geglu, out_planes) + pf_conv2d(a_planes, res = x). */
int pf_mlp_geglu_fused(const float* x, int batch, int l, const float* ln_gamma, const float* ln_beta, float ln_eps,
                       const void* w1_bf16x3, const float* b1, const void* w2_bf16x3, const float* b2,
                       float* out, void* out_planes, void* stream);
/* The same launch with the SpatialTransformer's closing 1x1 convolution chained on (unet_attention.py:77-79 `proj_out(x) + x_in` after the
 * last transformer layer): out = res3 + b3 + W3 . (x + ff(norm3(x))), the ff result never leaving the chip.  w3_bf16x3: bf16x3 packing
 * of proj_out ([256][256]); res3: the block input [batch*l][256]; stats3 (optional): per-64-row-tile channel (sum, sumsq) of `out`,
 * [batch][l/64][256][2], for the GroupNorm that consumes it.  Bit-identical to pf_mlp_geglu_fused(out_planes) + pf_conv2d(a_planes, res). */
int pf_mlp_geglu_proj_fused(const float* x, int batch, int l, const float* ln_gamma, const float* ln_beta, float ln_eps,
                            const void* w1_bf16x3, const float* b1, const void* w2_bf16x3, const float* b2,
                            const void* w3_bf16x3, const float* b3, const float* res3, float* out, float* stats3, void* stream);
/* Implicit-GEMM convolution / linear on NHWC with fused prologue and epilogue (fp32 MFMA).
 *   ks 1|3, stride 1|2, ups 0|1 (nearest x2 folded into the input read, unet.py:236-238)
 *   prologue: 0 none | 1 y=silu(x*sc+sh) | 2 y=x*sc+sh (sc,sh [B][Cin]) | 3 LayerNorm (mean,rstd per row; sc,sh = gamma,beta [Cin])
 *   epilogue: + bias[n] + sbias[b][n] + res[m][n]; geglu!=0: out[m][j] = v*gelu_erf(g) on interleaved halves */
typedef struct pf_conv_args {
  const float* x0; int32_t c0; const float* x1; int32_t c1; /* input = concat(x0, x1) along channels */
  int32_t batch, hin, win;                                   /* input spatial size per sample (dense: hin=1, win=rows per sample) */
  int32_t ks, stride, ups;
  const float* w; int32_t n;                                 /* packed weights, logical N */
  int32_t prologue; const float* sc; const float* sh; const float* mean; const float* rstd;
  const float* bias; const float* sbias; int32_t ld_sbias; const float* res; int32_t ld_res;
  int32_t geglu;
  float* out; int32_t ld_out;
  int32_t precision;                                         /* PF_PREC_F32 (w = fp32 packing) | PF_PREC_BF16X3 (w = bf16x3 packing) */
  float* stats_out;                                          /* optional [B][tiles][N][2] per-tile (sum, sumsq) of the outputs; see pf_conv_stats_tiles */
  void* splitk_ws; size_t splitk_ws_bytes;                   /* optional scratch for split-K (small-M layers); see pf_conv_splitk_ws_bytes */
  int32_t a_planes;                                          /* x0 is NOT fp32 but bf16 hi/lo planes [M][K] | [M][K] from a producer's out_planes (ks=1, prologue 0, bf16x3) */
  void* out_planes;                                          /* optional: write the result as bf16 hi/lo planes [M][ld_out] | [M][ld_out] instead of fp32 `out` */
  void* qkv_planes;                                          /* optional (ks=1, N = 3C, d_head 64): instead of `out`, write the fused q|k|v projection
                                                                as bf16 hi/lo planes [qh|ql|kh|kl|vTh|vTl] (each B*L*C) for pf_attention_bf16x3 */
  /* optional fused 1x1 projection of a second tensor, accumulated into the same output tile (the ResBlock's
   * skip_connection conv folded into its second 3x3 conv, unet.py:262-277): out += skip_w . concat(skip_x0, skip_x1) + skip_bias.
   * bf16x3, ks = 3, stride 1, no upsampling; skip_w is the bf16x3 packing of the [n][skip_c0+skip_c1] weight, channel
   * counts multiples of 32. */
  /* ups = 1 only: `w` is the PARITY-FOLDED bf16x3 packing (pf_pack_upfold_weight_bf16x3) - nearest x2 upsampling followed by a
   * 3x3 conv equals, per output parity (y&1, x&1), a 2x2 conv on the source grid with row/column-summed weights: 4/9 of the MACs */
  int32_t ups_fold;
  const float* skip_x0; int32_t skip_c0;
  const float* skip_x1; int32_t skip_c1;
  const void* skip_w;
  const float* skip_bias;
  /* optional (bf16x3, prologue 1 or 2): GroupNorm scale/shift computed INSIDE the launch from the producers' per-tile statistics
   * (what pf_gn_finalize_tiles computes in its own launch): every workgroup reduces its sample's statistics at start-up and writes
   * the sample's rows of `sc` / `sh` (identical values from every workgroup of the sample) before it reads them.  gn_stats0 != NULL
   * enables it; tensors with many tiles per sample (the 128x128 / 64x64 levels) are better served by the separate launch. */
  const float* gn_stats0; int32_t gn_tiles0;
  const float* gn_stats1; int32_t gn_tiles1;
  const float* gn_gamma; const float* gn_beta;
  float gn_eps; int32_t gn_groups;
  /* optional: row of `sbias` per sample - sbias[sbias_rows[b]] instead of sbias[b] (the hoisted time-bias table indexed by the
   * device-resident t[b]); rows outside [0, sbias_nrows) are clamped */
  const int64_t* sbias_rows; int32_t sbias_nrows;
  int32_t no_t16;          /* != 0: never pick the 16x16-pixel tile (PF_OPT_CONV_T16 = off) */
  int32_t no_pp;           /* != 0: never run the two-group ping-pong form of the 128-wide tile (PF_OPT_CONV_PP = off) */
  /* measurement aids (tools/sweep_conv.py): 0 = the library's own choice.  force_tile: 1 + tile index (1: 128 px x 128 ch, 2: 128 px x 64 ch,
   * 3: 64 px x 64 ch; bf16x3 3x3 stride 1 and planes GEMMs without GeGLU, tile 1 needs n % 128 == 0); force_ksplit: K slices across workgroups (bf16x3 3x3 stride 1,
   * must divide the 32-channel chunks; > 1 needs splitk_ws).  Results are the same up to summation order. */
  int32_t force_tile, force_ksplit;
  /* > 0: the SECOND sources (x1, skip_x1, gn_stats1) hold only x1_bmod samples; sample b reads sample b % x1_bmod of them (the shared skip
   * tensors of pf_unet_forward_cfg) */
  int32_t x1_bmod;
  /* Fused Winograd F(2x2, 3x3) form of the ResBlock conv (bf16x3, ks = 3, stride 1, prologue 1, hin / win multiples of 16, n a multiple
   * of 64, no fused skip projection): w_wino = the layer's weights transformed and packed by pf_pack_wino_weight_bf16x3; wino != 0 asks for it.
   * 2.25x fewer matrix-pipe operations; equal to the direct form up to rounding (the transforms amplify the split's operand rounding
   * ~1.5x, tools/micro/winograd_numerics.py).  A launch that does not qualify runs the direct form on `w`. */
  const void* w_wino; int32_t wino;
  /* optional range telemetry: a device word (fp32 bit pattern, start at 0) that receives max(word, largest |value| this launch stores)
   * by atomic max - see pf_unet_track_absmax */
  void* absmax_slot;
} pf_conv_args;
/* scratch bytes a launch with these arguments would like for split-K (0 = the launch does not split) */
size_t pf_conv_splitk_ws_bytes(const pf_conv_args* a);
/* number of per-sample tiles a pf_conv2d launch with these arguments emits into stats_out (0 on error) */
int pf_conv_stats_tiles(const pf_conv_args* a);
/* GroupNorm scale/shift from per-tile statistics of up to two channel-concatenated producers (no pass over the tensors) */
int pf_gn_finalize_tiles(const float* stats0, int tiles0, int c0, const float* stats1, int tiles1, int c1, int batch, int hw,
                         int groups, float eps, const float* gamma, const float* beta, float* scale, float* shift, void* stream);
int pf_conv2d(const pf_conv_args* a, void* stream);

/* softmax(q k^T * d_head^-0.5) v per (batch, head); q [B,Lq,*], k/v [B,Lk,*] with row strides ld*, heads packed
 * along the last dim (unet_attention.py:261-293). d_head in {32,64}. */
/* self-attention on the pre-split planes written by pf_conv2d(qkv_planes=...): d_head 64, L % 128 == 0 (bf16x3 split MFMA);
 * the result goes to fp32 `o` or, when o_planes != NULL, to bf16 hi/lo planes [M][C] | [M][C] for a following planes GEMM */
int pf_attention_bf16x3(const void* qkv_planes, float* o, int ldo, void* o_planes, int batch, int n_heads, int l, int form /* PF_OPT_AUTO | 0: 128-query | 1: 256-query workgroups */,
                        void* stream);
/* The same attention for SMALL batches (128-query form): when (l / 128) * n_heads * batch workgroups would leave three quarters of the CUs idle and
 * l % 512 == 0 (batch 1 / 2 at L = 1024), the keys of every query tile are split over four workgroups - each leaves its un-normalised partial result,
 * running maximum and row sum in `scratch` - and a second launch merges them (the online-softmax combination; equal to the one-launch form up to
 * summation order).  scratch_bytes >= pf_attention_split_scratch_bytes(batch, n_heads, l) (0 = this shape does not split: one launch, scratch unused). */
size_t pf_attention_split_scratch_bytes(int batch, int n_heads, int l);
int pf_attention_bf16x3_split(const void* qkv_planes, float* o, int ldo, void* o_planes, int batch, int n_heads, int l, void* scratch, size_t scratch_bytes,
                              void* stream);
int pf_attention(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo,
                 int batch, int n_heads, int d_head, int lq, int lk, void* stream);

/* Multi-GPU: one process per GPU.  The path shards over the batch with no exchange in the step loop; the single collective
 * (weight broadcast at start-up) goes either through the host's torch.distributed (backend "nccl" = RCCL over xGMI) on the packed
 * blob or through the library's own pf_comm_* entry points above (librccl opened with dlopen). */

#ifdef __cplusplus
}
#endif
#endif /* PFHIP_H */
