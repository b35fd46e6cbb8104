/*
 * crowdmod_hip.h -- C ABI of libcrowdmod_hip.so: the MI355X-native DDPM-UNet
 * denoiser + sampler hot path of marcemq/crowdmod-ddpm-4D.
 *
 * The reference has no FFI of its own: the seam is three Python call
 * signatures plus the checkpoint format.  Each entry
 * point below names the reference interface it replaces.  All tensors that
 * cross this ABI use the REFERENCE layout [B, C, H(rows), W(cols), L(frames)]
 * fp32 contiguous, L innermost (utils/dataset.py:48-51); the library permutes
 * to its internal channels-last layout on the device.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure;
 *     cm_last_error() returns a thread-local description of the last failure;
 *   - no exceptions cross the ABI; no torch / C++ types in signatures;
 *   - handles are opaque, created and destroyed by the caller;
 *   - the caller owns every buffer it passes in;
 *   - "d_" pointers are device pointers on the handle's device, "h_" host;
 *   - `stream` is a hipStream_t passed as void* (NULL = the handle's stream);
 *   - one handle is not thread-safe; distinct handles on distinct devices are.
 */
#ifndef CROWDMOD_HIP_H
#define CROWDMOD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CM_ABI_VERSION 3
#define CM_MAX_LEVELS 8

typedef struct cm_model cm_model;       /* UNet denoiser: weights + workspace   */
typedef struct cm_schedule cm_schedule; /* ForwardSampler / DDPM schedule tables */

/* Hyper-parameters of reference UNet.__init__ (models/backbones/unet.py:11-25)
 * plus the tensor geometry the reference takes from cfg.MACROPROPS / cfg.DATASET
 * (models/diffusion/ddpm.py:211). */
typedef struct cm_unet_config {
  int32_t in_channels;                    /* mprops_count: 3 or 4               */
  int32_t out_channels;
  int32_t num_res_blocks;                 /* NUM_RES_BLOCKS                     */
  int32_t base_channels;                  /* BASE_CH (multiple of 8)            */
  int32_t n_levels;                       /* len(BASE_CH_MULT)                  */
  int32_t channel_mult[CM_MAX_LEVELS];    /* BASE_CH_MULT                       */
  int32_t apply_attention[CM_MAX_LEVELS]; /* APPLY_ATTENTION (first n_levels)   */
  int32_t time_multiple;                  /* TIME_EMB_MULT                      */
  int32_t rows, cols;                     /* MACROPROPS.ROWS / COLS             */
  int32_t past_len, future_len;           /* DATASET.PAST_LEN / FUTURE_LEN      */
  int32_t max_batch;                      /* workspace is sized for this batch  */
  int32_t device;                         /* HIP device ordinal; < 0: host-only handle (state_dict plan:
                                             names, shapes, set / get) that can never be finalized      */
} cm_unet_config;

/* ---- errors / info ------------------------------------------------------ */
const char *cm_last_error(void);
int cm_abi_version(void);
int cm_device_count(int *count);

/* ---- raw device memory (for hosts without a tensor library) ------------- */
int cm_malloc(int device, void **d_ptr, size_t bytes);
int cm_free(int device, void *d_ptr);
int cm_memcpy_h2d(int device, void *d_dst, const void *h_src, size_t bytes);
int cm_memcpy_d2h(int device, void *h_dst, const void *d_src, size_t bytes);
int cm_memcpy_d2d(int device, void *d_dst, const void *d_src, size_t bytes);
int cm_device_synchronize(int device);

/* ---- denoiser: replaces UNet(...) / .load_state_dict / .state_dict ------ */
/* unet.py:11-122 (ctor). */
int cm_model_create(const cm_unet_config *cfg, cm_model **out);
int cm_model_destroy(cm_model *m);
/* state_dict() enumeration: names and shapes are the reference's (169 tensors
 * for config/ATC.yml; conv weights [Co,Ci,kH,kW,kL]). */
int cm_model_num_params(const cm_model *m, int32_t *count);
int cm_model_param_info(const cm_model *m, int32_t index, const char **name, int64_t shape[5],
                        int32_t *ndim);
/* load_state_dict(): ddpm.py:161,288.  `numel` must match the tensor's size. */
int cm_model_set_param(cm_model *m, const char *name, const float *h_data, int64_t numel);
int cm_model_get_param(const cm_model *m, const char *name, float *h_data, int64_t numel);
/* Matrix-core operand type of the inference plan, chosen before cm_model_finalize.  CM_PRECISION_F32 (default):
 * fp32 arithmetic everywhere -- fp32 tensors and accumulation; products on fp32 matrix instructions or, in the 3x3x3 layers that
 * carry most of the FLOPs, formed from exact-remainder splits of both operands on the 16-bit matrix instructions (f16 two-way
 * splits / three cross terms where the operand range is bounded, bf16 three-way splits / six cross terms otherwise and in the
 * training step): the measured error against the reference is that of an fp32 chain (rms 4e-7).
 * CM_PRECISION_F16: the Winograd 3x3x3 layers (every stride-1 3x3x3 conv with an even
 * in-plane grid: 71-98 % of the FLOPs) contract f16 operands with fp32 accumulation (v_mfma_f32_32x32x16_f16);
 * GroupNorm statistics, SiLU, residuals, attention and the sampler update stay fp32.  The reference's analogue is
 * torch.amp.autocast around the denoiser (models/diffusion/ddpm.py:116-120).  Tolerance: tests/test_gpu_f16.py.
 * CM_PRECISION_F32R ("relaxed fp32"): fp32 tensors and fp32 accumulation as in the default plan, but the layers whose fp32
 * products are formed from bf16 splits keep only the three leading cross terms of the six (hi x hi, hi x mid, mid x hi: two-way
 * splits, ~16 mantissa bits per product instead of 24) -- the analogue of PyTorch's default conv arithmetic on the reference's own
 * GPUs (torch.backends.cudnn.allow_tf32 = True: 10 mantissa bits), six bits finer.  Inference only; the forward stays inside the
 * 1e-4 max-abs bound against the reference (tests/test_gpu_relaxed.py), not inside the default plan's 2e-6.
 * CM_PRECISION_F32X ("strict fp32", round 3's arithmetic): the default plan with the f16 two-way-split form switched off -- every
 * split layer forms its products from exact three-way bf16 splits, six cross terms (all 24 mantissa bits of both operands, dropped
 * terms <= 2^-24 of a product, fp32's whole exponent range).  Same measured error as the default plan, 16 % slower; for callers
 * that want the stronger per-product statement, and the reference point of the A/B in DESIGN.md section 4. */
enum { CM_PRECISION_F32 = 0, CM_PRECISION_F16 = 1, CM_PRECISION_F32R = 2, CM_PRECISION_F32X = 3 };
int cm_model_set_precision(cm_model *m, int32_t precision);
/* Packs weights into MFMA fragment order and precomputes the time-embedding
 * tables; must be called after the last cm_model_set_param and before any
 * forward.  Fails if a tensor was never set. */
int cm_model_finalize(cm_model *m);

/* UNet.forward(future, t, past) in eval mode -- unet.py:124-167.
 *   d_future [B,C,H,W,F] f32, d_t [B] i64 in [0,1000), d_past [B,C,H,W,P] f32
 *   d_out    [B,C,H,W,F] f32 (eps_hat).  B <= max_batch. */
int cm_unet_forward(cm_model *m, const float *d_future, const int64_t *d_t, const float *d_past,
                    float *d_out, int32_t B, void *stream);
/* Same with host buffers (staged through the workspace; synchronous). */
int cm_unet_forward_host(cm_model *m, const float *h_future, const int64_t *h_t, const float *h_past,
                         float *h_out, int32_t B);
/* Training-mode forward -- unet.py:124-167 with nn.Dropout3d active (layers.py:42,71): one
 * keep-mask/(1-p) value per (sample, ResnetBlock, channel).  d_dropmask [B][width] injects the
 * masks (width from cm_model_dropout_width: the blocks in state_dict order, Cout entries each);
 * NULL draws them from the device Philox stream (seed, sample_id_base + b).  Used by
 * DDPM_model._train_step (ddpm.py:111-121) when only the prediction is wanted; the full step
 * (loss, backward, Adam) is cm_train_step below. */
int cm_model_dropout_width(const cm_model *m, int32_t *width);
int cm_unet_forward_train(cm_model *m, const float *d_future, const int64_t *d_t, const float *d_past,
                          const float *d_dropmask, float p, uint64_t seed, int64_t sample_id_base,
                          float *d_out, int32_t B, void *stream);
/* F.mse_loss(pred, target) with reduction='mean' (ddpm.py:120); the scalar lands in *h_loss. */
int cm_mse_loss(cm_model *m, const float *d_pred, const float *d_target, int64_t n, float *h_loss,
                void *stream);
/* Test hook: copy an internal activation (by reference module name, e.g.
 * "encoder_blocks.0") of the last forward to the host in reference layout
 * [B,C,H,W,L].  `shape` receives {B,C,H,W,L}. */
int cm_debug_activation(cm_model *m, const char *name, float *h_out, int64_t capacity,
                        int64_t shape[5]);

/* ---- schedule: replaces ForwardSampler.__init__ (forward.py:10-27) ------ */
int cm_schedule_create(int32_t timesteps, float scale, float beta_start, float beta_end,
                       int32_t device, cm_schedule **out);
int cm_schedule_destroy(cm_schedule *s);
enum {
  CM_TAB_BETA = 0,
  CM_TAB_ALPHA = 1,
  CM_TAB_ALPHA_BAR = 2,
  CM_TAB_SQRT_ALPHA_BAR = 3,
  CM_TAB_ONE_BY_SQRT_ALPHA = 4,
  CM_TAB_SQRT_ONE_MINUS_ALPHA_BAR = 5
};
/* The six [T] buffers the reference exposes as attributes (ddpm.py:246-248). */
int cm_schedule_table(const cm_schedule *s, int32_t which, float *h_out, int32_t capacity);

/* ---- sampler steps ------------------------------------------------------- */
/* ForwardSampler.forward (forward.py:29-36) with the noise supplied by the
 * caller: d_xt = sqrt_alpha_bar[t_b] * x0 + sqrt_one_minus_alpha_bar[t_b] * eps.
 * per_sample = C*H*W*F. */
int cm_q_sample(const cm_schedule *s, const float *d_x0, const int64_t *d_t, const float *d_eps,
                float *d_xt, int32_t B, int64_t per_sample, void *stream);
/* DDPM.step (ddpm.py:25-38): d_x is updated in place.  d_noise == NULL draws
 * z from the device Philox stream (seed, sample_id_base + b, step t); it is
 * ignored (z = 0) when t == 0, as in the reference. */
int cm_ddpm_step(const cm_schedule *s, const float *d_eps, float *d_x, int32_t t, const float *d_noise,
                 uint64_t seed, int64_t sample_id_base, int32_t B, int64_t per_sample, void *stream);

/* ---- whole reverse loop --------------------------------------------------- */
enum { CM_SAMPLER_DDPM = 0, CM_SAMPLER_DDIM = 1, CM_SAMPLER_FM_EULER = 2 };
enum { CM_GUIDANCE_NONE = 0, CM_GUIDANCE_SPARSITY = 1 };

typedef struct cm_sample_opts {
  int32_t sampler;        /* cfg.MODEL.DDPM.SAMPLER                              */
  int32_t guidance;       /* cfg.MODEL.DDPM.GUIDANCE ("None" | "Sparsity")       */
  float lambda_guidance;  /* cfg.MODEL.DDPM.LAMBDA_GUIDANCE                      */
  float ddim_sigma;       /* cfg.MODEL.DDPM.SIGMA                                */
  int32_t ddim_divider;   /* cfg.MODEL.DDPM.DDIM_DIVIDER: taus = arange(0,T-1,d) */
  int32_t first_steps;    /* >0: run only the first n visited steps (benchmarks) */
  uint64_t seed;          /* device RNG seed (used when x_T / noise are NULL)    */
  int64_t sample_id_base; /* global index of sample 0 (batch sharding)           */
  int32_t use_graph;      /* 1: capture one step as a hipGraph (per-step scalars in a device table) and replay
                             it for the remaining steps; the call then returns after the loop has finished   */
  int32_t check_finite;   /* 1: after the last step count the non-finite elements of x_0; the call then returns
                             after the loop has finished, with a non-zero status (and the count in
                             cm_last_error) if any element is NaN / Inf -- the sampler-output health check
                             (the reference's analogue is the NaN stop of its training loop, ddpm.py:183-192) */
  /* CM_SAMPLER_FM_EULER -- FM_model.sampling_with_euler (models/flow_matching/flow_matching.py:203-224):
   * x <- x + (1/N) u(x, idx_i, past) for t_i = linspace(0,1,N)[i], idx_i = clamp(t_i * TIME_MAX_POS, 0,
   * TIME_MAX_POS-1) truncated; the schedule handle is not consulted. */
  int32_t fm_steps;        /* cfg.MODEL.FM.INTEGRATOR_STEPS.EULER                 */
  int32_t fm_time_max_pos; /* cfg.MODEL.FM.TIME_MAX_POS (<= 1000 table rows)      */
} cm_sample_opts;

/* DDPM_model._generate_ddpm / _generate_ddim (ddpm.py:206-282): all T (or
 * len(taus)) steps run on the device behind this one call.
 *   d_past   [B,C,H,W,P]
 *   d_xT     [B,C,H,W,F] or NULL (draw from the device RNG)
 *   d_noise  [nsteps,B,C,H,W,F] in visiting order, or NULL (device RNG).
 *            DDPM: step k is t = T-1-k, rows for t = 0 are not read
 *            (nsteps = T-1 suffices); DDIM: step k is reversed(taus)[k].
 *   d_out    [B,C,H,W,F] final sample x_0
 *   d_history NULL, or [nsteps+1,B,C,H,W,F]: x_T followed by x after every step
 *            (the `history=True` list of the reference). */
int cm_sample_loop(cm_model *m, const cm_schedule *s, const float *d_past, const float *d_xT,
                   const float *d_noise, const cm_sample_opts *opts, float *d_out, float *d_history,
                   int32_t B, void *stream);
/* Host-buffer variant (synchronous). */
int cm_sample_loop_host(cm_model *m, const cm_schedule *s, const float *h_past, const float *h_xT,
                        const float *h_noise, const cm_sample_opts *opts, float *h_out,
                        float *h_history, int32_t B);
/* Number of UNet evaluations cm_sample_loop performs for these options. */
int cm_sample_num_steps(const cm_schedule *s, const cm_sample_opts *opts, int32_t *nsteps);

/* ---- instrumentation ------------------------------------------------------ */
/* Per-kernel-class device time of the LAST forward/loop, measured with HIP
 * events on the launch stream when enabled (adds synchronisation; off by
 * default).  classes: 0 conv3x3x3, 1 conv1x1x1/GEMM, 2 stats+GN, 3 attention,
 * 4 elementwise (assemble/step).  `ms` and `launches` hold 8 entries. */
int cm_profile_enable(cm_model *m, int32_t on);
int cm_profile_read(cm_model *m, float ms[8], int64_t launches[8]);
/* Per class: length (ms) of the UNION of its launch intervals over all batch lanes of the last profiled call -- with
   two lanes the launches of a class overlap, so the sum of their durations in cm_profile_read exceeds the wall time
   they occupied; with one lane the two agree. */
int cm_profile_read_union(cm_model *m, float ms[8]);
/* Text table of the per-launch averages behind cm_profile_read (one line per op). */
int cm_profile_report(cm_model *m, char *buf, int64_t capacity);
/* Algorithmic FLOPs and bytes of one forward at batch B. */
int cm_model_cost(const cm_model *m, int32_t B, double *flops, double *bytes);
/* Algorithmic FLOPs of one forward per kernel class (same indices as cm_profile_read). */
int cm_model_class_flops(const cm_model *m, int32_t B, double flops[8]);
/* Matrix-core FLOPs the plan EXECUTES per class (parity-form upsample convs: 8 of 27 taps; Winograd F(2x2,3x3)
 * layers: 16 multiplies per 2x2 outputs and z tap instead of 36) -- the hardware-utilisation side of the roofline. */
int cm_model_exec_flops(const cm_model *m, int32_t B, double flops[8]);
/* The same FLOPs by the matrix instruction that issues them: `f32` = issued as fp32 instructions (v_mfma_f32_32x32x2_f32),
 * `b16` = ISSUED FLOPs of 16-bit-operand instructions (v_mfma_f32_32x32x16_{bf16,f16}): a six-term layer -- fp32 products
 * from exact three-way bf16 splits -- issues six of them per fp32-equivalent product, an f16-plan layer one.  The time the
 * matrix pipe needs at its peaks is f32 / peak_fp32 + b16 / peak_16bit: the numerator of bench.py's roofline fraction. */
int cm_model_issue_flops(const cm_model *m, int32_t B, double f32[8], double b16[8]);

/* ---- sampling metrics: the per-frame reductions of utils/metrics/metricsGenerator.py:70-92,
 * 120-186,293-339 (PSNR, masked PSNR, relative density error, total variation) on the device.
 *   d_pred, d_gt [N,C,H,W,F] f32 (device);  h_out [N][C][F][8] f64: sse, masked sse, masked count
 *   (mask = gt[:,0] > 1e-5), tv_pred, tv_gt, sum_pred, sum_gt, 0;  h_minmax [N][C][F][2] f32: min / max
 *   of the gt plane.  The log10 / max-over-repeats tail on [N,C,F] numbers stays on the host. */
int cm_frame_metrics(int32_t device, const float *d_pred, const float *d_gt, int32_t N, int32_t C, int32_t H,
                     int32_t W, int32_t F, double *h_out, float *h_minmax);

/* ---- tuning hooks (tools/tune_tiles.py): time one convolution of the plan with an explicit tile
 * geometry.  Diagnostics only -- the product path never calls them. */
/* Process-wide diagnostic switches of the conv kernels (the CM_CONV_DBG bit mask of DESIGN.md; 4096 = XCD tile
 * remap off); flags < 0 returns to the environment value.  Results are only defined for 0 / 4096 / 2048 / 512. */
int cm_debug_conv_flags(int32_t flags);
int cm_debug_conv_count(const cm_model *m, int32_t *count);
int cm_debug_conv_info(const cm_model *m, int32_t index, char *buf, int64_t capacity);
/* Test hook: conv op `index` alone on caller data (no GroupNorm / SiLU / time row / residual / fused skip; bias stays).
 * h_in0 / h_in1: host channels-last [B][Zs][Ys][Xs][C0 / C1]; h_out: host [B][Zo][Yo][Xo][C of the output tensor -- the last
 * field of cm_debug_conv_info];
 * mode 0 = the six-term bf16 form where the plan has one (raw operands are unbounded: never the h2 form), 1 = the same layer on
 * fp32 matrix instructions (split fragments withheld), 2 = the h2 form (f16 two-way splits) where the plan has one: the caller
 * keeps |x| inside the bound the plan guarantees for that layer (8000; tests/test_gpu_h2.py). */
int cm_debug_conv_io(cm_model *m, int32_t index, int32_t mode, const float *h_in0, const float *h_in1, float *h_out, int32_t B);
int cm_debug_time_conv(cm_model *m, int32_t index, int32_t MB, int32_t bz, int32_t by, int32_t bx,
                       int32_t B, int32_t iters, float *us);

/* ---- training step: replaces DDPM_model._train_step + optimizer.step() -----
 * (models/diffusion/ddpm.py:111-121,142-144; optimizer built at ddpm.py:53-56:
 * torch.optim.Adam(lr, betas, weight_decay) -- L2 coupled into the gradient.)
 * cm_train_init allocates master weights / gradients / Adam moments on the device
 * (state_dict order) and the backward workspace; call once after cm_model_finalize. */
int cm_train_init(cm_model *m, float lr, float beta1, float beta2, float eps, float weight_decay,
                  float dropout_rate /* cfg DROPOUT_RATE: nn.Dropout3d(p), layers.py:42 */);
int cm_train_set_lr(cm_model *m, float lr);
/* Data-parallel training: global index of this rank's sample 0 (rank * per-rank batch).  The device
 * Philox streams of eps and of the Dropout3d masks are addressed by (seed, step, GLOBAL sample index), so
 * ranks draw different noise and a job's samples draw the same numbers however it is sharded.  Default 0. */
int cm_train_set_sample_base(cm_model *m, int64_t sample_id_base);
/* One step on device buffers:
 *   x_t = q_sample(d_future, d_t, d_eps)          (forward.py:29-35 with the caller's noise)
 *   eps_hat = UNet(x_t, d_t, d_past) in train mode (Dropout3d masks: d_dropmask [B][width] or
 *             NULL -> device Philox stream keyed by (seed, step))
 *   *h_loss = mse(eps_hat, d_eps); backward; if apply_update != 0: Adam step + weight re-pack.
 * Synchronous (returns after the step has finished). */
int cm_train_step(cm_model *m, const cm_schedule *s, const float *d_future, const float *d_past,
                  const int64_t *d_t, const float *d_eps, const float *d_dropmask, uint64_t seed,
                  float *h_loss, int32_t B, int32_t apply_update, void *stream);
/* d_eps may be NULL: eps ~ N(0,1) is then drawn from the device Philox stream (seed, step, sample),
 * the analogue of torch.randn_like in forward.py:33.
 * Data-parallel use: cm_train_step(apply_update=0) on every rank, all-reduce(mean) the flat gradient
 * buffer from cm_train_flat_grads, then cm_train_apply (Adam + re-pack). */
int cm_train_flat_grads(cm_model *m, void **d_grads, int64_t *numel);
int cm_train_apply(cm_model *m, void *stream);
/* Adam state for the "opt" entry of a checkpoint (utils/utils.py:140-147): which = 0 exp_avg,
 * 1 exp_avg_sq; cm_train_opt_step reads (set=0) or writes (set=1) the step counter. */
int cm_train_get_opt_state(cm_model *m, const char *name, int32_t which, float *h_out, int64_t numel);
int cm_train_set_opt_state(cm_model *m, const char *name, int32_t which, const float *h_in, int64_t numel);
int cm_train_opt_step(cm_model *m, int32_t *step, int32_t set);
/* The same step with the caller's noised input and regression target (flow matching,
 * flow_matching.py:128-146: x_t and u_target from w_linear / w_conic, d_t = (t * TIME_MAX_POS).long()):
 * eps_hat = UNet(d_xt, d_t, d_past) in train mode; *h_loss = mse(eps_hat, d_target); backward; update. */
int cm_train_step_xt(cm_model *m, const float *d_xt, const float *d_past, const int64_t *d_t,
                     const float *d_target, const float *d_dropmask, uint64_t seed, float *h_loss,
                     int32_t B, int32_t apply_update, void *stream);
/* Gradient of one state_dict tensor after the last cm_train_step (reference layout). */
int cm_train_get_grad(cm_model *m, const char *name, float *h_out, int64_t numel);
/* Copy the trained master weights back into the handle's state_dict (cm_model_get_param then
 * returns them) and refresh the eval-mode time-embedding table. */
int cm_train_sync(cm_model *m);

#ifdef __cplusplus
}
#endif
#endif /* CROWDMOD_HIP_H */
