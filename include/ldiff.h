/* ldiff.h -- C ABI of libldiff_hip.so: the MI355X (gfx950) Laplace-diffusion sampling path.
 *
 * The reference (Lweihan/LDiffusion, /root/reference) has no FFI of its own: its sampler loops call
 * duck-typed python objects from `diffusers`.  Each entry point below names the
 * reference call it replaces (file:line into /root/reference); the python shims under ldiffusion_amd/ bind them with
 * ctypes behind shim objects that keep the reference's attribute surface, and INTEGRATION.md shows
 * the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 (LDIFF_OK) or a negative ldiff_status; the message of the last failure
 *     on the calling thread is ldiff_last_error().  -1 = bad argument/shape (python: ValueError),
 *     -2 = HIP runtime failure, -3 = handle not ready (weights/context missing) (python: RuntimeError).
 *   - tensors crossing the boundary are plain device pointers in the layouts the reference uses:
 *     float32, NCHW, contiguous (torch_dtype=torch.float32 everywhere: ldiffusion.py:67, segmentor.py:77).
 *     Internally activations are NHWC fp16 with fp32 accumulation; the residual stream is kept as fp16 hi|lo pairs
 *     (ldiff_unet_set_precision).
 *   - handles own device weights and workspace; caller-owned buffers are never retained past a call.
 *   - all work is enqueued on the caller's HIP stream (`stream` = hipStream_t, e.g. torch's current
 *     stream); no hidden synchronisation.  A handle is thread-compatible, not thread-safe; handles on different
 *     devices are independent (no process-wide device state in the library).
 *   - there is no CPU fallback anywhere: a missing GPU or code object is an error.
 *   - Non-finite detection.  Activations are stored as fp16 (also under precision 1 / 2: the hi half of a split tensor is an fp16), so a checkpoint /
 *     input whose activations leave +-65504 overflows where the reference's fp32 graph does not -- e.g. the SD-v1.x decoder fed z / 0.18215 of UN-scaled
 *     latents (pixel_latent_vector.py:73,81).  Every graph therefore carries a sticky flag, set on the device by the GroupNorm-statistics kernels when a
 *     group's totals are not finite (every activation reaches one: an inf from a conv / GEMM epilogue, or the NaN it becomes downstream; no extra launch,
 *     no extra read).  It is reported as LDIFF_ERR_NONFINITE (-4; python: RuntimeError subclass NonFiniteError)
 *       (a) by ldiff_{unet,vae,pipeline}_check_finite(handle, stream): synchronises `stream` (and the decode side stream), returns the verdict
 *           for everything enqueued so far and clears the flag -- the python shims call it wherever they hand results to the host;
 *       (b) at the latest by the NEXT ldiff_unet_forward / ldiff_vae_encode / ldiff_vae_decode / ldiff_sample on the handle, at entry, for work that has
 *           completed by then (no synchronisation; the flag is cleared when reported).
 *     The results of a flagged call are garbage.  LDIFF_TRACE_ABSMAX=1 prints max |activation| per graph stage to stderr (diagnostic; synchronises).
 */
#ifndef LDIFF_H
#define LDIFF_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LDIFF_VERSION 160 /* 0.1.6.0: + non-finite detection (LDIFF_ERR_NONFINITE, ldiff_*_check_finite), ldiff_conv_args.splitk (split launches emit statistics); 0.1.5.2: + ldiff_conv_args.n_real (tap-folded conv_out kernel), c3d_ups (upsampling convs on the dataflow kernel); 0.1.5.1: dataflow GEMM (gemm_df), shortcut conv folded into the dataflow conv3x3 (sc_*) */
#define LDIFF_MAX_BLOCKS 8

typedef enum { LDIFF_OK = 0, LDIFF_ERR_INVALID = -1, LDIFF_ERR_RUNTIME = -2, LDIFF_ERR_STATE = -3, LDIFF_ERR_NONFINITE = -4 } ldiff_status;
typedef enum { LDIFF_F32 = 0, LDIFF_F16 = 1, LDIFF_BF16 = 2 } ldiff_dtype; /* host dtypes accepted by *_load */

int ldiff_version(void);
const char* ldiff_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * UNet2DConditionModel  --  replaces `unet(latents, t, text_embeddings)`
 *   segmentor.py:103,444,526   pixel_latent_vector.py:78   ldiffusion.py:160,238   utils.py:201
 * cfg mirrors diffusers' unet/config.json.
 * ---------------------------------------------------------------------------------------------- */
typedef struct ldiff_unet ldiff_unet;
typedef struct {
  int in_channels, out_channels;
  int n_blocks;
  int block_out_channels[LDIFF_MAX_BLOCKS];
  int down_has_attn[LDIFF_MAX_BLOCKS]; /* CrossAttnDownBlock2D = 1, DownBlock2D = 0 */
  int up_has_attn[LDIFF_MAX_BLOCKS];   /* CrossAttnUpBlock2D = 1, UpBlock2D = 0 */
  int layers_per_block;
  int heads;               /* config.json "attention_head_dim" (SD-v1.5: it is the head COUNT) */
  int cross_attention_dim;
  int norm_num_groups;
  float norm_eps;
  int flip_sin_to_cos;
  float freq_shift;
} ldiff_unet_cfg;

int ldiff_unet_create(ldiff_unet** out, const ldiff_unet_cfg* cfg, int device);
/* Copy one tensor of diffusion_pytorch_model.safetensors (diffusers key names, torch layouts) to the device.
 * (from_pretrained: segmentor.py:79, ldiffusion.py:67) */
int ldiff_unet_load(ldiff_unet*, const char* name, const void* host_ptr, int dtype, const int64_t* shape, int ndim);
/* Storage policy of the graph (the reference computes in fp32: ldiffusion.py:67; every MFMA operand here is fp16):
 *   0 = all activations fp16 in HBM (fastest; ~2e-3 of the output range per UNet pass)
 *   1 = residual stream kept as fp16 hi|lo pairs (adds to fp32 round-off), stream-carrying contractions on split operands (default)
 *   2 = every conv / linear operand split (K doubled): ~1e-4 */
int ldiff_unet_set_precision(ldiff_unet*, int mode);
/* One forward is ~390-450 kernel launches (384 at B = 8, 443 at B = 1 at SD-v1.5 size: ldiff_unet_graph_nodes).  With graphs on (default) the launch sequence of a (B, h, w, precision, context)
 * configuration is captured into a hipGraph on its second use and replayed from then on (input, timestep and output pass through
 * handle-owned staging buffers: any caller pointers, any timestep; bit-identical results).  Off: every forward is launched
 * eagerly.  Forwards issued while per-launch profiling is enabled, or on a stream that is itself being captured, run eagerly. */
int ldiff_unet_set_graph(ldiff_unet*, int on);
int64_t ldiff_unet_graph_replays(ldiff_unet*); /* forwards served by graph replay so far */
int64_t ldiff_unet_graph_nodes(ldiff_unet*);   /* kernel launches of the currently captured forward (0: none captured yet) */
/* number of expected tensors not loaded yet; names via ldiff_unet_missing_name(i) */
int ldiff_unet_missing(ldiff_unet*);
const char* ldiff_unet_missing_name(ldiff_unet*, int i);
/* encoder_hidden_states [B_ctx, L, cross_attention_dim] float32 on device; precomputes the cross-attention
 * K/V of all transformer blocks (they do not depend on the timestep).  B_ctx is 1 (broadcast) or the batch. */
int ldiff_unet_set_context(ldiff_unet*, const void* ctx_dev, int B_ctx, int L, void* stream);
/* sample [B,in_channels,h,w] f32 NCHW -> out [B,out_channels,h,w] f32 NCHW */
int ldiff_unet_forward(ldiff_unet*, const void* sample_dev, int B, int h, int w, float timestep, void* out_dev, void* stream);
/* ControlNet inputs of the NEXT ldiff_unet_forward (diffusers' down_block_additional_residuals / mid_block_additional_residual,
 * segmentor.py:357-375): n_down float32 NCHW device tensors in skip-stack order (conv_in output first; shapes of the skip tensors),
 * added to the skip connections, and one tensor added to the mid block's output (either may be absent: n_down = 0 / NULL).  The
 * pointers must stay valid until that forward has been enqueued; they are consumed by it. */
int ldiff_unet_set_additional_residuals(ldiff_unet*, const void* const* down_dev, int n_down, const void* mid_dev);
/* non-finite detector (see the conventions above): LDIFF_OK or LDIFF_ERR_NONFINITE for all forwards enqueued on `stream` so far; clears the flag */
int ldiff_unet_check_finite(ldiff_unet*, void* stream);
void ldiff_unet_destroy(ldiff_unet*);

/* ------------------------------------------------------------------------------------------------
 * AutoencoderKL  --  replaces vae.encode(x).latent_dist / vae.decode(z).sample / pipeline.decode_latents
 *   segmentor.py:99,106,339,379,437,447,519,529  pixel_latent_vector.py:73,81  ldiffusion.py:228,240  utils.py:190,204
 * ---------------------------------------------------------------------------------------------- */
typedef struct ldiff_vae ldiff_vae;
typedef struct {
  int in_channels, out_channels, latent_channels;
  int n_blocks;
  int block_out_channels[LDIFF_MAX_BLOCKS];
  int layers_per_block;
  int norm_num_groups;
  float scaling_factor;
} ldiff_vae_cfg;

int ldiff_vae_create(ldiff_vae** out, const ldiff_vae_cfg* cfg, int device);
int ldiff_vae_load(ldiff_vae*, const char* name, const void* host_ptr, int dtype, const int64_t* shape, int ndim);
/* storage policy of the encoder and of the decoder graph (see ldiff_unet_set_precision); defaults: encoder 2 (its error is
 * inherited by every later pass of the sampler and it runs once per patch), decoder 0 (its output is only consumed as uint8
 * images / luma, never fed back into the latents: measured at SD-v1.5 width, 512x512, 5 passes, the luma features are within one
 * grey level of the fp32 oracle in all three modes -- 6.6 % / 3.5 % / 2.9 % of the pixels differ by one -- and the probe-head
 * masks are identical in all three; modes 1 / 2 cost +20 % / +96 % decode time) */
int ldiff_vae_set_precision(ldiff_vae*, int encoder_mode, int decoder_mode);
int ldiff_vae_missing(ldiff_vae*);
const char* ldiff_vae_missing_name(ldiff_vae*, int i);
/* x [B,3,H,W] f32 NCHW -> moments [B, 2*latent, H/8, W/8] f32 NCHW (mean | logvar), i.e. quant_conv(encoder(x)) */
int ldiff_vae_encode(ldiff_vae*, const void* x_dev, int B, int H, int W, void* moments_dev, void* stream);
/* z [B,latent,h,w] f32 NCHW, multiplied by z_scale first (decode_latents passes 1/scaling_factor, vae.decode passes 1).
 * Any of the outputs may be NULL:
 *   sample_nchw [B,3,8h,8w] f32            = vae.decode(z).sample
 *   image_nhwc  [B,8h,8w,3] f32            = (sample/2+0.5).clamp(0,1)            (decode_latents)
 *   rgb_u8      [B,8h,8w,3] u8             = (image*255).round()  half-even       (numpy_to_pil)
 *   luma_u8     [B,n_slots,8h,8w] u8, slot = PIL convert("L") of rgb_u8           (pixel_latent_vector.py:85) */
int ldiff_vae_decode(ldiff_vae*, const void* z_dev, int B, int h, int w, float z_scale, void* sample_nchw, void* image_nhwc,
                     void* rgb_u8, void* luma_u8, int n_slots, int slot, void* stream);
/* non-finite detector: LDIFF_OK or LDIFF_ERR_NONFINITE for all encodes / decodes enqueued so far (on `stream` and on the decode side stream) */
int ldiff_vae_check_finite(ldiff_vae*, void* stream);
void ldiff_vae_destroy(ldiff_vae*);

/* ------------------------------------------------------------------------------------------------
 * Sampler arithmetic
 * ---------------------------------------------------------------------------------------------- */
/* PNDM/PLMS update as one linear combination out = sum_i coef[i] * ops[i] over n float32 elements
 * (scheduler.step(...).prev_sample: segmentor.py:104,445,527  pixel_latent_vector.py:79); coefficients
 * are computed on the host from alphas_cumprod exactly as PNDMScheduler._get_prev_sample does. */
int ldiff_pndm_step(const float* coef, const void* const* ops, int nops, void* out, int64_t n, void* stream);
/* (host) the two float32 coefficients of _get_prev_sample for alphas_cumprod values a_t, a_prev:
 * prev_sample = sample_coeff * sample + eps_coeff * model_output.  Both the python scheduler shim and
 * ldiff_sample use this one routine, so the two drivers agree bit for bit on every machine. */
int ldiff_pndm_coeffs(float a_t, float a_prev, float* sample_coeff, float* eps_coeff);
/* alphas_cumprod table of the SD-v1.5 scheduler config (1000 float32) */
int ldiff_pndm_alphas_cumprod(float* out_host, int n);
/* out = z0 + Laplace(0, scale) given the uniform draw u (or u = NULL: counter-based Philox stream seed/offset)
 * (ldiffusion.py:234-237; torch.distributions.Laplace.rsample) */
int ldiff_laplace_add(const void* z0, float scale, const void* u_or_null, uint64_t seed, uint64_t offset, void* out, int64_t n, void* stream);
/* logits [B,C,H,W] f32 -> mask [B,H,W] u8 = argmax over C  (segmentor.py:536-537) */
int ldiff_argmax_u8(const void* logits, int B, int C, int H, int W, void* mask_u8, void* stream);
/* Mask tail over the per-pixel latent vectors in ONE launch: features [B,N,H,W] u8 (the N luma planes of a pixel are its latent vector,
 * pixel_latent_vector.py:85-93), weight [C,N] f32, bias [C] f32 or NULL -> mask [B,H,W] u8 = first argmax_c of
 *   logit_c = bias_c + sum_n weight[c,n] * (feature_n * scale)      (argmax(softmax(.)): segmentor.py:536-537)
 * with the arithmetic pinned (float32, x_n = fl(f_n * scale), acc = bias_c, acc = fl(acc + fl(w * x_n)) in plane order, no fused multiply-add)
 * so that a host statement reproduces the mask bit for bit (oracle/noise_post.py probe_argmax).  C <= 32, N <= 64, H*W % 4 == 0. */
int ldiff_probe_argmax_u8(const void* features_u8, int B, int N, int H, int W, const void* weight, const void* bias_or_null, float scale, int C, void* mask_u8,
                          void* stream);
/* One tile of a sliding-window / tile merge: acc[:, y0:y0+th, x0:x0+tw] += pred * weight;  cnt[y0:.., x0:..] += weight (weight NULL: 1)
 * acc [C,H,W], cnt [H,W], pred [C,th,tw], weight [th,tw]; dtypes 0: all float32, 1: all float16, 3: float16 accumulators and weight with a
 * float32 prediction (the product then stays float32, as type promotion makes it); product and sum rounded separately, as the tensor
 * formulation does (nnU-Net predictor, model/nnunetv2/inference/predict_from_raw_data.py:563-570,
 * called from segmentor.py:388-488; float16 accumulators there). */
int ldiff_window_accumulate(void* acc, void* cnt, const void* pred, const void* weight_or_null, int C, int H, int W, int th, int tw, int y0, int x0, int dtypes,
                            void* stream);
/* rgb [B,3,H,W] f32 -> gray [B,1,H,W] f32 = (rgb*[0.2989,0.5870,0.1140]).sum(1)  (ldiffusion.py:241-242) */
int ldiff_luma_float(const void* rgb_nchw, void* gray, int B, int H, int W, void* stream);
/* F.interpolate(x, size=(out_h,out_w), mode="bilinear", align_corners=False) on fp32 NCHW (ldiffusion.py:240,250: the decoded
 * image is resized to 64x64 before the float luma of the training-time features). */
int ldiff_bilinear_resize(const void* x_nchw_f32, void* y_nchw_f32, int B, int C, int H, int W, int out_h, int out_w, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused sampler  --  replaces the whole per-image loop body of
 *   pixel_latent_vector.py:72-86 (mode LDIFF_SAMPLE_PLMS)  and  segmentor.py:99-107 / 519-530 (same, n_passes = 1)
 * for a batch of B patches:  z = vae.encode(x).mean;  set_timesteps;  for t: eps = unet(z,t,ctx);
 * z = scheduler.step(eps,t,z);  rgb = numpy_to_pil(decode_latents(z));  luma[:, pass] = convert("L").
 * ---------------------------------------------------------------------------------------------- */
typedef struct ldiff_pipeline ldiff_pipeline;
int ldiff_pipeline_create(ldiff_pipeline** out, ldiff_unet*, ldiff_vae*); /* borrows both handles */
/* Replace the built-in alphas_cumprod table (1000 float32, host) with the caller's scheduler.alphas_cumprod
 * (ldiffusion.py:198,234 reads that attribute); the built-in one agrees with torch's to ~2 ulp. */
int ldiff_pipeline_set_alphas_cumprod(ldiff_pipeline*, const float* abar_host, int n);
/* ldiff_sample runs the VAE decode of pass k (it only feeds the feature tensor) on the VAE's side stream beside the UNet
 * pass k+1 (+7 % patches/s: the UNet's 16x16 / 8x8 levels leave most CUs idle).  mode 0: everything on the caller's stream;
 * mode 1 (default): side stream, the caller's stream joins before ldiff_sample returns; mode 2: side stream, join deferred:
 * features / rgb of that call may only be read after ldiff_pipeline_join(p, stream) -- with two pipelines on the same
 * unet/vae used alternately, the encoder and UNet passes of batch k+1 then run under the trailing decodes of batch k.
 * Results are bit-identical in all modes. */
int ldiff_pipeline_set_overlap(ldiff_pipeline*, int mode);
int ldiff_pipeline_join(ldiff_pipeline*, void* stream);
/* images [B,3,H,W] f32.  n_passes = number of UNet passes N (set_timesteps(N-1) for N >= 3, set_timesteps(1) for N = 1;
 * N = 2 is rejected: the reference's set_timesteps(1) then yields a single pass, pixel_latent_vector.py:74).
 * Outputs (any may be NULL): latents_out [B,latent,H/8,W/8] f32 (after the last pass),
 * features_u8 [B,N,H,W] (luma of every pass), rgb_u8 [B,H,W,3] (last pass). */
int ldiff_sample(ldiff_pipeline*, const void* images, int B, int H, int W, int n_passes, void* latents_out, void* features_u8,
                 void* rgb_u8, void* stream);
/* non-finite detector over both graphs of the pipeline: joins a deferred decode, synchronises `stream`, LDIFF_OK or LDIFF_ERR_NONFINITE; clears the flags */
int ldiff_pipeline_check_finite(ldiff_pipeline*, void* stream);
/* PLMS timesteps the sampler will visit for n_passes (host); returns the count written. */
int ldiff_plms_timesteps(int n_passes, int64_t* out, int cap);
void ldiff_pipeline_destroy(ldiff_pipeline*);

/* ------------------------------------------------------------------------------------------------
 * Single-kernel entry points (internal NHWC fp16 layout) -- used by the parity tests and the roofline
 * bench so that every kernel is reachable through the C ABI.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const void* x;  const void* x2; int C1, C2;       /* NHWC f16 sources (x2 optional channel concat) */
  int B, Hin, Win, Hout, Wout, ks, stride, pad_t, pad_l, ups;
  const void* w;  int N, Nrows;                     /* [Nrows][ks*ks*(C1+C2)] f16, rows >= N zero */
  const void* gn_scale; const void* gn_shift; int silu_in;   /* f32 [B, C1+C2] or NULL */
  const void* bias;                                 /* f32 [Nrows] or NULL */
  const void* temb; int ld_temb;                    /* f32 [B, ld_temb] or NULL */
  const void* res; int ld_res;                      /* f16 [M, ld_res] or NULL */
  void* y; int ldy; int out_f32;
  void* stats;                                      /* optional f32 [B][N][R][2]: fused GroupNorm partial sums of y (R from ldiff_op_conv_stats_blocks) */
  int geglu;                                        /* 1x1 / linear only: the weight rows are the [x | gate] rows of diffusers' GEGLU projection
                                                       interleaved by 16 (row r of x -> 32*(r/16) + r%16, of gate -> 32*(r/16) + 16 + r%16);
                                                       y[m, 0..N/2) = x * gelu_erf(gate), ldy counts those N/2 columns (N % 32 == 0) */
  int ld1, ld2;                                     /* row pitch (elements) of x / x2; 0 = C1 / C2 */
  int res_lo;                                       /* > 0: res is a split tensor (value = hi + lo), lo half res_lo elements after the hi half */
  int y_lo;                                         /* > 0: write y split: hi at column n, lo = f16(v - hi) at column y_lo + n */
  int short_runs;                                   /* 1: the persistent conv kernels retire a workgroup after ONE unit / tile (what ldiff_sample sets for the VAE
                                                       decodes that run beside the next UNet pass); 0: one workgroup per CU walks its whole share */
  int lo8_slab0;                                    /* > 0: x is a split operand whose lo half is fp8 (ldiff_op_norm_apply_lo8): rows of [C fp16 | C e4m3] = 3C
                                                       bytes, C1 = 3C/2, lo8_slab0 = C/64, w built by ldiff_op_lo8_weights; 3x3 stride 1, C % 128 == 0, N % 128 == 0,
                                                       split output with statistics, maps that fill the chip with 16 x 16 tiles; anything else: LDIFF_ERR_INVALID */
  const void* lo8_scale;                            /* the int ldiff_op_lo8_weights wrote (device memory) */
  int gemm_df;                                      /* 1x1 / linear only, producer / consumer ("dataflow") GEMM: 0 = where the executors would pick it (unit list
                                                       fills the chip), -1 = never, 1 = always where the shape is eligible (LDIFF_ERR_INVALID otherwise), 16 mt + ntw
                                                       (mt 4 | 8, ntw 2 | 4 | 5) = always, with units of 16 mt rows x 64 ntw columns (tests, timing) */
  const void* sc_x;                                 /* 3x3 stride-1 GroupNorm + SiLU convs on the dataflow kernel only: the 1x1 conv_shortcut of a ResnetBlock2D that changes width
                                                       (diffusers ResnetBlock2D.conv_shortcut), folded into the block's second conv: y = conv3x3(silu(gn(x))) + sc_w . sc_x + bias +
                                                       sc_bias.  sc_x [B, Hin, Win, sc_C] fp16 (row pitch sc_ld, 0 = sc_C; sc_C % 64 == 0), sc_w [Nrows, sc_C] fp16 K-major,
                                                       sc_bias [Nrows] fp32 or NULL; no res then.  Shapes the dataflow kernel does not take: LDIFF_ERR_INVALID */
  int sc_C, sc_ld;
  const void* sc_w;
  const void* sc_bias;
  int c3d_ups;                                      /* ups = 1 convs on the dataflow kernel: 1 = wherever the shape is eligible (tests, timing); 0 = the executors' choice (never: it does
                                                       not pay in the sampler's step) */
  int n_real;                                       /* output channels of the layer where N (the stored columns, a multiple of 4) rounds them up; 0 = not stated.  The tap-folded
                                                       3x3 kernel for <= 3 output channels (the VAE's conv_out) runs only where this says so */
  int splitk;                                       /* 0 = the executors' plan (none when `stats` is given: this entry point's historical behaviour); 2..16 = that many K splits
                                                       (tests, timing): fp32 partials + the reduce kernel, which then also emits `stats` in blocks of 32 rows
                                                       (ldiff_op_conv_stats_blocks accounts for it; needs 32 | Hout * Wout).  fp16 output, no GEGLU, no parity-folded upsampling */
} ldiff_conv_args;
int ldiff_op_conv(const ldiff_conv_args*, void* stream);
/* row blocks per image the launch would emit statistics for (0 = unsupported for this shape) */
int ldiff_op_conv_stats_blocks(const ldiff_conv_args*);
/* finalize producer-fused partial sums into per-(b,channel) scale/shift; part2 (second concat source) may be NULL */
int ldiff_op_gn_finalize(const void* part1, int R1, int C1, const void* part2, int R2, int C2, int B, int HW, int groups, float eps,
                         const void* gamma, const void* beta, void* scale, void* shift, void* stream);
int ldiff_op_attention(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int ldo, int B, int heads,
                       int Lq, int Lk, int d, int64_t q_bstride, int64_t kv_bstride, int64_t o_bstride, float scale, void* stream);
/* The same with q ALREADY multiplied by scale * log2(e) (the executors do that in the fp32 epilogue of the q/k/v projection, so q is still rounded
 * once): the kernel then lets the MFMAs subtract the running softmax reference (a 1.0 in K's padding column against -reference in Q's) and skips the
 * per-score FMA.  Head dims with a free column in the last 32-wide k-step only (d = 40, 80: the UNet's levels 0 and 1); others: LDIFF_ERR_INVALID. */
int ldiff_op_attention_prescaled(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int ldo, int B, int heads,
                                 int Lq, int Lk, int d, int64_t q_bstride, int64_t kv_bstride, int64_t o_bstride, void* stream);
/* sources are (pointer, channels C, row pitch ld (0 = C), lo offset (0 = plain, > 0 = split tensor hi|lo)) */
int ldiff_op_gn_stats(const void* x, int C1, int ld1, int lo1, const void* x2, int C2, int ld2, int lo2, int B, int HW, int groups, float eps,
                      const void* gamma, const void* beta, void* scale, void* shift, void* stream);
int ldiff_op_layernorm(const void* x, int ldx, int x_lo, void* y, int rows, int C, const void* gamma, const void* beta, float eps, void* stream);
/* LayerNorm folded into the linear layer that consumes it (BasicTransformerBlock norm1 -> to_q/k/v, norm2 -> attn2.to_q, norm3 -> ff.net.0.proj):
 *   y[rows, N] = LayerNorm(x; gamma, beta, eps)[rows, C] . w[N, C]^T + bias,  x as in ldiff_op_layernorm (x_lo > 0: split rows hi | lo),
 *   the normalised operand rounded ONCE to fp16 (as the two-launch form does); geglu as in ldiff_conv_args (y has N/2 columns).
 * qcols > 0 (a multiple of 64, no GEGLU): columns [0, qcols) are multiplied by qscale in fp32 before the rounding (q of a fused q/k/v projection for
 * ldiff_op_attention_prescaled).
 * Returns LDIFF_ERR_INVALID for shapes the kernel does not take (C != 320, N % 64 != 0, ...): callers then use ldiff_op_layernorm + ldiff_op_conv. */
int ldiff_op_ln_linear(const void* x, int ldx, int x_lo, int rows, int C, const void* gamma, const void* beta, float eps, const void* w, int N, int Nrows,
                       const void* bias_or_null, int geglu, void* y, int ldy, int qcols, float qscale, void* stream);
/* y[m, c] = act(x[m, c] * scale[b, c] + shift[b, c]) (GroupNorm-apply, optional SiLU) over the concat of one or two sources,
 * written plain (y_lo = 0) or split */
int ldiff_op_norm_apply(const void* x, int C1, int ld1, int lo1, const void* x2, int C2, int ld2, int lo2, int B, int HW, const void* scale,
                        const void* shift, int silu, void* y, int ldy, int y_lo, void* stream);
/* weights of a contraction over a split operand: per tap [a(Ca) b(Cb) ..] -> [a a b b 0..] */
int ldiff_op_dup_weights(const void* w, void* wd, int Nrows, int taps, int src_tap_stride, int Ca, int Cb, int dst_tap_stride, void* stream);
/* Split conv operand with an fp8 lo half (the lo product is a 2^-11 correction: three mantissa bits reproduce it to within everything else's noise,
 * profiles/r04_precision_study_lo8.txt; the block-scaled MFMA runs e4m3 at twice the fp16 rate).  norm_apply_lo8: y rows [C1 fp16 | C1 e4m3 of
 * lo * 2^15], 3 C1 bytes, C1 % 16 == 0.  lo8_weights: [Nrows][taps][Cin] fp16 -> [Nrows][taps][Cin fp16 | Cin e4m3 of w * 2^sw] (3 Cin bytes per tap),
 * scale_out[0] = 127 - sw (one int, device memory), Cin % 128 == 0. */
int ldiff_op_norm_apply_lo8(const void* x, int C1, int ld1, int lo1, int B, int HW, const void* scale, const void* shift, int silu, void* y, void* stream);
int ldiff_op_lo8_weights(const void* w, void* wd, void* scale_out, int Nrows, int taps, int Cin, void* stream);
int ldiff_op_geglu(const void* x, void* y, int64_t M, int C4, void* stream);
/* lo_off > 0: also store the rounding remainder of channel c at channel lo_off + c (split first-layer input) */
int ldiff_op_nchw_to_nhwc(const void* x_f32, void* y_f16, int B, int C, int H, int W, int Cpad, int lo_off, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Backward-pass primitives of the fine-tuning step  --  `engine.backward(loss)` / `engine.step()`
 *   ldiffusion.py:227-255 (V5 loop), model/loss.py:44-126 (loss); the reference trains at 64x64 images = 8x8 latents, where a step
 * is bound by the 859 M weights it reads and writes, so the contractions of the backward pass reuse ldiff_op_conv:
 *   dgrad: dx = ldiff_op_conv(dy, W rearranged to [Cin][ky'][kx'][Cout] with the taps flipped)          (a layout cast by the caller)
 *   wgrad: dW[n][tap*C+c] = ldiff_op_conv as a GEMM over K = M on dyT = ldiff_op_transpose(dy) and xcolT = ldiff_op_im2col_t(x)
 * and the entry points below add what has no forward counterpart.  Activations / gradients NHWC f16, parameter gradients f32.
 * ldiffusion_amd/autograd.py wraps them as torch.autograd.Function objects (the tape is torch's; every FLOP is in this library).
 * ---------------------------------------------------------------------------------------------- */
/* out[(tap*C + c)][m] (rows of Mpad columns, Mpad % 8 == 0, zero padded) = the im2col matrix of x, transposed; m = (b, oy, ox) */
int ldiff_op_im2col_t(const void* x, void* out, int B, int H, int W, int C, int ks, int stride, int pad, int ups, int Ho, int Wo, int Mpad, void* stream);
/* out[n][m] = x[m][n], rows of Mpad columns (zero padded) */
int ldiff_op_transpose(const void* x, void* out, int M, int N, int ldx, int Mpad, void* stream);
/* db[n] = sum_m dy[m, n]  (bias gradient, f32) */
int ldiff_op_colsum(const void* dy, void* db_f32, int M, int N, int ld, void* stream);
/* GroupNorm (+SiLU) forward that keeps mean / rstd [B, groups] f32, and its backward: dx f16; dgamma / dbeta f32 are ACCUMULATED
 * (atomics over the batch; zero them first) */
int ldiff_op_gn_train_fwd(const void* x, void* y, const void* gamma, const void* beta, void* mean, void* rstd, int B, int HW, int C, int groups, float eps,
                          int silu, void* stream);
int ldiff_op_gn_train_bwd(const void* x, const void* dy, const void* gamma, const void* beta, const void* mean, const void* rstd, void* dx, void* dgamma,
                          void* dbeta, int B, int HW, int C, int groups, int silu, void* stream);
/* LayerNorm backward (statistics recomputed from x); dgamma / dbeta accumulated */
int ldiff_op_ln_bwd(const void* x, const void* dy, const void* gamma, void* dx, void* dgamma, void* dbeta, int rows, int C, float eps, void* stream);
/* GEGLU backward: x [M, 2*C4] = [h | gate], dy [M, C4] -> dx [M, 2*C4] */
int ldiff_op_geglu_bwd(const void* x, const void* dy, void* dx, int64_t M, int C4, void* stream);
/* SiLU on n float16 elements and its backward (the time-embedding MLP of the step: time_embedding.act, the SiLU in front of every time_emb_proj) */
int ldiff_op_silu(const void* x, void* y, int64_t n, void* stream);
int ldiff_op_silu_bwd(const void* x, const void* dy, void* dx, int64_t n, void* stream);
/* softmax(scale Q K^T) V backward for short sequences (Lq * Lk <= 8192: the 8x8-latent fine-tuning step); layouts as ldiff_op_attention */
int ldiff_op_attention_bwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* dO, int ldo, void* dq, void* dk, void* dv,
                           int B, int heads, int Lq, int Lk, int d, int64_t q_bstride, int64_t kv_bstride, int64_t o_bstride, float scale, void* stream);
/* one AdamW update of n f32 parameters (torch.optim.AdamW semantics; step counts from 1) */
int ldiff_op_adamw(void* p, const void* g, void* m, void* v, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                   void* stream);
/* Contrastive (InfoNCE) feature loss of the fine-tuning step for GIVEN sample triples, forward and gradient in one launch
 * (/root/reference/model/loss.py:89-109; the random draws of :62-87 stay on the host, ldiffusion_amd/loss.py sample_triples):
 *   features f32 [B, n, HW] (n <= 32 planes);  triple t = image bi[t], anchor pixel ai[t], positive pi[t], negatives ni[t*K .. t*K+K) (int32, device)
 *   loss[0] = mean_t CE([a.p | a.n_k] / temperature, target 0);  dfeatures [B, n, HW] = d loss / d features.  Both are overwritten.
 *   T_dev (may be NULL): device int32 holding the actual number of triples (<= T, which is then the capacity of the index arrays and the
 *   launch size), read at execution time -- the launch is shape-stable, e.g. inside a captured graph whose batches yield varying counts. */
int ldiff_op_infonce(const void* features, int B, int n, int64_t HW, const void* bi, const void* ai, const void* pi, const void* ni, int T, const void* T_dev,
                     int K, float temperature, void* loss, void* dfeatures, void* stream);
/* Weight layouts of the training step (the float32 master [Cout, Cin, k, k] of torch / diffusers -> what ldiff_op_conv reads):
 *   mode 0, forward: dst[n][ky][kx][c] = w[n][c][ky][kx]           rows >= Cout, Cpad >= Cin, the rest zero
 *   mode 1, dgrad:   dst[c][ky][kx][n] = w[n][c][k-1-ky][k-1-kx]   rows >= Cin,  Cpad >= Cout, the rest zero
 * and back: the wgrad GEMM's g[n][tap*Cx + c] (row pitch ldg) -> dw[n][c][ky][kx] (loss.backward() of /root/reference/ldiffusion.py:254). */
int ldiff_op_pack_weight(const void* w_f32, void* dst_f16, int Cout, int Cin, int k, int rows, int Cpad, int mode, void* stream);
int ldiff_op_unpack_wgrad(const void* g_f32, void* dw_f32, int Cout, int Cin, int k, int Cx, int ldg, void* stream);
/* ldiff_op_pack_weight for MANY (tensor, layout) pairs in one launch.  entries: device array of n_entries records of 48 bytes
 *   { const float* w; _Float16* dst; int32 Cout, Cin, kk (= k*k: 1 or 9), rows, Cpad, mode, tiles_x, 0 }
 * tile_prefix: device int32 [n_entries + 1], tile_prefix[e] = first workgroup of entry e, tile_prefix[n_entries] = n_tiles.  Tiles per entry:
 *   mode 0: tiles_x = ceil(Cpad / 256) times (kk == 1 ? ceil(rows / 8) : rows);  mode 1: tiles_x = ceil(Cpad / 64) times ceil(rows / (kk == 1 ? 64 : 16)). */
int ldiff_op_pack_weight_multi(const void* entries, const void* tile_prefix, int n_entries, int n_tiles, void* stream);

/* All parameters of a model in one launch (the reference's optimiser step, /root/reference/ldiffusion.py:168-171,255: engine.step()).
 * Device tables: tensors[t] = {float* p, float* m, float* v, int64 n} (32 bytes), grads[t] = const float* (gradient of tensor t),
 * chunks[c] = {int32 tensor, int32 pad, int64 first element} (16 bytes): workgroup c updates elements [first, first + 16384) of its tensor. */
int ldiff_op_adamw_multi(const void* tensors, const void* grads, const void* chunks, int64_t nchunks, float lr, float beta1, float beta2, float eps,
                         float weight_decay, int step, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Live measurement for bench.py's roofline line: when enabled, every conv/linear, attention and
 * GroupNorm-statistics launch is bracketed by two HIP events recorded on the launch stream.
 * ldiff_prof_collect waits for them and returns one row per kernel (time, launches, algorithmic flops/bytes).
 * Two events per launch cost ~6 % of a sampler step when every launch carries them, so ldiff_prof_set_filter(name)
 * restricts the brackets to launches of ONE kernel (exact row name, e.g. "conv3x3<8x16,128,gn>"; NULL = all):
 * bench.py profiles every kernel in an untimed warm-up step and only the dominant one inside the timed region.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { char name[64]; int64_t launches; double ms, flops, bytes; } ldiff_prof_row;
int ldiff_prof_enable(int on);
int ldiff_prof_set_filter(const char* kernel_name_or_null);
int ldiff_prof_collect(ldiff_prof_row* rows, int cap); /* returns the number of rows (may exceed cap) or <0 */

/* ------------------------------------------------------------------------------------------------
 * CU-restricted streams (measurement of chip partitioning between the UNet / encoder stream and the
 * decode side stream; no reference counterpart).  A stream made here only runs on the CUs whose index
 * modulo 32 lies in [lo32, hi32) (both multiples of 8): every XCD keeps the same share of its CUs.
 * ldiff_vae_set_side_cu_share re-creates the VAE's decode side stream with such a share (0, 32 = whole chip).
 * ---------------------------------------------------------------------------------------------- */
int ldiff_stream_create_cu_share(int lo32, int hi32, void** stream_out);
int ldiff_stream_destroy(void* stream);
int ldiff_vae_set_side_cu_share(ldiff_vae* v, int lo32, int hi32);

#ifdef __cplusplus
}
#endif
#endif /* LDIFF_H */
