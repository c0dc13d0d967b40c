/*
 * kd_engine.h — C ABI of libkd_engine.so, the MI355X (gfx950) denoising engine that sits
 * under the drop-in `imagen_pytorch` package of this repo.
 *
 * The reference (jameshball/kidney-diffusion) has no FFI: its hot path is the Python API of
 * the third-party `imagen-pytorch==1.18.5` (reference requirements.txt:37).  Each entry point
 * below names the reference-side call it replaces (file:line in /root/reference) and, for the
 * arithmetic, the SURVEY.md Appendix A item it implements.
 *
 * Conventions
 *   - every pointer named d_* is a DEVICE pointer (HBM), borrowed, never freed by the library;
 *   - `stream` is a hipStream_t passed as void*; all launches are stream-ordered.  What synchronises:
 *     kd_unet_create[_shared] (allocates, packs the weights, device-synchronises); the FIRST
 *     kd_sample_* call on a plan (hipMalloc of the sampler scratch, capture + instantiate of the step
 *     graph); a kd_sample_* call whose schedule differs from the previous call's (one stream
 *     synchronise + one asynchronous table upload; an unchanged schedule uploads nothing);
 *     kd_unet_profile and the single-kernel test entry points.  Successive calls on one plan must be
 *     issued on one stream (or otherwise ordered): the plan owns ONE workspace;
 *   - every function returns 0 on success, non-zero on failure; kd_last_error() gives the text;
 *   - images are fp32 NCHW exactly as the reference passes them (sample_ultra_res.py:183-195);
 *     feature maps inside the engine are fp32 NHWC (DESIGN.md "Data layout").
 */
#ifndef KD_ENGINE_H
#define KD_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KD_MAX_LEVELS 8

const char* kd_last_error(void);
/* ABI version of this header: bumped whenever an entry point changes its arguments or a struct its layout, so that a
 * caller built against an older header can refuse the library instead of passing short structs.  History:
 *   1  rounds 1-3
 *   2  round 4/5: kd_conv3x3_winograd4_nhwc gained `gemm_bf16x3` (before `stream`); kd_unet_config_t gained
 *      `gemm_bf16x3`, `x3_linear` and `wino4_max_images`, kd_sample_args_t `cond_table_max_mb`; kd_unet_cond_table_refused_bytes, kd_linear_bf16x3 (+ _seg_rows), kd_downsample_bf16x3, kd_layernorm_ex and kd_layernorm_linear_bf16x3 added;
 *      kd_unet_cond_table_build_ms takes a non-const handle (it reads the build's events on demand) */
#define KD_ENGINE_ABI_VERSION 2
int kd_version(void);
/* sha256 prefix (16 hex digits) of the sources this binary was compiled from (csrc/build_id.py); a build with
 * EXTRA flags carries the suffix "+experiment".  The Python binding refuses a library whose id differs from
 * the sources lying next to it. */
const char* kd_build_id(void);

/* ------------------------------------------------------------------------------------------
 * UNet.  Replaces `Unet(...)` construction + `Unet.forward` of imagen-pytorch as configured at
 * train_ultra_res.py:29-60, train.py:30-65, train_uncond.py:30-61.
 * ---------------------------------------------------------------------------------------- */
typedef struct kd_unet_config {
  int dim;
  int num_levels;
  int dim_mults[KD_MAX_LEVELS];
  int num_resnet_blocks[KD_MAX_LEVELS];
  int layer_attns[KD_MAX_LEVELS];
  int layer_cross_attns[KD_MAX_LEVELS];
  int cond_dim;               /* already defaulted to dim by the host */
  int channels;               /* 3 */
  int cond_images_channels;   /* 0 / 3 / 4 / 6 */
  int lowres_cond;            /* set by Imagen for unets after the first */
  int memory_efficient;
  int init_conv_to_final_conv_residual;
  int cond_on_text;
  int text_tokens;            /* number of pooled text tokens handed to forward (0 if none) */
  int attn_heads;             /* 8 */
  int attn_dim_head;          /* 64 */
  int ff_mult_x2;             /* 2*ff_mult, integer (4 for ff_mult=2) */
  int num_time_tokens;        /* 2 */
  int sinu_dim;               /* learned_sinu_pos_emb_dim, 16 */
  int resnet_groups;          /* 8 */
  int attend_at_middle;       /* 1 */
  int use_gca;                /* use_global_context_attn, 1 */
  /* static shape of the plan */
  int batch;
  int image_size;             /* S: x is [batch, channels, S, S] */
  /* 0 = auto: for the ResnetBlock 3x3 convs Winograd F(2x2,3x3) as 16 batched GEMMs from Cin >= 256,
   * the fused Winograd kernel below that where the map is a multiple of 16x16, Cout of 64 and the
   * launch fills the chip, direct implicit GEMM elsewhere; 1 = direct implicit GEMM everywhere
   * (bitwise the k-ordered fmaf chain); 2 = as 0 without the fused kernel; 3 = as 0 with the fused
   * kernel wherever its shape rules allow (tests); n >= 32 = batched-GEMM Winograd from Cin >= n
   * (experiments / tests) */
  int conv_algo;
  /* Attention similarity variant - imagen-pytorch changed it between versions, and the reference's pinned
   * 1.18.5 is not available to check: 0 = q * dim_head^-0.5 . k (the library default
   * the reference's configs get); 1 = `Unet(cosine_sim_attn=True)`: l2norm(q) . l2norm(k) * 16;
   * 2 = learned per-channel `q_scale` / `k_scale` on the normalised q / k, * 8 (later versions; the host
   * switches to it when a checkpoint carries those keys).  Applies to self-, cross- and text-pooling attention. */
  int attn_qk_norm;
  /* Structural forks between library versions, named by a checkpoint's keys / shapes:
   * downsample_conv4 = 1: Downsample is `Conv2d(dim, dim_out, 4, stride 2, pad 1)` (parameters `<pre>.weight`
   * [d_out, d, 4, 4], `<pre>.bias`) instead of pixel-unshuffle + 1x1 conv (`<pre>.1.weight` [d_out, 4 d, 1, 1]);
   * mid_attn_plain = 1: mid_attn is a residual multi-query attention without a feed-forward (parameters
   * `mid_attn.fn.fn.*`) instead of a TransformerBlock (`mid_attn.layers.0.{0,1}.*`). */
  int downsample_conv4;
  int mid_attn_plain;
  /* Batched-GEMM Winograd layers: cap of the V + D transform buffers per slice of tiles in MiB (the map is walked in
   * slices, bit-identical results); 0 = one slice (default: fastest, largest workspace). */
  int wino_slice_mb;
  /* Winograd F(4x4,3x3) (36 batched GEMMs over tiles of 4x4 outputs) for the ResnetBlock 3x3 convs with at least this
   * many input channels whose GEMMs fill the chip; 0 = default (512; 128 where the GEMMs run on bf16x3), < 0 = never.  fp32 throughout; per-conv relative
   * L2 against fp64 3-4e-6 (F(2x2,3x3): 5e-7) - inside the 2e-5 the UNet forward is held to. */
  int wino43_min_cin;
  /* The position GEMMs of those F(4x4,3x3) layers on the bf16 matrix pipe, every fp32 operand carried as three bf16
   * pieces (a = ah + am + al exactly) and six exact products accumulated in fp32 per k-step (kernels_gemm_bf16x3.hip):
   * fp32-class results - error against fp64 not above the fp32 MFMA path's - at 3/8 of its matrix cycles.
   * 0 = default (where tiles % 256 == 0, Cout % 128 == 0; with it F(4x4,3x3) is taken from Cin >= 128), < 0 = never
   * (fp32 MFMA).  The transformed input V reaches the GEMM either as the three planes, written by the input transform
   * (1 = always), or as fp32 that the GEMM's loader waves split on the way into LDS (2 = always): a third less V traffic
   * for vector work beside the MFMA waves.  0 picks per layer: fp32 where the GEMM waits for HBM rather than for the matrix
   * pipe (Cin Cout / (6 Cin + 4 Cout) < 40: the 64 x 64 level and above of the SR UNet), planes elsewhere.  Same results
   * bit for bit either way. */
  int gemm_bf16x3;
  /* Token GEMMs and 1x1 convs (attention projections, feed-forward, skip convs without output statistics) on the same
   * bf16x3 kernel in its epilogue form (bias / residual / gate, strided rows; the fp32 activations are split by the
   * kernel's loader waves): 0 = default (K >= 512 input channels, rows % 256 == 0, Cout % 128 == 0, at least 64 tiles of
   * 256 x 128; needs gemm_bf16x3 >= 0 and conv_algo == 0), n > 0 = K >= n, < 0 = never (conv_buf_kernel, fp32 MFMA). */
  int x3_linear;
  /* F(4x4,3x3) layers - and the 1x1 convs on the bf16x3 kernel - run in sets of at most this many images, one set of
   * launches after the other (V and D of one set) (0 = default: the whole batch, or - where V / D / the maps of the whole
   * batch pass the 4 GB a buffer resource spans, unet3's outer levels at batch 8 - the largest divisor of the batch that
   * fits).  A test knob: results equal the whole-batch plan's to fp32 rounding (the k-cut of left-over tiles follows the
   * tile count). */
  int wino4_max_images;
} kd_unet_config_t;

/* One named parameter tensor of the UNet's state_dict (key WITHOUT the `unets.N.` prefix,
 * e.g. "downs.0.1.block1.project.weight"), fp32, contiguous, torch layout, on the device. */
typedef struct kd_param {
  const char* name;
  const float* d_data;
  int64_t numel;
} kd_param_t;

typedef struct kd_unet kd_unet_t;

/* Builds the execution plan, re-packs the weights into the engine's layouts (copies; the
 * caller may free its tensors afterwards) and allocates the activation workspace. */
int kd_unet_create(const kd_unet_config_t* cfg, const kd_param_t* params, int n_params,
                   kd_unet_t** out);
/* Same, for another (batch, image_size[, conv_algo]) plan of the SAME UNet: the new plan shares the
 * packed weights of `share_with` (and packs only the forms that plan does not have yet), so a UNet
 * sampled at several batch sizes keeps one copy of its weights in HBM.  `params` must hold the same
 * values as when `share_with` was created.  The store lives until the last plan is destroyed. */
int kd_unet_create_shared(const kd_unet_config_t* cfg, const kd_param_t* params, int n_params,
                          const kd_unet_t* share_with, kd_unet_t** out);
void kd_unet_destroy(kd_unet_t* u);
/* bytes of HBM held (weights + workspace) and algorithmic MACs of one forward (whole batch) */
int64_t kd_unet_hbm_bytes(const kd_unet_t* u);
int64_t kd_unet_weight_bytes(const kd_unet_t* u);  /* the (possibly shared) packed-weight store alone */
int64_t kd_unet_macs(const kd_unet_t* u);
/* MACs the conv / GEMM launches of one step actually issue on the matrix cores: smaller than
 * kd_unet_macs where a 3x3 conv runs as Winograd F(2x2,3x3) (16/36 of its MACs) and by the
 * step-invariant share of the init / final conv that is hoisted out of the step */
int64_t kd_unet_mfma_macs(const kd_unet_t* u);
/* bf16 MACs the plan's bf16x3 GEMMs issue per forward (six per fp32 MAC of theirs; not included in kd_unet_mfma_macs) */
int64_t kd_unet_mfma_bf16_macs(const kd_unet_t* u);
int kd_unet_num_launches(const kd_unet_t* u);
/* ... of which conditioning launches (functions of log_snr / lowres_log_snr / text only): the sampler replaces them by ONE
 * gather per iteration when it runs from the conditioning table (kd_sample_args_t::cond_table). */
int kd_unet_num_cond_launches(const kd_unet_t* u);
/* device time (ms), row count (*rows) and runs of the conditioning ops (*runs; both may be NULL) of the plan's last
 * conditioning-table build, < 0 if none was built
 * yet.  Rows are built on demand for the schedule steps a call walks, B steps per run of the conditioning ops (their rows
 * are independent over the batch): T / B runs for a whole schedule. */
float kd_unet_cond_table_build_ms(kd_unet_t* u, int* rows, int* runs);
/* > 0: the last sampling call wanted a conditioning table of this many bytes and did not get it (larger than
 * kd_sample_args_t::cond_table_max_mb / an eighth of the free HBM, or the allocation failed): its steps then compute
 * the conditioning themselves - same results, more launches per step */
int64_t kd_unet_cond_table_refused_bytes(const kd_unet_t* u);

/* Step-invariant text conditioning (the text branch of Unet.forward; reached by the
 * reference through sample_cond.py:36-48 / sample.py:51-60): text_to_cond, null-embedding select,
 * PerceiverResampler pooling, to_text_non_attn_cond.  Run once per sample call.
 *   d_text_embeds [B,L,text_embed_dim], d_text_mask [B,L] as 0/1 floats, L <= max_text_len;
 *   drop = 1 gives the null conditioning (cond_drop_prob = 1, used for classifier-free guidance);
 *   out: d_text_tokens [B,text_tokens,cond_dim], d_text_hiddens [B,time_cond_dim]. */
int kd_unet_text_cond(kd_unet_t* u, const float* d_text_embeds, const float* d_text_mask, int L, int drop,
                      float* d_text_tokens, float* d_text_hiddens, void* stream);

/* Diagnostic: per-launch device time of one forward as CSV "index,label,macs,avg_us,mfma_macs" (macs: the
 * algorithmic MACs of the launch as the reference computes them, mfma_macs: what it issues on the matrix
 * cores; uses the inputs of the preceding kd_unet_forward call; synchronises the stream). */
int kd_unet_profile(kd_unet_t* u, int iters, char* buf, size_t buflen, void* stream);

/* Replaces `unet.forward_with_cond_scale(x, log_snr(t), lowres_cond_img=..,
 * lowres_noise_times=.., cond_images=.., text_embeds=..)` at cond_scale == 1.
 *   d_x            [B,3,S,S]
 *   d_lowres       [B,3,S,S] or NULL (required iff cfg.lowres_cond)
 *   d_cond_images  [B,Cc,S,S] already nearest-resized to S, or NULL (required iff Cc > 0)
 *   d_log_snr      [B] log-SNR of the current time; d_lowres_log_snr [B] or NULL
 *   d_text_tokens  [B,text_tokens,cond_dim] pooled text tokens (pre norm_cond) or NULL
 *   d_text_hiddens [B,time_cond_dim] or NULL
 *   d_out          [B,3,S,S] */
int kd_unet_forward(kd_unet_t* u, const float* d_x, const float* d_lowres,
                    const float* d_cond_images, const float* d_log_snr,
                    const float* d_lowres_log_snr, const float* d_text_tokens,
                    const float* d_text_hiddens, float* d_out, void* stream);

/* Classifier-free guidance combine of `Unet.forward_with_cond_scale` (sample.py:55-59 reaches it with
 * cond_scale != 1): out = null + (cond - null) * cond_scale over n floats; out may alias cond. */
int kd_cfg_combine(const float* d_cond, const float* d_null, float* d_out, float cond_scale, int64_t n,
                   void* stream);

/* ------------------------------------------------------------------------------------------
 * Sampler.  Replaces `Imagen.p_sample_loop` / `p_sample` / `p_mean_variance` (SURVEY §3.2,
 * A.2) as entered from sample_ultra_res.py:183-195, outpainting.py:146-157,
 * sample_cond.py:40-48, sample_uncond.py:49-55.
 * ---------------------------------------------------------------------------------------- */
enum { KD_OBJ_NOISE = 0, KD_OBJ_V = 1, KD_OBJ_X_START = 2 };

/* Per-(timestep) scalars precomputed by the host in fp32 exactly as the library's torch ops
 * produce them.  All arrays have T entries. */
typedef struct kd_schedule {
  int T;
  const float* log_snr;       /* log_snr(t_k)            -> UNet time input               */
  const float* alpha;         /* sqrt(sigmoid(log_snr))                                    */
  const float* sigma;         /* sqrt(sigmoid(-log_snr))                                   */
  const float* alpha_next;    /* alpha at t_{k+1}                                          */
  const float* sigma_next;    /* sigma at t_{k+1}                                          */
  const float* c;             /* -expm1(log_snr - log_snr_next)                            */
  const float* noise_scale;   /* [t_next != 0] * exp(0.5*log(max(sigma_next^2*c,1e-20)))   */
  /* inpainting re-noise t_next -> t:  x*rn_a + noise*rn_b ; 0-length use allowed if unused */
  const float* rn_a;          /* alpha_t/alpha_next                                         */
  const float* rn_b;          /* (sigma_t*alpha_next - sigma_next*alpha_t)/alpha_next       */
} kd_schedule_t;

typedef struct kd_sample_args {
  int objective;              /* KD_OBJ_*  (train_ultra_res.py:87, train.py:90) */
  int dynamic_threshold;      /* 1: s = max(1, quantile_0.95 |x0|) per sample   */
  float percentile;           /* 0.95 */
  int resample_times;         /* inpaint_resample_times if inpainting else 1 (sample_ultra_res.py:192) */
  /* conditioning, constant over the loop */
  const float* d_lowres;          /* [B,3,S,S] noised low-res conditioning or NULL */
  const float* d_lowres_log_snr;  /* [B] or NULL */
  const float* d_cond_images;     /* [B,Cc,S,S] or NULL */
  const float* d_text_tokens;     /* or NULL */
  const float* d_text_hiddens;    /* or NULL */
  /* inpainting (both NULL or both set): image already normalised to [-1,1] and resized,
   * mask [B,1,S,S] as 0/1 floats (sample_ultra_res.py:149-174) */
  const float* d_inpaint_images;
  const float* d_inpaint_masks;
  /* noise: explicit tensors (parity tests) or on-device Philox (seed) when the pointer is NULL.
   * d_noise_step    [T*R,B,3,S,S]   index (k*R + (R-1-r))
   * d_noise_inpaint [T*R,B,3,S,S]   same indexing (only read when inpainting)
   * d_noise_renoise [T*R,B,3,S,S]   same indexing (only read when inpainting and r != 0) */
  const float* d_noise_step;
  const float* d_noise_inpaint;
  const float* d_noise_renoise;
  uint64_t seed;
  int use_graph;              /* 1: capture one step into a hipGraph and replay it */
  /* classifier-free guidance (sample.py:55-59): pred = null + (cond - null) * cond_scale, two UNet
   * forwards per step when cond_scale != 1; the null conditioning comes from kd_unet_text_cond(drop=1) */
  float cond_scale;           /* 1.0 = off */
  const float* d_null_text_tokens;
  const float* d_null_text_hiddens;
  /* Conditioning table: when the UNet has no text conditioning and (if it is low-res conditioned) the caller states that
   * all B entries of d_lowres_log_snr hold ONE value (what Imagen.sample passes), the time conditioning of a step - time
   * embeddings, FiLM scale / shift of every ResnetBlock, time tokens and their cross-attention K / V - is a function of
   * the schedule index alone: it is computed once per (schedule, value) into a table and restored per iteration by one
   * gather.  Bit-identical results either way.  cond_table: 0 = use it when possible, < 0 = never.  The table is
   * allocated only if T * cond_bytes fits cond_table_max_mb (0 = 4096) and 1/8 of the free HBM (a failed allocation
   * falls back to computing the conditioning in the step); its rows are built for the steps a call walks. */
  int lowres_log_snr_uniform;   /* 1: all entries of d_lowres_log_snr equal lowres_log_snr_value */
  float lowres_log_snr_value;
  int cond_table;
  int cond_table_max_mb;
} kd_sample_args_t;

/* In: d_img = x_T [B,3,S,S].  Out: d_img = unnormalised sample in [0,1] (clamp, final inpaint
 * paste and (x+1)/2 included).  Runs all T*R denoising iterations on `stream`. */
int kd_sample_loop(kd_unet_t* u, const kd_schedule_t* sched, const kd_sample_args_t* args,
                   float* d_img, void* stream);
/* Runs iterations [k_begin, k_end) only, without the final clamp/unnormalise (used by bench.py
 * to time exactly K steps, and by tests to compare intermediate states). */
int kd_sample_steps(kd_unet_t* u, const kd_schedule_t* sched, const kd_sample_args_t* args,
                    float* d_img, int k_begin, int k_end, void* stream);
/* Builds (force != 0: rebuilds) the conditioning-table rows of schedule steps [k_begin, k_end) without sampling, so
 * that the first sampling call of a schedule does not pay for them; *built (may be NULL) = rows built, 0 when the plan
 * runs without a table.  Device time of the build: kd_unet_cond_table_build_ms. */
int kd_sample_build_cond_table(kd_unet_t* u, const kd_schedule_t* sched, const kd_sample_args_t* args, int k_begin,
                               int k_end, int force, int* built, void* stream);
/* The tail of p_sample_loop on its own: clamp(-1,1), paste of the known inpaint pixels,
 * (x+1)/2.  kd_sample_loop == kd_sample_steps(0,T) + kd_sample_finalize. */
int kd_sample_finalize(kd_unet_t* u, const kd_sample_args_t* args, float* d_img, void* stream);
/* What the last executed iteration left behind (parity checks against p_mean_variance of the library):
 * which = 0: the UNet's output eps-hat / v-hat after guidance [B,3,S,S]; 1: the x0 estimate before the
 * threshold clamp [B,3,S,S]; 2: the per-sample dynamic thresholds max(1, quantile_p |x0|) [B].
 * Stream-ordered device-to-device copy into d_out. */
int kd_sample_last(kd_unet_t* u, int which, float* d_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Individual kernels, exported so that tests/ can check each one against the oracle through
 * the same ABI the plan uses internally.
 * ---------------------------------------------------------------------------------------- */

/* 2-D convolution as implicit GEMM on fp32 MFMA.  x: NHWC [B,Hi,Wi,Cin]; w: torch OIHW
 * [Cout,Cin,KH,KW] (re-packed internally on each call of this test entry); y: NHWC.
 * act: 0 none, 1 SiLU, 2 GELU(erf); | 0x100: the row-run K layout of the init convs; | 0x200 (1x1 convs): the
 * upsample form, PixelShuffle(2) of the conv output, y is [B, 2 Ho, 2 Wo, Cout / 4]. */
int kd_conv2d_nhwc(const float* d_x, const float* d_w_oihw, const float* d_bias, float* d_y,
                   int B, int Hi, int Wi, int Cin, int Cout, int KH, int KW, int stride, int pad,
                   int act, void* stream);
/* The same 3x3 / stride-1 / pad-1 convolution through the plan's Winograd F(2x2,3x3) path (used for
 * the deep ResnetBlock convs): fp32, differs from kd_conv2d_nhwc by re-association
 * only.  Needs even H, W; B*H*W/4 % 256 == 0; Cin % 32 == 0; Cout > 32. */
int kd_conv3x3_winograd_nhwc(const float* d_x, const float* d_w_oihw, const float* d_bias, float* d_y,
                             int B, int H, int W, int Cin, int Cout, void* stream);
/* The same convolution (+ optional residual d_res, NHWC with Cout channels) through the plan's Winograd F(4x4,3x3)
 * path for the deepest layers (kernels_wino4.hip): weight transform, input transform, 36 batched GEMMs, output transform.
 * Needs H % 4 == 0, W % 4 == 0, B (H/4) (W/4) % 128 == 0, Cin % 32 == 0, Cout % 64 == 0.  d_out_stats (may be NULL):
 * [B, G, 2] = (mean, rstd) of y per image and group of Cout / G channels from the partial sums the output transform
 * leaves for the next GroupNorm; needs (Cout / G) % 16 == 0.  gemm_bf16x3 = 1: the GEMMs on the bf16 matrix pipe as the
 * plan runs them by default (V in plane form), 2: V as fp32, split by the GEMM's loader waves (kd_unet_config_t::gemm_bf16x3;
 * both need B (H/4) (W/4) % 256 == 0 and Cout % 128 == 0 as well). */
int kd_conv3x3_winograd4_nhwc(const float* d_x, const float* d_w_oihw, const float* d_bias, const float* d_res,
                              float* d_y, int B, int H, int W, int Cin, int Cout, int G, float eps,
                              float* d_out_stats, int gemm_bf16x3, void* stream);
/* C[g][M][N] = A[g][M][K] B[g][N][K]^T (fp32, row-major) through the bf16x3 GEMM of kernels_gemm_bf16x3.hip: both
 * operands split into three bf16 planes, six bf16 MFMAs per k-step, fp32 accumulation.  a_planes != 0: both operands
 * split beforehand (the plan's form: weights once, activations by the kernel that writes them); a_planes = 0: A stays fp32
 * and is split by the kernel's loader waves on its way into LDS.  Same planes, same bits.  Needs M % 256 == 0, N % 128 == 0, K % 32 == 0, 6 G M K and 6 G N K below 2^32 bytes. */
int kd_gemm_bf16x3(const float* d_a, const float* d_b, float* d_c, int G, int M, int N, int K, int a_planes, void* stream);
/* The token-GEMM / 1x1-conv form of the same kernel, as the plan runs the attention projections, the feed-forward and
 * the skip convs with K >= 512 (kd_unet_config_t::x3_linear): y[m][n] = sum_k x[m][k] w[n][k] + bias[n]
 * (+ gate_src[m][n] gate[m / hw][n]) (+ res[m][n]), fp32 rows with strides ldx / ldres / ldgs / ldy (0 = dense), the
 * activations split into their three bf16 pieces by the kernel's loader waves, the weights once.  Optional pointers may
 * be NULL.  d_seg (optional): the (sum, sum of squares) partials of y the launch leaves for a GroupNorm that reads it,
 * fp64 [M / hw][N / 16][hw / rows][2] with rows = kd_linear_bf16x3_seg_rows(M, N, K) (32, or 8 where the tiles are cut in
 * k).  act (0 none, 1 SiLU, 2 GELU, 3 sigmoid) is applied to the product + bias.  pixshuf_wo > 0: the Upsample form
 * (conv1x1 -> SiLU -> PixelShuffle(2)): the rows are the pixels of maps of width pixshuf_wo, d_w's rows are packed
 * n' = (2 i + j) N/4 + c, y is the [4 M][ldy] map of N / 4 channels and the statistics chunks are 4 per 32 input pixels
 * (one per sub-position).  Replaces nn.Linear / 1x1 nn.Conv2d + the residual / GlobalContext-gate adds around them and
 * Upsample's conv + activation + rearrange. */
int kd_linear_bf16x3(const float* d_x, int ldx, const float* d_w, const float* d_bias, const float* d_res, int ldres,
                     const float* d_gate_src, int ldgs, const float* d_gate, int hw, float* d_y, int ldy, int M, int N, int K,
                     int act, int pixshuf_wo, double* d_seg, void* stream);
int kd_linear_bf16x3_seg_rows(int M, int N, int K);
/* The library's Downsample (Rearrange 'b c (h s1) (w s2) -> b (c s1 s2) h w' + Conv2d(4 C, O, 1): a 2 x 2 / stride-2 conv,
 * SURVEY A.1) on the same kernel: x NHWC [B][H][W][ldx] (C channels used), d_w the torch weight [O][4 C], y NHWC
 * [B][H/2][W/2][O]; the kernel's loader waves gather the four input pixels of an output pixel.  d_seg as above with
 * hw = (H/2) (W/2) and rows = kd_linear_bf16x3_seg_rows(B hw, O, 4 C). */
int kd_downsample_bf16x3(const float* d_x, int ldx, const float* d_w, const float* d_bias, float* d_y, int B, int H, int W, int C,
                         int O, double* d_seg, void* stream);
/* ResnetBlock `Block` in one pass over x: conv3x3(SiLU(FiLM(GroupNorm_G(x)))) + bias (+ d_res), the form the
 * plan uses for those layers: statistics, a per-(image, channel) affine fold, and the fused Winograd kernel
 * with the activation applied to the raw patch in LDS (the activated map is never written).  d_scale_shift
 * ([B, 2 Cin] = [scale | shift]) and d_res (NHWC, Cout channels) may be NULL.  Needs H % 16 == 0, W % 16 == 0, Cin % 4 == 0,
 * Cout % 64 == 0 (non-zero return otherwise), Cin <= 2048, Cin % G == 0, (Cin / G) % 4 == 0, 16-byte aligned d_y / d_bias / d_res.  Runs the kernel the plan would
 * pick for the shape: items of 16 x 8 pixels x 128 output channels where Cout % 128 == 0
 * (kernels_wino_fused128.hip), 16 x 16 pixels x 64 channels otherwise (kernels_wino_fused.hip). */
int kd_gn_conv3x3_winograd_fused_nhwc(const float* d_x, const float* d_gamma, const float* d_beta,
                                      const float* d_scale_shift, const float* d_w_oihw,
                                      const float* d_bias, const float* d_res, float* d_y, int B, int H,
                                      int W, int Cin, int Cout, int G, float eps, float* d_out_stats,
                                      int ldx, void* stream);

/* (ldx: row stride of d_x in floats, >= Cin and a multiple of 4, 0 = dense: the plan hands this kernel channel-slice
 * views of wider buffers - a skip tensor living in the concat it will join.) */
/* (d_out_stats, may be NULL: [B, G, 2] = (mean, rstd) of y per image and group of Cout / G channels, reduced
 * from the partial sums the kernel's epilogue leaves for the GroupNorm of the next layer; needs
 * (Cout / G) % 16 == 0.) */
/* The UNet's initial cross-embed convolution (three stride-1 convs k = 3 / 7 / 15) over a 3-plane NCHW
 * image in the plan's one-kernel form: y NHWC [B][S][S][n3+n7+n15] = cat(conv3, conv7, conv15)(x) + bias.
 * d_w3 / d_w7 / d_w15: OIHW [n][3][k][k].  Needs S % 32 == 0, n3 <= 64, n7 <= 32, n15 <= 32, each a multiple of 4. */
int kd_init_conv_nchw(const float* d_x, const float* d_w3, const float* d_w7, const float* d_w15,
                      const float* d_bias, float* d_y, int B, int S, int n3, int n7, int n15, void* stream);
/* GroupNorm(G) + optional FiLM (scale+1, shift: [B,2C] = [scale | shift]) + SiLU, NHWC. */
int kd_groupnorm_silu_nhwc(const float* d_x, const float* d_gamma, const float* d_beta,
                           const float* d_scale_shift, float* d_y, int B, int HW, int C, int G,
                           float eps, void* stream);
/* LayerNorm over the last dim of [rows, C]; beta may be NULL (gain-only). */
int kd_layernorm(const float* d_x, const float* d_g, const float* d_beta, float* d_y,
                 int rows, int C, float eps, void* stream);
/* The forms the TransformerBlock's plan uses: y = LN(f(x)) g (+ beta) (+ res) with f = in_act (0 none, 1 SiLU, 2 GELU:
 * FeedForward's Linear -> GELU -> LayerNorm when the GEMM stores the raw product), and, with d_g2 / d_y2 given, also
 * y2 = LN(y) g2 in the same pass (attention's to_out LayerNorm + residual followed by the feed-forward's first). */
int kd_layernorm_ex(const float* d_x, const float* d_g, const float* d_beta, const float* d_res, float* d_y, int rows, int C,
                    float eps, int in_act, const float* d_g2, float* d_y2, void* stream);
/* LayerNorm -> Linear as the TransformerBlock's plan runs them where the GEMM is a bf16x3 one: the LayerNorm leaves
 * y = LN(f(x)) g (+ beta) as the GEMM's three bf16 planes ([3][C / 16][rows][16] bf16 in d_planes, 6 rows C bytes; the
 * planes of a value add up to it exactly), the GEMM reads them with its DMA loaders: d_y [rows, N] (row stride ldy) =
 * y @ w[N, C]^T + bias + res.  rows % 256 == 0, N % 128 == 0, C % 32 == 0, C <= 4096.  Same products in the same order as
 * kd_layernorm_ex followed by kd_linear_bf16x3: bit-identical results. */
int kd_layernorm_linear_bf16x3(const float* d_x, const float* d_g, const float* d_beta, int rows, int C, float eps, int in_act,
                               const float* d_w, const float* d_bias, const float* d_res, int ldres, float* d_y, int ldy,
                               int N, void* d_planes, void* stream);
/* Attention with fp32 softmax.  q [B,Nq,H,D] (already scaled), k/v [B,Nk,Hkv,D] with
 * Hkv in {1,H}; out [B,Nq,H,D].  D must be 64. */
int kd_attention(const float* d_q, const float* d_k, const float* d_v, float* d_out,
                 int B, int Nq, int Nk, int H, int Hkv, int D, void* stream);
/* Per-sample linear-interpolated quantile of |x| over n values (torch.quantile semantics). */
int kd_quantile_abs(const float* d_x, float* d_out, int B, int64_t n, float q, void* d_workspace,
                    size_t workspace_bytes, void* stream);
size_t kd_quantile_workspace_bytes(int B);
/* Standard-normal Philox4x32-10 fill (the generator kd_sample_loop uses when no noise tensors
 * are given): element i of stream `stream_id` under `seed`. */
int kd_philox_normal(float* d_out, int64_t n, uint64_t seed, uint64_t stream_id, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KD_ENGINE_H */
