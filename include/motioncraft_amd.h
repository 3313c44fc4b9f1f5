/* C-ABI of libmotioncraft_amd.so -- the MI355X (gfx950) STMoGen sampling hot path.
 *
 * This is the drop-in boundary for the per-step denoising path of cure-lab/MotionCraft
 * (SURVEY.md section 8b).  The reference is pure Python/PyTorch and has no FFI of its own; the
 * entry points below are what a binding for that path replaces, each citing the reference
 * interface (paths relative to the reference root):
 *
 *   mc_model_*            weights of  STMoGenTransformer               mogen/models/transformers/stmogen.py:626-653
 *                         (state dict as loaded by mmcv load_checkpoint, tools/test.py:99)
 *   mc_ctx_set_timesteps  timestep_map of SpacedDiffusion/_WrappedModel  mogen/models/utils/gaussian_diffusion.py:1416-1463
 *                         + time_embed / emb_layers hoists             diffusion_transformer.py:89-93,206-208; stylization_block.py:17-20
 *   mc_ctx_set_condition  model_kwargs {xf_out, motion_mask}           mogen/models/architectures/diffusion_architecture.py:166-174
 *                         + per-layer text_moe K/V hoist               mogen/models/attentions/st_attention.py:116-118
 *   mc_ctx_set_control    ControlT2MHalf.forward_c + before_proj           mogen/models/transformers/controlnet.py:66,186-199
 *   mc_denoise            model(x, ts, **model_kwargs)                 diffusion_transformer.py:186-238 -> stmogen.py:725-761
 *   mc_sample_step        GaussianDiffusion.p_sample / ddim_sample     gaussian_diffusion.py:634-696, 799-852
 *   mc_sample_step_inpaint  the same with y = {gt, outpainting_mask}   gaussian_diffusion.py:492-501, 855-877
 *   mc_sample_step_seeded   the same with pre_seq / transl_req         gaussian_diffusion.py:664-674, 816-820
 *   mc_textenc_*          DiffusionTransformer.encode_text (CLIP text tower, text_pre_proj, textTransEncoder, text_ln)
 *                                                                      mogen/models/transformers/diffusion_transformer.py:109-172
 *   mc_evalenc_*          T2MContrastiveModel_SMPLX.encode_motion / encode_text (evaluation embeddings)
 *                                                                      mogen/models/rnns/t2m_bigru_smplx.py:66-437
 *   mc_t2meval_*          T2MContrastiveModel.encode_motion / encode_text (HumanML3D / KIT evaluation embeddings)
 *                                                                      mogen/models/rnns/t2m_bigru.py:72-299
 *   mc_wavenc_*           WavEncoder (audio condition pre-encoder)     mogen/models/utils/blocks.py:11-71; controlnet.py:90-105,187
 *   mc_postprocess_smplx  de-normalise + SMPL-X re-pack + temporal filter  tools/visualize.py:39-44,217-246; tools/s2g_test.py:289-297
 *   mc_op_renoise         GaussianDiffusion._undo (resampling jumps)   gaussian_diffusion.py:429-435, 1113-1118
 *
 * Conventions: plain pointers and sizes only.  `*_dev` pointers are device (HBM) addresses owned
 * by the caller (e.g. torch allocations); `stream` is a hipStream_t passed as void*.  All tensors
 * are fp32, contiguous, row-major.  Every function returns 0 on success or an MC_ERR_* code;
 * mc_last_error() returns a thread-local description.  A handle is re-entrant across handles but
 * not thread-safe per handle.  No call synchronises the device except mc_model_set_param
 * (synchronous H2D upload) and the *_create/_destroy functions.
 */
#ifndef MOTIONCRAFT_AMD_H
#define MOTIONCRAFT_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MC_OK 0
#define MC_ERR_ARG 1
#define MC_ERR_HIP 2
#define MC_ERR_STATE 3

typedef struct mc_model mc_model;
typedef struct mc_ctx mc_ctx;

/* the configs under configs/stmogen: model=dict(type='STMoGenTransformer', ...) */
typedef struct mc_model_config {
    int32_t input_feats;      /* 322 (SMPL-X motionx layout)                       */
    int32_t max_seq_len;      /* 196                                               */
    int32_t latent_dim;       /* L: per-part latent (ca_block_cfg.latent_dim)      */
    int32_t num_parts;        /* H: 12 body parts (ca_block_cfg.num_heads)         */
    int32_t num_layers;       /* NL                                                */
    int32_t ffn_dim;          /* F: SFFN hidden (ffn_cfg.ffn_dim)                  */
    int32_t time_embed_dim;   /* Te                                                */
    int32_t text_latent_dim;  /* Dt                                                */
    int32_t max_text_len;     /* Nt = 77                                           */
    int32_t num_experts;      /* E = 16                                            */
    int32_t topk;             /* 2                                                 */
    int32_t dyn_heads;        /* 8 (st_attention.py:95)                            */
    float capacity_factor;    /* 1.5 (st_attention.py:33)                          */
    float cfg_scale;          /* scale_func_cfg.scale = 6.5                        */
    /* plug-and-play control branch ControlT2MHalf (mogen/models/transformers/controlnet.py:107-183); 0 = none */
    int32_t num_ctrl_layers;  /* copy_blocks_num                                   */
    int32_t ctrl_cond_feats;  /* width of the condition fed to control_cond_input  */
    int32_t ctrl_condition_cfg; /* condition_encode_cfg.condition_cfg: zero c in the uncond half */
} mc_model_config;

/* per-step scalars of the sampler, fp64 schedule tables cast to fp32 like _extract_into_tensor
 * (gaussian_diffusion.py:1330-1343) */
typedef struct mc_step_coefs {
    int32_t mode;             /* 0 = DDPM p_sample, 1 = DDIM ddim_sample           */
    float text_coef;          /* w = 1 + scale * t_orig / 1000 (stmogen.py:655-659) */
    float none_coef;          /* 1 - w                                             */
    float c1, c2;             /* posterior_mean_coef1/2[i]                          */
    float log_var;            /* log(append(posterior_variance[1], betas[1:]))[i]   */
    float sqrt_recip, sqrt_recipm1, ab, ab_prev, eta;   /* DDIM                     */
    float nonzero;            /* (t != 0)                                           */
} mc_step_coefs;

/* RePaint / outpainting operands of one step (long-sequence windows): model_kwargs['y'] of the reference */
typedef struct mc_inpaint {
    const float* gt_dev;          /* y['gt'] [B,T,C]                                                  */
    const uint8_t* keep_dev;      /* y['outpainting_mask'] [B,T,C], one byte per element (torch.bool)  */
    const float* gt_noise_dev;    /* 2nd randn_like of ddim_sample (:868) [B,T,C]; unused for DDPM     */
    const float* blend_w_dev;     /* linspace(0,1,overlap_len) (:873); may be NULL when blend_len == 0 */
    int32_t blend_len;            /* overlap_len if sqrt(1-alpha_bar_prev) < 0.2 and opt.addBlend, else 0 */
} mc_inpaint;

/* pre_seq / transl_req seeding of one step (p_sample :664-674, ddim_sample :816-820): before the network sees x_t,
 * x[:, :pre_len, :] = q_sample(pre_seq, t, randn_like(pre_seq)) and x[:, :2, channel_k] = q_sample(transl_k, t, randn(2)) */
#define MC_MAX_TRANSL 8
typedef struct mc_seed {
    const float* pre_seq_dev;     /* [B, pre_len, C]; may be NULL when pre_len == 0                                  */
    const float* pre_noise_dev;   /* [B, pre_len, C] the randn_like(pre_seq) of THIS step                            */
    int32_t pre_len;              /* pre_seq.shape[1]                                                                */
    float sqrt_ab, sqrt_1mab;     /* sqrt_alphas_cumprod[i], sqrt_one_minus_alphas_cumprod[i] cast to fp32           */
    int32_t num_transl;           /* len(transl_req) <= MC_MAX_TRANSL (DDPM only)                                     */
    int32_t transl_channel[MC_MAX_TRANSL];   /* item[0]                                                              */
    float transl_value[MC_MAX_TRANSL][2];    /* q_sample(item[1:], t, randn(2)): 2 scalars, evaluated by the host     */
} mc_seed;

const char* mc_last_error(void);
int mc_device_count(int* n);
int mc_set_device(int dev);

/* ---- model (weights) -------------------------------------------------------------- */
int mc_model_create(const mc_model_config* cfg, mc_model** out);
void mc_model_destroy(mc_model* m);
/* Upload one packed fp32 parameter from host memory.  Names and layouts: motioncraft_amd/weights.py */
int mc_model_set_param(mc_model* m, const char* name, const float* host, int64_t numel);
int mc_model_finalize(mc_model* m);

/* ---- context = workspace for one (batch, frames) shape ------------------------------ */
int mc_ctx_create(mc_model* m, int32_t batch, int32_t frames, int32_t max_steps, mc_ctx** out);
void mc_ctx_destroy(mc_ctx* c);
int64_t mc_ctx_workspace_bytes(const mc_ctx* c);
/* Synchronises `stream` and reports a sticky device-side error of the context (MC_ERR_STATE): today the one error a kernel can
 * raise is the grid-barrier time-out of the cooperative routing kernel (route_coop_k; its grid is reserved out of the device's
 * resident-workgroup capacity per context at mc_ctx_create, so only barrier kernels of ANOTHER process can starve it).  The
 * Python sampler loops call it once after the last step -- the reference has no counterpart (its routing is tutel's, on the
 * framework's stream: st_attention.py:28-45). */
int mc_ctx_check(mc_ctx* c, void* stream);
/* Measurement hook (bench.py `roofline.dominant_kernel`; no reference counterpart): while on, every FiLM out_layers GEMM launch
 * (h += Linear(a), stylization_block.py:39 -- the dominant kernel of a step) is bracketed by HIP events recorded on the stream it is
 * launched on.  mc_ctx_profile_read waits for them and returns the average duration (us), the number of launches and the
 * algorithmic GFLOP of one launch (2 rows D^2); rows_filter > 0 keeps only launches of that many rows. */
int mc_ctx_profile(mc_ctx* c, int32_t on);
int mc_ctx_profile_read(mc_ctx* c, int64_t rows_filter, double* avg_us, int32_t* count, double* gflop_per_launch);
/* the precision the per-step kernels of this context actually run in: MC_PREC_F32 also when a reduced-precision mode is set but
 * the batch is at most MC_HALF_MIN_ROWS (512) residual rows, where the fp32 small-batch kernels are the faster ones (B = 1) */
int mc_ctx_effective_precision(const mc_ctx* c);
/* 1 if this context routes its layers with the one-launch cooperative kernel (0: one-workgroup kernels or the launch sequence) */
int mc_ctx_uses_coop_routing(const mc_ctx* c);
/* tests: keep the routing decisions (expert ids, combine weights; 0 = dropped) of every layer of the
 * last mc_denoise call in buffers "cap_idx" / "cap_w" */
int mc_ctx_enable_capture(mc_ctx* c);
/* MFMA operand precision of the per-step GEMM-shaped kernels of this context (FiLM out_layers GEMMs, expert and SFFN MLPs,
 * control after_proj): the "fp16 MFMA" mode of BASELINE.json configs[4]; the reference hook is mmcv's wrap_fp16_model
 * (tools/test.py:95-97).  MC_PREC_F32 (default): exact fp32 MFMA.  MC_PREC_F16: operands rounded to fp16, fp32
 * accumulate (~2e-4 relative per GEMM stage).  MC_PREC_F16X3: operands split x = hi + lo in fp16, three products
 * hi*hi + hi*lo + lo*hi accumulated in fp32 -- fp32-class results at 3/16 of the fp32 MFMA time.  The gate, routing,
 * LayerNorm statistics, softmaxes and all elementwise work stay fp32 in every mode (tutel forces fp32_gate,
 * st_attention.py:31).  The fp16 weight planes are built once per model on the first call.
 * Tolerance: MC_PREC_F16X3 meets every bound of the fp32 path (tests/test_gpu_parity.py holds it to the same 2e-4 per call /
 * 1e-3 per trajectory).  MC_PREC_F16 is OUTSIDE the north-star tolerance on the x0 prediction (8e-3 observed at B = 16, t = 640: one
 * fp16 rounding per operand, amplified by the classifier-free-guidance weights); what it meets is 1e-3 on x_{t-1} of every sampler
 * step (1.1e-4 observed), since the update damps the x0 error -- the same class of result as the reference under
 * wrap_fp16_model, which is what configs[4] names. */
#define MC_PREC_F32 0
#define MC_PREC_F16 1
#define MC_PREC_F16X3 2
int mc_ctx_set_precision(mc_ctx* c, int32_t precision);
/* tutel boundary (parity unpinned): order of tokens with EXACTLY equal importance (max gate score) at an
 * expert's capacity cut.  tutel ranks by `importance_scores.argsort(dim=0)` -- not a stable sort, so the order of ties is
 * implementation-defined there.  MC_TIE_STABLE (default): lower token index first (what a stable sort / radix sort
 * gives; oracle/tutel_restated.py TIE_POLICY='stable'); MC_TIE_REVERSE: higher index first.  One switch in the kernel
 * and one in the oracle, so a golden from a real tutel install can be matched without kernel work.  Call before
 * mc_ctx_set_condition (the hoisted text K/V are routed too). */
#define MC_TIE_STABLE 0
#define MC_TIE_REVERSE 1
int mc_ctx_set_tie_policy(mc_ctx* c, int32_t policy);
/* Kernel-selection switches of ONE context (no reference counterpart: the reference has one code path; these exist for A/B
 * measurement and for the tests that pin alternative kernels to each other).  The MC_* environment variables of the same
 * meaning only seed the defaults of contexts created afterwards; two contexts of one process may differ.  Keys:
 *   "chain" (bit mask), "big_tokens", "split_groups", "small_gemm_rows", "split_rows_expert",
 *   "split_rows_sffn", "split_expert", "split_sffn", "temporal_split", "rowchain_split", "gemm_tune", "small_tile_n",
 *   "gemm_wp_grid", "half_min_rows", "gate_small", "route_reg", "route_small", "route_coop", "route_per",
 *   "dbg_delay_us" (tests: holds the second sample group's stream that long in front of every layer tail, so the two-stream
 *   schedule runs far out of phase; results must not change).
 * Results never depend on them beyond fp32 summation order where DESIGN.md says so.  Unknown key -> MC_ERR_ARG.
 * Not while a captured graph exists (mc_ctx_graph_release first). */
int mc_ctx_set_option(mc_ctx* c, const char* key, int64_t value);
int mc_ctx_set_timesteps(mc_ctx* c, const int32_t* t_orig_host, int32_t num_steps, void* stream);
int mc_ctx_set_condition(mc_ctx* c, const float* xf_out_dev, const float* mask_dev, void* stream);
/* control condition (ControlT2MHalf.forward_c + controlnet[0].before_proj, controlnet.py:186-199, 66):
 * c_feat_dev [B, Tc, ctrl_cond_feats] = output of the (step-invariant) condition_pre_encoder, Tc <= frames;
 * NULL disables the branch for this context (forward_test with c=None, controlnet.py:405-413). */
int mc_ctx_set_control(mc_ctx* c, const float* c_feat_dev, int32_t Tc, void* stream);

/* x_t_dev [B,T,C] at schedule index step_index -> out2_dev [2B,T,C] (text half, then uncond half);
 * out2_dev may be NULL (result stays in the context, buffer "out2").
 * stop_after_layers < 0: full model; >= 0: run only that many decoder layers and skip the pose
 * decoder (tests read intermediates through mc_ctx_get_buffer). */
int mc_denoise(mc_ctx* c, const float* x_t_dev, int32_t step_index, float* out2_dev,
               int32_t stop_after_layers, void* stream);
/* denoise + CFG combine + sampler update in one call; x0_dev may be NULL; x_prev_dev may alias x_t_dev */
int mc_sample_step(mc_ctx* c, const float* x_t_dev, int32_t step_index, const mc_step_coefs* coefs,
                   const float* noise_dev, float* x_prev_dev, float* x0_dev, void* stream);

/* The sampler LOOP as one call -- GaussianDiffusion.p_sample_loop / ddim_sample_loop, gaussian_diffusion.py:698-797, 925-1049 (the
 * mode is in coefs[k].mode): runs schedule indices step_indices_host[0 .. num_steps) in order (the reference walks
 * num_timesteps-1 .. 0) with coefs_host[k], x_dev [B,T,C] updated IN PLACE, no return to the host language between steps.
 * The per-step th.randn_like(x) (gaussian_diffusion.py:684 / 847) is
 *   noise_dev != NULL: read from noise_dev [num_steps][B,T,C]  (parity runs on the reference's own seeds), or
 *   noise_dev == NULL: drawn inside the sampler-update kernel -- Philox4x32-10 keyed by `seed`, counter = (element / 4,
 *                      noise_draw0 + k), Box-Muller; mc_op_philox_normal writes the same draws to memory (bit for bit).
 * x0_last_dev (may be NULL) receives the last step's x0 prediction.  Asynchronous on `stream` like every other entry point. */
int mc_sample_loop(mc_ctx* c, float* x_dev, const int32_t* step_indices_host, const mc_step_coefs* coefs_host, int32_t num_steps,
                   const float* noise_dev, uint64_t seed, uint64_t noise_draw0, float* x0_last_dev, void* stream);
/* draw `draw` of the Philox stream keyed by `seed`: out_dev[n] = normals, bits_dev[n] = the raw 32-bit words (either may be NULL) */
int mc_op_philox_normal(float* out_dev, uint32_t* bits_dev, int64_t n, uint64_t seed, uint64_t draw, void* stream);

/* hipGraph replay of mc_sample_step (BASELINE.json configs[4] "hipGraph-captured 50-step DDIM"): ONE graph serves every step
 * of the schedule -- the step index is a device-side integer, the FiLM tables and the sampler coefficients are addressed
 * with it inside the kernels, no host sync or per-step H2D copy (SURVEY.md section 3.1 lists the reference's per-step
 * host syncs).  capture: coefs_host[num_steps] = the mc_step_coefs of every schedule index (num_steps must equal the
 * schedule of mc_ctx_set_timesteps); x_dev [B,T,C] is updated IN PLACE by every replay, noise_dev [B,T,C] is read by
 * every replay (the caller refills it between steps); both pointers are baked into the graph.  `stream` must be a
 * non-default stream.  step: runs schedule index step_index on `stream` (bit-identical to mc_sample_step). */
int mc_ctx_graph_capture(mc_ctx* c, float* x_dev, const float* noise_dev, const mc_step_coefs* coefs_host, int32_t num_steps,
                         void* stream);
int mc_ctx_graph_step(mc_ctx* c, int32_t step_index, void* stream);
int mc_ctx_graph_release(mc_ctx* c);

/* mc_sample_step with the seeding above; x_t_dev is MODIFIED IN PLACE on the seeded elements (the reference writes
 * into `img`) before the denoiser and the sampler update read it */
int mc_sample_step_seeded(mc_ctx* c, float* x_t_dev, int32_t step_index, const mc_step_coefs* coefs,
                          const float* noise_dev, const mc_seed* seed, float* x_prev_dev, float* x0_dev, void* stream);

/* mc_sample_step with the kept region of x0 / of the new sample taken from gt (RePaint) */
int mc_sample_step_inpaint(mc_ctx* c, const float* x_t_dev, int32_t step_index, const mc_step_coefs* coefs,
                           const float* noise_dev, const mc_inpaint* inpaint, float* x_prev_dev, float* x0_dev,
                           void* stream);

/* ---- introspection for tests --------------------------------------------------------- */
/* named context buffers: "h","z","proj","mf","qkv","ys","yt","a" (fp32 rows; refused while a reduced-precision context keeps fp16
 * planes there),"a_tail" (the deferred last FiLM block's fp32 rows),"z2","out2","emb","ss","tf",
 * "idx","gate","comb_w","key","cap_idx","cap_w" (layer selects tf / ss / cap slices) */
int mc_ctx_get_buffer(mc_ctx* c, const char* name, int32_t layer, void** dev_ptr, int64_t* numel);

/* FLOP ledger for the per-kernel roofline (tools/kernel_roofline.py; off by default): while enabled, every launcher books the useful
 * multiply-add work (x 2) of each launch under its kernel's name.  mc_debug_flop_ledger(1) clears and starts, (0) stops;
 * mc_debug_flop_ledger_dump writes "kernel<TAB>calls<TAB>flops" lines into buf (truncated at cap) and returns the size needed. */
int mc_debug_flop_ledger(int32_t enable);
int64_t mc_debug_flop_ledger_dump(char* buf, int64_t cap);

/* op-level entry points (kernel parity tests call these through the same ABI) */
int mc_op_gemm(const float* a_dev, const float* w_dev, const float* bias_dev, const float* res_dev,
               float* c_dev, int32_t M, int32_t N, int32_t K, int32_t ldw, int32_t act, void* stream);
/* C = A W^T + bias + res on the fp16 MFMA (split != 0: hi/lo three-product form); N % 128 == 0, K % 32 == 0; synchronises */
int mc_op_gemm_f16(const float* a_dev, const float* w_dev, const float* bias_dev, const float* res_dev, float* c_dev,
                   int32_t M, int32_t N, int32_t K, int32_t split, void* stream);
/* the folded decoder tail as one op (stmogen.py:505-544 + 757-760 after the CFG combination, motioncraft_amd/csrc/mc_gemm.hip):
 *   C[r] = (wc h[r] + wu h[r + M]) W[0]^T + (wc a[r] + wu a[r + M]) W[1]^T + bias[0] + bias[1]
 * h, a [2 M][K] (conditional rows, then the unconditional ones), W [2][N][K], bias [2][N], C [M][N]; c2_dev: scratch [M][N] (the
 * block-range form writes one partial product per K group, added here by a second launch and by the sampler-update kernel in the step;
 * may be NULL for variant 1); variant 0 = the default kernel choice, 1 = the column-tile form (gemm_tail_k), 2 = the block-range form
 * (gemm_tail2_k; N <= 336) */
int mc_op_gemm_tail(const float* h_dev, const float* a_dev, const float* w_dev, const float* bias_dev, float* c_dev, float* c2_dev,
                    int32_t M, int32_t N, int32_t K, float wc, float wu, int32_t variant, void* stream);
int mc_op_ln_rows(const float* x_dev, int64_t ldx, const float* gamma_dev, const float* beta_dev,
                  const float* add_dev, int32_t add_mod, float* y_dev, int64_t rows, int32_t L, void* stream);
int mc_op_sampler_update(const float* x_t_dev, const float* out_text_dev, const float* out_none_dev,
                         const float* noise_dev, float* x_prev_dev, float* x0_dev, int64_t n,
                         const mc_step_coefs* coefs, void* stream);

/* Result post-processing of the 322-d motion.  pred_dev [B,T,322] normalised; lengths_dev [B]
 * int32 valid frames (NULL = T); mean/std_dev [322] fp64; taps_dev [4][MC_POST_MAXTAP] fp64 = normalised Gaussian
 * taps of the 4 channel groups (body+jaw, hands, trans, expressions) centred at radius[g]; radius[g] < 0 leaves the
 * group unfiltered.  stats_f32 != 0 de-normalises in fp32 like numpy does with float32 mean/std files.
 * Outputs fp64: poses [B,T,165], expressions [B,T,100], trans [B,T,3]; frames >= length are 0. */
#define MC_POST_MAXTAP 129
int mc_postprocess_smplx(const float* pred_dev, const int32_t* lengths_dev, const double* mean_dev,
                         const double* std_dev, const double* taps_dev, const int32_t radius[4], int32_t stats_f32,
                         int32_t B, int32_t T, int32_t C, double* poses_dev, double* expr_dev, double* trans_dev,
                         void* stream);
/* The same over the STITCHED sequence the tools save when several --text / --motion_length intervals are given
 * (tools/visualize.py:216-246: the intervals' valid frames are concatenated FIRST, the Gaussian filter then runs over
 * the whole sequence, so smoothing crosses the seams).  rows_dev int32 [n_frames]: row (b*T + t) of pred_dev [*,322]
 * that stitched frame i shows; filter support is clamped to [0, n_frames).  Outputs [n_frames, 165|100|3] fp64. */
int mc_postprocess_smplx_stitched(const float* pred_dev, const int32_t* rows_dev, int32_t n_frames,
                                  const double* mean_dev, const double* std_dev, const double* taps_dev,
                                  const int32_t radius[4], int32_t stats_f32, int32_t C, double* poses_dev,
                                  double* expr_dev, double* trans_dev, void* stream);

/* ---- text condition encoder (encode_text, diffusion_transformer.py:142-172); run once per prompt batch -------- */
typedef struct mc_textenc mc_textenc;
typedef struct mc_textenc_config {
    int32_t clip_dim;         /* 512: width of the CLIP text features                          */
    int32_t text_latent_dim;  /* text_encoder.latent_dim (256)                                 */
    int32_t num_layers;       /* text_encoder.num_layers (2) of nn.TransformerEncoder          */
    int32_t ff_size;          /* text_encoder.ff_size (2048)                                   */
    int32_t num_heads;        /* text_encoder.num_heads (4); head_dim must be 64               */
    int32_t max_len;          /* 77 tokens                                                     */
    int32_t clip_layers;      /* 12 (0: the CLIP tower is not used, clip_feat is an input)     */
    int32_t clip_heads;       /* 8                                                             */
    int32_t clip_ff;          /* 2048                                                          */
    int32_t vocab;            /* 49408                                                         */
} mc_textenc_config;
int mc_textenc_create(const mc_textenc_config* cfg, mc_textenc** out);
void mc_textenc_destroy(mc_textenc* e);
/* fp32 parameters from host memory under the reference's own state-dict keys (relative to the denoiser):
 * "text_pre_proj.weight/bias", "textTransEncoder.layers.{i}.{self_attn.in_proj_weight,...}", "text_ln.weight/bias",
 * and optionally "clip.token_embedding.weight", "clip.positional_embedding", "clip.transformer.resblocks.{i}.*",
 * "clip.ln_final.*" */
int mc_textenc_set_param(mc_textenc* e, const char* name, const float* host, int64_t numel);
int mc_textenc_finalize(mc_textenc* e);
/* clip_feat_dev [B, max_len, clip_dim] -> xf_out_dev [B, max_len, text_latent_dim]  (encode_text with clip_feat given) */
int mc_textenc_forward_feat(mc_textenc* e, const float* clip_feat_dev, int32_t B, float* xf_out_dev, void* stream);
/* tokens_dev int32 [B, max_len] (clip.tokenize ids) -> xf_out_dev; clip_feat_out_dev may be NULL */
int mc_textenc_forward_tokens(mc_textenc* e, const int32_t* tokens_dev, int32_t B, float* clip_feat_out_dev,
                              float* xf_out_dev, void* stream);

/* ---- Evaluation embedding model (T2MContrastiveModel_SMPLX; mogen/models/rnns/t2m_bigru_smplx.py:396-437) --------
 * motion side: ActorAgnosticEncoder (:66-195); text side: DistilbertActorAgnosticEncoder after tokenisation (:198-394).
 * The embeddings feed FID / R-precision / matching score / diversity / multimodality (mogen/core/evaluation/). */
typedef struct mc_evalenc mc_evalenc;
typedef struct mc_evalenc_config {
    int32_t nfeats;           /* motion_encoder.nfeats (322)                                                  */
    int32_t latent_dim;       /* 256                                                                          */
    int32_t ff_size;          /* 1024                                                                         */
    int32_t num_layers;       /* 4 layers of nn.TransformerEncoder (post-LN, GELU)                            */
    int32_t num_heads;        /* 4; head_dim must be 64                                                       */
    int32_t pe_len;           /* rows of sequence_pos_encoding.pe (5000)                                      */
    int32_t bert_dim;         /* DistilBERT hidden size (768); 0 = motion side only                           */
    int32_t bert_layers;      /* 6                                                                            */
    int32_t bert_heads;       /* 12; head_dim must be 64                                                      */
    int32_t bert_ff;          /* 3072                                                                         */
    int32_t bert_vocab;       /* 30522                                                                        */
    int32_t bert_max_pos;     /* 512                                                                          */
} mc_evalenc_config;
int mc_evalenc_create(const mc_evalenc_config* cfg, mc_evalenc** out);
void mc_evalenc_destroy(mc_evalenc* e);
/* fp32 parameters from host memory under the evaluator checkpoint's own keys ("motionencoder.skel_embedding.weight",
 * "motionencoder.mu_token", "motionencoder.sequence_pos_encoding.pe", "motionencoder.seqTransEncoder.layers.{i}.*",
 * "textencoder.text_model.embeddings.*", "textencoder.text_model.transformer.layer.{i}.*", "textencoder.projection.1.*",
 * "textencoder.mu_token", ... as split by load_pretrained, t2m_bigru_smplx.py:417-435) */
int mc_evalenc_set_param(mc_evalenc* e, const char* name, const float* host, int64_t numel);
int mc_evalenc_finalize(mc_evalenc* e);
/* encode_motion(motion, motion_length).loc: motion_dev [B, T, nfeats], lengths_dev int32 [B] -> mu_out_dev [B, latent_dim] */
int mc_evalenc_encode_motion(mc_evalenc* e, const float* motion_dev, const int32_t* lengths_dev, int32_t B, int32_t T,
                             float* mu_out_dev, void* stream);
/* encode_text(...).loc after the tokenizer: ids_dev int32 [B, S], mask_dev uint8 [B, S] (attention_mask) -> mu_out_dev */
int mc_evalenc_encode_text(mc_evalenc* e, const int32_t* ids_dev, const uint8_t* mask_dev, int32_t B, int32_t S,
                           float* mu_out_dev, void* stream);

/* ---- HumanML3D / KIT evaluation embedding model (T2MContrastiveModel; mogen/models/rnns/t2m_bigru.py:284-299) --------
 * motion side: T2MMotionEncoder = MovementConvEncoder + MotionEncoderBiGRUCo (:72-110, :226-282); text side:
 * TextEncoderBiGRUCo on word vectors + part-of-speech one-hots (:186-223; the GloVe lookup stays with the dataset). */
typedef struct mc_t2meval mc_t2meval;
typedef struct mc_t2meval_config {
    int32_t input_size;       /* motion_encoder.input_size (263 / 251); the last 4 channels are dropped            */
    int32_t movement_hidden;  /* 512                                                                               */
    int32_t movement_latent;  /* 512                                                                               */
    int32_t motion_hidden;    /* 1024                                                                              */
    int32_t motion_latent;    /* 512                                                                               */
    int32_t word_size;        /* 300; 0 = motion side only                                                         */
    int32_t pos_size;         /* 15                                                                                */
    int32_t text_hidden;      /* 512                                                                               */
    int32_t text_out;         /* 512                                                                               */
} mc_t2meval_config;
int mc_t2meval_create(const mc_t2meval_config* cfg, mc_t2meval** out);
void mc_t2meval_destroy(mc_t2meval* e);
/* fp32 parameters from host memory: the checkpoint's three state dicts flattened with their names as prefixes
 * ("movement_encoder.main.0.weight" [O, C, 4] as stored, "motion_encoder.gru.weight_ih_l0_reverse", "text_encoder.hidden", ...) */
int mc_t2meval_set_param(mc_t2meval* e, const char* name, const float* host, int64_t numel);
int mc_t2meval_finalize(mc_t2meval* e);
/* motion_dev [B, T, input_size], lengths_dev int32 [B] (frames, >= 4) -> out_dev [B, motion_latent] */
int mc_t2meval_encode_motion(mc_t2meval* e, const float* motion_dev, const int32_t* lengths_dev, int32_t B, int32_t T,
                             float* out_dev, void* stream);
/* word_emb_dev [B, S, word_size], pos_onehot_dev [B, S, pos_size], sent_len_dev int32 [B] (>= 1) -> out_dev [B, text_out] */
int mc_t2meval_encode_text(mc_t2meval* e, const float* word_emb_dev, const float* pos_onehot_dev, const int32_t* sent_len_dev,
                           int32_t B, int32_t S, float* out_dev, void* stream);

/* ---- WavEncoder: step-invariant audio condition encoder of the speech-to-gesture configs ------------------ */
typedef struct mc_wavenc mc_wavenc;
int mc_wavenc_create(int32_t audio_in, int32_t out_dim, mc_wavenc** out);
void mc_wavenc_destroy(mc_wavenc* e);
/* BatchNorm-folded, tap-major conv weights from host memory: "b{i}.conv1.w" [planes][ceil4(15*cin)], "b{i}.conv1.b",
 * "b{i}.conv2.w" [planes][15*planes], "b{i}.conv2.b", "b{i}.down.w", "b{i}.down.b" (blocks 0,1,3,5), i = 0..5 */
int mc_wavenc_set_param(mc_wavenc* e, const char* name, const float* host, int64_t numel);
int mc_wavenc_finalize(mc_wavenc* e);
int mc_wavenc_out_len(const mc_wavenc* e, int32_t samples, int32_t* frames);
/* wav_dev [B, samples, audio_in] -> out_dev [B, frames, out_dim] (channels-last, like WavEncoder.forward's return) */
int mc_wavenc_forward(mc_wavenc* e, const float* wav_dev, int32_t B, int32_t samples, float* out_dev, void* stream);

/* out = a * x + b * noise over n elements (out may alias x) */
int mc_op_renoise(const float* x_dev, const float* noise_dev, float a, float b, float* out_dev, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MOTIONCRAFT_AMD_H */
