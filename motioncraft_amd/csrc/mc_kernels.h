// Launchers of the non-GEMM kernels (mc_kernels.hip, mc_route.hip, mc_attn.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// CFG twin aliasing in base layer 0: while *split_flag == 0 the rows >= `from` of mf / qkv / ys are bit-identical to
// rows (r - from); producers neither compute nor store them, consumers read the twin's.  flag == nullptr: off.
struct TwinAlias {
    const int* split_flag = nullptr;
    long from = 0;
};

// ---- mc_kernels.hip -------------------------------------------------------------------
// Y[r][0:L] = LN_L(X[r*ldx + x_col : +L]) * gamma + beta (+ add[(r % add_mod)*L + c])
int mc_launch_ln_rows(const float* X, long ldx, int x_col, const float* gamma, const float* beta,
                      const float* add, int add_mod, float* Y, long ldy, long rows, int L, hipStream_t s);
// A[r][0:D] = silu( LN_D(Y1[r] (+ Y2[r])) * gamma + beta ) * (1 + ss[0:D]) + ss[D:2D] ) -- StylizationBlock prologue
// y1_alias: Y1 rows (global row index row0 + r) >= from are read from row - from while the flag is clear
// Step index read on the device (hipGraph replay, mc_ctx_graph_*): while `ptr` is set, per-step tables are addressed with
// *ptr inside the kernel instead of through host-computed pointers / by-value scalars, so ONE captured graph serves all steps.
struct StepRef {
    const int* ptr = nullptr;
    long stride = 0;             // floats per step of the table behind the pointer argument
};
int mc_launch_film_rows(const float* Y1, const float* Y2, const float* gamma, const float* beta,
                        const float* ss, float* A, long rows, int D, hipStream_t s, TwinAlias y1_alias = TwinAlias(), long row0 = 0,
                        StepRef step = StepRef(), int y1_parts = 1, long y1_pstride = 0,    // y1_parts > 1: Y1 = sum of that many partial planes
                        int planes = 0, long plane_stride = 0);   // planes 1 / 2: A is written as fp16 plane(s) [rows][D] halves (hi | lo plane_stride halves behind) for gemm_hd_k; + 4: FRAGMENT-MAJOR planes for gemm_hf_k (rows % 32 == 0)
// te[s][0:D] = cat(cos(t_s f), sin(t_s f))
int mc_launch_timestep_embedding(const int* t_orig, float* te, int S, int D, hipStream_t s);
int mc_launch_silu(const float* X, float* Y, long n, hipStream_t s);
// out = a + b + bias (each optional), rows x D
int mc_launch_add_rows(float* out, const float* a, const float* b, const float* bias, long rows, int D, hipStream_t s);
// out = softmax(W[H][H], dim=1)
int mc_launch_softmax_rows_small(const float* W, float* out, int rows, int cols, hipStream_t s);

struct SamplerCoefs {
    int mode;          // 0 ddpm, 1 ddim
    float text_coef, none_coef;
    float c1, c2;      // ddpm: posterior_mean_coef1/2
    float log_var;     // ddpm: model_log_variance (FIXED_LARGE)
    float sqrt_recip, sqrt_recipm1, ab, ab_prev, eta;   // ddim
    float nonzero;     // 0 at i == 0
};
// x_prev = sampler(x_t, x0 = text_coef*out_text + none_coef*out_none, noise); optionally also writes x0
// table != nullptr (graph replay): the step's coefficients are table[*step_ptr]; only text_coef / none_coef of `c` are used
// rng (with noise == nullptr): the noise is drawn inside the kernel, Philox4x32-10 keyed by the seed, counter (element / 4, draw index)
struct RngArgs { uint32_t seed_lo = 0, seed_hi = 0, draw_lo = 0, draw_hi = 0; };
// pad (mc_sample_loop): x_prev is also written as rows of Cp floats into xpad (the next step's pose-encoder GEMM operand; the pad
// columns are not touched)
struct PadOut { float* xpad = nullptr; int C = 0, Cp = 0; };
int mc_launch_sampler_update(const float* x_t, const float* out_text, const float* out_none,
                             const float* noise, float* x_prev, float* x0_out, long n,
                             SamplerCoefs c, hipStream_t s, const SamplerCoefs* table = nullptr, const int* step_ptr = nullptr,
                             const RngArgs* rng = nullptr, const PadOut* pad = nullptr);
// out[n] = the normals / bits[n] = the raw 32-bit words of draw `rng` (either may be null)
int mc_launch_philox_fill(float* out, uint32_t* bits, long n, RngArgs rng, hipStream_t s);
// out = table[*step].text_coef * x + table[*step].none_coef * y   (CFG combine of the two halves under graph replay)
int mc_launch_cfg_combine_tab(const float* x, const float* y, const SamplerCoefs* table, const int* step_ptr, float* out, long n,
                              hipStream_t s);
int mc_launch_set_int(int* dst, int value, hipStream_t s);
int mc_launch_spin(long ticks, hipStream_t s);      // tests: hold stream s for `ticks` of the 100 MHz wall clock

// RePaint / outpainting (gaussian_diffusion.py:492-501, 855-877)
struct InpaintArgs {
    const float* gt;            // [B][T][C]
    const uint8_t* keep;        // [B][T][C] bool outpainting_mask
    const float* gt_noise;      // [B][T][C] second randn_like of ddim_sample (DDIM only)
    const float* blend_w;       // [blend_len] linspace(0, 1, overlap_len)
    int blend_len;              // 0 = no cross-fade at this step
    int T, C;
};
int mc_launch_sampler_inpaint(const float* x_t, const float* out_text, const float* out_none, const float* noise,
                              InpaintArgs ip, float* x_prev, float* x0_out, long n, SamplerCoefs c, hipStream_t s);
// Y[r][0:Cp] = X[r][0:C], zero padded (aligned rows for the pose-encoder GEMM)
int mc_launch_pad_rows(const float* X, float* Y, long rows, int C, int Cp, hipStream_t s);
// pre_seq / transl_req seeding of p_sample / ddim_sample (gaussian_diffusion.py:664-674, 816-820), fused into the pad pass:
// X[b][t < pre_len][:] = sqrt_ab * pre[b][t][:] + sqrt_1mab * pre_noise[b][t][:], then X[b][t < 2][ch_k] = transl_value[k][t],
// written back to X IN PLACE (the sampler update reads x_t again) and to the padded copy
struct SeedArgs {
    const float* pre = nullptr;        // [B][pre_len][C]
    const float* pre_noise = nullptr;  // [B][pre_len][C]
    int pre_len = 0, T = 0;
    float sqrt_ab = 0.f, sqrt_1mab = 0.f;
    int num_transl = 0;
    int transl_channel[8];
    float transl_value[8][2];
};
int mc_launch_pad_rows_seeded(float* X, float* Y, long rows, int C, int Cp, const SeedArgs& sd, hipStream_t s);
// out[M][N] (contiguous) = sum_s part[s][M][N] + bias[N] + res[M][N]
int mc_launch_splitk_reduce(const float* part, int S, long M, int N, const float* bias, const float* res, float* out,
                            hipStream_t s, const int* dyn_tiles = nullptr);   // dyn_tiles: S = mc_mlp_dyn_ways(*dyn_tiles) on the device
int mc_launch_axpby(const float* x, const float* y, float a, float b, float* out, long n, hipStream_t s);
// out0 = a x0 + b y0 and out1 = a x1 + b y1 in one launch; table != null: (a, b) = table[*step_ptr].{text_coef, none_coef}
int mc_launch_axpby_pair(const float* x0, const float* y0, float* out0, const float* x1, const float* y1, float* out1, float a, float b,
                         const SamplerCoefs* table, const int* step_ptr, long n, hipStream_t s);

// ---- mc_post.hip ----------------------------------------------------------------------
// de-normalise + 322 -> (poses 165, expressions 100, trans 3) + Gaussian temporal filter (tools/visualize.py:217-246)
int mc_launch_smplx_post(const float* pred, const int* lengths, const int* rows, const double* mean, const double* stdv,
                         const double* taps, const int* radius, int stats_f32, int B, int T, int C,
                         double* poses, double* expr, double* trans, hipStream_t s);
int mc_smplx_post_maxtap();

// ---- mc_route.hip ---------------------------------------------------------------------
struct RouteBufs {
    // per (token, choice)
    int* idx;          // [N][2] expert index
    float* gate;       // [N][2] normalised gate
    uint32_t* key;     // [N]    float bits of max score (importance)
    float* comb_w;     // [N][2] gate if kept else 0
    int* src_row;      // [2N]   slot -> token
    int* dst_row;      // [2N]   slot -> 2*token + choice
    // tile map
    int* tile_group; int* tile_row0; int* tile_nrows;   // [2][max_tiles] (one map per slot group)
    int* state;        // small int block, layout in mc_route.hip
    int max_tiles;
    uint32_t tie_xor = 0xFFFFFFFFu;   // order of equal-importance tokens at the capacity cut: ~0 = lower index first (stable), 0 = higher first
    bool coop = true;                 // larger batches: the cooperative one-launch routing kernel (false: the 12-launch sequence; env MC_ROUTE_COOP=0, tests)
    int coop_per = 0;                 // route_coop_k: (token, choice) pairs per thread, 10 or 16 (0: 10 up to 256 workgroups, 16 beyond; option "route_per")
    long small_pairs = -1;            // >= 0: overrides MC_ROUTE_SMALL for this context (tests force the large-batch paths on small configs)
    bool reg_kernel = true;           // small batches: the register-resident one-workgroup routing kernel (false: the L2-streaming form at every size; env MC_ROUTE_REG=0, tests)
};
size_t mc_route_state_ints(int E);
bool mc_route_is_small(long N);
bool mc_route_cleans_counts(const RouteBufs& rb, long N);   // the routing kernel that will run for N tokens hands the (choice, expert) counts back zeroed
size_t mc_route_barrier_offset();                          // ints into the state block: grid-barrier words, to be zeroed once
size_t mc_route_error_offset();                             // ints into the state block: sticky "grid barrier timed out" word of route_coop_k
int mc_route_coop_wgs(long N);                              // workgroups route_coop_k launches for N tokens (0: beyond its size range)
int mc_route_coop_slots(int dev);                           // route_coop_k workgroups device `dev` (the current one at the first query) holds at once (occupancy x CUs)
bool mc_route_coop_reserve(int dev, int nwg);                        // reserve / give back resident workgroups for one context (per device, process-wide)
void mc_route_coop_release(int dev, int nwg);
size_t mc_route_barrier_ints();   // routing of N tokens runs as the one-workgroup kernel, which also leaves the (choice, expert) counts zeroed
// proj [N][256] (cosine_projector output incl. bias) -> idx/gate/key + per-expert choice counts
int mc_launch_gate_finish(const float* proj, const float* sim_n, const float* logit_scale, long N, int E,
                          RouteBufs rb, hipStream_t s);
// capacity/BPR drop decision + slot compaction + tile map (tile rows = 128)
// Nsrc = N, or N/2 (twin mode); tokens >= gsplit get their own slot group (tile map at [max_tiles, 2 max_tiles))
int mc_launch_route(long N, long Nsrc, long gsplit, int E, int capacity, RouteBufs rb, hipStream_t s);
const int* mc_route_num_tiles_ptr(const RouteBufs& rb, int group = 0);
// twin mode: 1 after routing iff some second-half token was kept/dropped differently from its first-half twin
const int* mc_route_split_flag_ptr(const RouteBufs& rb);

// ---- mc_attn.hip ----------------------------------------------------------------------
// static + dynamic body topology: ys[(b,t)][h*L + c]
int mc_launch_body(const float* mf, long ldmf, const float* qkv, const float* wsm, float* ys,
                   long frames, int H, int L, int G, hipStream_t s, TwinAlias alias = TwinAlias(), long frame0 = 0);
// temporal linear attention over text (+) motion tokens: yt[(b,t)][h*L + c]
// samples [b0, b0 + nb) of the CFG-doubled batch (B = samples per CFG half)
int mc_launch_temporal(const float* mf, const float* tf, const float* mask, float* yt,
                       int b0, int nb, int B, int T, int Nt, int H, int L, hipStream_t s, const int* twin_flag = nullptr,
                       long lsplit_max = 96,     // up to this many (sample, part) workgroups the output columns are sliced over blockIdx.y
                       bool pair = false,        // L = 64: two parts per workgroup
                       bool skip_text = false);  // the unconditional half skips whole leading blocks of its (masked, zero-valued) text rows: the same bits       // L = 64, even H, beyond lsplit_max: two adjacent parts per workgroup
