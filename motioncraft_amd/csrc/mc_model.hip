// Host side of libmotioncraft_amd.so: model/context handles, workspace layout in HBM and the
// per-step kernel schedule of the STMoGen denoiser (C-ABI declared in include/motioncraft_amd.h).
//
// HBM layout (all fp32, row-major; B2 = 2B CFG-doubled batch, N = B2*T*H tokens, rows = B2*T):
//   h    [rows, D]      residual stream (text-conditioned half first, stmogen.py:736-744)
//   z    [N, L]         LN(x) + motion_moe.embedding            proj [N, 256] cosine projector out
//   hbuf [2N, 4L]       expert hidden, slot-major               y2   [N, 2, L] expert outputs per choice
//   mf   [N, 4L]        [body_value | key | value | query]      qkv  [N, 3L]   dynamic-topology q,k,v
//   ys/yt/a/z2 [rows,D] static+dynamic / temporal / FiLM prologue / SFFN output
//   fh   [rows, H*F]    SFFN hidden                             out2 [rows, C]  pose decoder output
//   tf[layer] [B2*Nt, 2L]  step-invariant text K/V              ss[layer][blk][S][2D] FiLM scale|shift
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/motioncraft_amd.h"
#include "mc_common.h"
#include "mc_gemm.h"
#include "mc_kernels.h"
#include "mc_chain.h"
#include "mc_half.h"

struct HalfW {              // fp16 hi / lo planes of one weight (mc_half.h); lo directly behind hi in one allocation
    mc_half* hi = nullptr;
    mc_half* lo = nullptr;
};

struct mc_model {
    mc_model_config cfg;
    std::map<std::string, std::pair<float*, int64_t>> params;
    std::map<std::string, HalfW> half;     // built on the first mc_ctx_set_precision(.., MC_PREC_F16*) (shared by all contexts)
    std::mutex half_mu;                    // guards `half` (filled from per-context calls)
    size_t half_bytes = 0;                 // device bytes of the fp16 hi / lo planes (reported by mc_ctx_workspace_bytes of a reduced-precision context)
    bool finalized = false;
    int Cp = 0;  // input_feats padded to a multiple of 32 (row stride of enc.w and of the padded pose rows)
};

struct MoeW {
    const float *emb, *gate_w, *gate_b, *sim_n, *sim_nT, *scale, *fc1_w, *fc1_b, *fc2_wt, *fc2_b, *proj_w, *proj_b;
    int din, dout;
};

struct LayerW {
    const float *norm_g, *norm_b, *tnorm_g, *tnorm_b, *wsm;
    MoeW mm, tm;
    const float *dyn_g, *dyn_b, *qkv_w, *qkv_b;
    const float *ca_film_w, *ca_film_b, *ca_ln_g, *ca_ln_b, *ca_out_w, *ca_out_b;
    const float *ffn_w1, *ffn_b1, *ffn_w2, *ffn_b2;
    const float *ffn_film_w, *ffn_film_b, *ffn_ln_g, *ffn_ln_b, *ffn_out_w, *ffn_out_b;
    // control-branch copies only (ControlT2MBlock): zero-init projections around the copied DecoderLayer
    const float *before_w = nullptr, *before_b = nullptr, *after_w = nullptr, *after_b = nullptr;
    // fp16 planes of the per-step GEMM weights (reduced-precision mode; null until mc_ctx_set_precision builds them)
    HalfW h_ca_out, h_ffn_out, h_fc1, h_fc2, h_w1, h_w2, h_after, h_proj, h_qkv;
};

// Per-context switches (mc_ctx_set_option).  Defaults = the measured best; a variable of the same meaning in the environment
// (MC_CHAIN, MC_SPLIT, MC_GEMM_TUNE, ...) overrides the default of contexts created afterwards, mc_ctx_set_option overrides both for
// one context -- two models of one process can run different kernel selections.
struct McOptions {
    // chain bits: 0 fused expert / SFFN MLP (mlp2_k), 1 fused gate (gate_k), 2 chained proj / qkv (rowchain_k), 3 temporal branch on the
    // side stream at any batch size, 4 CFG twin dedupe in base layer 0, 5 two sample groups on two streams (large batches), 6 one
    // expert-MLP launch per sample group, 7 last FiLM Linear + pose decoder on the CFG-combined rows (folded), 8 twin aliasing of the
    // mf / qkv / ys rows in base layer 0, 9 the sample groups stay on their streams across the control-branch ops between layers,
    // 10 proj + body LN + q/k/v in one kernel (large batches), 11 folded decoder tail as one grouped GEMM + sum in the sampler kernel,
    // 12 small batches: the SFFN's split-hidden partial sums are added up by the FiLM row kernel, 13 B=1 sizes: the expert MLP picks
    // 3 or 4 hidden slices on the device, 14 small batches: temporal branch on the main stream, LN + q/k/v + body on the side stream,
    // 15 (round 4) pqbody_k: bit 10's kernel also runs the body-topology attention (q/k/v never in HBM; L = 128, 12 parts, fp32),
    // 16 (round 4) the twin layer's gate / experts / front kernels run as two sample sub-groups on the two streams
    // 17 (round 4) reduced-precision contexts: film_rows_k writes the FiLM GEMM's A operand as fp16 planes, gemm_hd_k reads them by LDS-DMA
    // 18 (round 4) the fused expert / SFFN MLPs (fp32 and fp16, L = 128 / 64) stage their weight chunks by LDS-DMA (mlp2d_k / mlp2hd_k; same bits)
    // 19 (round 4) mc_sample_loop: the sampler update also writes x_{t-1} at the padded stride of the next step's pose-encoder GEMM
    // 20 (round 4) reduced-precision contexts: temporal linear attention on the fp16 MFMA (temporal_h_k)
    // 22 (round 4) L = 64 models: temporal_k takes two adjacent parts per workgroup (all four waves on the MFMA)
    // 21 (round 4) large batches: the folded decoder tail with the CFG combination in its A staging, one pass over both K groups (gemm_tail_k)
    // 23 (round 5) two-stream schedule: the last sample group's gate launch is cut at a whole number of workgroup rounds; the partial last round
    //    runs as gate_small_k (32-token workgroups, the same bits) on the OTHER group's stream, beside the big launch instead of behind it.
    //    OFF by default: measured SLOWER (B=64 19.47 -> 19.55 ms/step, B=32 10.10 -> 10.14, same-box A/B twice): the 608 small workgroups
    //    (each wave re-reads its projector chunks from L2, wave 0 walks the logit chain alone) take longer than the partial round they replace
    // 24 (round 5) temporal_k: the unconditional CFG half skips whole leading blocks of its text rows (keys at -1e6, values x 0: exact zeros
    //    as long as the sample has a valid frame) -- the same bits, ~20 % less of that half's kernel
    // 25 (round 5) twin layer: the fused front of a sample sub-group covers that sub-group's aliased twins in the same launch (pqbody_k's second
    //    token range) instead of a launch of its own behind the cross-join
    //    OFF by default: the same bits, but measured SLOWER (B=64 19.20 -> 19.30, 19.25 -> 19.36 ms/step): the 125 us by which the separate no-op
    //    launch held the second group's stream back were doing useful work -- they kept the two chains out of phase (see bit 26)
    // 26 (round 5) the second sample group's first FiLM block (proj_out) starts behind the first group's FiLM ROW kernel (one event per layer):
    //    the two HBM-bound row kernels never run against each other and the groups' GEMM / SFFN launches leave the seam half a kernel apart
    //    instead of in lockstep (B=64 19.08 -> 18.94, 19.14 -> 19.01, 19.12 -> 19.00 ms/step; a 60 us spin at the same place: the same).
    //    Only in exact-fp32 contexts at L = 128: measured slower in the fp16 modes (f16 6.86 -> 7.08) and at L = 64 (M2D 17.62 -> 18.00), neutral at batch 32
    // 27 (round 6) reduced-precision contexts: the FiLM plane GEMM requests the residual rows of its epilogue at the TOP of the tile, into registers of their own
    //    (gemm_hd_k<., false, true>: in flight during the DMA prologue; no load -> add -> store chain at the end of every tile) -- the same order (sum + bias) + R,
    //    the same bits (tools/gemm_h6_lab.hip: the kernel runs at the package power cap, this chain is the part that moves).  Same-box A/B: f16 B=64 6.87 -> 6.82,
    //    B=32 3.98 -> 3.81 ms/step; f16x3 10.72 -> 10.62, 5.52 -> 5.45
    // 28 (round 6, OFF) plain f16 only: the accumulators START as R + bias instead (stores-only epilogue; f16 B=32 3.75 against bit 27's 3.81) -- another fp32
    //    summation order (2.6e-4 on h after one layer, 1.1e-3 on x0 against the exact order: inside plain f16's own error, but not free): a switch, not the default
    // 29 (round 6) reduced-precision contexts: the FiLM operand planes are written FRAGMENT-MAJOR and the plane GEMM reads its A fragments straight into registers
    //    (gemm_hf_k: only W rides the LDS-DMA ring; tools/gemm_h6_lab.hip ha_k); launches whose rows are whole 32-row blocks only; the same bits as gemm_hd_k.
    //    film_rows_k stages the 4 rows of a workgroup in LDS and writes 64-byte runs, the 8 workgroups of a 32-row block share an XCD.  Serial schedule: the GEMM
    //    198 -> 183 us (B=64 f16), the row kernel 74.7 -> 77.2; two-stream step, same-box A/B: f16x3 10.85 -> 10.56 (B=64), 5.50 -> 5.37 (B=32); f16 neutral (7.05 / 7.06, 3.76 / 3.75)
    int chain = 65527 | (1 << 16) | (1 << 17) | (1 << 18) | (1 << 19) | (1 << 20) | (1 << 21) | (1 << 22) | (1 << 24) | (1 << 26) | (1 << 27) | (1 << 29);     // (all but bits 3, 23, 25 and 28)
    long small_gemm_rows = 5600;       // plain GEMMs of up to this many rows take the small-M kernels (round 4: 6400 -> 5600, measured per batch: at
                                       // 6272 rows -- a sample group of 32 x 196 frames -- gemm_wp_k / gemm_tail_k now win: B=32 step 10.21 -> 10.03 ms,
                                       // S2G at 32 per GPU 27.65 -> 27.14; at 4704 rows (B=24) the small kernels still do, 7.85 vs 7.91)
    long split_rows_expert = 2048, split_rows_sffn = 8192;      // residual rows up to which the fused MLPs split their hidden dimension
    long temporal_split = 96;          // (sample, part) workgroups up to which temporal_k slices its output columns
    long big_tokens = 65536;           // above this many motion tokens: the large-batch schedule (two sample groups on two streams, projqkv / pqbody)
    long rowchain_split = 20480;       // tokens up to which rowchain_k slices its output chunks over blockIdx.y
    int gemm_tune = -1, small_tile_n = 0, gemm_wp_grid = 0;      // GemmArgs::tune / small_tile_n / wp_grid (-1 / 0: library defaults)
};
static McOptions default_options() {
    McOptions o;
    if (const char* e = getenv("MC_CHAIN")) o.chain = atoi(e);
    if (const char* e = getenv("MC_SMALL_GEMM_ROWS")) o.small_gemm_rows = atol(e);
    if (const char* e = getenv("MC_SPLIT_ROWS_EXPERT")) o.split_rows_expert = atol(e);
    if (const char* e = getenv("MC_SPLIT_ROWS_SFFN")) o.split_rows_sffn = atol(e);
    if (const char* e = getenv("MC_TEMPORAL_SPLIT")) o.temporal_split = atol(e);
    if (const char* e = getenv("MC_BIG_TOKENS")) o.big_tokens = atol(e);
    if (const char* e = getenv("MC_ROWCHAIN_SPLIT")) o.rowchain_split = atol(e);
    if (const char* e = getenv("MC_GEMM_TUNE")) o.gemm_tune = atoi(e);
    if (const char* e = getenv("MC_SMALL_TILE_N")) o.small_tile_n = atoi(e);
    if (const char* e = getenv("MC_GEMM_WP_GRID")) o.gemm_wp_grid = atoi(e);
    return o;
}

struct ProfRec { hipEvent_t e0 = nullptr, e1 = nullptr; long rows = 0; };

struct mc_ctx {
    mc_model* m = nullptr;
    McOptions opt = default_options();
    int B = 0, T = 0, S = 0, maxS = 0;
    long N = 0, rows = 0, Ntxt = 0;
    std::vector<LayerW> lw;
    const float *enc_w, *enc_b, *seq_emb, *time_w0, *time_b0, *time_w2, *time_b2, *dec_w, *dec_b;
    const float *dec_wf = nullptr, *dec_bf = nullptr;   // decoder folded with the last StylizationBlock Linear (optional)
    float *dec_cat_w = nullptr, *dec_cat_b = nullptr;   // [2][C][D] = dec_w | dec_wf and [2][C] = dec_bf | 0: the folded tail as ONE grouped GEMM
    const float *ctrl_in_w = nullptr, *ctrl_in_b = nullptr;
    int NLA = 0;                 // base + control layers (weights, text K/V and FiLM tables are per layer slot)
    float *hc = nullptr, *cb = nullptr, *cenc = nullptr;   // control stream, before_proj(c) [rows,D], forward_c(c) [B*T,D]
    bool have_ctrl = false;
    // workspace
    std::vector<void*> allocs;
    int64_t bytes = 0;
    float *h, *z, *proj, *hbuf, *y2, *mf, *qkv, *ys, *yt, *a, *z2, *fh, *out2, *xpad;
    // reduced-precision contexts: the fp32 rows of the DEFERRED last FiLM block (defer_last_gemm) get their own array -- `a` holds the
    // fp16 hi | lo planes of BOTH sample groups there, and fp32 rows written into it by one group's stream would land on the plane rows
    // the other group's gemm_hd_k may still be reading; null in fp32 contexts (then `a` itself holds fp32 rows only)
    float* a_tail = nullptr;
    long dbg_delay_us = 0;       // tests (option "dbg_delay_us"): hold the second (> 0) or the first (< 0) sample group's stream this long in front of every layer tail, so the groups run far out of phase
    size_t hbuf_cap = 0;        // floats allocated behind hbuf
    size_t hbuf_floats = 0;     // > 0: hbuf is free scratch (fused expert path), used for split-K partial sums
    bool cnt_clean = false;     // the routing state's (choice, expert) counts are known to be zero on the stream (route_small_k cleans up after itself)
    long led_nsrc = 0, led_gsplit = 0;      // FLOP ledger only: routed source tokens of the last routing and its slot-group boundary
    float *xfn, *tf;          // tf: [NL][B2*Nt][2L]
    const float* mask = nullptr;   // = mask_own after mc_ctx_set_condition (a private copy: the pointer is baked into captured graphs)
    float* mask_own = nullptr;
    int* t_orig;
    float *te, *e1, *emb, *semb, *ss;   // ss: [NL][2][maxS][2D]
    RouteBufs rb;
    int device = 0;             // the HIP device the context was created on (reservations are per device)
    int coop_reserved = 0;      // resident route_coop_k workgroups this context holds (mc_route_coop_reserve)
    bool prof_on = false;       // mc_ctx_profile: HIP events around the FiLM out_layers GEMM launches (bench.py's dominant-kernel figure)
    std::vector<ProfRec> prof;
    bool have_cond = false;
    // side stream: the temporal branch of STMA needs only the motion-MoE output, so it runs beside
    // (LN + qkv -> body attention) of the same layer (fork after the MoE projection, join before proj_out)
    hipStream_t side = nullptr;
    bool xpad_ready = false;           // mc_sample_loop: xpad already holds this step's padded x_t (written by the previous sampler update)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_gate = nullptr;      // chain bit 23: the last group's rows are ready (recorded on its stream in front of its gate launch)
    // large batches: the batch is cut into `nparts` groups of whole samples, group k > 0 runs on parts[k-1]
    bool no_alias = false;          // introspection runs (stop_after_layers < num_layers): every intermediate row is materialised
    bool defer_last_gemm = false;   // sampler entry points: the last FiLM GEMM runs on the CFG-combined rows (see denoise_combined)
    hipStream_t parts[3] = {nullptr, nullptr, nullptr};
    // Two HIP streams only overlap when the runtime maps them to different HARDWARE queues (GPU_MAX_HW_QUEUES, default 4, handed out round-robin as streams
    // are created): with an RCCL process group initialised in the host process the caller's stream and the one side stream of round 5 landed on the SAME
    // queue and the two-stream schedule ran serially (+7 % step time at B=64: 19.05 -> 20.47 ms, profiles/r06_hw_queue_collision.txt).  The context creates
    // SIDE_CAND candidates back to back (they cover consecutive queues) and times, once per caller stream, which of them really runs beside it.
    static constexpr int SIDE_CAND = 4;
    hipStream_t side_cand[SIDE_CAND] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t side_for = nullptr;         // the caller stream the current `side` was picked for
    bool side_picked = false;
    float side_probe_ms[SIDE_CAND] = {0.f, 0.f, 0.f, 0.f};
    // earlier answers (a host that alternates between a few caller streams must not re-probe -- and re-synchronise -- on every call)
    static constexpr int SIDE_MEMO = 4;
    hipStream_t side_memo_for[SIDE_MEMO] = {nullptr, nullptr, nullptr, nullptr};
    int side_memo_pick[SIDE_MEMO] = {-1, -1, -1, -1};
    int side_memo_n = 0;
    hipEvent_t ev_parts[3] = {nullptr, nullptr, nullptr};
    int nparts = 2;
    // hipGraph replay of the sampler step (mc_ctx_graph_capture / _step): ONE graph for all steps of the schedule; the step
    // index lives in device memory (gstep) and the per-step tables are addressed inside the kernels (StepRef)
    int* gstep = nullptr;
    SamplerCoefs* gcoefs = nullptr;     // [S] schedule coefficients of every step
    bool graph_mode = false;            // set while capturing: launch with device-indexed step parameters
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    int graph_steps = 0;
    int prec = MC_PREC_F32;      // MFMA operand precision of the per-step GEMM-shaped kernels (mc_ctx_set_precision)
    long half_min_rows = 512;    // env MC_HALF_MIN_ROWS at context creation: see use_half()
    int split_sffn = 0;          // env MC_SPLIT_SFFN at context creation: hidden-dimension split of the SFFN (0 = load model)
    int split_expert = 0;        // env MC_SPLIT_EXPERT at context creation: hidden-dimension split of the small-batch expert MLP (0 = load model)
    long gate_small_tokens = 12000;   // env MC_GATE_SMALL at context creation: up to this many tokens the gate runs as gate_small_k
    int* cap_idx = nullptr;      // [NL][2N] routing capture (tests): expert ids ...
    float* cap_w = nullptr;      // ... and combine weights (0 = dropped) of every layer
};

namespace {

template <class T>
int ws_alloc(mc_ctx* c, T** p, size_t n) {
    void* d = nullptr;
    size_t bytes = (n * sizeof(T) + 255) & ~(size_t)255;
    if (bytes == 0) bytes = 256;
    MC_HIP(hipMalloc(&d, bytes));
    c->allocs.push_back(d);
    c->bytes += (int64_t)bytes;
    *p = (T*)d;
    return MC_OK;
}

int get_param(mc_model* m, const std::string& name, int64_t numel, const float** out) {
    auto it = m->params.find(name);
    if (it == m->params.end()) {
        mc_set_error("missing parameter '%s'", name.c_str());
        return MC_ERR_STATE;
    }
    if (it->second.second != numel) {
        mc_set_error("parameter '%s' has %ld elements, expected %ld", name.c_str(), (long)it->second.second, (long)numel);
        return MC_ERR_STATE;
    }
    *out = it->second.first;
    return MC_OK;
}

#define GP(ptr, name, numel)                                         \
    do {                                                             \
        int _r = get_param(m, (name), (int64_t)(numel), &(ptr));     \
        if (_r != MC_OK) return _r;                                  \
    } while (0)

int bind_moe(mc_model* m, const std::string& pre, int din, int dout, int seq_rows, MoeW* w) {
    const int E = m->cfg.num_experts;
    w->din = din;
    w->dout = dout;
    GP(w->emb, pre + "emb", (int64_t)seq_rows * din);
    GP(w->gate_w, pre + "gate_w", 256 * din);
    GP(w->gate_b, pre + "gate_b", 256);
    GP(w->sim_n, pre + "sim_n", 256 * E);
    GP(w->sim_nT, pre + "sim_nT", 32 * 256);
    GP(w->scale, pre + "scale", 1);
    GP(w->fc1_w, pre + "fc1_w", (int64_t)E * 4 * din * din);
    GP(w->fc1_b, pre + "fc1_b", (int64_t)E * 4 * din);
    GP(w->fc2_wt, pre + "fc2_wt", (int64_t)E * din * 4 * din);
    GP(w->fc2_b, pre + "fc2_b", (int64_t)E * din);
    GP(w->proj_w, pre + "proj_w", (int64_t)dout * din);
    GP(w->proj_b, pre + "proj_b", dout);
    return MC_OK;
}

int bind_weights(mc_ctx* c) {
    mc_model* m = c->m;
    const mc_model_config& g = m->cfg;
    const int L = g.latent_dim, H = g.num_parts, D = L * H, F = g.ffn_dim, Te = g.time_embed_dim;
    GP(c->enc_w, "enc.w", (int64_t)D * m->Cp);
    GP(c->enc_b, "enc.b", D);
    GP(c->seq_emb, "seq_emb", (int64_t)g.max_seq_len * D);
    GP(c->time_w0, "time.w0", (int64_t)Te * D);
    GP(c->time_b0, "time.b0", Te);
    GP(c->time_w2, "time.w2", (int64_t)Te * Te);
    GP(c->time_b2, "time.b2", Te);
    GP(c->dec_w, "dec.w", (int64_t)g.input_feats * D);
    GP(c->dec_b, "dec.b", g.input_feats);
    if (m->params.count("dec.wf") && m->params.count("dec.bf")) {
        GP(c->dec_wf, "dec.wf", (int64_t)g.input_feats * D);
        GP(c->dec_bf, "dec.bf", g.input_feats);
    }
    c->NLA = g.num_layers + g.num_ctrl_layers;
    c->lw.resize(c->NLA);
    if (g.num_ctrl_layers > 0) {
        GP(c->ctrl_in_w, "ctrl_in.w", (int64_t)D * ((g.ctrl_cond_feats + 3) / 4 * 4));
        GP(c->ctrl_in_b, "ctrl_in.b", D);
    }
    for (int i = 0; i < c->NLA; ++i) {
        LayerW& w = c->lw[i];
        const bool is_ctrl = i >= g.num_layers;
        const std::string p = (is_ctrl ? "c" + std::to_string(i - g.num_layers) : "l" + std::to_string(i)) + ".";
        if (is_ctrl) {
            if (i == g.num_layers) {
                GP(w.before_w, p + "before_w", (int64_t)D * D);
                GP(w.before_b, p + "before_b", D);
            }
            GP(w.after_w, p + "after_w", (int64_t)D * D);
            GP(w.after_b, p + "after_b", D);
        }
        GP(w.norm_g, p + "norm.g", L);
        GP(w.norm_b, p + "norm.b", L);
        GP(w.tnorm_g, p + "text_norm.g", g.text_latent_dim);
        GP(w.tnorm_b, p + "text_norm.b", g.text_latent_dim);
        GP(w.wsm, p + "body_wsm", H * H);
        int r = bind_moe(m, p + "mm.", L, 4 * L, g.max_seq_len * H, &w.mm);
        if (r != MC_OK) return r;
        r = bind_moe(m, p + "tm.", g.text_latent_dim, 2 * L, g.max_text_len, &w.tm);
        if (r != MC_OK) return r;
        GP(w.dyn_g, p + "dyn.norm.g", L);
        GP(w.dyn_b, p + "dyn.norm.b", L);
        GP(w.qkv_w, p + "dyn.qkv_w", 3 * L * L);
        GP(w.qkv_b, p + "dyn.qkv_b", 3 * L);
        GP(w.ca_film_w, p + "ca.film_w", (int64_t)2 * D * Te);
        GP(w.ca_film_b, p + "ca.film_b", 2 * D);
        GP(w.ca_ln_g, p + "ca.ln_g", D);
        GP(w.ca_ln_b, p + "ca.ln_b", D);
        GP(w.ca_out_w, p + "ca.out_w", (int64_t)D * D);
        GP(w.ca_out_b, p + "ca.out_b", D);
        GP(w.ffn_w1, p + "ffn.w1", (int64_t)H * F * L);
        GP(w.ffn_b1, p + "ffn.b1", H * F);
        GP(w.ffn_w2, p + "ffn.w2", (int64_t)H * L * F);
        GP(w.ffn_b2, p + "ffn.b2", H * L);
        GP(w.ffn_film_w, p + "ffn.film_w", (int64_t)2 * D * Te);
        GP(w.ffn_film_b, p + "ffn.film_b", 2 * D);
        GP(w.ffn_ln_g, p + "ffn.ln_g", D);
        GP(w.ffn_ln_b, p + "ffn.ln_b", D);
        GP(w.ffn_out_w, p + "ffn.out_w", (int64_t)D * D);
        GP(w.ffn_out_b, p + "ffn.out_b", D);
    }
    return MC_OK;
}

// context-only derived weights (bind_weights itself is a pure lookup / validation pass, also run by mc_model_finalize):
// [2][C][D] = dec_w | dec_wf and [2][C] = dec_bf | 0 -- the folded decoder tail as ONE grouped GEMM
int build_ctx_weights(mc_ctx* c) {
    if (!c->dec_wf) return MC_OK;
    const mc_model_config& g = c->m->cfg;
    const int D = g.latent_dim * g.num_parts;
    const size_t wn = (size_t)g.input_feats * D, bn = (size_t)g.input_feats;
    int rr;
    if ((rr = ws_alloc(c, &c->dec_cat_w, 2 * wn)) != MC_OK || (rr = ws_alloc(c, &c->dec_cat_b, 2 * bn)) != MC_OK) return rr;
    MC_HIP(hipMemcpy(c->dec_cat_w, c->dec_w, wn * sizeof(float), hipMemcpyDeviceToDevice));
    MC_HIP(hipMemcpy(c->dec_cat_w + wn, c->dec_wf, wn * sizeof(float), hipMemcpyDeviceToDevice));
    MC_HIP(hipMemcpy(c->dec_cat_b, c->dec_bf, bn * sizeof(float), hipMemcpyDeviceToDevice));
    MC_HIP(hipMemset(c->dec_cat_b + bn, 0, bn * sizeof(float)));
    return MC_OK;
}

// fp16 hi / lo planes of weight `name` ([rows][K] fp32 on the device), built once per model; chain = K axis stored in the
// chain-permuted order of mc_half.h (second GEMM of the fused MLP)
int half_weight(mc_model* m, const std::string& name, long rows, int K, bool chain, HalfW* out) {
    // the plane cache belongs to the MODEL and is filled from per-context calls (mc_ctx_set_precision): one lock per model, so two
    // contexts switching precision from two threads build each plane once
    std::lock_guard<std::mutex> lock(m->half_mu);
    auto it = m->half.find(name);
    if (it != m->half.end()) { *out = it->second; return MC_OK; }
    const float* src = nullptr;
    int r = get_param(m, name, (int64_t)rows * K, &src);
    if (r != MC_OK) return r;
    HalfW h;
    MC_HIP(hipMalloc((void**)&h.hi, sizeof(mc_half) * 2 * (size_t)rows * K));
    h.lo = h.hi + (size_t)rows * K;
    r = chain ? mc_launch_split_f16_chainperm(src, h.hi, h.lo, rows, K, nullptr) : mc_launch_split_f16(src, h.hi, h.lo, rows * K, nullptr);
    if (r != MC_OK) { (void)hipFree(h.hi); return r; }
    MC_HIP(hipStreamSynchronize(nullptr));
    m->half[name] = h;
    m->half_bytes += sizeof(mc_half) * 2 * (size_t)rows * K;
    *out = h;
    return MC_OK;
}

int bind_half_weights(mc_ctx* c) {
    mc_model* m = c->m;
    const mc_model_config& g = m->cfg;
    const int L = g.latent_dim, H = g.num_parts, D = L * H, F = g.ffn_dim, E = g.num_experts;
    int r;
    for (int i = 0; i < c->NLA; ++i) {
        LayerW& w = c->lw[i];
        const bool is_ctrl = i >= g.num_layers;
        const std::string p = (is_ctrl ? "c" + std::to_string(i - g.num_layers) : "l" + std::to_string(i)) + ".";
        if (D % 128 == 0) {
            if ((r = half_weight(m, p + "ca.out_w", D, D, false, &w.h_ca_out))) return r;
            if ((r = half_weight(m, p + "ffn.out_w", D, D, false, &w.h_ffn_out))) return r;
            if (is_ctrl && (r = half_weight(m, p + "after_w", D, D, false, &w.h_after))) return r;
        }
        if (mc_mlp_h_supported(L, 4 * L)) {
            if ((r = half_weight(m, p + "mm.proj_w", 4 * L, L, false, &w.h_proj))) return r;
            if ((r = half_weight(m, p + "dyn.qkv_w", 3 * L, L, true, &w.h_qkv))) return r;
            if ((r = half_weight(m, p + "mm.fc1_w", (long)E * 4 * L, L, false, &w.h_fc1))) return r;
            if ((r = half_weight(m, p + "mm.fc2_wt", (long)E * L, 4 * L, true, &w.h_fc2))) return r;
        }
        if (mc_mlp_h_supported(L, F)) {
            if ((r = half_weight(m, p + "ffn.w1", (long)H * F, L, false, &w.h_w1))) return r;
            if ((r = half_weight(m, p + "ffn.w2", (long)H * L, F, true, &w.h_w2))) return r;
        }
    }
    return MC_OK;
}

// The fp16-MFMA kernels have no small-batch variants (128-row workgroups, no hidden / K split): up to this many residual
// rows (B=1 at 196 frames: 392) the fp32 small-batch kernels are faster and the reduced-precision modes run on them
// (B=1 50-step DDIM: 68.3 ms on the fp16 kernels, 57.3 ms on the fp32 ones; from B=2 the fp16 kernels win).
static bool use_half(const mc_ctx* c) { return c->prec != MC_PREC_F32 && c->rows > c->half_min_rows; }

// residual rows up to which the fused MLPs split their hidden dimension over workgroups (which = 0 experts, 1 SFFN)
static long split_rows(const mc_ctx* c, int which) { return which ? c->opt.split_rows_sffn : c->opt.split_rows_expert; }
static const McOptions& options_of(const mc_ctx* c) {
    static const McOptions dflt = default_options();      // context-free ops (mc_op_gemm)
    return c ? c->opt : dflt;
}
static long small_gemm_rows(const mc_ctx* c) { return options_of(c).small_gemm_rows; }
static bool chain_on(const mc_ctx* c, int which) { return (options_of(c).chain >> which) & 1; }
static void gemm_opts(const mc_ctx* c, GemmArgs& g) {
    const McOptions& o = options_of(c);
    g.tune = o.gemm_tune; g.small_tile_n = o.small_tile_n; g.wp_grid = o.gemm_wp_grid;
}

int dense(const mc_ctx* c, const float* A, long lda, const float* W, long ldw, const float* bias, const float* R, long ldr,
          float* C, long ldc, long M, int N, int K, int act, hipStream_t s) {
    GemmArgs g;
    gemm_opts(c, g);
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.R = R; g.ldr = ldr;
    g.C = C; g.ldc = ldc; g.M = (int)M; g.N = N; g.K = K; g.act = act;
    if (M <= small_gemm_rows(c) && act == ACT_NONE && K % 32 == 0 && lda % 4 == 0 && ldw % 4 == 0)
        return mc_launch_gemm_small(g, s);          // latency-bound sizes: 64 x 64 tiles (see gemm_small_k)
    return mc_launch_gemm(GM_PLAIN, g, 1, 0, s);
}

// C = A W^T + bias + R on the fp16 MFMA (reduced-precision mode); `hw` = the weight's fp16 planes
int dense_h(const mc_ctx* c, const float* A, const HalfW& hw, const float* bias, const float* R, float* C, long M, int N, int K,
            hipStream_t s) {
    GemmHArgs g;
    g.A = A; g.lda = K; g.Wh = hw.hi; g.Wl = hw.lo; g.bias = bias; g.R = R; g.ldr = N; g.C = C; g.ldc = N;
    g.M = (int)M; g.N = N; g.K = K;
    return mc_launch_gemm_h(g, c->prec == MC_PREC_F16X3, s);
}

// Small batches: a [M x K] x [K x N] GEMM with fewer than ~128 output tiles leaves most of the 256 CUs idle while each
// tile walks the whole K serially.  Split K across workgroups (the grouped launch with column offsets as "group"
// strides), partial sums in `ws`, reduced in a fixed order: C = sum_s A[:, s] W[:, s]^T + bias + R.
int dense_splitk(const mc_ctx* c, const float* A, const float* W, const float* bias, const float* R, float* C, long M, int N, int K,
                 float* ws, size_t ws_floats, hipStream_t s) {
    const int tiles = cdiv(M, 128) * cdiv(N, 128);
    int S = 1;
    // (up to two rounds of the 512 workgroup slots: measured better than stopping at one, B=4 -4.5 %)
    while (S < 8 && tiles * (S * 2) <= 1024 && K % (S * 2 * 32) == 0 && (size_t)(S * 2) * M * N <= ws_floats) S *= 2;
    if (S == 1) return dense(c, A, K, W, K, bias, R, N, C, N, M, N, K, ACT_NONE, s);
    GemmArgs g;
    gemm_opts(c, g);
    g.A = A; g.lda = K; g.a_gstride = K / S;          // split s reads columns [s K/S, (s+1) K/S) of A and of W
    g.W = W; g.ldw = K; g.w_gstride = K / S;
    g.C = ws; g.ldc = N; g.c_gstride = M * N;
    g.M = (int)M; g.N = N; g.K = K / S;
    int r = mc_launch_gemm(GM_PLAIN, g, S, 0, s);
    if (r != MC_OK) return r;
    return mc_launch_splitk_reduce(ws, S, M, N, bias, R, C, s);
}

// One tutel MoE layer + GELU + proj (class MOE, st_attention.py:49-56) over Ntok tokens whose
// gate/expert input `z` ([Ntok, din], embedding already added) is in HBM.
// `gated`: idx/gate/key/counts were already produced (fused gate_k); otherwise run projector + gate finish here.
// expert FFN over the slots of one slot group (mc_route.hip): y2[dst_row] = FC2(gelu(FC1(z[src_row])))
int moe_experts(mc_ctx* c, const MoeW& w, const float* z, long Ntok, int group, hipStream_t s, const HalfW* hw = nullptr,
                const HalfW* hw2 = nullptr) {
    const int E = c->m->cfg.num_experts, din = w.din, hid = 4 * w.din;
    const int max_tiles = cdiv(2 * Ntok, 128) + E;
    const long to = (long)group * c->rb.max_tiles;
    int r;
    if (chain_on(c, 0) && mc_mlp_supported(din, hid)) {
        // fused expert FFN: hidden activations stay on chip (mc_chain.hip)
        MlpArgs m;
        m.dma = chain_on(c, 18) ? 1 : 0;
        {   // top-2 slots of this slot group's source tokens (twins of base layer 0 have no slots of their own)
            const long nsrc = c->led_nsrc > 0 ? c->led_nsrc : Ntok, gs = c->led_gsplit;
            m.ledger_rows = 2 * ((gs <= 0 || gs >= nsrc) ? nsrc : (group == 0 ? gs : nsrc - gs));
        }
        m.X = z; m.ldx = din; m.W1 = w.fc1_w; m.b1 = w.fc1_b; m.W2t = w.fc2_wt; m.b2 = w.fc2_b;
        m.Y = c->y2; m.ldy = din; m.L = din; m.hidden = hid;
        m.tile_group = c->rb.tile_group + to; m.tile_row0 = c->rb.tile_row0 + to; m.tile_nrows = c->rb.tile_nrows + to;
        m.num_tiles = mc_route_num_tiles_ptr(c->rb, group); m.src_row = c->rb.src_row; m.dst_row = c->rb.dst_row;
        if (hw && hw->hi && hw2 && hw2->hi && use_half(c))      // reduced-precision mode: the same fused MLP on the fp16 MFMA
            return mc_launch_mlp_h(MLP_EXPERT, m, hw->hi, hw->lo, hw2->hi, hw2->lo, c->prec == MC_PREC_F16X3, 1, max_tiles, s);
        // small batches (a few dozen tiles, each walking all hidden chunks serially): split the hidden dimension 4 ways,
        // partial FC2 sums in hbuf, reduced in a fixed order (rows of dropped pairs stay unwritten garbage: never read)
        // how many ways: the launch is bound by the busiest CU (n = ceil(workgroups / 256) of them land on it, two co-resident
        // ones finish in 1.45x one -- the load model of mc_launch_gemm_small) times the hidden chunks per workgroup; ~2 Ntok / 128
        // + E / 2 tiles are real.  4 ways except where 3 take a whole round off: B = 2 at 196 frames (620 -> 465 workgroups,
        // 3 -> 2 rounds: 80.5 -> 77.7 ms per 50-step DDIM; at B = 1 the tile count straddles 256 / 3 and 3 ways lose 2 %).
        int S = c->split_expert;                   // env MC_SPLIT_EXPERT at context creation (0: the model below)
        if (S <= 0) {
            const long tiles = 2 * Ntok / 128 + E / 2;
            auto load = [&](int ways) {
                const long n = cdiv(tiles * ways, 256);
                return (1.45 * (double)(n / 2) + (double)(n % 2)) * (double)cdiv(hid / 32, ways);
            };
            S = 4;
            if (hid / 32 >= 4 && load(3) < 0.92 * load(4) && tiles * 3 > 300) S = 3;     // (tiles * 3 <= 300: B = 1, see above)
        }
        if (z == c->z && c->rows <= split_rows(c, 0) && c->N <= c->opt.big_tokens && S > 1 && hid / 32 >= S &&      // (N <= big_tokens: the one-stream schedule, one user of hbuf)
            c->hbuf_floats >= (size_t)S * 2 * Ntok * din) {
            m.Y = c->hbuf; m.nsplit = S; m.y_sstride = 2 * Ntok * din;
            // B = 1 sizes (the estimated tile count straddles 256 / 3): 3 or 4 ways decided on the device from the real count
            const bool dyn = S == 4 && c->split_expert == 0 && (2 * Ntok / 128 + E / 2) * 3 <= 300 && chain_on(c, 13);
            m.dyn_split = dyn ? 1 : 0;
            if ((r = mc_launch_mlp(MLP_EXPERT, m, 1, max_tiles, s))) return r;
            return mc_launch_splitk_reduce(c->hbuf, S, 2 * Ntok, din, nullptr, nullptr, c->y2, s,
                                           dyn ? mc_route_num_tiles_ptr(c->rb, group) : nullptr);
        }
        return mc_launch_mlp(MLP_EXPERT, m, 1, max_tiles, s);
    }
    GemmArgs a;
    gemm_opts(c, a);
    a.A = z; a.lda = din; a.src_row = c->rb.src_row;
    a.W = w.fc1_w; a.ldw = din; a.w_gstride = (long)hid * din;
    a.bias = w.fc1_b; a.b_gstride = hid; a.act = ACT_GELU;
    a.C = c->hbuf; a.ldc = hid; a.N = hid; a.K = din;
    a.tile_group = c->rb.tile_group + to; a.tile_row0 = c->rb.tile_row0 + to; a.tile_nrows = c->rb.tile_nrows + to;
    a.num_tiles = mc_route_num_tiles_ptr(c->rb, group);
    if ((r = mc_launch_gemm(GM_EXP1, a, 1, max_tiles, s))) return r;
    GemmArgs b;
    gemm_opts(c, b);
    b.A = c->hbuf; b.lda = hid; b.W = w.fc2_wt; b.ldw = hid; b.w_gstride = (long)din * hid;
    b.bias = w.fc2_b; b.b_gstride = din; b.dst_row = c->rb.dst_row;
    b.C = c->y2; b.ldc = din; b.N = din; b.K = hid;
    b.tile_group = a.tile_group; b.tile_row0 = a.tile_row0; b.tile_nrows = a.tile_nrows; b.num_tiles = a.num_tiles;
    return mc_launch_gemm(GM_EXP2, b, 1, max_tiles, s);
}

// One tutel MoE layer + GELU + proj (class MOE, st_attention.py:49-56) over Ntok tokens whose
// gate/expert input `z` ([Ntok, din], embedding already added) is in HBM.
// `gated`: idx/gate/key/counts were already produced (fused gate_k); otherwise run projector + gate finish here.
// `gsplit` < Ntok: two slot groups; then only the routing runs here and the caller launches moe_experts per group.
int run_moe(mc_ctx* c, const MoeW& w, const float* z, long Ntok, float* out, long ldout, bool gated, bool twin, long gsplit,
            hipStream_t s, const HalfW* hw = nullptr, const HalfW* hw2 = nullptr) {
    const mc_model_config& g = c->m->cfg;
    const int E = g.num_experts, din = w.din;
    int r;
    if (!gated) {
        // cosine projector (tutel/gates/cosine_top.py): proj = z Wp^T + bp
        if ((r = dense(c, z, din, w.gate_w, din, w.gate_b, nullptr, 0, c->proj, 256, Ntok, 256, din, ACT_NONE, s))) return r;
        if ((r = mc_launch_gate_finish(c->proj, w.sim_n, w.scale, Ntok, E, c->rb, s))) return r;
    }
    const int capacity = g.topk * (int)((double)g.capacity_factor * (double)((Ntok + E - 1) / E));  // tutel extract_critical
    c->cnt_clean = false;
    c->led_nsrc = twin ? Ntok / 2 : Ntok;
    c->led_gsplit = gsplit;
    if ((r = mc_launch_route(Ntok, twin ? Ntok / 2 : Ntok, gsplit, E, capacity, c->rb, s))) return r;
    c->cnt_clean = mc_route_cleans_counts(c->rb, Ntok);
    if (gsplit < Ntok) return MC_OK;
    if ((r = moe_experts(c, w, z, Ntok, 0, s, hw, hw2))) return r;
    if (!out) return MC_OK;                        // the caller launches the projection itself (row ranges)
    if (chain_on(c, 2) && mc_mlp_supported(din, 32) && w.dout % 32 == 0) {
        RowChainArgs p;
        p.split_tokens = c->opt.rowchain_split;
        p.X = c->y2; p.comb_w = c->rb.comb_w; p.W = w.proj_w; p.bias = w.proj_b;
        p.Y = out; p.ldy = ldout; p.N = Ntok; p.L = din; p.Nout = w.dout;
        p.twin_from = twin ? Ntok / 2 : 0;
        return mc_launch_rowchain(0, p, s);
    }
    GemmArgs p;
    gemm_opts(c, p);
    p.A = c->y2; p.lda = din; p.comb_w = c->rb.comb_w;
    p.W = w.proj_w; p.ldw = din; p.bias = w.proj_b;
    p.C = out; p.ldc = ldout; p.M = (int)Ntok; p.N = w.dout; p.K = din;
    return mc_launch_gemm(GM_COMB, p, 1, 0, s);
}

// the array the deferred last FiLM block's fp32 rows live in (read back by denoise_combined)
static float* deferred_a(const mc_ctx* c) { return c->a_tail ? c->a_tail : c->a; }

// ONE answer per context to "does `a` hold fp16 hi | lo planes instead of fp32 rows?" -- asked by film_block (which writes them) and by
// mc_ctx_get_buffer("a") (which must not hand planes out as fp32 rows): reduced-precision context, chain bit 17, the plane launcher's
// shape preconditions (N % 128, K % 64 via D % 128; 32-bit byte offsets into a plane).  Every per-step FiLM weight of a reduced-precision
// context has planes (mc_ctx_set_precision builds them for all layers or fails), so the weight is not part of the answer.
bool a_holds_planes(const mc_ctx* c) {
    const long D = (long)c->m->cfg.latent_dim * c->m->cfg.num_parts;
    return use_half(c) && chain_on(c, 17) && D % 128 == 0 && c->rows * D * 2 < (1L << 31);
}

// rows [row0, row0 + nrows) of:  a = silu(LN(y1 (+ y2)) * (1 + scale) + shift);  h += Linear(a)   (StylizationBlock)
int film_block(mc_ctx* c, float* hs, const float* y1, const float* y2, const float* ln_g, const float* ln_b,
               const float* ss, const float* out_w, const float* out_b, long row0, long nrows, hipStream_t s,
               bool prologue_only = false, TwinAlias y1_alias = TwinAlias(), const HalfW* hw = nullptr, int y1_parts = 1,
               hipEvent_t ev_after_rows = nullptr) {
    // ev_after_rows: recorded on s behind the row kernel (the other sample group's FiLM block may be ordered behind it: run_layer)
    // y1_parts > 1: y1 = that many partial planes of [nrows][D] starting AT y1 (rows relative to row0), summed by the row kernel
    const int D = c->m->cfg.latent_dim * c->m->cfg.num_parts;
    const long o = row0 * D;
    int r;
    StepRef sref;
    if (c->graph_mode) { sref.ptr = c->gstep; sref.stride = 2L * D; }     // `ss` then is the table's row of step 0
    // reduced-precision contexts: `a` is consumed by the fp16-MFMA GEMM only -> written as fp16 planes (hi [rows][D] | lo) into the same
    // buffer: the rows of this range start at halves offset o, the lo plane sits rows * D halves behind the hi plane
    const bool half_gemm = !prologue_only && hw && hw->hi && use_half(c);
    // (the plane path's launcher needs N % 128 == 0, K % 64 == 0 and 32-bit byte offsets into a plane: anything else stays on dense_h)
    const bool planes = half_gemm && a_holds_planes(c);      // (one answer per context: plane rows and fp32 rows never share `a`)
    const long pstride = c->rows * D;
    float* a_rows = prologue_only ? deferred_a(c) : c->a;
    float* a_out = planes ? reinterpret_cast<float*>(reinterpret_cast<mc_half*>(c->a) + o) : a_rows + o;
    // chain bit 29 (round 6): the planes of THIS launch's rows are written fragment-major and the GEMM reads its A fragments straight into registers
    // (gemm_hf_k); needs whole 32-row blocks (a sample group of B x 196 frames with B % 8 == 0 has them), else the row-major planes + gemm_hd_k
    const bool frag_major = planes && chain_on(c, 29) && row0 % 32 == 0 && nrows % 32 == 0 && D % 64 == 0 && D >= 192;
    if ((r = mc_launch_film_rows(y1_parts > 1 ? y1 : y1 + o, y2 ? y2 + o : nullptr, ln_g, ln_b, ss, a_out, nrows, D, s, y1_alias, row0, sref,
                                 y1_parts, nrows * D, planes ? ((c->prec == MC_PREC_F16X3 ? 2 : 1) | (frag_major ? 4 : 0)) : 0, pstride))) return r;
    if (ev_after_rows) MC_HIP(hipEventRecord(ev_after_rows, s));
    if (prologue_only) return MC_OK;
    // h = h + Linear(a)          (st_attention.py:172 / stmogen.py:606)
    if (planes) {
        GemmHArgs g;
        g.Ah = reinterpret_cast<mc_half*>(c->a) + o; g.Al = g.Ah + pstride;
        g.Wh = hw->hi; g.Wl = hw->lo; g.bias = out_b; g.R = hs + o; g.ldr = D; g.C = hs + o; g.ldc = D;
        g.M = (int)nrows; g.N = D; g.K = D;
        g.a_fm = frag_major ? 1 : 0;
        g.acc_init = (chain_on(c, 27) ? 1 : 0) | (chain_on(c, 28) ? 2 : 0);      // 1: residual rows prefetched into registers (same bits), 2: accumulators start as R + bias (plain f16 only)
        return mc_launch_gemm_h(g, c->prec == MC_PREC_F16X3, s);
    }
    if (half_gemm)
        return dense_h(c, c->a + o, *hw, out_b, hs + o, hs + o, nrows, D, D, s);
    if (nrows <= small_gemm_rows(c) && D % 64 == 0) {     // up to a few thousand rows: 64 x 64 tiles, short MFMA chains, no K split (B=8: -11 % per step)
        GemmArgs q;
        gemm_opts(c, q);
        q.A = c->a + o; q.lda = D; q.W = out_w; q.ldw = D; q.bias = out_b; q.R = hs + o; q.ldr = D; q.C = hs + o; q.ldc = D;
        q.M = (int)nrows; q.N = D; q.K = D;
        return mc_launch_gemm_small(q, s);
    }
    if (nrows <= 2048 && c->hbuf_floats)       // few output tiles: split K (hbuf is free scratch on the fused path)
        return dense_splitk(c, c->a + o, out_w, out_b, hs + o, hs + o, nrows, D, D, c->hbuf, c->hbuf_floats, s);
    if (!c->prof_on) return dense(c, c->a + o, D, out_w, D, out_b, hs + o, D, hs + o, D, nrows, D, D, ACT_NONE, s);
    // mc_ctx_profile: HIP events around this launch ON THE STREAM IT IS LAUNCHED ON (the sample group's stream)
    ProfRec pr;
    pr.rows = nrows;
    if (hipEventCreate(&pr.e0) != hipSuccess || hipEventCreate(&pr.e1) != hipSuccess) { mc_set_error("hipEventCreate failed"); return MC_ERR_HIP; }
    MC_HIP(hipEventRecord(pr.e0, s));
    r = dense(c, c->a + o, D, out_w, D, out_b, hs + o, D, hs + o, D, nrows, D, D, ACT_NONE, s);
    MC_HIP(hipEventRecord(pr.e1, s));
    c->prof.push_back(pr);
    return r;
}

// Everything of a DecoderLayer AFTER the expert MLP, restricted to residual-stream rows [row0, row0 + nrows)
// (whole samples): MoE combine + proj, body LN + q/k/v, body and temporal attention, proj_out FiLM block, SFFN, its
// FiLM block.  Every kernel here is row-independent, so disjoint row ranges can run on different streams.
// `phase`: 0 = everything; 1 = the front only (combine + proj, LN + q/k/v, body topology); 2 = the temporal attention only
// (the twin layer of the large-batch schedule runs the front per sample sub-group and the rest per CFG half: run_layer)
int layer_rows(mc_ctx* c, int i, float* hs, int step, bool twin, long row0, long nrows, hipStream_t s, hipStream_t st, int phase = 0, long rows2_0 = -1) {
    // rows2_0 >= 0 (phase 1, fused front only): the same launch also covers rows [rows2_0, rows2_0 + nrows) -- the aliased twins of this range
    // twin layer: rows of the second CFG half whose routing equals their twin's are aliased, not recomputed
    TwinAlias tok_alias, frame_alias;
    const int* twin_flag = nullptr;
    if (twin && chain_on(c, 8) && !c->no_alias) {
        twin_flag = mc_route_split_flag_ptr(c->rb);
        tok_alias.split_flag = frame_alias.split_flag = twin_flag;
        tok_alias.from = c->N / 2;
        frame_alias.from = c->rows / 2;
    }
    const mc_model_config& g = c->m->cfg;
    const int L = g.latent_dim, H = g.num_parts, D = L * H;
    const LayerW& w = c->lw[i];
    const long tok0 = row0 * H, ntok = nrows * H;
    int r;
    // ---- post-score combine + GELU + MOE.proj -> mf [N][4L] ----
    // large batches: proj + body LayerNorm + q/k/v as one kernel (small ones keep them apart: the temporal branch then
    // starts on the side stream right after the projection)
    const bool pq_fused = chain_on(c, 2) && chain_on(c, 10) && c->N > c->opt.big_tokens && mc_mlp_supported(L, 32) && (4 * L) % 32 == 0;
    // ... and the body-topology attention too (pqbody_k: frame-aligned tiles, q/k/v never leave the chip): fp32 path, L = 128, 12 parts
    const bool body_fused = pq_fused && chain_on(c, 15) && H == 12 && g.dyn_heads == 8 && (L == 128 || L == 64);
    if (phase == 2) {
        // (front done elsewhere)
    } else if (chain_on(c, 2) && mc_mlp_supported(L, 32) && (4 * L) % 32 == 0) {
        RowChainArgs p;
        p.split_tokens = c->opt.rowchain_split;
        p.X = c->y2; p.comb_w = c->rb.comb_w; p.W = w.mm.proj_w; p.bias = w.mm.proj_b;
        p.Y = c->mf; p.ldy = 4 * L; p.tok0 = tok0; p.N = tok0 + ntok; p.L = L; p.Nout = 4 * L;
        p.twin_from = twin ? c->N / 2 : 0;
        p.alias = tok_alias;
        if (pq_fused) {       // + the dynamic body topology's shared LayerNorm and q/k/v on the body_value columns
            p.gamma = w.dyn_g; p.beta = w.dyn_b; p.W2 = w.qkv_w; p.bias2 = w.qkv_b; p.Y2 = c->qkv; p.ldy2 = 3 * L;
            p.pad_row = c->N;      // mf / qkv carry 128 padding rows (mc_ctx_create): projqkv_k's stores are unconditional
            if (body_fused) {
                p.wsm = w.wsm; p.ys = c->ys;
                if (rows2_0 >= 0) { p.tok2 = rows2_0 * H; p.N2 = (rows2_0 + nrows) * H; p.nblk1 = 1; }
                if (use_half(c) && w.h_proj.hi && w.h_qkv.hi) {
                    if ((r = mc_launch_pqbody_h(p, H, w.h_proj.hi, w.h_proj.lo, w.h_qkv.hi, w.h_qkv.lo, c->prec == MC_PREC_F16X3, s))) return r;
                } else if ((r = mc_launch_pqbody(p, H, s))) return r;
            } else if (use_half(c) && w.h_proj.hi && w.h_qkv.hi) {
                if ((r = mc_launch_projqkv_h(p, w.h_proj.hi, w.h_proj.lo, w.h_qkv.hi, w.h_qkv.lo, c->prec == MC_PREC_F16X3, s))) return r;
            } else if ((r = mc_launch_projqkv(p, s))) return r;
        } else if ((r = mc_launch_rowchain(0, p, s))) return r;
    } else {
        GemmArgs p;
        gemm_opts(c, p);
        p.A = c->y2 + 2 * tok0 * L; p.lda = L; p.comb_w = c->rb.comb_w + 2 * tok0;
        p.W = w.mm.proj_w; p.ldw = L; p.bias = w.mm.proj_b;
        p.C = c->mf + tok0 * 4 * L; p.ldc = 4 * L; p.M = (int)ntok; p.N = 4 * L; p.K = L;
        if ((r = mc_launch_gemm(GM_COMB, p, 1, 0, s))) return r;
    }
    // ---- temporal linear attention: needs only mf; on `st` (a second stream for small batches) or inline ----
    const float* tfl = c->tf + (long)i * c->Ntxt * 2 * L;
    // small batches: the longer branch (temporal) stays on `s`, LN + q/k/v + body go to the side stream -- the fork latency is
    // then paid by the short branch and the join event has fired long before `s` reaches it (B=1: -13 us per layer vs the
    // temporal branch on the side stream)
    hipStream_t sb = s;          // stream of the body branch
    hipStream_t stt = s;         // stream of the temporal branch
    if (st != s) {
        MC_HIP(hipEventRecord(c->ev_fork, s));
        MC_HIP(hipStreamWaitEvent(st, c->ev_fork, 0));
        if (chain_on(c, 14) && c->rows <= 1200) sb = st; else stt = st;      // (B <= 3 at 196 frames: -1.5 .. -3 %; B = 4: +1 %)
        if (c->dbg_delay_us != 0) {        // (tests) hold the side stream (> 0) or the main stream (< 0) behind the fork
            int r2 = mc_launch_spin((c->dbg_delay_us > 0 ? c->dbg_delay_us : -c->dbg_delay_us) * 100, c->dbg_delay_us > 0 ? st : s);
            if (r2 != MC_OK) return r2;
        }
    }
    // ---- dynamic body topology: shared LayerNorm + q/k/v ----
    if (pq_fused || phase == 2) {
        // q/k/v were produced by projqkv_k above
    } else if (chain_on(c, 2) && mc_mlp_supported(L, 32)) {
        RowChainArgs q;
        q.split_tokens = c->opt.rowchain_split;
        q.X = c->mf; q.ldx = 4 * L; q.gamma = w.dyn_g; q.beta = w.dyn_b; q.W = w.qkv_w; q.bias = w.qkv_b;
        q.Y = c->qkv; q.ldy = 3 * L; q.tok0 = tok0; q.N = tok0 + ntok; q.L = L; q.Nout = 3 * L;
        q.alias = tok_alias;
        if ((r = mc_launch_rowchain(1, q, sb))) return r;
    } else {
        if ((r = mc_launch_ln_rows(c->mf + tok0 * 4 * L, 4 * L, 0, w.dyn_g, w.dyn_b, nullptr, 1, c->z + tok0 * L, L, ntok, L, sb))) return r;
        if ((r = dense(c, c->z + tok0 * L, L, w.qkv_w, L, w.qkv_b, nullptr, 0, c->qkv + tok0 * 3 * L, 3 * L, ntok, 3 * L, L, ACT_NONE, sb))) return r;
    }
    if (!body_fused && phase != 2 &&
        (r = mc_launch_body(c->mf + tok0 * 4 * L, 4 * L, c->qkv + tok0 * 3 * L, w.wsm, c->ys + row0 * D, nrows, H, L, g.dyn_heads, sb,
                            frame_alias, row0))) return r;
    if (sb != s) MC_HIP(hipEventRecord(c->ev_join, sb));
    if (phase == 1) return MC_OK;
    const int tnb = (int)(nrows / c->T);
    if (use_half(c) && chain_on(c, 20) && (L == 128 || L == 64) && (long)tnb * H > c->opt.temporal_split) {
        // reduced-precision mode: both contractions on the fp16 MFMA (whole-(sample, part) workgroups; the sliced small-batch form stays fp32)
        if ((r = mc_launch_temporal_h(c->mf, tfl, c->mask, c->yt, (int)(row0 / c->T), tnb, c->B, c->T, g.max_text_len, H, L,
                                      c->prec == MC_PREC_F16X3, stt, twin_flag, chain_on(c, 24)))) return r;
    } else if ((r = mc_launch_temporal(c->mf, tfl, c->mask, c->yt, (int)(row0 / c->T), tnb, c->B, c->T,
                                       g.max_text_len, H, L, stt, twin_flag, c->opt.temporal_split, chain_on(c, 22), chain_on(c, 24)))) return r;
    if (stt != s) MC_HIP(hipEventRecord(c->ev_join, stt));
    if (st != s) MC_HIP(hipStreamWaitEvent(s, c->ev_join, 0));
    return MC_OK;
}

int layer_rows_tail(mc_ctx* c, int i, float* hs, int step, bool twin, long row0, long nrows, hipStream_t s, hipEvent_t ev_rows = nullptr,
                    hipEvent_t wait_first = nullptr) {
    // ev_rows: recorded behind the first FiLM block's row kernel; wait_first: this range's tail starts behind that event of the other group
    if (wait_first) MC_HIP(hipStreamWaitEvent(s, wait_first, 0));
    const mc_model_config& g = c->m->cfg;
    const int L = g.latent_dim, H = g.num_parts, D = L * H, F = g.ffn_dim;
    const LayerW& w = c->lw[i];
    int r;
    const float* ss0 = c->ss + ((long)(i * 2 + 0) * c->maxS + step) * 2 * D;
    TwinAlias ys_alias;
    if (twin && chain_on(c, 8) && !c->no_alias) { ys_alias.split_flag = mc_route_split_flag_ptr(c->rb); ys_alias.from = c->rows / 2; }
    if ((r = film_block(c, hs, c->ys, c->yt, w.ca_ln_g, w.ca_ln_b, ss0, w.ca_out_w, w.ca_out_b, row0, nrows, s, false, ys_alias, &w.h_ca_out, 1, ev_rows))) return r;
    // ---- SFFN (stmogen.py:596-607): 12 part-wise FFNs as grouped GEMMs ----
    const long o = row0 * D;
    int z2_parts = 1;
    if (chain_on(c, 0) && mc_mlp_supported(L, F)) {
        MlpArgs m;
        m.dma = chain_on(c, 18) ? 1 : 0;
        m.X = hs + o; m.ldx = D; m.x_gstride = L;
        m.W1 = w.ffn_w1; m.b1 = w.ffn_b1; m.W2t = w.ffn_w2; m.b2 = w.ffn_b2;
        m.Y = c->z2 + o; m.ldy = D; m.y_gstride = L; m.M = (int)nrows; m.L = L; m.hidden = F;
        // hidden split of the part-wise FFNs (partial sums folded into the FiLM row kernel): ways by the load model of
        // mc_launch_gemm_small plus ~2 chunk times of fixed cost per workgroup -- 4 ways up to B = 5 (and at B = 8: 184.5 -> 178.0 ms
        // per 50-step DDIM), 2 at B = 6 and B = 16 (293.7 -> 287.5), none at B = 12 or beyond 8192 rows
        int S = c->split_sffn;                     // env MC_SPLIT_SFFN at context creation (0: the model below)
        if (S <= 0) {
            S = 1;
            if (nrows <= split_rows(c, 1) && nrows == c->rows) {      // (one launch over the whole batch: the partial planes live in the one hbuf -- not in the two-stream schedule)
                auto load = [&](int ways) {
                    const long n = cdiv((long)cdiv(nrows, 128) * H * ways, 256);
                    return (1.45 * (double)(n / 2) + (double)(n % 2)) * ((double)cdiv(F / 32, ways) + 2.0);
                };
                double best = load(1);
                for (int ways = 2; ways <= 4; ways *= 2)
                    if (F / 32 >= ways && load(ways) <= 1.03 * best) { S = ways; best = load(ways) < best ? load(ways) : best; }
            }
        }
        if (use_half(c) && w.h_w1.hi && w.h_w2.hi) {
            if ((r = mc_launch_mlp_h(MLP_PARTS, m, w.h_w1.hi, w.h_w1.lo, w.h_w2.hi, w.h_w2.lo, c->prec == MC_PREC_F16X3, H, 0, s))) return r;
        } else
        if (nrows <= split_rows(c, 1) && nrows == c->rows && S > 1 && F / 32 >= S && c->hbuf_floats >= (size_t)S * nrows * D) {     // small batches: see moe_experts
            m.Y = c->hbuf; m.nsplit = S; m.y_sstride = nrows * D;
            if ((r = mc_launch_mlp(MLP_PARTS, m, H, 0, s))) return r;
            if (chain_on(c, 12)) z2_parts = S;          // the FiLM row kernel adds the partial planes up itself
            else if ((r = mc_launch_splitk_reduce(c->hbuf, S, nrows, D, nullptr, nullptr, c->z2 + o, s))) return r;
        } else if ((r = mc_launch_mlp(MLP_PARTS, m, H, 0, s))) return r;
    } else {
        GemmArgs f1;
        gemm_opts(c, f1);
        f1.A = hs + o; f1.lda = D; f1.a_gstride = L;
        f1.W = w.ffn_w1; f1.ldw = L; f1.w_gstride = (long)F * L;
        f1.bias = w.ffn_b1; f1.b_gstride = F; f1.act = ACT_GELU;
        f1.C = c->fh + row0 * H * F; f1.ldc = (long)H * F; f1.c_gstride = F;
        f1.M = (int)nrows; f1.N = F; f1.K = L;
        if ((r = mc_launch_gemm(GM_PLAIN, f1, H, 0, s))) return r;
        GemmArgs f2;
        gemm_opts(c, f2);
        f2.A = c->fh + row0 * H * F; f2.lda = (long)H * F; f2.a_gstride = F;
        f2.W = w.ffn_w2; f2.ldw = F; f2.w_gstride = (long)L * F;
        f2.bias = w.ffn_b2; f2.b_gstride = L;
        f2.C = c->z2 + o; f2.ldc = D; f2.c_gstride = L;
        f2.M = (int)nrows; f2.N = L; f2.K = F;
        if ((r = mc_launch_gemm(GM_PLAIN, f2, H, 0, s))) return r;
    }
    const float* ss1 = c->ss + ((long)(i * 2 + 1) * c->maxS + step) * 2 * D;
    return film_block(c, hs, z2_parts > 1 ? c->hbuf : c->z2, nullptr, w.ffn_ln_g, w.ffn_ln_b, ss1, w.ffn_out_w, w.ffn_out_b, row0, nrows, s,
                      c->defer_last_gemm && i == g.num_layers - 1, TwinAlias(), &w.h_ffn_out, z2_parts);
}

// groups of whole samples for the multi-stream schedule: group k = rows [part_row0(k), part_row0(k + 1))
long part_row0(const mc_ctx* c, int k) { return ((long)2 * c->B * k / c->nparts) * c->T; }
// Pick the side stream that runs BESIDE the caller's stream `s` (see mc_ctx::side_cand): a 60 us spin kernel on `s` and one on the candidate, started
// together -- ~65 us when the two streams sit on different hardware queues, ~125 us when they share one.  Once per (context, caller stream); host-synchronous
// (~0.5 ms), so never inside a stream capture (a captured step keeps the stream picked by the eager calls before it).  MC_SIDE_PROBE=0 switches it off.
int pick_side_stream(mc_ctx* c, hipStream_t s) {
    if (c->side_picked && c->side_for == s) return MC_OK;
    static const bool enabled = [] { const char* e = getenv("MC_SIDE_PROBE"); return !e || atoi(e) != 0; }();
    if (!enabled || c->graph_mode || c->graph_exec) return MC_OK;
    for (int k = 0; k < c->side_memo_n; ++k)
        if (c->side_memo_for[k] == s) {          // answered before for this caller stream: switch without a probe (the previous call joined its side work)
            c->side = c->side_cand[c->side_memo_pick[k]];
            c->parts[0] = c->side;
            c->side_for = s;
            return MC_OK;
        }
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return MC_OK; }
    hipEvent_t e0 = nullptr, e1 = nullptr, eq = nullptr;
    MC_HIP(hipEventCreate(&e0));
    MC_HIP(hipEventCreate(&e1));
    MC_HIP(hipEventCreateWithFlags(&eq, hipEventDisableTiming));
    int best = 0, r = MC_OK;
    float best_ms = 1e30f;
    for (int rep = 0; rep < 2 && r == MC_OK; ++rep)           // (the first round also pays the candidates' first-use cost: the second one decides)
        for (int k = 0; k < mc_ctx::SIDE_CAND && r == MC_OK; ++k) {
            hipStream_t q = c->side_cand[k];
            bool ok = hipEventRecord(e0, s) == hipSuccess && hipStreamWaitEvent(q, e0, 0) == hipSuccess;
            ok = ok && mc_launch_spin(6000, s) == MC_OK && mc_launch_spin(6000, q) == MC_OK;
            ok = ok && hipEventRecord(eq, q) == hipSuccess && hipStreamWaitEvent(s, eq, 0) == hipSuccess && hipEventRecord(e1, s) == hipSuccess &&
                 hipEventSynchronize(e1) == hipSuccess;
            float ms = 0.f;
            ok = ok && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
            if (!ok) { mc_set_error("side-stream probe failed: %s", hipGetErrorString(hipGetLastError())); r = MC_ERR_HIP; break; }
            if (rep == 1) {
                c->side_probe_ms[k] = ms;
                if (ms < best_ms - 0.02f) { best_ms = ms; best = k; }      // (20 us margin: near ties keep the earlier candidate)
            }
        }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(eq);
    if (r != MC_OK) return r;
    c->side = c->side_cand[best];
    c->parts[0] = c->side;
    c->side_for = s;
    c->side_picked = true;
    {
        const int slot = c->side_memo_n < mc_ctx::SIDE_MEMO ? c->side_memo_n++ : 0;      // (more than SIDE_MEMO caller streams: the oldest answer is re-probed)
        c->side_memo_for[slot] = s;
        c->side_memo_pick[slot] = best;
    }
    if (getenv("MC_SIDE_PROBE_VERBOSE"))
        fprintf(stderr, "[motioncraft_amd] side stream for caller stream %p: candidate %d (spin pair %.0f %.0f %.0f %.0f us)\n", (void*)s, best,
                c->side_probe_ms[0] * 1e3f, c->side_probe_ms[1] * 1e3f, c->side_probe_ms[2] * 1e3f, c->side_probe_ms[3] * 1e3f);
    return MC_OK;
}

hipStream_t part_stream(const mc_ctx* c, int k, hipStream_t s) { return k == 0 ? s : c->parts[k - 1]; }
int parts_fork(mc_ctx* c, hipStream_t s) {
    MC_HIP(hipEventRecord(c->ev_fork, s));
    for (int k = 1; k < c->nparts; ++k) MC_HIP(hipStreamWaitEvent(c->parts[k - 1], c->ev_fork, 0));
    return MC_OK;
}
int parts_join(mc_ctx* c, hipStream_t s) {
    for (int k = 1; k < c->nparts; ++k) {
        MC_HIP(hipEventRecord(c->ev_parts[k - 1], c->parts[k - 1]));
        MC_HIP(hipStreamWaitEvent(s, c->ev_parts[k - 1], 0));
    }
    return MC_OK;
}

// One DecoderLayer (STMA + SFFN, stmogen.py:610-623) in place on the residual stream `hs` [rows, D];
// `i` selects the layer slot (weights, text K/V, FiLM tables): base layers first, control copies after.
// `split`: 0 = one stream; 1 = CFG halves on two streams, joined at the end of the layer; 2 = same, but the halves
// stay apart across layers (the caller joins after the last one) and the gate is split too -- the streams only meet at
// the routing step, the one place where tokens of the whole batch are ranked against each other.
int run_layer(mc_ctx* c, int i, float* hs, int step, bool twin_ok, int split, hipStream_t s) {
    const mc_model_config& g = c->m->cfg;
    const int L = g.latent_dim, H = g.num_parts;
    const LayerW& w = c->lw[i];
    int r;
    // ---- STMA: gate + routing + experts (the only part that couples tokens across the batch) ----
    const bool fused_gate = chain_on(c, 1) && mc_mlp_supported(L, 32);
    // The two CFG halves enter base layer 0 with the same residual stream (the pose encoder output is written to
    // both, stmogen.py:736-740), so gate scores and expert outputs of token i + N/2 equal those of token i:
    // gate and experts run on the first half only, routing still ranks all N tokens ("twin" mode, mc_route.hip).
    const bool twin = twin_ok && fused_gate && chain_on(c, 2) && chain_on(c, 4) && (c->N % 2 == 0) &&
                      c->rb.tie_xor == 0xFFFFFFFFu;   // (the dedupe relies on a twin ranking right behind its original: stable tie order)
    // Twin layer of the large-batch schedule: gate, experts and the front kernels exist for the FIRST CFG half only (the second one
    // aliases it), so the half is cut into two sample sub-groups that go down the two streams (otherwise one stream idles for the
    // first ~1.75 ms of every step at B=64); the streams cross-join behind the front (the temporal kernel of either CFG half reads
    // mf rows of the whole first half) and continue per CFG half as in every other layer.
    const long sub_rows = ((long)c->B / 2) * c->T;          // rows of the first sub-group (whole samples)
    const bool twin_split = twin && split == 2 && c->nparts == 2 && chain_on(c, 6) && chain_on(c, 8) && chain_on(c, 16) && !c->no_alias &&
                            sub_rows > 0;
    if (fused_gate) {
        GateArgs ga;
        ga.X = hs; ga.ldx = L; ga.gamma = w.norm_g; ga.beta = w.norm_b; ga.emb = w.mm.emb; ga.emb_mod = c->T * H;
        ga.Z = c->z; ga.Wp = w.mm.gate_w; ga.bp = w.mm.gate_b; ga.sim_nT = w.mm.sim_nT; ga.logit_scale = w.mm.scale;
        ga.E = g.num_experts; ga.L = L; ga.small_tokens = c->gate_small_tokens;
        ga.idx = c->rb.idx; ga.gate = c->rb.gate; ga.key = c->rb.key; ga.cnt = c->rb.state;
        if (split == 2 && (!twin || twin_split)) {
            if (!c->cnt_clean) MC_HIP(hipMemsetAsync(ga.cnt, 0, sizeof(int) * 32, s));
            c->cnt_clean = false;
            if ((r = parts_fork(c, s))) return r;
            ga.zero_cnt = 0;
            for (int k = 0; k < c->nparts; ++k) {
                if (twin_split) { ga.tok0 = k ? sub_rows * H : 0; ga.N = k ? c->N / 2 : sub_rows * H; }
                else { ga.tok0 = part_row0(c, k) * H; ga.N = part_row0(c, k + 1) * H; }
                // The group that reaches the join last runs its gate ALONE on the chip (the other stream already waits for the routing): 1176
                // tiles of 128 tokens on 512 workgroup slots are 2.3 rounds -- the third one 30 % full.  Cut the launch at whole rounds; the
                // rest goes to gate_small_k (32-token workgroups whose waves split the projector chunks: bit-identical scores, tested) on
                // the waiting stream, so it runs BESIDE the big launch.
                const long slots = 2L * mc_device_cus(), tiles = cdiv(ga.N - ga.tok0, 128L), rem = tiles % slots;
                if (!twin_split && c->nparts == 2 && k == 1 && chain_on(c, 23) && tiles > slots && rem > 0 && 8 * rem <= 5 * slots) {
                    hipStream_t sk = part_stream(c, k, s);
                    MC_HIP(hipEventRecord(c->ev_gate, sk));                      // (the group's rows: behind its last FiLM GEMM)
                    const long cut = ga.tok0 + (tiles - rem) * 128;
                    GateArgs gb = ga;
                    gb.N = cut;
                    if ((r = mc_launch_gate(gb, sk))) return r;
                    GateArgs gs = ga;
                    gs.tok0 = cut;
                    gs.small_tokens = ga.N - cut;                                // -> gate_small_k
                    MC_HIP(hipStreamWaitEvent(s, c->ev_gate, 0));
                    if ((r = mc_launch_gate(gs, s))) return r;
                    continue;
                }
                if ((r = mc_launch_gate(ga, part_stream(c, k, s)))) return r;
            }
        } else {
            ga.N = twin ? c->N / 2 : c->N;
            ga.zero_cnt = c->cnt_clean ? 0 : 1;      // small batches: the previous layer's routing kernel left the counts zeroed
            c->cnt_clean = false;
            if ((r = mc_launch_gate(ga, s))) return r;
        }
        if (split == 2 && (r = parts_join(c, s))) return r;      // routing ranks the whole batch: every group must have arrived
    } else {
        if ((r = mc_launch_ln_rows(hs, L, 0, w.norm_g, w.norm_b, w.mm.emb, c->T * H, c->z, L, c->N, L, s))) return r;
    }
    // two slot groups when the two sample groups run on two streams: each group's expert MLP joins its own chain
    const bool grouped = split == 2 && c->nparts == 2 && chain_on(c, 6);
    const long gsplit = twin_split ? sub_rows * H : grouped ? part_row0(c, 1) * H : c->N;
    if ((r = run_moe(c, w.mm, c->z, c->N, nullptr, 0, fused_gate, twin, gsplit, s, &w.h_fc1, &w.h_fc2))) return r;   // routing (+ experts if one group)
    if (c->cap_idx) {
        if (twin) {     // expert ids exist for the first half only: the twins have the same ones
            MC_HIP(hipMemcpyAsync(c->cap_idx + (long)i * 2 * c->N, c->rb.idx, sizeof(int) * c->N, hipMemcpyDeviceToDevice, s));
            MC_HIP(hipMemcpyAsync(c->cap_idx + (long)i * 2 * c->N + c->N, c->rb.idx, sizeof(int) * c->N, hipMemcpyDeviceToDevice, s));
        } else {
            MC_HIP(hipMemcpyAsync(c->cap_idx + (long)i * 2 * c->N, c->rb.idx, sizeof(int) * 2 * c->N, hipMemcpyDeviceToDevice, s));
        }
        MC_HIP(hipMemcpyAsync(c->cap_w + (long)i * 2 * c->N, c->rb.comb_w, sizeof(float) * 2 * c->N, hipMemcpyDeviceToDevice, s));
    }
    // ---- the row-independent rest of the layer ----
    const long half = (long)c->B * c->T;        // rows of one CFG half
    // (measured per shape, same-box A/B: helps the exact-fp32 L = 128 step; the fp16 modes -- whose GEMMs are a small part of the chain -- lose 0.1 - 0.2 ms and the
    //  L = 64 models (M2D) 0.35 ms with it, batch 32 is neutral: applied where it helps)
    const bool evstag = split == 2 && c->nparts == 2 && chain_on(c, 26) && !chain_on(c, 23) && !use_half(c) && L == 128;      // (bit 23 uses the same event)
    if (split) {
        // Large batches: the two CFG halves go down two streams.  Each kernel of the chain fills 4.59 "waves" of
        // workgroups at B=64, so ~8 % of every launch is a tail on a partly idle chip; with two independent chains in
        // flight the next kernel of one half starts inside the tail of the other (same effect as two batches in flight).
        if (twin_split) {
            const long half_rows = (long)c->B * c->T;
            hipStream_t s1 = c->parts[0];
            if ((r = parts_fork(c, s))) return r;
            if ((r = moe_experts(c, w.mm, c->z, c->N, 0, s, &w.h_fc1, &w.h_fc2))) return r;
            if ((r = moe_experts(c, w.mm, c->z, c->N, 1, s1, &w.h_fc1, &w.h_fc2))) return r;
            // (round 5, chain bit 25) the fused front of a sub-group also covers that sub-group's twins in the second CFG half: in the usual case
            // (no twin pair split by a capacity cut) those workgroups exit at once, and as part of THIS launch they start inside its tail --
            // as a launch of their own behind the cross-join they queued for LDS behind the other stream's temporal kernel (~125 us per step
            // in front of the second group's temporal kernel: profiles/r05_b64_timeline.txt).  A twin reads its ORIGINAL's expert rows, and the
            // original belongs to the same sub-group, i.e. the same stream: no new dependency.
            const int L_ = g.latent_dim;
            const bool front_covers_twins = chain_on(c, 25) && chain_on(c, 2) && chain_on(c, 10) && chain_on(c, 15) && c->N > c->opt.big_tokens &&
                                            mc_mlp_supported(L_, 32) && H == 12 && g.dyn_heads == 8 && (L_ == 128 || L_ == 64);
            if ((r = layer_rows(c, i, hs, step, twin, 0, sub_rows, s, s, 1, front_covers_twins ? half_rows : -1))) return r;
            if ((r = layer_rows(c, i, hs, step, twin, sub_rows, half_rows - sub_rows, s1, s1, 1, front_covers_twins ? half_rows + sub_rows : -1))) return r;
            // cross-join: each stream waits for the other's front
            MC_HIP(hipEventRecord(c->ev_join, s));
            MC_HIP(hipEventRecord(c->ev_parts[0], s1));
            MC_HIP(hipStreamWaitEvent(s1, c->ev_join, 0));
            MC_HIP(hipStreamWaitEvent(s, c->ev_parts[0], 0));
            // the second CFG half's own front: exits at once while no twin pair was split by a capacity cut (the usual case)
            if (!front_covers_twins && (r = layer_rows(c, i, hs, step, twin, half_rows, half_rows, s1, s1, 1))) return r;
            if ((r = layer_rows(c, i, hs, step, twin, 0, half_rows, s, s, 2))) return r;
            if ((r = layer_rows(c, i, hs, step, twin, half_rows, half_rows, s1, s1, 2))) return r;
            for (int k = 0; k < c->nparts; ++k) {
                if (c->dbg_delay_us != 0 && k == (c->dbg_delay_us > 0 ? 1 : 0) &&
                    (r = mc_launch_spin((c->dbg_delay_us > 0 ? c->dbg_delay_us : -c->dbg_delay_us) * 100, part_stream(c, k, s)))) return r;
                if ((r = layer_rows_tail(c, i, hs, step, twin, part_row0(c, k), part_row0(c, k + 1) - part_row0(c, k), part_stream(c, k, s),
                                         (evstag && k == 0) ? c->ev_gate : nullptr, (evstag && k == 1) ? c->ev_gate : nullptr))) return r;
            }
            return MC_OK;
        }
        if (grouped && twin) {         // group 1 combines group 0's expert rows (its own tokens have no slots): fork after them
            if ((r = moe_experts(c, w.mm, c->z, c->N, 0, s, &w.h_fc1, &w.h_fc2))) return r;
            if ((r = parts_fork(c, s))) return r;
        } else {
            if ((r = parts_fork(c, s))) return r;
            if (grouped) {
                if ((r = moe_experts(c, w.mm, c->z, c->N, 0, s, &w.h_fc1, &w.h_fc2))) return r;
                if ((r = moe_experts(c, w.mm, c->z, c->N, 1, c->parts[0], &w.h_fc1, &w.h_fc2))) return r;
            }
        }
        for (int k = 0; k < c->nparts; ++k) {
            hipStream_t sk = part_stream(c, k, s);
            if ((r = layer_rows(c, i, hs, step, twin, part_row0(c, k), part_row0(c, k + 1) - part_row0(c, k), sk, sk))) return r;
            if (k == 0 && twin && chain_on(c, 8) && !c->no_alias) {
                // twin aliasing: the other groups read group 0's mf / ys instead of producing their own
                MC_HIP(hipEventRecord(c->ev_join, s));
                for (int j = 1; j < c->nparts; ++j) MC_HIP(hipStreamWaitEvent(c->parts[j - 1], c->ev_join, 0));
            }
        }
        for (int k = 0; k < c->nparts; ++k) {
            // (tests) hold one sample group's stream: > 0 the second group, < 0 the first
            if (c->dbg_delay_us != 0 && k == (c->dbg_delay_us > 0 ? 1 : 0) &&
                (r = mc_launch_spin((c->dbg_delay_us > 0 ? c->dbg_delay_us : -c->dbg_delay_us) * 100, part_stream(c, k, s)))) return r;
            if ((r = layer_rows_tail(c, i, hs, step, twin, part_row0(c, k), part_row0(c, k + 1) - part_row0(c, k), part_stream(c, k, s),
                                     (evstag && k == 0) ? c->ev_gate : nullptr, (evstag && k == 1) ? c->ev_gate : nullptr))) return r;
        }
        if (split == 1 && (r = parts_join(c, s))) return r;
        return MC_OK;
    }
    // Small batches: the temporal branch runs on the side stream beside LN + qkv + body (measured +2.4 % at B=8).
    const bool side_temporal = c->side && (chain_on(c, 3) || c->N <= c->opt.big_tokens);
    if ((r = layer_rows(c, i, hs, step, twin, 0, 2 * half, s, side_temporal ? c->side : s))) return r;
    return layer_rows_tail(c, i, hs, step, twin, 0, 2 * half, s);
}

// The largest grid any routing call of this context can launch cooperatively (mc_ctx_create and mc_ctx_set_option("route_coop") share
// it): the motion MoE routes N tokens, the text MoE (mc_ctx_set_condition) Ntxt -- either may be the one inside route_coop_k's size
// range; sizes in the one-workgroup regime or beyond the kernel's range need nothing.  Counted at 10 pairs per thread, the larger of
// the two grids the launch may pick.
int coop_grid_needed(const mc_ctx* c) {
    const long small = c->rb.small_pairs >= 0 ? c->rb.small_pairs : -1;
    int nwg = 0;
    for (long n : {c->N, c->Ntxt}) {
        const bool one_wg = small >= 0 ? 2 * n <= small : mc_route_is_small(n);
        if (!one_wg && mc_route_coop_wgs(n) > nwg) nwg = mc_route_coop_wgs(n);
    }
    return nwg;
}

}  // namespace

extern "C" {

int mc_device_count(int* n) {
    MC_HIP(hipGetDeviceCount(n));
    return MC_OK;
}
int mc_set_device(int dev) {
    MC_HIP(hipSetDevice(dev));
    return MC_OK;
}

int mc_model_create(const mc_model_config* cfg, mc_model** out) {
    MC_REQUIRE(cfg && out, "null argument");
    MC_REQUIRE(cfg->latent_dim == 32 || cfg->latent_dim == 64 || cfg->latent_dim == 128,
               "latent_dim=%d unsupported (32, 64, 128)", cfg->latent_dim);
    MC_REQUIRE(cfg->topk == 2, "topk=%d unsupported (reference configs use 2)", cfg->topk);
    MC_REQUIRE(cfg->num_experts >= 2 && cfg->num_experts <= 16, "num_experts=%d unsupported", cfg->num_experts);
    MC_REQUIRE(cfg->latent_dim % cfg->dyn_heads == 0, "latent_dim %% dyn_heads != 0");
    MC_REQUIRE(cfg->num_ctrl_layers >= 0 && cfg->num_ctrl_layers < cfg->num_layers, "copy_blocks_num=%d must be in [0, num_layers)", cfg->num_ctrl_layers);
    MC_REQUIRE(cfg->num_ctrl_layers == 0 || cfg->ctrl_cond_feats >= 1, "ctrl_cond_feats must be >= 1");
    MC_REQUIRE(cfg->ffn_dim % 4 == 0 && cfg->time_embed_dim % 4 == 0 && cfg->text_latent_dim % 4 == 0, "dims must be multiples of 4");
    {
        const int q = cfg->text_latent_dim / 4;
        MC_REQUIRE(q >= 1 && q <= 64 && (q & (q - 1)) == 0, "text_latent_dim=%d unsupported", cfg->text_latent_dim);
    }
    mc_model* m = new mc_model();
    m->cfg = *cfg;
    m->Cp = (cfg->input_feats + 31) / 32 * 32;
    *out = m;
    return MC_OK;
}

void mc_model_destroy(mc_model* m) {
    if (!m) return;
    for (auto& kv : m->params) (void)hipFree(kv.second.first);
    for (auto& kv : m->half) (void)hipFree(kv.second.hi);
    delete m;
}

int mc_model_set_param(mc_model* m, const char* name, const float* host, int64_t numel) {
    MC_REQUIRE(m && name && host && numel > 0, "bad argument");
    auto it = m->params.find(name);
    // replacing a weight of a finalized model: contexts hold raw pointers into the old allocation and the fp16 planes built from it
    // would go stale -- the model is immutable once contexts can exist (build a new model for new weights).  Checked BEFORE anything
    // is allocated (a rejected call must not leak the new buffer).
    MC_REQUIRE(it == m->params.end() || !m->finalized, "mc_model_set_param(%s): the model is finalized; weights are immutable from then on", name);
    float* d = nullptr;
    MC_HIP(hipMalloc(&d, (size_t)numel * sizeof(float)));
    if (hipMemcpy(d, host, (size_t)numel * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(d);
        mc_set_error("mc_model_set_param(%s): host-to-device copy failed", name);
        return MC_ERR_HIP;
    }
    if (it != m->params.end()) (void)hipFree(it->second.first);
    m->params[name] = std::make_pair(d, numel);
    return MC_OK;
}

int mc_model_finalize(mc_model* m) {
    MC_REQUIRE(m, "null model");
    mc_ctx probe;
    probe.m = m;
    int r = bind_weights(&probe);
    if (r != MC_OK) return r;
    m->finalized = true;
    return MC_OK;
}

int mc_ctx_create(mc_model* m, int32_t batch, int32_t frames, int32_t max_steps, mc_ctx** out) {
    MC_REQUIRE(m && out, "null argument");
    MC_REQUIRE(m->finalized, "model not finalized");
    MC_REQUIRE(batch >= 1 && frames >= 1 && frames <= m->cfg.max_seq_len, "bad batch/frames (%d, %d)", batch, frames);
    MC_REQUIRE(max_steps >= 1, "max_steps < 1");
    const mc_model_config& g = m->cfg;
    mc_ctx* c = new mc_ctx();
    if (const char* e = getenv("MC_HALF_MIN_ROWS")) c->half_min_rows = atol(e);      // (the tests lift it to run the fp16 kernels at their small sizes)
    if (const char* e = getenv("MC_GATE_SMALL")) c->gate_small_tokens = atol(e);
    if (const char* e = getenv("MC_SPLIT_EXPERT")) c->split_expert = atoi(e);
    if (const char* e = getenv("MC_SPLIT_SFFN")) c->split_sffn = atoi(e);
    if (const char* e = getenv("MC_ROUTE_REG")) c->rb.reg_kernel = atoi(e) != 0;
    if (const char* e = getenv("MC_ROUTE_COOP")) c->rb.coop = atoi(e) != 0;
    if (const char* e = getenv("MC_ROUTE_SMALL_CTX")) c->rb.small_pairs = atol(e);
    c->m = m;
    c->B = batch;
    c->T = frames;
    c->maxS = max_steps;
    const int L = g.latent_dim, H = g.num_parts, D = L * H, F = g.ffn_dim, Te = g.time_embed_dim, Dt = g.text_latent_dim;
    const long B2 = 2L * batch;
    c->rows = B2 * frames;
    c->N = c->rows * H;
    c->Ntxt = B2 * g.max_text_len;
    int r = bind_weights(c);
    if (r == MC_OK) r = build_ctx_weights(c);
    if (r != MC_OK) { mc_ctx_destroy(c); return r; }
    bool cand_ok = true;
    for (int k = 0; k < mc_ctx::SIDE_CAND; ++k) cand_ok = cand_ok && hipStreamCreateWithFlags(&c->side_cand[k], hipStreamNonBlocking) == hipSuccess;
    c->side = c->side_cand[0];
    if (!cand_ok ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_gate, hipEventDisableTiming) != hipSuccess) {
        mc_set_error("could not create the side stream / events");
        mc_ctx_destroy(c);
        return MC_ERR_HIP;
    }
    {
        const char* e = getenv("MC_SPLIT");
        c->nparts = e ? atoi(e) : 2;
        if (c->nparts < 2) c->nparts = 2;
        if (c->nparts > 4) c->nparts = 4;
        if (c->nparts > 2 * batch) c->nparts = 2 * batch;
        c->parts[0] = c->side;
        for (int k = 0; k < c->nparts - 1; ++k) {
            if ((k > 0 && hipStreamCreateWithFlags(&c->parts[k], hipStreamNonBlocking) != hipSuccess) ||
                hipEventCreateWithFlags(&c->ev_parts[k], hipEventDisableTiming) != hipSuccess) {
                mc_set_error("could not create the part streams / events");
                mc_ctx_destroy(c);
                return MC_ERR_HIP;
            }
        }
    }
    const long Nmax = c->N > c->Ntxt ? c->N : c->Ntxt;
    const size_t zsz = (size_t)(c->N * L > c->Ntxt * Dt ? c->N * L : c->Ntxt * Dt);
    const size_t hsz = (size_t)(2 * c->N * 4 * L > 2 * c->Ntxt * 4 * Dt ? 2 * c->N * 4 * L : 2 * c->Ntxt * 4 * Dt);
#define WS(p, n) do { if ((r = ws_alloc(c, &(p), (size_t)(n))) != MC_OK) { mc_ctx_destroy(c); return r; } } while (0)
    WS(c->h, c->rows * D);
    WS(c->z, zsz);
    WS(c->proj, Nmax * 256);
    WS(c->hbuf, hsz);
    c->hbuf_cap = hsz;
    if (chain_on(c, 0) && mc_mlp_supported(L, 4 * L)) c->hbuf_floats = hsz;   // (the text MoE only touches hbuf in set_condition)
    WS(c->y2, 2 * zsz);
    WS(c->mf, (c->N + 128) * 4 * L);        // + 128 padding rows: projqkv_k stores unconditionally (invalid lanes land there)
    WS(c->qkv, (c->N + 128) * 3 * L);
    WS(c->ys, c->rows * D);
    WS(c->yt, c->rows * D);
    WS(c->a, c->rows * D);
    WS(c->z2, c->rows * D);
    WS(c->fh, c->rows * H * F);
    WS(c->out2, c->rows * g.input_feats);
    WS(c->xpad, (long)batch * frames * m->Cp);
    WS(c->xfn, c->Ntxt * Dt);
    WS(c->mask_own, (long)batch * frames);
    WS(c->tf, (long)c->NLA * c->Ntxt * 2 * L);
    WS(c->t_orig, max_steps);
    WS(c->te, (long)max_steps * D);
    WS(c->e1, (long)max_steps * Te);
    WS(c->emb, (long)max_steps * Te);
    WS(c->semb, (long)max_steps * Te);
    WS(c->ss, (long)c->NLA * 2 * max_steps * 2 * D);
    if (g.num_ctrl_layers > 0) {
        WS(c->hc, c->rows * D);
        WS(c->cb, c->rows * D);
        WS(c->cenc, (long)batch * frames * D);
    }
    WS(c->rb.idx, 2 * Nmax);
    WS(c->rb.gate, 2 * Nmax);
    WS(c->rb.key, Nmax);
    WS(c->rb.comb_w, 2 * Nmax);
    WS(c->rb.src_row, 2 * Nmax);
    WS(c->rb.dst_row, 2 * Nmax);
    c->rb.max_tiles = cdiv(2 * Nmax, 128) + g.num_experts;
    WS(c->rb.tile_group, 2 * c->rb.max_tiles);
    WS(c->rb.tile_row0, 2 * c->rb.max_tiles);
    WS(c->rb.tile_nrows, 2 * c->rb.max_tiles);
    WS(c->rb.state, mc_route_state_ints(g.num_experts));
    MC_HIP(hipMemset(c->rb.state + mc_route_barrier_offset(), 0, mc_route_barrier_ints() * sizeof(int)));      // grid-barrier words of the cooperative routing kernel
    // the cooperative routing kernel needs its whole grid resident: reserve it out of what the device holds, or run the
    // launch sequence instead (no env var needed: a fifth concurrent B = 64 context, or a CPX partition, simply falls back)
    // Reserved for the LARGEST grid any routing call of this context can launch cooperatively: the motion MoE routes N tokens, the
    // text MoE (mc_ctx_set_condition) Ntxt -- either may be the one inside route_coop_k's size range.  Nothing to reserve (both in
    // the one-workgroup regime or beyond the kernel's range) or no room -> coop off for the context, so no unreserved grid can launch.
    MC_HIP(hipGetDevice(&c->device));
    if (c->rb.coop) {
        const int nwg = coop_grid_needed(c);
        if (nwg > 0 && mc_route_coop_reserve(c->device, nwg)) c->coop_reserved = nwg;
        else c->rb.coop = false;
    }
#undef WS
    *out = c;
    return MC_OK;
}

static void graph_release(mc_ctx* c);
static void prof_clear(mc_ctx* c);

void mc_ctx_destroy(mc_ctx* c) {
    if (!c) return;
    graph_release(c);
    if (c->coop_reserved) { mc_route_coop_release(c->device, c->coop_reserved); c->coop_reserved = 0; }
    prof_clear(c);
    for (int k = 0; k < mc_ctx::SIDE_CAND; ++k)
        if (c->side_cand[k]) { (void)hipStreamSynchronize(c->side_cand[k]); (void)hipStreamDestroy(c->side_cand[k]); }
    for (int k = 1; k < 3; ++k)
        if (c->parts[k]) { (void)hipStreamSynchronize(c->parts[k]); (void)hipStreamDestroy(c->parts[k]); }
    for (int k = 0; k < 3; ++k)
        if (c->ev_parts[k]) (void)hipEventDestroy(c->ev_parts[k]);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_gate) (void)hipEventDestroy(c->ev_gate);
    for (void* p : c->allocs) (void)hipFree(p);
    delete c;
}

int64_t mc_ctx_workspace_bytes(const mc_ctx* c) { return c ? (int64_t)(c->bytes + (c->prec != MC_PREC_F32 ? c->m->half_bytes : 0)) : 0; }

int mc_ctx_check(mc_ctx* c, void* stream) {
    MC_REQUIRE(c, "null context");
    int flag = 0;
    MC_HIP(hipMemcpyAsync(&flag, c->rb.state + mc_route_error_offset(), sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    MC_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (flag != 0) {
        mc_set_error("the cooperative routing kernel's grid barrier timed out (its workgroups were not all resident: another process "
                     "is holding the GPU with barrier kernels of its own); the results of this context are invalid -- destroy it and "
                     "create the context with MC_ROUTE_COOP=0");
        return MC_ERR_STATE;
    }
    return MC_OK;
}

int mc_ctx_effective_precision(const mc_ctx* c) { return c ? (use_half(c) ? c->prec : MC_PREC_F32) : MC_PREC_F32; }

int mc_ctx_uses_coop_routing(const mc_ctx* c) { return c && c->rb.coop && c->coop_reserved > 0 ? 1 : 0; }

static void prof_clear(mc_ctx* c) {
    for (auto& p : c->prof) { if (p.e0) (void)hipEventDestroy(p.e0); if (p.e1) (void)hipEventDestroy(p.e1); }
    c->prof.clear();
}

int mc_ctx_profile(mc_ctx* c, int32_t on) {
    MC_REQUIRE(c, "null context");
    MC_REQUIRE(!c->graph_mode, "profiling inside a graph capture");
    prof_clear(c);
    c->prof_on = on != 0;
    return MC_OK;
}

int mc_ctx_profile_read(mc_ctx* c, int64_t rows_filter, double* avg_us, int32_t* count, double* gflop_per_launch) {
    MC_REQUIRE(c && avg_us && count && gflop_per_launch, "null argument");
    const int D = c->m->cfg.latent_dim * c->m->cfg.num_parts;
    double sum = 0.0, rows = 0.0;
    int n = 0;
    for (auto& p : c->prof) {
        MC_HIP(hipEventSynchronize(p.e1));
        if (rows_filter > 0 && p.rows != rows_filter) continue;
        float ms = 0.f;
        MC_HIP(hipEventElapsedTime(&ms, p.e0, p.e1));
        sum += ms * 1e3;
        rows += (double)p.rows;
        ++n;
    }
    *count = n;
    *avg_us = n ? sum / n : 0.0;
    *gflop_per_launch = n ? 2.0 * (rows / n) * D * D / 1e9 : 0.0;
    return MC_OK;
}

int mc_ctx_set_tie_policy(mc_ctx* c, int32_t policy) {
    MC_REQUIRE(c, "null context");
    MC_REQUIRE(policy == MC_TIE_STABLE || policy == MC_TIE_REVERSE, "tie policy %d (MC_TIE_STABLE or MC_TIE_REVERSE)", policy);
    c->rb.tie_xor = policy == MC_TIE_STABLE ? 0xFFFFFFFFu : 0u;
    c->have_cond = false;      // the hoisted text K/V were routed under the previous policy: set the condition again
    return MC_OK;
}

int mc_ctx_set_precision(mc_ctx* c, int32_t precision) {
    MC_REQUIRE(c, "null context");
    MC_REQUIRE(precision == MC_PREC_F32 || precision == MC_PREC_F16 || precision == MC_PREC_F16X3, "precision %d", precision);
    if (precision != MC_PREC_F32) {
        int r = bind_half_weights(c);
        if (r != MC_OK) return r;
        if (!c->a_tail && (r = ws_alloc(c, &c->a_tail, (size_t)c->rows * c->m->cfg.latent_dim * c->m->cfg.num_parts)) != MC_OK) return r;
    }
    c->prec = precision;
    return MC_OK;
}

// Per-context kernel-selection switches (the MC_* environment variables only seed the defaults of contexts created later).
int mc_ctx_set_option(mc_ctx* c, const char* key, int64_t value) {
    MC_REQUIRE(c && key, "null argument");
    MC_REQUIRE(!c->graph_exec, "mc_ctx_set_option: a captured graph holds the old kernel selection (mc_ctx_graph_release first)");
    const std::string k(key);
    McOptions& o = c->opt;
    if (k == "chain") {
        o.chain = (int)value;
        c->hbuf_floats = (chain_on(c, 0) && mc_mlp_supported(c->m->cfg.latent_dim, 4 * c->m->cfg.latent_dim)) ? c->hbuf_cap : 0;
    } else if (k == "big_tokens") o.big_tokens = value;
    else if (k == "small_gemm_rows") o.small_gemm_rows = value;
    else if (k == "split_rows_expert") o.split_rows_expert = value;
    else if (k == "split_rows_sffn") o.split_rows_sffn = value;
    else if (k == "temporal_split") o.temporal_split = value;
    else if (k == "rowchain_split") o.rowchain_split = value;
    else if (k == "gemm_tune") o.gemm_tune = (int)value;
    else if (k == "small_tile_n") o.small_tile_n = (int)value;
    else if (k == "gemm_wp_grid") o.gemm_wp_grid = (int)value;
    else if (k == "half_min_rows") c->half_min_rows = value;
    else if (k == "dbg_delay_us") { MC_REQUIRE(value >= -100000 && value <= 100000, "dbg_delay_us: -100000 .. 100000"); c->dbg_delay_us = value; }
    else if (k == "gate_small") c->gate_small_tokens = value;
    else if (k == "split_expert") c->split_expert = (int)value;
    else if (k == "split_sffn") c->split_sffn = (int)value;
    else if (k == "route_reg") c->rb.reg_kernel = value != 0;
    else if (k == "route_per") { MC_REQUIRE(value == 0 || value == 10 || value == 16, "route_per: 0 (default), 10 or 16"); c->rb.coop_per = (int)value; }
    else if (k == "route_small") { MC_REQUIRE(value <= 131072, "route_small: the one-workgroup kernels hold at most 131072 pairs"); c->rb.small_pairs = value; }
    else if (k == "route_coop") {
        // off: give the reservation back; on: only if the grid can be reserved now
        if (!value) {
            if (c->coop_reserved) { mc_route_coop_release(c->device, c->coop_reserved); c->coop_reserved = 0; }
            c->rb.coop = false;
        } else if (!c->rb.coop) {
            const int nwg = coop_grid_needed(c);       // 0: no routing call of this context is in route_coop_k's size range (nothing to switch on)
            MC_REQUIRE(nwg == 0 || mc_route_coop_reserve(c->device, nwg), "route_coop: the cooperative routing grid cannot be reserved on this device");
            c->coop_reserved = nwg;
            c->rb.coop = nwg > 0;
        }
    } else if (k == "split_groups") {
        MC_REQUIRE(value >= 2 && value <= 4 && value <= 2 * c->B, "split_groups: 2..4 groups of whole samples (batch %d)", c->B);
        for (int j = 1; j < (int)value - 1; ++j) {
            if (!c->parts[j]) MC_HIP(hipStreamCreateWithFlags(&c->parts[j], hipStreamNonBlocking));
        }
        for (int j = 0; j < (int)value - 1; ++j) {
            if (!c->ev_parts[j]) MC_HIP(hipEventCreateWithFlags(&c->ev_parts[j], hipEventDisableTiming));
        }
        c->nparts = (int)value;
    } else {
        mc_set_error("mc_ctx_set_option: unknown key '%s'", key);
        return MC_ERR_ARG;
    }
    return MC_OK;
}

int mc_ctx_enable_capture(mc_ctx* c) {
    MC_REQUIRE(c, "null context");
    if (c->cap_idx) return MC_OK;
    int r;
    if ((r = ws_alloc(c, &c->cap_idx, (size_t)c->NLA * 2 * c->N)) != MC_OK) return r;
    return ws_alloc(c, &c->cap_w, (size_t)c->NLA * 2 * c->N);
}

// time_embed (diffusion_transformer.py:89-93,206-208) and every StylizationBlock.emb_layers
// (stylization_block.py:17-20,34-35) depend only on the timestep, which is identical for the whole
// batch -> evaluated once for all S steps of the schedule as M = S row GEMMs.
int mc_ctx_set_timesteps(mc_ctx* c, const int32_t* t_orig_host, int32_t S, void* stream) {
    MC_REQUIRE(c && t_orig_host, "null argument");
    MC_REQUIRE(S >= 1 && S <= c->maxS, "num_steps=%d exceeds context max_steps=%d", S, c->maxS);
    hipStream_t s = (hipStream_t)stream;
    const mc_model_config& g = c->m->cfg;
    const int D = g.latent_dim * g.num_parts, Te = g.time_embed_dim;
    MC_HIP(hipMemcpyAsync(c->t_orig, t_orig_host, sizeof(int) * S, hipMemcpyHostToDevice, s));
    MC_HIP(hipStreamSynchronize(s));  // t_orig_host may be a temporary of the caller
    int r;
    if ((r = mc_launch_timestep_embedding(c->t_orig, c->te, S, D, s))) return r;
    if ((r = dense(c, c->te, D, c->time_w0, D, c->time_b0, nullptr, 0, c->e1, Te, S, Te, D, ACT_SILU, s))) return r;
    if ((r = dense(c, c->e1, Te, c->time_w2, Te, c->time_b2, nullptr, 0, c->emb, Te, S, Te, Te, ACT_NONE, s))) return r;
    if ((r = mc_launch_silu(c->emb, c->semb, (long)S * Te, s))) return r;
    for (int i = 0; i < c->NLA; ++i) {
        const LayerW& w = c->lw[i];
        float* ss0 = c->ss + ((long)(i * 2 + 0) * c->maxS) * 2 * D;
        float* ss1 = c->ss + ((long)(i * 2 + 1) * c->maxS) * 2 * D;
        if ((r = dense(c, c->semb, Te, w.ca_film_w, Te, w.ca_film_b, nullptr, 0, ss0, 2 * D, S, 2 * D, Te, ACT_NONE, s))) return r;
        if ((r = dense(c, c->semb, Te, w.ffn_film_w, Te, w.ffn_film_b, nullptr, 0, ss1, 2 * D, S, 2 * D, Te, ACT_NONE, s))) return r;
    }
    c->S = S;
    return MC_OK;
}

// Step-invariant text K/V of every layer (st_attention.py:116-118): text_moe over the CFG-doubled
// condition batch (the MoE capacity couples both halves, so both are routed together).
int mc_ctx_set_condition(mc_ctx* c, const float* xf_out_dev, const float* mask_dev, void* stream) {
    MC_REQUIRE(c && xf_out_dev && mask_dev, "null argument");
    hipStream_t s = (hipStream_t)stream;
    const mc_model_config& g = c->m->cfg;
    const int L = g.latent_dim, Dt = g.text_latent_dim, Nt = g.max_text_len;
    const long half = (long)c->B * Nt;
    MC_HIP(hipMemcpyAsync(c->mask_own, mask_dev, sizeof(float) * c->B * c->T, hipMemcpyDeviceToDevice, s));
    c->mask = c->mask_own;
    int r;
    for (int i = 0; i < c->NLA; ++i) {
        const LayerW& w = c->lw[i];
        if ((r = mc_launch_ln_rows(xf_out_dev, Dt, 0, w.tnorm_g, w.tnorm_b, w.tm.emb, Nt, c->xfn, Dt, half, Dt, s))) return r;
        MC_HIP(hipMemcpyAsync(c->xfn + half * Dt, c->xfn, sizeof(float) * half * Dt, hipMemcpyDeviceToDevice, s));
        if ((r = run_moe(c, w.tm, c->xfn, c->Ntxt, c->tf + (long)i * c->Ntxt * 2 * L, 2 * L, false, false, c->Ntxt, s))) return r;
    }
    c->have_cond = true;
    return MC_OK;
}

// ControlT2MHalf.forward_c (controlnet.py:186-199) with condition_pre_encoder = identity, followed by
// `c * all_cond_type` (controlnet.py:377-379) and controlnet[0].before_proj (controlnet.py:66): all
// step-invariant, evaluated once per condition batch.
int mc_ctx_set_control(mc_ctx* c, const float* c_feat_dev, int32_t Tc, void* stream) {
    MC_REQUIRE(c, "null context");
    const mc_model_config& g = c->m->cfg;
    if (!c_feat_dev) { c->have_ctrl = false; return MC_OK; }
    MC_REQUIRE(g.num_ctrl_layers > 0, "the model has no control branch");
    MC_REQUIRE(Tc >= 1 && Tc <= c->T, "control length %d outside [1, %d]", Tc, c->T);
    hipStream_t s = (hipStream_t)stream;
    const int D = g.latent_dim * g.num_parts, Fc = g.ctrl_cond_feats;
    const long BT = (long)c->B * c->T;
    int r;
    MC_HIP(hipMemsetAsync(c->cenc, 0, sizeof(float) * BT * D, s));           // zero padding for t >= Tc
    {
        GemmArgs e;
        gemm_opts(c, e);   // per sample: [Tc, Fc] x [D, Fc]^T + bias + sequence_embedding[:Tc]
        e.A = c_feat_dev; e.lda = Fc; e.a_gstride = (long)Tc * Fc;
        e.W = c->ctrl_in_w; e.ldw = (Fc + 3) / 4 * 4; e.bias = c->ctrl_in_b;
        e.add = c->seq_emb; e.add_mod = Tc; e.ld_add = D;
        e.C = c->cenc; e.ldc = D; e.c_gstride = (long)c->T * D; e.dup_rows = 0;
        e.M = Tc; e.N = D; e.K = Fc;
        if ((r = mc_launch_gemm(GM_ENC, e, c->B, 0, s))) return r;
    }
    const LayerW& w0 = c->lw[g.num_layers];
    // text-conditioned half: before_proj(c); unconditional half: before_proj(c * 0) = bias when condition_cfg
    if ((r = dense(c, c->cenc, D, w0.before_w, D, w0.before_b, nullptr, 0, c->cb, D, BT, D, D, ACT_NONE, s))) return r;
    if (g.ctrl_condition_cfg) {
        if ((r = mc_launch_add_rows(c->cb + BT * D, nullptr, nullptr, w0.before_b, BT, D, s))) return r;
    } else {
        MC_HIP(hipMemcpyAsync(c->cb + BT * D, c->cb, sizeof(float) * BT * D, hipMemcpyDeviceToDevice, s));
    }
    c->have_ctrl = true;
    return MC_OK;
}

static int denoise_impl(mc_ctx* c, const float* x_t, int32_t step, float* out2_dev, int32_t stop_after, void* stream,
                        const SeedArgs* seed);

int mc_denoise(mc_ctx* c, const float* x_t, int32_t step, float* out2_dev, int32_t stop_after, void* stream) {
    return denoise_impl(c, x_t, step, out2_dev, stop_after, stream, nullptr);
}

// `seed` != nullptr: x_t is overwritten in place on its first frames before the network reads it (mc_sample_step_seeded)
static int denoise_impl(mc_ctx* c, const float* x_t, int32_t step, float* out2_dev, int32_t stop_after, void* stream,
                        const SeedArgs* seed) {
    MC_REQUIRE(c && x_t, "null argument");
    MC_REQUIRE(c->have_cond, "mc_ctx_set_condition not called");
    MC_REQUIRE(step >= 0 && step < c->S, "step_index %d outside the %d-step schedule", step, c->S);
    if (c->graph_mode) step = 0;       // host-side table pointers address step 0; the kernels add *gstep rows
    hipStream_t s = (hipStream_t)stream;
    const mc_model_config& g = c->m->cfg;
    const int L = g.latent_dim, H = g.num_parts, D = L * H, C = g.input_feats;
    const long BT = (long)c->B * c->T;
    int r;
    if ((r = pick_side_stream(c, s))) return r;        // once per caller stream: the side stream that really runs beside it (hardware queues)
    // PoseEncoder as one dense [C -> D] GEMM with the scattered weight, + sequence_embedding[:T],
    // written to both CFG halves (stmogen.py:336-353; diffusion_transformer.py:215-218; stmogen.py:740)
    {
        const bool padded = c->xpad_ready && !seed;        // mc_sample_loop: the previous step's sampler update wrote the rows already
        c->xpad_ready = false;
        if (seed) { if ((r = mc_launch_pad_rows_seeded(const_cast<float*>(x_t), c->xpad, BT, C, c->m->Cp, *seed, s))) return r; }
        else if (!padded && (r = mc_launch_pad_rows(x_t, c->xpad, BT, C, c->m->Cp, s))) return r;
        GemmArgs e;
        gemm_opts(c, e);
        e.A = c->xpad; e.lda = c->m->Cp;
        e.W = c->enc_w; e.ldw = c->m->Cp; e.bias = c->enc_b;
        e.add = c->seq_emb; e.add_mod = c->T; e.ld_add = D;
        e.C = c->h; e.ldc = D; e.dup_rows = BT;
        e.M = (int)BT; e.N = D; e.K = c->m->Cp;
        if (BT <= small_gemm_rows(c) && D % 64 == 0 && c->m->Cp % 32 == 0) { if ((r = mc_launch_gemm_small(e, s))) return r; }
        else if ((r = mc_launch_gemm(GM_ENC, e, 1, 0, s))) return r;
    }
    const int nl = stop_after >= 0 ? (stop_after < g.num_layers ? stop_after : g.num_layers) : g.num_layers;
    c->no_alias = stop_after >= 0 && stop_after < g.num_layers;
    const int NC = c->have_ctrl ? g.num_ctrl_layers : 0;
    // large batches: the CFG halves run on two streams (see run_layer); with a control branch the extra whole-batch
    // ops between layers need both halves, so the halves re-join after every layer
    const bool fused = chain_on(c, 1) && chain_on(c, 2) && mc_mlp_supported(L, 32);
    const int split = (c->side && chain_on(c, 5) && c->N > c->opt.big_tokens && fused) ? ((NC > 0 && !chain_on(c, 9)) ? 1 : 2) : 0;
    // row-wise op between layers, on the stream of the sample group that owns the rows (one launch when not split)
    auto by_group = [&](auto&& fn) -> int {
        if (split != 2) return fn(0L, c->rows, s);
        for (int k = 0; k < c->nparts; ++k) {
            const long r0 = part_row0(c, k);
            if (int e = fn(r0, part_row0(c, k + 1) - r0, part_stream(c, k, s))) return e;
        }
        return MC_OK;
    };
    for (int i = 0; i < nl; ++i) {
        // ControlT2MHalf.forward_test (controlnet.py:372-413): base block 0, then for index 1..copy:
        //   c, c_skip = controlnet[index-1](x=h, c=c);  h = base[index](h + c_skip)
        if (i >= 1 && i <= NC) {
            const int j = i - 1, slot = g.num_layers + j;
            if (j == 0) {                                                                              // x + before_proj(c)
                if ((r = by_group([&](long r0, long n, hipStream_t sk) {
                         return mc_launch_add_rows(c->hc + r0 * D, c->h + r0 * D, c->cb + r0 * D, nullptr, n, D, sk); })))
                    return r;
            }
            if ((r = run_layer(c, slot, c->hc, step, false, split, s))) return r;                       // copied_block
            const LayerW& cw = c->lw[slot];
            if ((r = by_group([&](long r0, long n, hipStream_t sk) {                                   // h += after_proj(c)
                     if (use_half(c) && cw.h_after.hi)
                         return dense_h(c, c->hc + r0 * D, cw.h_after, cw.after_b, c->h + r0 * D, c->h + r0 * D, n, D, D, sk);
                     return dense(c, c->hc + r0 * D, D, cw.after_w, D, cw.after_b, c->h + r0 * D, D, c->h + r0 * D, D, n, D, D, ACT_NONE, sk); })))
                return r;
        }
        if ((r = run_layer(c, i, c->h, step, i == 0, split, s))) return r;
    }
    if (split == 2 && (r = parts_join(c, s))) return r;      // the groups meet again before the pose decoder
    if (stop_after >= 0) return MC_OK;
    // PoseDecoder as one dense [D -> C] GEMM (stmogen.py:505-544), /2 folded into the packed weight
    float* o = out2_dev ? out2_dev : c->out2;
    return dense(c, c->h, D, c->dec_w, D, c->dec_b, nullptr, 0, o, C, c->rows, C, D, ACT_NONE, s);
}

static SamplerCoefs to_coefs(const mc_step_coefs* k) {
    SamplerCoefs c;
    c.mode = k->mode; c.text_coef = k->text_coef; c.none_coef = k->none_coef; c.c1 = k->c1; c.c2 = k->c2;
    c.log_var = k->log_var; c.sqrt_recip = k->sqrt_recip; c.sqrt_recipm1 = k->sqrt_recipm1; c.ab = k->ab;
    c.ab_prev = k->ab_prev; c.eta = k->eta; c.nonzero = k->nonzero;
    return c;
}

// The pose decoder is affine, so  w dec(h_text) + (1 - w) dec(h_none) = dec(w h_text + (1 - w) h_none):  the sampler
// entry points combine the two CFG halves of the residual stream first and decode B*T rows instead of 2*B*T
// (mc_denoise, which hands out both decoded halves, keeps the reference's order).
// -> x0 = *x0a (+ *x0b when it is not null: the two partial products of the folded tail, summed by the sampler kernel)
static int denoise_combined(mc_ctx* c, const float* x_t, int32_t step, const mc_step_coefs* k, void* stream, const float** x0a,
                            const float** x0b, const SeedArgs* seed = nullptr) {
    const mc_model_config& g = c->m->cfg;
    // ... and so is the Linear of the very last StylizationBlock (h += a W^T + b): it, too, runs once on the combined
    // rows  h_c = comb(h) + comb(a) W^T + b  instead of on both halves.
    const bool defer = chain_on(c, 7);
    c->defer_last_gemm = defer;
    int r = denoise_impl(c, x_t, step, nullptr, g.num_layers, stream, seed);   // all layers (minus that GEMM), no decoder
    c->defer_last_gemm = false;
    if (r != MC_OK) return r;
    hipStream_t s = (hipStream_t)stream;
    const int D = g.latent_dim * g.num_parts, C = g.input_feats;
    const long BT = (long)c->B * c->T;
    const LayerW& w = c->lw[g.num_layers - 1];
    *x0b = nullptr;
    auto combine = [&](const float* x, const float* y, float* out) -> int {      // w x + (1 - w) y, w = this step's CFG weight
        if (c->graph_mode) return mc_launch_cfg_combine_tab(x, y, c->gcoefs, c->gstep, out, BT * D, s);
        return mc_launch_axpby(x, y, k->text_coef, k->none_coef, out, BT * D, s);
    };
    if (defer && c->dec_cat_w && chain_on(c, 11) && chain_on(c, 21) && BT > c->opt.small_gemm_rows && D % 32 == 0) {
        // large batches: CFG combination, both K groups and the biases in ONE GEMM pass (gemm_tail_k): no axpby_pair_k, no partial outputs
        TailArgs t;
        t.H = c->h; t.Af = deferred_a(c); t.half = BT * D; t.lda = D;
        t.W = c->dec_cat_w; t.ldw = D; t.w_gstride = (long)C * D; t.bias = c->dec_cat_b; t.b_gstride = C;
        t.C = c->out2; t.ldc = C; t.M = (int)BT; t.N = C; t.K = D;
        t.wc = k->text_coef; t.wu = k->none_coef;
        if (c->graph_mode) {
            static_assert(sizeof(SamplerCoefs) % sizeof(float) == 0 && offsetof(SamplerCoefs, none_coef) == offsetof(SamplerCoefs, text_coef) + sizeof(float), "SamplerCoefs layout");
            t.coef_table = reinterpret_cast<const float*>(c->gcoefs) + offsetof(SamplerCoefs, text_coef) / sizeof(float);
            t.coef_stride = sizeof(SamplerCoefs) / sizeof(float);
            t.step_ptr = c->gstep;
        }
        t.tune = options_of(c).gemm_tune;
        t.C2 = c->out2 + BT * C;               // (out2 holds [2 B T][C]: room for one partial product per K group)
        if ((r = mc_launch_gemm_tail(t, s))) return r;
        *x0a = c->out2;
        if (mc_gemm_tail_two_outputs(t)) *x0b = t.C2;      // gemm_tail2_k: x0 = C + C2, added by the sampler-update kernel
        return MC_OK;
    }
    if (defer && c->dec_cat_w && chain_on(c, 11)) {
        // h_c and a_c in one launch
        if ((r = mc_launch_axpby_pair(c->h, c->h + BT * D, c->z2, deferred_a(c), deferred_a(c) + BT * D, c->z2 + BT * D, k->text_coef, k->none_coef,
                                      c->graph_mode ? c->gcoefs : nullptr, c->graph_mode ? c->gstep : nullptr, BT * D, s))) return r;
    } else if ((r = combine(c->h, c->h + BT * D, c->z2))) return r;     // h_c
    if (defer) {
        if (c->dec_cat_w && chain_on(c, 11)) {
            // ... and that Linear composed with the decoder is one [C, D] matrix (folded at pack time):
            //   x0 = [dec(h_c) + Wd b] + [a_c (Wd W)^T]
            // the two skinny products (N = C = 322: 294 tiles each, half a chip) are the two groups of ONE grouped GEMM
            // over (h_c | a_c) x (Wd | Wd W); the sampler kernel adds the two partial outputs
            GemmArgs t;
            gemm_opts(c, t);
            t.A = c->z2; t.lda = D; t.a_gstride = BT * D; t.W = c->dec_cat_w; t.ldw = D; t.w_gstride = (long)C * D;
            t.bias = c->dec_cat_b; t.b_gstride = C; t.C = c->out2; t.ldc = C; t.c_gstride = BT * C;
            t.M = (int)BT; t.N = C; t.K = D;
            if (D % 32 == 0) { if ((r = mc_launch_gemm_small(t, s, 2))) return r; }
            else if ((r = mc_launch_gemm(GM_PLAIN, t, 2, 0, s))) return r;
            *x0a = c->out2;
            *x0b = c->out2 + BT * C;
            return MC_OK;
        }
        float* const ad = deferred_a(c);
        if ((r = combine(ad, ad + BT * D, ad))) return r;      // a_c
        if (c->dec_wf) {
            if ((r = dense(c, c->z2, D, c->dec_w, D, c->dec_bf, nullptr, 0, c->out2, C, BT, C, D, ACT_NONE, s))) return r;
            if ((r = dense(c, ad, D, c->dec_wf, D, nullptr, c->out2, C, c->out2, C, BT, C, D, ACT_NONE, s))) return r;
            *x0a = c->out2;
            return MC_OK;
        }
        if ((r = dense(c, ad, D, w.ffn_out_w, D, w.ffn_out_b, c->z2, D, c->z2, D, BT, D, D, ACT_NONE, s))) return r;  // h_c += a_c W^T + b
    }
    if ((r = dense(c, c->z2, D, c->dec_w, D, c->dec_b, nullptr, 0, c->out2, C, BT, C, D, ACT_NONE, s))) return r;
    *x0a = c->out2;
    return MC_OK;
}

static SamplerCoefs combined_coefs(const mc_step_coefs* k, bool two_parts) {
    SamplerCoefs sc = to_coefs(k);
    sc.text_coef = 1.f;                  // x0 is already the CFG-combined prediction ...
    sc.none_coef = two_parts ? 1.f : 0.f;   // ... or the sum of its two partial products
    return sc;
}

int mc_sample_step(mc_ctx* c, const float* x_t, int32_t step, const mc_step_coefs* k, const float* noise,
                   float* x_prev, float* x0, void* stream) {
    MC_REQUIRE(c && x_t && k && noise && x_prev, "null argument");
    const float *x0a = nullptr, *x0b = nullptr;
    int r = denoise_combined(c, x_t, step, k, stream, &x0a, &x0b);
    if (r != MC_OK) return r;
    const long n = (long)c->B * c->T * c->m->cfg.input_feats;
    return mc_launch_sampler_update(x_t, x0a, x0b ? x0b : x0a, noise, x_prev, x0, n, combined_coefs(k, x0b != nullptr), (hipStream_t)stream,
                                    c->graph_mode ? c->gcoefs : nullptr, c->graph_mode ? c->gstep : nullptr);
}

static RngArgs rng_args(uint64_t seed, uint64_t draw) {
    RngArgs r;
    r.seed_lo = (uint32_t)seed; r.seed_hi = (uint32_t)(seed >> 32); r.draw_lo = (uint32_t)draw; r.draw_hi = (uint32_t)(draw >> 32);
    return r;
}

// The whole p_sample_loop / ddim_sample_loop of the reference (gaussian_diffusion.py:698-797, 925-1049) as ONE call: x is updated in
// place through the schedule indices step_indices[0 .. num_steps) (the reference walks num_timesteps-1 .. 0), no return to the host
// language between steps.  The per-step randn_like (gaussian_diffusion.py:684, 847) is either read from noise_dev
// [num_steps][B,T,C] (parity runs on the reference's seeds) or, noise_dev == NULL, drawn inside the sampler-update kernel
// (Philox4x32-10, draw index noise_draw0 + k): no noise tensor, no generator launch.
int mc_sample_loop(mc_ctx* c, float* x, const int32_t* step_indices, const mc_step_coefs* coefs, int32_t num_steps,
                   const float* noise, uint64_t seed, uint64_t noise_draw0, float* x0_last, void* stream) {
    MC_REQUIRE(c && x && step_indices && coefs, "null argument");
    MC_REQUIRE(num_steps >= 0, "num_steps < 0");
    const long n = (long)c->B * c->T * c->m->cfg.input_feats;
    // the sampler update of step k also writes x_{t-1} at the padded row stride the pose-encoder GEMM of step k + 1 stages from
    // (chain bit 19: no pad_rows_k launch between the steps; the first step of a loop pads, which also zeroes the pad columns)
    PadOut po;
    po.xpad = c->xpad; po.C = c->m->cfg.input_feats; po.Cp = c->m->Cp;
    const bool fold_pad = chain_on(c, 19) && c->xpad;
    c->xpad_ready = false;
    for (int k = 0; k < num_steps; ++k) {
        const float *x0a = nullptr, *x0b = nullptr;
        int r = denoise_combined(c, x, step_indices[k], &coefs[k], stream, &x0a, &x0b);
        if (r != MC_OK) { c->xpad_ready = false; return r; }
        const RngArgs rng = rng_args(seed, noise_draw0 + (uint64_t)k);
        const bool more = fold_pad && k + 1 < num_steps;
        r = mc_launch_sampler_update(x, x0a, x0b ? x0b : x0a, noise ? noise + (long)k * n : nullptr, x, k + 1 == num_steps ? x0_last : nullptr, n,
                                     combined_coefs(&coefs[k], x0b != nullptr), (hipStream_t)stream, nullptr, nullptr, noise ? nullptr : &rng,
                                     more ? &po : nullptr);
        if (r != MC_OK) { c->xpad_ready = false; return r; }
        c->xpad_ready = more;
    }
    c->xpad_ready = false;
    return MC_OK;
}

int mc_op_philox_normal(float* out, uint32_t* bits, int64_t n, uint64_t seed, uint64_t draw, void* stream) {
    MC_REQUIRE(out || bits, "null argument");
    return mc_launch_philox_fill(out, bits, n, rng_args(seed, draw), (hipStream_t)stream);
}

// ---- hipGraph replay of mc_sample_step -------------------------------------------------------------------------------
static void graph_release(mc_ctx* c) {
    if (c->graph_exec) { (void)hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }
    if (c->graph) { (void)hipGraphDestroy(c->graph); c->graph = nullptr; }
    c->graph_steps = 0;
}

int mc_ctx_graph_capture(mc_ctx* c, float* x_dev, const float* noise_dev, const mc_step_coefs* coefs_host, int32_t num_steps,
                         void* stream) {
    MC_REQUIRE(c && x_dev && noise_dev && coefs_host, "null argument");
    MC_REQUIRE(c->have_cond, "mc_ctx_set_condition not called");
    MC_REQUIRE(num_steps >= 1 && num_steps == c->S, "num_steps=%d must equal the schedule set by mc_ctx_set_timesteps (%d)", num_steps, c->S);
    MC_REQUIRE(!c->cap_idx, "routing capture (tests) and graph replay are mutually exclusive");
    hipStream_t s = (hipStream_t)stream;
    MC_REQUIRE(s != nullptr, "graph capture needs a non-default stream");
    graph_release(c);
    int r;
    if (!c->gstep && (r = ws_alloc(c, &c->gstep, 1)) != MC_OK) return r;
    if (!c->gcoefs && (r = ws_alloc(c, &c->gcoefs, (size_t)c->maxS)) != MC_OK) return r;
    std::vector<SamplerCoefs> tab(num_steps);
    for (int i = 0; i < num_steps; ++i) tab[i] = to_coefs(&coefs_host[i]);
    MC_HIP(hipMemcpyAsync(c->gcoefs, tab.data(), sizeof(SamplerCoefs) * num_steps, hipMemcpyHostToDevice, s));
    MC_HIP(hipMemsetAsync(c->gstep, 0, sizeof(int), s));
    MC_HIP(hipStreamSynchronize(s));                 // (`tab` is a temporary)
    MC_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    c->graph_mode = true;
    c->cnt_clean = false;        // a replay must not assume anything about the routing counts it starts from: the first gate clears them
    r = mc_sample_step(c, x_dev, 0, &coefs_host[0], noise_dev, x_dev, nullptr, stream);     // in place: x_prev aliases x_t
    c->graph_mode = false;
    c->cnt_clean = false;        // (nothing of the captured step ran)
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(s, &g);
    if (r != MC_OK) { if (g) (void)hipGraphDestroy(g); return r; }
    if (e != hipSuccess || !g) { mc_set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return MC_ERR_HIP; }
    c->graph = g;
    const hipError_t ei = hipGraphInstantiate(&c->graph_exec, g, nullptr, nullptr, 0);
    if (ei != hipSuccess) { mc_set_error("hipGraphInstantiate: %s", hipGetErrorString(ei)); graph_release(c); return MC_ERR_HIP; }
    c->graph_steps = num_steps;
    return MC_OK;
}

int mc_ctx_graph_step(mc_ctx* c, int32_t step, void* stream) {
    MC_REQUIRE(c && c->graph_exec, "no captured graph (mc_ctx_graph_capture)");
    MC_REQUIRE(step >= 0 && step < c->graph_steps, "step_index %d outside the %d captured steps", step, c->graph_steps);
    hipStream_t s = (hipStream_t)stream;
    int r = mc_launch_set_int(c->gstep, step, s);
    if (r != MC_OK) return r;
    MC_HIP(hipGraphLaunch(c->graph_exec, s));
    return MC_OK;
}

int mc_ctx_graph_release(mc_ctx* c) {
    MC_REQUIRE(c, "null context");
    graph_release(c);
    return MC_OK;
}

int mc_sample_step_seeded(mc_ctx* c, float* x_t, int32_t step, const mc_step_coefs* k, const float* noise,
                          const mc_seed* sd, float* x_prev, float* x0, void* stream) {
    MC_REQUIRE(c && x_t && k && noise && x_prev && sd, "null argument");
    MC_REQUIRE(sd->pre_len >= 0 && sd->pre_len <= c->T, "pre_seq has %d frames, the window %d", sd->pre_len, c->T);
    MC_REQUIRE(sd->num_transl >= 0 && sd->num_transl <= MC_MAX_TRANSL, "num_transl=%d outside [0, %d]", sd->num_transl, MC_MAX_TRANSL);
    MC_REQUIRE(sd->num_transl == 0 || c->T >= 2, "transl_req seeds frames 0 and 1: the window has %d", c->T);
    SeedArgs a;
    a.pre = sd->pre_seq_dev; a.pre_noise = sd->pre_noise_dev; a.pre_len = sd->pre_len; a.T = c->T;
    a.sqrt_ab = sd->sqrt_ab; a.sqrt_1mab = sd->sqrt_1mab; a.num_transl = sd->num_transl;
    for (int i = 0; i < sd->num_transl; ++i) {
        MC_REQUIRE(sd->transl_channel[i] >= 0 && sd->transl_channel[i] < c->m->cfg.input_feats, "transl_req channel %d", sd->transl_channel[i]);
        a.transl_channel[i] = sd->transl_channel[i];
        a.transl_value[i][0] = sd->transl_value[i][0];
        a.transl_value[i][1] = sd->transl_value[i][1];
    }
    const float *x0a = nullptr, *x0b = nullptr;
    int r = denoise_combined(c, x_t, step, k, stream, &x0a, &x0b, &a);
    if (r != MC_OK) return r;
    const long n = (long)c->B * c->T * c->m->cfg.input_feats;
    return mc_launch_sampler_update(x_t, x0a, x0b ? x0b : x0a, noise, x_prev, x0, n, combined_coefs(k, x0b != nullptr), (hipStream_t)stream);
}

int mc_sample_step_inpaint(mc_ctx* c, const float* x_t, int32_t step, const mc_step_coefs* k, const float* noise,
                           const mc_inpaint* ip, float* x_prev, float* x0, void* stream) {
    MC_REQUIRE(c && x_t && k && noise && x_prev && ip, "null argument");
    const float *x0a = nullptr, *x0b = nullptr;
    int r = denoise_combined(c, x_t, step, k, stream, &x0a, &x0b);
    if (r != MC_OK) return r;
    const int C = c->m->cfg.input_feats;
    const long n = (long)c->B * c->T * C;
    InpaintArgs a;
    a.gt = ip->gt_dev; a.keep = ip->keep_dev; a.gt_noise = ip->gt_noise_dev; a.blend_w = ip->blend_w_dev;
    a.blend_len = ip->blend_len; a.T = c->T; a.C = C;
    MC_REQUIRE(a.blend_len >= 0 && a.blend_len <= c->T, "blend_len %d outside the %d-frame window", a.blend_len, c->T);
    return mc_launch_sampler_inpaint(x_t, x0a, x0b ? x0b : x0a, noise, a, x_prev, x0, n, combined_coefs(k, x0b != nullptr), (hipStream_t)stream);
}

int mc_postprocess_smplx(const float* pred, const int32_t* lengths, const double* mean, const double* stdv,
                         const double* taps, const int32_t radius[4], int32_t stats_f32, int32_t B, int32_t T, int32_t C,
                         double* poses, double* expr, double* trans, void* stream) {
    MC_REQUIRE(pred && mean && stdv && taps && radius && poses && expr && trans, "null argument");
    static_assert(MC_POST_MAXTAP == 129, "header / kernel tap table size");
    MC_REQUIRE(mc_smplx_post_maxtap() == MC_POST_MAXTAP, "tap table size mismatch");
    return mc_launch_smplx_post(pred, lengths, nullptr, mean, stdv, taps, radius, stats_f32, B, T, C, poses, expr, trans,
                                (hipStream_t)stream);
}

int mc_postprocess_smplx_stitched(const float* pred, const int32_t* rows, int32_t n_frames, const double* mean,
                                  const double* stdv, const double* taps, const int32_t radius[4], int32_t stats_f32,
                                  int32_t C, double* poses, double* expr, double* trans, void* stream) {
    MC_REQUIRE(pred && rows && mean && stdv && taps && radius && poses && expr && trans && n_frames >= 0, "bad argument");
    MC_REQUIRE(mc_smplx_post_maxtap() == MC_POST_MAXTAP, "tap table size mismatch");
    return mc_launch_smplx_post(pred, nullptr, rows, mean, stdv, taps, radius, stats_f32, 1, n_frames, C, poses, expr,
                                trans, (hipStream_t)stream);
}

int mc_op_renoise(const float* x, const float* noise, float a, float b, float* out, int64_t n, void* stream) {
    MC_REQUIRE(x && noise && out && n >= 0, "bad argument");
    return mc_launch_axpby(x, noise, a, b, out, (long)n, (hipStream_t)stream);
}

int mc_ctx_get_buffer(mc_ctx* c, const char* name, int32_t layer, void** dev_ptr, int64_t* numel) {
    MC_REQUIRE(c && name && dev_ptr && numel, "null argument");
    const mc_model_config& g = c->m->cfg;
    const int L = g.latent_dim, H = g.num_parts, D = L * H;
    const std::string n(name);
    void* p = nullptr;
    int64_t cnt = 0;
    if (n == "h") { p = c->h; cnt = c->rows * D; }
    else if (n == "z") { p = c->z; cnt = c->N * L; }
    else if (n == "proj") { p = c->proj; cnt = c->N * 256; }
    else if (n == "mf") { p = c->mf; cnt = c->N * 4 * L; }
    else if (n == "qkv") { p = c->qkv; cnt = c->N * 3 * L; }
    else if (n == "y2") { p = c->y2; cnt = c->N * 2 * L; }
    else if (n == "ys") { p = c->ys; cnt = c->rows * D; }
    else if (n == "yt") { p = c->yt; cnt = c->rows * D; }
    else if (n == "a") {
        // reduced-precision contexts keep fp16 hi | lo PLANES in `a` (film_block): not the fp32 [rows][D] rows this call promises
        MC_REQUIRE(!a_holds_planes(c), "buffer 'a' holds fp16 planes in a reduced-precision context (clear chain bit 17 to read fp32 rows)");
        p = c->a; cnt = c->rows * D;
    }
    else if (n == "a_tail") { p = deferred_a(c); cnt = c->rows * D; }
    else if (n == "z2") { p = c->z2; cnt = c->rows * D; }
    else if (n == "out2") { p = c->out2; cnt = c->rows * g.input_feats; }
    else if (n == "emb") { p = c->emb; cnt = (int64_t)c->S * g.time_embed_dim; }
    else if (n == "ss") { p = c->ss + (long)layer * c->maxS * 2 * D; cnt = (int64_t)c->S * 2 * D; }
    else if (n == "tf") { p = c->tf + (long)layer * c->Ntxt * 2 * L; cnt = c->Ntxt * 2 * L; }
    else if (n == "idx") { p = c->rb.idx; cnt = 2 * c->N; }
    else if (n == "gate") { p = c->rb.gate; cnt = 2 * c->N; }
    else if (n == "comb_w") { p = c->rb.comb_w; cnt = 2 * c->N; }
    else if (n == "key") { p = c->rb.key; cnt = c->N; }
    else if (n == "hc" && c->hc) { p = c->hc; cnt = c->rows * D; }
    else if (n == "cb" && c->cb) { p = c->cb; cnt = c->rows * D; }
    else if (n == "cap_idx" && c->cap_idx) { p = c->cap_idx + (long)layer * 2 * c->N; cnt = 2 * c->N; }
    else if (n == "cap_w" && c->cap_w) { p = c->cap_w + (long)layer * 2 * c->N; cnt = 2 * c->N; }
    else { mc_set_error("unknown buffer '%s'", name); return MC_ERR_ARG; }
    *dev_ptr = p;
    *numel = cnt;
    return MC_OK;
}

int mc_op_gemm(const float* a, const float* w, const float* bias, const float* res, float* cdev, int32_t M, int32_t N,
               int32_t K, int32_t ldw, int32_t act, void* stream) {
    MC_REQUIRE(a && w && cdev && M > 0 && N > 0 && K > 0 && K % 4 == 0 && ldw % 4 == 0 && ldw >= K, "bad gemm args");
    return dense(nullptr, a, K, w, ldw, bias, res, N, cdev, N, M, N, K, act, (hipStream_t)stream);
}

int mc_op_gemm_f16(const float* a, const float* w, const float* bias, const float* res, float* cdev, int32_t M, int32_t N,
                   int32_t K, int32_t split, void* stream) {
    MC_REQUIRE(a && w && cdev && M > 0 && N > 0 && K > 0, "bad gemm args");
    hipStream_t s = (hipStream_t)stream;
    mc_half* planes = nullptr;
    MC_HIP(hipMalloc((void**)&planes, sizeof(mc_half) * 2 * (size_t)N * K));
    int r = mc_launch_split_f16(w, planes, planes + (size_t)N * K, (long)N * K, s);
    if (r == MC_OK) {
        GemmHArgs g;
        g.A = a; g.lda = K; g.Wh = planes; g.Wl = planes + (size_t)N * K; g.bias = bias; g.R = res; g.ldr = N; g.C = cdev; g.ldc = N;
        g.M = M; g.N = N; g.K = K;
        r = mc_launch_gemm_h(g, split != 0, s);
    }
    (void)hipStreamSynchronize(s);
    (void)hipFree(planes);
    return r;
}

int mc_op_gemm_tail(const float* h, const float* a, const float* w, const float* bias, float* cdev, float* c2dev, int32_t M, int32_t N,
                    int32_t K, float wc, float wu, int32_t variant, void* stream) {
    MC_REQUIRE(h && a && w && bias && cdev && M > 0 && N > 0 && K > 0 && K % 32 == 0, "bad gemm_tail args");
    MC_REQUIRE(variant >= 0 && variant <= 2, "gemm_tail variant %d (0 default, 1 column tiles, 2 block ranges)", variant);
    TailArgs t;
    t.H = h; t.Af = a; t.half = (long)M * K; t.lda = K; t.W = w; t.ldw = K; t.w_gstride = (long)N * K; t.bias = bias; t.b_gstride = N;
    t.C = cdev; t.ldc = N; t.M = M; t.N = N; t.K = K; t.wc = wc; t.wu = wu;
    t.tune = options_of(nullptr).gemm_tune;
    if (variant) {
        if (t.tune < 0) t.tune = mc_gemm_default_tune();       // MC_GEMM_TUNE, or the built-in default
        t.tune = variant == 1 ? (t.tune & ~1024) : (t.tune | 1024);
    }
    t.C2 = c2dev;
    MC_REQUIRE(variant != 2 || c2dev, "gemm_tail variant 2 needs the scratch output c2_dev [M][N]");
    MC_REQUIRE(variant != 2 || mc_gemm_tail_two_outputs(t),
               "gemm_tail variant 2: the block-range form is not eligible for M=%d N=%d K=%d (N <= 336, K %% 16 == 0, K >= 48, <= 17 blocks per wave)", M, N, K);
    int r = mc_launch_gemm_tail(t, (hipStream_t)stream);
    if (r == MC_OK && mc_gemm_tail_two_outputs(t))       // the sampler-update kernel adds the two partial products in the step; here: C += C2
        r = mc_launch_axpby(cdev, c2dev, 1.f, 1.f, cdev, (long)M * N, (hipStream_t)stream);
    return r;
}

int mc_op_ln_rows(const float* x, int64_t ldx, const float* gamma, const float* beta, const float* add, int32_t add_mod,
                  float* y, int64_t rows, int32_t L, void* stream) {
    return mc_launch_ln_rows(x, ldx, 0, gamma, beta, add, add_mod, y, L, rows, L, (hipStream_t)stream);
}

int mc_op_sampler_update(const float* x_t, const float* ot, const float* on, const float* noise, float* x_prev, float* x0,
                         int64_t n, const mc_step_coefs* k, void* stream) {
    MC_REQUIRE(x_t && ot && on && noise && x_prev && k, "null argument");
    return mc_launch_sampler_update(x_t, ot, on, noise, x_prev, x0, n, to_coefs(k), (hipStream_t)stream);
}

}  // extern "C"
