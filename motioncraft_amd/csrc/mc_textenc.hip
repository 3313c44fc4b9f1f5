// Text condition encoders on the device, step-invariant, run once per prompt batch:
//   stage B (optional)  CLIP ViT-B/32 text transformer: token + positional embedding, 12 pre-LN residual attention
//                       blocks (causal mask, QuickGELU MLP), ln_final          diffusion_transformer.py:144-151
//                       (architecture of the un-vendored `clip` package, openai/CLIP clip/model.py: ResidualAttentionBlock)
//   stage A             text_pre_proj (512 -> text_latent_dim) -> nn.TransformerEncoder (post-LN, GELU, no mask)
//                       -> text_ln = xf_out [B, 77, text_latent_dim]           diffusion_transformer.py:109-141,156-158
// Both stages are the same generic encoder-layer schedule over [B*77, d] rows: the Linear layers are launches of the
// fp32 MFMA GEMM (bias / GELU / QuickGELU / residual fused in its epilogue), LayerNorm is a wave-per-row kernel and the
// 77 x 77 attention of one (sample, head) is one workgroup with K/V in LDS.
#include "mc_common.h"
#include "mc_gemm.h"
#include "mc_kernels.h"
#include "mc_enc.h"
#include "../../include/motioncraft_amd.h"
#include <map>
#include <string>
#include <vector>

namespace {

// LayerNorm over rows of L floats (L % 4 == 0, L <= 4096): one wavefront per row, two-pass variance.
__global__ __launch_bounds__(256) void ln_wide_k(const float* __restrict__ X, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, float* __restrict__ Y, long rows, int L,
                                                 float eps, int relu) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int n4 = L >> 2;
    const float* x = X + r * L;
    float s = 0.f;
    for (int i = lane; i < n4; i += 64) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * i);
        s += v[0] + v[1] + v[2] + v[3];
    }
    const float mean = group_sum(s, 64) / (float)L;
    float q = 0.f;
    for (int i = lane; i < n4; i += 64) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * i);
#pragma unroll
        for (int j = 0; j < 4; ++j) q += (v[j] - mean) * (v[j] - mean);
    }
    const float rstd = rsqrtf(group_sum(q, 64) / (float)L + eps);
    for (int i = lane; i < n4; i += 64) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * i);
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 4 * i);
        const f32x4 b = *reinterpret_cast<const f32x4*>(beta + 4 * i);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[j] = (v[j] - mean) * rstd * g[j] + b[j];
            if (relu) o[j] = fmaxf(o[j], 0.f);
        }
        *reinterpret_cast<f32x4*>(Y + r * L + 4 * i) = o;
    }
}

// x[b][s][:] = token_embedding[ids[b][s]][:] + positional_embedding[s][:]
__global__ __launch_bounds__(256) void embed_tokens_k(const int* __restrict__ ids, const float* __restrict__ emb,
                                                      const float* __restrict__ pos, float* __restrict__ X, long rows,
                                                      int S, int d, int vocab) {
    const long n4 = rows * (d >> 2);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long r = i / (d >> 2);
        const int c = (int)(i % (d >> 2)) * 4;
        int id = ids[r];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        const f32x4 e = *reinterpret_cast<const f32x4*>(emb + (long)id * d + c);
        const f32x4 p = *reinterpret_cast<const f32x4*>(pos + (long)(r % S) * d + c);
        *reinterpret_cast<f32x4*>(X + r * d + c) = f32x4{e[0] + p[0], e[1] + p[1], e[2] + p[2], e[3] + p[3]};
    }
}

// Multi-head attention over a short sequence (S <= 128, head_dim 64): one workgroup per (sample, head).
// qkv [B*S][3d] (torch in_proj order q | k | v), out [B*S][d].  K rows are padded to 65 floats (lane j reads row j:
// conflict-free), V rows are read with lane = channel.  Wave w takes queries w, w+4, ...
constexpr int MHA_S = 128, MHA_HD = 64;
__global__ __launch_bounds__(256) void mha_small_k(const float* __restrict__ qkv, float* __restrict__ out, int S, int d,
                                                   int heads, int causal) {
    __shared__ float Ks[MHA_S * (MHA_HD + 1)];
    __shared__ float Vs[MHA_S * MHA_HD];
    __shared__ float Qs[4][MHA_HD];
    __shared__ float Ps[4][MHA_S];
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* base = qkv + (long)b * S * 3 * d + h * MHA_HD;
    for (int i = tid; i < S * MHA_HD; i += 256) {
        const int s = i / MHA_HD, c = i % MHA_HD;
        Ks[s * (MHA_HD + 1) + c] = base[(long)s * 3 * d + d + c];
        Vs[s * MHA_HD + c] = base[(long)s * 3 * d + 2 * d + c];
    }
    __syncthreads();
    const float scale = 0.125f;                       // 1 / sqrt(64)
    for (int q = wave; q < S; q += 4) {
        Qs[wave][lane] = base[(long)q * 3 * d + lane] * scale;     // torch scales q before the product
        __builtin_amdgcn_wave_barrier();
        float sc[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = lane + 64 * u;
            float a = -INFINITY;
            if (j < S && !(causal && j > q)) {
                a = 0.f;
#pragma unroll 16
                for (int c = 0; c < MHA_HD; ++c) a += Qs[wave][c] * Ks[j * (MHA_HD + 1) + c];
            }
            sc[u] = a;
        }
        const float m = group_max(fmaxf(sc[0], sc[1]), 64);
        const float e0 = sc[0] == -INFINITY ? 0.f : expf(sc[0] - m), e1 = sc[1] == -INFINITY ? 0.f : expf(sc[1] - m);
        const float inv = 1.f / group_sum(e0 + e1, 64);
        Ps[wave][lane] = e0 * inv;
        Ps[wave][lane + 64] = e1 * inv;
        __builtin_amdgcn_wave_barrier();
        float o = 0.f;
        for (int j = 0; j < S; ++j) o += Ps[wave][j] * Vs[j * MHA_HD + lane];
        out[((long)b * S + q) * d + h * MHA_HD + lane] = o;
        __builtin_amdgcn_wave_barrier();
    }
}

// General form: any S, optional key mask, optional causal mask.  One workgroup per (sample, head, block of 16 queries);
// keys are streamed through LDS 64 at a time with the running-max ("online") softmax, so that LDS use is independent of
// S.  Wave w owns queries 4w .. 4w+3 of the block; lane = key within the chunk for the scores, lane = channel for P V.
constexpr int MHA_QB = 16;
__global__ __launch_bounds__(256) void mha_masked_k(const float* __restrict__ qkv, const uint8_t* __restrict__ valid,
                                                    float* __restrict__ out, int S, int d, int heads, int causal) {
    __shared__ float Ks[64 * (MHA_HD + 1)];
    __shared__ float Vs[64 * MHA_HD];
    __shared__ float Qs[MHA_QB][MHA_HD];
    __shared__ float Ps[4][64];
    const int b = blockIdx.x / heads, h = blockIdx.x % heads, q0 = blockIdx.y * MHA_QB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* base = qkv + (long)b * S * 3 * d + h * MHA_HD;
    for (int i = tid; i < MHA_QB * MHA_HD; i += 256) {
        const int qi = i / MHA_HD, c = i % MHA_HD, q = q0 + qi;
        Qs[qi][c] = q < S ? base[(long)q * 3 * d + c] * 0.125f : 0.f;
    }
    float m[4], l[4], o[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { m[u] = -INFINITY; l[u] = 0.f; o[u] = 0.f; }
    for (int k0 = 0; k0 < S; k0 += 64) {
        __syncthreads();
        for (int i = tid; i < 64 * MHA_HD; i += 256) {
            const int j = i / MHA_HD, c = i % MHA_HD, key = k0 + j;
            const bool in = key < S;
            const long off = (long)(in ? key : 0) * 3 * d + c;
            const float kv = base[off + d], vv = base[off + 2 * d];
            Ks[j * (MHA_HD + 1) + c] = in ? kv : 0.f;
            Vs[j * MHA_HD + c] = in ? vv : 0.f;
        }
        __syncthreads();
        const int key = k0 + lane;
        const bool ok = key < S && (valid == nullptr || valid[(long)b * S + (key < S ? key : 0)] != 0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int qi = wave * 4 + u, q = q0 + qi;
            float a = 0.f;
#pragma unroll 16
            for (int c = 0; c < MHA_HD; ++c) a += Qs[qi][c] * Ks[lane * (MHA_HD + 1) + c];
            const bool use = ok && !(causal && key > q);
            a = use ? a : -INFINITY;
            const float mn = fmaxf(m[u], group_max(a, 64));
            const float p = use ? expf(a - mn) : 0.f;
            const float alpha = m[u] == -INFINITY ? 0.f : expf(m[u] - mn);
            l[u] = l[u] * alpha + group_sum(p, 64);
            Ps[wave][lane] = p;
            __builtin_amdgcn_wave_barrier();
            float acc = 0.f;
#pragma unroll 16
            for (int j = 0; j < 64; ++j) acc += Ps[wave][j] * Vs[j * MHA_HD + lane];
            o[u] = o[u] * alpha + acc;
            m[u] = mn;
            __builtin_amdgcn_wave_barrier();
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int q = q0 + wave * 4 + u;
        if (q < S) out[((long)b * S + q) * d + h * MHA_HD + lane] = l[u] > 0.f ? o[u] / l[u] : 0.f;
    }
}

}  // namespace

int mc_enc_dense(const float* A, long lda, const float* W, long ldw, const float* bias, const float* R, long ldr, float* C, long ldc,
                 long M, int N, int K, int act, hipStream_t s) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.R = R; g.ldr = ldr;
    g.C = C; g.ldc = ldc; g.M = (int)M; g.N = N; g.K = K; g.act = act;
    return mc_launch_gemm(GM_PLAIN, g, 1, 0, s);
}

int mc_enc_ln(const float* X, const float* g, const float* b, float* Y, long rows, int L, float eps, int relu, hipStream_t s) {
    MC_REQUIRE(L % 4 == 0, "layer norm width %d", L);
    hipLaunchKernelGGL(ln_wide_k, dim3(cdiv(rows, 4)), dim3(256), 0, s, X, g, b, Y, rows, L, eps, relu);
    MC_LAUNCH_CHECK();
    return MC_OK;
}

int mc_enc_embed_tokens(const int* ids, const float* emb, const float* pos, float* X, long rows, int S, int d, int vocab,
                        hipStream_t s) {
    MC_REQUIRE(d % 4 == 0, "embedding width %d", d);
    hipLaunchKernelGGL(embed_tokens_k, dim3(cdiv(rows * (d / 4), 256)), dim3(256), 0, s, ids, emb, pos, X, rows, S, d, vocab);
    MC_LAUNCH_CHECK();
    return MC_OK;
}

int mc_enc_layer(const EncLayer& p, float* x, float* qkv, float* att, float* y, float* hid, long rows, int B, int S, int d,
                 int heads, int ff, bool pre_ln, int act, int causal, const uint8_t* valid, float eps, hipStream_t s) {
    MC_REQUIRE(d == heads * MHA_HD, "encoder layer: width %d with %d heads (head_dim must be %d)", d, heads, MHA_HD);
    int r;
    const float* src = x;
    if (pre_ln) {
        if ((r = mc_enc_ln(x, p.n1_g, p.n1_b, y, rows, d, eps, 0, s))) return r;
        src = y;
    }
    if ((r = mc_enc_dense(src, d, p.in_w, d, p.in_b, nullptr, 0, qkv, 3 * d, rows, 3 * d, d, ACT_NONE, s))) return r;
    if (valid == nullptr && S <= MHA_S)
        hipLaunchKernelGGL(mha_small_k, dim3(B * heads), dim3(256), 0, s, qkv, att, S, d, heads, causal);
    else
        hipLaunchKernelGGL(mha_masked_k, dim3(B * heads, cdiv(S, MHA_QB)), dim3(256), 0, s, qkv, valid, att, S, d, heads, causal);
    MC_LAUNCH_CHECK();
    if (pre_ln) {
        if ((r = mc_enc_dense(att, d, p.out_w, d, p.out_b, x, d, x, d, rows, d, d, ACT_NONE, s))) return r;   // x += out_proj(att)
        if ((r = mc_enc_ln(x, p.n2_g, p.n2_b, y, rows, d, eps, 0, s))) return r;
        if ((r = mc_enc_dense(y, d, p.l1_w, d, p.l1_b, nullptr, 0, hid, ff, rows, ff, d, act, s))) return r;
        return mc_enc_dense(hid, ff, p.l2_w, ff, p.l2_b, x, d, x, d, rows, d, ff, ACT_NONE, s);                // x += c_proj(...)
    }
    if ((r = mc_enc_dense(att, d, p.out_w, d, p.out_b, x, d, y, d, rows, d, d, ACT_NONE, s))) return r;       // y = x + out_proj(att)
    if ((r = mc_enc_ln(y, p.n1_g, p.n1_b, x, rows, d, eps, 0, s))) return r;
    if ((r = mc_enc_dense(x, d, p.l1_w, d, p.l1_b, nullptr, 0, hid, ff, rows, ff, d, act, s))) return r;
    if ((r = mc_enc_dense(hid, ff, p.l2_w, ff, p.l2_b, x, d, y, d, rows, d, ff, ACT_NONE, s))) return r;      // y = x + linear2(...)
    return mc_enc_ln(y, p.n2_g, p.n2_b, x, rows, d, eps, 0, s);
}

struct mc_textenc {
    mc_textenc_config cfg;
    std::map<std::string, std::pair<float*, int64_t>> params;
    std::vector<EncLayer> ft, clip;
    const float *pre_w = nullptr, *pre_b = nullptr, *ln_g = nullptr, *ln_b = nullptr;
    const float *tok = nullptr, *pos = nullptr, *lnf_g = nullptr, *lnf_b = nullptr;
    bool finalized = false, has_clip = false;
    float* ws = nullptr;
    size_t ws_floats = 0;
};

namespace {

int getp(mc_textenc* e, const std::string& name, int64_t numel, const float** out) {
    auto it = e->params.find(name);
    if (it == e->params.end()) { mc_set_error("text encoder: missing parameter '%s'", name.c_str()); return MC_ERR_STATE; }
    if (it->second.second != numel) {
        mc_set_error("text encoder: parameter '%s' has %ld elements, expected %ld", name.c_str(), (long)it->second.second, (long)numel);
        return MC_ERR_STATE;
    }
    *out = it->second.first;
    return MC_OK;
}

int ensure_ws(mc_textenc* e, size_t floats, hipStream_t s) {
    if (floats <= e->ws_floats) return MC_OK;
    if (e->ws) { MC_HIP(hipStreamSynchronize(s)); MC_HIP(hipFree(e->ws)); e->ws = nullptr; e->ws_floats = 0; }
    MC_HIP(hipMalloc((void**)&e->ws, floats * sizeof(float)));
    e->ws_floats = floats;
    return MC_OK;
}

}  // namespace

extern "C" {

int mc_textenc_create(const mc_textenc_config* cfg, mc_textenc** out) {
    MC_REQUIRE(cfg && out, "null argument");
    MC_REQUIRE(cfg->max_len >= 1 && cfg->max_len <= MHA_S, "text encoder: max_len=%d unsupported (<= %d)", cfg->max_len, MHA_S);
    MC_REQUIRE(cfg->text_latent_dim == cfg->num_heads * MHA_HD && cfg->clip_dim % 4 == 0 && cfg->ff_size % 4 == 0,
               "text encoder: text_latent_dim=%d with %d heads unsupported (head_dim must be 64)", cfg->text_latent_dim, cfg->num_heads);
    MC_REQUIRE(cfg->clip_layers == 0 || cfg->clip_dim == cfg->clip_heads * MHA_HD, "text encoder: clip width %d / %d heads unsupported",
               cfg->clip_dim, cfg->clip_heads);
    mc_textenc* e = new mc_textenc();
    e->cfg = *cfg;
    *out = e;
    return MC_OK;
}

void mc_textenc_destroy(mc_textenc* e) {
    if (!e) return;
    for (auto& kv : e->params) (void)hipFree(kv.second.first);
    if (e->ws) (void)hipFree(e->ws);
    delete e;
}

int mc_textenc_set_param(mc_textenc* e, const char* name, const float* host, int64_t numel) {
    MC_REQUIRE(e && name && host && numel > 0, "bad argument");
    float* d = nullptr;
    MC_HIP(hipMalloc((void**)&d, (size_t)numel * sizeof(float)));
    MC_HIP(hipMemcpy(d, host, (size_t)numel * sizeof(float), hipMemcpyHostToDevice));
    auto it = e->params.find(name);
    if (it != e->params.end()) (void)hipFree(it->second.first);
    e->params[name] = {d, numel};
    e->finalized = false;
    return MC_OK;
}

int mc_textenc_finalize(mc_textenc* e) {
    MC_REQUIRE(e, "null encoder");
    const mc_textenc_config& c = e->cfg;
    const int d = c.text_latent_dim, ff = c.ff_size;
    int r;
#define TP(ptr, name, n) if ((r = getp(e, (name), (int64_t)(n), &(ptr)))) return r
    e->pre_w = e->pre_b = nullptr;
    if (c.clip_dim != d) {                       // text_pre_proj is nn.Identity when the widths agree (:124-127)
        TP(e->pre_w, "text_pre_proj.weight", (int64_t)d * c.clip_dim);
        TP(e->pre_b, "text_pre_proj.bias", d);
    }
    e->ft.assign(c.num_layers, EncLayer());
    for (int i = 0; i < c.num_layers; ++i) {
        const std::string p = "textTransEncoder.layers." + std::to_string(i) + ".";
        EncLayer& L = e->ft[i];
        TP(L.in_w, p + "self_attn.in_proj_weight", (int64_t)3 * d * d);  TP(L.in_b, p + "self_attn.in_proj_bias", 3 * d);
        TP(L.out_w, p + "self_attn.out_proj.weight", (int64_t)d * d);    TP(L.out_b, p + "self_attn.out_proj.bias", d);
        TP(L.l1_w, p + "linear1.weight", (int64_t)ff * d);               TP(L.l1_b, p + "linear1.bias", ff);
        TP(L.l2_w, p + "linear2.weight", (int64_t)d * ff);               TP(L.l2_b, p + "linear2.bias", d);
        TP(L.n1_g, p + "norm1.weight", d);  TP(L.n1_b, p + "norm1.bias", d);
        TP(L.n2_g, p + "norm2.weight", d);  TP(L.n2_b, p + "norm2.bias", d);
    }
    TP(e->ln_g, "text_ln.weight", d);
    TP(e->ln_b, "text_ln.bias", d);
    e->has_clip = false;
    if (c.clip_layers > 0 && e->params.count("clip.token_embedding.weight")) {
        const int w = c.clip_dim, cf = c.clip_ff;
        TP(e->tok, "clip.token_embedding.weight", (int64_t)c.vocab * w);
        TP(e->pos, "clip.positional_embedding", (int64_t)c.max_len * w);
        TP(e->lnf_g, "clip.ln_final.weight", w);
        TP(e->lnf_b, "clip.ln_final.bias", w);
        e->clip.assign(c.clip_layers, EncLayer());
        for (int i = 0; i < c.clip_layers; ++i) {
            const std::string p = "clip.transformer.resblocks." + std::to_string(i) + ".";
            EncLayer& L = e->clip[i];
            TP(L.in_w, p + "attn.in_proj_weight", (int64_t)3 * w * w);  TP(L.in_b, p + "attn.in_proj_bias", 3 * w);
            TP(L.out_w, p + "attn.out_proj.weight", (int64_t)w * w);    TP(L.out_b, p + "attn.out_proj.bias", w);
            TP(L.l1_w, p + "mlp.c_fc.weight", (int64_t)cf * w);         TP(L.l1_b, p + "mlp.c_fc.bias", cf);
            TP(L.l2_w, p + "mlp.c_proj.weight", (int64_t)w * cf);       TP(L.l2_b, p + "mlp.c_proj.bias", w);
            TP(L.n1_g, p + "ln_1.weight", w);  TP(L.n1_b, p + "ln_1.bias", w);
            TP(L.n2_g, p + "ln_2.weight", w);  TP(L.n2_b, p + "ln_2.bias", w);
        }
        e->has_clip = true;
    }
#undef TP
    e->finalized = true;
    return MC_OK;
}

// clip_feat_dev [B, max_len, clip_dim] (= ln_final output of the CLIP text transformer) -> xf_out_dev [B, max_len, text_latent_dim]
int mc_textenc_forward_feat(mc_textenc* e, const float* clip_feat, int32_t B, float* xf_out, void* stream) {
    MC_REQUIRE(e && clip_feat && xf_out && B >= 1, "bad argument");
    MC_REQUIRE(e->finalized, "text encoder not finalized");
    hipStream_t s = (hipStream_t)stream;
    const mc_textenc_config& c = e->cfg;
    const int S = c.max_len, d = c.text_latent_dim, ff = c.ff_size;
    const long rows = (long)B * S;
    int r;
    if ((r = ensure_ws(e, (size_t)rows * (d + 3 * d + d + d + ff) + 256, s))) return r;
    float* x = e->ws;
    float* qkv = x + rows * d;
    float* att = qkv + rows * 3 * d;
    float* y = att + rows * d;
    float* hid = y + rows * d;
    if (e->pre_w) {
        if ((r = mc_enc_dense(clip_feat, c.clip_dim, e->pre_w, c.clip_dim, e->pre_b, nullptr, 0, x, d, rows, d, c.clip_dim, ACT_NONE, s))) return r;
    } else {
        MC_HIP(hipMemcpyAsync(x, clip_feat, (size_t)rows * d * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    for (int i = 0; i < c.num_layers; ++i)
        if ((r = mc_enc_layer(e->ft[i], x, qkv, att, y, hid, rows, B, S, d, c.num_heads, ff, false, ACT_GELU, 0, nullptr, 1e-5f, s))) return r;
    return mc_enc_ln(x, e->ln_g, e->ln_b, xf_out, rows, d, 1e-5f, 0, s);
}

// tokens_dev int32 [B, max_len] (clip.tokenize output) -> clip_feat_out_dev (optional) and xf_out_dev
int mc_textenc_forward_tokens(mc_textenc* e, const int32_t* tokens, int32_t B, float* clip_feat_out, float* xf_out, void* stream) {
    MC_REQUIRE(e && tokens && xf_out && B >= 1, "bad argument");
    MC_REQUIRE(e->finalized, "text encoder not finalized");
    MC_REQUIRE(e->has_clip, "text encoder: no clip.* weights were loaded (pass clip_feat to mc_textenc_forward_feat instead)");
    hipStream_t s = (hipStream_t)stream;
    const mc_textenc_config& c = e->cfg;
    const int S = c.max_len, w = c.clip_dim, cf = c.clip_ff, d = c.text_latent_dim;
    const long rows = (long)B * S;
    int r;
    const size_t stageA = (size_t)rows * (d + 3 * d + d + d + c.ff_size) + 256;
    const size_t stageB = (size_t)rows * (w + 3 * w + w + w + cf + w) + 256;
    if ((r = ensure_ws(e, stageA + stageB, s))) return r;
    float* x = e->ws + stageA;
    float* qkv = x + rows * w;
    float* att = qkv + rows * 3 * w;
    float* y = att + rows * w;
    float* hid = y + rows * w;
    float* feat = hid + rows * cf;
    if ((r = mc_enc_embed_tokens(tokens, e->tok, e->pos, x, rows, S, w, c.vocab, s))) return r;
    for (int i = 0; i < c.clip_layers; ++i)
        if ((r = mc_enc_layer(e->clip[i], x, qkv, att, y, hid, rows, B, S, w, c.clip_heads, cf, true, ACT_QUICKGELU, 1, nullptr, 1e-5f, s))) return r;
    float* f = clip_feat_out ? clip_feat_out : feat;
    if ((r = mc_enc_ln(x, e->lnf_g, e->lnf_b, f, rows, w, 1e-5f, 0, s))) return r;
    return mc_textenc_forward_feat(e, f, B, xf_out, stream);
}

}  // extern "C"
