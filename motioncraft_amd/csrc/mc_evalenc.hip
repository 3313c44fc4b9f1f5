// Evaluation embedding model on the device: T2MContrastiveModel_SMPLX
//   encode_motion   ActorAgnosticEncoder: skel_embedding -> [mu_token, logvar_token, frames] + sinusoid table ->
//                   post-LN nn.TransformerEncoder with the length mask as key-padding mask -> row 0 (mu = .loc)
//                                                                mogen/models/rnns/t2m_bigru_smplx.py:66-195,404-410
//   encode_text     DistilbertActorAgnosticEncoder after tokenisation: DistilBERT (embeddings LN, post-LN layers, eps
//                   1e-12, attention mask) -> ReLU -> Linear -> the same token/transformer tail
//                                                                mogen/models/rnns/t2m_bigru_smplx.py:198-394,412-414
// Both run on the generic encoder-layer schedule of mc_enc.h (fp32 MFMA GEMMs + masked streaming attention).
#include "mc_common.h"
#include "mc_gemm.h"
#include "mc_kernels.h"
#include "mc_enc.h"
#include "../../include/motioncraft_amd.h"
#include <map>
#include <string>
#include <vector>

namespace {

// xs[b][0] = mu_token + pe[0]; xs[b][1] = logvar_token + pe[1]; xs[b][2 + t] = rows[b][t] (pe already added by the GEMM)
// valid[b][0..1] = 1; valid[b][2 + t] = t < lengths[b]  (or mask_in[b][t])
__global__ __launch_bounds__(256) void assemble_tokens_k(const float* __restrict__ rows, const float* __restrict__ mu_tok,
                                                         const float* __restrict__ lv_tok, const float* __restrict__ pe,
                                                         const int* __restrict__ lengths, const uint8_t* __restrict__ mask_in,
                                                         float* __restrict__ xs, uint8_t* __restrict__ valid, int B, int S, int d) {
    const int S2 = S + 2, d4 = d >> 2;
    const long n = (long)B * S2 * d4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % d4) * 4;
        const long r = i / d4;
        const int b = (int)(r / S2), t = (int)(r % S2);
        f32x4 v;
        if (t < 2) {
            const f32x4 tk = *reinterpret_cast<const f32x4*>((t == 0 ? mu_tok : lv_tok) + c);
            const f32x4 p = *reinterpret_cast<const f32x4*>(pe + (long)t * d + c);
            v = f32x4{tk[0] + p[0], tk[1] + p[1], tk[2] + p[2], tk[3] + p[3]};
        } else {
            v = *reinterpret_cast<const f32x4*>(rows + ((long)b * S + (t - 2)) * d + c);
        }
        *reinterpret_cast<f32x4*>(xs + r * d + c) = v;
        if (c == 0) valid[r] = t < 2 ? 1 : (lengths ? (uint8_t)((t - 2) < lengths[b]) : (uint8_t)(mask_in[(long)b * S + (t - 2)] != 0));
    }
}

__global__ __launch_bounds__(256) void relu_k(const float* __restrict__ X, float* __restrict__ Y, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = reinterpret_cast<const f32x4*>(X)[i];
        reinterpret_cast<f32x4*>(Y)[i] = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
    }
}

struct Tail {                       // mu/logvar tokens + sinusoid table + seqTransEncoder
    const float *mu = nullptr, *lv = nullptr, *pe = nullptr;
    std::vector<EncLayer> layers;
};

}  // namespace

struct mc_evalenc {
    mc_evalenc_config cfg;
    std::map<std::string, std::pair<float*, int64_t>> params;
    std::vector<float*> owned;       // buffers packed at finalize
    Tail motion, text;
    const float *skel_w = nullptr, *skel_b = nullptr;        // skel_w zero padded to [d][Cp]
    const float *tok = nullptr, *pos = nullptr, *eln_g = nullptr, *eln_b = nullptr, *proj_w = nullptr, *proj_b = nullptr;
    std::vector<EncLayer> bert;
    int Cp = 0;
    bool finalized = false, has_text = false;
    float* ws = nullptr;
    size_t ws_floats = 0;
};

namespace {

int getp(mc_evalenc* e, const std::string& name, int64_t numel, const float** out) {
    auto it = e->params.find(name);
    if (it == e->params.end()) { mc_set_error("evaluation encoder: missing parameter '%s'", name.c_str()); return MC_ERR_STATE; }
    if (it->second.second != numel) {
        mc_set_error("evaluation encoder: parameter '%s' has %ld elements, expected %ld", name.c_str(), (long)it->second.second, (long)numel);
        return MC_ERR_STATE;
    }
    *out = it->second.first;
    return MC_OK;
}

int ensure_ws(mc_evalenc* e, size_t floats, hipStream_t s) {
    if (floats <= e->ws_floats) return MC_OK;
    if (e->ws) { MC_HIP(hipStreamSynchronize(s)); MC_HIP(hipFree(e->ws)); e->ws = nullptr; e->ws_floats = 0; }
    MC_HIP(hipMalloc((void**)&e->ws, floats * sizeof(float)));
    e->ws_floats = floats;
    return MC_OK;
}

#define TP(ptr, name, n) if ((r = getp(e, (name), (int64_t)(n), &(ptr)))) return r

int bind_tail(mc_evalenc* e, const std::string& pre, Tail& t) {
    const mc_evalenc_config& c = e->cfg;
    const int d = c.latent_dim, ff = c.ff_size;
    int r;
    TP(t.mu, pre + "mu_token", d);
    TP(t.lv, pre + "logvar_token", d);
    TP(t.pe, pre + "sequence_pos_encoding.pe", (int64_t)c.pe_len * d);
    t.layers.assign(c.num_layers, EncLayer());
    for (int i = 0; i < c.num_layers; ++i) {
        const std::string p = pre + "seqTransEncoder.layers." + std::to_string(i) + ".";
        EncLayer& L = t.layers[i];
        TP(L.in_w, p + "self_attn.in_proj_weight", (int64_t)3 * d * d);  TP(L.in_b, p + "self_attn.in_proj_bias", 3 * d);
        TP(L.out_w, p + "self_attn.out_proj.weight", (int64_t)d * d);    TP(L.out_b, p + "self_attn.out_proj.bias", d);
        TP(L.l1_w, p + "linear1.weight", (int64_t)ff * d);               TP(L.l1_b, p + "linear1.bias", ff);
        TP(L.l2_w, p + "linear2.weight", (int64_t)d * ff);               TP(L.l2_b, p + "linear2.bias", d);
        TP(L.n1_g, p + "norm1.weight", d);  TP(L.n1_b, p + "norm1.bias", d);
        TP(L.n2_g, p + "norm2.weight", d);  TP(L.n2_b, p + "norm2.bias", d);
    }
    return MC_OK;
}

// rows [B*S][d] (pe already added) -> tokens + transformer -> mu_out [B][d]
int run_tail(mc_evalenc* e, const Tail& t, const float* rows_in, const int* lengths, const uint8_t* mask_in, int B, int S,
             float* buf, float* mu_out, hipStream_t s) {
    const mc_evalenc_config& c = e->cfg;
    const int d = c.latent_dim, ff = c.ff_size, S2 = S + 2;
    const long rows = (long)B * S2;
    float* xs = buf;
    float* qkv = xs + rows * d;
    float* att = qkv + rows * 3 * d;
    float* y = att + rows * d;
    float* hid = y + rows * d;
    uint8_t* valid = reinterpret_cast<uint8_t*>(hid + rows * ff);
    hipLaunchKernelGGL(assemble_tokens_k, dim3((unsigned)std::min<long>(cdiv(rows * (d / 4), 256), 4096)), dim3(256), 0, s, rows_in,
                       t.mu, t.lv, t.pe, lengths, mask_in, xs, valid, B, S, d);
    MC_LAUNCH_CHECK();
    int r;
    for (int i = 0; i < c.num_layers; ++i)
        if ((r = mc_enc_layer(t.layers[i], xs, qkv, att, y, hid, rows, B, S2, d, c.num_heads, ff, false, ACT_GELU, 0, valid, 1e-5f, s)))
            return r;
    MC_HIP(hipMemcpy2DAsync(mu_out, (size_t)d * sizeof(float), xs, (size_t)S2 * d * sizeof(float), (size_t)d * sizeof(float), B,
                            hipMemcpyDeviceToDevice, s));
    return MC_OK;
}

size_t tail_floats(const mc_evalenc_config& c, int B, int S) {
    const size_t rows = (size_t)B * (S + 2);
    return rows * (size_t)(c.latent_dim * 6 + c.ff_size) + rows / 4 + 64;
}

}  // namespace

extern "C" {

int mc_evalenc_create(const mc_evalenc_config* cfg, mc_evalenc** out) {
    MC_REQUIRE(cfg && out, "null argument");
    MC_REQUIRE(cfg->nfeats >= 1 && cfg->num_layers >= 1 && cfg->latent_dim == cfg->num_heads * 64 && cfg->ff_size % 4 == 0,
               "evaluation encoder: latent_dim=%d with %d heads unsupported (head_dim must be 64)", cfg->latent_dim, cfg->num_heads);
    MC_REQUIRE(cfg->bert_dim == 0 || (cfg->bert_dim == cfg->bert_heads * 64 && cfg->bert_ff % 4 == 0 && cfg->bert_layers >= 1),
               "evaluation encoder: DistilBERT width %d with %d heads unsupported (head_dim must be 64)", cfg->bert_dim, cfg->bert_heads);
    MC_REQUIRE(cfg->pe_len >= 3, "evaluation encoder: pe_len=%d", cfg->pe_len);
    mc_evalenc* e = new mc_evalenc();
    e->cfg = *cfg;
    e->Cp = (cfg->nfeats + 3) / 4 * 4;
    *out = e;
    return MC_OK;
}

void mc_evalenc_destroy(mc_evalenc* e) {
    if (!e) return;
    for (auto& kv : e->params) (void)hipFree(kv.second.first);
    for (float* p : e->owned) (void)hipFree(p);
    if (e->ws) (void)hipFree(e->ws);
    delete e;
}

int mc_evalenc_set_param(mc_evalenc* e, const char* name, const float* host, int64_t numel) {
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

int mc_evalenc_finalize(mc_evalenc* e) {
    MC_REQUIRE(e, "null encoder");
    const mc_evalenc_config& c = e->cfg;
    const int d = c.latent_dim;
    int r;
    for (float* p : e->owned) (void)hipFree(p);
    e->owned.clear();
    const float* w = nullptr;
    TP(w, "motionencoder.skel_embedding.weight", (int64_t)d * c.nfeats);
    TP(e->skel_b, "motionencoder.skel_embedding.bias", d);
    if (e->Cp != c.nfeats) {                                   // GEMM rows are read 16 bytes at a time
        float* wp = nullptr;
        MC_HIP(hipMalloc((void**)&wp, (size_t)d * e->Cp * sizeof(float)));
        e->owned.push_back(wp);
        MC_HIP(hipMemset(wp, 0, (size_t)d * e->Cp * sizeof(float)));
        MC_HIP(hipMemcpy2D(wp, (size_t)e->Cp * sizeof(float), w, (size_t)c.nfeats * sizeof(float), (size_t)c.nfeats * sizeof(float), d,
                           hipMemcpyDeviceToDevice));
        w = wp;
    }
    e->skel_w = w;
    if ((r = bind_tail(e, "motionencoder.", e->motion))) return r;
    e->has_text = false;
    if (c.bert_dim > 0 && e->params.count("textencoder.projection.1.weight")) {
        const int bw = c.bert_dim, bf = c.bert_ff;
        const std::string t = "textencoder.text_model.";
        TP(e->tok, t + "embeddings.word_embeddings.weight", (int64_t)c.bert_vocab * bw);
        TP(e->pos, t + "embeddings.position_embeddings.weight", (int64_t)c.bert_max_pos * bw);
        TP(e->eln_g, t + "embeddings.LayerNorm.weight", bw);
        TP(e->eln_b, t + "embeddings.LayerNorm.bias", bw);
        e->bert.assign(c.bert_layers, EncLayer());
        for (int i = 0; i < c.bert_layers; ++i) {
            const std::string p = t + "transformer.layer." + std::to_string(i) + ".";
            EncLayer& L = e->bert[i];
            // q_lin | k_lin | v_lin stacked into one [3w][w] projection (the order mc_enc_layer's attention expects)
            float *pw = nullptr, *pb = nullptr;
            MC_HIP(hipMalloc((void**)&pw, (size_t)3 * bw * bw * sizeof(float)));  e->owned.push_back(pw);
            MC_HIP(hipMalloc((void**)&pb, (size_t)3 * bw * sizeof(float)));       e->owned.push_back(pb);
            const char* names[3] = {"q_lin", "k_lin", "v_lin"};
            for (int j = 0; j < 3; ++j) {
                const float *sw = nullptr, *sb = nullptr;
                TP(sw, p + "attention." + names[j] + ".weight", (int64_t)bw * bw);
                TP(sb, p + "attention." + names[j] + ".bias", bw);
                MC_HIP(hipMemcpy(pw + (size_t)j * bw * bw, sw, (size_t)bw * bw * sizeof(float), hipMemcpyDeviceToDevice));
                MC_HIP(hipMemcpy(pb + (size_t)j * bw, sb, (size_t)bw * sizeof(float), hipMemcpyDeviceToDevice));
            }
            L.in_w = pw; L.in_b = pb;
            TP(L.out_w, p + "attention.out_lin.weight", (int64_t)bw * bw);  TP(L.out_b, p + "attention.out_lin.bias", bw);
            TP(L.n1_g, p + "sa_layer_norm.weight", bw);                     TP(L.n1_b, p + "sa_layer_norm.bias", bw);
            TP(L.l1_w, p + "ffn.lin1.weight", (int64_t)bf * bw);            TP(L.l1_b, p + "ffn.lin1.bias", bf);
            TP(L.l2_w, p + "ffn.lin2.weight", (int64_t)bw * bf);            TP(L.l2_b, p + "ffn.lin2.bias", bw);
            TP(L.n2_g, p + "output_layer_norm.weight", bw);                 TP(L.n2_b, p + "output_layer_norm.bias", bw);
        }
        TP(e->proj_w, "textencoder.projection.1.weight", (int64_t)d * bw);
        TP(e->proj_b, "textencoder.projection.1.bias", d);
        if ((r = bind_tail(e, "textencoder.", e->text))) return r;
        e->has_text = true;
    }
    e->finalized = true;
    return MC_OK;
}
#undef TP

int mc_evalenc_encode_motion(mc_evalenc* e, const float* motion, const int32_t* lengths, int32_t B, int32_t T, float* mu_out,
                             void* stream) {
    MC_REQUIRE(e && motion && lengths && mu_out && B >= 1 && T >= 1, "bad argument");
    MC_REQUIRE(e->finalized, "evaluation encoder not finalized");
    const mc_evalenc_config& c = e->cfg;
    MC_REQUIRE(T + 2 <= c.pe_len, "evaluation encoder: %d frames exceed the positional table (%d)", T, c.pe_len);
    hipStream_t s = (hipStream_t)stream;
    const int d = c.latent_dim, Cp = e->Cp;
    const long rows = (long)B * T;
    int r;
    const size_t head = (size_t)rows * (Cp + d) + 64;
    if ((r = ensure_ws(e, head + tail_floats(c, B, T), s))) return r;
    float* pad = e->ws;
    float* emb = pad + rows * Cp;
    const float* A = motion;
    if (Cp != c.nfeats) {
        if ((r = mc_launch_pad_rows(motion, pad, rows, c.nfeats, Cp, s))) return r;
        A = pad;
    }
    GemmArgs g;                                       // emb[(b,t)] = skel_embedding(motion[b][t]) + pe[2 + t]
    g.A = A; g.lda = Cp; g.W = e->skel_w; g.ldw = Cp; g.bias = e->skel_b; g.C = emb; g.ldc = d;
    g.M = (int)rows; g.N = d; g.K = Cp; g.act = ACT_NONE;
    g.add = e->motion.pe + 2 * d; g.add_mod = T; g.ld_add = d;
    if ((r = mc_launch_gemm(GM_ENC, g, 1, 0, s))) return r;
    return run_tail(e, e->motion, emb, lengths, nullptr, B, T, e->ws + head, mu_out, s);
}

int mc_evalenc_encode_text(mc_evalenc* e, const int32_t* ids, const uint8_t* mask, int32_t B, int32_t S, float* mu_out, void* stream) {
    MC_REQUIRE(e && ids && mask && mu_out && B >= 1 && S >= 1, "bad argument");
    MC_REQUIRE(e->finalized, "evaluation encoder not finalized");
    MC_REQUIRE(e->has_text, "evaluation encoder: no textencoder.* weights were loaded");
    const mc_evalenc_config& c = e->cfg;
    MC_REQUIRE(S <= c.bert_max_pos && S + 2 <= c.pe_len, "evaluation encoder: %d tokens exceed the position tables", S);
    hipStream_t s = (hipStream_t)stream;
    const int d = c.latent_dim, w = c.bert_dim, bf = c.bert_ff;
    const long rows = (long)B * S;
    int r;
    const size_t head = (size_t)rows * (w * 6 + bf + d) + 64;
    if ((r = ensure_ws(e, head + tail_floats(c, B, S), s))) return r;
    float* x = e->ws;
    float* qkv = x + rows * w;
    float* att = qkv + rows * 3 * w;
    float* y = att + rows * w;
    float* hid = y + rows * w;
    float* emb = hid + rows * bf;
    if ((r = mc_enc_embed_tokens(ids, e->tok, e->pos, x, rows, S, w, c.bert_vocab, s))) return r;
    if ((r = mc_enc_ln(x, e->eln_g, e->eln_b, x, rows, w, 1e-12f, 0, s))) return r;
    for (int i = 0; i < c.bert_layers; ++i)
        if ((r = mc_enc_layer(e->bert[i], x, qkv, att, y, hid, rows, B, S, w, c.bert_heads, bf, false, ACT_GELU, 0, mask, 1e-12f, s)))
            return r;
    hipLaunchKernelGGL(relu_k, dim3((unsigned)std::min<long>(cdiv(rows * (w / 4), 256), 4096)), dim3(256), 0, s, x, y, rows * (w / 4));
    MC_LAUNCH_CHECK();
    GemmArgs g;                                       // emb[(b,j)] = projection(relu(h[b][j])) + pe[2 + j]
    g.A = y; g.lda = w; g.W = e->proj_w; g.ldw = w; g.bias = e->proj_b; g.C = emb; g.ldc = d;
    g.M = (int)rows; g.N = d; g.K = w; g.act = ACT_NONE;
    g.add = e->text.pe + 2 * d; g.add_mod = S; g.ld_add = d;
    if ((r = mc_launch_gemm(GM_ENC, g, 1, 0, s))) return r;
    return run_tail(e, e->text, emb, nullptr, mask, B, S, e->ws + head, mu_out, s);
}

}  // extern "C"
