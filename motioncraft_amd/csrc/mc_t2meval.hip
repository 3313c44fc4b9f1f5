// HumanML3D / KIT evaluation embedding model on the device: T2MContrastiveModel (mogen/models/rnns/t2m_bigru.py)
//   encode_motion   T2MMotionEncoder (:72-110): MovementConvEncoder (two Conv1d k=4 s=2 p=1 + LeakyReLU(0.2), Linear;
//                   :226-246) on motion[..., :-4], then MotionEncoderBiGRUCo (:249-282) with lengths // 4
//   encode_text     TextEncoderBiGRUCo (:186-223) on word vectors + pos_emb(one-hot)
// The convolutions are GEMMs over a zero-padded channels-last copy (row t of the im2col matrix is the contiguous
// 4 x C window starting at padded frame 2t: lda = 2C < K = 4C, one GEMM group per sample).  The GRU input products of
// all steps and both directions are one grouped GEMM; the recurrence is one grouped [B, H] x [H, 3H] GEMM + one gate
// kernel per step (both directions together).  A packed sequence = sample b is updated at steps s < len[b]; the
// reverse direction visits t = len[b]-1-s.
#include "mc_common.h"
#include "mc_gemm.h"
#include "mc_kernels.h"
#include "mc_enc.h"
#include "../../include/motioncraft_amd.h"
#include <map>
#include <string>
#include <vector>

namespace {

// Y[b][0] = Y[b][T+1] = 0; Y[b][1+t][0:C] = X[b][t][0:C] (source rows of ldx floats), zero up to Cp
__global__ __launch_bounds__(256) void pad_time_k(const float* __restrict__ X, float* __restrict__ Y, int B, int T, int C, int ldx, int Cp) {
    const long n = (long)B * (T + 2) * Cp;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cp);
        const long r = i / Cp;
        const int t = (int)(r % (T + 2)) - 1, b = (int)(r / (T + 2));
        Y[i] = (t >= 0 && t < T && c < C) ? X[((long)b * T + t) * ldx + c] : 0.f;
    }
}

__global__ __launch_bounds__(256) void lrelu_k(float* __restrict__ X, long n, float slope) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = X[i];
        X[i] = v >= 0.f ? v : slope * v;
    }
}

// h[b][d][:] = hidden[d][:]
__global__ __launch_bounds__(256) void gru_init_k(const float* __restrict__ hidden, float* __restrict__ h, int B, int H) {
    const long n = (long)B * 2 * H;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) h[i] = hidden[i % (2 * H)];
}

// one recurrence step of both directions: gi [2][B*S][3H] (input products + b_ih), gh [B][2][3H] (W_hh h + b_hh), h [B][2][H]
__global__ __launch_bounds__(256) void gru_gate_k(const float* __restrict__ gi, const float* __restrict__ gh, float* __restrict__ h,
                                                  const int* __restrict__ lens, int len_div, int s, int B, int S, int H) {
    const long n = (long)B * 2 * H;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i % H), d = (int)((i / H) % 2), b = (int)(i / (2L * H));
        int len = lens[b] / len_div;
        len = len > S ? S : len;
        if (s >= len) continue;
        const int t = d == 0 ? s : len - 1 - s;
        const float* a = gi + ((long)d * B * S + (long)b * S + t) * 3 * H;
        const float* g = gh + ((long)b * 2 + d) * 3 * H;
        const float r = 1.f / (1.f + expf(-(a[j] + g[j])));
        const float z = 1.f / (1.f + expf(-(a[H + j] + g[H + j])));
        const float nn = tanhf(a[2 * H + j] + r * g[2 * H + j]);
        h[i] = (1.f - z) * nn + z * h[i];
    }
}

struct Head {                        // input_emb + BiGRU + output_net
    int din = 0, hid = 0, dout = 0;
    const float *in_w = nullptr, *in_b = nullptr, *hidden = nullptr, *o0_w = nullptr, *o0_b = nullptr, *ln_g = nullptr, *ln_b = nullptr,
                *o3_w = nullptr, *o3_b = nullptr;
    float *wih = nullptr, *whh = nullptr, *bih = nullptr, *bhh = nullptr;      // [2][3H][H] / [2][3H], packed at finalize
};

int grid_for(long n) { return (int)std::min<long>(cdiv(n, 256), 4096); }

}  // namespace

struct mc_t2meval {
    mc_t2meval_config cfg;
    std::map<std::string, std::pair<float*, int64_t>> params;
    std::vector<float*> owned;
    const float *c1_w = nullptr, *c1_b = nullptr, *c2_w = nullptr, *c2_b = nullptr, *mo_w = nullptr, *mo_b = nullptr;
    const float *pos_w = nullptr, *pos_b = nullptr;
    Head motion, text;
    int Cp = 0, Pp = 0;
    bool finalized = false, has_text = false;
    float* ws = nullptr;
    size_t ws_floats = 0;
};

namespace {

int getp(mc_t2meval* e, const std::string& name, int64_t numel, const float** out) {
    auto it = e->params.find(name);
    if (it == e->params.end()) { mc_set_error("t2m evaluator: missing parameter '%s'", name.c_str()); return MC_ERR_STATE; }
    if (it->second.second != numel) {
        mc_set_error("t2m evaluator: parameter '%s' has %ld elements, expected %ld", name.c_str(), (long)it->second.second, (long)numel);
        return MC_ERR_STATE;
    }
    *out = it->second.first;
    return MC_OK;
}

int ensure_ws(mc_t2meval* e, size_t floats, hipStream_t s) {
    if (floats <= e->ws_floats) return MC_OK;
    if (e->ws) { MC_HIP(hipStreamSynchronize(s)); MC_HIP(hipFree(e->ws)); e->ws = nullptr; e->ws_floats = 0; }
    MC_HIP(hipMalloc((void**)&e->ws, floats * sizeof(float)));
    e->ws_floats = floats;
    return MC_OK;
}

#define TP(ptr, name, n) if ((r = getp(e, (name), (int64_t)(n), &(ptr)))) return r

int bind_head(mc_t2meval* e, const std::string& pre, int din, int hid, int dout, Head& h) {
    int r;
    h.din = din; h.hid = hid; h.dout = dout;
    TP(h.in_w, pre + "input_emb.weight", (int64_t)hid * din);  TP(h.in_b, pre + "input_emb.bias", hid);
    TP(h.hidden, pre + "hidden", 2 * hid);
    TP(h.o0_w, pre + "output_net.0.weight", (int64_t)hid * 2 * hid);  TP(h.o0_b, pre + "output_net.0.bias", hid);
    TP(h.ln_g, pre + "output_net.1.weight", hid);  TP(h.ln_b, pre + "output_net.1.bias", hid);
    TP(h.o3_w, pre + "output_net.3.weight", (int64_t)dout * hid);  TP(h.o3_b, pre + "output_net.3.bias", dout);
    const size_t wn = (size_t)3 * hid * hid, bn = (size_t)3 * hid;
    float** dst[4] = {&h.wih, &h.whh, &h.bih, &h.bhh};
    const char* base[4] = {"gru.weight_ih_l0", "gru.weight_hh_l0", "gru.bias_ih_l0", "gru.bias_hh_l0"};
    for (int k = 0; k < 4; ++k) {
        const size_t n = k < 2 ? wn : bn;
        MC_HIP(hipMalloc((void**)dst[k], 2 * n * sizeof(float)));
        e->owned.push_back(*dst[k]);
        for (int d = 0; d < 2; ++d) {
            const float* src = nullptr;
            TP(src, pre + base[k] + (d ? "_reverse" : ""), (int64_t)n);
            MC_HIP(hipMemcpy(*dst[k] + d * n, src, n * sizeof(float), hipMemcpyDeviceToDevice));
        }
    }
    return MC_OK;
}

// x [B*S][din] -> out [B][dout];  buf: emb [B*S][H] | gi [2][B*S][3H] | h [B][2H] | gh [B][6H] | y [B][H]
int run_head(const Head& p, const float* x, const int* lens, int len_div, int B, int S, float* buf, float* out, hipStream_t s) {
    const int H = p.hid;
    const long BS = (long)B * S;
    float* emb = buf;
    float* gi = emb + BS * H;
    float* h = gi + 2 * BS * 3 * H;
    float* gh = h + (long)B * 2 * H;
    float* y = gh + (long)B * 6 * H;
    int r;
    if ((r = mc_enc_dense(x, p.din, p.in_w, p.din, p.in_b, nullptr, 0, emb, H, BS, H, p.din, ACT_NONE, s))) return r;
    GemmArgs g;                                        // gi[d] = emb W_ih[d]^T + b_ih[d]
    g.A = emb; g.lda = H; g.W = p.wih; g.ldw = H; g.w_gstride = (long)3 * H * H; g.bias = p.bih; g.b_gstride = 3 * H;
    g.C = gi; g.ldc = 3 * H; g.c_gstride = BS * 3 * H; g.M = (int)BS; g.N = 3 * H; g.K = H;
    if ((r = mc_launch_gemm(GM_PLAIN, g, 2, 0, s))) return r;
    hipLaunchKernelGGL(gru_init_k, dim3(grid_for((long)B * 2 * H)), dim3(256), 0, s, p.hidden, h, B, H);
    MC_LAUNCH_CHECK();
    GemmArgs q;                                        // gh[b][d] = h[b][d] W_hh[d]^T + b_hh[d]
    q.A = h; q.lda = 2 * H; q.a_gstride = H; q.W = p.whh; q.ldw = H; q.w_gstride = (long)3 * H * H; q.bias = p.bhh; q.b_gstride = 3 * H;
    q.C = gh; q.ldc = 6 * H; q.c_gstride = 3 * H; q.M = B; q.N = 3 * H; q.K = H;
    for (int st = 0; st < S; ++st) {
        if ((r = mc_launch_gemm(GM_PLAIN, q, 2, 0, s))) return r;
        hipLaunchKernelGGL(gru_gate_k, dim3(grid_for((long)B * 2 * H)), dim3(256), 0, s, gi, gh, h, lens, len_div, st, B, S, H);
        MC_LAUNCH_CHECK();
    }
    if ((r = mc_enc_dense(h, 2 * H, p.o0_w, 2 * H, p.o0_b, nullptr, 0, y, H, B, H, 2 * H, ACT_NONE, s))) return r;
    if ((r = mc_enc_ln(y, p.ln_g, p.ln_b, y, B, H, 1e-5f, 0, s))) return r;
    hipLaunchKernelGGL(lrelu_k, dim3(grid_for((long)B * H)), dim3(256), 0, s, y, (long)B * H, 0.2f);
    MC_LAUNCH_CHECK();
    return mc_enc_dense(y, H, p.o3_w, H, p.o3_b, nullptr, 0, out, p.dout, B, p.dout, H, ACT_NONE, s);
}

size_t head_floats(const Head& p, int B, int S) {
    return (size_t)B * S * p.hid * 7 + (size_t)B * p.hid * 9 + 64;
}

// host-side repack of an uploaded parameter: conv weight [O][C][4] -> tap-major [O][4][Cp]; pos_emb [O][P] -> [O][Pp]
std::vector<float> repack_conv(const float* w, int O, int C, int Cp) {
    std::vector<float> out((size_t)O * 4 * Cp, 0.f);
    for (int o = 0; o < O; ++o)
        for (int c = 0; c < C; ++c)
            for (int k = 0; k < 4; ++k) out[((size_t)o * 4 + k) * Cp + c] = w[((size_t)o * C + c) * 4 + k];
    return out;
}

}  // namespace

extern "C" {

int mc_t2meval_create(const mc_t2meval_config* cfg, mc_t2meval** out) {
    MC_REQUIRE(cfg && out, "null argument");
    MC_REQUIRE(cfg->input_size > 4 && cfg->movement_hidden % 4 == 0 && cfg->movement_latent % 4 == 0 && cfg->motion_hidden % 4 == 0 &&
                   cfg->motion_latent % 4 == 0,
               "t2m evaluator: motion widths must be multiples of 4");
    MC_REQUIRE(cfg->word_size == 0 || (cfg->word_size % 4 == 0 && cfg->text_hidden % 4 == 0 && cfg->text_out % 4 == 0 && cfg->pos_size >= 1),
               "t2m evaluator: text widths must be multiples of 4");
    mc_t2meval* e = new mc_t2meval();
    e->cfg = *cfg;
    e->Cp = (cfg->input_size - 4 + 3) / 4 * 4;
    e->Pp = (cfg->pos_size + 3) / 4 * 4;
    *out = e;
    return MC_OK;
}

void mc_t2meval_destroy(mc_t2meval* e) {
    if (!e) return;
    for (auto& kv : e->params) (void)hipFree(kv.second.first);
    for (float* p : e->owned) (void)hipFree(p);
    if (e->ws) (void)hipFree(e->ws);
    delete e;
}

int mc_t2meval_set_param(mc_t2meval* e, const char* name, const float* host, int64_t numel) {
    MC_REQUIRE(e && name && host && numel > 0, "bad argument");
    const mc_t2meval_config& c = e->cfg;
    const std::string n = name;
    std::vector<float> packed;
    const int C = c.input_size - 4;
    if (n == "movement_encoder.main.0.weight") {
        MC_REQUIRE(numel == (int64_t)c.movement_hidden * C * 4, "t2m evaluator: %s has %ld elements", name, (long)numel);
        packed = repack_conv(host, c.movement_hidden, C, e->Cp);
    } else if (n == "movement_encoder.main.3.weight") {
        MC_REQUIRE(numel == (int64_t)c.movement_latent * c.movement_hidden * 4, "t2m evaluator: %s has %ld elements", name, (long)numel);
        packed = repack_conv(host, c.movement_latent, c.movement_hidden, c.movement_hidden);
    } else if (n == "text_encoder.pos_emb.weight") {
        MC_REQUIRE(numel == (int64_t)c.word_size * c.pos_size, "t2m evaluator: %s has %ld elements", name, (long)numel);
        packed.assign((size_t)c.word_size * e->Pp, 0.f);
        for (int o = 0; o < c.word_size; ++o)
            for (int k = 0; k < c.pos_size; ++k) packed[(size_t)o * e->Pp + k] = host[(size_t)o * c.pos_size + k];
    }
    if (!packed.empty()) { host = packed.data(); numel = (int64_t)packed.size(); }
    float* d = nullptr;
    MC_HIP(hipMalloc((void**)&d, (size_t)numel * sizeof(float)));
    MC_HIP(hipMemcpy(d, host, (size_t)numel * sizeof(float), hipMemcpyHostToDevice));
    auto it = e->params.find(n);
    if (it != e->params.end()) (void)hipFree(it->second.first);
    e->params[n] = {d, numel};
    e->finalized = false;
    return MC_OK;
}

int mc_t2meval_finalize(mc_t2meval* e) {
    MC_REQUIRE(e, "null evaluator");
    const mc_t2meval_config& c = e->cfg;
    int r;
    for (float* p : e->owned) (void)hipFree(p);
    e->owned.clear();
    TP(e->c1_w, "movement_encoder.main.0.weight", (int64_t)c.movement_hidden * 4 * e->Cp);
    TP(e->c1_b, "movement_encoder.main.0.bias", c.movement_hidden);
    TP(e->c2_w, "movement_encoder.main.3.weight", (int64_t)c.movement_latent * 4 * c.movement_hidden);
    TP(e->c2_b, "movement_encoder.main.3.bias", c.movement_latent);
    TP(e->mo_w, "movement_encoder.out_net.weight", (int64_t)c.movement_latent * c.movement_latent);
    TP(e->mo_b, "movement_encoder.out_net.bias", c.movement_latent);
    if ((r = bind_head(e, "motion_encoder.", c.movement_latent, c.motion_hidden, c.motion_latent, e->motion))) return r;
    e->has_text = false;
    if (c.word_size > 0 && e->params.count("text_encoder.pos_emb.weight")) {
        TP(e->pos_w, "text_encoder.pos_emb.weight", (int64_t)c.word_size * e->Pp);
        TP(e->pos_b, "text_encoder.pos_emb.bias", c.word_size);
        if ((r = bind_head(e, "text_encoder.", c.word_size, c.text_hidden, c.text_out, e->text))) return r;
        e->has_text = true;
    }
    e->finalized = true;
    return MC_OK;
}
#undef TP

int mc_t2meval_encode_motion(mc_t2meval* e, const float* motion, const int32_t* lengths, int32_t B, int32_t T, float* out, void* stream) {
    MC_REQUIRE(e && motion && lengths && out && B >= 1 && T >= 4, "bad argument");
    MC_REQUIRE(e->finalized, "t2m evaluator not finalized");
    const mc_t2meval_config& c = e->cfg;
    hipStream_t s = (hipStream_t)stream;
    const int C = c.input_size - 4, Cp = e->Cp, Hm = c.movement_hidden, Lm = c.movement_latent;
    const int T1 = (T + 2 - 4) / 2 + 1, T2 = (T1 + 2 - 4) / 2 + 1;
    MC_REQUIRE(T2 >= 1, "t2m evaluator: %d frames are too few", T);
    const size_t n_pad1 = (size_t)B * (T + 2) * Cp + 4 * Cp, n_pad2 = (size_t)B * (T1 + 2) * Hm + 4 * Hm, n_c2 = (size_t)B * T2 * Lm;
    int r;
    if ((r = ensure_ws(e, n_pad1 + n_pad2 + 2 * n_c2 + head_floats(e->motion, B, T2) + 64, s))) return r;
    float* pad1 = e->ws;
    float* pad2 = pad1 + n_pad1;
    float* c2 = pad2 + n_pad2;
    float* mov = c2 + n_c2;
    float* buf = mov + n_c2;
    hipLaunchKernelGGL(pad_time_k, dim3(grid_for((long)B * (T + 2) * Cp)), dim3(256), 0, s, motion, pad1, B, T, C, c.input_size, Cp);
    MC_LAUNCH_CHECK();
    MC_HIP(hipMemsetAsync(pad2, 0, n_pad2 * sizeof(float), s));
    GemmArgs g;                                        // conv 1 -> rows 1 .. T1 of the padded buffer of conv 2
    g.A = pad1; g.lda = 2 * Cp; g.a_gstride = (long)(T + 2) * Cp; g.W = e->c1_w; g.ldw = 4 * Cp; g.bias = e->c1_b;
    g.C = pad2 + Hm; g.ldc = Hm; g.c_gstride = (long)(T1 + 2) * Hm; g.M = T1; g.N = Hm; g.K = 4 * Cp;
    if ((r = mc_launch_gemm(GM_PLAIN, g, B, 0, s))) return r;
    // LeakyReLU(0.2) keeps zeros: applying it to the whole padded buffer leaves the pad rows zero
    hipLaunchKernelGGL(lrelu_k, dim3(grid_for((long)n_pad2)), dim3(256), 0, s, pad2, (long)B * (T1 + 2) * Hm, 0.2f);
    MC_LAUNCH_CHECK();
    GemmArgs q;                                        // conv 2
    q.A = pad2; q.lda = 2 * Hm; q.a_gstride = (long)(T1 + 2) * Hm; q.W = e->c2_w; q.ldw = 4 * Hm; q.bias = e->c2_b;
    q.C = c2; q.ldc = Lm; q.c_gstride = (long)T2 * Lm; q.M = T2; q.N = Lm; q.K = 4 * Hm;
    if ((r = mc_launch_gemm(GM_PLAIN, q, B, 0, s))) return r;
    hipLaunchKernelGGL(lrelu_k, dim3(grid_for((long)n_c2)), dim3(256), 0, s, c2, (long)n_c2, 0.2f);
    MC_LAUNCH_CHECK();
    if ((r = mc_enc_dense(c2, Lm, e->mo_w, Lm, e->mo_b, nullptr, 0, mov, Lm, (long)B * T2, Lm, Lm, ACT_NONE, s))) return r;
    return run_head(e->motion, mov, lengths, 4, B, T2, buf, out, s);
}

int mc_t2meval_encode_text(mc_t2meval* e, const float* word_emb, const float* pos_onehot, const int32_t* sent_len, int32_t B, int32_t S,
                           float* out, void* stream) {
    MC_REQUIRE(e && word_emb && pos_onehot && sent_len && out && B >= 1 && S >= 1, "bad argument");
    MC_REQUIRE(e->finalized, "t2m evaluator not finalized");
    MC_REQUIRE(e->has_text, "t2m evaluator: no text_encoder.* weights were loaded");
    const mc_t2meval_config& c = e->cfg;
    hipStream_t s = (hipStream_t)stream;
    const long BS = (long)B * S;
    int r;
    if ((r = ensure_ws(e, (size_t)BS * c.word_size + head_floats(e->text, B, S) + 64, s))) return r;
    float* x = e->ws;
    GemmArgs g;                                        // x = word_emb + pos_emb(pos_onehot)   (K = pos_size: unaligned rows)
    g.A = pos_onehot; g.lda = c.pos_size; g.W = e->pos_w; g.ldw = e->Pp; g.bias = e->pos_b; g.R = word_emb; g.ldr = c.word_size;
    g.C = x; g.ldc = c.word_size; g.M = (int)BS; g.N = c.word_size; g.K = c.pos_size;
    if ((r = mc_launch_gemm(GM_ENC, g, 1, 0, s))) return r;
    return run_head(e->text, x, sent_len, 1, B, S, x + BS * c.word_size, out, s);
}

}  // extern "C"
