// WavEncoder (reference mogen/models/utils/blocks.py:11-71): the step-invariant audio condition encoder of the
// speech-to-gesture configs (ConditionEncoder, controlnet.py:90-105) -- 6 residual BasicBlocks of
// Conv1d(k=15) + BatchNorm1d + LeakyReLU, total stride 5*6*6*3 = 540 (16 kHz audio -> ~30 fps features).
// ~250 GFLOP per 196-frame sample, i.e. ~13 % of a 50-step DDIM run -> worth MFMA.
//
// MI355X design: activations are channels-last [B][T][C] with the conv's zero padding materialised in the
// buffer, so im2col is a VIEW: row (b,t) of the implicit GEMM is the 15*Cin contiguous floats starting at
// x[b][t*stride][0] (row stride = stride*Cin, overlapping rows).  Each convolution is therefore one launch of the
// fp32 MFMA GEMM (mc_gemm.hip) with batch as the group dimension; eval-mode BatchNorm is folded into the
// weights/bias at pack time (motioncraft_amd/wav_encoder.py), LeakyReLU and the residual add (+ LeakyReLU after
// it) run in the GEMM epilogue, and each block writes straight into the interior of the next block's padded buffer.
#include "mc_common.h"
#include "mc_gemm.h"
#include "../../include/motioncraft_amd.h"
#include <map>
#include <string>
#include <vector>

namespace {
constexpr int KS = 15;
struct ConvW { const float* w = nullptr; const float* b = nullptr; long ldw = 0; };
struct BlockSpec { int cin, planes, stride, pad; bool down; };
}  // namespace

struct mc_wavenc {
    int audio_in = 0, out_dim = 0;
    BlockSpec spec[6];
    ConvW c1[6], c2[6], dn[6];
    std::map<std::string, std::pair<float*, int64_t>> params;
    bool finalized = false;
    float* ws = nullptr;
    size_t ws_floats = 0;
};

namespace {

int conv_len(int Tin, int pad, int stride) { return (Tin + 2 * pad - KS) / stride + 1; }

// one Conv1d(k=15, stride, padding materialised in X) as a grouped GEMM; X row pitch = Tp*Cin per batch item
int conv_gemm(const float* X, long Tp, int Cin, int stride, const ConvW& w, int Cout, int B, int Tout,
              float* C, long c_rows_per_item, const float* R, long ldr, long r_item_stride, int act, int act_after_res,
              hipStream_t s) {
    GemmArgs g;
    g.A = X; g.lda = (long)stride * Cin; g.a_gstride = Tp * Cin;
    g.W = w.w; g.ldw = w.ldw; g.bias = w.b;
    g.C = C; g.ldc = Cout; g.c_gstride = c_rows_per_item * Cout;
    g.R = R; g.ldr = ldr; g.r_gstride = r_item_stride;
    g.act = act; g.act_after_res = act_after_res;
    g.M = Tout; g.N = Cout; g.K = KS * Cin;
    const bool aligned = (g.K % 4 == 0) && (g.lda % 4 == 0) && (g.a_gstride % 4 == 0);
    return mc_launch_gemm(aligned ? GM_PLAIN : GM_ENC, g, B, 0, s);
}

int get(mc_wavenc* e, const std::string& name, int64_t numel, const float** out) {
    auto it = e->params.find(name);
    if (it == e->params.end()) { mc_set_error("wav encoder: missing parameter '%s'", name.c_str()); return MC_ERR_STATE; }
    if (it->second.second != numel) {
        mc_set_error("wav encoder: parameter '%s' has %ld elements, expected %ld", name.c_str(), (long)it->second.second, (long)numel);
        return MC_ERR_STATE;
    }
    *out = it->second.first;
    return MC_OK;
}

}  // namespace

extern "C" {

int mc_wavenc_create(int32_t audio_in, int32_t out_dim, mc_wavenc** out) {
    MC_REQUIRE(out && audio_in >= 1 && out_dim >= 16 && out_dim % 16 == 0, "wav encoder: audio_in=%d out_dim=%d unsupported", audio_in, out_dim);
    mc_wavenc* e = new mc_wavenc();
    e->audio_in = audio_in;
    e->out_dim = out_dim;
    const int D = out_dim;
    // blocks.py:57-64: (inplanes, planes, stride, first_dilation used as padding, downsample)
    const BlockSpec sp[6] = {{audio_in, D / 4, 5, 1600, true}, {D / 4, D / 4, 6, 0, true}, {D / 4, D / 4, 1, 7, false},
                             {D / 4, D / 2, 6, 0, true},       {D / 2, D / 2, 1, 7, false}, {D / 2, D, 3, 0, true}};
    for (int i = 0; i < 6; ++i) e->spec[i] = sp[i];
    *out = e;
    return MC_OK;
}

void mc_wavenc_destroy(mc_wavenc* e) {
    if (!e) return;
    for (auto& kv : e->params) (void)hipFree(kv.second.first);
    if (e->ws) (void)hipFree(e->ws);
    delete e;
}

int mc_wavenc_set_param(mc_wavenc* e, const char* name, const float* host, int64_t numel) {
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

// parameters (BatchNorm folded, kernel-tap-major): b{i}.conv1.w [planes][ld(15*cin)], b{i}.conv1.b [planes],
// b{i}.conv2.w [planes][15*planes], b{i}.conv2.b, and for downsample blocks b{i}.down.w / b{i}.down.b;
// rows padded to a multiple of 4 floats (ld = ceil4(15*cin)).
int mc_wavenc_finalize(mc_wavenc* e) {
    MC_REQUIRE(e, "null encoder");
    for (int i = 0; i < 6; ++i) {
        const BlockSpec& b = e->spec[i];
        const std::string pre = "b" + std::to_string(i) + ".";
        const long ld1 = (KS * b.cin + 3) / 4 * 4, ld2 = (long)KS * b.planes;
        int r;
        if ((r = get(e, pre + "conv1.w", (int64_t)b.planes * ld1, &e->c1[i].w))) return r;
        if ((r = get(e, pre + "conv1.b", b.planes, &e->c1[i].b))) return r;
        e->c1[i].ldw = ld1;
        if ((r = get(e, pre + "conv2.w", (int64_t)b.planes * ld2, &e->c2[i].w))) return r;
        if ((r = get(e, pre + "conv2.b", b.planes, &e->c2[i].b))) return r;
        e->c2[i].ldw = ld2;
        if (b.down) {
            if ((r = get(e, pre + "down.w", (int64_t)b.planes * ld1, &e->dn[i].w))) return r;
            if ((r = get(e, pre + "down.b", b.planes, &e->dn[i].b))) return r;
            e->dn[i].ldw = ld1;
        }
    }
    e->finalized = true;
    return MC_OK;
}

int mc_wavenc_out_len(const mc_wavenc* e, int32_t samples, int32_t* frames) {
    MC_REQUIRE(e && frames, "null argument");
    int T = samples;
    for (int i = 0; i < 6; ++i) {
        T = conv_len(T, e->spec[i].pad, e->spec[i].stride);
        if (T < 1) { *frames = 0; return MC_OK; }
    }
    *frames = T;
    return MC_OK;
}

int mc_wavenc_forward(mc_wavenc* e, const float* wav, int32_t B, int32_t samples, float* out, void* stream) {
    MC_REQUIRE(e && wav && out && B >= 1 && samples >= 1, "bad argument");
    MC_REQUIRE(e->finalized, "wav encoder not finalized");
    hipStream_t s = (hipStream_t)stream;
    // lengths and workspace layout: per block  X (padded input), H (conv1 output padded by 7), S (shortcut)
    int Tin[7];
    Tin[0] = samples;
    size_t need = 0;
    std::vector<size_t> offX(7), offH(6), offS(6);
    for (int i = 0; i < 6; ++i) {
        const BlockSpec& b = e->spec[i];
        Tin[i + 1] = conv_len(Tin[i], b.pad, b.stride);
        MC_REQUIRE(Tin[i + 1] >= 1, "wav encoder: %d samples are too few", samples);
        offX[i] = need; need += (size_t)B * (Tin[i] + 2 * b.pad) * b.cin;
        need = (need + 63) & ~(size_t)63;
        offH[i] = need; need += (size_t)B * (Tin[i + 1] + 2 * (KS / 2)) * b.planes;
        need = (need + 63) & ~(size_t)63;
        offS[i] = need; if (b.down) need += (size_t)B * Tin[i + 1] * b.planes;
        need = (need + 63) & ~(size_t)63;
    }
    if (need > e->ws_floats) {
        if (e->ws) { MC_HIP(hipStreamSynchronize(s)); MC_HIP(hipFree(e->ws)); e->ws = nullptr; e->ws_floats = 0; }
        MC_HIP(hipMalloc((void**)&e->ws, need * sizeof(float)));
        e->ws_floats = need;
    }
    // zero everything once (padding borders), then drop the audio into the interior of X0
    MC_HIP(hipMemsetAsync(e->ws, 0, need * sizeof(float), s));
    {
        const BlockSpec& b = e->spec[0];
        const size_t row = (size_t)samples * b.cin * sizeof(float);
        MC_HIP(hipMemcpy2DAsync(e->ws + offX[0] + (size_t)b.pad * b.cin, (size_t)(samples + 2 * b.pad) * b.cin * sizeof(float),
                                wav, row, row, B, hipMemcpyDeviceToDevice, s));
    }
    int r;
    for (int i = 0; i < 6; ++i) {
        const BlockSpec& b = e->spec[i];
        const long Tp = Tin[i] + 2 * b.pad, T1 = Tin[i + 1], Hp = T1 + 2 * (KS / 2);
        const float* X = e->ws + offX[i];
        float* H = e->ws + offH[i];
        // conv1 + bn1 + LeakyReLU -> interior of H
        if ((r = conv_gemm(X, Tp, b.cin, b.stride, e->c1[i], b.planes, B, (int)T1, H + (KS / 2) * b.planes, Hp, nullptr, 0, -1,
                           ACT_LRELU, 0, s))) return r;
        // shortcut: downsample conv + bn, or the block input itself (stride 1, same width)
        const float* R;
        long ldr, rstride;
        if (b.down) {
            float* S = e->ws + offS[i];
            if ((r = conv_gemm(X, Tp, b.cin, b.stride, e->dn[i], b.planes, B, (int)T1, S, T1, nullptr, 0, -1, ACT_NONE, 0, s))) return r;
            R = S; ldr = b.planes; rstride = T1 * b.planes;
        } else {
            R = X + (long)b.pad * b.cin; ldr = b.cin; rstride = Tp * b.cin;
        }
        // conv2 + bn2, + shortcut, LeakyReLU -> interior of the next block's input (or the result)
        float* C;
        long crows;
        if (i + 1 < 6) {
            const int pn = e->spec[i + 1].pad;
            C = e->ws + offX[i + 1] + (long)pn * b.planes;
            crows = T1 + 2 * pn;
        } else {
            C = out;
            crows = T1;
        }
        if ((r = conv_gemm(H, Hp, b.planes, 1, e->c2[i], b.planes, B, (int)T1, C, crows, R, ldr, rstride, ACT_LRELU, 1, s))) return r;
    }
    return MC_OK;
}

}  // extern "C"
