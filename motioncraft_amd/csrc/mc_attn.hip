// MC-Attn cores (reference mogen/models/attentions/st_attention.py:105-179):
//   body_k      static 12x12 body topology + dynamic topology (EfficientSelfAttention over the
//               H body parts of one frame, efficient_attention.py:25-46), joint tile in LDS
//   temporal_k  temporal linear attention over text (+) motion tokens of one (sample, part):
//               column softmax of K over the sequence, A2 = K^T V and Y = softmax_L(Q) A2 on
//               v_mfma_f32_32x32x2_f32, operands staged through LDS
#include "mc_common.h"
#include "mc_kernels.h"

namespace {

// ---------------------------------------------------------------------------------------
// One workgroup per frame (b,t).  bv = motion_feat[..., 0:L] of the frame's H tokens,
// qkv = [query | key | value] of LN(bv) (GEMM done before).  Writes
//   ys[h][c] = sum_l softmax(body_weight)[h][l] bv[l][c]  +  bv[h][c] + (q A)[h][c]
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void body_k(const float* __restrict__ mf, long ldmf, const float* __restrict__ qkv,
                                              const float* __restrict__ wsm, float* __restrict__ ys, int H, int L, int G) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int HL = H * L;
    const int hd = L / G;
    float* s_bv = sm;
    float* s_q = s_bv + HL;
    float* s_k = s_q + HL;
    float* s_v = s_k + HL;
    float* s_A = s_v + HL;          // [G][hd][hd]
    float* s_w = s_A + L * hd;      // [H][H]
    const long frame = blockIdx.x;
    const long tok0 = frame * H;
    const int tid = threadIdx.x;
    for (int i = tid; i < HL / 4; i += 256) {
        const int h = (i * 4) / L, c = (i * 4) % L;
        *reinterpret_cast<f32x4*>(s_bv + h * L + c) = *reinterpret_cast<const f32x4*>(mf + (tok0 + h) * ldmf + c);
        const float* qr = qkv + (tok0 + h) * 3 * L + c;
        *reinterpret_cast<f32x4*>(s_q + h * L + c) = *reinterpret_cast<const f32x4*>(qr);
        *reinterpret_cast<f32x4*>(s_k + h * L + c) = *reinterpret_cast<const f32x4*>(qr + L);
        *reinterpret_cast<f32x4*>(s_v + h * L + c) = *reinterpret_cast<const f32x4*>(qr + 2 * L);
    }
    for (int i = tid; i < H * H; i += 256) s_w[i] = wsm[i];
    __syncthreads();
    // query: softmax over the hd channels of a head   (efficient_attention.py:35)
    for (int i = tid; i < H * G; i += 256) {
        float* p = s_q + (i / G) * L + (i % G) * hd;
        float m = p[0];
        for (int d = 1; d < hd; ++d) m = fmaxf(m, p[d]);
        float s = 0.f;
        for (int d = 0; d < hd; ++d) { p[d] = expf(p[d] - m); s += p[d]; }
        for (int d = 0; d < hd; ++d) p[d] /= s;
    }
    // key: softmax over the H body parts (dim=1)      (efficient_attention.py:36), mask == 1
    for (int c = tid; c < L; c += 256) {
        float m = s_k[c];
        for (int h = 1; h < H; ++h) m = fmaxf(m, s_k[h * L + c]);
        float s = 0.f;
        for (int h = 0; h < H; ++h) { const float e = expf(s_k[h * L + c] - m); s_k[h * L + c] = e; s += e; }
        for (int h = 0; h < H; ++h) s_k[h * L + c] /= s;
    }
    __syncthreads();
    // A[g][d][l] = sum_h k[h][g,d] v[h][g,l]
    for (int o = tid; o < L * hd; o += 256) {
        const int g = o / (hd * hd), d = (o / hd) % hd, l = o % hd;
        float a = 0.f;
        for (int h = 0; h < H; ++h) a += s_k[h * L + g * hd + d] * s_v[h * L + g * hd + l];
        s_A[o] = a;
    }
    __syncthreads();
    float* out = ys + frame * HL;
    for (int o = tid; o < HL; o += 256) {
        const int h = o / L, c = o % L, g = c / hd, l = c % hd;
        float st = 0.f;
        for (int j = 0; j < H; ++j) st += s_w[h * H + j] * s_bv[j * L + c];
        float dy = 0.f;
        for (int d = 0; d < hd; ++d) dy += s_q[h * L + g * hd + d] * s_A[(g * hd + d) * hd + l];
        out[o] = st + (s_bv[o] + dy);
    }
}

// ---------------------------------------------------------------------------------------
// Temporal linear attention, one workgroup per (sample b of the CFG-doubled batch, part h).
// ---------------------------------------------------------------------------------------
template <int L>
__global__ __launch_bounds__(256) void temporal_k(const float* __restrict__ mf, const float* __restrict__ tf,
                                                  const float* __restrict__ mask, float* __restrict__ yt,
                                                  int B, int T, int Nt, int H) {
    constexpr int MT = (L == 128) ? 2 : 1;
    constexpr int LP = L + 4;
    constexpr int LQ = L + 1;
    constexpr int C4 = L / 4;            // float4 columns per row
    constexpr int NSL = 256 / C4;        // row slices in the stats pass
    __shared__ __attribute__((aligned(16))) float sm[2 * L + 2 * NSL * L + 2 * 32 * LP + L * LP];
    float* s_m = sm;
    float* s_s = s_m + L;
    float* s_pm = s_s + L;               // [NSL][L]
    float* s_ps = s_pm + NSL * L;        // [NSL][L]
    float* Ks = s_ps + NSL * L;          // [32][LP]
    float* Vs = Ks + 32 * LP;            // [32][LP]
    float* A2s = Vs + 32 * LP;           // [L][LP]
    float* Qs = Ks;                      // [32][LQ]  (aliases Ks/Vs after phase 2)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const float cnd = b < B ? 1.f : 0.f;             // text-conditioned half first (stmogen.py:736-739)
    const float* mrow = mask + (long)(b % B) * T;
    const int Nseq = Nt + T;
    const float NEG = -1000000.f;
    const long D4 = 4 * L;

    auto load_kv = [&](int n, int c4, f32x4& kk, f32x4& vv) {
        if (n < Nt) {
            const float* r = tf + ((long)b * Nt + n) * 2 * L + c4;
            kk = *reinterpret_cast<const f32x4*>(r);
            vv = *reinterpret_cast<const f32x4*>(r + L);
            const float add = (1.f - cnd) * NEG;
#pragma unroll
            for (int j = 0; j < 4; ++j) { kk[j] += add; vv[j] *= cnd; }
        } else {
            const int t = n - Nt;
            const float m = mrow[t];
            const float* r = mf + (((long)b * T + t) * H + h) * D4 + c4;
            kk = *reinterpret_cast<const f32x4*>(r + L);
            vv = *reinterpret_cast<const f32x4*>(r + 2 * L);
            const float add = (1.f - m) * NEG;
#pragma unroll
            for (int j = 0; j < 4; ++j) { kk[j] += add; vv[j] *= m; }
        }
    };

    // ---- phase 1: column max / sum over the sequence (softmax dim=1, st_attention.py:155) ----
    {
        const int c4 = (tid % C4) * 4, sl = tid / C4;
        f32x4 m = {-3e38f, -3e38f, -3e38f, -3e38f}, s = {0.f, 0.f, 0.f, 0.f};
        for (int n = sl; n < Nseq; n += NSL) {
            f32x4 kk, vv;
            load_kv(n, c4, kk, vv);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float nm = fmaxf(m[j], kk[j]);
                s[j] = s[j] * expf(m[j] - nm) + expf(kk[j] - nm);
                m[j] = nm;
            }
        }
        *reinterpret_cast<f32x4*>(s_pm + sl * L + c4) = m;
        *reinterpret_cast<f32x4*>(s_ps + sl * L + c4) = s;
    }
    __syncthreads();
    if (tid < L) {
        float M = -3e38f;
        for (int i = 0; i < NSL; ++i) M = fmaxf(M, s_pm[i * L + tid]);
        float S = 0.f;
        for (int i = 0; i < NSL; ++i) S += s_ps[i * L + tid] * expf(s_pm[i * L + tid] - M);
        s_m[tid] = M;
        s_s[tid] = S;
    }
    __syncthreads();

    // ---- phase 2: A2[d][l] = sum_n softmaxK[n][d] V[n][l]  (st_attention.py:167) ----
    const bool mm_active = (L >= 64) || wave == 0;
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[MT][MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < MT; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    const int nch = (Nseq + 31) / 32;
    for (int ch = 0; ch < nch; ++ch) {
#pragma unroll
        for (int j = 0; j < (32 * C4) / 256; ++j) {
            const int i = tid + 256 * j;
            const int row = i / C4, c4 = (i % C4) * 4;
            const int n = ch * 32 + row;
            f32x4 e = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (n < Nseq) {
                f32x4 kk;
                load_kv(n, c4, kk, vv);
#pragma unroll
                for (int q = 0; q < 4; ++q) e[q] = expf(kk[q] - s_m[c4 + q]) / s_s[c4 + q];
            }
            *reinterpret_cast<f32x4*>(Ks + row * LP + c4) = e;
            *reinterpret_cast<f32x4*>(Vs + row * LP + c4) = vv;
        }
        __syncthreads();
        if (mm_active) {
#pragma unroll 4
            for (int ks = 0; ks < 16; ++ks) {
                const int n = 2 * ks + (lane >> 5);
                float a[MT], bb[MT];
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) a[mi] = Ks[n * LP + (wm * MT + mi) * 32 + (lane & 31)];
#pragma unroll
                for (int ni = 0; ni < MT; ++ni) bb[ni] = Vs[n * LP + (wn * MT + ni) * 32 + (lane & 31)];
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                    for (int ni = 0; ni < MT; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], bb[ni], acc[mi][ni], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    if (mm_active) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int ni = 0; ni < MT; ++ni)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int d = (wm * MT + mi) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                    const int l = (wn * MT + ni) * 32 + (lane & 31);
                    A2s[d * LP + l] = acc[mi][ni][reg];
                }
    }
    __syncthreads();

    // ---- phase 3: y_t[t] = softmax_L(q[t]) A2   (st_attention.py:164-169) ----
    const int ntc = (T + 31) / 32;
    for (int tc = 0; tc < ntc; ++tc) {
        {
            const int row = tid >> 3, sub = tid & 7;
            const int t = tc * 32 + row;
            constexpr int SEG = L / 8;
            float v[SEG];
            float mx = -3e38f;
            if (t < T) {
                const float* r = mf + (((long)b * T + t) * H + h) * D4 + 3 * L + sub * SEG;
#pragma unroll
                for (int j = 0; j < SEG; j += 4) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(r + j);
                    v[j] = x[0]; v[j + 1] = x[1]; v[j + 2] = x[2]; v[j + 3] = x[3];
                }
#pragma unroll
                for (int j = 0; j < SEG; ++j) mx = fmaxf(mx, v[j]);
            }
            mx = group_max(mx, 8);
            float s = 0.f;
            if (t < T) {
#pragma unroll
                for (int j = 0; j < SEG; ++j) { v[j] = expf(v[j] - mx); s += v[j]; }
            }
            s = group_sum(s, 8);
#pragma unroll
            for (int j = 0; j < SEG; ++j) Qs[row * LQ + sub * SEG + j] = (t < T) ? v[j] / s : 0.f;
        }
        __syncthreads();
        if (wave < L / 32) {
            f32x16 o;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll 4
            for (int ks = 0; ks < L / 2; ++ks) {
                const int d = 2 * ks + (lane >> 5);
                const float a = Qs[(lane & 31) * LQ + d];
                const float bb = A2s[d * LP + wave * 32 + (lane & 31)];
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, o, 0, 0, 0);
            }
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int t = tc * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                if (t < T) yt[((long)b * T + t) * (H * L) + h * L + wave * 32 + (lane & 31)] = o[reg];
            }
        }
        __syncthreads();
    }
}

}  // namespace

int mc_launch_body(const float* mf, long ldmf, const float* qkv, const float* wsm, float* ys,
                   long frames, int H, int L, int G, hipStream_t s) {
    MC_REQUIRE(L % 4 == 0 && L % G == 0, "body: L=%d G=%d unsupported", L, G);
    const size_t lds = sizeof(float) * ((size_t)4 * H * L + (size_t)L * (L / G) + (size_t)H * H);
    MC_REQUIRE(lds <= 160 * 1024, "body: tile does not fit LDS");
    if (frames <= 0) return MC_OK;
    hipLaunchKernelGGL(body_k, dim3((unsigned)frames), dim3(256), lds, s, mf, ldmf, qkv, wsm, ys, H, L, G);
    MC_LAUNCH_CHECK();
    return MC_OK;
}

int mc_launch_temporal(const float* mf, const float* tf, const float* mask, float* yt,
                       int B2, int B, int T, int Nt, int H, int L, hipStream_t s) {
    dim3 grid(B2 * H), blk(256);
    if (L == 128) hipLaunchKernelGGL(temporal_k<128>, grid, blk, 0, s, mf, tf, mask, yt, B, T, Nt, H);
    else if (L == 64) hipLaunchKernelGGL(temporal_k<64>, grid, blk, 0, s, mf, tf, mask, yt, B, T, Nt, H);
    else if (L == 32) hipLaunchKernelGGL(temporal_k<32>, grid, blk, 0, s, mf, tf, mask, yt, B, T, Nt, H);
    else { mc_set_error("temporal: latent_dim=%d unsupported (32, 64, 128)", L); return MC_ERR_ARG; }
    MC_LAUNCH_CHECK();
    return MC_OK;
}
