// MC-Attn cores (reference mogen/models/attentions/st_attention.py:105-179):
//   body_k      static 12x12 body topology + dynamic topology (EfficientSelfAttention over the
//               H body parts of one frame, efficient_attention.py:25-46), joint tile in LDS
//   temporal_k  temporal linear attention over text (+) motion tokens of one (sample, part):
//               column softmax of K over the sequence, A2 = K^T V and Y = softmax_L(Q) A2 on
//               v_mfma_f32_32x32x2_f32, operands staged through LDS
#include "mc_common.h"
#include "mc_kernels.h"
#include <utility>
#include <stdlib.h>

namespace {

// ---------------------------------------------------------------------------------------
// Body topology, register-resident: HD = L/8 lanes per (frame, head g); lane l of a group owns
// channel c = g*HD + l of all H parts: bv[h][c], q[h][c], k[h][c], v[h][c] (4H registers).
//   static : ys[h][c]  = sum_j softmax(body_weight)[h][j] * bv[j][c]              (in-lane, st_attention.py:123-128)
//   dynamic: k softmax over the H parts (in-lane), q softmax over the HD channels of the head
//            (group shuffles), A[d][l] = sum_h k[h][d] v[h][l] and y[h][l] = sum_d q[h][d] A[d][l]
//            with k/q broadcast inside the HD-lane group   (efficient_attention.py:25-46, mask == 1)
// No LDS, no barriers; every load/store instruction of a wave covers 256 contiguous bytes.
// ---------------------------------------------------------------------------------------
template <int HD, int H>
__global__ __launch_bounds__(256) void body_reg_k(const float* __restrict__ mf, long ldmf, const float* __restrict__ qkv,
                                                   const float* __restrict__ wsm, float* __restrict__ ys, long frames,
                                                   TwinAlias alias, long frame0) {
    constexpr int G = 8, L = G * HD;
    constexpr int GPW = 64 / HD;                 // (frame, head) groups per wave
    __shared__ float s_w[H * H];
    for (int i = threadIdx.x; i < H * H; i += 256) s_w[i] = wsm[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long grp = wave * GPW + lane / HD;     // global (frame, head) index
    const long frame = grp / G;
    const int gh = (int)(grp % G), l = lane % HD;
    if (frame >= frames) return;                 // whole HD-lane groups leave together
    if (alias.split_flag && frame0 + frame >= alias.from && *alias.split_flag == 0) return;   // identical to the twin frame: not produced
    const int c = gh * HD + l;
    const long tok0 = frame * H;
    float bv[H], q[H], k[H], v[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        bv[h] = mf[(tok0 + h) * ldmf + c];
        const float* r = qkv + (tok0 + h) * 3 * L + c;
        q[h] = r[0];
        k[h] = r[L];
        v[h] = r[2 * L];
    }
    // key: softmax over the H body parts (dim=1)
    {
        float m = k[0];
#pragma unroll
        for (int h = 1; h < H; ++h) m = fmaxf(m, k[h]);
        float s = 0.f;
#pragma unroll
        for (int h = 0; h < H; ++h) { k[h] = fast_exp2((k[h] - m) * LOG2E); s += k[h]; }
        const float rs = __frcp_rn(s);
#pragma unroll
        for (int h = 0; h < H; ++h) k[h] *= rs;
    }
    // query: softmax over the HD channels of the head
#pragma unroll
    for (int h = 0; h < H; ++h) {
        const float m = group_max(q[h], HD);
        const float e = fast_exp2((q[h] - m) * LOG2E);
        q[h] = e * __frcp_rn(group_sum(e, HD));
    }
    float* out = ys + frame * (H * L) + c;
    if constexpr (HD == 16) {
        // The head is one DPP row: both contractions over d walk the row by rotation.  With
        // src(s) = the lane a rotation by s reads, A_[s] = A[d = src(s)][l] and q[h] read through the SAME
        // rotation pair up again in y[h][l] = sum_s q[h][src(s)] A_[s] -- no lane ever needs to know d.
        float A_[16];
        auto contract_kv = [&](auto S) {
            float a = 0.f;
#pragma unroll
            for (int h = 0; h < H; ++h) a += row_ror<decltype(S)::value>(k[h]) * v[h];
            A_[decltype(S)::value] = a;
        };
        static_for_16(contract_kv);
#pragma unroll
        for (int h = 0; h < H; ++h) {
            float st = 0.f;
#pragma unroll
            for (int j = 0; j < H; ++j) st += s_w[h * H + j] * bv[j];
            float dy = 0.f;
            auto contract_qa = [&](auto S) { dy += row_ror<decltype(S)::value>(q[h]) * A_[decltype(S)::value]; };
            static_for_16(contract_qa);
            out[h * L] = st + (bv[h] + dy);
        }
    } else {
        // A[d][l] = sum_h k[h][d] v[h][l]  (this lane: column l, all d)
        float A[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            float a = 0.f;
#pragma unroll
            for (int h = 0; h < H; ++h) a += __shfl(k[h], d, HD) * v[h];
            A[d] = a;
        }
#pragma unroll
        for (int h = 0; h < H; ++h) {
            float st = 0.f;
#pragma unroll
            for (int j = 0; j < H; ++j) st += s_w[h * H + j] * bv[j];
            float dy = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) dy += __shfl(q[h], d, HD) * A[d];
            out[h * L] = st + (bv[h] + dy);
        }
    }
}

// ---------------------------------------------------------------------------------------
// Temporal linear attention, one workgroup per (sample b of the CFG-doubled batch, part h)
// (st_attention.py:137-170):
//   K = concat(key_text + (1-c)(-1e6), key_motion + (1-mask)(-1e6)) -> softmax over the 77+T tokens
//   V = concat(value_text * c, value_motion * mask);  A2 = K^T V  [L x L];  y_t = softmax_L(Q) A2
// Wave w owns output columns l in [32w, 32w+32) for ALL d: A2 lives in 16*L/32 accumulator
// registers per lane and -- C^T fragment == next B operand -- feeds the second contraction
// directly from registers (no A2 round trip through LDS).  K/V rows stream through one 32-row LDS
// chunk, softmax_L(Q) rows through a [32][L+4] slab read as b128 A fragments.
// ---------------------------------------------------------------------------------------
// LSPLIT (small batches: a few dozen workgroups, each bound by the serial MFMA chain of its 4 waves): the L output
// columns are cut into 32-wide slices on blockIdx.y; inside a workgroup wave w then owns ONE 32 x 32 tile of A2 (d tile w
// of the slice's columns) instead of all L/32 d tiles of its own columns, and the partial products of the second
// contraction (one d tile per wave) are summed through LDS.  4x the workgroups, a quarter of the MFMA chain per wave.
// PAIR (L = 64 models, even H; round 4): one workgroup owns TWO adjacent parts (h, h + 1) of a sample.  At L = 64 a part has two 32-wide
// d tiles, i.e. two MFMA waves, and half of every workgroup only staged and waited at the chunk barriers (temporal_k<64> ran at 33 % of
// the MFMA rate).  The pair's K / V / Q rows are staged as one 128-column slab (part p in columns [64 p, 64 p + 64)); waves 2 p, 2 p + 1 own
// part p's two output column tiles.  Same per-part arithmetic, half the workgroups, every wave on the MFMA.
template <int L, bool LSPLIT, bool PAIR = false>
__global__ __launch_bounds__(256, 3) void temporal_k(const float* __restrict__ mf, const float* __restrict__ tf,
                                                     const float* __restrict__ mask, float* __restrict__ yt,
                                                     int b0, int B, int T, int Nt, int H, const int* twin_flag, int skip_text) {
    static_assert(!PAIR || (L == 64 && !LSPLIT), "temporal_k: PAIR is the L = 64 whole-part form");
    constexpr int LW = PAIR ? 2 * L : L; // columns of the staged slabs
    constexpr int NT = L / 32;           // 32-wide d tiles of a part (= MFMA waves per part)
    constexpr int LP = LW + 4;
    constexpr int C4 = LW / 4;           // float4 columns per row
    constexpr int NSL = 256 / C4;        // row slices in the stats pass
    constexpr int NACC = LSPLIT ? 1 : NT;
    __shared__ __attribute__((aligned(16))) float sm[2 * LW + 2 * NSL * LW + 2 * 32 * LP + (LSPLIT ? NT * 32 * 33 : 0)];
    float* s_m = sm;
    float* s_s = s_m + LW;
    float* s_pm = s_s + LW;              // [NSL][LW]
    float* s_ps = s_pm + NSL * LW;       // [NSL][LW]
    float* Ks = s_ps + NSL * LW;         // [32][LP]
    float* Vs = Ks + 32 * LP;            // [32][LP]
    float* Qs = Ks;                      // [32][LP]  phase 3 reuses the K slab (43 KB total -> 3 workgroups per CU)
    float* Ps = Vs + 32 * LP;            // LSPLIT: [NT][32][33] partial y_t tiles of the waves
    const int ls = LSPLIT ? (int)blockIdx.y : 0;      // output column slice (LSPLIT)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int HB = PAIR ? H / 2 : H;                 // workgroups per sample
    const int b = b0 + blockIdx.x / HB, h = PAIR ? 2 * (blockIdx.x % HB) : blockIdx.x % HB;
    const int wp = PAIR ? wave >> 1 : 0, wl = PAIR ? wave & 1 : wave;      // MFMA role of this wave: part of the pair, 32-wide output column tile
    const float cnd = b < B ? 1.f : 0.f;             // text-conditioned half first (stmogen.py:736-739)
    // CFG twin aliasing (base layer 0): the motion rows of sample b >= B were not produced, they equal sample b - B's
    const int bm = (twin_flag && b >= B && *twin_flag == 0) ? b - B : b;
    const float* mrow = mask + (long)(b % B) * T;
    const int Nseq = Nt + T;
    const float NEG = -1000000.f;
    const long D4 = 4 * L;
    const int hf = lane >> 5;

    // Branch-free row fetch: the loads of a batch are issued back to back and nothing touches their
    // results until finish_kv (mask arithmetic at load time makes the compiler wait per row: one exposed
    // L2/HBM latency per row).  Row n >= Nseq is clamped (its result is discarded by the caller).
    auto row_ptr = [&](int n, int c4, int& t) -> const float* {
        const int nc = n < Nseq ? n : Nseq - 1;
        const bool txt = nc < Nt;
        t = txt ? 0 : nc - Nt;
        const int pp = PAIR ? c4 / L : 0, cc = PAIR ? c4 % L : c4;                     // part of the pair, column inside the part
        const float* rt = tf + ((long)b * Nt + (txt ? nc : 0)) * 2 * L + cc;            // [key | value] (the text rows serve every part)
        const float* rm = mf + (((long)bm * T + t) * H + h + pp) * D4 + L + cc;         // [.. | key | value | ..]
        return txt ? rt : rm;
    };
    auto issue_kv = [&](int n, int c4, f32x4& kk, f32x4& vv, float& mv) {
        int t;
        const float* r = row_ptr(n, c4, t);
        kk = *reinterpret_cast<const f32x4*>(r);
        vv = *reinterpret_cast<const f32x4*>(r + L);
        const float mm = mrow[t];
        mv = n < Nt ? cnd : mm;
    };
    auto issue_k = [&](int n, int c4, f32x4& kk, float& mv) {
        int t;
        const float* r = row_ptr(n, c4, t);
        kk = *reinterpret_cast<const f32x4*>(r);
        const float mm = mrow[t];
        mv = n < Nt ? cnd : mm;
    };

    // Unconditional half (round 5): its text keys all carry the -1e6 of st_attention.py:153 and its text values are multiplied by c = 0
    // (:161), so -- as long as the sample has at least ONE valid frame, whose key then sets the column maximum -- every text row's
    // softmax numerator underflows to exactly 0 and its value row is exactly 0: the rows contribute nothing, bit for bit.  Whole leading
    // blocks of them are skipped: multiples of the stats pass's row batch (so every remaining row keeps its slice / slot, i.e. its place in
    // the online max / sum) and of the 32-row chunks of phase 2 (so the MFMA pairs (n, n + 1) and their order stay what they were).
    constexpr int BATCH = LSPLIT ? 12 : 8;           // (small batches: 3 round trips to L2 instead of 5 for the 273 rows; the extra registers are free there)
    int skip1 = 0, skip2 = 0;
    if constexpr (!LSPLIT) {
        if (b >= B && skip_text) {                    // (uniform per workgroup)
            int v = 0;
            for (int t = tid; t < T; t += 256) v |= mrow[t] != 0.f;
            if (__syncthreads_or(v)) { skip1 = Nt / (NSL * BATCH) * (NSL * BATCH); skip2 = Nt / 32 * 32; }
        }
    }

    // ---- phase 1: column max / sum over the sequence (softmax dim=1, st_attention.py:155) ----
    // rows are taken in batches of 8 independent loads (a row-by-row online update serialises one
    // global-load latency per row); one rescale per batch instead of per row
    {
        const int c4 = (tid % C4) * 4, sl = tid / C4;
        // log2 domain: k2 = k * log2(e); exp(k - m) = exp2(k2 - m2) is one v_exp_f32
        f32x4 m = {-3e38f, -3e38f, -3e38f, -3e38f}, s = {0.f, 0.f, 0.f, 0.f};
        for (int n0 = skip1 + sl; n0 < Nseq; n0 += NSL * BATCH) {
            f32x4 kk[BATCH];
            float mv[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) issue_k(n0 + u * NSL, c4, kk[u], mv[u]);
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const bool ok = n0 + u * NSL < Nseq;
                const float add = (1.f - mv[u]) * NEG;
#pragma unroll
                for (int j = 0; j < 4; ++j) kk[u][j] = ok ? (kk[u][j] + add) * LOG2E : -3e38f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float nm = m[j];
#pragma unroll
                for (int u = 0; u < BATCH; ++u) nm = fmaxf(nm, kk[u][j]);
                float acc = s[j] * fast_exp2(m[j] - nm);
#pragma unroll
                for (int u = 0; u < BATCH; ++u) acc += fast_exp2(kk[u][j] - nm);    // exp2(-3e38 - nm) = 0 for the padding slots
                s[j] = acc;
                m[j] = nm;
            }
        }
        *reinterpret_cast<f32x4*>(s_pm + sl * LW + c4) = m;
        *reinterpret_cast<f32x4*>(s_ps + sl * LW + c4) = s;
    }
    __syncthreads();
    if (tid < LW) {
        float M = -3e38f;
        for (int i = 0; i < NSL; ++i) M = fmaxf(M, s_pm[i * LW + tid]);
        float S = 0.f;
        for (int i = 0; i < NSL; ++i) S += s_ps[i * LW + tid] * fast_exp2(s_pm[i * LW + tid] - M);
        s_m[tid] = M;                 // column max in the log2 domain
        s_s[tid] = 1.f / S;           // reciprocal of the column sum
    }
    __syncthreads();

    // ---- phase 2: A2[d][l] = sum_n softmaxK[n][d] V[n][l]  (st_attention.py:167) ----
    const bool mm_active = PAIR || wave < NT;
    f32x16 acc[NACC];
#pragma unroll
    for (int dt = 0; dt < NACC; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
    const int nch = (Nseq + 31) / 32;
    // register prefetch: the K/V rows of chunk ch+1 are requested before the MFMAs of chunk ch
    constexpr int SPT = (32 * C4) / 256;            // float4 (row, column) slots per thread per chunk
    f32x4 pk[SPT], pv[SPT];
    float pm[SPT];
    auto prefetch_kv = [&](int ch) {
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            const int i = tid + 256 * j;
            issue_kv(ch * 32 + i / C4, (i % C4) * 4, pk[j], pv[j], pm[j]);
        }
    };
    const int ch0 = skip2 / 32;
    prefetch_kv(ch0);
    for (int ch = ch0; ch < nch; ++ch) {
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            const int i = tid + 256 * j;
            const int row = i / C4, c4 = (i % C4) * 4;
            const int n = ch * 32 + row;
            f32x4 e = {0.f, 0.f, 0.f, 0.f}, v = {0.f, 0.f, 0.f, 0.f};
            if (n < Nseq) {
                const float add = (1.f - pm[j]) * NEG;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    e[q] = fast_exp2(fmaf(pk[j][q] + add, LOG2E, -s_m[c4 + q])) * s_s[c4 + q];
                    v[q] = pv[j][q] * pm[j];
                }
            }
            *reinterpret_cast<f32x4*>(Ks + row * LP + c4) = e;
            *reinterpret_cast<f32x4*>(Vs + row * LP + c4) = v;
        }
        __syncthreads();
        if (ch + 1 < nch) prefetch_kv(ch + 1);
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch loads ABOVE the MFMAs (the scheduler otherwise sinks them to their use)
        if (mm_active) {
#pragma unroll 4
            for (int ks = 0; ks < 16; ++ks) {
                const int n = 2 * ks + hf;
                if constexpr (LSPLIT) {      // tile (d tile = wave, column slice ls)
                    const float bb = Vs[n * LP + ls * 32 + (lane & 31)];
                    const float a = Ks[n * LP + wave * 32 + (lane & 31)];
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[0], 0, 0, 0);
                } else {
                    const float bb = Vs[n * LP + wp * L + wl * 32 + (lane & 31)];
#pragma unroll
                    for (int dt = 0; dt < NT; ++dt) {
                        const float a = Ks[n * LP + wp * L + dt * 32 + (lane & 31)];
                        acc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[dt], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- phase 3: y_t[t][l] = sum_d softmax_L(q[t])[d] A2[d][l]  (st_attention.py:164-169) ----
    // lane (l = 32*wave + (lane&31), half hf) holds A2[d][l] for d = 32 dt + 8 q + 4 hf + i in acc[dt][4q+i]:
    // exactly the B operand of k-group (dt, q); the A operand Q[t][same d] is one b128 read.
    const int ntc = (T + 31) / 32;
    constexpr int SEG = LW / 8;                      // 8 threads per query row (PAIR: 4 per part)
    constexpr int QG = PAIR ? 4 : 8;                 // threads that share one softmax row
    const int qrow = tid >> 3, qsub = tid & 7;
    const int qpart = PAIR ? qsub >> 2 : 0, qcol = PAIR ? (qsub & 3) * SEG : qsub * SEG;
    float qv[SEG];
    auto prefetch_q = [&](int tc) {
        const int t = tc * 32 + qrow;
        if (t < T) {
            const float* r = mf + (((long)bm * T + t) * H + h + qpart) * D4 + 3 * L + qcol;
#pragma unroll
            for (int j = 0; j < SEG; j += 4) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(r + j);
                qv[j] = x[0]; qv[j + 1] = x[1]; qv[j + 2] = x[2]; qv[j + 3] = x[3];
            }
        }
    };
    // (LSPLIT launches deal the 32-row output chunks over blockIdx.z: both workgroups of a (sample, part, slice) build the
    // same A2 tile, each projects half of the query rows)
    const int tc0 = LSPLIT ? (int)blockIdx.z : 0, tcs = LSPLIT ? (int)gridDim.z : 1;
    prefetch_q(tc0);
    for (int tc = tc0; tc < ntc; tc += tcs) {
        {
            const int t = tc * 32 + qrow;
            float mx = -3e38f;
            if (t < T) {
#pragma unroll
                for (int j = 0; j < SEG; ++j) mx = fmaxf(mx, qv[j]);
            }
            mx = group_max(mx, QG);
            float s = 0.f;
            if (t < T) {
#pragma unroll
                for (int j = 0; j < SEG; ++j) { qv[j] = fast_exp2((qv[j] - mx) * LOG2E); s += qv[j]; }
            }
            s = 1.f / group_sum(s, QG);
#pragma unroll
            for (int j = 0; j < SEG; j += 4) {
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
                if (t < T) o = f32x4{qv[j] * s, qv[j + 1] * s, qv[j + 2] * s, qv[j + 3] * s};
                *reinterpret_cast<f32x4*>(Qs + qrow * LP + qsub * SEG + j) = o;
            }
        }
        __syncthreads();
        if (tc + tcs < ntc) prefetch_q(tc + tcs);
        __builtin_amdgcn_sched_barrier(0);
        if (mm_active) {
            f32x16 o;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.f;
            const float* qp = Qs + (lane & 31) * LP + wp * L + 4 * hf;
            if constexpr (LSPLIT) {          // this wave's d tile only: a partial sum over d
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(qp + wave * 32 + 8 * q);
#pragma unroll
                    for (int i = 0; i < 4; ++i) o = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], acc[0][4 * q + i], o, 0, 0, 0);
                }
#pragma unroll
                for (int reg = 0; reg < 16; ++reg)
                    Ps[(wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hf) * 33 + (lane & 31)] = o[reg];
            } else {
#pragma unroll
                for (int dt = 0; dt < NT; ++dt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(qp + dt * 32 + 8 * q);
#pragma unroll
                        for (int i = 0; i < 4; ++i) o = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], acc[dt][4 * q + i], o, 0, 0, 0);
                    }
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int t = tc * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hf;
                    if (t < T) yt[((long)b * T + t) * (H * L) + (h + wp) * L + wl * 32 + (lane & 31)] = o[reg];
                }
            }
        }
        __syncthreads();
        if constexpr (LSPLIT) {              // y_t tile = sum of the waves' partials in d-tile order (fixed: deterministic)
            for (int i = tid; i < 32 * 32; i += 256) {
                const int tr = i >> 5, l = i & 31, t = tc * 32 + tr;
                float v = Ps[tr * 33 + l];
#pragma unroll
                for (int w = 1; w < NT; ++w) v += Ps[(w * 32 + tr) * 33 + l];
                if (t < T) yt[((long)b * T + t) * (H * L) + h * L + ls * 32 + l] = v;
            }
            // (Ps is rewritten only after the next chunk's barrier; Qs is rewritten before it, but no one reads Qs here)
        }
    }
}

}  // namespace

int mc_launch_body(const float* mf, long ldmf, const float* qkv, const float* wsm, float* ys,
                   long frames, int H, int L, int G, hipStream_t s, TwinAlias alias, long frame0) {
    MC_REQUIRE(G == 8 && (H == 12 || H == 8) && (L == 32 || L == 64 || L == 128), "body: H=%d L=%d G=%d unsupported", H, L, G);
    if (frames <= 0) return MC_OK;
    const int hd = L / G;
    const long groups = frames * G;               // (frame, head) groups of hd lanes
    const long waves = (groups * hd + 63) / 64;
    dim3 grid((unsigned)((waves + 3) / 4));
    MC_LEDGER("body_reg_k", grid, (double)frames * (2.0 * H * H * L + G * 2.0 * (2.0 * H * hd * hd)));       // static mix + per-head linear attention over the H parts
#define MC_BODY_CASE(HH)                                                                                          \
    if (hd == 16) hipLaunchKernelGGL((body_reg_k<16, HH>), grid, dim3(256), 0, s, mf, ldmf, qkv, wsm, ys, frames, alias, frame0);    \
    else if (hd == 8) hipLaunchKernelGGL((body_reg_k<8, HH>), grid, dim3(256), 0, s, mf, ldmf, qkv, wsm, ys, frames, alias, frame0); \
    else hipLaunchKernelGGL((body_reg_k<4, HH>), grid, dim3(256), 0, s, mf, ldmf, qkv, wsm, ys, frames, alias, frame0);
    if (H == 12) { MC_BODY_CASE(12) } else { MC_BODY_CASE(8) }   // motionx: 12 parts; human_ml3d / kit_ml: 8
#undef MC_BODY_CASE
    MC_LAUNCH_CHECK();
    return MC_OK;
}

int mc_launch_temporal(const float* mf, const float* tf, const float* mask, float* yt,
                       int b0, int nb, int B, int T, int Nt, int H, int L, hipStream_t s, const int* twin_flag, long lsplit_max, bool pair, bool skip_text) {
    const int sk = skip_text ? 1 : 0;
    if (nb <= 0) return MC_OK;
    dim3 grid(nb * H), blk(256);
    if (mc_ledger_on) {       // per (sample, part): K^T V over Nt + T keys [L x L] and Q (K^T V) over T queries (efficient_attention.py:25-46)
        char name[32];
        snprintf(name, sizeof(name), "temporal_k<%d", L);
        dim3 lg = grid;          // (the grid of the form launched below)
        if (L >= 64 && (long)nb * H <= lsplit_max) { lg.y = L / 32; lg.z = (long)nb * H * (L / 32) * 2 <= 256 ? 2 : 1; }
        else if (L == 64 && pair && H % 2 == 0) lg.x = nb * (H / 2);
        MC_LEDGER(name, lg, (double)nb * H * (2.0 * (Nt + T) * L * L + 2.0 * T * L * L));
    }
    // small batches: a few dozen (sample, part) workgroups, each bound by its waves' serial MFMA chain -> cut the L output
    // columns into 32-wide slices on blockIdx.y (temporal_k<L, true>).  50-step DDIM, 196 frames: B=1 70.4 -> 66.4 ms, B=2 96.1 ->
    // 92.3, B=4 128.3 -> 127.0; from B=8 (192 workgroups) the unsplit kernel is faster again
    if (L >= 64 && (long)nb * H <= lsplit_max) {
        grid.y = L / 32;
        grid.z = (long)nb * H * (L / 32) * 2 <= 256 ? 2 : 1;      // still one workgroup per CU: the output rows halved as well (B=1: 96 -> 192 workgroups)
        if (L == 128) hipLaunchKernelGGL((temporal_k<128, true>), grid, blk, 0, s, mf, tf, mask, yt, b0, B, T, Nt, H, twin_flag, sk);
        else hipLaunchKernelGGL((temporal_k<64, true>), grid, blk, 0, s, mf, tf, mask, yt, b0, B, T, Nt, H, twin_flag, sk);
        MC_LAUNCH_CHECK();
        return MC_OK;
    }
    if (L == 128) hipLaunchKernelGGL((temporal_k<128, false>), grid, blk, 0, s, mf, tf, mask, yt, b0, B, T, Nt, H, twin_flag, sk);
    else if (L == 64 && pair && H % 2 == 0) {       // two parts per workgroup: every wave on the MFMA
        grid.x = nb * (H / 2);
        hipLaunchKernelGGL((temporal_k<64, false, true>), grid, blk, 0, s, mf, tf, mask, yt, b0, B, T, Nt, H, twin_flag, sk);
    } else if (L == 64) hipLaunchKernelGGL((temporal_k<64, false>), grid, blk, 0, s, mf, tf, mask, yt, b0, B, T, Nt, H, twin_flag, sk);
    else if (L == 32) hipLaunchKernelGGL((temporal_k<32, false>), grid, blk, 0, s, mf, tf, mask, yt, b0, B, T, Nt, H, twin_flag, sk);
    else { mc_set_error("temporal: latent_dim=%d unsupported (32, 64, 128)", L); return MC_ERR_ARG; }
    MC_LAUNCH_CHECK();
    return MC_OK;
}
