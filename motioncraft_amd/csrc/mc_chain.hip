// Register-chained row-panel kernels (gfx950, fp32 MFMA 32x32x2).
//
// Key property used here: with the weights as the MFMA "A" operand a wave computes C^T fragments,
// lane l owning output row (l & 31) and, per accumulator quad q, the 4 consecutive columns
// 8q + 4(l >> 5) + {0..3} of a 32-column chunk.  That is EXACTLY the B-operand fragment layout of
// the next GEMM over the same rows (k-group j = 4*chunk + q, k = 8j + 4(l >> 5) + i): chained
// per-row GEMMs (and the elementwise / per-row-LayerNorm work between them) need no data movement
// at all -- accumulators become operands.  Each wave owns 32 rows, a 128-row workgroup shares only
// the weight chunks, which stream through a double-buffered LDS slab with register prefetch
// (one barrier per 32-column chunk = per 64 MFMAs).
//
//   mlp2_k        Y = GELU(X W1^T + b1) W2 + b2       experts (gathered rows) and SFFN parts
//   gate_k        z = LN(x) + emb;  p = z Wp^T + bp;  cosine logits, softmax, top-2, counts
//   rowchain_k<0> a = GELU(w0 y0 + w1 y1);  mf = a Wproj^T + b
//   rowchain_k<1> qkv = LN(mf[:, :L]) Wqkv^T + b
#include "mc_common.h"
#include "mc_chain.h"
#include "mc_bodyphase.h"
#include <stdlib.h>

namespace {

// ---- weight chunk streaming -------------------------------------------------------------------
// A "chunk" is ROWS x KW floats of a row-major matrix with leading dimension ld, staged in LDS
// with row stride KW + 4 (conflict-free b128 fragment reads).
template <int ROWS, int KW>
struct ChunkStage {
    static constexpr int LDS_LD = KW + 4;
    static constexpr int F4 = ROWS * KW / 4;            // float4 per chunk
    static constexpr int PT = (F4 + 255) / 256;         // float4 per thread
    f32x4 r[PT];
    __device__ __forceinline__ void fetch(const float* __restrict__ W, long ld, int row0, int col0, int tid) {
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int idx = tid + 256 * i;
            if (F4 % 256 == 0 || idx < F4) {
                const int rr = idx / (KW / 4), cc = (idx % (KW / 4)) * 4;
                r[i] = *reinterpret_cast<const f32x4*>(W + (long)(row0 + rr) * ld + col0 + cc);
            }
        }
    }
    __device__ __forceinline__ void commit(float* lds, int tid) const {
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int idx = tid + 256 * i;
            if (F4 % 256 == 0 || idx < F4) {
                const int rr = idx / (KW / 4), cc = (idx % (KW / 4)) * 4;
                *reinterpret_cast<f32x4*>(lds + rr * LDS_LD + cc) = r[i];
            }
        }
    }
};

// acc[32 out][32 rows] = Wc[32 out][8*NJ] * X[32 rows][8*NJ] with CH independent accumulator chains
// (k-groups dealt round-robin, summed at the end).
template <int NJ, int CH = 2>
__device__ __forceinline__ f32x16 chunk_mma(const float* Wc, const f32x4 (&xf)[NJ], int lane) {
    constexpr int LDW = 8 * NJ + 4;
    f32x16 acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[c][q] = 0.f;
    const float* wp = Wc + (lane & 31) * LDW + (lane >> 5) * 4;
    // explicit software pipeline: the weight fragments of k-groups j+CH.. are read while the
    // 4*CH MFMAs of k-groups j.. execute
    f32x4 w[CH], wn[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) w[c] = *reinterpret_cast<const f32x4*>(wp + 8 * c);
#pragma unroll
    for (int j = 0; j < NJ; j += CH) {
        if (j + CH < NJ) {
#pragma unroll
            for (int c = 0; c < CH; ++c) wn[c] = *reinterpret_cast<const f32x4*>(wp + 8 * (j + CH + c));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[c][i], xf[j + c][i], acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < CH; ++c) w[c] = wn[c];
    }
#pragma unroll
    for (int c = 1; c < CH; ++c)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[0][q] += acc[c][q];
    return acc[0];
}

// per-row LayerNorm of a fragment-distributed row: lane l and lane l^32 hold the two halves
template <int NJ>
__device__ __forceinline__ void frag_layernorm(f32x4 (&x)[NJ], const float* __restrict__ gamma,
                                               const float* __restrict__ beta, int kq) {
    constexpr int L = 8 * NJ;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) s += x[j][0] + x[j][1] + x[j][2] + x[j][3];
    s += __shfl_xor(s, 32, 64);
    const float mean = s / (float)L;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            x[j][i] -= mean;
            q += x[j][i] * x[j][i];
        }
    q += __shfl_xor(q, 32, 64);
    const float rstd = rsqrtf(q / (float)L + 1e-5f);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 8 * j + kq);
        const f32x4 b = *reinterpret_cast<const f32x4*>(beta + 8 * j + kq);
#pragma unroll
        for (int i = 0; i < 4; ++i) x[j][i] = x[j][i] * rstd * g[i] + b[i];
    }
}

// =================================================================================================
// Fused 2-layer MLP (reference: tutel FusedExpertsNetwork; SFFN stmogen.py:596-607)
// =================================================================================================
template <int L, int MODE>
__global__ __launch_bounds__(256, 2) void mlp2_k(MlpArgs g) {
    constexpr int NJ = L / 8, NT = L / 32, HC = 32;
    using S1 = ChunkStage<HC, L>;      // W1 chunk  [32 hidden][L]
    using S2 = ChunkStage<L, HC>;      // W2^T chunk [L out][32 hidden]
    constexpr int MAXHID = 1024;
    __shared__ __attribute__((aligned(16))) float smem[2 * (HC * S1::LDS_LD + L * S2::LDS_LD) + MAXHID];
    constexpr int BUFSZ = HC * S1::LDS_LD + L * S2::LDS_LD;
    float* s_b1 = smem + 2 * BUFSZ;      // first-layer bias of this group (hidden <= 1024)
    auto W1s = [&](int b) { return smem + b * BUFSZ; };
    auto W2s = [&](int b) { return smem + b * BUFSZ + HC * S1::LDS_LD; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int grp, row0, nrows;
    int zs = (int)blockIdx.z, ns = g.nsplit;              // hidden slice of this workgroup, number of slices
    if constexpr (MODE == MLP_EXPERT) {
        const int real = *g.num_tiles;
        int bt = (int)blockIdx.x;
        if (g.dyn_split) {                                    // 1-D launch: ways chosen from the real tile count
            ns = mc_mlp_dyn_ways(real);
            zs = bt / max(real, 1);
            bt -= zs * real;
            if (zs >= ns) return;
        }
        if (bt >= real) return;
        const int t = xcd_remap(bt, real);
        grp = g.tile_group[t];
        row0 = g.tile_row0[t];
        nrows = g.tile_nrows[t];
    } else {
        grp = blockIdx.y;
        row0 = blockIdx.x * 128;
        nrows = min(128, g.M - row0);
    }
    const float* __restrict__ W1 = g.W1 + (long)grp * g.hidden * L;
    const float* __restrict__ W2t = g.W2t + (long)grp * L * g.hidden;
    const float* __restrict__ b1 = g.b1 + (long)grp * g.hidden;
    const float* __restrict__ b2 = g.b2 + (long)grp * L;

    for (int i = tid; i < g.hidden; i += 256) s_b1[i] = b1[i];
    const int r = wave * 32 + (lane & 31);
    const bool rok = r < nrows;
    const int kq = (lane >> 5) * 4;
    f32x4 xf[NJ];
    {
        long srow = rok ? row0 + r : row0;
        if constexpr (MODE == MLP_EXPERT) srow = rok ? g.src_row[row0 + r] : 0;
        const float* xp = g.X + (long)grp * g.x_gstride + srow * g.ldx + kq;
        // (rows past nrows read a valid row -- srow 0 / the last row of the matrix -- and are never stored)
#pragma unroll
        for (int j = 0; j < NJ; ++j) xf[j] = *reinterpret_cast<const f32x4*>(xp + 8 * j);
    }
    f32x16 acc2[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc2[t][q] = 0.f;

    // hidden chunks [hc0, nch) of this workgroup (all of them unless the hidden dimension is split over blockIdx.z)
    const int nch_all = g.hidden / HC;
    const int hc0 = (int)((long)nch_all * zs / ns), nch = (int)((long)nch_all * (zs + 1) / ns);
    S1 s1;
    S2 s2;
    s1.fetch(W1, L, hc0 * HC, 0, tid);
    s2.fetch(W2t, g.hidden, 0, hc0 * HC, tid);
    s1.commit(W1s(hc0 & 1), tid);
    s2.commit(W2s(hc0 & 1), tid);
    if (hc0 + 1 < nch) {
        s1.fetch(W1, L, (hc0 + 1) * HC, 0, tid);
        s2.fetch(W2t, g.hidden, 0, (hc0 + 1) * HC, tid);
    }
    __syncthreads();
    for (int hc = hc0; hc < nch; ++hc) {
        const int buf = hc & 1;
        if (hc + 1 < nch) {
            s1.commit(W1s(buf ^ 1), tid);
            s2.commit(W2s(buf ^ 1), tid);
        }
        if (hc + 2 < nch) {
            s1.fetch(W1, L, (hc + 2) * HC, 0, tid);
            s2.fetch(W2t, g.hidden, 0, (hc + 2) * HC, tid);
        }
        // FC1 chunk: 32 hidden units of this wave's 32 rows
        const f32x16 a1 = chunk_mma<NJ, 1>(W1s(buf), xf, lane);
        // bias + exact GELU; the C^T fragment IS the B-operand fragment of FC2 (k-group q)
        f32x4 hf[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(s_b1 + hc * HC + 8 * q + kq);
#pragma unroll
            for (int i = 0; i < 4; ++i) hf[q][i] = gelu_exact(a1[4 * q + i] + bb[i]);
        }
        // FC2 partial: out[L] += W2t[:, chunk] h
        const float* w2p = W2s(buf) + (lane & 31) * S2::LDS_LD + kq;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f32x4 w2 = *reinterpret_cast<const f32x4*>(w2p + t * 32 * S2::LDS_LD + 8 * q);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[i], hf[q][i], acc2[t], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    if (!rok) return;
    long drow = row0 + r;
    if constexpr (MODE == MLP_EXPERT) drow = g.dst_row[row0 + r];
    float* yrow = g.Y + (long)zs * g.y_sstride + (long)grp * g.y_gstride + drow * g.ldy;
    const bool add_b2 = zs == 0;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = t * 32 + 8 * q + kq;
            const f32x4 bb = add_b2 ? *reinterpret_cast<const f32x4*>(b2 + n) : f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 v = {acc2[t][4 * q] + bb[0], acc2[t][4 * q + 1] + bb[1], acc2[t][4 * q + 2] + bb[2], acc2[t][4 * q + 3] + bb[3]};
            *reinterpret_cast<f32x4*>(yrow + n) = v;
        }
}

// =================================================================================================
// mlp2d_k: mlp2_k<L, MODE> (L = 128 or 64) with the weight chunks staged by LDS-DMA (global_load_lds_dwordx4) instead of registers + ds_write.
// Per chunk every thread of mlp2_k issues 8 global loads, waits for them, and writes 8 x 16 bytes into LDS; here a wave issues 8 DMAs
// and nothing else (no staging registers, no store instructions, no compiler-placed wait in the MFMA stream).  DMA writes lane-linear,
// so the LDS images are unpadded and XOR-swizzled instead of padded:
//   W1 chunk [32 hidden rows][128 floats]: 16-byte chunk c of row r sits at position (c & 16) | ((c & 15) ^ (r & 15))   (512-byte rows:
//            the 16 rows of a b128 lane group hit 16 different slots of the 256-byte bank row)
//   W2 chunk [128 out rows][32 floats]:    chunk c of row r at position c ^ ((r >> 1) & 7)   (128-byte rows: row parity picks the half of
//            the bank row, (r >> 1) & 7 the slot -- conflict-free for the b128 lane groups {0-3, 12-15, 20-27})
// Same MFMA order and operands as mlp2_k: the same bits.
// =================================================================================================
__device__ __forceinline__ void dma16c(unsigned voff, const float* sbase, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte) : "memory");
}

template <int L, int MODE>
__global__ __launch_bounds__(256, 2) void mlp2d_k(MlpArgs g) {
    static_assert(L == 128 || L == 64, "mlp2d_k: L");
    constexpr int NJ = L / 8, NT = L / 32, HC = 32;
    constexpr int C1 = HC * L, C2 = L * HC, BUFSZ = C1 + C2, MAXHID = 1024;      // floats
    constexpr int NP = L / 32;             // 1 KB DMA pieces per matrix and wave
    constexpr int CPR = L / 4;             // 16-byte chunks per W1 row (a DMA instruction covers 256 / CPR rows)
    __shared__ __attribute__((aligned(16))) float smem[2 * BUFSZ + MAXHID];
    float* s_b1 = smem + 2 * BUFSZ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    int grp, row0, nrows;
    if constexpr (MODE == MLP_EXPERT) {
        const int real = *g.num_tiles;
        const int bt = (int)blockIdx.x;
        if (bt >= real) return;
        const int t = xcd_remap(bt, real);
        grp = g.tile_group[t];
        row0 = g.tile_row0[t];
        nrows = g.tile_nrows[t];
    } else {
        grp = blockIdx.y;
        row0 = blockIdx.x * 128;
        nrows = min(128, g.M - row0);
    }
    const int grp_u = __builtin_amdgcn_readfirstlane(grp);
    const float* __restrict__ W1 = g.W1 + (long)grp_u * g.hidden * L;
    const float* __restrict__ W2t = g.W2t + (long)grp_u * L * g.hidden;
    const float* __restrict__ b1 = g.b1 + (long)grp * g.hidden;
    const float* __restrict__ b2 = g.b2 + (long)grp * L;
    for (int i = tid; i < g.hidden; i += 256) s_b1[i] = b1[i];
    // DMA byte offsets of this lane (NP pieces per matrix and wave)
    //   W1: piece q = 64 / CPR rows: row (64 / CPR) (NP wave + q) + lane / CPR, LDS position p = lane % CPR <- logical chunk (p & 16) | ((p & 15) ^ (row & 15))
    //   W2: piece q = 8 rows: row 8 (NP wave + q) + (lane >> 3), LDS position lane & 7  <- logical chunk p ^ ((row >> 1) & 7)
    unsigned vo1[NP], vo2[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int r1 = (64 / CPR) * (NP * wave + q) + lane / CPR, p1 = lane % CPR;
        vo1[q] = (unsigned)((r1 * L + ((p1 & 16) | ((p1 & 15) ^ (r1 & 15))) * 4) * 4);
        const int r2 = 8 * (NP * wave + q) + (lane >> 3), p2 = lane & 7;
        vo2[q] = (unsigned)(((long)r2 * g.hidden + (p2 ^ ((r2 >> 1) & 7)) * 4) * 4);
    }
    const unsigned lds0 = (unsigned)(size_t)smem;
    auto issue = [&](int hc) {
        const unsigned l1 = lds0 + (unsigned)((hc & 1) * BUFSZ * 4) + (unsigned)(NP * wave_u) * 1024;
        const unsigned l2 = l1 + C1 * 4;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            dma16c(vo1[q], W1 + (long)hc * HC * L, l1 + q * 1024);
            dma16c(vo2[q], W2t + hc * HC, l2 + q * 1024);
        }
    };
    const int r = wave * 32 + (lane & 31);
    const bool rok = r < nrows;
    const int hf = lane >> 5, kq = hf * 4;
    f32x4 xf[NJ];
    {
        long srow = rok ? row0 + r : row0;
        if constexpr (MODE == MLP_EXPERT) srow = rok ? g.src_row[row0 + r] : 0;
        const float* xp = g.X + (long)grp * g.x_gstride + srow * g.ldx + kq;
#pragma unroll
        for (int j = 0; j < NJ; ++j) xf[j] = *reinterpret_cast<const f32x4*>(xp + 8 * j);
    }
    f32x16 acc2[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc2[t][q] = 0.f;
    const int nch = g.hidden / HC;
    issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int fr = lane & 31, s1x = fr & 15, s2x = (fr >> 1) & 7;
    for (int hc = 0; hc < nch; ++hc) {
        if (hc + 1 < nch) issue(hc + 1);          // the other buffer: last read in iteration hc - 1, behind the barrier every wave has passed
        const float* W1c = smem + (hc & 1) * BUFSZ + fr * L;
        const float* W2c = smem + (hc & 1) * BUFSZ + C1;
        // FC1 chunk: one dependent accumulator chain over the 16 k-groups (chunk_mma<NJ, 1>'s order), fragments one k-group ahead
        f32x16 a1;
#pragma unroll
        for (int q = 0; q < 16; ++q) a1[q] = 0.f;
        auto w1frag = [&](int j) { return *reinterpret_cast<const f32x4*>(W1c + (((2 * j + hf) & 16) | (((2 * j + hf) & 15) ^ s1x)) * 4); };
        f32x4 w = w1frag(0), wn;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (j + 1 < NJ) wn = w1frag(j + 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[i], xf[j][i], a1, 0, 0, 0);
            w = wn;
        }
        f32x4 hfr[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(s_b1 + hc * HC + 8 * q + kq);
#pragma unroll
            for (int i = 0; i < 4; ++i) hfr[q][i] = gelu_exact(a1[4 * q + i] + bb[i]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f32x4 w2 = *reinterpret_cast<const f32x4*>(W2c + (t * 32 + fr) * HC + (((2 * q + hf) ^ s2x) * 4));
#pragma unroll
                for (int i = 0; i < 4; ++i) acc2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[i], hfr[q][i], acc2[t], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of chunk hc + 1 have landed
        __syncthreads();
    }
    if (!rok) return;
    long drow = row0 + r;
    if constexpr (MODE == MLP_EXPERT) drow = g.dst_row[row0 + r];
    float* yrow = g.Y + (long)grp * g.y_gstride + drow * g.ldy;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = t * 32 + 8 * q + kq;
            const f32x4 bb = *reinterpret_cast<const f32x4*>(b2 + n);
            f32x4 v = {acc2[t][4 * q] + bb[0], acc2[t][4 * q + 1] + bb[1], acc2[t][4 * q + 2] + bb[2], acc2[t][4 * q + 3] + bb[3]};
            *reinterpret_cast<f32x4*>(yrow + n) = v;
        }
}

// The gate's per-token finish on fragment-distributed logits: lane l and lane l ^ 32 hold the two halves of token (l & 31)
// (this lane: experts e(r) = (r & 3) + 8 (r >> 2) + 4 hf, r = 0..7); ss = this lane's part of |p|^2.
// cosine logits, softmax, top-2 (lowest index on ties), renormalised gates, importance key, (choice, expert) counts in LDS.
__device__ __forceinline__ void gate_tail(const GateArgs& g, float ss, const float (&l8)[8], int hf, int lane, bool rok, long tok,
                                          int* s_cnt) {
    constexpr int MAXE = 16;
    ss += __shfl_xor(ss, 32, 64);
    const float denom = fmaxf(sqrtf(ss), 1e-12f);                    // F.normalize(dim=1)
    const float scale = g.logit_scale[0];
    // this lane holds the logits of experts e(r) = (r & 3) + 8 (r >> 2) + 4 hf, r = 0..7; its partner the other 8
    float lg[8];
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int e = (r & 3) + 8 * (r >> 2) + 4 * hf;
        lg[r] = e < g.E ? (l8[r] / denom) * scale : -INFINITY;
        mx = fmaxf(mx, lg[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        lg[r] = expf(lg[r] - mx);          // exp(-inf) = 0 for e >= E
        sum += lg[r];
    }
    sum += __shfl_xor(sum, 32, 64);
    auto merge = [&](float& v, int& e) {   // combine with the partner half: larger score, lowest index on ties
        const float pv = __shfl_xor(v, 32, 64);
        const int pe = __shfl_xor(e, 32, 64);
        if (pv > v || (pv == v && pe < e)) { v = pv; e = pe; }
    };
    float m1 = -1.f, m2 = -1.f;
    int c1 = 99, c2 = 99;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int e = (r & 3) + 8 * (r >> 2) + 4 * hf;
        lg[r] = lg[r] / sum;
        if (e < g.E && lg[r] > m1) { m1 = lg[r]; c1 = e; }
    }
    merge(m1, c1);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int e = (r & 3) + 8 * (r >> 2) + 4 * hf;
        if (e < g.E && e != c1 && lg[r] > m2) { m2 = lg[r]; c2 = e; }
    }
    merge(m2, c2);
    if (rok && lane < 32) {
        const float den = fmaxf(m1 + m2, 1.1920928955078125e-07f);   // normalize_gate, finfo(float32).eps
        g.idx[tok * 2] = c1;
        g.idx[tok * 2 + 1] = c2;
        g.gate[tok * 2] = m1 / den;
        g.gate[tok * 2 + 1] = m2 / den;
        g.key[tok] = __float_as_uint(m1);
        atomicAdd(&s_cnt[c1], 1);
        atomicAdd(&s_cnt[MAXE + c2], 1);
    }
}

// =================================================================================================
// Gate: LayerNorm + embedding + cosine projector + logits + softmax + top-2
// (st_attention.py:116-120 LN; MOE.forward :49-51; tutel cosine_top gate + moe_layer routing())
// =================================================================================================
template <int L>
__global__ __launch_bounds__(256, 2) void gate_k(GateArgs g) {
    constexpr int NJ = L / 8, PC = 256 / 32, MAXE = 16, LDS_SIM = 256 + 4;
    using SP = ChunkStage<32, L>;
    __shared__ __attribute__((aligned(16))) float smem[2 * 32 * SP::LDS_LD + 32 * LDS_SIM + 256];
    __shared__ int s_cnt[2 * MAXE];
    auto Ws = [&](int b) { return smem + b * 32 * SP::LDS_LD; };
    float* s_simT = smem + 2 * 32 * SP::LDS_LD;    // [32 expert rows (>= E zero)][256 + 4]: sim_n^T as an MFMA "A" operand
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 32 * 64; i += 256) {         // 32 rows x 64 float4 (pre-transposed at pack time: coalesced)
        const int e = i >> 6, j4 = (i & 63) * 4;
        *reinterpret_cast<f32x4*>(s_simT + e * LDS_SIM + j4) = *reinterpret_cast<const f32x4*>(g.sim_nT + e * 256 + j4);
    }
    float* s_bp = s_simT + 32 * LDS_SIM;           // projector bias [256]
    s_bp[tid] = g.bp[tid];
    if (tid < 2 * MAXE) s_cnt[tid] = 0;
    const long tok = g.tok0 + (long)blockIdx.x * 128 + wave * 32 + (lane & 31);
    const bool rok = tok < g.N;
    const int hf = lane >> 5, kq = hf * 4;
    // z = LN(x) * gamma + beta + embedding[(t,h)]
    f32x4 zf[NJ];
    {
        // rows past N read row N-1 instead (never stored): loads under `if (rok)` cost one exposed latency each
        const long tk = rok ? tok : g.N - 1;
        const float* xp = g.X + tk * g.ldx + kq;
        const float* ep = g.emb + (tk % g.emb_mod) * L + kq;
#pragma unroll
        for (int j = 0; j < NJ; ++j) zf[j] = *reinterpret_cast<const f32x4*>(xp + 8 * j);
        frag_layernorm<NJ>(zf, g.gamma, g.beta, kq);
        // embedding rows: one batch of NJ unconditional loads after the LayerNorm (holding them across it spills)
        f32x4 ef[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) ef[j] = *reinterpret_cast<const f32x4*>(ep + 8 * j);
#pragma unroll
        for (int j = 0; j < NJ; ++j) zf[j] += ef[j];
        if (rok) {
            float* zp = g.Z + tok * L + kq;
#pragma unroll
            for (int j = 0; j < NJ; ++j) *reinterpret_cast<f32x4*>(zp + 8 * j) = zf[j];
        }
    }
    // p = z Wp^T + bp in 8 chunks of 32 columns; |p|^2 on the VALU, p . sim_n chained on the MFMA
    // (the C^T fragment of a chunk is the B operand of the [32 -> experts] product)
    float ss = 0.f;
    f32x16 lacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    SP sp;                                 // (requesting this chunk before the row loads spills in this kernel: 256 VGPRs)
    sp.fetch(g.Wp, L, 0, 0, tid);
    sp.commit(Ws(0), tid);
    sp.fetch(g.Wp, L, 32, 0, tid);
    __syncthreads();
    const float* simp = s_simT + (lane & 31) * LDS_SIM + kq;
    for (int c = 0; c < PC; ++c) {
        const int buf = c & 1;
        if (c + 1 < PC) sp.commit(Ws(buf ^ 1), tid);
        if (c + 2 < PC) sp.fetch(g.Wp, L, (c + 2) * 32, 0, tid);
        const f32x16 p = chunk_mma<NJ>(Ws(buf), zf, lane);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(s_bp + c * 32 + 8 * q + kq);
            const f32x4 sv = *reinterpret_cast<const f32x4*>(simp + c * 32 + 8 * q);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = p[4 * q + i] + bb[i];
                ss += v * v;
                lacc = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[i], v, lacc, 0, 0, 0);
            }
        }
        __syncthreads();
    }
    {
        float l8[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) l8[r] = lacc[r];
        gate_tail(g, ss, l8, hf, lane, rok, tok, s_cnt);
    }
    __syncthreads();
    if (tid < 2 * MAXE && s_cnt[tid]) atomicAdd(&g.cnt[tid], s_cnt[tid]);
}

// Small batches (a few thousand tokens): gate_k's 128-token workgroups leave most CUs idle while each wave walks all 8
// projector chunks.  Here a workgroup owns 32 tokens and its 4 waves split the 8 chunks of p = z Wp^T + bp (2 each;
// every wave rebuilds the same row fragment): 4x the workgroups, a 4x shorter projector chain per wave.  Weight fragments
// are wave-private, so they come straight from L2 into registers (no LDS staging, no barrier in the chunk loop).
// The p fragments meet in LDS and wave 0 runs |p|^2 and the [256 -> experts] MFMA chain over them in gate_k's order
// (chunk, quad, element ascending), so scores, expert choices and importance keys are bit-identical to gate_k's: which
// of the two kernels a batch size selects never changes a routing decision.
template <int L>
__global__ __launch_bounds__(256) void gate_small_k(GateArgs g) {
    constexpr int NJ = L / 8, MAXE = 16, PC = 8;
    __shared__ float s_v[PC][16][64];          // p (+ bias) in C^T fragment layout: [chunk][accumulator element][lane]
    __shared__ int s_cnt[2 * MAXE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 2 * MAXE) s_cnt[tid] = 0;
    const long tok = g.tok0 + (long)blockIdx.x * 32 + (lane & 31);
    const bool rok = tok < g.N;
    const int hf = lane >> 5, kq = hf * 4;
    const int c0 = 2 * wave;
    // this wave's first weight chunk is requested before the row loads
    const float* wrow = g.Wp + (long)(lane & 31) * L + kq;
    f32x4 w[NJ], wn[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) w[j] = *reinterpret_cast<const f32x4*>(wrow + (long)c0 * 32 * L + 8 * j);
    f32x4 zf[NJ];
    {
        const long tk = rok ? tok : g.N - 1;
        const float* xp = g.X + tk * g.ldx + kq;
        const float* ep = g.emb + (tk % g.emb_mod) * L + kq;
#pragma unroll
        for (int j = 0; j < NJ; ++j) zf[j] = *reinterpret_cast<const f32x4*>(xp + 8 * j);
        f32x4 ef[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) ef[j] = *reinterpret_cast<const f32x4*>(ep + 8 * j);
        frag_layernorm<NJ>(zf, g.gamma, g.beta, kq);
#pragma unroll
        for (int j = 0; j < NJ; ++j) zf[j] += ef[j];
        if (rok && wave == 0) {
            float* zp = g.Z + tok * L + kq;
#pragma unroll
            for (int j = 0; j < NJ; ++j) *reinterpret_cast<f32x4*>(zp + 8 * j) = zf[j];
        }
    }
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
        const int c = c0 + cc;
        if (cc == 0) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) wn[j] = *reinterpret_cast<const f32x4*>(wrow + (long)(c + 1) * 32 * L + 8 * j);
        }
        f32x4 bb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bb[q] = *reinterpret_cast<const f32x4*>(g.bp + c * 32 + 8 * q + kq);
        // the same two round-robin accumulator chains as chunk_mma<NJ, 2> (gate_k), summed the same way
        f32x16 acc[2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][q] = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j][i], zf[j][i], acc[j & 1], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) s_v[c][4 * q + i][lane] = (acc[0][4 * q + i] + acc[1][4 * q + i]) + bb[q][i];
        if (cc == 0) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) w[j] = wn[j];
        }
    }
    // wave 0: sim_n^T fragments of the first chunks are requested before the barrier
    const float* simp = g.sim_nT + (lane & 31) * 256 + kq;
    f32x4 sv[4], svn[4];
    if (wave == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) sv[q] = *reinterpret_cast<const f32x4*>(simp + 8 * q);
    }
    __syncthreads();
    if (wave == 0) {
        float ss = 0.f;
        f32x16 lacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < PC; ++c) {
            if (c + 1 < PC) {
#pragma unroll
                for (int q = 0; q < 4; ++q) svn[q] = *reinterpret_cast<const f32x4*>(simp + (c + 1) * 32 + 8 * q);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = s_v[c][4 * q + i][lane];
                    ss += v * v;
                    lacc = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[q][i], v, lacc, 0, 0, 0);
                }
#pragma unroll
            for (int q = 0; q < 4; ++q) sv[q] = svn[q];
        }
        float l8[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) l8[r] = lacc[r];
        gate_tail(g, ss, l8, hf, lane, rok, tok, s_cnt);
    }
    __syncthreads();
    if (tid < 2 * MAXE && s_cnt[tid]) atomicAdd(&g.cnt[tid], s_cnt[tid]);
}

// =================================================================================================
// combproj_k: a = GELU(w0 y0 + w1 y1) (post-score combine, st_attention.py:52) ; mf = a Wproj^T + b
// lnqkv_k   : q|k|v = LN(body_value) Wqkv^T + b   (efficient_attention.py:32-38, one shared LayerNorm)
// Both: one B-operand fragment per wave (32 rows x L in L/2 VGPRs), N streamed in 32-column chunks.
// =================================================================================================
template <int L, int KIND>   // KIND 0: combproj, 1: lnqkv
__global__ __launch_bounds__(256, 2) void rowchain_k(RowChainArgs g) {
    constexpr int NJ = L / 8;
    using SP = ChunkStage<32, L>;
    __shared__ __attribute__((aligned(16))) float smem[2 * 32 * SP::LDS_LD + 4 * L];
    auto Ws = [&](int b) { return smem + b * 32 * SP::LDS_LD; };
    float* s_bias = smem + 2 * 32 * SP::LDS_LD;      // whole bias vector (Nout <= 4L): a global load per chunk would sit on the critical path
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < g.Nout; i += 256) s_bias[i] = g.bias[i];
    const long tok = g.tok0 + (long)blockIdx.x * 128 + wave * 32 + (lane & 31);
    // CFG twin aliasing (base layer 0): rows identical to their twin's are not produced at all
    const bool aliasing = g.alias.split_flag && *g.alias.split_flag == 0;
    if (aliasing && g.tok0 + (long)blockIdx.x * 128 >= g.alias.from) return;
    const bool rok = tok < g.N && !(aliasing && tok >= g.alias.from);
    const int kq = (lane >> 5) * 4;
    // output chunks [cs, ce) of this workgroup: all of them, or (small batches, gridDim.y > 1) one slice per blockIdx.y --
    // the row fragment is rebuilt by every slice, the serial chunk chain gets gridDim.y times shorter
    const int nc_all = g.Nout / 32;
    const int cs = nc_all * (int)blockIdx.y / (int)gridDim.y, ce = nc_all * ((int)blockIdx.y + 1) / (int)gridDim.y;
    SP sp;
    sp.fetch(g.W, L, cs * 32, 0, tid);    // first weight chunk requested before the row loads: its latency hides behind them
    f32x4 xf[NJ];
    if constexpr (KIND == 0) {
        const long tk = tok < g.N ? tok : 0;
        const float w0 = rok ? g.comb_w[2 * tk] : 0.f, w1 = rok ? g.comb_w[2 * tk + 1] : 0.f;
        const long ty = (g.twin_from > 0 && tk >= g.twin_from) ? tk - g.twin_from : tk;    // CFG twin: same expert outputs
        const float* y0 = g.X + 2 * ty * L + kq;
        // all 2*NJ row loads are issued unconditionally and back to back (a load under `if (w != 0)` makes the
        // compiler wait for each one in its own basic block); rows of dropped choices were never written, so
        // their (possibly NaN) contents are discarded by a select, not multiplied by 0
        f32x4 ya[NJ], yb[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            ya[j] = *reinterpret_cast<const f32x4*>(y0 + 8 * j);
            yb[j] = *reinterpret_cast<const f32x4*>(y0 + L + 8 * j);
        }
        const bool k0 = w0 != 0.f, k1 = w1 != 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) xf[j][i] = gelu_exact((k0 ? w0 * ya[j][i] : 0.f) + (k1 ? w1 * yb[j][i] : 0.f));
    } else {
        // rows past N read row N-1 instead (never stored): a load under `if (rok)` costs one exposed latency per j
        const float* xp = g.X + (tok < g.N ? tok : g.N - 1) * g.ldx + kq;
#pragma unroll
        for (int j = 0; j < NJ; ++j) xf[j] = *reinterpret_cast<const f32x4*>(xp + 8 * j);
        frag_layernorm<NJ>(xf, g.gamma, g.beta, kq);
    }
    sp.commit(Ws(0), tid);
    if (cs + 1 < ce) sp.fetch(g.W, L, (cs + 1) * 32, 0, tid);
    __syncthreads();
    float* orow = g.Y + tok * g.ldy + kq;
    // Order inside an iteration: MFMAs(c) -> commit(c+1) -> stores(c) -> fetch(c+2).  gfx9 counts loads and
    // stores in ONE in-order vmcnt, and the wait the compiler places before the commit must be valid on the
    // loop-entry path too, so it always drains everything issued before the fetch it waits for.  With the fetch
    // issued AFTER the stores of the same iteration, what gets drained is one MFMA phase old (free); with the
    // fetch issued before them (the obvious order) every chunk stalls on the write latency of its own stores.
    for (int c = cs; c < ce; ++c) {
        const f32x16 a = chunk_mma<NJ>(Ws((c - cs) & 1), xf, lane);
        if (c + 1 < ce) sp.commit(Ws(((c - cs) & 1) ^ 1), tid);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(s_bias + c * 32 + 8 * q + kq);
            const f32x4 v = {a[4 * q] + bb[0], a[4 * q + 1] + bb[1], a[4 * q + 2] + bb[2], a[4 * q + 3] + bb[3]};
            if (rok) *reinterpret_cast<f32x4*>(orow + c * 32 + 8 * q) = v;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (c + 2 < ce) sp.fetch(g.W, L, (c + 2) * 32, 0, tid);
        __syncthreads();
    }
}

}  // namespace

bool mc_mlp_supported(int L, int hidden) { return (L == 32 || L == 64 || L == 128) && hidden % 32 == 0 && hidden >= 32 && hidden <= 1024; }

int mc_launch_mlp(int mode, const MlpArgs& g, int groups, int max_tiles, hipStream_t s) {
    MC_REQUIRE(mc_mlp_supported(g.L, g.hidden), "fused mlp: L=%d hidden=%d unsupported", g.L, g.hidden);
    MC_REQUIRE(g.ldx % 4 == 0 && g.ldy % 4 == 0 && g.x_gstride % 4 == 0 && g.y_gstride % 4 == 0, "fused mlp: unaligned strides");
    dim3 grid;
    MC_REQUIRE(g.nsplit >= 1 && g.hidden / 32 >= g.nsplit, "fused mlp: %d hidden chunks cannot be split %d ways", g.hidden / 32, g.nsplit);
    if (mode == MLP_EXPERT) {
        if (max_tiles <= 0) return MC_OK;
        MC_REQUIRE(!g.dyn_split || g.nsplit == 4, "fused mlp: dyn_split launches are sized for 4 ways");
        grid = g.dyn_split ? dim3(max_tiles * 4, 1, 1) : dim3(max_tiles, 1, g.nsplit);
    } else {
        if (g.M <= 0) return MC_OK;
        grid = dim3(cdiv(g.M, 128), groups, g.nsplit);
    }
    // L = 128 / 64 without a hidden split: the LDS-DMA staged form (needs 32-bit byte offsets into one group's weights)
    const bool dma_form = g.dma && (g.L == 128 || g.L == 64) && g.nsplit == 1 && !g.dyn_split && (long)g.hidden * 128 * 4 < (1L << 31);
    if (mc_ledger_on) {
        char name[48];
        snprintf(name, sizeof(name), "%s<%d, %d>", dma_form ? "mlp2d_k" : "mlp2_k", g.L, mode == MLP_EXPERT ? 0 : 1);
        const double rows = mode == MLP_EXPERT ? (double)g.ledger_rows : (double)g.M * groups;
        MC_LEDGER(name, grid, rows * 4.0 * g.L * g.hidden);          // FC1 + FC2, multiply-add = 2
    }
    if (dma_form) {
        if (g.L == 128) {
            if (mode == MLP_EXPERT) hipLaunchKernelGGL((mlp2d_k<128, MLP_EXPERT>), grid, dim3(256), 0, s, g);
            else hipLaunchKernelGGL((mlp2d_k<128, MLP_PARTS>), grid, dim3(256), 0, s, g);
        } else {
            if (mode == MLP_EXPERT) hipLaunchKernelGGL((mlp2d_k<64, MLP_EXPERT>), grid, dim3(256), 0, s, g);
            else hipLaunchKernelGGL((mlp2d_k<64, MLP_PARTS>), grid, dim3(256), 0, s, g);
        }
        MC_LAUNCH_CHECK();
        return MC_OK;
    }
#define MC_MLP_CASE(LL)                                                                              \
    case LL:                                                                                         \
        if (mode == MLP_EXPERT) hipLaunchKernelGGL((mlp2_k<LL, MLP_EXPERT>), grid, dim3(256), 0, s, g); \
        else hipLaunchKernelGGL((mlp2_k<LL, MLP_PARTS>), grid, dim3(256), 0, s, g);                  \
        break;
    switch (g.L) {
        MC_MLP_CASE(128)
        MC_MLP_CASE(64)
        MC_MLP_CASE(32)
        default: mc_set_error("fused mlp: L=%d unsupported", g.L); return MC_ERR_ARG;
    }
#undef MC_MLP_CASE
    MC_LAUNCH_CHECK();
    return MC_OK;
}

int mc_launch_gate(const GateArgs& g, hipStream_t s) {
    MC_REQUIRE(g.E >= 2 && g.E <= 16, "gate: num_experts=%d unsupported (2..16)", g.E);
    if (g.zero_cnt) MC_HIP(hipMemsetAsync(g.cnt, 0, sizeof(int) * 32, s));
    if (g.N <= g.tok0) return MC_OK;
    if (mc_ledger_on) {
        char name[32];
        snprintf(name, sizeof(name), "%s<%d>", g.N - g.tok0 <= g.small_tokens ? "gate_small_k" : "gate_k", g.L);
        const dim3 lg(cdiv(g.N - g.tok0, g.N - g.tok0 <= g.small_tokens ? 32 : 128));
        MC_LEDGER(name, lg, 2.0 * (double)(g.N - g.tok0) * (g.L * 256.0 + 256.0 * g.E));       // cosine projector + logits
    }
    if (g.N - g.tok0 <= g.small_tokens) {        // latency-bound sizes (B <= 2 at 196 frames; B=1 -2.8 ms per 50 steps, B=4 +1 ms): 32-token workgroups
        dim3 grid(cdiv(g.N - g.tok0, 32));
        switch (g.L) {
            case 128: hipLaunchKernelGGL(gate_small_k<128>, grid, dim3(256), 0, s, g); break;
            case 64: hipLaunchKernelGGL(gate_small_k<64>, grid, dim3(256), 0, s, g); break;
            case 32: hipLaunchKernelGGL(gate_small_k<32>, grid, dim3(256), 0, s, g); break;
            default: mc_set_error("gate: L=%d unsupported", g.L); return MC_ERR_ARG;
        }
        MC_LAUNCH_CHECK();
        return MC_OK;
    }
    dim3 grid(cdiv(g.N - g.tok0, 128));
    switch (g.L) {
        case 128: hipLaunchKernelGGL(gate_k<128>, grid, dim3(256), 0, s, g); break;
        case 64: hipLaunchKernelGGL(gate_k<64>, grid, dim3(256), 0, s, g); break;
        case 32: hipLaunchKernelGGL(gate_k<32>, grid, dim3(256), 0, s, g); break;
        default: mc_set_error("gate: L=%d unsupported", g.L); return MC_ERR_ARG;
    }
    MC_LAUNCH_CHECK();
    return MC_OK;
}

// =================================================================================================
// projqkv_k: rowchain_k<0> and rowchain_k<1> in one pass.  The first L output columns of the projection
// (body_value) are exactly the row fragment the q/k/v GEMM needs: they stay in VGPRs, get the shared
// LayerNorm in fragment layout and become the B operand of the second weight stream -- body_value is
// never re-read from HBM and the 12 q/k/v chunks share the tile prologue of the 16 projection chunks.
// =================================================================================================
template <int L>
__global__ __launch_bounds__(256, 2) void projqkv_k(RowChainArgs g) {
    constexpr int NJ = L / 8, NC0 = 4 * L / 32, NC1 = 3 * L / 32, NKEEP = L / 32;
    using SP = ChunkStage<32, L>;
    __shared__ __attribute__((aligned(16))) float smem[2 * 32 * SP::LDS_LD + 7 * L];
    auto Ws = [&](int b) { return smem + b * 32 * SP::LDS_LD; };
    float* s_bias = smem + 2 * 32 * SP::LDS_LD;      // proj bias [4L] | qkv bias [3L]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4 * L; i += 256) s_bias[i] = g.bias[i];
    for (int i = tid; i < 3 * L; i += 256) s_bias[4 * L + i] = g.bias2[i];
    const long tok = g.tok0 + (long)blockIdx.x * 128 + wave * 32 + (lane & 31);
    const bool aliasing = g.alias.split_flag && *g.alias.split_flag == 0;
    if (aliasing && g.tok0 + (long)blockIdx.x * 128 >= g.alias.from) return;
    const bool rok = tok < g.N && !(aliasing && tok >= g.alias.from);
    const int kq = (lane >> 5) * 4;
    SP sp;
    // global chunk cg: 0 .. NC0-1 = projection rows, NC0 .. NC0+NC1-1 = q/k/v rows
    auto fetch = [&](int cg) {
        if (cg < NC0) sp.fetch(g.W, L, cg * 32, 0, tid);
        else sp.fetch(g.W2, L, (cg - NC0) * 32, 0, tid);
    };
    fetch(0);
    f32x4 xf[NJ];
    {
        const long tk = tok < g.N ? tok : 0;
        const float w0 = rok ? g.comb_w[2 * tk] : 0.f, w1 = rok ? g.comb_w[2 * tk + 1] : 0.f;
        const long ty = (g.twin_from > 0 && tk >= g.twin_from) ? tk - g.twin_from : tk;
        const float* y0 = g.X + 2 * ty * L + kq;
        f32x4 ya[NJ], yb[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            ya[j] = *reinterpret_cast<const f32x4*>(y0 + 8 * j);
            yb[j] = *reinterpret_cast<const f32x4*>(y0 + L + 8 * j);
        }
        const bool k0 = w0 != 0.f, k1 = w1 != 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) xf[j][i] = gelu_exact((k0 ? w0 * ya[j][i] : 0.f) + (k1 ? w1 * yb[j][i] : 0.f));
    }
    sp.commit(Ws(0), tid);
    fetch(1);
    __syncthreads();
    // Stores are UNCONDITIONAL: invalid lanes (rows past N, aliased twins of a partly aliased tile) write their garbage into the
    // 128 padding rows the caller allocates behind the last real token (RowChainArgs::pad_row), so every chunk
    // issues exactly 4 store instructions per wave with no branch around them -- and the weight fetch for chunk c+2 is issued
    // BEFORE the stores of chunk c, so the (in-order) vmcnt wait in front of the next commit reads "at most the 4 younger stores may
    // still be in flight" instead of draining them: round 3 measured 0.8 ms per step of store latency exposed through that drain
    // at B=64 (timing ablation without the stores).
    const long trow = rok ? tok : g.pad_row + (long)(wave * 32 + (lane & 31));
    float* orow = g.Y + trow * g.ldy + kq;
    float* qrow = g.Y2 + trow * g.ldy2 + kq;
    f32x4 bvf[NJ];                                   // body_value fragment = k-groups 4c + q of output chunks c < L/32
    // one chunk: MFMAs -> commit(next) -> fetch(next + 1) -> bias + stores
#define MC_PQ_CHUNK(cg, XF, OUT, BIAS0, KEEP)                                                              \
    {                                                                                                      \
        const f32x16 a = chunk_mma<NJ>(Ws((cg) & 1), XF, lane);                                           \
        if ((cg) + 1 < NC0 + NC1) sp.commit(Ws(((cg) & 1) ^ 1), tid);                                     \
        if ((cg) + 2 < NC0 + NC1) fetch((cg) + 2);                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                   \
            const f32x4 bb = *reinterpret_cast<const f32x4*>(s_bias + (BIAS0) + 8 * q + kq);              \
            const f32x4 v = {a[4 * q] + bb[0], a[4 * q + 1] + bb[1], a[4 * q + 2] + bb[2], a[4 * q + 3] + bb[3]}; \
            *reinterpret_cast<f32x4*>((OUT) + 8 * q) = v;                                                 \
            KEEP                                                                                           \
        }                                                                                                  \
        __syncthreads();                                                                                   \
    }
#pragma unroll
    for (int c = 0; c < NKEEP; ++c) MC_PQ_CHUNK(c, xf, orow + c * 32, c * 32, bvf[4 * c + q] = v;)
    // (runtime loops: unrolled, the per-chunk weight pointers get hoisted and spill; the loop-entry state of the vmcnt bookkeeping
    // equals the back-edge state -- the unrolled chunks above leave the same queue -- so the counted waits survive)
#pragma unroll 1
    for (int c = NKEEP; c < NC0; ++c) MC_PQ_CHUNK(c, xf, orow + c * 32, c * 32, )
    frag_layernorm<NJ>(bvf, g.gamma, g.beta, kq);
#pragma unroll 1
    for (int c = 0; c < NC1; ++c) MC_PQ_CHUNK(NC0 + c, bvf, qrow + c * 32, 4 * L + c * 32, )
#undef MC_PQ_CHUNK
}

int mc_launch_projqkv(const RowChainArgs& g, hipStream_t s) {
    MC_REQUIRE(g.Nout == 4 * g.L && g.ldy % 4 == 0 && g.ldy2 % 4 == 0 && g.W2 && g.bias2 && g.Y2, "projqkv: bad arguments");
    MC_REQUIRE(g.pad_row >= g.N, "projqkv: pad_row (128 padding rows of Y / Y2 behind the last token) not set");
    if (g.N <= g.tok0) return MC_OK;
    dim3 grid(cdiv(g.N - g.tok0, 128));
    if (mc_ledger_on) {
        char name[32];
        snprintf(name, sizeof(name), "projqkv_k<%d>", g.L);
        MC_LEDGER(name, grid, 2.0 * (double)mc_ledger_tokens(g) * 7.0 * g.L * g.L);        // proj [4L, L] + q/k/v [3L, L]
    }
    switch (g.L) {
        case 128: hipLaunchKernelGGL(projqkv_k<128>, grid, dim3(256), 0, s, g); break;
        case 64: hipLaunchKernelGGL(projqkv_k<64>, grid, dim3(256), 0, s, g); break;
        case 32: hipLaunchKernelGGL(projqkv_k<32>, grid, dim3(256), 0, s, g); break;
        default: mc_set_error("projqkv: L=%d unsupported", g.L); return MC_ERR_ARG;
    }
    MC_LAUNCH_CHECK();
    return MC_OK;
}

// =================================================================================================
// pqbody_k: projqkv_k AND the body-topology attention (static 12 x 12 + EfficientSelfAttention over the H parts of a frame,
// st_attention.py:123-134, efficient_attention.py:25-46) in one pass over FRAME-ALIGNED tiles: a workgroup owns FR = 128 / H whole
// frames (H = 12: 10 frames = 120 token rows, the last 8 rows of the 128-row MFMA tile idle), so every frame's q / k / v meet inside
// one workgroup.  The q/k/v weight stream is walked per 32-channel group (= 2 dynamic heads) in the order q, k, v; each 32 x 32
// C^T fragment (lane = token, registers = channels) is parked in one of two 18 KB LDS slots and read back transposed (lane =
// channel of one (frame, head), registers = the H parts): exactly body_reg_k's register layout, whose DPP-row contractions run
// unchanged.  q/k/v never exist in HBM and ys is written from here; only the raw body_value chunk comes back from L2 (this
// workgroup stored it into mf during the projection phase).
//   slot plan (S0 / S1 swap every channel group; one barrier per weight chunk, as in projqkv_k):
//     [q chunk -> Sq] b [k chunk -> Sk ; read q(Sq), softmax_16] b [v chunk -> Sq ; read k(Sk), softmax_H] b [read v(Sq): A = k^T v, y = q A]
//   20 (frame, head) units of 16 lanes per channel group = 5 wave passes over 4 waves: wave w takes frames 2w, 2w+1, the fifth pass
//   (frames 8, 9) rotates over the waves with the channel group.
// =================================================================================================
// NQ = H: the set produces all H parts of its frames; NQ < H: parts [h0, h0 + NQ) only (the fifth pass is cut 4 ways by parts:
// every wave rebuilds A = k^T v of the two odd frames and projects H / 4 of their query rows)
template <int L, int H>
__global__ __launch_bounds__(256, 2) void pqbody_k(RowChainArgs g_in) {
    RowChainArgs g = g_in;                       // (uniform copy: workgroups of the second token range swap in its bounds)
    long bidx = blockIdx.x;
    if (g.nblk1 > 0 && bidx >= g.nblk1) { bidx -= g.nblk1; g.tok0 = g.tok2; g.N = g.N2; }
    if (g.alias.split_flag && *g.alias.split_flag == 0 && g.tok0 + bidx * (long)BodyPhase<L, H>::TR >= g.alias.from) return;      // (before anything is staged)
    constexpr int NJ = L / 8, NC0 = 4 * L / 32, NG = L / 32, NKEEP = L / 32, NSEQ = NC0 + 3 * NG;
    using BP = BodyPhase<L, H>;
    constexpr int TR = BP::TR, XS = BP::XS;         // token rows of a tile, exchange slot row stride
    using SP = ChunkStage<32, L>;
    __shared__ __attribute__((aligned(16))) float smem[2 * 32 * SP::LDS_LD + 7 * L + BP::LDS_FLOATS];
    auto Ws = [&](int b) { return smem + b * 32 * SP::LDS_LD; };
    float* s_bias = smem + 2 * 32 * SP::LDS_LD;      // proj bias [4L] | qkv bias [3L]
    float* s_x = s_bias + 7 * L;                     // two exchange slots [128][XS]
    float* s_w = s_x + 2 * BP::SROWS * XS;           // softmax(body_weight) [H][H]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4 * L; i += 256) s_bias[i] = g.bias[i];
    for (int i = tid; i < 3 * L; i += 256) s_bias[4 * L + i] = g.bias2[i];
    for (int i = tid; i < H * H; i += 256) s_w[i] = g.wsm[i];
    const long tile_tok0 = g.tok0 + bidx * TR;
    const bool aliasing = g.alias.split_flag && *g.alias.split_flag == 0;
    if (aliasing && tile_tok0 >= g.alias.from) return;
    const int r = wave * 32 + (lane & 31);
    const long tok = tile_tok0 + r;
    const bool rok = r < TR && tok < g.N && !(aliasing && tok >= g.alias.from);
    const int kq = (lane >> 5) * 4;
    SP sp;
    // chunk sequence: 0 .. NC0-1 projection rows; then per channel group cg: q rows 32cg, k rows L + 32cg, v rows 2L + 32cg
    auto fetch = [&](int seq) {
        if (seq < NC0) sp.fetch(g.W, L, seq * 32, 0, tid);
        else {
            const int u = seq - NC0, cgi = u / 3, j = u - 3 * cgi;
            sp.fetch(g.W2, L, j * L + cgi * 32, 0, tid);
        }
    };
    fetch(0);
    f32x4 xf[NJ];
    {
        const long tk = tok < g.N ? tok : 0;
        const float w0 = rok ? g.comb_w[2 * tk] : 0.f, w1 = rok ? g.comb_w[2 * tk + 1] : 0.f;
        const long ty = (g.twin_from > 0 && tk >= g.twin_from) ? tk - g.twin_from : tk;
        const float* y0 = g.X + 2 * ty * L + kq;
        f32x4 ya[NJ], yb[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            ya[j] = *reinterpret_cast<const f32x4*>(y0 + 8 * j);
            yb[j] = *reinterpret_cast<const f32x4*>(y0 + L + 8 * j);
        }
        const bool k0 = w0 != 0.f, k1 = w1 != 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) xf[j][i] = gelu_exact((k0 ? w0 * ya[j][i] : 0.f) + (k1 ? w1 * yb[j][i] : 0.f));
    }
    sp.commit(Ws(0), tid);
    fetch(1);
    __syncthreads();
    // projection phase: as projqkv_k (unconditional stores, invalid lanes into the padding rows behind the last token)
    const long trow = rok ? tok : g.pad_row + (long)(wave * 32 + (lane & 31));
    float* orow = g.Y + trow * g.ldy + kq;
    f32x4 bvf[NJ];
#define MC_PB_CHUNK(seq, OUT, BIAS0, KEEP)                                                                 \
    {                                                                                                      \
        const f32x16 a = chunk_mma<NJ>(Ws((seq) & 1), xf, lane);                                          \
        sp.commit(Ws(((seq) & 1) ^ 1), tid);                                                               \
        fetch((seq) + 2);                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                   \
            const f32x4 bb = *reinterpret_cast<const f32x4*>(s_bias + (BIAS0) + 8 * q + kq);              \
            const f32x4 v = {a[4 * q] + bb[0], a[4 * q + 1] + bb[1], a[4 * q + 2] + bb[2], a[4 * q + 3] + bb[3]}; \
            *reinterpret_cast<f32x4*>((OUT) + 8 * q) = v;                                                 \
            KEEP                                                                                           \
        }                                                                                                  \
        __syncthreads();                                                                                   \
    }
#pragma unroll
    for (int c = 0; c < NKEEP; ++c) MC_PB_CHUNK(c, orow + c * 32, c * 32, bvf[4 * c + q] = v;)
#pragma unroll 1
    for (int c = NKEEP; c < NC0; ++c) MC_PB_CHUNK(c, orow + c * 32, c * 32, )
#undef MC_PB_CHUNK
    frag_layernorm<NJ>(bvf, g.gamma, g.beta, kq);

    // ---- q/k/v + body phase ----
    // one weight chunk: MFMAs on the LayerNormed body_value fragment -> commit / fetch of the stream -> fragment (+ bias) into an
    // exchange slot [token row][32 channels]
    auto qkv_chunk = [&](int seq, int bias0, float* slot) {
        const f32x16 a = chunk_mma<NJ>(Ws(seq & 1), bvf, lane);
        if (seq + 1 < NSEQ) sp.commit(Ws((seq & 1) ^ 1), tid);
        if (seq + 2 < NSEQ) fetch(seq + 2);
        __builtin_amdgcn_sched_barrier(0);
        float* xr = slot + r * XS + kq;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(s_bias + bias0 + 8 * q + kq);
            const f32x4 v = {a[4 * q] + bb[0], a[4 * q + 1] + bb[1], a[4 * q + 2] + bb[2], a[4 * q + 3] + bb[3]};
            *reinterpret_cast<f32x4*>(xr + 8 * q) = v;
        }
    };
    BodyPhase<L, H> bp(g, s_x, s_w, tile_tok0, aliasing, lane, wave);
#pragma unroll 1
    for (int cg = 0; cg < NG; ++cg) {
        const int seq = NC0 + 3 * cg;
        float* Sq = bp.slot(cg & 1);
        float* Sk = bp.slot((cg & 1) ^ 1);
        bp.begin_group();
        qkv_chunk(seq, 4 * L + cg * 32, Sq);                      // q
        __syncthreads();
        qkv_chunk(seq + 1, 5 * L + cg * 32, Sk);                  // k
        bp.after_k(Sq);
        __syncthreads();
        qkv_chunk(seq + 2, 6 * L + cg * 32, Sq);                  // v (every q read of Sq happened before the barrier above)
        bp.after_v(Sk);
        __syncthreads();
        bp.finish(Sq, cg);
        // (next group: q -> Sk, whose k reads are behind the barrier above; k -> Sq only after the next barrier, which every wave
        //  reaches after its v reads)
    }
}

int mc_launch_pqbody(const RowChainArgs& g, int H, hipStream_t s) {
    MC_REQUIRE((g.L == 128 || g.L == 64) && H == 12, "pqbody: L=%d H=%d unsupported (128 / 64, 12)", g.L, H);
    MC_REQUIRE(g.Nout == 4 * g.L && g.ldy == 4 * g.L && g.W2 && g.bias2 && g.wsm && g.ys, "pqbody: bad arguments");
    MC_REQUIRE(g.pad_row >= g.N, "pqbody: pad_row (128 padding rows of Y behind the last token) not set");
    MC_REQUIRE(g.tok0 % H == 0 && g.N % H == 0, "pqbody: token range [%ld, %ld) is not made of whole frames", g.tok0, g.N);
    if (g.N <= g.tok0) return MC_OK;
    const long frames = (g.N - g.tok0) / H;
    RowChainArgs gg = g;
    dim3 grid(cdiv(frames, 128 / H));
    gg.nblk1 = 0;
    if (g.nblk1 != 0) {           // second token range in the same launch (any non-zero nblk1 asks for it; the real count is set here)
        MC_REQUIRE(g.tok2 % H == 0 && g.N2 % H == 0 && g.N2 >= g.tok2 && g.pad_row >= g.N2, "pqbody: bad second token range [%ld, %ld)", g.tok2, g.N2);
        gg.nblk1 = (int)grid.x;
        grid.x += cdiv((g.N2 - g.tok2) / H, 128 / H);
    }
    if (mc_ledger_on) {       // proj + q/k/v per token; per frame the static topology (H x H mix of L-vectors) and the dynamic one (8 heads of linear attention over the H parts: k^T v and q (k^T v), [hd x hd] each)
        char name[32];
        snprintf(name, sizeof(name), "pqbody_k<%d", g.L);
        const double toks = (double)(mc_ledger_tokens(g)), hd = g.L / 8.0;      // (aliased twins -- the tail of the range, or the optional second range -- exit at once in the usual case)
        MC_LEDGER(name, grid, 2.0 * toks * 7.0 * g.L * g.L + (toks / H) * (2.0 * H * H * g.L + 8 * 2.0 * (2.0 * H * hd * hd)));
    }
    if (g.L == 128) hipLaunchKernelGGL((pqbody_k<128, 12>), grid, dim3(256), 0, s, gg);
    else hipLaunchKernelGGL((pqbody_k<64, 12>), grid, dim3(256), 0, s, gg);
    MC_LAUNCH_CHECK();
    return MC_OK;
}

int mc_launch_rowchain(int kind, const RowChainArgs& g, hipStream_t s) {
    MC_REQUIRE(g.Nout % 32 == 0 && g.ldy % 4 == 0, "rowchain: Nout=%d / ldy unsupported", g.Nout);
    if (g.N <= g.tok0) return MC_OK;
    dim3 grid(cdiv(g.N - g.tok0, 128), (g.N - g.tok0 <= g.split_tokens && (g.Nout / 32) % 4 == 0) ? 4 : 1);
    if (mc_ledger_on) {
        char name[32];
        snprintf(name, sizeof(name), "rowchain_k<%d, %d>", g.L, kind ? 1 : 0);
        MC_LEDGER(name, grid, 2.0 * (double)mc_ledger_tokens(g) * g.Nout * g.L);
    }
#define MC_RC_CASE(LL)                                                                    \
    case LL:                                                                              \
        if (kind == 0) hipLaunchKernelGGL((rowchain_k<LL, 0>), grid, dim3(256), 0, s, g);  \
        else hipLaunchKernelGGL((rowchain_k<LL, 1>), grid, dim3(256), 0, s, g);            \
        break;
    switch (g.L) {
        MC_RC_CASE(128)
        MC_RC_CASE(64)
        MC_RC_CASE(32)
        default: mc_set_error("rowchain: L=%d unsupported", g.L); return MC_ERR_ARG;
    }
#undef MC_RC_CASE
    MC_LAUNCH_CHECK();
    return MC_OK;
}
