// Fused two-layer MLP  Y = GELU(X W1^T + b1) W2 + b2  over row tiles (gfx950, fp32 MFMA).
//
// Used for the MoE expert FFNs (tutel FusedExpertsNetwork, gathered token rows per expert) and for
// the 12 part-wise SFFN FFNs (reference stmogen.py:596-607).  K1 = N2 = L (32/64/128), hidden = 4L
// or ffn_dim.  The hidden activations never touch HBM:
//   * each wave owns 32 rows of a 128-row tile; its X fragment (32 x L) stays in VGPRs for the
//     whole tile in MFMA B-operand layout (L/2 registers per lane),
//   * the hidden dimension is processed in chunks of 32: FC1 chunk (32 rows x 32 hidden, K = L)
//     -> bias + exact GELU -> 32 x 36 LDS slab private to the wave -> FC2 partial (32 rows x L,
//     K = 32) accumulated in 16*L/32 accumulator registers,
//   * only the weight chunks (W1[32][L], W2^T[L][32]) are shared by the 4 waves: staged through LDS,
//     next chunk prefetched into registers while the current one is multiplied.
// 54 KB LDS + <= 256 VGPRs -> 2 workgroups per CU.  As in mc_gemm.hip the weights are the MFMA "A"
// operand (C^T fragments) so hidden slabs and outputs are written as float4 per lane.
#include "mc_common.h"
#include "mc_mlp.h"

namespace {

constexpr int HC = 32;        // hidden chunk
constexpr int LDH = HC + 4;   // row stride of the hidden slab / W2 chunk

template <int L, int MODE>
__global__ __launch_bounds__(256, 2) void mlp_k(MlpArgs g) {
    constexpr int NJ = L / 8;          // 8-wide k groups of the first GEMM
    constexpr int NT = L / 32;         // 32-col output tiles of the second GEMM
    constexpr int LDX = L + 4;         // row stride of the W1 chunk
    constexpr int WPT = L / 32;        // float4 per thread per weight chunk (both chunks are 32*L floats)
    __shared__ __attribute__((aligned(16))) float smem[128 * LDH + HC * LDX + L * LDH];
    float* Hs = smem;                  // [128][LDH]
    float* W1s = Hs + 128 * LDH;       // [HC][LDX]
    float* W2s = W1s + HC * LDX;       // [L][LDH]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int grp, row0, nrows;
    if constexpr (MODE == MLP_EXPERT) {
        const int real = *g.num_tiles;
        if ((int)blockIdx.x >= real) return;
        const int t = xcd_remap(blockIdx.x, real);
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

    // ---- X fragment of this wave's 32 rows, resident in registers ----
    const int r = wave * 32 + (lane & 31);
    const bool rok = r < nrows;
    const int kq = (lane >> 5) * 4;
    f32x4 xf[NJ];
    {
        long srow = row0 + r;
        if constexpr (MODE == MLP_EXPERT) srow = rok ? g.src_row[row0 + r] : 0;
        const float* xp = g.X + (long)grp * g.x_gstride + srow * g.ldx + kq;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            xf[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (rok) xf[j] = *reinterpret_cast<const f32x4*>(xp + 8 * j);
        }
    }

    // ---- weight chunk staging ----
    f32x4 p1[WPT], p2[WPT];
    auto prefetch = [&](int hc) {
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int idx = tid + 256 * i;
            const int r1 = idx / (L / 4), c1 = (idx % (L / 4)) * 4;       // W1 chunk [HC][L]
            p1[i] = *reinterpret_cast<const f32x4*>(W1 + (long)(hc * HC + r1) * L + c1);
            const int r2 = idx >> 3, c2 = (idx & 7) * 4;                  // W2^T chunk [L][HC]
            p2[i] = *reinterpret_cast<const f32x4*>(W2t + (long)r2 * g.hidden + hc * HC + c2);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int idx = tid + 256 * i;
            *reinterpret_cast<f32x4*>(W1s + (idx / (L / 4)) * LDX + (idx % (L / 4)) * 4) = p1[i];
            *reinterpret_cast<f32x4*>(W2s + (idx >> 3) * LDH + (idx & 7) * 4) = p2[i];
        }
    };

    f32x16 acc2[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc2[t][q] = 0.f;

    const int nch = g.hidden / HC;
    prefetch(0);
    commit();
    __syncthreads();
    float* hrow = Hs + (wave * 32 + (lane & 31)) * LDH;
    for (int hc = 0; hc < nch; ++hc) {
        if (hc + 1 < nch) prefetch(hc + 1);
        // FC1: D[hid][row] += W1[hid][k] X[row][k], two accumulator chains
        f32x16 a1 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x16 a1b = a1;
        const float* w1p = W1s + (lane & 31) * LDX + kq;
#pragma unroll
        for (int j = 0; j < NJ; j += 2) {
            const f32x4 wa = *reinterpret_cast<const f32x4*>(w1p + 8 * j);
            const f32x4 wb = *reinterpret_cast<const f32x4*>(w1p + 8 * (j + 1));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[i], xf[j][i], a1, 0, 0, 0);
                a1b = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[i], xf[j + 1][i], a1b, 0, 0, 0);
            }
        }
        // bias + exact GELU -> this wave's rows of the hidden slab (lane: row, 4 consecutive hidden units per quad)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int hq = 8 * q + kq;
            const f32x4 bb = *reinterpret_cast<const f32x4*>(b1 + hc * HC + hq);
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = gelu_exact(a1[4 * q + i] + a1b[4 * q + i] + bb[i]);
            *reinterpret_cast<f32x4*>(hrow + hq) = v;
        }
        __builtin_amdgcn_wave_barrier();   // slab rows are produced and consumed by the same wave (LDS is in-order per wave)
        // FC2 partial: D[out][row] += W2t[out][hid] H[row][hid]
#pragma unroll
        for (int j = 0; j < HC / 8; ++j) {
            const f32x4 hf = *reinterpret_cast<const f32x4*>(hrow + 8 * j + kq);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f32x4 w2 = *reinterpret_cast<const f32x4*>(W2s + (t * 32 + (lane & 31)) * LDH + 8 * j + kq);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[i], hf[i], acc2[t], 0, 0, 0);
            }
        }
        __syncthreads();                   // every wave is done with this weight chunk
        if (hc + 1 < nch) {
            commit();
            __syncthreads();
        }
    }

    // ---- epilogue: lane owns one output row, 4 consecutive columns per accumulator quad ----
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

template <int MODE>
int launch(const MlpArgs& g, dim3 grid, hipStream_t s) {
    switch (g.L) {
        case 128: hipLaunchKernelGGL((mlp_k<128, MODE>), grid, dim3(256), 0, s, g); break;
        case 64: hipLaunchKernelGGL((mlp_k<64, MODE>), grid, dim3(256), 0, s, g); break;
        case 32: hipLaunchKernelGGL((mlp_k<32, MODE>), grid, dim3(256), 0, s, g); break;
        default: mc_set_error("fused mlp: L=%d unsupported (32, 64, 128)", g.L); return MC_ERR_ARG;
    }
    MC_LAUNCH_CHECK();
    return MC_OK;
}

}  // namespace

bool mc_mlp_supported(int L, int hidden) { return (L == 32 || L == 64 || L == 128) && hidden % HC == 0 && hidden >= HC; }

int mc_launch_mlp(int mode, const MlpArgs& g, int groups, int max_tiles, hipStream_t s) {
    MC_REQUIRE(mc_mlp_supported(g.L, g.hidden), "fused mlp: L=%d hidden=%d unsupported", g.L, g.hidden);
    MC_REQUIRE(g.ldx % 4 == 0 && g.ldy % 4 == 0 && g.x_gstride % 4 == 0 && g.y_gstride % 4 == 0, "fused mlp: unaligned strides");
    if (mode == MLP_EXPERT) {
        if (max_tiles <= 0) return MC_OK;
        return launch<MLP_EXPERT>(g, dim3(max_tiles), s);
    }
    if (g.M <= 0) return MC_OK;
    return launch<MLP_PARTS>(g, dim3(cdiv(g.M, 128), groups), s);
}
