// fp32 MFMA GEMM family for the STMoGen denoiser (gfx950).
//
//   C[M,N] = epilogue( prologue(A)[M,K] * W[N,K]^T + bias[N] )
//
// All matrices fp32 row-major, K contiguous in both operands (nn.Linear weight layout).
// One 256-thread workgroup (4 waves, 2x2) computes a 128x128 tile with
// v_mfma_f32_32x32x2_f32 (exact fp32, 157 TF chip peak); each wave owns 64x64 = 2x2 MFMA
// tiles (64 accumulator VGPRs).  Operands are staged through LDS in [rows][32+4] tiles,
// double buffered, read back as one ds_read_b128 per 4 MFMA k-steps: inside an 8-wide k-group
// lanes 0-31 take k = 8j..8j+3 and lanes 32-63 take k = 8j+4..8j+7, which is a permutation of the
// k order applied identically to A and W (the dot product is order-independent up to rounding).
// The +4 pad makes both the b128 reads and the b128 staging writes bank-conflict free.
//
// Modes (one kernel instantiation each):
//   GM_PLAIN  dense / part-grouped (blockIdx.y = group) with optional activation, residual,
//             row-periodic add table (positional embedding) and duplicate-row write (CFG halves)
//   GM_EXP1   expert FC1 over device-built slot tiles: A row = src_row[slot], C row = slot
//   GM_EXP2   expert FC2: A row = slot, C row = dst_row[slot]  (= 2*token + choice)
//   GM_COMB   A[r][k] = gelu(w0[r]*Y[2r][k] + w1[r]*Y[2r+1][k]) (post-score combine of the two
//             expert outputs of a token, dropped choices have w = 0), then dense GEMM
#include "mc_common.h"
#include "mc_gemm.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32, LD = BK + 4;

template <int MODE>
__global__ __launch_bounds__(256, 2) void gemm_k(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LD];
    float* As = smem;                  // [2][BM*LD]
    float* Bs = smem + 2 * BM * LD;    // [2][BN*LD]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int ntn = (g.N + BN - 1) / BN;
    int bid, grp = blockIdx.y;
    if constexpr (MODE == GM_EXP1 || MODE == GM_EXP2) {
        // the launch covers the worst-case tile count; remap over the tiles that exist
        const int real = *g.num_tiles * ntn;
        if ((int)blockIdx.x >= real) return;
        bid = xcd_remap(blockIdx.x, real);
    } else {
        bid = xcd_remap(blockIdx.x, gridDim.x);
    }
    const int tm = bid / ntn, tn = bid % ntn;
    int row0, nrows;
    if constexpr (MODE == GM_EXP1 || MODE == GM_EXP2) {
        grp = g.tile_group[tm];
        row0 = g.tile_row0[tm];
        nrows = g.tile_nrows[tm];
    } else {
        row0 = tm * BM;
        nrows = min(BM, g.M - row0);
        if (nrows <= 0) return;
    }
    const float* __restrict__ Ab = g.A + (long)grp * g.a_gstride + g.a_col;
    const float* __restrict__ Wb = g.W + (long)grp * g.w_gstride;

    // staging assignment: thread -> rows (tid>>3) + 32*i, k-column (tid&7)*4
    const int sr = tid >> 3;
    const int sk = (tid & 7) * 4;
    long arow_off[4];
    bool arow_ok[4];
    float cw0[4], cw1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int r = sr + 32 * i;
        arow_ok[i] = r < nrows;
        long srow = row0 + r;
        if constexpr (MODE == GM_EXP1) srow = arow_ok[i] ? g.src_row[row0 + r] : 0;
        if constexpr (MODE == GM_COMB) {
            arow_off[i] = 2 * srow * g.lda;
            cw0[i] = arow_ok[i] ? g.comb_w[2 * srow] : 0.f;
            cw1[i] = arow_ok[i] ? g.comb_w[2 * srow + 1] : 0.f;
        } else {
            arow_off[i] = srow * g.lda;
        }
    }
    long wrow_off[4];
    bool wrow_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int n = tn * BN + sr + 32 * i;
        wrow_ok[i] = n < g.N;
        wrow_off[i] = (long)n * g.ldw;
    }

    f32x4 ra[4], rb[4];
    auto load_tile = [&](int k0) {
        const int k = k0 + sk;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (arow_ok[i] && k < g.K) {
                if constexpr (MODE == GM_COMB) {
                    f32x4 y0 = {0.f, 0.f, 0.f, 0.f}, y1 = {0.f, 0.f, 0.f, 0.f};
                    if (cw0[i] != 0.f) y0 = *reinterpret_cast<const f32x4*>(Ab + arow_off[i] + k);
                    if (cw1[i] != 0.f) y1 = *reinterpret_cast<const f32x4*>(Ab + arow_off[i] + g.lda + k);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = gelu_exact(cw0[i] * y0[j] + cw1[i] * y1[j]);
                } else if (MODE == GM_PLAIN && g.a_scalar) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (k + j < g.K) v[j] = Ab[arow_off[i] + k + j];
                } else {
                    v = *reinterpret_cast<const f32x4*>(Ab + arow_off[i] + k);
                }
            }
            ra[i] = v;
            f32x4 w = {0.f, 0.f, 0.f, 0.f};
            if (wrow_ok[i] && k < g.K) w = *reinterpret_cast<const f32x4*>(Wb + wrow_off[i] + k);
            rb[i] = w;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4*>(As + buf * BM * LD + (sr + 32 * i) * LD + sk) = ra[i];
            *reinterpret_cast<f32x4*>(Bs + buf * BN * LD + (sr + 32 * i) * LD + sk) = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int nk = (g.K + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    const int frow = lane & 31;
    const int fk = (lane >> 5) * 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile((kt + 1) * BK);
        const float* Ap = As + buf * BM * LD + (wm * 64 + frow) * LD + fk;
        const float* Bp = Bs + buf * BN * LD + (wn * 64 + frow) * LD + fk;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 a0 = *reinterpret_cast<const f32x4*>(Ap + j * 8);
            f32x4 a1 = *reinterpret_cast<const f32x4*>(Ap + 32 * LD + j * 8);
            f32x4 b0 = *reinterpret_cast<const f32x4*>(Bp + j * 8);
            f32x4 b1 = *reinterpret_cast<const f32x4*>(Bp + 32 * LD + j * 8);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b0[i], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b1[i], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b0[i], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b1[i], acc[1][1], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // epilogue
    const float* __restrict__ bias = g.bias ? g.bias + (long)grp * g.b_gstride : nullptr;
    float* __restrict__ Cb = g.C + (long)grp * g.c_gstride + g.c_col;
    const float* __restrict__ Rb = g.R ? g.R + (long)grp * g.c_gstride + g.c_col : nullptr;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int r = wm * 64 + mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            if (r >= nrows) continue;
            long drow = row0 + r;
            if constexpr (MODE == GM_EXP2) drow = g.dst_row[row0 + r];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int n = tn * BN + wn * 64 + ni * 32 + (lane & 31);
                if (n >= g.N) continue;
                float v = acc[mi][ni][reg];
                if (bias) v += bias[n];
                v = apply_act(v, g.act);
                if (g.add) v += g.add[(long)((row0 + r) % g.add_mod) * g.ld_add + n];
                if (Rb) v += Rb[drow * g.ldr + n];
                Cb[drow * g.ldc + n] = v;
                if (g.dup_rows > 0) Cb[(drow + g.dup_rows) * g.ldc + n] = v;
            }
        }
    }
}

}  // namespace

int mc_launch_gemm(int mode, const GemmArgs& g, int groups, int max_tiles, hipStream_t stream) {
    const int ntn = cdiv(g.N, BN);
    int ntm = (mode == GM_EXP1 || mode == GM_EXP2) ? max_tiles : cdiv(g.M, BM);
    if (ntm <= 0 || ntn <= 0) return MC_OK;
    dim3 grid(ntm * ntn, groups > 0 ? groups : 1, 1);
    switch (mode) {
        case GM_PLAIN: hipLaunchKernelGGL(gemm_k<GM_PLAIN>, grid, dim3(256), 0, stream, g); break;
        case GM_EXP1: hipLaunchKernelGGL(gemm_k<GM_EXP1>, grid, dim3(256), 0, stream, g); break;
        case GM_EXP2: hipLaunchKernelGGL(gemm_k<GM_EXP2>, grid, dim3(256), 0, stream, g); break;
        case GM_COMB: hipLaunchKernelGGL(gemm_k<GM_COMB>, grid, dim3(256), 0, stream, g); break;
        default: mc_set_error("bad gemm mode %d", mode); return MC_ERR_ARG;
    }
    MC_LAUNCH_CHECK();
    return MC_OK;
}
