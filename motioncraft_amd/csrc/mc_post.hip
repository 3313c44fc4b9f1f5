// Result post-processing of the 322-d SMPL-X motion vector, on the device:
//   de-normalise (pred * std + mean)                                    tools/visualize.py:217-220
//   re-pack 322 -> poses[165] (body 0:66, jaw 66:69 <- 156:159, hands 75:165 <- 66:156),
//                  expressions[100] <- 209:309, trans[3] <- 309:312        tools/visualize.py:236-243, s2g_test.py:289-297
//   Gaussian temporal filter per channel, scipy.ndimage.gaussian_filter(mode="nearest") semantics
//   (taps = normalised exp(-x^2 / 2 sigma^2), radius int(4 sigma + .5), edge frames replicated)
//                                                                       tools/visualize.py:39-44,244-246
// One thread per (sample, frame, output channel); the whole [T,322] slab of a sample is a few hundred KB and
// stays in L2, so the 2r+1 strided taps are cache hits: HBM traffic = one read of pred + one write of the outputs.
#include "mc_common.h"
#include "mc_kernels.h"

namespace {

constexpr int NPOSE = 165, NEXPR = 100, NTRANS = 3, NOUT = NPOSE + NEXPR + NTRANS;

struct PostArgs {
    const float* pred;        // [B][T][322] normalised
    const int* lengths;       // [B] valid frames (filter support is clamped to [0, len))
    const int* rows;          // stitched mode (B == 1, T == stitched frames): frame t reads row rows[t] of pred [*, 322]
    const double* mean;       // [322]
    const double* stdv;       // [322]
    const double* taps;       // 4 tables of MAXTAP doubles: centre at [radius[g]]
    int radius[4];            // per group: body+jaw, hands, trans, expr;  -1 = not filtered
    int stats_f32;            // de-normalise in fp32 (two roundings, numpy float32 * float32 + float32)
    int B, T, C;
    double* poses;            // [B][T][165]
    double* expr;             // [B][T][100]
    double* trans;            // [B][T][3]
};

constexpr int MAXTAP = 129;

// numpy evaluates pred * std + mean as two ufunc passes (two roundings): keep the compiler from fusing them into
// one FMA (hipcc contracts by default, and HIP's __fmul_rn / __fadd_rn are plain operators).
template <typename F>
__device__ __forceinline__ F mul_then_add(F p, F s, F m) {
#pragma clang fp contract(off)
    const F q = p * s;
    return q + m;
}

__global__ __launch_bounds__(256) void smplx_post_k(PostArgs a) {
    const long total = (long)a.B * a.T * NOUT;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i % NOUT);
        const int t = (int)((i / NOUT) % a.T);
        const int b = (int)(i / ((long)NOUT * a.T));
        int src, grp;
        double* out;
        if (j < NPOSE) {
            out = a.poses + ((long)b * a.T + t) * NPOSE + j;
            if (j < 66) { src = j; grp = 0; }
            else if (j < 69) { src = 156 + (j - 66); grp = 0; }
            else if (j < 75) { src = -1; grp = 0; }                       // eye poses: not generated
            else { src = 66 + (j - 75); grp = 1; }
        } else if (j < NPOSE + NEXPR) {
            out = a.expr + ((long)b * a.T + t) * NEXPR + (j - NPOSE);
            src = 209 + (j - NPOSE); grp = 3;
        } else {
            out = a.trans + ((long)b * a.T + t) * NTRANS + (j - NPOSE - NEXPR);
            src = 309 + (j - NPOSE - NEXPR); grp = 2;
        }
        const int len = a.lengths ? a.lengths[b] : a.T;
        if (src < 0 || t >= len) { *out = 0.0; continue; }
        const float* col = a.pred + (long)b * a.T * a.C + src;      // (stitched mode: B == 1 -> b == 0)
        const double m = a.mean[src], s = a.stdv[src];
        const float mf = (float)m, sf = (float)s;
        auto denorm = [&](int tt) -> double {
            const float p = col[(long)(a.rows ? a.rows[tt] : tt) * a.C];
            if (a.stats_f32) return (double)mul_then_add<float>(p, sf, mf);
            return mul_then_add<double>((double)p, s, m);
        };
        const int r = a.radius[grp];
        if (r < 0) { *out = denorm(t); continue; }
        const double* w = a.taps + grp * MAXTAP;
        double acc = denorm(t) * w[r];
        for (int k = 1; k <= r; ++k) {
            const int lo = t - k < 0 ? 0 : t - k, hi = t + k > len - 1 ? len - 1 : t + k;
            acc += (denorm(lo) + denorm(hi)) * w[r + k];
        }
        *out = acc;
    }
}

}  // namespace

int mc_launch_smplx_post(const float* pred, const int* lengths, const int* rows, const double* mean, const double* stdv,
                         const double* taps, const int* radius, int stats_f32, int B, int T, int C,
                         double* poses, double* expr, double* trans, hipStream_t s) {
    MC_REQUIRE(C == 322, "smplx post-processing: input_feats=%d (the SMPL-X layout is 322-d)", C);
    MC_REQUIRE(!rows || (B == 1 && !lengths), "smplx post-processing: the stitched mode takes one sequence of `T` mapped frames");
    PostArgs a;
    a.pred = pred; a.lengths = lengths; a.rows = rows; a.mean = mean; a.stdv = stdv; a.taps = taps;
    for (int g = 0; g < 4; ++g) {
        MC_REQUIRE(radius[g] < (MAXTAP + 1) / 2, "smplx post-processing: filter radius %d too large", radius[g]);
        a.radius[g] = radius[g];
    }
    a.stats_f32 = stats_f32; a.B = B; a.T = T; a.C = C;
    a.poses = poses; a.expr = expr; a.trans = trans;
    const long total = (long)B * T * NOUT;
    if (total <= 0) return MC_OK;
    int blocks = cdiv(total, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(smplx_post_k, dim3(blocks), dim3(256), 0, s, a);
    MC_LAUNCH_CHECK();
    return MC_OK;
}

int mc_smplx_post_maxtap() { return MAXTAP; }
