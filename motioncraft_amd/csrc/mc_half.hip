// fp16-MFMA variants of the GEMM-shaped kernels (gfx950: v_mfma_f32_32x32x16_f16, fp32 accumulate) -- the reduced-precision
// mode of BASELINE.json configs[4]; modes and error model in mc_half.h.
//
// Fragment layout of v_mfma_f32_32x32x16_f16 (guide section 3): the A operand of lane l is 8 consecutive-k halves of
// row (l & 31), k = 8 (l >> 5) + 0..7; the B operand the same for column (l & 31); C/D as for every 32x32 MFMA:
// lane l holds D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31], r = 0..15.  As in the fp32 kernels the WEIGHT tile is the
// "A" operand and the activation rows the "B" operand, so a lane owns one output row and 4 consecutive columns per
// accumulator quad (float4 epilogues), and an accumulator fragment can be chained as the next B operand: its 16
// values are the k-slots (hf, i), i = 0..7, of two 16-wide k-blocks in the order  k = 16 blk + 4 hf + (i & 3) + 8 (i >> 2)
// -- a permutation inside each 32-chunk that the second weight matrix is stored in (mc_launch_split_f16_chainperm).
//
// LDS tiles are [rows][K + 8] halves: a row stride of 80 / 144 / 272 bytes (20 / 36 / 68 banks) makes the ds_read_b128
// fragment reads of any 16 distinct rows and the ds_write_b128 staging writes bank-conflict free.
#include "mc_common.h"
#include "mc_half.h"
#include "mc_bodyphase.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// x = hi + lo with hi = fp16(x), lo = fp16(x - hi).  The operand is first pinned as a materialised fp32 value: with
// -ffp-contract=fast hipcc otherwise folds the producer's last multiply into the conversion (v_fma_mix*_f16 = ONE
// rounding of the exact product) for `hi` while `x - hi` still sees the fp32-rounded product converted separately -- near an
// fp16 rounding tie the two disagree by one fp16 ulp and hi + lo is off by 2^-11 |x| (seen as 3e-4 errors in 0.16 % of rows).
__device__ __forceinline__ float pinned(float v) {
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x = pinned(a[i]), y = pinned(b[i]);
        const _Float16 hx = (_Float16)x, hy = (_Float16)y;
        hi[i] = hx;
        lo[i] = (_Float16)(x - pinned((float)hx));
        hi[4 + i] = hy;
        lo[4 + i] = (_Float16)(y - pinned((float)hy));
    }
}

template <bool SPLIT>
__device__ __forceinline__ f32x16 mma3(const f16x8& wh, const f16x8& wl, const f16x8& xh, const f16x8& xl, f32x16 acc) {
    if constexpr (SPLIT) {      // small terms first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, acc, 0, 0, 0);
    }
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc, 0, 0, 0);
}

__global__ __launch_bounds__(256) void split_f16_k(const float* __restrict__ x, mc_half* __restrict__ hi, mc_half* __restrict__ lo,
                                                   long n, int K, int perm) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        long src = i;
        if (perm) {     // position p of a 32-chunk <- source column 16 blk + 4 hf + (i & 3) + 8 (i >> 2)
            const long r = i / K;
            const int c = (int)(i % K), q = c & 31, blk = q >> 4, hf = (q >> 3) & 1, j = q & 7;
            src = r * K + (c - q) + 16 * blk + 4 * hf + (j & 3) + 8 * (j >> 2);
        }
        const float v = x[src];
        const _Float16 h = (_Float16)v;
        hi[i] = h;
        if (lo) lo[i] = (_Float16)(v - (float)h);
    }
}

// =================================================================================================
// C = act(A W^T + bias) + R,  128 x 128 x 32 tiles, 4 waves x (2 x 2) MFMA tiles, double-buffered LDS, one barrier per
// k-tile; A is fp32 in HBM and split while staging (registers -> LDS), W comes pre-split
// =================================================================================================
constexpr int HLD = 40;                 // halves per LDS row: 32 + 8 pad
constexpr int HTILE = 128 * HLD;        // halves per plane tile

template <bool SPLIT>
__global__ __launch_bounds__(256, 2) void gemm_h_k(GemmHArgs g) {
    constexpr int P = SPLIT ? 2 : 1;
    __shared__ __attribute__((aligned(16))) _Float16 smem[2 * 2 * P * HTILE];     // [buf][A | W][plane][128][HLD]: 80 / 40 KB
    auto tile = [&](int buf, int op, int p) { return smem + ((buf * 2 + op) * P + p) * HTILE; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, frow = lane & 31, hf = lane >> 5;
    const int ntn = g.N / 128;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = bid / ntn, tn = bid % ntn;
    const int row0 = tm * 128, nrows = min(128, g.M - row0);
    // staging: thread -> 8-column piece sq of rows srow and srow + 64 (A: 8 floats, W: 8 halves per plane)
    const int sq = tid & 3, srow = tid >> 2;
    const float* pa[2];
    const mc_half* pw[P][2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int r = min(row0 + srow + 64 * c, g.M - 1);          // rows past M re-read the last row (never stored)
        pa[c] = g.A + (long)r * g.lda + sq * 8;
        const long wo = (long)(tn * 128 + srow + 64 * c) * g.K + sq * 8;
        pw[0][c] = g.Wh + wo;
        if constexpr (SPLIT) pw[1][c] = g.Wl + wo;
    }
    f32x4 ra[2][2];
    u32x4 rw[P][2];
    auto load = [&](int kt) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            ra[c][0] = *reinterpret_cast<const f32x4*>(pa[c] + kt * 32);
            ra[c][1] = *reinterpret_cast<const f32x4*>(pa[c] + kt * 32 + 4);
#pragma unroll
            for (int p = 0; p < P; ++p) rw[p][c] = *reinterpret_cast<const u32x4*>(pw[p][c] + kt * 32);
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int o = (srow + 64 * c) * HLD + sq * 8;
            f16x8 hi, lo;
            split8(ra[c][0], ra[c][1], hi, lo);
            *reinterpret_cast<f16x8*>(tile(buf, 0, 0) + o) = hi;
            if constexpr (SPLIT) *reinterpret_cast<f16x8*>(tile(buf, 0, 1) + o) = lo;
#pragma unroll
            for (int p = 0; p < P; ++p) *reinterpret_cast<u32x4*>(tile(buf, 1, p) + o) = rw[p][c];
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    const int nk = g.K / 32;
    load(0);
    store(0);
    if (nk > 1) load(1);
    __syncthreads();
    const int fo_a = (wm * 64 + frow) * HLD + hf * 8, fo_w = (wn * 64 + frow) * HLD + hf * 8;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        auto kstep = [&](int s) {
            f16x8 fa[2][P], fw[2][P];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    fa[i][p] = *reinterpret_cast<const f16x8*>(tile(buf, 0, p) + fo_a + i * 32 * HLD + s * 16);
                    fw[i][p] = *reinterpret_cast<const f16x8*>(tile(buf, 1, p) + fo_w + i * 32 * HLD + s * 16);
                }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = mma3<SPLIT>(fw[ni][0], fw[ni][P - 1], fa[mi][0], fa[mi][P - 1], acc[mi][ni]);
        };
        kstep(0);
        // staging in the middle of the MFMA stream (as gemm_k): tile kt+1 (requested one iteration ago) -> other buffer,
        // then tile kt+2 is requested into the same registers
        if (kt + 1 < nk) store(buf ^ 1);
        if (kt + 2 < nk) load(kt + 2);
        kstep(1);
        __syncthreads();
    }
    // epilogue: lane owns row m and 4 consecutive columns per accumulator quad
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int m = wm * 64 + mi * 32 + frow;
        if (m >= nrows) continue;
        float* crow = g.C + (long)(row0 + m) * g.ldc;
        const float* rrow = g.R ? g.R + (long)(row0 + m) * g.ldr : nullptr;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = tn * 128 + wn * 64 + ni * 32 + 8 * q + 4 * hf;
                f32x4 v = {acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]};
                if (g.bias) v += *reinterpret_cast<const f32x4*>(g.bias + n);
                if (g.act != ACT_NONE) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = apply_act(v[j], g.act);
                }
                if (rrow) v += *reinterpret_cast<const f32x4*>(rrow + n);
                *reinterpret_cast<f32x4*>(crow + n) = v;
            }
    }
}

// =================================================================================================
// gemm_hd_k: the same product fed from fp16 PLANES of A (written once by the producer: film_rows_k in a reduced-precision
// context) instead of splitting the fp32 activations while staging -- gemm_h_k converts every A element once per COLUMN tile (12
// times at N = 1536) and that VALU work was ~30 % of its MFMA time.  Both operands arrive by LDS-DMA (global_load_lds_dwordx4: no
// staging registers, no ds_write): the LDS image of a plane tile is the unpadded [128 rows][BK halves], BK = 64 (f16: 128-byte
// rows, 16-byte chunk c of row r at position c ^ (r & 7), gemm_dma_k's layout) or 32 (split: 64-byte rows, c ^ ((r >> 2) & 3),
// gemm_wp_k's layout), double buffered: 2 x 32 KB in either mode, one barrier per k-tile (16 / 24 MFMAs per wave).
// =================================================================================================
// 16 bytes per lane from sbase + voff (bytes) straight into LDS at lds_byte + 16 lane (M0 form; asm: the builtin does not survive host-side
// instantiation inside a kernel TEMPLATE, and the compiler must not count / drain it as an ordinary load anyway -- mc_gemm.hip dma16)
__device__ __forceinline__ void dma16h(unsigned voff, const mc_half* sbase, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte) : "memory");
}

template <bool SPLIT, bool INIT = false, bool PRE = false>
__global__ __launch_bounds__(256, 2) void gemm_hd_k(GemmHArgs g) {
    constexpr int P = SPLIT ? 2 : 1;
    constexpr int BKH = SPLIT ? 32 : 64;            // halves per k-tile
    constexpr int CH = BKH / 8;                     // 16-byte chunks per row
    constexpr int RPI = 64 / CH;                    // rows per wave-wide DMA instruction
    constexpr int PT = 128 * BKH;                   // halves per plane tile
    __shared__ __attribute__((aligned(16))) _Float16 smem[2 * 2 * P * PT];      // [buf][A | W][plane][128][BKH] = 64 KB
    auto tile = [&](int buf, int op, int p) { return smem + ((buf * 2 + op) * P + p) * PT; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, frow = lane & 31, hf = lane >> 5;
    const int ntn = g.N / 128;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = bid / ntn, tn = bid % ntn;
    const int row0 = tm * 128, nrows = min(128, g.M - row0);
    // DMA: wave w moves rows [32 w, 32 w + 32) of every plane tile, RPI rows per instruction: lane -> row q RPI + lane / CH,
    // LDS position lane % CH, global chunk (lane % CH) ^ swz(row)
    const int dr = lane / CH, dpos = lane % CH;
    auto swz = [&](int r) { return SPLIT ? ((r >> 2) & 3) : (r & 7); };
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    constexpr int NQ = 32 / RPI;                    // instructions per plane tile per wave (2 / 4)
    unsigned goa[NQ], gow[NQ];                      // byte offsets into the planes (< 2^32: checked by the launcher)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int r = 32 * wave + q * RPI + dr;     // row inside the tile
        const int ar = min(row0 + r, g.M - 1);      // rows past M re-read the last row (never stored)
        goa[q] = (unsigned)(((long)ar * g.K + (dpos ^ swz(r)) * 8) * 2);
        gow[q] = (unsigned)(((long)(tn * 128 + r) * g.K + (dpos ^ swz(r)) * 8) * 2);
    }
    const unsigned lds0 = (unsigned)(size_t)smem;
    auto issue = [&](int kt, int buf) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const mc_half* ap = (p ? g.Al : g.Ah) + kt * BKH;
            const mc_half* wp = (p ? g.Wl : g.Wh) + kt * BKH;
            const unsigned la = lds0 + (unsigned)((((buf * 2 + 0) * P + p) * PT + 32 * wave_u * BKH) * 2);
            const unsigned lw = lds0 + (unsigned)((((buf * 2 + 1) * P + p) * PT + 32 * wave_u * BKH) * 2);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                dma16h(goa[q], ap, la + q * RPI * BKH * 2);
                dma16h(gow[q], wp, lw + q * RPI * BKH * 2);
            }
        }
    };
    // PRE (round 6): the residual rows this lane adds in the epilogue are requested HERE, into 64 registers of their own, and fly during the DMA prologue and the
    // first k-tile (whose vmcnt(0) covers them): the epilogue adds them in the SAME order as before -- (sum + bias) + R, the same bits -- without a
    // load -> add -> store chain at the end of every tile.  (sched_barrier: the scheduler must not sink the loads down to their uses)
    f32x4 rpre[PRE ? 2 : 1][2][4];
    if constexpr (PRE) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int m = min(wm * 64 + mi * 32 + frow, nrows - 1);
            const float* rrow = g.R + (long)(row0 + m) * g.ldr;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int q = 0; q < 4; ++q) rpre[mi][ni][q] = *reinterpret_cast<const f32x4*>(rrow + tn * 128 + wn * 64 + ni * 32 + 8 * q + 4 * hf);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    f32x16 acc[2][2];
    if constexpr (INIT) {
        // (round 6, tools/gemm_h6_lab.hip) the kernel runs at the package power cap, where the cost of its parts adds up; what the lab found movable is
        // the load -> add -> store chain at the end of every tile: with R + bias as the accumulators' start value the loads fly during the DMA prologue
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int m = min(wm * 64 + mi * 32 + frow, nrows - 1);
            const float* rrow = g.R + (long)(row0 + m) * g.ldr;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = tn * 128 + wn * 64 + ni * 32 + 8 * q + 4 * hf;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(rrow + n) + *reinterpret_cast<const f32x4*>(g.bias + n);
                    acc[mi][ni][4 * q] = v[0]; acc[mi][ni][4 * q + 1] = v[1]; acc[mi][ni][4 * q + 2] = v[2]; acc[mi][ni][4 * q + 3] = v[3];
                }
        }
    } else {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    }
    const int nk = g.K / BKH;
    const int sw = swz(frow);                        // rows frow, frow + 32, + 64, + 96 share the swizzle term
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) issue(kt + 1, buf ^ 1);     // in flight during the MFMAs below; drained in front of the barrier
        const int ra = (wm * 64 + frow) * BKH, rw = (wn * 64 + frow) * BKH;
#pragma unroll
        for (int s = 0; s < BKH / 16; ++s) {
            const int pos = ((2 * s + hf) ^ sw) * 8;
            f16x8 fa[2][P], fw[2][P];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    fa[i][p] = *reinterpret_cast<const f16x8*>(tile(buf, 0, p) + ra + i * 32 * BKH + pos);
                    fw[i][p] = *reinterpret_cast<const f16x8*>(tile(buf, 1, p) + rw + i * 32 * BKH + pos);
                }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = mma3<SPLIT>(fw[ni][0], fw[ni][P - 1], fa[mi][0], fa[mi][P - 1], acc[mi][ni]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's part of tile kt + 1 has landed
        __syncthreads();
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int m = wm * 64 + mi * 32 + frow;
        if (m >= nrows) continue;
        float* crow = g.C + (long)(row0 + m) * g.ldc;
        const float* rrow = g.R ? g.R + (long)(row0 + m) * g.ldr : nullptr;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = tn * 128 + wn * 64 + ni * 32 + 8 * q + 4 * hf;
                f32x4 v = {acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]};
                if constexpr (INIT) {
                    *reinterpret_cast<f32x4*>(crow + n) = v;
                    continue;
                }
                if (g.bias) v += *reinterpret_cast<const f32x4*>(g.bias + n);
                if (g.act != ACT_NONE) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = apply_act(v[j], g.act);
                }
                if constexpr (PRE) v += rpre[mi][ni][q];
                else if (rrow) v += *reinterpret_cast<const f32x4*>(rrow + n);
                *reinterpret_cast<f32x4*>(crow + n) = v;
            }
    }
}

// =================================================================================================
// gemm_hf_k (round 6, tools/gemm_h6_lab.hip ha_k): gemm_hd_k whose A operand NEVER touches LDS.  film_rows_k writes the activation planes FRAGMENT-MAJOR --
// per 32-row block and 16-wide k-step the 64 lanes' 16-byte MFMA operands contiguous (1 KB) -- so a wave fetches the fragments of ITS 32 rows with one
// coalesced global_load_dwordx4 per k-step and plane, straight into VGPRs: no LDS-DMA piece, no ds_read, no barrier dependency for A.  4 waves, each 32 rows x
// all 128 columns of the tile (4 accumulator blocks); only W rides the LDS-DMA ring (3 slots of [plane][128 rows][BKH halves]: 48 KB), ONE raw barrier per
// slab, A fragments and W pieces requested two slabs ahead and retired together by counted vmcnt (loads return in order; everything inline asm, so the compiler
// neither counts nor drains them).  Lab, plain f16, same passes: 193 us against 218 at 25088 rows, 82 vs 92 at 12544, 48.5 vs 57 at 6272; bit-equal to
// gemm_hd_k (the same k order per output, the same three products per operand pair in the split mode).
// The residual rows are prefetched at the top of the tile as in gemm_hd_k<., ., true>.
// =================================================================================================
__device__ __forceinline__ void gload16h(f16x8& dst, const mc_half* base, unsigned voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(dst) : "v"(voff), "s"(base) : "memory");
}

template <bool SPLIT>
__global__ __launch_bounds__(256, 2) void gemm_hf_k(GemmHArgs g) {
    constexpr int P = SPLIT ? 2 : 1;
    constexpr int BKH = SPLIT ? 32 : 64, KS = BKH / 16;      // halves per slab, k-steps per slab
    constexpr int CH = BKH / 8, RPI = 64 / CH;               // 16-byte chunks per W row, W rows per wave-wide DMA instruction
    constexpr int PT = 128 * BKH;                            // halves per W plane tile
    constexpr int ST = 3;
    constexpr int NQ = 32 / RPI;                             // DMA instructions per plane tile per wave (4 / 2)
    constexpr int NVM = P * NQ + P * KS;                     // vector-memory operations of one slab per wave: W pieces + A fragments (= 8 in both modes)
    static_assert(NVM == 8, "gemm_hf_k: the counted wait below is written for 8 operations per slab");
    __shared__ __attribute__((aligned(16))) _Float16 smem[ST * P * PT];     // [slot][plane][128][BKH] = 48 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int frow = lane & 31, hf = lane >> 5;
    const int ntn = g.N / 128;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = bid / ntn, tn = bid % ntn;
    const int row0 = tm * 128, nrows = min(128, g.M - row0);
    auto swz = [&](int r) { return SPLIT ? ((r >> 2) & 3) : ((r >> 1) & 7); };
    const int dr = lane / CH, dpos = lane % CH;
    unsigned gow[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int r = 32 * wave + q * RPI + dr;              // W row inside the tile
        gow[q] = (unsigned)(((long)(tn * 128 + r) * g.K + (dpos ^ swz(r)) * 8) * 2);
    }
    const unsigned lds0 = (unsigned)(size_t)smem;
    auto issue_w = [&](int s) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const mc_half* wp = (p ? g.Wl : g.Wh) + s * BKH;
            const unsigned lw = lds0 + (unsigned)((((s % ST) * P + p) * PT + 32 * wave_u * BKH) * 2);
#pragma unroll
            for (int q = 0; q < NQ; ++q) dma16h(gow[q], wp, lw + q * RPI * BKH * 2);
        }
    };
    // this wave's row block of the fragment-major planes (rows past M: the last block -- valid memory, never stored)
    const int nrb = g.M >> 5;
    const int rb = min((row0 >> 5) + wave_u, nrb - 1);
    const unsigned abase = (unsigned)(((long)rb * (g.K >> 4) * 64 + lane) * 16);
    f16x8 fa[3][P][KS];
    auto issue_a = [&](int s, int set) {
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int k = 0; k < KS; ++k) gload16h(fa[set][p][k], p ? g.Al : g.Ah, abase + (unsigned)((s * KS + k) * 1024));
    };
    // residual rows of the epilogue, requested first (they are the oldest loads: every counted wait below covers them)
    const int m = wave * 32 + frow;
    f32x4 rpre[4][4];
    const bool has_r = g.R != nullptr;
    if (has_r) {
        const float* rrow = g.R + (long)(row0 + min(m, nrows - 1)) * g.ldr + tn * 128 + 4 * hf;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int q = 0; q < 4; ++q) rpre[ni][q] = *reinterpret_cast<const f32x4*>(rrow + ni * 32 + 8 * q);
        __builtin_amdgcn_sched_barrier(0);
    }
    f32x16 acc[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;
    const int ns = g.K / BKH;
    issue_w(0); issue_a(0, 0);
    if (ns > 1) { issue_w(1); issue_a(1, 1); }
    const int sw = swz(frow);                                 // rows frow, frow + 32, + 64, + 96 share the swizzle term
    auto body = [&](int s, int set, int nset) {
        // outstanding, oldest first: W(s) A(s) W(s + 1) A(s + 1): slab s has landed when at most the 8 operations of slab s + 1 remain
        if (s + 1 < ns) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                         // every wave's pieces of slab s are in LDS; every wave has read slab s - 1
        if (s + 2 < ns) { issue_w(s + 2); issue_a(s + 2, nset); }
        const _Float16* sl = smem + (s % ST) * P * PT;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const int pos = ((2 * k + hf) ^ sw) * 8;
            f16x8 fw[4][P];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int p = 0; p < P; ++p) fw[ni][p] = *reinterpret_cast<const f16x8*>(sl + p * PT + (ni * 32 + frow) * BKH + pos);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[ni] = mma3<SPLIT>(fw[ni][0], fw[ni][P - 1], fa[set][0][k], fa[set][P - 1][k], acc[ni]);
        }
    };
    // (the register ring is indexed statically: three slabs per trip)
    int s = 0;
    for (; s + 2 < ns; s += 3) {
        body(s, 0, 2);
        body(s + 1, 1, 0);
        body(s + 2, 2, 1);
    }
    if (s < ns) body(s, 0, 2);
    if (s + 1 < ns) body(s + 1, 1, 0);
    if (m >= nrows) return;
    float* crow = g.C + (long)(row0 + m) * g.ldc;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = tn * 128 + ni * 32 + 8 * q + 4 * hf;
            f32x4 v = {acc[ni][4 * q], acc[ni][4 * q + 1], acc[ni][4 * q + 2], acc[ni][4 * q + 3]};
            if (g.bias) v += *reinterpret_cast<const f32x4*>(g.bias + n);
            if (g.act != ACT_NONE) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = apply_act(v[j], g.act);
            }
            if (has_r) v += rpre[ni][q];
            *reinterpret_cast<f32x4*>(crow + n) = v;
        }
}

// =================================================================================================
// Fused 2-layer MLP on the fp16 MFMA (structure of mlp2_k, mc_chain.hip): X fragment in VGPRs, hidden in 32-wide chunks
// whose FC1 accumulator -- bias + exact GELU in fp32, then split -- is directly the B operand of FC2
// =================================================================================================
template <int L, int MODE, bool SPLIT>
__global__ __launch_bounds__(256, 2) void mlp2_h_k(MlpArgs g, const mc_half* __restrict__ W1h, const mc_half* __restrict__ W1l,
                                                   const mc_half* __restrict__ W2h, const mc_half* __restrict__ W2l) {
    constexpr int P = SPLIT ? 2 : 1, NKB = L / 16, NT = L / 32, HC = 32;
    constexpr int LD1 = L + 8, LD2 = HC + 8;            // halves per LDS row of the W1 chunk [32][L] / W2^T chunk [L][32]
    constexpr int S1 = HC * LD1, S2 = L * LD2, BUF = P * (S1 + S2);
    constexpr int MAXHID = 1024;
    constexpr int PC1 = HC * L / 8, PC2 = L * HC / 8;    // 16-byte pieces per plane chunk
    constexpr int N1 = (PC1 + 255) / 256, N2 = (PC2 + 255) / 256;
    __shared__ __attribute__((aligned(16))) _Float16 smem[2 * BUF + 2 * MAXHID];
    float* s_b1 = reinterpret_cast<float*>(smem + 2 * BUF);
    auto W1s = [&](int b, int p) { return smem + b * BUF + p * S1; };
    auto W2s = [&](int b, int p) { return smem + b * BUF + P * S1 + p * S2; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5;
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
    const mc_half* w1p[P];
    const mc_half* w2p[P];
    w1p[0] = W1h + (long)grp * g.hidden * L;
    w2p[0] = W2h + (long)grp * L * g.hidden;
    if constexpr (SPLIT) {
        w1p[1] = W1l + (long)grp * g.hidden * L;
        w2p[1] = W2l + (long)grp * L * g.hidden;
    }
    const float* __restrict__ b1 = g.b1 + (long)grp * g.hidden;
    const float* __restrict__ b2 = g.b2 + (long)grp * L;
    for (int i = tid; i < g.hidden; i += 256) s_b1[i] = b1[i];

    u32x4 r1[P][N1], r2[P][N2];
    auto fetch = [&](int hc) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
#pragma unroll
            for (int i = 0; i < N1; ++i) {
                const int idx = tid + 256 * i;
                if (PC1 % 256 == 0 || idx < PC1)
                    r1[p][i] = *reinterpret_cast<const u32x4*>(w1p[p] + (long)(hc * HC + idx / (L / 8)) * L + (idx % (L / 8)) * 8);
            }
#pragma unroll
            for (int i = 0; i < N2; ++i) {
                const int idx = tid + 256 * i;
                if (PC2 % 256 == 0 || idx < PC2)
                    r2[p][i] = *reinterpret_cast<const u32x4*>(w2p[p] + (long)(idx >> 2) * g.hidden + hc * HC + (idx & 3) * 8);
            }
        }
    };
    auto commit = [&](int b) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
#pragma unroll
            for (int i = 0; i < N1; ++i) {
                const int idx = tid + 256 * i;
                if (PC1 % 256 == 0 || idx < PC1)
                    *reinterpret_cast<u32x4*>(W1s(b, p) + (idx / (L / 8)) * LD1 + (idx % (L / 8)) * 8) = r1[p][i];
            }
#pragma unroll
            for (int i = 0; i < N2; ++i) {
                const int idx = tid + 256 * i;
                if (PC2 % 256 == 0 || idx < PC2) *reinterpret_cast<u32x4*>(W2s(b, p) + (idx >> 2) * LD2 + (idx & 3) * 8) = r2[p][i];
            }
        }
    };

    const int r = wave * 32 + (lane & 31);
    const bool rok = r < nrows;
    f16x8 xh[NKB], xl[NKB];
    {
        long srow = rok ? row0 + r : row0;
        if constexpr (MODE == MLP_EXPERT) srow = rok ? g.src_row[row0 + r] : 0;
        const float* xp = g.X + (long)grp * g.x_gstride + srow * g.ldx + hf * 8;
        // (rows past nrows read a valid row and are never stored)
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(xp + 16 * kb), b = *reinterpret_cast<const f32x4*>(xp + 16 * kb + 4);
            split8(a, b, xh[kb], xl[kb]);
        }
    }
    f32x16 acc2[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc2[t][q] = 0.f;

    const int nch = g.hidden / HC;
    fetch(0);
    commit(0);
    if (1 < nch) fetch(1);
    __syncthreads();
    const int fo1 = (lane & 31) * LD1 + hf * 8, fo2 = (lane & 31) * LD2 + hf * 8;
    for (int hc = 0; hc < nch; ++hc) {
        const int buf = hc & 1;
        if (hc + 1 < nch) commit(buf ^ 1);
        if (hc + 2 < nch) fetch(hc + 2);
        // FC1 chunk: 32 hidden units of this wave's 32 rows
        f32x16 a1;
#pragma unroll
        for (int q = 0; q < 16; ++q) a1[q] = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const f16x8 wh = *reinterpret_cast<const f16x8*>(W1s(buf, 0) + fo1 + 16 * kb);
            const f16x8 wl = SPLIT ? *reinterpret_cast<const f16x8*>(W1s(buf, P - 1) + fo1 + 16 * kb) : wh;
            a1 = mma3<SPLIT>(wh, wl, xh[kb], xl[kb], a1);
        }
        // bias + exact GELU in fp32; accumulator registers 0..7 / 8..15 are the two k-blocks of the FC2 B operand
        f16x8 hh[2], hl[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            f32x4 v[2];
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int q = 2 * blk + qq;
                const f32x4 bb = *reinterpret_cast<const f32x4*>(s_b1 + hc * HC + 8 * q + 4 * hf);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[qq][i] = gelu_exact(a1[4 * q + i] + bb[i]);
            }
            split8(v[0], v[1], hh[blk], hl[blk]);
        }
        // FC2 partial: out[L] += W2t[:, chunk] h
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f16x8 wh = *reinterpret_cast<const f16x8*>(W2s(buf, 0) + fo2 + t * 32 * LD2 + 16 * blk);
                const f16x8 wl = SPLIT ? *reinterpret_cast<const f16x8*>(W2s(buf, P - 1) + fo2 + t * 32 * LD2 + 16 * blk) : wh;
                acc2[t] = mma3<SPLIT>(wh, wl, hh[blk], hl[blk], acc2[t]);
            }
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
            const int n = t * 32 + 8 * q + 4 * hf;
            const f32x4 bb = *reinterpret_cast<const f32x4*>(b2 + n);
            const f32x4 v = {acc2[t][4 * q] + bb[0], acc2[t][4 * q + 1] + bb[1], acc2[t][4 * q + 2] + bb[2], acc2[t][4 * q + 3] + bb[3]};
            *reinterpret_cast<f32x4*>(yrow + n) = v;
        }
}

// per-row LayerNorm of a fragment-distributed row (lane l and lane l ^ 32 hold the two halves; column of x[j][i] = 8 j + 4 hf + i)
// =================================================================================================
// mlp2hd_k: mlp2_h_k<L, MODE, SPLIT> (L = 128 or 64) with the weight chunks of both layers staged by LDS-DMA (the fp16 twin of mlp2d_k, mc_chain.hip):
// no staging registers, no ds_write, no padding.  LDS images per plane:
//   W1 chunk [32 hidden rows][L halves]: 16-byte chunk c of row r at position c ^ (r & 15) (L = 128: 256-byte rows) or c ^ ((r >> 1) & 7) (L = 64)
//   W2 chunk [L out rows][32 halves]     (64-byte rows):  chunk c of row r at position c ^ ((r >> 2) & 3)        (gemm_wp_k's layout)
// both conflict-free for the b128 fragment reads (mlp2_h_k's padded rows were not: 13 % of its LDS cycles were bank conflicts).
// Same MFMA order and operands: the same bits.
// =================================================================================================
template <int L, int MODE, bool SPLIT>
__global__ __launch_bounds__(256, 2) void mlp2hd_k(MlpArgs g, const mc_half* __restrict__ W1h, const mc_half* __restrict__ W1l,
                                                   const mc_half* __restrict__ W2h, const mc_half* __restrict__ W2l) {
    static_assert(L == 128 || L == 64, "mlp2hd_k: L");
    constexpr int P = SPLIT ? 2 : 1, NKB = L / 16, NT = L / 32, HC = 32;
    constexpr int NP = L / 64;             // 1 KB DMA pieces per plane, matrix and wave
    constexpr int CPR = L / 8;             // 16-byte chunks per W1 row: 16 (256-byte rows, swizzle r & 15) or 8 (128-byte rows, (r >> 1) & 7)
    constexpr int C1 = HC * L, C2 = L * HC, BUF = P * (C1 + C2), MAXHID = 1024;       // halves
    __shared__ __attribute__((aligned(16))) _Float16 smem[2 * BUF + 2 * MAXHID];
    float* s_b1 = reinterpret_cast<float*>(smem + 2 * BUF);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
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
    const int grp_u = __builtin_amdgcn_readfirstlane(grp);
    const mc_half* w1p[P];
    const mc_half* w2p[P];
    w1p[0] = W1h + (long)grp_u * g.hidden * L;
    w2p[0] = W2h + (long)grp_u * L * g.hidden;
    if constexpr (SPLIT) {
        w1p[1] = W1l + (long)grp_u * g.hidden * L;
        w2p[1] = W2l + (long)grp_u * L * g.hidden;
    }
    const float* __restrict__ b1 = g.b1 + (long)grp * g.hidden;
    const float* __restrict__ b2 = g.b2 + (long)grp * L;
    for (int i = tid; i < g.hidden; i += 256) s_b1[i] = b1[i];
    // DMA byte offsets of this lane (NP instructions per plane, matrix and wave)
    //   W1: instruction q = 64 / CPR rows: row (64 / CPR) (NP wave + q) + lane / CPR, position p = lane % CPR <- logical chunk p ^ swizzle(row)
    //   W2: instruction q = 16 rows: row 16 (NP wave + q) + (lane >> 2), position lane & 3 <- logical chunk p ^ ((row >> 2) & 3)
    unsigned vo1[NP], vo2[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int r1 = (64 / CPR) * (NP * wave + q) + lane / CPR, p1 = lane % CPR;
        vo1[q] = (unsigned)((r1 * L + (p1 ^ (L == 128 ? (r1 & 15) : ((r1 >> 1) & 7))) * 8) * 2);
        const int r2 = 16 * (NP * wave + q) + (lane >> 2), p2 = lane & 3;
        vo2[q] = (unsigned)(((long)r2 * g.hidden + (p2 ^ ((r2 >> 2) & 3)) * 8) * 2);
    }
    const unsigned lds0 = (unsigned)(size_t)smem;
    auto issue = [&](int hc) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const unsigned l1 = lds0 + (unsigned)(((hc & 1) * BUF + p * C1) * 2) + (unsigned)(NP * wave_u) * 1024;
            const unsigned l2 = lds0 + (unsigned)(((hc & 1) * BUF + P * C1 + p * C2) * 2) + (unsigned)(NP * wave_u) * 1024;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                dma16h(vo1[q], w1p[p] + (long)hc * HC * L, l1 + q * 1024);
                dma16h(vo2[q], w2p[p] + hc * HC, l2 + q * 1024);
            }
        }
    };
    const int r = wave * 32 + (lane & 31);
    const bool rok = r < nrows;
    f16x8 xh[NKB], xl[NKB];
    {
        long srow = rok ? row0 + r : row0;
        if constexpr (MODE == MLP_EXPERT) srow = rok ? g.src_row[row0 + r] : 0;
        const float* xp = g.X + (long)grp * g.x_gstride + srow * g.ldx + hf * 8;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(xp + 16 * kb), b = *reinterpret_cast<const f32x4*>(xp + 16 * kb + 4);
            split8(a, b, xh[kb], xl[kb]);
        }
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
    const int fr = lane & 31, s1x = L == 128 ? (fr & 15) : ((fr >> 1) & 7), s2x = (fr >> 2) & 3;
    for (int hc = 0; hc < nch; ++hc) {
        if (hc + 1 < nch) issue(hc + 1);
        const _Float16* B0 = smem + (hc & 1) * BUF;
        f32x16 a1;
#pragma unroll
        for (int q = 0; q < 16; ++q) a1[q] = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const int o = fr * L + ((2 * kb + hf) ^ s1x) * 8;
            const f16x8 wh = *reinterpret_cast<const f16x8*>(B0 + o);
            const f16x8 wl = SPLIT ? *reinterpret_cast<const f16x8*>(B0 + (P - 1) * C1 + o) : wh;
            a1 = mma3<SPLIT>(wh, wl, xh[kb], xl[kb], a1);
        }
        f16x8 hh[2], hl[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            f32x4 v[2];
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int q = 2 * blk + qq;
                const f32x4 bb = *reinterpret_cast<const f32x4*>(s_b1 + hc * HC + 8 * q + 4 * hf);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[qq][i] = gelu_exact(a1[4 * q + i] + bb[i]);
            }
            split8(v[0], v[1], hh[blk], hl[blk]);
        }
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int o = P * C1 + (t * 32 + fr) * HC + ((2 * blk + hf) ^ s2x) * 8;
                const f16x8 wh = *reinterpret_cast<const f16x8*>(B0 + o);
                const f16x8 wl = SPLIT ? *reinterpret_cast<const f16x8*>(B0 + (P - 1) * C2 + o) : wh;
                acc2[t] = mma3<SPLIT>(wh, wl, hh[blk], hl[blk], acc2[t]);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
            const int n = t * 32 + 8 * q + 4 * hf;
            const f32x4 bb = *reinterpret_cast<const f32x4*>(b2 + n);
            const f32x4 v = {acc2[t][4 * q] + bb[0], acc2[t][4 * q + 1] + bb[1], acc2[t][4 * q + 2] + bb[2], acc2[t][4 * q + 3] + bb[3]};
            *reinterpret_cast<f32x4*>(yrow + n) = v;
        }
}

template <int NJ>
__device__ __forceinline__ void frag_layernorm_h(f32x4 (&x)[NJ], const float* __restrict__ gamma, const float* __restrict__ beta, int kq) {
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
// projqkv_k (mc_chain.hip) on the fp16 MFMA: a = GELU(w0 y0 + w1 y1);  mf = a Wproj^T + b;  q|k|v = LN(mf[:, :L]) Wqkv^T + b2.
// The combine / GELU / LayerNorm arithmetic is fp32; only the two weight streams and the row fragments are fp16 (split).
// The kept body_value accumulators (output chunks c < L/32) become the B operand of the second GEMM in chain order, so
// Wqkv is stored chain-permuted along K (mc_launch_split_f16_chainperm).
// =================================================================================================
template <int L, bool SPLIT>
__global__ __launch_bounds__(256, 2) void projqkv_h_k(RowChainArgs g, const mc_half* __restrict__ Wph, const mc_half* __restrict__ Wpl,
                                                      const mc_half* __restrict__ Wqh, const mc_half* __restrict__ Wql) {
    constexpr int P = SPLIT ? 2 : 1, NKB = L / 16, NJ = L / 8, NC0 = 4 * L / 32, NC1 = 3 * L / 32, NKEEP = L / 32;
    constexpr int LD1 = L + 8, S1 = 32 * LD1;             // weight chunk [32 out rows][L] per plane
    constexpr int PC = 32 * L / 8, NP = (PC + 255) / 256;  // 16-byte pieces per plane chunk
    __shared__ __attribute__((aligned(16))) _Float16 smem[2 * P * S1 + 2 * 7 * L];
    float* s_bias = reinterpret_cast<float*>(smem + 2 * P * S1);      // proj bias [4L] | qkv bias [3L]
    auto Ws = [&](int b, int p) { return smem + (b * P + p) * S1; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5, kq = hf * 4;
    for (int i = tid; i < 4 * L; i += 256) s_bias[i] = g.bias[i];
    for (int i = tid; i < 3 * L; i += 256) s_bias[4 * L + i] = g.bias2[i];
    const long tok = g.tok0 + (long)blockIdx.x * 128 + wave * 32 + (lane & 31);
    const bool aliasing = g.alias.split_flag && *g.alias.split_flag == 0;
    if (aliasing && g.tok0 + (long)blockIdx.x * 128 >= g.alias.from) return;
    const bool rok = tok < g.N && !(aliasing && tok >= g.alias.from);
    u32x4 rw[P][NP];
    auto fetch = [&](int cg) {           // chunk cg: 0 .. NC0-1 = projection rows, then the q/k/v rows
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const mc_half* src = cg < NC0 ? (p ? Wpl : Wph) + (long)cg * 32 * L : (p ? Wql : Wqh) + (long)(cg - NC0) * 32 * L;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int idx = tid + 256 * i;
                if (PC % 256 == 0 || idx < PC) rw[p][i] = *reinterpret_cast<const u32x4*>(src + (long)(idx / (L / 8)) * L + (idx % (L / 8)) * 8);
            }
        }
    };
    auto commit = [&](int b) {
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int idx = tid + 256 * i;
                if (PC % 256 == 0 || idx < PC) *reinterpret_cast<u32x4*>(Ws(b, p) + (idx / (L / 8)) * LD1 + (idx % (L / 8)) * 8) = rw[p][i];
            }
    };
    fetch(0);
    f16x8 xh[NKB], xl[NKB];
    {
        const long tk = tok < g.N ? tok : 0;
        const float w0 = rok ? g.comb_w[2 * tk] : 0.f, w1 = rok ? g.comb_w[2 * tk + 1] : 0.f;
        const long ty = (g.twin_from > 0 && tk >= g.twin_from) ? tk - g.twin_from : tk;
        const float* y0 = g.X + 2 * ty * L + hf * 8;
        const bool k0 = w0 != 0.f, k1 = w1 != 0.f;
        // rows of dropped choices were never written: discarded by a select, not multiplied by 0 (see rowchain_k)
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            f32x4 v[2];
#pragma unroll
            for (int hq = 0; hq < 2; ++hq) {
                const f32x4 ya = *reinterpret_cast<const f32x4*>(y0 + 16 * kb + 4 * hq);
                const f32x4 yb = *reinterpret_cast<const f32x4*>(y0 + L + 16 * kb + 4 * hq);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[hq][i] = gelu_exact((k0 ? w0 * ya[i] : 0.f) + (k1 ? w1 * yb[i] : 0.f));
            }
            split8(v[0], v[1], xh[kb], xl[kb]);
        }
    }
    commit(0);
    fetch(1);
    __syncthreads();
    const int fo = (lane & 31) * LD1 + hf * 8;
    float* orow = g.Y + tok * g.ldy + kq;
    float* qrow = g.Y2 + tok * g.ldy2 + kq;
    f32x4 bvf[NJ];                    // body_value row fragment (fp32) for the shared LayerNorm
    f16x8 bh[NKB], bl[NKB];
    auto chunk = [&](int cg, const f16x8 (&fh)[NKB], const f16x8 (&fl)[NKB]) {
        f32x16 a;
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const f16x8 wh = *reinterpret_cast<const f16x8*>(Ws(cg & 1, 0) + fo + 16 * kb);
            const f16x8 wl = SPLIT ? *reinterpret_cast<const f16x8*>(Ws(cg & 1, P - 1) + fo + 16 * kb) : wh;
            a = mma3<SPLIT>(wh, wl, fh[kb], fl[kb], a);
        }
        return a;
    };
    // one chunk: MFMAs -> commit(next) -> bias + stores -> fetch(next + 1)   (order: see rowchain_k)
#define MC_PQH_CHUNK(cg, FH, FL, OUT, KEEP)                                                                 \
    {                                                                                                       \
        const f32x16 a = chunk((cg), FH, FL);                                                               \
        if ((cg) + 1 < NC0 + NC1) commit(((cg) & 1) ^ 1);                                                   \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                     \
            const f32x4 bb = *reinterpret_cast<const f32x4*>(s_bias + (cg) * 32 + 8 * q + kq);              \
            const f32x4 v = {a[4 * q] + bb[0], a[4 * q + 1] + bb[1], a[4 * q + 2] + bb[2], a[4 * q + 3] + bb[3]}; \
            if (rok) *reinterpret_cast<f32x4*>((OUT) + 8 * q) = v;                                          \
            KEEP                                                                                            \
        }                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        if ((cg) + 2 < NC0 + NC1) fetch((cg) + 2);                                                          \
        __syncthreads();                                                                                    \
    }
    // (the qkv bias sits behind the 4L proj biases in s_bias, so chunk cg of either stream reads s_bias + 32 cg)
#pragma unroll
    for (int c = 0; c < NKEEP; ++c) MC_PQH_CHUNK(c, xh, xl, orow + c * 32, bvf[4 * c + q] = v;)
    for (int c = NKEEP; c < NC0; ++c) MC_PQH_CHUNK(c, xh, xl, orow + c * 32, )
    frag_layernorm_h<NJ>(bvf, g.gamma, g.beta, kq);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) split8(bvf[2 * kb], bvf[2 * kb + 1], bh[kb], bl[kb]);
    for (int c = 0; c < NC1; ++c) MC_PQH_CHUNK(NC0 + c, bh, bl, qrow + c * 32, )
#undef MC_PQH_CHUNK
}

// =================================================================================================
// pqbody_h_k: projqkv_h_k + the body-topology attention (mc_bodyphase.h) over frame-aligned tiles, as pqbody_k (mc_chain.hip) does
// for the fp32 path: the q/k/v chunks are walked per 32-channel group in the order q, k, v and handed to the body phase through two
// LDS slots; q/k/v never reach HBM, ys is written from here.  Body arithmetic fp32 (identical to body_reg_k's).
// =================================================================================================
template <int L, int H, bool SPLIT>
__global__ __launch_bounds__(256, 2) void pqbody_h_k(RowChainArgs g_in, const mc_half* __restrict__ Wph, const mc_half* __restrict__ Wpl,
                                                     const mc_half* __restrict__ Wqh, const mc_half* __restrict__ Wql) {
    using BP = BodyPhase<L, H>;
    RowChainArgs g = g_in;                       // (second token range: see pqbody_k)
    long bidx = blockIdx.x;
    if (g.nblk1 > 0 && bidx >= g.nblk1) { bidx -= g.nblk1; g.tok0 = g.tok2; g.N = g.N2; }
    if (g.alias.split_flag && *g.alias.split_flag == 0 && g.tok0 + bidx * (long)BP::TR >= g.alias.from) return;
    constexpr int TR = BP::TR, XS = BP::XS;
    constexpr int P = SPLIT ? 2 : 1, NKB = L / 16, NJ = L / 8, NC0 = 4 * L / 32, NG = L / 32, NKEEP = L / 32, NSEQ = NC0 + 3 * NG;
    constexpr int LD1 = L + 8, S1 = 32 * LD1;             // weight chunk [32 out rows][L] per plane
    constexpr int PC = 32 * L / 8, NP = (PC + 255) / 256;  // 16-byte pieces per plane chunk
    __shared__ __attribute__((aligned(16))) _Float16 smem[2 * P * S1 + 2 * 7 * L + 2 * BP::LDS_FLOATS];
    float* s_bias = reinterpret_cast<float*>(smem + 2 * P * S1);      // proj bias [4L] | qkv bias [3L]
    float* s_x = s_bias + 7 * L;
    float* s_w = s_x + 2 * BP::SROWS * XS;
    auto Ws = [&](int b, int p) { return smem + (b * P + p) * S1; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5, kq = hf * 4;
    for (int i = tid; i < 4 * L; i += 256) s_bias[i] = g.bias[i];
    for (int i = tid; i < 3 * L; i += 256) s_bias[4 * L + i] = g.bias2[i];
    for (int i = tid; i < H * H; i += 256) s_w[i] = g.wsm[i];
    const long tile_tok0 = g.tok0 + bidx * TR;
    const bool aliasing = g.alias.split_flag && *g.alias.split_flag == 0;
    if (aliasing && tile_tok0 >= g.alias.from) return;
    const int r = wave * 32 + (lane & 31);
    const long tok = tile_tok0 + r;
    const bool rok = r < TR && tok < g.N && !(aliasing && tok >= g.alias.from);
    u32x4 rw[P][NP];
    auto fetch = [&](int seq) {           // chunk sequence: projection rows, then per channel group q, k, v rows
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const mc_half* src;
            if (seq < NC0) src = (p ? Wpl : Wph) + (long)seq * 32 * L;
            else {
                const int u = seq - NC0, cgi = u / 3, j = u - 3 * cgi;
                src = (p ? Wql : Wqh) + (long)(j * L + cgi * 32) * L;
            }
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int idx = tid + 256 * i;
                if (PC % 256 == 0 || idx < PC) rw[p][i] = *reinterpret_cast<const u32x4*>(src + (long)(idx / (L / 8)) * L + (idx % (L / 8)) * 8);
            }
        }
    };
    auto commit = [&](int b) {
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int idx = tid + 256 * i;
                if (PC % 256 == 0 || idx < PC) *reinterpret_cast<u32x4*>(Ws(b, p) + (idx / (L / 8)) * LD1 + (idx % (L / 8)) * 8) = rw[p][i];
            }
    };
    fetch(0);
    f16x8 xh[NKB], xl[NKB];
    {
        const long tk = tok < g.N ? tok : 0;
        const float w0 = rok ? g.comb_w[2 * tk] : 0.f, w1 = rok ? g.comb_w[2 * tk + 1] : 0.f;
        const long ty = (g.twin_from > 0 && tk >= g.twin_from) ? tk - g.twin_from : tk;
        const float* y0 = g.X + 2 * ty * L + hf * 8;
        const bool k0 = w0 != 0.f, k1 = w1 != 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            f32x4 v[2];
#pragma unroll
            for (int hq = 0; hq < 2; ++hq) {
                const f32x4 ya = *reinterpret_cast<const f32x4*>(y0 + 16 * kb + 4 * hq);
                const f32x4 yb = *reinterpret_cast<const f32x4*>(y0 + L + 16 * kb + 4 * hq);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[hq][i] = gelu_exact((k0 ? w0 * ya[i] : 0.f) + (k1 ? w1 * yb[i] : 0.f));
            }
            split8(v[0], v[1], xh[kb], xl[kb]);
        }
    }
    commit(0);
    fetch(1);
    __syncthreads();
    const int fo = (lane & 31) * LD1 + hf * 8;
    float* orow = g.Y + tok * g.ldy + kq;
    f32x4 bvf[NJ];
    f16x8 bh[NKB], bl[NKB];
    auto chunk = [&](int seq, const f16x8 (&fh)[NKB], const f16x8 (&fl)[NKB]) {
        f32x16 a;
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const f16x8 wh = *reinterpret_cast<const f16x8*>(Ws(seq & 1, 0) + fo + 16 * kb);
            const f16x8 wl = SPLIT ? *reinterpret_cast<const f16x8*>(Ws(seq & 1, P - 1) + fo + 16 * kb) : wh;
            a = mma3<SPLIT>(wh, wl, fh[kb], fl[kb], a);
        }
        return a;
    };
#define MC_PBH_CHUNK(seq, KEEP)                                                                             \
    {                                                                                                       \
        const f32x16 a = chunk((seq), xh, xl);                                                              \
        commit(((seq) & 1) ^ 1);                                                                            \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                     \
            const f32x4 bb = *reinterpret_cast<const f32x4*>(s_bias + (seq) * 32 + 8 * q + kq);             \
            const f32x4 v = {a[4 * q] + bb[0], a[4 * q + 1] + bb[1], a[4 * q + 2] + bb[2], a[4 * q + 3] + bb[3]}; \
            if (rok) *reinterpret_cast<f32x4*>(orow + (seq) * 32 + 8 * q) = v;                              \
            KEEP                                                                                            \
        }                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        fetch((seq) + 2);                                                                                   \
        __syncthreads();                                                                                    \
    }
#pragma unroll
    for (int c = 0; c < NKEEP; ++c) MC_PBH_CHUNK(c, bvf[4 * c + q] = v;)
    for (int c = NKEEP; c < NC0; ++c) MC_PBH_CHUNK(c, )
#undef MC_PBH_CHUNK
    frag_layernorm_h<NJ>(bvf, g.gamma, g.beta, kq);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) split8(bvf[2 * kb], bvf[2 * kb + 1], bh[kb], bl[kb]);
    auto qkv_chunk = [&](int seq, int bias0, float* sl) {
        const f32x16 a = chunk(seq, bh, bl);
        if (seq + 1 < NSEQ) commit((seq & 1) ^ 1);
        float* xr = sl + r * XS + kq;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(s_bias + bias0 + 8 * q + kq);
            const f32x4 v = {a[4 * q] + bb[0], a[4 * q + 1] + bb[1], a[4 * q + 2] + bb[2], a[4 * q + 3] + bb[3]};
            *reinterpret_cast<f32x4*>(xr + 8 * q) = v;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (seq + 2 < NSEQ) fetch(seq + 2);
    };
    BP bp(g, s_x, s_w, tile_tok0, aliasing, lane, wave);
#pragma unroll 1
    for (int cg = 0; cg < NG; ++cg) {
        const int seq = NC0 + 3 * cg;
        float* Sq = bp.slot(cg & 1);
        float* Sk = bp.slot((cg & 1) ^ 1);
        bp.begin_group();
        qkv_chunk(seq, 4 * L + cg * 32, Sq);
        __syncthreads();
        qkv_chunk(seq + 1, 5 * L + cg * 32, Sk);
        bp.after_k(Sq);
        __syncthreads();
        qkv_chunk(seq + 2, 6 * L + cg * 32, Sq);
        bp.after_v(Sk);
        __syncthreads();
        bp.finish(Sq, cg);
    }
}

// =================================================================================================
// temporal_h_k: temporal linear attention (temporal_k, mc_attn.hip; st_attention.py:137-170) with both contractions on the fp16
// MFMA.  One workgroup per (sample of the CFG-doubled batch, part); wave w owns output columns [32 w, 32 w + 32).
//   phase 1  column max / sum of K over the 77 + T tokens: fp32, as temporal_k
//   phase 2  A2[d][l] = sum_n e[n][d] V[n][l], e = exp2(k2 - max2) UNNORMALISED (<= 1: fp16-friendly; the 1 / sum of column d is
//            applied to the fp32 accumulator rows afterwards).  The contraction index n must be the 8 consecutive halves of a lane,
//            so 32-row chunks of K and V are TRANSPOSED while staging: a thread owns one column and 8 consecutive rows (8 scalar
//            loads, coalesced over the wave), converts / splits them and writes one b128 per plane into [column][32 n] tiles (64-byte
//            rows, 16-byte chunk c of row r at position c ^ ((r >> 2) & 3): conflict-free b128 fragment reads; the lane -> column map
//            swaps column bits 1..3 around so that the 8-lane groups of ds_write_b128 hit 8 different slots)
//   phase 3  y[t][l] = (1 / sum_t) sum_d q'[t][d] A2[d][l], q' = exp2(q2 - max2) unnormalised; the A2 accumulator fragment is the B
//            operand (split once into fp16 planes in registers), so the Q slab is stored with the chain permutation of its 32-chunks
//            (mc_half.h) in [32 t][L] tiles (swizzled as mlp2hd_k's W1 chunk)
// Softmax statistics, masks and the final scaling stay fp32.  SPLIT: three-product hi/lo form (fp32-class), else plain fp16 operands.
// =================================================================================================
template <int L, bool SPLIT>
__global__ __launch_bounds__(256, 2) void temporal_h_k(const float* __restrict__ mf, const float* __restrict__ tf,
                                                       const float* __restrict__ mask, float* __restrict__ yt,
                                                       int b0, int B, int T, int Nt, int H, const int* twin_flag, int skip_text) {
    static_assert(L == 128 || L == 64, "temporal_h_k: L");
    constexpr int NT = L / 32, C4 = L / 4, NSL = 256 / C4, P = SPLIT ? 2 : 1;
    constexpr int PL = L * 32;                  // halves of one transposed plane chunk [L][32 n] (= one Q slab plane [32 t][L])
    constexpr int NGT = L / 32;                 // 8-row groups of a chunk per staging thread (128 threads per matrix)
    __shared__ __attribute__((aligned(16))) float sf[2 * L + 2 * NSL * L + 32];
    __shared__ __attribute__((aligned(16))) _Float16 sh[2 * P * PL];
    float* s_m = sf;
    float* s_s = s_m + L;
    float* s_pm = s_s + L;
    float* s_ps = s_pm + NSL * L;
    float* s_qr = s_ps + NSL * L;               // [32] 1 / sum of the query rows of the current chunk
    _Float16* Kp = sh;                          // K planes (hi, lo), then V planes; phase 3 reuses the K planes as the Q slab
    _Float16* Vp = sh + P * PL;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = b0 + blockIdx.x / H, h = blockIdx.x % H;
    const float cnd = b < B ? 1.f : 0.f;
    const int bm = (twin_flag && b >= B && *twin_flag == 0) ? b - B : b;
    const float* mrow = mask + (long)(b % B) * T;
    const int Nseq = Nt + T;
    const float NEG = -1000000.f;
    const long D4 = 4 * L;
    const int hf = lane >> 5, fr = lane & 31;

    // row n of the [text | motion] sequence: pointer to its key vector (value = + L), clamped past the end
    auto row_ptr = [&](int n, int& t) -> const float* {
        const int nc = n < Nseq ? n : Nseq - 1;
        const bool txt = nc < Nt;
        t = txt ? 0 : nc - Nt;
        const float* rt = tf + ((long)b * Nt + (txt ? nc : 0)) * 2 * L;
        const float* rm = mf + (((long)bm * T + t) * H + h) * D4 + L;
        return txt ? rt : rm;
    };
    // unconditional half: whole leading 32-row chunks of its text rows (numerators exactly 0, values exactly +-0 while the sample has a
    // valid frame) are skipped in phase 2 -- see temporal_k (mc_attn.hip); the stats pass keeps all rows (its row batch is wider than the text)
    int ch0 = 0;
    if (b >= B && skip_text) {                    // (uniform per workgroup)
        int v = 0;
        for (int t = tid; t < T; t += 256) v |= mrow[t] != 0.f;
        if (__syncthreads_or(v)) ch0 = Nt / 32;
    }
    // ---- phase 1: column max / sum (log2 domain), temporal_k's pass ----
    {
        const int c4 = (tid % C4) * 4, sl = tid / C4;
        f32x4 m = {-3e38f, -3e38f, -3e38f, -3e38f}, s = {0.f, 0.f, 0.f, 0.f};
        constexpr int BATCH = 12;      // (3 round trips for the 273 rows of the 196-frame configs; no accumulators live yet)
        for (int n0 = sl; n0 < Nseq; n0 += NSL * BATCH) {
            f32x4 kk[BATCH];
            float mv[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                int t;
                const float* r = row_ptr(n0 + u * NSL, t);
                kk[u] = *reinterpret_cast<const f32x4*>(r + c4);
                const float mm = mrow[t];
                mv[u] = n0 + u * NSL < Nt ? cnd : mm;
            }
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
                for (int u = 0; u < BATCH; ++u) acc += fast_exp2(kk[u][j] - nm);
                s[j] = acc;
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
        for (int i = 0; i < NSL; ++i) S += s_ps[i * L + tid] * fast_exp2(s_pm[i * L + tid] - M);
        s_m[tid] = M;
        s_s[tid] = 1.f / S;
    }
    __syncthreads();

    // ---- phase 2 ----
    // staging role of this thread: matrix (0 K, 1 V), column, first 8-row group
    const int mat = tid >> 7, wsub = (tid >> 6) & 1;
    const int pcol = (lane & 1) | (((lane >> 3) & 1) << 1) | (((lane >> 1) & 3) << 2) | (lane & 0x30);     // lane bits 1..3 -> column bits 2, 3, 1
    const int col = (L == 128 ? 64 * wsub : 0) + pcol;
    const int ng0 = L == 128 ? 0 : 2 * wsub;
    const float colmax = s_m[col];
    _Float16* myplane = (mat ? Vp : Kp) + col * 32;
    const int wsw = (col >> 2) & 3;
    // split mode: two register sets -- chunk ch + 1 is requested BEFORE chunk ch is converted, so its round trip to L2 / HBM overlaps the
    // conversion, the barrier and the MFMAs of chunk ch (the workgroup count per CU is set by the split planes' registers anyway)
    float pfA[NGT * 8], pfB[NGT * 8];
    float pmvA, pmvB;               // mask value of row ch * 32 + fr (lanes fr and fr + 32 hold the same)
    auto prefetch = [&](int ch, float (&pf)[NGT * 8], float& pmv) {
#pragma unroll
        for (int j = 0; j < NGT; ++j)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                int t;
                const float* rp = row_ptr(ch * 32 + 8 * (ng0 + j) + r, t);
                pf[j * 8 + r] = rp[col + (mat ? L : 0)];
            }
        int t;
        const int n = ch * 32 + fr;
        (void)row_ptr(n, t);
        const float mm = mrow[t];
        pmv = n < Nt ? cnd : mm;
    };
    auto commit = [&](int ch, const float (&pf)[NGT * 8], const float pmv) {
#pragma unroll
        for (int j = 0; j < NGT; ++j) {
            f32x4 v[2];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int rr = 8 * (ng0 + j) + r;                       // row of the chunk (wave-uniform)
                const float mv = __shfl(pmv, rr, 64);
                float x;
                if (mat == 0) x = fast_exp2(fmaf(pf[j * 8 + r] + (1.f - mv) * NEG, LOG2E, -colmax));
                else x = pf[j * 8 + r] * mv;
                if (ch * 32 + rr >= Nseq) x = 0.f;
                v[r >> 2][r & 3] = x;
            }
            f16x8 hi, lo;
            split8(v[0], v[1], hi, lo);
            const int pos = ((ng0 + j) ^ wsw) * 8;
            *reinterpret_cast<f16x8*>(myplane + pos) = hi;
            if constexpr (SPLIT) *reinterpret_cast<f16x8*>(myplane + PL + pos) = lo;
        }
    };
    const bool mm_active = wave < NT;
    f32x16 acc[NT];
#pragma unroll
    for (int dt = 0; dt < NT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
    const int nch = (Nseq + 31) / 32;
    const int rsw = (fr >> 2) & 3;
    auto chunk_mma = [&]() {
        if (mm_active) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int o = ((2 * ks + hf) ^ rsw) * 8;
                const _Float16* vp = Vp + (wave * 32 + fr) * 32 + o;
                const f16x8 vh = *reinterpret_cast<const f16x8*>(vp);
                const f16x8 vl = SPLIT ? *reinterpret_cast<const f16x8*>(vp + PL) : vh;
#pragma unroll
                for (int dt = 0; dt < NT; ++dt) {
                    const _Float16* kp = Kp + (dt * 32 + fr) * 32 + o;
                    const f16x8 kh = *reinterpret_cast<const f16x8*>(kp);
                    const f16x8 kl = SPLIT ? *reinterpret_cast<const f16x8*>(kp + PL) : kh;
                    acc[dt] = mma3<SPLIT>(kh, kl, vh, vl, acc[dt]);
                }
            }
        }
    };
    prefetch(ch0, pfA, pmvA);
    if constexpr (SPLIT) {
        for (int ch = ch0; ch < nch; ch += 2) {
            if (ch + 1 < nch) prefetch(ch + 1, pfB, pmvB);
            __builtin_amdgcn_sched_barrier(0);
            commit(ch, pfA, pmvA);
            __syncthreads();
            chunk_mma();
            __syncthreads();
            if (ch + 1 < nch) {
                if (ch + 2 < nch) prefetch(ch + 2, pfA, pmvA);
                __builtin_amdgcn_sched_barrier(0);
                commit(ch + 1, pfB, pmvB);
                __syncthreads();
                chunk_mma();
                __syncthreads();
            }
        }
    } else {
        // plain fp16: one register set (128 VGPRs: four workgroups per CU hide the round trips better than the second set does at two)
        for (int ch = ch0; ch < nch; ++ch) {
            commit(ch, pfA, pmvA);
            __syncthreads();
            if (ch + 1 < nch) prefetch(ch + 1, pfA, pmvA);
            __builtin_amdgcn_sched_barrier(0);
            chunk_mma();
            __syncthreads();
        }
    }
    // 1 / column sum on the accumulator rows, then the fragment becomes the B operand of phase 3 (fp16 planes in registers)
    f16x8 a2h[NT][2], a2l[NT][2];
#pragma unroll
    for (int dt = 0; dt < NT; ++dt)
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            f32x4 v[2];
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int q = 2 * blk + qq;
                const f32x4 rs = *reinterpret_cast<const f32x4*>(s_s + dt * 32 + 8 * q + 4 * hf);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[qq][i] = acc[dt][4 * q + i] * rs[i];
            }
            split8(v[0], v[1], a2h[dt][blk], a2l[dt][blk]);
        }

    // ---- phase 3 ----
    const int ntc = (T + 31) / 32;
    constexpr int SEG = L / 8;
    const int qrow = tid >> 3, qsub = tid & 7;
    const int qsw = L == 128 ? (qrow & 15) : ((qrow >> 1) & 7);
    float qv[SEG];
    auto prefetch_q = [&](int tc) {
        const int t = tc * 32 + qrow;
        if (t < T) {
            const float* r = mf + (((long)bm * T + t) * H + h) * D4 + 3 * L + qsub * SEG;
#pragma unroll
            for (int j = 0; j < SEG; j += 4) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(r + j);
                qv[j] = x[0]; qv[j + 1] = x[1]; qv[j + 2] = x[2]; qv[j + 3] = x[3];
            }
        }
    };
    const int fsw = L == 128 ? (fr & 15) : ((fr >> 1) & 7);
    prefetch_q(0);
    for (int tc = 0; tc < ntc; ++tc) {
        {
            const int t = tc * 32 + qrow;
            float mx = -3e38f;
            if (t < T) {
#pragma unroll
                for (int j = 0; j < SEG; ++j) mx = fmaxf(mx, qv[j]);
            }
            mx = group_max(mx, 8);
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < SEG; ++j) {
                qv[j] = t < T ? fast_exp2((qv[j] - mx) * LOG2E) : 0.f;
                s += qv[j];
            }
            s = group_sum(s, 8);
            if (qsub == 0) s_qr[qrow] = t < T ? 1.f / s : 0.f;
            // quads of 4 consecutive d -> k-slot positions of the chain permutation
#pragma unroll
            for (int jq = 0; jq < SEG / 4; ++jq) {
                const int d0 = SEG * qsub + 4 * jq, s0 = d0 & 31;
                const int p0 = 16 * (s0 >> 4) + 8 * ((s0 >> 2) & 1) + 4 * ((s0 >> 3) & 1);
                const int hpos = (d0 & ~31) + p0;                        // half index inside the row
                const int a = qrow * L + (((hpos >> 3) ^ qsw) << 3) + (hpos & 4);
                typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
                f16x4 hi, lo;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float x = pinned(qv[4 * jq + i]);
                    const _Float16 hx = (_Float16)x;
                    hi[i] = hx;
                    lo[i] = (_Float16)(x - pinned((float)hx));
                }
                *reinterpret_cast<f16x4*>(Kp + a) = hi;
                if constexpr (SPLIT) *reinterpret_cast<f16x4*>(Kp + PL + a) = lo;
            }
        }
        __syncthreads();
        if (tc + 1 < ntc) prefetch_q(tc + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (mm_active) {
            f32x16 o;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
            for (int dt = 0; dt < NT; ++dt)
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    const _Float16* qp = Kp + fr * L + (((2 * (2 * dt + blk) + hf) ^ fsw) << 3);
                    const f16x8 qh = *reinterpret_cast<const f16x8*>(qp);
                    const f16x8 ql = SPLIT ? *reinterpret_cast<const f16x8*>(qp + PL) : qh;
                    o = mma3<SPLIT>(qh, ql, a2h[dt][blk], a2l[dt][blk], o);
                }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 rs = *reinterpret_cast<const f32x4*>(s_qr + 8 * q + 4 * hf);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int t = tc * 32 + 8 * q + 4 * hf + i;
                    if (t < T) yt[((long)b * T + t) * (H * L) + h * L + wave * 32 + fr] = o[4 * q + i] * rs[i];
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace

int mc_launch_split_f16(const float* x, mc_half* hi, mc_half* lo, long n, hipStream_t s) {
    if (n <= 0) return MC_OK;
    int blocks = cdiv(n, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(split_f16_k, dim3(blocks), dim3(256), 0, s, x, hi, lo, n, 1, 0);
    MC_LAUNCH_CHECK();
    return MC_OK;
}

int mc_launch_split_f16_chainperm(const float* x, mc_half* hi, mc_half* lo, long rows, int K, hipStream_t s) {
    MC_REQUIRE(K % 32 == 0, "chain-permuted split: K=%d is not a multiple of 32", K);
    const long n = rows * K;
    if (n <= 0) return MC_OK;
    int blocks = cdiv(n, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(split_f16_k, dim3(blocks), dim3(256), 0, s, x, hi, lo, n, K, 1);
    MC_LAUNCH_CHECK();
    return MC_OK;
}

int mc_launch_gemm_h(const GemmHArgs& g, bool split, hipStream_t s) {
    if (g.Ah) {            // A comes as fp16 planes [M][K] (K contiguous, no padding): the LDS-DMA kernel
        MC_REQUIRE(g.Wh && g.C && (!split || (g.Wl && g.Al)), "fp16 gemm (planes): null operand");
        MC_REQUIRE(g.N % 128 == 0 && g.K % 64 == 0 && g.ldc % 4 == 0 && (!g.R || g.ldr % 4 == 0) && (long)g.M * g.K * 2 < (1L << 31) &&
                       (long)g.N * g.K * 2 < (1L << 31),
                   "fp16 gemm (planes): unsupported shape (M=%d N=%d K=%d)", g.M, g.N, g.K);
        if (g.M <= 0) return MC_OK;
        dim3 grid(cdiv(g.M, 128) * (g.N / 128));
        if (!g.a_fm) MC_LEDGER(split ? "gemm_hd_k<true>" : "gemm_hd_k<false>", grid, 2.0 * g.M * g.N * g.K);      // (fp32-equivalent product: the split form runs 3 fp16 MFMAs per operand pair)
        if (g.a_fm) {          // fragment-major planes: A straight into registers (M % 32 == 0, K % 64 == 0; the producer wrote the rows of THIS launch)
            MC_REQUIRE(g.M % 32 == 0 && g.K >= 192, "fp16 gemm (fragment-major planes): M=%d K=%d", g.M, g.K);
            // its A fragments arrive by inline-asm loads the compiler does not track: a spilled (copied) fragment register would be read before it landed
            static const int scratch = [] {
                hipFuncAttributes a0, a1;
                if (hipFuncGetAttributes(&a0, (const void*)gemm_hf_k<false>) != hipSuccess || hipFuncGetAttributes(&a1, (const void*)gemm_hf_k<true>) != hipSuccess) return -1;
                return (int)(a0.localSizeBytes + a1.localSizeBytes);
            }();
            MC_REQUIRE(scratch == 0, "gemm_hf_k of this build uses scratch memory (%d bytes; -1 = attributes unreadable): its asynchronous fragment loads are unsafe", scratch);
            MC_LEDGER(split ? "gemm_hf_k<true>" : "gemm_hf_k<false>", grid, 2.0 * g.M * g.N * g.K);
            if (split) hipLaunchKernelGGL(gemm_hf_k<true>, grid, dim3(256), 0, s, g);
            else hipLaunchKernelGGL(gemm_hf_k<false>, grid, dim3(256), 0, s, g);
            MC_LAUNCH_CHECK();
            return MC_OK;
        }
        const bool pre = (g.acc_init & 1) && g.R, init = (g.acc_init & 2) && g.R && g.bias && g.act == ACT_NONE;
        if (split) {
            if (pre) hipLaunchKernelGGL((gemm_hd_k<true, false, true>), grid, dim3(256), 0, s, g);
            else hipLaunchKernelGGL(gemm_hd_k<true>, grid, dim3(256), 0, s, g);
        } else if (init) hipLaunchKernelGGL((gemm_hd_k<false, true>), grid, dim3(256), 0, s, g);
        else if (pre) hipLaunchKernelGGL((gemm_hd_k<false, false, true>), grid, dim3(256), 0, s, g);
        else hipLaunchKernelGGL(gemm_hd_k<false>, grid, dim3(256), 0, s, g);
        MC_LAUNCH_CHECK();
        return MC_OK;
    }
    MC_REQUIRE(g.A && g.Wh && g.C && (!split || g.Wl), "fp16 gemm: null operand");
    MC_REQUIRE(g.N % 128 == 0 && g.K % 32 == 0 && g.lda % 4 == 0 && g.ldc % 4 == 0 && (!g.R || g.ldr % 4 == 0),
               "fp16 gemm: unsupported shape (M=%d N=%d K=%d)", g.M, g.N, g.K);
    if (g.M <= 0) return MC_OK;
    dim3 grid(cdiv(g.M, 128) * (g.N / 128));
    MC_LEDGER(split ? "gemm_h_k<true>" : "gemm_h_k<false>", grid, 2.0 * g.M * g.N * g.K);
    if (split) hipLaunchKernelGGL(gemm_h_k<true>, grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL(gemm_h_k<false>, grid, dim3(256), 0, s, g);
    MC_LAUNCH_CHECK();
    return MC_OK;
}

bool mc_mlp_h_supported(int L, int hidden) { return (L == 32 || L == 64 || L == 128) && hidden % 32 == 0 && hidden >= 32 && hidden <= 1024; }

int mc_launch_mlp_h(int mode, const MlpArgs& g, const mc_half* W1h, const mc_half* W1l, const mc_half* W2h, const mc_half* W2l,
                    bool split, int groups, int max_tiles, hipStream_t s) {
    MC_REQUIRE(mc_mlp_h_supported(g.L, g.hidden), "fp16 fused mlp: L=%d hidden=%d unsupported", g.L, g.hidden);
    MC_REQUIRE(g.ldx % 4 == 0 && g.ldy % 4 == 0 && g.x_gstride % 4 == 0 && g.y_gstride % 4 == 0 && g.nsplit == 1, "fp16 fused mlp: strides / nsplit");
    MC_REQUIRE(W1h && W2h && (!split || (W1l && W2l)), "fp16 fused mlp: null weight plane");
    dim3 grid;
    if (mode == MLP_EXPERT) {
        if (max_tiles <= 0) return MC_OK;
        grid = dim3(max_tiles, 1, 1);
    } else {
        if (g.M <= 0) return MC_OK;
        grid = dim3(cdiv(g.M, 128), groups, 1);
    }
    if (mc_ledger_on) {
        char name[48];
        snprintf(name, sizeof(name), "%s<%d, %d, %s>", (g.dma && (g.L == 128 || g.L == 64)) ? "mlp2hd_k" : "mlp2_h_k", g.L, mode == MLP_EXPERT ? 0 : 1,
                 split ? "true" : "false");
        const double rows = mode == MLP_EXPERT ? (double)g.ledger_rows : (double)g.M * groups;
        MC_LEDGER(name, grid, rows * 4.0 * g.L * g.hidden);
    }
    if (g.dma && (g.L == 128 || g.L == 64)) {     // LDS-DMA staged weight chunks (chain bit 18): the same bits
#define MC_MLPHD(LL, MM, SS) hipLaunchKernelGGL((mlp2hd_k<LL, MM, SS>), grid, dim3(256), 0, s, g, W1h, W1l, W2h, W2l)
        if (g.L == 128) {
            if (mode == MLP_EXPERT) { if (split) MC_MLPHD(128, MLP_EXPERT, true); else MC_MLPHD(128, MLP_EXPERT, false); }
            else { if (split) MC_MLPHD(128, MLP_PARTS, true); else MC_MLPHD(128, MLP_PARTS, false); }
        } else {
            if (mode == MLP_EXPERT) { if (split) MC_MLPHD(64, MLP_EXPERT, true); else MC_MLPHD(64, MLP_EXPERT, false); }
            else { if (split) MC_MLPHD(64, MLP_PARTS, true); else MC_MLPHD(64, MLP_PARTS, false); }
        }
#undef MC_MLPHD
        MC_LAUNCH_CHECK();
        return MC_OK;
    }
#define MC_MLPH(LL, MM, SS) hipLaunchKernelGGL((mlp2_h_k<LL, MM, SS>), grid, dim3(256), 0, s, g, W1h, W1l, W2h, W2l)
#define MC_MLPH_CASE(LL)                                                       \
    case LL:                                                                   \
        if (mode == MLP_EXPERT) { if (split) MC_MLPH(LL, MLP_EXPERT, true); else MC_MLPH(LL, MLP_EXPERT, false); } \
        else { if (split) MC_MLPH(LL, MLP_PARTS, true); else MC_MLPH(LL, MLP_PARTS, false); }                      \
        break;
    switch (g.L) {
        MC_MLPH_CASE(128)
        MC_MLPH_CASE(64)
        MC_MLPH_CASE(32)
        default: mc_set_error("fp16 fused mlp: L=%d unsupported", g.L); return MC_ERR_ARG;
    }
#undef MC_MLPH_CASE
#undef MC_MLPH
    MC_LAUNCH_CHECK();
    return MC_OK;
}

int mc_launch_pqbody_h(const RowChainArgs& g, int H, const mc_half* Wph, const mc_half* Wpl, const mc_half* Wqh, const mc_half* Wql, bool split,
                       hipStream_t s) {
    MC_REQUIRE((g.L == 128 || g.L == 64) && H == 12, "fp16 pqbody: L=%d H=%d unsupported (128 / 64, 12)", g.L, H);
    MC_REQUIRE(g.Nout == 4 * g.L && g.ldy == 4 * g.L && g.bias && g.bias2 && g.wsm && g.ys && Wph && Wqh && (!split || (Wpl && Wql)),
               "fp16 pqbody: bad arguments");
    MC_REQUIRE(g.pad_row >= g.N, "fp16 pqbody: pad_row (128 padding rows of Y behind the last token) not set");
    MC_REQUIRE(g.tok0 % H == 0 && g.N % H == 0, "fp16 pqbody: token range [%ld, %ld) is not made of whole frames", g.tok0, g.N);
    if (g.N <= g.tok0) return MC_OK;
    dim3 grid(cdiv((g.N - g.tok0) / H, 128 / H));
    RowChainArgs gg = g;
    gg.nblk1 = 0;
    if (g.nblk1 != 0) {           // second token range in the same launch (mc_launch_pqbody)
        MC_REQUIRE(g.tok2 % H == 0 && g.N2 % H == 0 && g.N2 >= g.tok2 && g.pad_row >= g.N2, "fp16 pqbody: bad second token range [%ld, %ld)", g.tok2, g.N2);
        gg.nblk1 = (int)grid.x;
        grid.x += cdiv((g.N2 - g.tok2) / H, 128 / H);
    }
    if (mc_ledger_on) {
        char name[40];
        snprintf(name, sizeof(name), "pqbody_h_k<%d, 12, %s>", g.L, split ? "true" : "false");
        const double toks = (double)(mc_ledger_tokens(g)), hd = g.L / 8.0;      // (aliased twins -- the tail of the range, or the optional second range -- exit at once in the usual case)
        MC_LEDGER(name, grid, 2.0 * toks * 7.0 * g.L * g.L + (toks / H) * (2.0 * H * H * g.L + 8 * 2.0 * (2.0 * H * hd * hd)));
    }
    if (g.L == 128) {
        if (split) hipLaunchKernelGGL((pqbody_h_k<128, 12, true>), grid, dim3(256), 0, s, gg, Wph, Wpl, Wqh, Wql);
        else hipLaunchKernelGGL((pqbody_h_k<128, 12, false>), grid, dim3(256), 0, s, gg, Wph, Wpl, Wqh, Wql);
    } else {
        if (split) hipLaunchKernelGGL((pqbody_h_k<64, 12, true>), grid, dim3(256), 0, s, gg, Wph, Wpl, Wqh, Wql);
        else hipLaunchKernelGGL((pqbody_h_k<64, 12, false>), grid, dim3(256), 0, s, gg, Wph, Wpl, Wqh, Wql);
    }
    MC_LAUNCH_CHECK();
    return MC_OK;
}

int mc_launch_projqkv_h(const RowChainArgs& g, const mc_half* Wph, const mc_half* Wpl, const mc_half* Wqh, const mc_half* Wql, bool split,
                        hipStream_t s) {
    MC_REQUIRE(g.Nout == 4 * g.L && g.ldy % 4 == 0 && g.ldy2 % 4 == 0 && g.bias && g.bias2 && g.Y2 && Wph && Wqh && (!split || (Wpl && Wql)),
               "fp16 projqkv: bad arguments");
    if (g.N <= g.tok0) return MC_OK;
    dim3 grid(cdiv(g.N - g.tok0, 128));
    if (mc_ledger_on) {
        char name[40];
        snprintf(name, sizeof(name), "projqkv_h_k<%d, %s>", g.L, split ? "true" : "false");
        MC_LEDGER(name, grid, 2.0 * (double)mc_ledger_tokens(g) * 7.0 * g.L * g.L);
    }
#define MC_PQH(LL) case LL: if (split) hipLaunchKernelGGL((projqkv_h_k<LL, true>), grid, dim3(256), 0, s, g, Wph, Wpl, Wqh, Wql); \
                            else hipLaunchKernelGGL((projqkv_h_k<LL, false>), grid, dim3(256), 0, s, g, Wph, Wpl, Wqh, Wql); break;
    switch (g.L) {
        MC_PQH(128)
        MC_PQH(64)
        MC_PQH(32)
        default: mc_set_error("fp16 projqkv: L=%d unsupported", g.L); return MC_ERR_ARG;
    }
#undef MC_PQH
    MC_LAUNCH_CHECK();
    return MC_OK;
}

int mc_launch_temporal_h(const float* mf, const float* tf, const float* mask, float* yt, int b0, int nb, int B, int T, int Nt, int H, int L,
                         bool split, hipStream_t s, const int* twin_flag, bool skip_text) {
    MC_REQUIRE(L == 128 || L == 64, "fp16 temporal attention: latent_dim=%d unsupported (128, 64)", L);
    const int sk = skip_text ? 1 : 0;
    if (nb <= 0) return MC_OK;
    dim3 grid(nb * H), blk(256);
    if (mc_ledger_on) {
        char name[40];
        snprintf(name, sizeof(name), "temporal_h_k<%d, %s>", L, split ? "true" : "false");
        MC_LEDGER(name, grid, (double)nb * H * (2.0 * (Nt + T) * L * L + 2.0 * T * L * L));
    }
    if (L == 128) {
        if (split) hipLaunchKernelGGL((temporal_h_k<128, true>), grid, blk, 0, s, mf, tf, mask, yt, b0, B, T, Nt, H, twin_flag, sk);
        else hipLaunchKernelGGL((temporal_h_k<128, false>), grid, blk, 0, s, mf, tf, mask, yt, b0, B, T, Nt, H, twin_flag, sk);
    } else {
        if (split) hipLaunchKernelGGL((temporal_h_k<64, true>), grid, blk, 0, s, mf, tf, mask, yt, b0, B, T, Nt, H, twin_flag, sk);
        else hipLaunchKernelGGL((temporal_h_k<64, false>), grid, blk, 0, s, mf, tf, mask, yt, b0, B, T, Nt, H, twin_flag, sk);
    }
    MC_LAUNCH_CHECK();
    return MC_OK;
}
