// Round-6 lab for the fp16 plane GEMM (NOT part of the library): C = A W^T + bias + R on v_mfma_f32_32x32x16_f16, FiLM shape
// (M x 1536 x 1536, fp16 planes of A and W, fp32 bias + residual epilogue).  VERDICT r05 item 1 asks for a new loop skeleton, gated in
// three stages: (i) MFMAs only, (ii) + operands, (iii) + fp32 residual epilogue.  One generic kernel, hr_k, spans the structures:
//   tile TM x TN, k-slabs of BK halves in an ST-deep LDS-DMA ring (counted vmcnt, ONE raw s_barrier per slab, no drain), WM x WN waves,
//   OCC workgroups per CU;  hd_k = the library's gemm_hd_k<false> (128 x 128 x 64, 2 stages, vmcnt(0) + __syncthreads per k-tile).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_h6_lab.hip -o tools/_bin/gemm_h6_lab && tools/_bin/gemm_h6_lab
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 half_t;

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int NX = 8;
    int q = nwg / NX, r = nwg % NX, xcd = bid % NX, idx = bid / NX;
    return ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
__device__ __forceinline__ void dma16h(unsigned voff, const half_t* sbase, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte) : "memory");
}

struct HArgs {
    const half_t *A, *W;
    const float *bias, *R;
    float* C;
    int M, N, K;
};

// ---------------------------------------------------------------------------------------------------------------------
// reference: the library's gemm_hd_k<false>
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void hd_k(HArgs g) {
    constexpr int BKH = 64, CH = 8, RPI = 8, PT = 128 * BKH, NQ = 4;
    __shared__ __attribute__((aligned(16))) half_t smem[2 * 2 * PT];
    auto tile = [&](int buf, int op) { return smem + (buf * 2 + op) * PT; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, frow = lane & 31, hf = lane >> 5;
    const int ntn = g.N / 128;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = bid / ntn, tn = bid % ntn;
    const int row0 = tm * 128, nrows = min(128, g.M - row0);
    const int dr = lane / CH, dpos = lane % CH;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    unsigned goa[NQ], gow[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int r = 32 * wave + q * RPI + dr;
        const int ar = min(tm * 128 + r, g.M - 1);
        goa[q] = (unsigned)(((long)ar * g.K + (dpos ^ (r & 7)) * 8) * 2);
        gow[q] = (unsigned)(((long)(tn * 128 + r) * g.K + (dpos ^ (r & 7)) * 8) * 2);
    }
    const unsigned lds0 = (unsigned)(size_t)smem;
    auto issue = [&](int kt, int buf) {
        const half_t* ap = g.A + kt * BKH;
        const half_t* wp = g.W + kt * BKH;
        const unsigned la = lds0 + (unsigned)(((buf * 2 + 0) * PT + 32 * wave_u * BKH) * 2);
        const unsigned lw = lds0 + (unsigned)(((buf * 2 + 1) * PT + 32 * wave_u * BKH) * 2);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            dma16h(goa[q], ap, la + q * RPI * BKH * 2);
            dma16h(gow[q], wp, lw + q * RPI * BKH * 2);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    const int nk = g.K / BKH;
    const int sw = frow & 7;
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) issue(kt + 1, buf ^ 1);
        const int ra = (wm * 64 + frow) * BKH, rw = (wn * 64 + frow) * BKH;
#pragma unroll
        for (int s = 0; s < BKH / 16; ++s) {
            const int pos = ((2 * s + hf) ^ sw) * 8;
            f16x8 fa[2], fw[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = *reinterpret_cast<const f16x8*>(tile(buf, 0) + ra + i * 32 * BKH + pos);
                fw[i] = *reinterpret_cast<const f16x8*>(tile(buf, 1) + rw + i * 32 * BKH + pos);
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[ni], fa[mi], acc[mi][ni], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int m = wm * 64 + mi * 32 + frow;
        if (m >= nrows) continue;
        float* crow = g.C + (long)(row0 + m) * g.N;
        const float* rrow = g.R + (long)(row0 + m) * g.N;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = tn * 128 + wn * 64 + ni * 32 + 8 * q + 4 * hf;
                f32x4 v = {acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]};
                v += *reinterpret_cast<const f32x4*>(g.bias + n);
                v += *reinterpret_cast<const f32x4*>(rrow + n);
                *reinterpret_cast<f32x4*>(crow + n) = v;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// hr_k: generic staged ring.  LDS slot = [TM rows of A | TN rows of W][BK halves], 16-byte chunk c of row r at position c ^ swz(r)
// (BK = 64: (r >> 1) & 7, BK = 32: (r >> 2) & 3 -- conflict-free for the ds_read_b128 lane groups of the guide's LDS table).
// FLAGS: 1 no residual read / (almost) no store, 2 no DMA inside the loop (stale operands), 4 no barrier in the loop (invalid, with 2),
//        8 odd workgroups start half a tile late (s_sleep) to put the two workgroups of a CU out of phase, 16 s_setprio(1) around the MFMAs,
//        32 the slab's DMA pieces are issued one at a time BETWEEN the MFMAs (pinned by sched_barrier) instead of as a burst behind the barrier,
//        64 the accumulators start as R + bias (loads issued at the top of the tile, in flight during the DMA prologue): the epilogue is stores only
//           (another fp32 summation order: not bit-equal to the library kernel), 128 both workgroups of a CU out of phase by a longer sleep
// ---------------------------------------------------------------------------------------------------------------------
template <int N_>
__device__ __forceinline__ void wait_vm() {
    if constexpr (N_ == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N_ == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N_ == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N_ == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N_ == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N_ == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N_ == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N_ == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N_ == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N_ == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if constexpr (N_ == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N_ == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N_ == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else if constexpr (N_ == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else static_assert(N_ < 0, "add the vmcnt literal");
}

template <int TM, int TN, int BK, int ST, int WM, int WN, int OCC, int FLAGS>
__global__ __launch_bounds__(WM * WN * 64, (OCC * WM * WN + 3) / 4) void hr_k(HArgs g) {
    constexpr int NW = WM * WN, NT = NW * 64;
    constexpr int RB = BK * 2, CH = BK / 8, RPI = 64 / CH;       // row bytes, 16-byte chunks per row, rows per wave-wide DMA instruction
    constexpr int ROWS = TM + TN, SLOT = ROWS * RB;              // bytes per ring slot
    constexpr int NP = ROWS / RPI, P = NP / NW;                  // DMA instructions per slab, per wave
    static_assert(NP % NW == 0 && TM % RPI == 0, "pieces must divide over the waves");
    constexpr int D = ST - 1;                                    // slabs in flight ahead of the one being read
    constexpr int MI = TM / WM / 32, NI = TN / WN / 32, KS = BK / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave / WN, wn = wave % WN, frow = lane & 31, hf = lane >> 5;
    const int ntn = g.N / TN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = bid / ntn, tn = bid % ntn;
    const int row0 = tm * TM, nrows = min(TM, g.M - row0);
    auto swz = [](int r) { return BK == 64 ? ((r >> 1) & 7) : ((r >> 2) & 3); };
    if (FLAGS & 8) {
        if (blockIdx.x >= 256 && blockIdx.x < 512)              // (the second workgroup slot of each CU, if the dispatcher fills one slot per CU first)
            for (int i = 0; i < ((FLAGS & 128) ? 12 : 4); ++i) __builtin_amdgcn_s_sleep(127);
    }
    // DMA pieces of this wave: piece p covers slab rows [(wave P + p) RPI, + RPI); lane -> row lane / CH, LDS position lane % CH
    const int dr = lane / CH, dpos = lane % CH;
    unsigned goff[P];
    bool isA[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int r = (wave_u * P + p) * RPI + dr;               // slab row
        isA[p] = (wave_u * P + p) * RPI < TM;                    // wave-uniform
        const int lr = isA[p] ? r : r - TM;                      // row inside its operand tile (TM % 8 == 0: the swizzle term of r and lr agree)
        const long grow = isA[p] ? (long)min(row0 + lr, g.M - 1) : (long)(tn * TN + lr);
        goff[p] = (unsigned)((grow * g.K + (dpos ^ swz(r)) * 8) * 2);
    }
    const unsigned lds0 = (unsigned)(size_t)smem;
    auto issue = [&](int s) {
        const unsigned slot = lds0 + (unsigned)((s % ST) * SLOT);
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const half_t* base = (isA[p] ? g.A : g.W) + s * BK;
            dma16h(goff[p], base, slot + (unsigned)((wave_u * P + p) * RPI * RB));
        }
    };
    f32x16 acc[MI][NI];
    if (FLAGS & 64) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = min(wm * (TM / WM) + i * 32 + frow, nrows - 1);
            const float* rrow = g.R + (long)(row0 + m) * g.N;
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = tn * TN + wn * (TN / WN) + j * 32 + 8 * q + 4 * hf;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(rrow + n) + *reinterpret_cast<const f32x4*>(g.bias + n);
                    acc[i][j][4 * q] = v[0]; acc[i][j][4 * q + 1] = v[1]; acc[i][j][4 * q + 2] = v[2]; acc[i][j][4 * q + 3] = v[3];
                }
        }
    } else {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    const int ns = g.K / BK;
    // fragment byte offsets inside a slot (row-dependent swizzle term folded per k-step below)
    int arow[MI], wrow[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) arow[i] = wm * (TM / WM) + i * 32 + frow;
#pragma unroll
    for (int j = 0; j < NI; ++j) wrow[j] = TM + wn * (TN / WN) + j * 32 + frow;
#pragma unroll
    for (int s = 0; s < D; ++s)
        if (s < ns) issue(s);
    for (int s = 0; s < ns; ++s) {
        if (!(FLAGS & 2) || s == 0) {
            // slabs s .. s + D - 1 are in flight: slab s has landed when at most (D - 1) P younger pieces are outstanding
            if (s + D - 1 < ns) wait_vm<(D - 1) * P>();
            else wait_vm<0>();                                   // (tail: fewer slabs behind this one)
        }
        if (!(FLAGS & 4)) __builtin_amdgcn_s_barrier();
        const bool dma_now = !(FLAGS & 2) && s + D < ns;
        if (!(FLAGS & 32) && dma_now) issue(s + D);            // into the slot of slab s - 1: every wave has passed the barrier, so has read it
        const unsigned char* sl = smem + (s % ST) * SLOT;
        if (FLAGS & 16) __builtin_amdgcn_s_setprio(1);
        if (FLAGS & 32) {
            // all fragments of the slab first (KS (MI + NI) ds_read_b128), then the MFMAs with one DMA piece behind every STRIDE-th of them
            f16x8 fa[KS][MI], fw[KS][NI];
#pragma unroll
            for (int k = 0; k < KS; ++k) {
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[k][i] = *reinterpret_cast<const f16x8*>(sl + arow[i] * RB + (((2 * k + hf) ^ swz(arow[i])) * 16));
#pragma unroll
                for (int j = 0; j < NI; ++j) fw[k][j] = *reinterpret_cast<const f16x8*>(sl + wrow[j] * RB + (((2 * k + hf) ^ swz(wrow[j])) * 16));
            }
            constexpr int NMF = KS * MI * NI, STRIDE = NMF / P > 0 ? NMF / P : 1;
            const unsigned slot = lds0 + (unsigned)(((s + D) % ST) * SLOT);
            int n = 0;
#pragma unroll
            for (int k = 0; k < KS; ++k)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[k][j], fa[k][i], acc[i][j], 0, 0, 0);
                        if (n % STRIDE == 0 && n / STRIDE < P) {
                            const int p = n / STRIDE;
                            if (dma_now) {
                                __builtin_amdgcn_sched_barrier(0);
                                dma16h(goff[p], (isA[p] ? g.A : g.W) + (s + D) * BK, slot + (unsigned)((wave_u * P + p) * RPI * RB));
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                        ++n;
                    }
        } else {
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                f16x8 fa[MI], fw[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[i] = *reinterpret_cast<const f16x8*>(sl + arow[i] * RB + (((2 * k + hf) ^ swz(arow[i])) * 16));
#pragma unroll
                for (int j = 0; j < NI; ++j) fw[j] = *reinterpret_cast<const f16x8*>(sl + wrow[j] * RB + (((2 * k + hf) ^ swz(wrow[j])) * 16));
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[j], fa[i], acc[i][j], 0, 0, 0);
            }
        }
        if (FLAGS & 16) __builtin_amdgcn_s_setprio(0);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = wm * (TM / WM) + i * 32 + frow;
        if (m >= nrows) continue;
        float* crow = g.C + (long)(row0 + m) * g.N;
        const float* rrow = g.R + (long)(row0 + m) * g.N;
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = tn * TN + wn * (TN / WN) + j * 32 + 8 * q + 4 * hf;
                f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                if (FLAGS & 64) {
                    *reinterpret_cast<f32x4*>(crow + n) = v;
                    continue;
                }
                v += *reinterpret_cast<const f32x4*>(g.bias + n);
                if (FLAGS & 1) {
                    if (v[0] == 1234.56789f) *reinterpret_cast<f32x4*>(crow + n) = v;
                } else {
                    v += *reinterpret_cast<const f32x4*>(rrow + n);
                    *reinterpret_cast<f32x4*>(crow + n) = v;
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// ha_k: the A operand NEVER touches LDS.  The producer would write the activation planes FRAGMENT-MAJOR -- for every 32-row block and every 16-wide k-step the
// 64 lanes' 16-byte MFMA operands contiguous (1 KB: [row block][k-step][lane][8 halves]) -- so a wave fetches the fragments of ITS 32 rows with one coalesced
// global_load_dwordx4 per k-step, straight into VGPRs (no LDS-DMA piece, no ds_read, no barrier dependency for A).  TM / 32 waves, each 32 rows x TN columns;
// only W rides the LDS-DMA ring (TN rows x BK halves per slot).  A fragments are requested two slabs ahead into a 3-deep register ring (inline asm, counted
// vmcnt together with the W pieces: loads return in order).  Result identical to the library kernel (the same k order per output).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void gload16(f16x8& dst, const half_t* base, unsigned voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(dst) : "v"(voff), "s"(base) : "memory");
}

template <int TM, int TN, int ST, int OCC, int FLAGS>
__global__ __launch_bounds__(TM / 32 * 64, (OCC * (TM / 32) + 3) / 4) void ha_k(HArgs g, const half_t* __restrict__ Af) {
    constexpr int BK = 64, NW = TM / 32, RB = BK * 2, RPI = 8, KS = BK / 16;
    constexpr int SLOT = TN * RB, NP = TN / RPI, P = NP / NW;          // W pieces per slab, per wave
    static_assert(NP % NW == 0, "W pieces must divide over the waves");
    constexpr int D = ST - 1, NI = TN / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int frow = lane & 31, hf = lane >> 5;
    const int ntn = g.N / TN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = bid / ntn, tn = bid % ntn;
    const int row0 = tm * TM, nrows = min(TM, g.M - row0);
    auto swz = [](int r) { return (r >> 1) & 7; };
    const int dr = lane / 8, dpos = lane % 8;
    unsigned goff[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int r = (wave_u * P + p) * RPI + dr;
        goff[p] = (unsigned)(((long)(tn * TN + r) * g.K + (dpos ^ swz(r)) * 8) * 2);
    }
    const unsigned lds0 = (unsigned)(size_t)smem;
    auto issue_w = [&](int s) {
        const unsigned slot = lds0 + (unsigned)((s % ST) * SLOT);
#pragma unroll
        for (int p = 0; p < P; ++p) dma16h(goff[p], g.W + s * BK, slot + (unsigned)((wave_u * P + p) * RPI * RB));
    };
    // fragment-major A: row block rb = (row0 + 32 wave) / 32 (rows past M: the last block, never stored), k-step ks -> 1 KB at ((rb * K/16 + ks) * 64 + lane) * 16 bytes
    const int nrb = (g.M + 31) / 32;
    const int rb = min((row0 >> 5) + wave_u, nrb - 1);
    const unsigned abase = (unsigned)(((long)rb * (g.K / 16) * 64 + lane) * 16);
    f16x8 fa[3][KS];
    auto issue_a = [&](int s, int set) {
#pragma unroll
        for (int k = 0; k < KS; ++k) gload16(fa[set][k], Af, abase + (unsigned)((s * KS + k) * 1024));
    };
    f32x16 acc[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int ns = g.K / BK;
    int wrow[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) wrow[j] = j * 32 + frow;
    // prologue: slabs 0 .. D - 1 of W and of the A fragments, interleaved in the order the loop keeps (W(s), A(s))
    static_assert(D == 2, "ha_k: the register ring is written for two slabs in flight");
    issue_w(0); issue_a(0, 0);
    issue_w(1); issue_a(1, 1);
    auto body = [&](int s, int set, int nset) {
        // outstanding, oldest first: W(s) A(s) W(s + 1) A(s + 1): slab s has landed when at most P + KS younger operations remain
        if (s + 1 < ns) wait_vm<P + KS>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (s + 2 < ns) { issue_w(s + 2); issue_a(s + 2, nset); }
        const unsigned char* sl = smem + (s % ST) * SLOT;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            f16x8 fw[NI];
#pragma unroll
            for (int j = 0; j < NI; ++j) fw[j] = *reinterpret_cast<const f16x8*>(sl + wrow[j] * RB + (((2 * k + hf) ^ swz(wrow[j])) * 16));
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[j], fa[set][k], acc[j], 0, 0, 0);
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
    const int m = wave * 32 + frow;
    if (m < nrows) {
        float* crow = g.C + (long)(row0 + m) * g.N;
        const float* rrow = g.R + (long)(row0 + m) * g.N;
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = tn * TN + j * 32 + 8 * q + 4 * hf;
                f32x4 v = {acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]};
                v += *reinterpret_cast<const f32x4*>(g.bias + n);
                if (FLAGS & 1) {
                    if (v[0] == 1234.56789f) *reinterpret_cast<f32x4*>(crow + n) = v;
                } else {
                    v += *reinterpret_cast<const f32x4*>(rrow + n);
                    *reinterpret_cast<f32x4*>(crow + n) = v;
                }
            }
    }
}

// bare MFMA stream: no memory, no barrier -- the clock-limited ceiling of v_mfma_f32_32x32x16_f16 on operands that change every instruction
__global__ __launch_bounds__(256, 2) void bare_k(float* out, int iters, const half_t* __restrict__ vals) {
    f32x16 acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[c][q] = 0.f;
    f16x8 a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a[j] = *reinterpret_cast<const f16x8*>(vals + ((threadIdx.x * 4 + j) & 1023) * 8);
        b[j] = *reinterpret_cast<const f16x8*>(vals + 8192 + ((threadIdx.x * 4 + j) & 1023) * 8);
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(j + c) & 3], b[j], acc[c], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int q = 0; q < 16; ++q) s += acc[c][q];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// ---------------------------------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------------------------------
static half_t *dA, *dW, *dAf;
static float *dB, *dR, *dC, *dC2;
static int NCU;

template <class F>
static float time_us(F&& launch, int reps = 20) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms * 1000.f / reps;
}
static void report(const char* name, int M, float us) {
    printf("  %-86s M=%5d  %8.1f us  %7.1f TFLOP/s  %.3f of 2.5 PF\n", name, M, us, 2.0 * M * 1536 * 1536 / us * 1e-6, 2.0 * M * 1536 * 1536 / us * 1e-6 / 2500.0);
    fflush(stdout);
}
static bool compare(int M, const char* what) {
    const int N = 1536;
    std::vector<float> a((size_t)M * N), b((size_t)M * N);
    hipMemcpy(a.data(), dC, a.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), dC2, b.size() * 4, hipMemcpyDeviceToHost);
    size_t nd = 0; double md = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        if (memcmp(&a[i], &b[i], 4)) ++nd;
        md = std::max(md, (double)std::fabs(a[i] - b[i]));
    }
    printf("    %s: %zu of %zu words differ, max |d| = %.3g\n", what, nd, a.size(), md);
    return nd == 0;
}

template <int TM, int TN, int BK, int ST, int WM, int WN, int OCC, int FLAGS>
static void run_hr(const char* name, int M, bool check) {
    HArgs g{dA, dW, dB, dR, dC2, M, 1536, 1536};
    const int grid = ((M + TM - 1) / TM) * (1536 / TN);
    constexpr int lds = ST * (TM + TN) * BK * 2;
    auto kern = hr_k<TM, TN, BK, ST, WM, WN, OCC, FLAGS>;
    static bool once = [&] { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds); return true; }();
    (void)once;
    if (check) hipMemset(dC2, 0xff, (size_t)M * 1536 * 4);
    const float us = time_us([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WN * 64), lds, 0, g); });
    char buf[160];
    snprintf(buf, sizeof(buf), "hr_k %3dx%3d BK%d ring %d, %d x %d waves, %d WG/CU, %3d KB LDS%s", TM, TN, BK, ST, WM, WN, OCC, lds / 1024, name);
    report(buf, M, us);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("    !! %s\n", hipGetErrorString(e)); exit(1); }
    if (check) compare(M, "vs the library kernel");
}

// run one kernel back to back for ~8 s so that rocm-smi can sample clock and power under it: 0 bare MFMA stream, 1 library kernel, 2 hr_k 256x128 without
// epilogue traffic, 3 hr_k 256x128 MFMAs + fragment reads only
static int power_loop(int which);
template <int TM, int TN, int ST, int OCC, int FLAGS>
static void run_ha(const char* name, int M, bool check) {
    HArgs g{dA, dW, dB, dR, dC2, M, 1536, 1536};
    const int grid = ((M + TM - 1) / TM) * (1536 / TN);
    constexpr int lds = ST * TN * 64 * 2;
    auto kern = ha_k<TM, TN, ST, OCC, FLAGS>;
    static bool once = [&] { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds); return true; }();
    (void)once;
    if (check) hipMemset(dC2, 0xff, (size_t)M * 1536 * 4);
    const float us = time_us([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(TM / 32 * 64), lds, 0, g, dAf); });
    char buf[160];
    snprintf(buf, sizeof(buf), "ha_k %3dx%3d BK64 ring %d, %d x 1 waves, %d WG/CU, %3d KB LDS, A fragment-major -> registers%s", TM, TN, ST, TM / 32, OCC, lds / 1024, name);
    report(buf, M, us);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("    !! %s\n", hipGetErrorString(e)); exit(1); }
    if (check) compare(M, "vs the library kernel");
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    NCU = prop.multiProcessorCount;
    printf("CUs: %d\n", NCU);
    const size_t MMAX = 32768, N = 1536, K = 1536;
    hipMalloc(&dA, MMAX * K * 2); hipMalloc(&dW, N * K * 2); hipMalloc(&dB, N * 4);
    hipMalloc(&dR, MMAX * N * 4); hipMalloc(&dC, MMAX * N * 4); hipMalloc(&dC2, MMAX * N * 4);
    std::vector<half_t> hA(MMAX * K), hW(N * K);
    std::vector<float> hR(MMAX * N), hB(N);
    srand(1);
    auto gauss = [] {
        const double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
        return (float)(std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2));
    };
    for (auto& x : hA) x = (half_t)gauss();
    for (auto& x : hR) x = gauss();
    for (auto& x : hW) x = (half_t)(gauss() * 0.0255f);
    for (auto& x : hB) x = gauss();
    hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
    {   // fragment-major copy of A: [row block of 32][k-step of 16][lane][8 halves], lane -> row lane & 31, k = 8 (lane >> 5) + i
        std::vector<half_t> hAf(MMAX * K);
        const size_t nks = K / 16;
        for (size_t rb = 0; rb < MMAX / 32; ++rb)
            for (size_t ks = 0; ks < nks; ++ks)
                for (size_t l = 0; l < 64; ++l)
                    for (size_t i = 0; i < 8; ++i) hAf[((rb * nks + ks) * 64 + l) * 8 + i] = hA[(rb * 32 + (l & 31)) * K + ks * 16 + 8 * (l >> 5) + i];
        hipMalloc(&dAf, MMAX * K * 2);
        hipMemcpy(dAf, hAf.data(), hAf.size() * 2, hipMemcpyHostToDevice);
    }
    hipMemcpy(dR, hR.data(), hR.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    {   // the clock-limited MFMA ceiling: 512 workgroups x 4 waves x iters x 32 MFMAs
        const int iters = 2000;
        const float us = time_us([&] { hipLaunchKernelGGL(bare_k, dim3(2 * NCU), dim3(256), 0, 0, dC2, iters, dA); });
        const double fl = 2.0 * NCU * 4 * (double)iters * 32 * 32768.0;
        printf("bare v_mfma_f32_32x32x16_f16 stream, random operands, 2 waves per SIMD: %.1f us -> %.0f TFLOP/s = %.3f of 2.5 PF (clock-limited ceiling)\n", us, fl / us * 1e-6,
               fl / us * 1e-6 / 2500.0);
    }
    const int only = argc > 1 ? atoi(argv[1]) : 0;
    const int mode = argc > 2 ? atoi(argv[2]) : 0;
    if (argc > 3) return power_loop(atoi(argv[3]));
    for (int pass = 0; pass < 2; ++pass) {
        printf("---- pass %d\n", pass);
        for (int M : {25088, 12544, 6272}) {
            if (only && M != only) continue;
            const bool chk = pass == 0;
            {
                HArgs g{dA, dW, dB, dR, dC, M, 1536, 1536};
                const int grid = ((M + 127) / 128) * 12;
                report("gemm_hd_k<false> as in the library (128 x 128 x 64, 2 stages, drain + barrier per k-tile)", M,
                       time_us([&] { hipLaunchKernelGGL(hd_k, dim3(grid), dim3(256), 0, 0, g); }));
            }
            if (mode == 0) {
            // ---- the same tile, deeper ring, counted vmcnt, more workgroups per CU
            run_hr<128, 128, 64, 2, 2, 2, 2, 0>("", M, chk);
            run_hr<128, 128, 32, 3, 2, 2, 3, 0>("", M, chk);
            run_hr<128, 128, 32, 4, 2, 2, 2, 0>("", M, chk);
            run_hr<128, 128, 32, 2, 2, 2, 4, 0>("", M, chk);
            run_hr<128, 128, 64, 3, 2, 2, 1, 0>("", M, chk);
            // ---- 256 x 128: 0.75 of the operand bytes per MFMA
            run_hr<256, 128, 32, 3, 2, 2, 2, 0>("", M, chk);
            run_hr<256, 128, 32, 3, 4, 2, 2, 0>("", M, chk);
            run_hr<256, 128, 32, 3, 2, 2, 2, 16>(" + setprio", M, false);
            run_hr<256, 128, 32, 3, 2, 2, 2, 8>(" + odd workgroups late", M, false);
            run_hr<128, 256, 32, 3, 2, 2, 2, 0>("", M, chk);
            // ---- 256 x 256: half the operand bytes per MFMA, one workgroup per CU
            run_hr<256, 256, 32, 4, 2, 4, 1, 0>("", M, chk);
            run_hr<256, 256, 64, 2, 2, 4, 1, 0>("", M, chk);
            run_hr<256, 256, 32, 4, 2, 4, 1, 16>(" + setprio", M, false);
            if (pass == 0) {
                // ---- the stages of the VERDICT's gate, on the candidates
                run_hr<128, 128, 32, 3, 2, 2, 3, 1>("  (no epilogue traffic)", M, false);
                run_hr<128, 128, 32, 3, 2, 2, 3, 3>("  (no DMA in the loop, no epilogue traffic)", M, false);
                run_hr<128, 128, 32, 3, 2, 2, 3, 7>("  (MFMAs + fragment reads only: no DMA, no barrier, no epilogue traffic)", M, false);
                run_hr<256, 128, 32, 3, 2, 2, 2, 1>("  (no epilogue traffic)", M, false);
                run_hr<256, 128, 32, 3, 2, 2, 2, 3>("  (no DMA in the loop, no epilogue traffic)", M, false);
                run_hr<256, 128, 32, 3, 2, 2, 2, 7>("  (MFMAs + fragment reads only)", M, false);
                run_hr<256, 256, 32, 4, 2, 4, 1, 1>("  (no epilogue traffic)", M, false);
                run_hr<256, 256, 32, 4, 2, 4, 1, 3>("  (no DMA in the loop, no epilogue traffic)", M, false);
                run_hr<256, 256, 32, 4, 2, 4, 1, 7>("  (MFMAs + fragment reads only)", M, false);
            }
            } else if (mode == 3) {
            // ---- fourth experiment set: A fragments straight from a fragment-major plane into registers, only W through LDS
            run_hr<128, 128, 64, 2, 2, 2, 2, 0>("", M, false);
            run_ha<128, 128, 3, 2, 0>("", M, chk);
            run_ha<128, 128, 3, 3, 0>("", M, chk);
            run_ha<128, 128, 3, 4, 0>("", M, chk);
            run_ha<128, 256, 3, 2, 0>("", M, chk);
            run_ha<256, 128, 3, 2, 0>("", M, chk);
            run_ha<256, 256, 3, 1, 0>("", M, chk);
            run_ha<128, 128, 3, 3, 1>("  (no epilogue traffic)", M, false);
            run_ha<128, 256, 3, 2, 1>("  (no epilogue traffic)", M, false);
            run_hr<128, 128, 64, 2, 2, 2, 2, 1>("  (no epilogue traffic)", M, false);
            } else if (mode == 2) {
            // ---- third experiment set: smaller tiles for the launches of a sample group (6272 / 12544 rows: 2.3 / 4.6 tiles of 128 x 128 per CU)
            run_hr<128, 128, 64, 2, 2, 2, 2, 0>("", M, false);
            run_hr<64, 128, 64, 2, 2, 2, 3, 0>("", M, chk);
            run_hr<64, 128, 32, 3, 2, 2, 4, 0>("", M, chk);
            run_hr<64, 128, 32, 2, 2, 2, 5, 0>("", M, chk);
            run_hr<128, 64, 64, 2, 2, 2, 3, 0>("", M, chk);
            run_hr<128, 64, 32, 3, 2, 2, 4, 0>("", M, chk);
            run_hr<64, 256, 32, 3, 2, 2, 3, 0>("", M, chk);
            run_hr<64, 256, 64, 2, 2, 2, 2, 0>("", M, chk);
            run_hr<128, 192, 32, 3, 2, 2, 2, 0>("", M, chk);
            run_hr<64, 192, 32, 3, 2, 2, 3, 0>("", M, chk);
            } else {
            // ---- second experiment set: DMA pieces between the MFMAs; R + bias as the accumulators' start value
            run_hr<256, 128, 32, 3, 2, 2, 2, 0>("", M, false);
            run_hr<256, 128, 32, 3, 2, 2, 2, 32>(" + DMA between MFMAs", M, chk);
            run_hr<256, 128, 32, 3, 2, 2, 2, 33>(" + DMA between MFMAs (no epilogue traffic)", M, false);
            run_hr<256, 128, 32, 3, 2, 2, 2, 64>(" + acc starts as R + b", M, chk);
            run_hr<256, 128, 32, 3, 2, 2, 2, 96>(" + both", M, chk);
            run_hr<256, 128, 32, 3, 2, 2, 2, 96 + 8 + 128>(" + both + slot-B workgroups 45 us late", M, false);
            run_hr<256, 128, 32, 3, 4, 2, 2, 32>(" + DMA between MFMAs", M, chk);
            run_hr<256, 128, 32, 3, 4, 2, 2, 96>(" + both", M, chk);
            run_hr<128, 128, 32, 3, 2, 2, 3, 32>(" + DMA between MFMAs", M, chk);
            run_hr<128, 128, 32, 3, 2, 2, 3, 96>(" + both", M, chk);
            run_hr<128, 128, 64, 2, 2, 2, 2, 64>(" + acc starts as R + b", M, chk);
            run_hr<128, 128, 64, 2, 2, 2, 2, 96>(" + both", M, chk);
            run_hr<256, 256, 32, 4, 2, 4, 1, 32>(" + DMA between MFMAs", M, chk);
            run_hr<256, 256, 32, 4, 2, 4, 1, 33>(" + DMA between MFMAs (no epilogue traffic)", M, false);
            run_hr<256, 256, 32, 4, 2, 4, 1, 96>(" + both", M, chk);
            run_hr<128, 256, 32, 3, 2, 2, 2, 96>(" + both", M, chk);
            }
        }
    }
    return 0;
}

static int power_loop(int which) {
    const int M = 25088;
    HArgs g{dA, dW, dB, dR, dC2, M, 1536, 1536};
    auto k2 = hr_k<256, 128, 32, 3, 2, 2, 2, 1>;
    auto k3 = hr_k<256, 128, 32, 3, 2, 2, 2, 7>;
    hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    hipFuncSetAttribute((const void*)k3, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    int n = 0;
    for (;; ++n) {
        if (which == 0) hipLaunchKernelGGL(bare_k, dim3(2 * NCU), dim3(256), 0, 0, dC2, 2000, dA);
        else if (which == 1) hipLaunchKernelGGL(hd_k, dim3(196 * 12), dim3(256), 0, 0, g);
        else if (which == 2) hipLaunchKernelGGL(k2, dim3(98 * 12), dim3(256), 72 * 1024, 0, g);
        else hipLaunchKernelGGL(k3, dim3(98 * 12), dim3(256), 72 * 1024, 0, g);
        if (n % 64 == 63) {
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms > 8000.f) { printf("power loop %d: %d launches in %.0f ms = %.1f us each\n", which, n + 1, ms, ms * 1000.f / (n + 1)); break; }
        }
    }
    return 0;
}
