// Round-5 lab for the fp16 plane GEMM of the reduced-precision modes (NOT part of the library): what bounds gemm_hd_k<false>
// on the FiLM shape (M x 1536 x 1536, fp16 planes of A and W, fp32 bias + residual epilogue), and candidate structures.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_h_lab.hip -o tools/_bin/gemm_h_lab && tools/_bin/gemm_h_lab
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
// the library's gemm_hd_k<false> (128 x 128 x 64, 4 waves x 64 x 64, two LDS-DMA stages, vmcnt(0) + barrier per k-tile) with
// diagnostic cuts: DIAG & 1: no residual read, (almost) no store; & 2: every workgroup stages tile (0, 0) (operands hot in
// L2 / no HBM); & 4: no DMA inside the loop; & 8: fragments read once, not per k-step; & 16: 128-byte rows swizzled by (row >> 1) & 7 instead of row & 7
// ---------------------------------------------------------------------------------------------------------------------
template <int DIAG>
__global__ __launch_bounds__(256, 2) void hd_k(HArgs g) {
    constexpr int BKH = 64, CH = 8, RPI = 8, PT = 128 * BKH, NQ = 4;
    __shared__ __attribute__((aligned(16))) half_t smem[2 * 2 * PT];
    auto tile = [&](int buf, int op) { return smem + (buf * 2 + op) * PT; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, frow = lane & 31, hf = lane >> 5;
    const int ntn = g.N / 128;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    int tm = bid / ntn, tn = bid % ntn;
    const int row0 = tm * 128, nrows = min(128, g.M - row0);
    if (DIAG & 2) { tm = 0; tn = 0; }
    const int dr = lane / CH, dpos = lane % CH;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    unsigned goa[NQ], gow[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int r = 32 * wave + q * RPI + dr;
        const int ar = min(tm * 128 + r, g.M - 1);
        const int sz = (DIAG & 16) ? ((r >> 1) & 7) : (r & 7);
        goa[q] = (unsigned)(((long)ar * g.K + (dpos ^ sz) * 8) * 2);
        gow[q] = (unsigned)(((long)(tn * 128 + r) * g.K + (dpos ^ sz) * 8) * 2);
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
    const int sw = (DIAG & 16) ? ((frow >> 1) & 7) : (frow & 7);
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f16x8 fa[2], fw[2];
    if (DIAG & 8) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            fa[i] = *reinterpret_cast<const f16x8*>(tile(0, 0) + (wm * 64 + frow + i * 32) * BKH + (hf ^ sw) * 8);
            fw[i] = *reinterpret_cast<const f16x8*>(tile(0, 1) + (wn * 64 + frow + i * 32) * BKH + (hf ^ sw) * 8);
        }
    }
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (!(DIAG & 4) && kt + 1 < nk) issue(kt + 1, buf ^ 1);
        const int ra = (wm * 64 + frow) * BKH, rw = (wn * 64 + frow) * BKH;
#pragma unroll
        for (int s = 0; s < BKH / 16; ++s) {
            const int pos = ((2 * s + hf) ^ sw) * 8;
            if (!(DIAG & 8)) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    fa[i] = *reinterpret_cast<const f16x8*>(tile(buf, 0) + ra + i * 32 * BKH + pos);
                    fw[i] = *reinterpret_cast<const f16x8*>(tile(buf, 1) + rw + i * 32 * BKH + pos);
                }
            } else {
                asm volatile("" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fw[0]), "+v"(fw[1]));
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
                const int n = (bid % ntn) * 128 + wn * 64 + ni * 32 + 8 * q + 4 * hf;
                f32x4 v = {acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]};
                v += *reinterpret_cast<const f32x4*>(g.bias + n);
                if (DIAG & 1) {
                    if (v[0] == 1234.56789f) *reinterpret_cast<f32x4*>(crow + n) = v;
                } else {
                    v += *reinterpret_cast<const f32x4*>(rrow + n);
                    *reinterpret_cast<f32x4*>(crow + n) = v;
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------------------------------
static half_t *dA, *dW;
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
static void report(const char* name, int M, int N, int K, float us) {
    printf("  %-74s M=%5d  %8.1f us  %7.1f TFLOP/s\n", name, M, us, 2.0 * M * N * K / us * 1e-6);
    fflush(stdout);
}
static bool compare(int M, int N, const char* what) {
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

template <int DIAG>
static void run_hd(const char* name, int M, float* C) {
    HArgs g{dA, dW, dB, dR, C, M, 1536, 1536};
    const int grid = ((M + 127) / 128) * 12;
    report(name, M, 1536, 1536, time_us([&] { hipLaunchKernelGGL(hd_k<DIAG>, dim3(grid), dim3(256), 0, 0, g); }));
}

#include "gemm_h_lab_cand.inc"

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
    hipMemcpy(dR, hR.data(), hR.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&dSlab, (size_t)NCU * 8 * 32 * 64 * 16); hipMalloc(&dSync, (2 + 2 * NCU) * 4);
    hipMemset(dSync, 0, (2 + 2 * NCU) * 4);
    hipMalloc(&dAl, MMAX * K * 2); hipMalloc(&dWl, N * K * 2);
    hipMemset(dAl, 0, MMAX * K * 2); hipMemset(dWl, 0, N * K * 2);
    const bool pmc = argc > 1 && !strcmp(argv[1], "pmc");
    if (pmc) {
        HArgs g{dA, dW, dB, dR, dC, 25088, 1536, 1536};
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(hd_k<0>, dim3(196 * 12), dim3(256), 0, 0, g);
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(hd_k<16>, dim3(196 * 12), dim3(256), 0, 0, g);
        pmc_candidates();
        hipDeviceSynchronize();
        return 0;
    }
    for (int pass = 0; pass < 2; ++pass) {
        printf("---- pass %d\n", pass);
        for (int M : {25088, 12544, 32768}) {
            run_hd<0>("gemm_hd_k<false> as in the library", M, dC);
            if (pass == 0) {
                run_hd<1>("  no residual read, no store", M, dC2);
                run_hd<2>("  every workgroup stages tile (0, 0)", M, dC2);
                run_hd<3>("  both", M, dC2);
                run_hd<4>("  no DMA in the loop", M, dC2);
                run_hd<5>("  no DMA in the loop, no epilogue traffic", M, dC2);
                run_hd<8>("  fragments read once (DMA on)", M, dC2);
                run_hd<13>("  MFMAs + barriers only", M, dC2);
            }
            run_hd<16>("  128-byte rows swizzled by (row >> 1) & 7", M, dC2);
            if (pass == 0) compare(M, 1536, "vs row & 7");
            run_candidates(M, pass == 0);
        }
    }
    return 0;
}
