// Where does the fp32 MFMA GEMM lose its ~20 %?  Standalone timing lab (NOT part of the library): the wave-private
// pipeline kernel of mc_gemm.hip with single ingredients removed.  Ablated variants compute WRONG results by
// construction -- they exist only here, only to be timed.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_lab.hip -o /tmp/gemm_lab && /tmp/gemm_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int NX = 8;
    int q = nwg / NX, r = nwg % NX, xcd = bid % NX, idx = bid / NX;
    return ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
__device__ __forceinline__ void dma16(unsigned voff, const float* sbase, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte) : "memory");
}
constexpr int WBK = 16, WSTAGE = 128 * WBK;
enum { F_DMA = 1, F_READ = 2, F_SPREAD = 4, F_XCD = 8 };   // ingredients present / options

template <int FL, int WPS>   // WPS: workgroups per CU the launch bounds allow (2: 2 waves/SIMD)
__global__ __launch_bounds__(256, WPS) void wp_k(const float* A, const float* W, float* C, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) float smem[4 * 2 * WSTAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int ntn = N / 128;
    const int bid = (FL & F_XCD) ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int tm = bid / ntn, tn = bid % ntn, row0 = tm * 128;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    float* wbase = smem + wave_u * 2 * WSTAGE;
    const unsigned lds0 = (unsigned)(size_t)wbase;
    const int dr = lane >> 2, dc = ((lane & 3) ^ ((dr >> 2) & 3)) * 4;
    unsigned voa[4], vow[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        voa[q] = (unsigned)(((long)(row0 + wm * 64 + 16 * q + dr) * K + dc) * 4);
        vow[q] = (unsigned)(((long)(tn * 128 + wn * 64 + 16 * q + dr) * K + dc) * 4);
    }
    auto issue_q = [&](int kt, int st, int q) {
        dma16(voa[q], A + kt * WBK, lds0 + st * WSTAGE * 4 + q * 1024);
        dma16(vow[q], W + kt * WBK, lds0 + st * WSTAGE * 4 + 4096 + q * 1024);
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    const int frow = lane & 31, hf = lane >> 5, sw = (frow >> 2) & 3;
    struct Frag { f32x4 a0, a1, b0, b1; };
    auto ld_frag = [&](int st, int j) {
        const float* S = wbase + st * WSTAGE + frow * WBK + ((2 * j + hf) ^ sw) * 4;
        Frag f;
        f.a0 = *reinterpret_cast<const f32x4*>(S);
        f.a1 = *reinterpret_cast<const f32x4*>(S + 32 * WBK);
        f.b0 = *reinterpret_cast<const f32x4*>(S + 64 * WBK);
        f.b1 = *reinterpret_cast<const f32x4*>(S + 96 * WBK);
        return f;
    };
    auto mma2 = [&](const Frag& f, int i0) {
#pragma unroll
        for (int i = i0; i < i0 + 2; ++i) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.b0[i], f.a0[i], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.b1[i], f.a0[i], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.b0[i], f.a1[i], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.b1[i], f.a1[i], acc[1][1], 0, 0, 0);
        }
    };
    const int nk = K / WBK;
#pragma unroll
    for (int q = 0; q < 4; ++q) issue_q(0, 0, q);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    Frag f0 = ld_frag(0, 0), f1 = ld_frag(0, 1);
    for (int kt = 0; kt < nk; ++kt) {
        const int st = kt & 1;
        const bool more = kt + 1 < nk;
        if constexpr ((FL & F_DMA) && !(FL & F_SPREAD)) {
            if (more) {
#pragma unroll
                for (int q = 0; q < 4; ++q) issue_q(kt + 1, st ^ 1, q);
            }
        }
        if constexpr (FL & F_READ) f1 = ld_frag(st, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma2(f0, 0);
        if constexpr ((FL & F_DMA) && (FL & F_SPREAD)) { if (more) { issue_q(kt + 1, st ^ 1, 0); issue_q(kt + 1, st ^ 1, 1); } __builtin_amdgcn_sched_barrier(0); }
        mma2(f0, 2);
        if constexpr ((FL & F_DMA) && (FL & F_SPREAD)) { if (more) { issue_q(kt + 1, st ^ 1, 2); issue_q(kt + 1, st ^ 1, 3); } }
        __builtin_amdgcn_sched_barrier(0);
        mma2(f1, 0);
        if (more) {
            if constexpr (FL & F_DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (FL & F_READ) f0 = ld_frag(st ^ 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma2(f1, 2);
    }
    const int m0 = row0 + wm * 64 + frow;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = tn * 128 + wn * 64 + ni * 32 + 8 * q + 4 * hf;
                f32x4 v = {acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]};
                *reinterpret_cast<f32x4*>(C + (long)(m0 + mi * 32) * N + n) = v;
            }
}

template <int FL, int WPS>
void run(const char* name, const float* A, const float* W, float* C, int M, int N, int K) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid((M / 128) * (N / 128));
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((wp_k<FL, WPS>), grid, dim3(256), 0, 0, A, W, C, M, N, K);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        for (int r = 0; r < 4; ++r) hipLaunchKernelGGL((wp_k<FL, WPS>), grid, dim3(256), 0, 0, A, W, C, M, N, K);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms / 4 < best) best = ms / 4;
    }
    const double tf = 2.0 * M * N * K / best / 1e9;
    printf("%-58s M=%6d tiles=%5d: %8.1f us  %6.1f TFLOP/s (%5.1f %% of 157.3)\n", name, M, grid.x, best * 1e3, tf, tf / 157.3 * 100);
}

int main() {
    const int N = 1536, K = 1536, MMAX = 25088;
    float *A, *W, *C;
    hipMalloc(&A, (size_t)MMAX * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&C, (size_t)MMAX * N * 4);
    std::vector<float> h((size_t)MMAX * K);
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    for (int M : {16384, 25088}) {     // 1536 tiles = 3 exact rounds of 512 slots; 2352 tiles = 4.59 rounds (the FiLM GEMM)
        run<F_DMA | F_READ | F_XCD, 2>("full (DMA at loop top)", A, W, C, M, N, K);
        run<F_DMA | F_READ | F_XCD | F_SPREAD, 2>("full, DMA spread between the MFMA groups", A, W, C, M, N, K);
        run<F_DMA | F_READ, 2>("full, no XCD remap", A, W, C, M, N, K);
        run<F_READ | F_XCD, 2>("no DMA in the loop (stale operands)", A, W, C, M, N, K);
        run<F_DMA | F_XCD, 2>("no LDS fragment reads in the loop", A, W, C, M, N, K);
        run<F_XCD, 2>("MFMA only", A, W, C, M, N, K);
        run<F_DMA | F_READ | F_XCD, 1>("full, 1 workgroup per CU (1 wave/SIMD)", A, W, C, M, N, K);
        run<F_XCD, 1>("MFMA only, 1 workgroup per CU", A, W, C, M, N, K);
    }
    return 0;
}
