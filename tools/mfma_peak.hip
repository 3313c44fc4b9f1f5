// Sustained fp32 MFMA rate of the part under load: the ceiling the GEMM-shaped kernels are priced against.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
// Each wave issues `iters` x CH v_mfma_f32_32x32x2_f32 on CH independent accumulators (no memory traffic at all).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CH>
__global__ __launch_bounds__(256, 2) void mfma_loop(float* out, int iters, float a, float b) {
    f32x16 acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[c][q] = (float)(threadIdx.x + c);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int q = 0; q < 16; ++q) s += acc[c][q];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// The same loop with operands that CHANGE from one MFMA to the next (8 random register pairs cycled), as in a real GEMM:
// the datapath toggles, the part draws more power and clocks down -- the rate a random-data fp32 GEMM is capped at.
template <int CH>
__global__ __launch_bounds__(256, 2) void mfma_loop_rnd(float* out, int iters, const float* __restrict__ vals) {
    f32x16 acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[c][q] = 0.f;
    float a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = vals[(threadIdx.x * 8 + j) & 4095]; b[j] = vals[4096 + ((threadIdx.x * 8 + j) & 4095)]; }
    for (int i = 0; i < iters; i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[(j + c) & 7], acc[c], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int q = 0; q < 16; ++q) s += acc[c][q];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CH>
void run_rnd(const char* name, int blocks, int iters, float* out, const float* vals, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop_rnd<CH>, dim3(blocks), dim3(256), 0, 0, out, iters, vals);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(mfma_loop_rnd<CH>, dim3(blocks), dim3(256), 0, 0, out, iters, vals);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double flops = (double)blocks * 4 * (double)iters * CH * 4096.0;
    printf("%-44s blocks %5d: %8.3f ms  %7.1f TFLOP/s  (%.1f %% of 157.3)\n", name, blocks, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
}

template <int CH>
void run(const char* name, int blocks, int iters, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<CH>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(mfma_loop<CH>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    const double flops = (double)blocks * 4 * (double)iters * CH * 4096.0;
    printf("%-44s blocks %5d: %8.3f ms  %7.1f TFLOP/s  (%.1f %% of 157.3)\n", name, blocks, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
}

int main() {
    float* out;
    hipMalloc(&out, 4096 * 256 * sizeof(float));
    const int iters = 1 << 15;
    run<1>("1 dependent chain, 2 waves/SIMD", 512, iters, out);
    run<4>("4 independent chains, 2 waves/SIMD", 512, iters / 4, out);
    run<4>("4 independent chains, 1 wave/SIMD", 256, iters / 4, out);
    run<1>("1 dependent chain, 1 wave/SIMD", 256, iters, out);
    run<4>("4 chains, 2 waves/SIMD, 4.59 rounds of blocks", 2352, iters / 16, out);
    // sustained: ~2 s of back-to-back launches, then measure again (clock under sustained MFMA load)
    for (int r = 0; r < 40; ++r) hipLaunchKernelGGL(mfma_loop<4>, dim3(512), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
    run<4>("4 independent chains, 2 waves/SIMD, after 2 s load", 512, iters / 4, out);
    // operands that change every MFMA (uniform random in [-1, 1), N(0,1)-like magnitudes): the power-capped rate
    std::vector<float> hv(8192);
    srand(7);
    for (auto& v : hv) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    float* vals;
    hipMalloc(&vals, hv.size() * sizeof(float));
    hipMemcpy(vals, hv.data(), hv.size() * sizeof(float), hipMemcpyHostToDevice);
    run_rnd<4>("4 chains, 2 waves/SIMD, CHANGING random operands", 512, iters / 4, out, vals, 5);
    run_rnd<4>("  the same, sustained (40 launches = 2 s)", 512, iters / 4, out, vals, 40);
    run_rnd<4>("4 chains, 1 wave/SIMD, CHANGING random operands", 256, iters / 4, out, vals, 10);
    run<4>("4 chains, 2 waves/SIMD, constant operands again", 512, iters / 4, out);
    hipFree(vals);
    hipFree(out);
    return 0;
}
