// Effective shader clock seen by a SHORT, partly-filled launch (the small-batch regime): every wave runs one dependent
// chain of `n` v_mfma_f32_32x32x2_f32 (64 cycles each, nothing else), `wgs` workgroups of 4 waves, launched back to back.
//   hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o tools/_bin/clock_probe && tools/_bin/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void chain(float* out, int n, float a, float b) {
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = (float)threadIdx.x;
    for (int i = 0; i < n; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += acc[q];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int cfg[][2] = {{168, 768}, {256, 768}, {1024, 768}, {168, 7680}, {1024, 7680}, {168, 76800}};
    for (auto& c : cfg) {
        const int reps = c[1] > 10000 ? 20 : 200;
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(chain, dim3(c[0]), dim3(256), 0, 0, out, c[1], 1.0f, 0.5f);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(chain, dim3(c[0]), dim3(256), 0, 0, out, c[1], 1.0f, 0.5f);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / reps, cyc = 64.0 * c[1];
        printf("wgs=%4d chain=%6d MFMAs: %.2f us per launch -> >= %.2f GHz if the chain were all of it (%.1f us at 2.4 GHz)\n", c[0], c[1], us, cyc / us / 1e3, cyc / 2400.0);
    }
    return 0;
}
