// Can the fp32 VALU pipe add FLOPs beside a saturated fp32 MFMA pipe?  (round 3; MI355X_MICROARCH.md: "a MFMA-only wave and a VALU-only
// wave on the same CU run concurrently".)  One 512-thread workgroup per CU: waves 0-3 (one per SIMD) loop v_mfma_f32_32x32x2_f32 on 4
// accumulators, waves 4-7 loop v_pk_fma_f32 on 16 independent register pairs; modes: MFMA only / VALU only / both.  Registers only.
//   hipcc --offload-arch=gfx950 -O3 tools/dual_pipe.hip -o /tmp/dual_pipe && /tmp/dual_pipe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>      // 1 MFMA waves work, 2 VALU waves work, 3 both
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, float a, float b) {
    const int wave = threadIdx.x >> 6;
    float s = 0.f;
    if (wave < 4) {
        if (MODE & 1) {
            f32x16 acc[4];
            for (int c = 0; c < 4; ++c) for (int q = 0; q < 16; ++q) acc[c][q] = (float)(threadIdx.x + c);
            for (int i = 0; i < iters; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
            for (int c = 0; c < 4; ++c) for (int q = 0; q < 16; ++q) s += acc[c][q];
        }
    } else if (MODE & 2) {
        f32x2 v[16];
        for (int c = 0; c < 16; ++c) v[c] = f32x2{(float)threadIdx.x + c, 1.f};
        const f32x2 x = {a, b}, y = {b, a};
        // per MFMA iteration above (4 MFMAs x 64 cycles = 256 cycles) a VALU wave can issue 256 / 4 = 64 packed FMAs at best
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 16; ++c) v[c] = __builtin_elementwise_fma(v[c], x, y);
        for (int c = 0; c < 16; ++c) s += v[c][0] + v[c][1];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters, 1.0f, 0.5f);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters, 1.0f, 0.5f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 10;
    const double mf = (MODE & 1) ? 256.0 * 4 * iters * 4 * 4096.0 : 0.0;                 // MFMA flops: 4 waves x iters x 4 MFMAs x 2*32*32*2
    const double vf = (MODE & 2) ? 256.0 * 4 * 64 * (double)iters * 64 * 2 * 2 : 0.0;   // VALU flops: 4 waves x 64 lanes x iters x 64 pk_fma x 2 lanes x 2
    printf("%-28s %8.3f ms   MFMA %6.1f TFLOP/s   VALU %6.1f TFLOP/s   total %6.1f\n", name, ms, mf / ms / 1e9, vf / ms / 1e9, (mf + vf) / ms / 1e9);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    const int iters = 1 << 14;
    for (int pass = 0; pass < 2; ++pass) {
        run<1>("MFMA waves only", out, iters);
        run<2>("VALU waves only", out, iters);
        run<3>("both", out, iters);
    }
    return 0;
}
