// Is an fp32-input MFMA a sequential fmaf chain over its k values, in lane-group order, for BOTH shapes?  (round 6: what a 16-token gate_k on
// v_mfma_f32_16x16x4_f32 would need to reproduce the scores of the 32-token kernel on v_mfma_f32_32x32x2_f32 bit for bit.)
//   D32[i][j] (32 x 32 x 2): lanes 0-31 supply k = 0, lanes 32-63 k = 1
//   D16[i][j] (16 x 16 x 4): lane group g = lane >> 4 supplies k = g
// One wave computes sum_k A[i][k] B[k][j] over K = 64 values per (i, j) three ways -- 32x32x2 over pairs (k, k + 1), 16x16x4 over quads, and a scalar
// fmaf chain in k order -- on random data with a wide exponent spread; the host compares bits.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_order_probe.hip -o tools/_bin/mfma_order_probe && tools/_bin/mfma_order_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int K = 64;

// A [32][K], B [K][32] row-major; out32 [32][32] (D[i][j]), out16 [16][16], outs [32][32]
__global__ void probe_k(const float* __restrict__ A, const float* __restrict__ B, float* out32, float* out16, float* outs) {
    const int lane = threadIdx.x;
    {   // 32x32x2: A operand lane -> row i = lane & 31, k = lane >> 5; B operand lane -> col j = lane & 31, k = lane >> 5
        f32x16 acc;
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int k0 = 0; k0 < K; k0 += 2) {
            const float a = A[(lane & 31) * K + k0 + (lane >> 5)], b = B[(k0 + (lane >> 5)) * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        for (int r = 0; r < 16; ++r) out32[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[r];
    }
    {   // 16x16x4: A lane -> row i = lane & 15, k = lane >> 4; B lane -> col j = lane & 15, k = lane >> 4; D lane -> rows 4 (lane >> 4) + r, col lane & 15
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < K; k0 += 4) {
            const float a = A[(lane & 15) * K + k0 + (lane >> 4)], b = B[(k0 + (lane >> 4)) * 32 + (lane & 15)];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        }
        for (int r = 0; r < 4; ++r) out16[(4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[r];
    }
    // scalar chain
    for (int e = lane; e < 32 * 32; e += 64) {
        const int i = e >> 5, j = e & 31;
        float s = 0.f;
        for (int k = 0; k < K; ++k) s = __builtin_fmaf(A[i * K + k], B[k * 32 + j], s);
        outs[e] = s;
    }
}

int main() {
    std::vector<float> hA(32 * K), hB(K * 32);
    srand(7);
    auto rnd = [] { return (float)((rand() / (double)RAND_MAX - 0.5) * std::exp2((double)(rand() % 24 - 12))); };
    for (auto& x : hA) x = rnd();
    for (auto& x : hB) x = rnd();
    float *dA, *dB, *d32, *d16, *ds;
    hipMalloc(&dA, hA.size() * 4); hipMalloc(&dB, hB.size() * 4); hipMalloc(&d32, 4096); hipMalloc(&d16, 1024); hipMalloc(&ds, 4096);
    hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe_k, dim3(1), dim3(64), 0, 0, dA, dB, d32, d16, ds);
    std::vector<float> o32(1024), o16(256), os(1024);
    hipMemcpy(o32.data(), d32, 4096, hipMemcpyDeviceToHost);
    hipMemcpy(o16.data(), d16, 1024, hipMemcpyDeviceToHost);
    hipMemcpy(os.data(), ds, 4096, hipMemcpyDeviceToHost);
    int d_32_s = 0, d_16_s = 0, d_16_32 = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) d_32_s += memcmp(&o32[i * 32 + j], &os[i * 32 + j], 4) != 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            d_16_s += memcmp(&o16[i * 16 + j], &os[i * 32 + j], 4) != 0;
            d_16_32 += memcmp(&o16[i * 16 + j], &o32[i * 32 + j], 4) != 0;
        }
    printf("v_mfma_f32_32x32x2_f32 vs the scalar fmaf chain in k order: %d of 1024 words differ\n", d_32_s);
    printf("v_mfma_f32_16x16x4_f32 vs the scalar fmaf chain in k order: %d of 256 words differ\n", d_16_s);
    printf("16x16x4 vs 32x32x2 on the shared 16 x 16 block:             %d of 256 words differ\n", d_16_32);
    return 0;
}
