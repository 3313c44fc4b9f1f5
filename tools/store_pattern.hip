// HBM write (and read) rate of the C^T-fragment access pattern of the MFMA kernels' epilogues vs full-line accesses (round 3).
//   pattern 0  "fragment": a wave instruction covers 32 rows x 2 pieces of 16 B (lane l -> row l & 31, byte offset 16 (l >> 5) + 32 q for the
//              q-th of 4 instructions): every 128-byte line of a row is written by 4 instructions x 2 lanes   (what gemm_* / projqkv_k do)
//   pattern 1  "full lines": a wave instruction covers 8 rows x 128 contiguous bytes (lane l -> row l >> 3, 16-byte piece l & 7)
// Both write (or read) the same [rows][ld] fp32 matrix, 32-column chunks like the chained kernels (ld = 512: mf; 1536: h).
//   hipcc --offload-arch=gfx950 -O3 tools/store_pattern.hip -o /tmp/store_pattern && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PATTERN, bool READ>
__global__ __launch_bounds__(256) void k(float* __restrict__ p, long rows, int ld, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunk = ld / 32;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long t = blockIdx.x; t < rows / 128; t += gridDim.x) {
        const long row0 = t * 128 + wave * 32;
        for (int c = 0; c < nchunk; ++c) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float* a;
                if (PATTERN == 0) a = p + (row0 + (lane & 31)) * ld + c * 32 + 8 * q + 4 * (lane >> 5);
                else a = p + (row0 + 8 * q + (lane >> 3)) * ld + c * 32 + 4 * (lane & 7);
                if (READ) acc += *reinterpret_cast<const f32x4*>(a);
                else *reinterpret_cast<f32x4*>(a) = f32x4{(float)c, (float)q, (float)lane, 1.f};
            }
        }
    }
    if (READ && acc[0] == 123.456f) *sink = acc[1];
}

template <int PATTERN, bool READ>
void run(const char* name, float* p, long rows, int ld, float* sink, int grid) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<PATTERN, READ>), dim3(grid), dim3(256), 0, 0, p, rows, ld, sink);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k<PATTERN, READ>), dim3(grid), dim3(256), 0, 0, p, rows, ld, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 10;
    printf("%-44s rows %7ld ld %5d grid %5d: %8.1f us  %6.2f TB/s\n", name, rows, ld, grid, ms * 1e3, rows * (double)ld * 4 / ms / 1e9);
}

int main() {
    const long rows = 301056;      // tokens of the B=64 step
    float *p, *sink;
    hipMalloc(&p, rows * 1536 * 4); hipMalloc(&sink, 4);
    hipMemset(p, 0, rows * 1536 * 4);
    for (int grid : {512, 2048}) {
        for (int ld : {512, 1536}) {
            const long r = ld == 512 ? rows : rows / 12 * 4;      // ~0.6 GB each
            run<0, false>("write, fragment pattern (32 B pieces)", p, r, ld, sink, grid);
            run<1, false>("write, full 128-byte lines", p, r, ld, sink, grid);
            run<0, true>("read,  fragment pattern (32 B pieces)", p, r, ld, sink, grid);
            run<1, true>("read,  full 128-byte lines", p, r, ld, sink, grid);
        }
    }
    return 0;
}
