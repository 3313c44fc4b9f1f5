// Lab (not part of the library): does the [token][4L] layout of `mf` camp on a subset of the L2 / HBM channels when temporal_k reads
// only the key (or key | value) quarter of every 2-KB token row at a 24-KB row stride?  Reads the same bytes three ways:
//   A  key quarter of [b][t][h][4L] rows (today's layout), one workgroup per (b, h), rows t = 0..T-1
//   B  the same volume from a contiguous [b][h][t][L] array
//   C  key | value halves (1 KB) of today's rows;  D  the same from a contiguous [b][h][t][2L] array
// build: hipcc --offload-arch=gfx950 -O3 tools/strided_read.hip -o gpurun_out/strided_read ; run: gpurun_out/strided_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int W>   // W floats per row read (128 or 256)
__global__ __launch_bounds__(256) void rd(const float* __restrict__ base, long row_stride, long wg_stride_b, long wg_stride_h, int H, int T, int off,
                                          float* __restrict__ out) {
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const float* p = base + b * wg_stride_b + h * wg_stride_h + off;
    constexpr int C4 = W / 4, NSL = 256 / C4;
    const int c4 = (threadIdx.x % C4) * 4, sl = threadIdx.x / C4;
    f32x4 s = {0, 0, 0, 0};
    for (int t0 = sl; t0 < T; t0 += NSL * 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int t = t0 + u * NSL < T ? t0 + u * NSL : T - 1; v[u] = *reinterpret_cast<const f32x4*>(p + t * row_stride + c4); }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    if (s[0] + s[1] + s[2] + s[3] == 12345.f) out[blockIdx.x] = s[0];
}
int main() {
    const int B2 = 128, T = 196, H = 12, L = 128;
    const long n = (long)B2 * T * H * 4 * L;
    float *mf, *out;
    hipMalloc(&mf, n * 4); hipMalloc(&out, 1 << 20);
    hipMemset(mf, 0, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char* name, auto launch, double mb) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-58s %8.1f us  %6.2f TB/s\n", name, ms / 20 * 1e3, mb / (ms / 20) / 1e3);
    };
    const double mbK = (double)B2 * T * H * L * 4 / 1e6;
    dim3 g(B2 * H), blk(256);
    time("A key quarter of [b][t][h][4L] rows (24-KB row stride)", [&] { hipLaunchKernelGGL(rd<128>, g, blk, 0, 0, mf, (long)H * 4 * L, (long)T * H * 4 * L, (long)4 * L, H, T, L, out); }, mbK);
    time("B the same bytes, contiguous [b][h][t][L]", [&] { hipLaunchKernelGGL(rd<128>, g, blk, 0, 0, mf, (long)L, (long)H * T * L, (long)T * L, H, T, 0, out); }, mbK);
    time("C key|value half of [b][t][h][4L] rows", [&] { hipLaunchKernelGGL(rd<256>, g, blk, 0, 0, mf, (long)H * 4 * L, (long)T * H * 4 * L, (long)4 * L, H, T, L, out); }, 2 * mbK);
    time("D the same bytes, contiguous [b][h][t][2L]", [&] { hipLaunchKernelGGL(rd<256>, g, blk, 0, 0, mf, (long)2 * L, (long)H * T * 2 * L, (long)T * 2 * L, H, T, 0, out); }, 2 * mbK);
    time("E whole [b][t][h][4L] rows by (b, h) workgroups", [&] { hipLaunchKernelGGL(rd<256>, g, blk, 0, 0, mf, (long)H * 4 * L, (long)T * H * 4 * L, (long)4 * L, H, T, 0, out);
                                                                  hipLaunchKernelGGL(rd<256>, g, blk, 0, 0, mf, (long)H * 4 * L, (long)T * H * 4 * L, (long)4 * L, H, T, 2 * L, out); }, 4 * mbK);
    return 0;
}
