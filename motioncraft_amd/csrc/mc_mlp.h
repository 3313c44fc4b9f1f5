// Fused two-layer MLP launcher (mc_mlp.hip).
#pragma once
#include <hip/hip_runtime.h>

enum { MLP_EXPERT = 0, MLP_PARTS = 1 };

struct MlpArgs {
    const float* X = nullptr;   // row r of group g at X[g*x_gstride + row(r)*ldx + 0..L)
    long ldx = 0, x_gstride = 0;
    const float* W1 = nullptr;  // [groups][hidden][L]
    const float* b1 = nullptr;  // [groups][hidden]
    const float* W2t = nullptr; // [groups][L][hidden]   (output-major, hidden contiguous)
    const float* b2 = nullptr;  // [groups][L]
    float* Y = nullptr;         // row r at Y[g*y_gstride + drow(r)*ldy + 0..L)
    long ldy = 0, y_gstride = 0;
    int M = 0, L = 0, hidden = 0;
    // MLP_EXPERT: device tile map + gather/scatter lists (mc_route.hip)
    const int* tile_group = nullptr;
    const int* tile_row0 = nullptr;
    const int* tile_nrows = nullptr;
    const int* num_tiles = nullptr;
    const int* src_row = nullptr;
    const int* dst_row = nullptr;
};

bool mc_mlp_supported(int L, int hidden);
int mc_launch_mlp(int mode, const MlpArgs& g, int groups, int max_tiles, hipStream_t s);
