// Argument block + launcher of the fp32 MFMA GEMM family (mc_gemm.hip).
#pragma once
#include <hip/hip_runtime.h>

enum { GM_PLAIN = 0, GM_EXP1 = 1, GM_EXP2 = 2, GM_COMB = 3, GM_ENC = 4 };

struct GemmArgs {
    // A operand: element (r,k) at A[grp*a_gstride + row(r)*lda + a_col + k]
    const float* A = nullptr;
    long lda = 0;
    int a_col = 0;
    long a_gstride = 0;
    // W operand [N][ldw], ldw % 4 == 0, rows zero padded to ldw
    const float* W = nullptr;
    long ldw = 0;
    long w_gstride = 0;
    const float* bias = nullptr;
    long b_gstride = 0;
    // output / residual: element (r,n) at C[grp*c_gstride + drow(r)*ldc + c_col + n]
    float* C = nullptr;
    long ldc = 0;
    int c_col = 0;
    long c_gstride = 0;
    const float* R = nullptr;
    long ldr = 0;
    long r_gstride = -1;         // group stride of R (-1: same as c_gstride)
    int act = 0;
    int act_after_res = 0;       // 0: act(acc + bias) + R    1: act(acc + bias + R)   (residual conv blocks)
    const float* add = nullptr;  // C += add[(r % add_mod) * ld_add + n]
    int add_mod = 1;
    long ld_add = 0;
    long dup_rows = 0;  // also write row r + dup_rows (CFG halves share the encoder output)
    int M = 0, N = 0, K = 0;
    // expert-grouped tile map (device arrays)
    const int* tile_group = nullptr;
    const int* tile_row0 = nullptr;
    const int* tile_nrows = nullptr;
    const int* num_tiles = nullptr;
    const int* src_row = nullptr;
    const int* dst_row = nullptr;
    const float* comb_w = nullptr;  // [M][2]
    // launch options (a context passes its own, mc_ctx_set_option; the defaults come from the environment once per process)
    int small_tile_n = 0;           // mc_launch_gemm_small: force the tile width (64, 48, 96); 0 = the load model's choice (env MC_SMALL_TILE_N)
    int wp_grid = 0;                // workgroups of the persistent gemm_wp_k launch; 0 = default 512 (env MC_GEMM_WP_GRID), < 0 one per tile
    int tune = -1;                  // -1 = process default (mc_gemm_default_tune(): env MC_GEMM_TUNE, else 1841 = 49 + 256 + 512 + 1024); bits: 0 mid-loop staging writes in gemm_k, 4 LDS-DMA kernels for full-tile plain launches, 5 (with 4) the persistent wave-private pipeline gemm_wp_k instead of gemm_dma_k, 6 no XCD remap in gemm_dma_k, 8 (round 4) XCD-aware tile order in gemm_small_k (a row block's column tiles share one L2), 9 (round 4) the aligned pose-encoder GEMM on gemm_wp_k (table + duplicate rows in its epilogue), 10 (round 5) the folded decoder tail as block ranges with A read once (gemm_tail2_k); bit 6 also switches the XCD remap of gemm_wp_k off
};

int mc_device_cus();      // compute units of the current device (cached)
int mc_launch_gemm(int mode, const GemmArgs& g, int groups, int max_tiles, hipStream_t stream);
// small-M plain GEMM (64 x 64 tiles, one MFMA tile per wave): C = A W^T + bias (+ add) + R, K % 32 == 0; any N (guarded scalar
// epilogue when rows are not 16-byte aligned); `groups` as in mc_launch_gemm (the *_gstride fields)
int mc_launch_gemm_small(const GemmArgs& g, hipStream_t stream, int groups = 1);

// The folded decoder tail of the large-batch schedule in one pass (gemm_tail_k): x0 = (wc H[r] + wu H[r + half]) W[0]^T +
// (wc Af[r] + wu Af[r + half]) W[1]^T + bias[0] + bias[1]; the CFG combination is formed while the A slab is staged.
struct TailArgs {
    const float* H = nullptr;       // residual stream [2 M][lda]: conditional rows, then (at + half elements) the unconditional ones
    const float* Af = nullptr;      // FiLM activations of the last StylizationBlock, same layout
    long half = 0, lda = 0;
    const float* W = nullptr;       // [2][N][ldw]: pose decoder, pose decoder x last FiLM Linear
    long ldw = 0, w_gstride = 0;
    const float* bias = nullptr;    // [2][N] at stride b_gstride
    long b_gstride = 0;
    float* C = nullptr;             // [M][ldc]
    float* C2 = nullptr;            // [M][ldc], optional: second partial product (gemm_tail2_k writes one per K group: x0 = C + C2; null: the one-output kernel)
    long ldc = 0;
    int M = 0, N = 0, K = 0;        // K per group
    float wc = 1.f, wu = 0.f;       // CFG weights ...
    const float* coef_table = nullptr;   // ... or (graph replay) read from coef_table[*step_ptr * coef_stride + {0, 1}]
    const int* step_ptr = nullptr;
    long coef_stride = 0;
    int tune = -1;
};
int mc_launch_gemm_tail(const TailArgs& g, hipStream_t stream);
int mc_gemm_default_tune();          // the resolved process default of GemmArgs::tune / TailArgs::tune
bool mc_gemm_tail_two_outputs(const TailArgs& g);      // true: this launch writes the two partial products C and C2 (x0 = C + C2), else C alone
