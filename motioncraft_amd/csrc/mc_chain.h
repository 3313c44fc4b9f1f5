// Register-chained row-panel kernels (mc_chain.hip): fused MLP, gate, combine+proj+LN+qkv.
#pragma once
#include <hip/hip_runtime.h>
#include "mc_kernels.h"
#include <stdint.h>

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
    // small batches: the hidden dimension is split over gridDim.z workgroups; split z writes its partial FC2 sums to
    // Y + z * y_sstride (b2 added by split 0 only); the caller reduces the partials in a fixed order
    int nsplit = 1;
    long y_sstride = 0;
    // MLP_EXPERT, nsplit == 4: the kernel picks 3 or 4 ways itself from the REAL tile count (mc_mlp_dyn_ways) -- with 3 ways
    // a B=1 layer's ~80 tiles are 240 workgroups, one per CU, instead of 320 (two rounds on 64 CUs); the tile count is
    // data-dependent and known on the device only.  The launch is then 1-D: workgroup b = (slice b / tiles, tile b % tiles).
    int dyn_split = 0;
    int dma = 0;                // L = 128, nsplit = 1: the LDS-DMA staged kernel mlp2d_k (same bits)
    long ledger_rows = 0;       // MLP_EXPERT: slots of the routing this launch walks (FLOP ledger only; the real tile count lives on the device)
};
// ways a small-batch expert launch splits its hidden dimension, from the number of real tiles (host + device)
__host__ __device__ inline int mc_mlp_dyn_ways(int real_tiles) { return (real_tiles * 4 <= 256 || real_tiles * 3 > 256) ? 4 : 3; }

struct GateArgs {
    const float* X = nullptr;      // token rows [N][ldx] (the residual stream viewed per part)
    long ldx = 0;
    const float* gamma = nullptr;  // STMA.norm
    const float* beta = nullptr;
    const float* emb = nullptr;    // MOE.embedding rows [(t,h)][L]
    int emb_mod = 1;
    float* Z = nullptr;            // [N][L]  LN(x)+embedding (expert input)
    const float* Wp = nullptr;     // cosine projector [256][L]
    const float* bp = nullptr;
    const float* sim_nT = nullptr; // [32][256]: unit columns of sim_matrix, transposed, rows >= E zero
    const float* logit_scale = nullptr;
    long tok0 = 0, N = 0;          // tokens [tok0, N)
    int zero_cnt = 1;              // launcher clears cnt first (0: the caller did, several launches accumulate)
    long small_tokens = 12000;     // launches of up to this many tokens run as gate_small_k (bit-identical; B <= 2 at 196 frames)
    int E = 0, L = 0;
    int* idx = nullptr;            // [N][2]
    float* gate = nullptr;         // [N][2]
    uint32_t* key = nullptr;       // [N]
    int* cnt = nullptr;            // [2][16] per (choice, expert) counts (zeroed by the launcher)
};

struct RowChainArgs {
    const float* X = nullptr;      // kind 0: Y2 [N][2][L] expert outputs per choice; kind 1: rows [N][ldx]
    long ldx = 0;
    const float* comb_w = nullptr; // kind 0: [N][2] gate if kept else 0
    const float* gamma = nullptr;  // kind 1: LayerNorm affine
    const float* beta = nullptr;
    const float* W = nullptr;      // [Nout][L]
    const float* bias = nullptr;
    float* Y = nullptr;            // [N][ldy]
    long ldy = 0;
    long tok0 = 0, N = 0;          // tokens [tok0, N)
    int L = 0, Nout = 0;
    long twin_from = 0;            // kind 0: tokens >= twin_from (> 0) read the expert outputs of token - twin_from
    TwinAlias alias;               // tokens >= alias.from are neither computed nor stored while *alias.split_flag == 0
    // projqkv (kind 0 followed by kind 1 on the first L output columns, in one kernel): the second weight stream
    const float* W2 = nullptr;     // [3L][L]
    const float* bias2 = nullptr;
    float* Y2 = nullptr;           // [N][ldy2]
    long ldy2 = 0;
    // pqbody (projqkv + the body-topology attention in one kernel): softmax(body_weight) [H][H] and the ys output [frames][H*L]
    const float* wsm = nullptr;
    float* ys = nullptr;
    long split_tokens = 20480;     // rowchain: launches of up to this many tokens slice the output chunks 4 ways over blockIdx.y
    long pad_row = -1;             // projqkv: first of 128 PADDING rows of Y and Y2 (behind the last real token): invalid lanes store there unconditionally
    // pqbody: an optional SECOND token range [tok2, N2) covered by the same launch (workgroups nblk1 .. of the grid; nblk1 = 0: none).
    // The twin layer's front launches it for the aliased rows of the second CFG half: those workgroups exit at once in the usual case,
    // and as part of the front's own launch they start inside its tail instead of queueing for LDS behind the next kernel
    long tok2 = 0, N2 = 0;
    int nblk1 = 0;
};

// FLOP ledger: tokens of [tok0, N) a row-chain launch really computes -- with twin aliasing armed (alias.split_flag set) the tokens from alias.from on
// are skipped while no twin pair is split by a capacity cut (the usual case; the flag lives on the device)
inline long mc_ledger_tokens(const RowChainArgs& g) {
    long hi = g.N;
    if (g.alias.split_flag && g.alias.from < hi) hi = g.alias.from;          // (a launch that starts at alias.from books nothing: it exits at once)
    return hi > g.tok0 ? hi - g.tok0 : 0;
}

bool mc_mlp_supported(int L, int hidden);
int mc_launch_mlp(int mode, const MlpArgs& g, int groups, int max_tiles, hipStream_t s);
int mc_launch_gate(const GateArgs& g, hipStream_t s);
int mc_launch_rowchain(int kind, const RowChainArgs& g, hipStream_t s);   // 0: combine+GELU+proj, 1: LN+linear
int mc_launch_pqbody(const RowChainArgs& g, int H, hipStream_t s);        // projqkv + body topology over frame-aligned tiles (L = 128, H = 12): q/k/v stay on chip, ys written
int mc_launch_projqkv(const RowChainArgs& g, hipStream_t s);              // both in one pass (gamma/beta = the LayerNorm of kind 1)
