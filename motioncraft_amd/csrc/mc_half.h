// fp16-MFMA variants of the GEMM-shaped kernels (mc_half.hip): the reduced-precision mode BASELINE.json configs[4]
// names ("fp16 MFMA").  Operands are fp16, accumulation is fp32 (v_mfma_f32_32x32x16_f16, 16x the fp32 MFMA rate).
//
//   MC_PREC_F16    every operand rounded to fp16 once (11-bit significand): ~2e-4 relative per GEMM stage
//   MC_PREC_F16X3  split operands x = hi + lo (two fp16, 22 bits) and three products hi*hi + hi*lo + lo*hi
//                  accumulated in fp32: fp32-class results (dropped lo*lo term and split residue ~2^-21) at 3 MFMAs
//
// Gate (LayerNorm + cosine projector + softmax + top-2), routing, LayerNorm statistics, softmaxes and every
// elementwise / normalisation op stay fp32 in both modes (SURVEY.md section 7 "hard parts": tutel forces fp32_gate).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mc_chain.h"

typedef _Float16 mc_half;

// hi[i] = fp16(x[i]); lo[i] = fp16(x[i] - hi[i])   (lo may be null).  Weights are split once per context.
int mc_launch_split_f16(const float* x, mc_half* hi, mc_half* lo, long n, hipStream_t s);
// The same for a [rows][K] matrix whose K axis is consumed in 32-wide chunks by a CHAINED second GEMM (mlp2_h_k FC2):
// inside every 32-chunk position p = 16 blk + 8 hf + i holds source column 16 blk + 4 hf + (i & 3) + 8 (i >> 2), the
// order in which an MFMA C^T accumulator fragment presents its 32 rows (mc_half.hip, mlp2_h_k).
int mc_launch_split_f16_chainperm(const float* x, mc_half* hi, mc_half* lo, long rows, int K, hipStream_t s);

struct GemmHArgs {
    const float* A = nullptr;     // [M][lda] fp32 activations (converted while staging) ...
    long lda = 0;
    const mc_half* Ah = nullptr;  // ... or, when set, fp16 planes [M][K] written by the producer (film_rows_k): hi
    const mc_half* Al = nullptr;  //     and lo (split mode)
    const mc_half* Wh = nullptr;  // [N][K] fp16 hi plane of the weight (nn.Linear layout)
    const mc_half* Wl = nullptr;  // lo plane (split mode)
    const float* bias = nullptr;  // [N]
    const float* R = nullptr;     // residual [M][ldr] (added after the activation)
    long ldr = 0;
    float* C = nullptr;           // [M][ldc]
    long ldc = 0;
    int M = 0, N = 0, K = 0;      // N % 128 == 0, K % 32 == 0, lda/ldc/ldr % 4 == 0
    int act = 0;
    int a_fm = 0;                 // the A planes are fragment-major (film_rows_k planes | 4): gemm_hf_k reads its fragments straight into registers
    int acc_init = 0;             // plane kernel, plain f16, R and bias set, no activation: the accumulators START as R + bias (loads in flight during the
                                  // DMA prologue; the epilogue is stores only) -- another fp32 summation order than (sum + bias) + R: not for the split mode
};
// C = act(A W^T + bias) + R with fp16 MFMA; split = three-product hi/lo form
int mc_launch_gemm_h(const GemmHArgs& g, bool split, hipStream_t s);

// mlp2_k (mc_chain.hip) on the fp16 MFMA: W1h/W1l [groups][hidden][L] and W2h/W2l [groups][L][hidden] (chain-permuted)
// fp16 planes replace MlpArgs::W1 / W2t; biases, X and Y stay fp32; GELU in fp32.  nsplit must be 1.
int mc_launch_mlp_h(int mode, const MlpArgs& g, const mc_half* W1h, const mc_half* W1l, const mc_half* W2h, const mc_half* W2l,
                    bool split, int groups, int max_tiles, hipStream_t s);
bool mc_mlp_h_supported(int L, int hidden);

// projqkv_k (mc_chain.hip) on the fp16 MFMA: Wph/Wpl [4L][L] planes of MOE.proj, Wqh/Wql [3L][L] planes of the q/k/v weight with
// the K axis chain-permuted; combine + GELU + LayerNorm stay fp32
int mc_launch_projqkv_h(const RowChainArgs& g, const mc_half* Wph, const mc_half* Wpl, const mc_half* Wqh, const mc_half* Wql, bool split,
                        hipStream_t s);
// projqkv_h + the body-topology attention over frame-aligned tiles (pqbody_k's fp16-MFMA twin; L = 128, H = 12): q/k/v stay on chip
int mc_launch_pqbody_h(const RowChainArgs& g, int H, const mc_half* Wph, const mc_half* Wpl, const mc_half* Wqh, const mc_half* Wql, bool split,
                       hipStream_t s);

// temporal_k (mc_attn.hip) with both contractions on the fp16 MFMA (softmax statistics, masks and scalings fp32); L = 128 / 64,
// whole-(sample, part) workgroups only (the column-sliced small-batch form stays on temporal_k)
int mc_launch_temporal_h(const float* mf, const float* tf, const float* mask, float* yt, int b0, int nb, int B, int T, int Nt, int H, int L,
                         bool split, hipStream_t s, const int* twin_flag, bool skip_text = false);
