// Shared pieces of the step-invariant transformer encoders (mc_textenc.hip defines them; mc_evalenc.hip reuses them):
// a generic encoder layer over [B*S, d] rows whose Linear layers are launches of the fp32 MFMA GEMM.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct EncLayer {
    const float *in_w, *in_b, *out_w, *out_b, *l1_w, *l1_b, *l2_w, *l2_b, *n1_g, *n1_b, *n2_g, *n2_b;
};

// C[M][N] = act(A[M][K] W[N][K]^T + bias) + R
int mc_enc_dense(const float* A, long lda, const float* W, long ldw, const float* bias, const float* R, long ldr, float* C,
                 long ldc, long M, int N, int K, int act, hipStream_t s);
// Y = LN_L(X) * g + b (rows of L floats, L % 4 == 0); relu: max(., 0) on the way out
int mc_enc_ln(const float* X, const float* g, const float* b, float* Y, long rows, int L, float eps, int relu, hipStream_t s);
// X[r][:] = emb[ids[r]][:] + pos[r % S][:]
int mc_enc_embed_tokens(const int* ids, const float* emb, const float* pos, float* X, long rows, int S, int d, int vocab,
                        hipStream_t s);
// one encoder layer over x [rows = B*S][d] in place; scratch: qkv [rows][3d], att [rows][d], y [rows][d], hid [rows][ff]
//   post-LN (nn.TransformerEncoderLayer norm_first=False, DistilBERT): x = LN1(x + SA(x)); x = LN2(x + FF(x))
//   pre-LN  (CLIP ResidualAttentionBlock):                              x = x + SA(LN1(x)); x = x + MLP(LN2(x))
// valid: uint8 [B][S] key mask (1 = may be attended) or nullptr; head_dim is 64
int mc_enc_layer(const EncLayer& p, float* x, float* qkv, float* att, float* y, float* hid, long rows, int B, int S, int d,
                 int heads, int ff, bool pre_ln, int act, int causal, const uint8_t* valid, float eps, hipStream_t s);
