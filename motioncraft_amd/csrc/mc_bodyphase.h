// Body-topology attention as a PHASE of the proj + q/k/v kernels (pqbody_k, mc_chain.hip; pqbody_h_k, mc_half.hip): static
// 12 x 12 topology + EfficientSelfAttention over the H parts of a frame (st_attention.py:123-134, efficient_attention.py:25-46) on
// q / k / v fragments parked in two LDS exchange slots [128 token rows][XS] (32 channels = 2 dynamic heads per slot).
//
// A workgroup owns FR = 128 / H whole frames (token row = frame * H + part).  The C^T fragments (lane = token, registers =
// channels) are written to a slot and read back transposed: lane = channel cc of one (frame, head), registers = the H parts --
// body_reg_k's register layout (mc_attn.hip), whose DPP-row contractions run unchanged.  Slot plan of one channel group (the
// caller places one workgroup barrier behind each weight chunk; Sq / Sk swap every group):
//     [q chunk -> Sq] b [k chunk -> Sk ; after_k: read q(Sq), softmax over the 16 channels] b
//     [v chunk -> Sq ; after_v: read k(Sk), softmax over the H parts] b [finish: read v(Sq): A = k^T v, y = q A, static + residual -> ys]
// 20 (frame, head) units of 16 lanes per group = 5 wave passes over 4 waves: wave w takes frames 2w, 2w+1; the fifth pass (frames
// 8, 9) is cut by body parts -- every wave softmaxes its keys and builds A, and projects H / 4 of the query rows (a whole extra
// pass on one wave per group made the other three wait at the chunk barriers: 2 passes on the critical path instead of 1.5).
#pragma once
#include "mc_common.h"
#include "mc_chain.h"

// NQ = H: the set produces all H parts of its frames; NQ < H: parts [h0, h0 + NQ) only
template <int H, int NQ>
struct BodySet {
    float q[NQ], k[H];
    int fl;        // frame of this lane inside the tile
    int h0;        // first part this set projects
    bool on;       // the frame exists (inside the tile, inside the range, not aliased)
};

// Lane-to-lane reads inside one dynamic head of HD = L / 8 lanes.  HD = 16: the head is one DPP row, read by rotation (row_ror:s).
// HD = 8: two heads share a DPP row, so the head is walked by XOR: lane i reads lane i ^ s -- s = 1..3 are quad permutations, and
// i ^ 4 .. i ^ 7 are the same permutations of the half-mirrored value (row_half_mirror: i -> i ^ 7 inside 8 lanes), which the caller
// forms once per register (xm).  Either way  A_[s] = sum_h read_s(k[h]) v[h]  holds A[d = src_s(lane)][lane], and q read through the
// SAME map pairs up again in  y[h][lane] = sum_s read_s(q[h]) A_[s]: no lane needs to know d.
template <int HD, int S>
__device__ __forceinline__ float head_read(float x, float xm) {
    if constexpr (HD == 16) return row_ror<S>(x);
    else {
        static_assert(HD == 8 && S < 8, "head_read: HD 16 or 8");
        if constexpr (S == 0) return x;
        else if constexpr (S == 1) return dpp_read<0xB1>(x);     // quad_perm [1,0,3,2]: i ^ 1
        else if constexpr (S == 2) return dpp_read<0x4E>(x);     // quad_perm [2,3,0,1]: i ^ 2
        else if constexpr (S == 3) return dpp_read<0x1B>(x);     // quad_perm [3,2,1,0]: i ^ 3
        else if constexpr (S == 4) return dpp_read<0x1B>(xm);    // (i ^ 3) ^ 7
        else if constexpr (S == 5) return dpp_read<0x4E>(xm);
        else if constexpr (S == 6) return dpp_read<0xB1>(xm);
        else return xm;                                          // i ^ 7
    }
}
template <class F, int... S>
__device__ __forceinline__ void static_for_n(F&& f, std::integer_sequence<int, S...>) { (f(std::integral_constant<int, S>{}), ...); }

template <int L, int H>
struct BodyPhase {
    static constexpr int HD = L / 8;                       // channels per dynamic head (8 heads)
    static_assert(HD == 16 || HD == 8, "BodyPhase: L = 128 (one head = one DPP row) or L = 64 (two heads per row)");
    static_assert(H % 4 == 0, "BodyPhase: the odd pass is cut 4 ways by parts");
    static constexpr int FR = 128 / H, TR = FR * H;        // frames / token rows of a tile
    static constexpr int NPASS = (FR + 1) / 2;             // wave passes (2 frames x 32 channels per pass)
    static_assert(NPASS <= 5, "BodyPhase: 4 full passes + one cut pass");
    static constexpr int XS = 36;                          // slot row stride in floats (b128 fragment writes conflict-free)
    static constexpr int SROWS = 128;                      // rows of a slot (120-row slots for pqbody_k<64>, 54.3 KB: no faster, M2D 18.40 either way)
    static constexpr int LDS_FLOATS = 2 * SROWS * XS + H * H;   // two slots + softmax(body_weight)

    const RowChainArgs& g;
    float* s_x;            // two exchange slots
    const float* s_w;      // softmax(body_weight) [H][H] in LDS
    long tile_tok0;
    bool aliasing;
    int lane, wave, cc;
    BodySet<H, H> b0;
    BodySet<H, H / 4> b1;

    __device__ __forceinline__ BodyPhase(const RowChainArgs& g_, float* s_x_, const float* s_w_, long tile_tok0_, bool aliasing_, int lane_, int wave_)
        : g(g_), s_x(s_x_), s_w(s_w_), tile_tok0(tile_tok0_), aliasing(aliasing_), lane(lane_), wave(wave_), cc(lane_ & 31) {}

    __device__ __forceinline__ float* slot(int i) const { return s_x + i * SROWS * XS; }

    template <class B>
    __device__ __forceinline__ void set_init(B& b, int pass, int h0) {
        b.fl = 2 * pass + (lane >> 5);
        b.h0 = h0;
        const long t0 = tile_tok0 + (long)b.fl * H;
        b.on = pass < NPASS && b.fl < FR && t0 < g.N && !(aliasing && t0 >= g.alias.from);
    }
    __device__ __forceinline__ void begin_group() {
        set_init(b0, wave, 0);
        set_init(b1, 4, wave * (H / 4));
    }
    template <class B>
    __device__ __forceinline__ void stage_q(B& b, const float* sl) {        // query: softmax over the 16 channels of the head
        constexpr int NQ = sizeof(b.q) / sizeof(float);
        if (!b.on) return;
        const float* x = sl + (b.fl * H + b.h0) * XS + cc;
#pragma unroll
        for (int h = 0; h < NQ; ++h) b.q[h] = x[h * XS];
#pragma unroll
        for (int h = 0; h < NQ; ++h) {
            const float m = group_max(b.q[h], HD);
            const float e = fast_exp2((b.q[h] - m) * LOG2E);
            b.q[h] = e * __frcp_rn(group_sum(e, HD));
        }
    }
    template <class B>
    __device__ __forceinline__ void stage_k(B& b, const float* sl) {        // key: softmax over the H body parts (in-lane)
        if (!b.on) return;
        const float* x = sl + (b.fl * H) * XS + cc;
#pragma unroll
        for (int h = 0; h < H; ++h) b.k[h] = x[h * XS];
        float m = b.k[0];
#pragma unroll
        for (int h = 1; h < H; ++h) m = fmaxf(m, b.k[h]);
        float sum = 0.f;
#pragma unroll
        for (int h = 0; h < H; ++h) { b.k[h] = fast_exp2((b.k[h] - m) * LOG2E); sum += b.k[h]; }
        const float rs = __frcp_rn(sum);
#pragma unroll
        for (int h = 0; h < H; ++h) b.k[h] *= rs;
    }
    template <class B>
    __device__ __forceinline__ void stage_v(B& b, const float* sl, int cg) {  // A = k^T v, y = q A (+ static topology + residual) -> ys
        constexpr int NQ = sizeof(b.q) / sizeof(float);
        if (!b.on) return;
        const long t0 = tile_tok0 + (long)b.fl * H;
        const float* x = sl + (b.fl * H) * XS + cc;
        const float* bvp = g.Y + t0 * g.ldy + cg * 32 + cc;      // raw body_value: stored by this workgroup in the projection phase
        float v[H], bv[H];
#pragma unroll
        for (int h = 0; h < H; ++h) bv[h] = bvp[h * g.ldy];
#pragma unroll
        for (int h = 0; h < H; ++h) v[h] = x[h * XS];
        float A_[HD];
        float km[H];                                   // HD = 8: the half-mirrored keys (head_read)
#pragma unroll
        for (int h = 0; h < H; ++h) km[h] = HD == 8 ? dpp_read<0x141>(b.k[h]) : 0.f;
        auto contract_kv = [&](auto S) {
            float a = 0.f;
#pragma unroll
            for (int h = 0; h < H; ++h) a += head_read<HD, decltype(S)::value>(b.k[h], km[h]) * v[h];
            A_[decltype(S)::value] = a;
        };
        static_for_n(contract_kv, std::make_integer_sequence<int, HD>{});
        float* out = g.ys + (t0 / H) * (long)(H * L) + cg * 32 + cc;
        if constexpr (NQ == H) {
#pragma unroll
            for (int h = 0; h < H; ++h) {
                float st = 0.f;
#pragma unroll
                for (int j = 0; j < H; ++j) st += s_w[h * H + j] * bv[j];
                float dy = 0.f;
                const float qm = HD == 8 ? dpp_read<0x141>(b.q[h]) : 0.f;
                auto contract_qa = [&](auto S) { dy += head_read<HD, decltype(S)::value>(b.q[h], qm) * A_[decltype(S)::value]; };
                static_for_n(contract_qa, std::make_integer_sequence<int, HD>{});
                out[h * L] = st + (bv[h] + dy);
            }
        } else {
#pragma unroll
            for (int hq = 0; hq < NQ; ++hq) {
                const int h = b.h0 + hq;               // wave-uniform
                float st = 0.f;
#pragma unroll
                for (int j = 0; j < H; ++j) st += s_w[h * H + j] * bv[j];
                float dy = 0.f;
                const float qm = HD == 8 ? dpp_read<0x141>(b.q[hq]) : 0.f;
                auto contract_qa = [&](auto S) { dy += head_read<HD, decltype(S)::value>(b.q[hq], qm) * A_[decltype(S)::value]; };
                static_for_n(contract_qa, std::make_integer_sequence<int, HD>{});
                float bvh = bv[0];                     // bv[h] for a runtime (wave-uniform) h without indexing the register array
#pragma unroll
                for (int j = 1; j < H; ++j) bvh = j == h ? bv[j] : bvh;
                out[h * L] = st + (bvh + dy);
            }
        }
    }
    __device__ __forceinline__ void after_k(const float* Sq) { stage_q(b0, Sq); stage_q(b1, Sq); }
    __device__ __forceinline__ void after_v(const float* Sk) { stage_k(b0, Sk); stage_k(b1, Sk); }
    __device__ __forceinline__ void finish(const float* Sv, int cg) { stage_v(b0, Sv, cg); stage_v(b1, Sv, cg); }
};
