// Common device/host helpers for the MotionCraft MI355X (gfx950 / CDNA4) sampling library.
// Written for gfx950 only: 64-wide wavefronts, f32-input MFMA (v_mfma_f32_32x32x2_f32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <utility>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MC_OK 0
#define MC_ERR_ARG 1
#define MC_ERR_HIP 2
#define MC_ERR_STATE 3

#define MC_HIP(expr)                                                                        \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            mc_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return MC_ERR_HIP;                                                              \
        }                                                                                   \
    } while (0)

#define MC_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            mc_set_error(__VA_ARGS__);        \
            return MC_ERR_ARG;                \
        }                                     \
    } while (0)

void mc_set_error(const char* fmt, ...);

#define MC_LAUNCH_CHECK() MC_HIP(hipGetLastError())

// debug FLOP ledger (mc_error.cpp; off unless mc_debug_flop_ledger(1)): launchers book the useful work of each launch under the kernel's name
extern bool mc_ledger_on;
void mc_ledger_add_(const char* kernel, long grid_threads, double flops);
// `grid` = the launch's dim3 grid of 256-thread workgroups: rows are keyed "kernel@<work-items>", the `grid` column of a rocprofv3 kernel trace
#define MC_LEDGER(kernel, grid, flops)                                                                                          \
    do {                                                                                                                        \
        if (mc_ledger_on) mc_ledger_add_((kernel), (long)(grid).x * (grid).y * (grid).z * 256, (double)(flops));              \
    } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- device math --------------------------------------------------------------------
constexpr float LOG2E = 1.44269504088896340736f;
// 2^x as the bare v_exp_f32 (1 ulp, flushes results below 2^-126 to 0): softmax numerators exp(x - max) are
// evaluated as fast_exp2(x * LOG2E - max * LOG2E), 2 VALU instructions instead of the ~20 of expf
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float gelu_exact(float x) {
    // nn.GELU() default / F.gelu: x * Phi(x), Phi(x) = 0.5 (1 + erf(x / sqrt 2)).
    // erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. fp32 round-off level; the resulting
    // GELU has max abs error 4.7e-7 over [-8, 8], the same as evaluating libm erff in fp32: 4.5e-7)
    // in 12 VALU + 2 transcendental instructions instead of the ~45 of the device erff (the fused MLPs pay for these lane-cycles:
    // fp32 VALU work shares the datapath with the fp32 MFMA).  With z = x / sqrt 2, t = 1 / (1 + p |z|),
    // h = 0.5 poly(t) t exp(-z^2):  gelu = x (1 - h) for x >= 0 and x h for x < 0, i.e. max(x, 0) - |x| h in both cases (no
    // cancellation for x < 0).  The 1 / sqrt 2 and the 0.5 are folded into the constants.
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, ax, 1.0f));   // v_rcp_f32, 1 ulp
    float p = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
    p = fmaf(p, t, 0.5f * 1.421413741f);
    p = fmaf(p, t, 0.5f * -0.284496736f);
    p = fmaf(p, t, 0.5f * 0.254829592f);
    const float h = p * t * fast_exp2(x * x * (-0.5f * LOG2E));
    return fmaf(-ax, h, fmaxf(x, 0.f));
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }

// activation codes shared by host + device
enum { ACT_NONE = 0, ACT_GELU = 1, ACT_SILU = 2, ACT_LRELU = 3, ACT_QUICKGELU = 4 };   // LRELU: nn.LeakyReLU() slope 0.01; QUICKGELU: x sigmoid(1.702 x) (CLIP)
__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACT_GELU) return gelu_exact(v);
    if (act == ACT_SILU) return silu_f(v);
    if (act == ACT_LRELU) return v >= 0.f ? v : 0.01f * v;
    if (act == ACT_QUICKGELU) return v / (1.0f + expf(-1.702f * v));
    return v;
}

// DPP cross-lane reads (gfx9 encodings): a modifier on the consuming VALU op, no LDS round trip --
// __shfl / __shfl_xor compile to ds_bpermute_b32 + s_waitcnt, ~50x the cost.
//   quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_mirror = 0x140, row_half_mirror = 0x141,
//   row_ror:n = 0x120 + n (rotate within each row of 16 lanes)
template <int CTRL>
__device__ __forceinline__ float dpp_read(float x) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), CTRL, 0xF, 0xF, true));
}
template <int N>
__device__ __forceinline__ float row_ror(float x) {
    if constexpr (N == 0) return x;
    else return dpp_read<0x120 + N>(x);
}

// f(integral_constant<int, s>) for s = 0..15: a DPP control word must be an immediate
template <class F, int... S>
__device__ __forceinline__ void static_for_seq(F&& f, std::integer_sequence<int, S...>) {
    (f(std::integral_constant<int, S>{}), ...);
}
template <class F>
__device__ __forceinline__ void static_for_16(F&& f) {
    static_for_seq(f, std::make_integer_sequence<int, 16>{});
}

// all-reduce across `width` consecutive lanes (width a power of two <= 64, groups aligned to width): the
// butterfly steps below 16 lanes are DPP (after the xor-1 and xor-2 steps a quad is uniform, so the mirrors
// act as xor-4 / xor-8); only the 16- and 32-lane steps go through the LDS crossbar.
__device__ __forceinline__ float group_sum(float v, int width) {
    if (width >= 2) v += dpp_read<0xB1>(v);
    if (width >= 4) v += dpp_read<0x4E>(v);
    if (width >= 8) v += dpp_read<0x141>(v);
    if (width >= 16) v += dpp_read<0x140>(v);
    if (width >= 32) v += __shfl_xor(v, 16, 64);
    if (width >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float group_max(float v, int width) {
    if (width >= 2) v = fmaxf(v, dpp_read<0xB1>(v));
    if (width >= 4) v = fmaxf(v, dpp_read<0x4E>(v));
    if (width >= 8) v = fmaxf(v, dpp_read<0x141>(v));
    if (width >= 16) v = fmaxf(v, dpp_read<0x140>(v));
    if (width >= 32) v = fmaxf(v, __shfl_xor(v, 16, 64));
    if (width >= 64) v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}

// XCD-aware block remap: consecutive remapped ids land on the same XCD (block b runs on XCD b % 8),
// bijective for any grid size (guide section 5, "XCD swizzle must be bijective").
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int NX = 8;
    int q = nwg / NX, r = nwg % NX;
    int xcd = bid % NX, idx = bid / NX;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

