// Common device/host helpers for the MotionCraft MI355X (gfx950 / CDNA4) sampling library.
// Written for gfx950 only: 64-wide wavefronts, f32-input MFMA (v_mfma_f32_32x32x2_f32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

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

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- device math --------------------------------------------------------------------
__device__ __forceinline__ float gelu_exact(float x) {
    // nn.GELU() default / F.gelu: 0.5 x (1 + erf(x / sqrt 2))
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }

// activation codes shared by host + device
enum { ACT_NONE = 0, ACT_GELU = 1, ACT_SILU = 2 };
__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACT_GELU) return gelu_exact(v);
    if (act == ACT_SILU) return silu_f(v);
    return v;
}

// reductions across `width` consecutive lanes (width power of two <= 64)
__device__ __forceinline__ float group_sum(float v, int width) {
    for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float group_max(float v, int width) {
    for (int o = width >> 1; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
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

// Static MFMA-pipe priority by hardware wave slot: the two waves that share a SIMD (one from each
// co-resident workgroup) otherwise run their MFMA-free stretches (staging, barrier, first fragment
// reads) in lockstep and leave the matrix pipe idle together.  Giving the odd slot priority 1 lets it
// drain its MFMA stream first, which de-phases the pair (MI355X_MICROARCH.md, "Two waves per SIMD").
__device__ __forceinline__ void prio_by_wave_slot() {
    const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | ((4 - 1) << 11));  // HW_REG_HW_ID.WAVE_ID[3:0]
    if (hw & 1) __builtin_amdgcn_s_setprio(1);
}
