// MoE routing for the tutel `cosine_top` gate as the reference configures it
// (st_attention.py:28-45; semantics restated in oracle/tutel_restated.py):
// top-2 of softmax(cosine logits), renormalised gates, batch-prioritised routing with
// capacity_factor 1.5 and token dropping.
//
// MI355X design: tutel materialises an [E, capacity, D] slot buffer addressed by the BPR
// "location" of every (token, choice).  The result only depends on the location through the
// drop test `location < capacity`, so this implementation never sorts: per (expert, choice)
// problem it radix-selects the capacity-th largest composite key
//        V = (float_bits(max score) << 32) | ~token_index        (all V distinct)
// which is exactly "rank by descending importance, stable in token order", and keeps the
// (token, choice) pairs with V >= V*.  Kept pairs are then compacted into contiguous per-expert
// slot ranges (order inside an expert is irrelevant: every row is an independent GEMM row) and
// a 128-row tile map is emitted for the grouped expert GEMMs.  Everything stays on the device:
// no host sync, graph-capturable.
#include "mc_common.h"
#include "mc_kernels.h"
#include <stdlib.h>

namespace {

constexpr int MAXE = 16;          // experts (lanes per token in the gate kernel)
constexpr int MAXP = 2 * MAXE;    // selection problems = (choice, expert)
constexpr int TILE_ROWS = 128;

// int state block layout
enum {
    ST_CNT = 0,                   // [2][MAXE]   tokens per (choice, expert)
    // slots are laid out per (slot group g, expert e), g = 0/1 = token below / at-or-above `gsplit`: the expert MLP of
    // each sample group can then be launched on its own stream (mc_model.hip); one group when gsplit >= N
    ST_KEPT = ST_CNT + MAXP,      // [2][MAXE]   kept pairs per (slot group, expert)
    ST_FILL = ST_KEPT + 2 * MAXE, // [2][MAXE]   compaction cursors
    ST_OFF = ST_FILL + 2 * MAXE,  // [2*MAXE+1]  slot range starts
    ST_ACTIVE = ST_OFF + 2 * MAXE + 1,  // [MAXP] 1: overflowed, needs selection; 0: keep all; -1: keep none
    ST_RANK = ST_ACTIVE + MAXP,   // [MAXP]      remaining rank during selection
    ST_ANY = ST_RANK + MAXP,      // [1]         any problem active
    ST_NTILES = ST_ANY + 1,       // [2]         tiles per slot group
    ST_DONE = ST_NTILES + 2,      // [1]         finished-workgroup counter of the histogram pass: the last one picks the bin
                                  // [1]  (ST_DONE + 1 = ST_SPLIT) twin mode: some token and its twin got different keep flags
    ST_SPLIT = ST_DONE + 1,
    ST_PREFIX = ST_DONE + 2,    // [MAXP][2]   (hi, lo) of the selected prefix / final threshold
    ST_HIST = ST_PREFIX + 2 * MAXP,  // [MAXP][256]
    ST_BAR = (ST_HIST + MAXP * 256 + 31) / 32 * 32,   // grid barrier words of route_coop_k (17 x 32 ints, zeroed once at context creation)
    ST_TOTAL = ST_BAR + 17 * 32
};

// tie_xor = 0xFFFFFFFF: among tokens of EQUAL importance the lower token index ranks first (a stable sort by descending
// score, what oracle/tutel_restated.py restates); tie_xor = 0: the higher index ranks first (the mirror order an
// implementation-defined, non-stable argsort could produce) -- RouteBufs::tie_xor, mc_ctx_set_tie_policy
__device__ __forceinline__ unsigned long long composite(uint32_t key, uint32_t tok, uint32_t tie_xor) {
    return ((unsigned long long)key << 32) | (unsigned long long)(tok ^ tie_xor);
}

// ---------------------------------------------------------------------------------------
// gate finish: 16 lanes per token (lane = expert).  tutel/gates/cosine_top.py + softmax + top-2.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gate_finish_k(const float* __restrict__ proj, const float* __restrict__ sim_n,
                                                     const float* __restrict__ logit_scale, long N, int E,
                                                     int* __restrict__ idx, float* __restrict__ gate,
                                                     uint32_t* __restrict__ key, int* __restrict__ state) {
    __shared__ float s_sim[256 * MAXE];
    __shared__ int s_cnt[MAXP];
    for (int i = threadIdx.x; i < 256 * MAXE; i += 256) {
        const int j = i / MAXE, e = i % MAXE;
        s_sim[i] = e < E ? sim_n[j * E + e] : 0.f;
    }
    if (threadIdx.x < MAXP) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int e = threadIdx.x & 15;
    const float scale = logit_scale[0];
    for (long tok = (long)blockIdx.x * 16 + (threadIdx.x >> 4); tok < N; tok += (long)gridDim.x * 16) {
        const float* p = proj + tok * 256;
        float ss = 0.f, dot = 0.f;
#pragma unroll 4
        for (int j = 0; j < 256; j += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p + j);
            ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
            dot += v[0] * s_sim[(j + 0) * MAXE + e] + v[1] * s_sim[(j + 1) * MAXE + e] +
                   v[2] * s_sim[(j + 2) * MAXE + e] + v[3] * s_sim[(j + 3) * MAXE + e];
        }
        const float denom = fmaxf(sqrtf(ss), 1e-12f);                 // F.normalize(dim=1)
        const float logit = e < E ? (dot / denom) * scale : -INFINITY;
        const float mx = group_max(logit, 16);
        const float ex = e < E ? expf(logit - mx) : 0.f;
        const float score = ex / group_sum(ex, 16);
        // top-2, lowest index wins ties
        const float m1 = group_max(score, 16);
        int c1 = (score == m1 && e < E) ? e : 99;
        for (int o = 8; o > 0; o >>= 1) c1 = min(c1, __shfl_xor(c1, o, 64));
        const float sc2 = (e == c1 || e >= E) ? -1.f : score;
        const float m2 = group_max(sc2, 16);
        int c2 = (sc2 == m2 && e < E && e != c1) ? e : 99;
        for (int o = 8; o > 0; o >>= 1) c2 = min(c2, __shfl_xor(c2, o, 64));
        if (e == 0) {
            const float den = fmaxf(m1 + m2, 1.1920928955078125e-07f);  // normalize_gate, finfo(float32).eps
            idx[tok * 2] = c1;
            idx[tok * 2 + 1] = c2;
            gate[tok * 2] = m1 / den;
            gate[tok * 2 + 1] = m2 / den;
            key[tok] = __float_as_uint(m1);
            atomicAdd(&s_cnt[c1], 1);
            atomicAdd(&s_cnt[MAXE + c2], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x < MAXP && s_cnt[threadIdx.x]) atomicAdd(&state[ST_CNT + threadIdx.x], s_cnt[threadIdx.x]);
}

// problem setup: limit per (choice, expert); active when the expert overflows.
__global__ void route_init_k(int* __restrict__ state, int E, int capacity, int cnt_mul) {
    const int p = threadIdx.x;  // 0..MAXP-1
    __shared__ int any;
    if (p == 0) any = 0;
    __syncthreads();
    if (p < MAXP) {
        const int choice = p / MAXE, e = p % MAXE;
        int act = 0, rank = 0;
        if (e < E) {
            const int c0 = state[ST_CNT + e] * cnt_mul;
            const int cnt = state[ST_CNT + p] * cnt_mul;
            const int limit = choice == 0 ? capacity : capacity - c0;   // second choices start at count_0[e]
            if (limit <= 0) act = cnt > 0 ? -1 : 0;
            else if (cnt > limit) { act = 1; rank = limit; }
        }
        state[ST_ACTIVE + p] = act;
        state[ST_RANK + p] = rank;
        state[ST_PREFIX + 2 * p] = 0;
        state[ST_PREFIX + 2 * p + 1] = 0;
        if (act == 1) atomicOr(&any, 1);
    }
    for (int i = threadIdx.x; i < MAXP * 256; i += blockDim.x) state[ST_HIST + i] = 0;
    if (threadIdx.x < 2 * MAXE) { state[ST_KEPT + threadIdx.x] = 0; state[ST_FILL + threadIdx.x] = 0; }
    if (threadIdx.x < 2) state[ST_DONE + threadIdx.x] = 0;
    __syncthreads();
    if (p == 0) state[ST_ANY] = any;
}

// one radix pass (byte `pass` from the top of the 64-bit composite key)
// (idx / gate / key hold the first Nsrc tokens; token tok >= Nsrc is the twin of tok - Nsrc: same scores, own index)
__global__ __launch_bounds__(256) void route_hist_k(const int* __restrict__ idx, const uint32_t* __restrict__ key,
                                                    long N, long Nsrc, int* __restrict__ state, int pass, uint32_t tie_xor) {
    if (state[ST_ANY] == 0) return;
    __shared__ int h[MAXP * 256];
    __shared__ int s_act[MAXP];
    __shared__ unsigned long long s_pre[MAXP];
    for (int i = threadIdx.x; i < MAXP * 256; i += 256) h[i] = 0;
    if (threadIdx.x < MAXP) {
        s_act[threadIdx.x] = state[ST_ACTIVE + threadIdx.x];
        s_pre[threadIdx.x] = ((unsigned long long)(uint32_t)state[ST_PREFIX + 2 * threadIdx.x] << 32) |
                             (uint32_t)state[ST_PREFIX + 2 * threadIdx.x + 1];
    }
    __syncthreads();
    const int shift = 56 - 8 * pass;
    for (long a = (long)blockIdx.x * 256 + threadIdx.x; a < 2 * N; a += (long)gridDim.x * 256) {
        const long tok = a >> 1;
        const long ts = tok >= Nsrc ? tok - Nsrc : tok;
        const int p = (int)(a & 1) * MAXE + idx[2 * ts + (a & 1)];
        if (s_act[p] != 1) continue;
        const unsigned long long V = composite(key[ts], (uint32_t)tok, tie_xor);
        if (pass > 0 && (V >> (shift + 8)) != s_pre[p]) continue;
        atomicAdd(&h[p * 256 + (int)((V >> shift) & 255)], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < MAXP * 256; i += 256)
        if (h[i]) atomicAdd(&state[ST_HIST + i], h[i]);
    // ---- the last workgroup to get here picks, per active problem, the bin holding the rank-th largest element ----
    __threadfence();
    __shared__ int s_last;
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(&state[ST_DONE], 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (!s_last) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int p = wave; p < MAXP; p += 4) {                       // one wavefront per problem, 4 bins per lane
        if (s_act[p] != 1) continue;
        int cb[4], t = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int* hp = &state[ST_HIST + p * 256 + 4 * lane + j];
            cb[j] = __hip_atomic_load(hp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // other CUs' atomics live in L2
            *hp = 0;
            t += cb[j];
        }
        int suf = t;                                             // inclusive suffix sum over lanes (bins descending)
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_down(suf, o, 64);
            if (lane + o < 64) suf += v;
        }
        const int rank = state[ST_RANK + p];
        int above = suf - t;                                     // elements in bins above this lane's 4
        int found = -1, above_found = 0;
#pragma unroll
        for (int j = 3; j >= 0; --j) {
            const int incl = above + cb[j];                      // suffix count including bin 4*lane+j
            if (found < 0 && incl >= rank && above < rank) { found = 4 * lane + j; above_found = above; }
            above = incl;
        }
        if (found >= 0) {
            unsigned long long pre = s_pre[p];
            pre = (pre << 8) | (unsigned long long)found;
            state[ST_PREFIX + 2 * p] = (int)(uint32_t)(pre >> 32);
            state[ST_PREFIX + 2 * p + 1] = (int)(uint32_t)pre;
            state[ST_RANK + p] = rank - above_found;
        }
    }
    if (threadIdx.x == 0) state[ST_DONE] = 0;
}

// slot ranges + 128-row tile maps: slot group g's tiles are written at [g * max_tiles, ...), its count at ST_NTILES + g
__global__ void route_plan_k(int* __restrict__ state, int E, int* __restrict__ tile_group,
                             int* __restrict__ tile_row0, int* __restrict__ tile_nrows, int max_tiles) {
    __shared__ int s_off[2 * MAXE + 1], s_t0[2][MAXE + 1], s_cnt2[2 * MAXE];
    if (threadIdx.x < 2 * MAXE) s_cnt2[threadIdx.x] = state[ST_KEPT + threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        int off = 0;
        for (int g = 0; g < 2; ++g) {
            int nt = 0;
            for (int e = 0; e < E; ++e) {
                s_off[g * MAXE + e] = off;
                s_t0[g][e] = nt;
                state[ST_OFF + g * MAXE + e] = off;
                const int cnt = s_cnt2[g * MAXE + e];
                off += cnt;
                nt += (cnt + TILE_ROWS - 1) / TILE_ROWS;
            }
            for (int e = E; e < MAXE; ++e) { s_off[g * MAXE + e] = off; state[ST_OFF + g * MAXE + e] = off; }
            s_t0[g][E] = nt;
            state[ST_NTILES + g] = min(nt, max_tiles);
        }
        s_off[2 * MAXE] = off;
        state[ST_OFF + 2 * MAXE] = off;
    }
    __syncthreads();
    for (int g = 0; g < 2; ++g)
        for (int e = 0; e < E; ++e) {                              // expert by expert: no per-tile search for its expert
            const int ve = g * MAXE + e, t1 = min(s_t0[g][e + 1], max_tiles);
            for (int t = s_t0[g][e] + threadIdx.x; t < t1; t += blockDim.x) {
                const int r = (t - s_t0[g][e]) * TILE_ROWS;
                tile_group[g * max_tiles + t] = e;
                tile_row0[g * max_tiles + t] = s_off[ve] + r;
                tile_nrows[g * max_tiles + t] = min(TILE_ROWS, s_off[ve + 1] - s_off[ve] - r);
            }
        }
}

// keep/drop + combine weights + per-expert kept counts
__global__ __launch_bounds__(256) void route_keep_k(const int* __restrict__ idx, const float* __restrict__ gate,
                                                    const uint32_t* __restrict__ key, long N, long Nsrc, long gsplit,
                                                    float* __restrict__ comb_w, int* __restrict__ state, uint32_t tie_xor) {
    __shared__ int s_act[MAXP];
    __shared__ unsigned long long s_thr[MAXP];
    __shared__ int s_kept[2 * MAXE];
    if (threadIdx.x < MAXP) {
        s_act[threadIdx.x] = state[ST_ACTIVE + threadIdx.x];
        s_thr[threadIdx.x] = ((unsigned long long)(uint32_t)state[ST_PREFIX + 2 * threadIdx.x] << 32) |
                             (uint32_t)state[ST_PREFIX + 2 * threadIdx.x + 1];
    }
    if (threadIdx.x < 2 * MAXE) s_kept[threadIdx.x] = 0;
    __syncthreads();
    for (long a = (long)blockIdx.x * 256 + threadIdx.x; a < 2 * N; a += (long)gridDim.x * 256) {
        const long tok = a >> 1;
        const long ts = tok >= Nsrc ? tok - Nsrc : tok;
        const long as = 2 * ts + (a & 1);
        const int e = idx[as];
        const int p = (int)(a & 1) * MAXE + e;
        bool keep = true;
        if (s_act[p] == -1) keep = false;
        else if (s_act[p] == 1) keep = composite(key[ts], (uint32_t)tok, tie_xor) >= s_thr[p];
        comb_w[a] = keep ? gate[as] : 0.f;
        if (tok >= Nsrc && s_act[p] == 1) {          // twin mode: would the original (same key, smaller index) decide differently?
            const bool keep_orig = composite(key[ts], (uint32_t)ts, tie_xor) >= s_thr[p];
            if (keep_orig != keep) state[ST_SPLIT] = 1;
        }
        if (keep && tok < Nsrc) atomicAdd(&s_kept[(tok >= gsplit ? MAXE : 0) + e], 1);   // expert slots exist for the first Nsrc tokens only
    }
    __syncthreads();
    if (threadIdx.x < 2 * MAXE && s_kept[threadIdx.x]) atomicAdd(&state[ST_KEPT + threadIdx.x], s_kept[threadIdx.x]);
}

// compaction: workgroup-local cursors in LDS, one global reservation per (workgroup, expert)
__global__ __launch_bounds__(256) void route_fill_k(const int* __restrict__ idx, const float* __restrict__ comb_w,
                                                    long N, long gsplit, int* __restrict__ state, int* __restrict__ src_row,
                                                    int* __restrict__ dst_row) {
    constexpr int PER = 8;  // pairs per thread
    __shared__ int s_cnt[2 * MAXE];
    __shared__ int s_base[2 * MAXE];
    if (threadIdx.x < 2 * MAXE) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const long a0 = ((long)blockIdx.x * 256 + threadIdx.x) * PER;
    int le[PER], lp[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const long a = a0 + i;
        le[i] = -1;
        if (a < 2 * N && comb_w[a] != 0.f) {
            le[i] = ((a >> 1) >= gsplit ? MAXE : 0) + idx[a];
            lp[i] = atomicAdd(&s_cnt[le[i]], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * MAXE && s_cnt[threadIdx.x])
        s_base[threadIdx.x] = state[ST_OFF + threadIdx.x] + atomicAdd(&state[ST_FILL + threadIdx.x], s_cnt[threadIdx.x]);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        if (le[i] >= 0) {
            const long a = a0 + i;
            const int slot = s_base[le[i]] + lp[i];
            src_row[slot] = (int)(a >> 1);
            dst_row[slot] = (int)a;
        }
    }
}

// ---------------------------------------------------------------------------------------
// The whole routing step of a layer as ONE launch for any batch that fits 10 pairs per thread on <= 256 workgroups
// (<= 655360 pairs: B <= 69 at 196 frames): the 12 launches above (init, 8 radix passes, keep, plan, fill) are the one
// place of the step where every stream waits -- routing ranks the tokens of the whole batch against each other -- so their
// launch gaps sit exposed on the critical path (~90 us per layer at B = 64, 60 us at B = 4).
// Workgroups are co-resident (256 threads, 33 KB of LDS) and meet at a grid barrier (arrival counter + generation word in
// the state block, device-scope atomics; the LAST workgroup to arrive runs the serial part of a phase -- pick the bins,
// plan the slot ranges -- before it releases the others).  Pairs live in registers as in route_small_k (importance bits +
// packed problem id), histograms go LDS -> global atomics, selection problems resolve early when a bin is taken whole,
// and the loop ends as soon as no problem is left selecting: 1 + (passes run, 4 unless scores tie exactly) + 1 barriers.
// Same integer decisions as the launch sequence above (test_cooperative_routing_kernel_equals_the_launch_sequence).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int ld_state(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Two-level arrival (MI355X_MICROARCH.md, barrier-xcd row): workgroups arrive at one of 8 group counters (group =
// blockIdx & 7, the XCD a block lands on under round-robin dispatch -- only speed depends on that), the last of a group
// arrives at the top counter, the last of all runs the hook and bumps the 8 group generation words the others poll.
// 32 arrivals / pollers per word instead of 256.  One release fence before the arrival, relaxed polling with s_sleep, ONE
// acquire fence after the wait.  Every word sits in its own 128-byte line: bar[0] top, bar[32 (1 + g)] group counter,
// bar[32 (9 + g)] group generation.
constexpr int BAR_STRIDE = 32, BAR_INTS = 17 * BAR_STRIDE;
// Returns false if the barrier timed out (the grid was not fully resident -- see mc_route_coop_reserve, which keeps that from
// happening inside one process; another PROCESS filling the GPU with barrier kernels of its own is the case left): the
// sticky error word bar[1] is raised, every workgroup leaves the kernel, and the host reports it (mc_ctx_check); the results of
// that routing call are invalid.  No trap: the process and the other contexts of the GPU keep running.
constexpr unsigned BAR_SPIN_LIMIT = 1u << 23;       // x s_sleep(1): a few seconds
template <class F>
__device__ __forceinline__ bool grid_sync(int* bar, int nwg, int* s_last, int* s_gen, F&& last_hook) {
    const int grp = (int)(blockIdx.x & 7), ngrp = nwg < 8 ? nwg : 8, members = (nwg - grp + 7) >> 3;
    int* cnt = bar + BAR_STRIDE * (1 + grp);
    int* gen = bar + BAR_STRIDE * (9 + grp);
    __syncthreads();
    if (threadIdx.x == 0) {
        *s_gen = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // (cannot advance before this workgroup arrives)
        __threadfence();                                          // this workgroup's writes are out before it signals
        int last = 0;
        if (__hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1) {
            __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = __hip_atomic_fetch_add(bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngrp - 1;
        }
        if (last) __threadfence();                                // acquire: everybody else's writes
        *s_last = last;
    }
    __syncthreads();
    if (*s_last) {
        last_hook();                                              // the whole workgroup
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_store(bar, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence();
        }
        __syncthreads();
        if ((int)threadIdx.x < ngrp) __hip_atomic_fetch_add(bar + BAR_STRIDE * (9 + (int)threadIdx.x), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (threadIdx.x == 0) {
        unsigned spins = 0;
        bool timed_out = false;
        while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == *s_gen) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > BAR_SPIN_LIMIT || __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { timed_out = true; break; }
        }
        if (timed_out) {
            __hip_atomic_store(bar + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // sticky; the other waiters see it and leave too
            *s_last = -1;
        } else {
            __threadfence();
        }
    }
    __syncthreads();
    return *s_last >= 0;
}

constexpr int COOP_PER = 10;
constexpr long COOP_MAX_WG = 512;      // grid limit of route_coop_k: 1.31 M (token, choice) pairs (B <= 139 at 196 frames; M2D 160 windows x 120 frames = 921 600);
                                       // round 3: was 256 -- beyond it the 12-launch sequence ran (M2D: 196 us per routing instead of ~50)
template <int PER>      // (token, choice) pairs per thread, register resident: 10, or 16 for fewer workgroups at the same pair count
__global__ __launch_bounds__(256) void route_coop_k(const int* __restrict__ idx, const float* __restrict__ gate,
                                                    const uint32_t* __restrict__ key, long N, long Nsrc, long gsplit, int E,
                                                    int capacity, int cnt_mul, float* __restrict__ comb_w, int* state,
                                                    int* __restrict__ src_row, int* __restrict__ dst_row,
                                                    int* __restrict__ tile_group, int* __restrict__ tile_row0,
                                                    int* __restrict__ tile_nrows, int max_tiles, uint32_t tie_xor, int skip_mid) {
    __shared__ int h[MAXP * 256];
    __shared__ int s_act[MAXP];
    __shared__ unsigned long long s_pre[MAXP];
    __shared__ int s_kept[2 * MAXE], s_cnt[2 * MAXE], s_base[2 * MAXE];
    __shared__ int s_off[2 * MAXE + 1], s_t0[2][MAXE + 1];
    __shared__ int s_last, s_gen, s_any;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwg = (int)gridDim.x;
    int* bar = state + ST_BAR;
    // pairs a = (blockIdx.x * PER + i) * 256 + tid: importance bits + problem id byte (-1: past the end / no slot)
    uint32_t kk[PER];
    int pp8[(PER + 3) / 4];
    auto get_p = [&](int i) { return (int)(pp8[i >> 2] << (24 - 8 * (i & 3))) >> 24; };
    auto set_p = [&](int i, int v) { pp8[i >> 2] = (pp8[i >> 2] & ~(0xFF << (8 * (i & 3)))) | ((v & 0xFF) << (8 * (i & 3))); };
    auto pair_of = [&](int i) { return ((long)blockIdx.x * PER + i) * 256 + tid; };
#pragma unroll
    for (int i = 0; i < (PER + 3) / 4; ++i) pp8[i] = -1;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const long a = pair_of(i);
        kk[i] = 0u;
        if (a < 2 * N) {
            const long tok = a >> 1;
            const long ts = tok >= Nsrc ? tok - Nsrc : tok;
            set_p(i, (int)(a & 1) * MAXE + idx[2 * ts + (a & 1)]);
            kk[i] = key[ts];
        }
    }
    // ---- problem setup (route_init_k) by workgroup 0; it also hands the (choice, expert) counts back zeroed ----
    if (blockIdx.x == 0) {
        if (tid == 0) s_any = 0;
        __syncthreads();
        if (tid < MAXP) {
            const int choice = tid / MAXE, e = tid % MAXE;
            int act = 0, rank = 0;
            if (e < E) {
                const int c0 = state[ST_CNT + e] * cnt_mul;
                const int cnt = state[ST_CNT + tid] * cnt_mul;
                const int limit = choice == 0 ? capacity : capacity - c0;
                if (limit <= 0) act = cnt > 0 ? -1 : 0;
                else if (cnt > limit) { act = 1; rank = limit; }
            }
            state[ST_ACTIVE + tid] = act;
            state[ST_RANK + tid] = rank;
            state[ST_PREFIX + 2 * tid] = 0;
            state[ST_PREFIX + 2 * tid + 1] = 0;
            if (act == 1) atomicOr(&s_any, 1);
        }
        for (int i = tid; i < MAXP * 256; i += 256) state[ST_HIST + i] = 0;
        if (tid < 2 * MAXE) { state[ST_KEPT + tid] = 0; state[ST_FILL + tid] = 0; }
        if (tid < 2) state[ST_DONE + tid] = 0;                    // (ST_DONE, ST_SPLIT)
        __syncthreads();
        if (tid < MAXP) state[ST_CNT + tid] = 0;
        if (tid == 0) state[ST_ANY] = s_any;
    }
    if (!grid_sync(bar, nwg, &s_last, &s_gen, [] {})) return;
    // ---- radix select, one byte per pass ----
    for (int pass = 0; pass < 8; ++pass) {
        if (ld_state(&state[ST_ANY]) == 0) break;                 // (uniform: written before the barrier every workgroup has passed)
        if (skip_mid && (pass == 4 || pass == 5)) continue;       // token indices < 2^16: the pick of pass 3 appended both constant bytes
        if (tid < MAXP) {
            s_act[tid] = ld_state(&state[ST_ACTIVE + tid]);
            s_pre[tid] = ((unsigned long long)(uint32_t)ld_state(&state[ST_PREFIX + 2 * tid]) << 32) | (uint32_t)ld_state(&state[ST_PREFIX + 2 * tid + 1]);
        }
        for (int i = tid; i < MAXP * 256; i += 256) h[i] = 0;
        __syncthreads();
        const int shift = 56 - 8 * pass;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int p = get_p(i);
            if (p < 0 || s_act[p] != 1) continue;
            const unsigned long long V = composite(kk[i], (uint32_t)(pair_of(i) >> 1), tie_xor);
            if (pass > 0 && (V >> (shift + 8)) != s_pre[p]) continue;
            atomicAdd(&h[p * 256 + (int)((V >> shift) & 255)], 1);
        }
        __syncthreads();
        for (int i = tid; i < MAXP * 256; i += 256)
            if (h[i]) atomicAdd(&state[ST_HIST + i], h[i]);
        const bool ok = grid_sync(bar, nwg, &s_last, &s_gen, [&] {
            // the last workgroup to arrive picks, per selecting problem, the bin that holds the rank-th largest key
            if (tid == 0) s_any = 0;
            __syncthreads();
            for (int p = wave; p < MAXP; p += 4) {
                if (s_act[p] != 1) continue;
                int cb[4], t = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int* hp = &state[ST_HIST + p * 256 + 4 * lane + j];
                    cb[j] = ld_state(hp);
                    __hip_atomic_store(hp, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    t += cb[j];
                }
                int suf = t;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int v = __shfl_down(suf, o, 64);
                    if (lane + o < 64) suf += v;
                }
                const int rank = ld_state(&state[ST_RANK + p]);
                int above = suf - t, found = -1, above_found = 0, incl_found = 0;
#pragma unroll
                for (int j = 3; j >= 0; --j) {
                    const int incl = above + cb[j];
                    if (found < 0 && incl >= rank && above < rank) { found = 4 * lane + j; above_found = above; incl_found = incl; }
                    above = incl;
                }
                if (found >= 0) {                                // exactly one lane finds the bin
                    unsigned long long pre = (s_pre[p] << 8) | (unsigned long long)found;
                    if (incl_found == rank && pass < 7) {        // the whole bin is kept: threshold = prefix, low bits zero
                        pre <<= shift;
                        state[ST_ACTIVE + p] = 2;
                    } else {
                        if (skip_mid && pass == 3) pre = (pre << 16) | (unsigned long long)((tie_xor >> 16) & 0xFFFFu);
                        state[ST_RANK + p] = rank - above_found;
                        s_any = 1;                               // (benign race: every writer stores 1)
                    }
                    state[ST_PREFIX + 2 * p] = (int)(uint32_t)(pre >> 32);
                    state[ST_PREFIX + 2 * p + 1] = (int)(uint32_t)pre;
                }
            }
            __syncthreads();
            if (tid == 0) state[ST_ANY] = s_any;
        });
        if (!ok) return;
    }
    // ---- keep / drop, combine weights, kept counts (route_keep_k) ----
    if (tid < MAXP) {
        s_act[tid] = ld_state(&state[ST_ACTIVE + tid]);
        s_pre[tid] = ((unsigned long long)(uint32_t)ld_state(&state[ST_PREFIX + 2 * tid]) << 32) | (uint32_t)ld_state(&state[ST_PREFIX + 2 * tid + 1]);
    }
    if (tid < 2 * MAXE) { s_kept[tid] = 0; s_cnt[tid] = 0; }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int p = get_p(i);
        if (p < 0) continue;
        const long a = pair_of(i);
        const long tok = a >> 1;
        const long ts = tok >= Nsrc ? tok - Nsrc : tok;
        const int act = s_act[p];
        bool keep = true;
        if (act == -1) keep = false;
        else if (act > 0) keep = composite(kk[i], (uint32_t)tok, tie_xor) >= s_pre[p];
        comb_w[a] = keep ? gate[2 * ts + (a & 1)] : 0.f;
        if (tok >= Nsrc && act > 0) {
            const bool keep_orig = composite(kk[i], (uint32_t)ts, tie_xor) >= s_pre[p];
            if (keep_orig != keep) state[ST_SPLIT] = 1;
        }
        if (!(keep && tok < Nsrc)) set_p(i, -1);                  // from here on: p >= 0 marks a pair that gets an expert slot
        else atomicAdd(&s_kept[(tok >= gsplit ? MAXE : 0) + (p & (MAXE - 1))], 1);
    }
    __syncthreads();
    if (tid < 2 * MAXE && s_kept[tid]) atomicAdd(&state[ST_KEPT + tid], s_kept[tid]);
    if (!grid_sync(bar, nwg, &s_last, &s_gen, [&] {
        // slot ranges + tile map (route_plan_k) by the last workgroup to arrive
        if (tid == 0) {
            int off = 0;
            for (int g = 0; g < 2; ++g) {
                int nt = 0;
                for (int e = 0; e < E; ++e) {
                    s_off[g * MAXE + e] = off;
                    s_t0[g][e] = nt;
                    state[ST_OFF + g * MAXE + e] = off;
                    const int cnt = ld_state(&state[ST_KEPT + g * MAXE + e]);
                    off += cnt;
                    nt += (cnt + TILE_ROWS - 1) / TILE_ROWS;
                }
                for (int e = E; e < MAXE; ++e) { s_off[g * MAXE + e] = off; state[ST_OFF + g * MAXE + e] = off; }
                s_t0[g][E] = nt;
                state[ST_NTILES + g] = min(nt, max_tiles);
            }
            s_off[2 * MAXE] = off;
            state[ST_OFF + 2 * MAXE] = off;
        }
        __syncthreads();
        for (int g = 0; g < 2; ++g)
            for (int e = 0; e < E; ++e) {
                const int ve = g * MAXE + e, t1 = min(s_t0[g][e + 1], max_tiles);
                for (int t = s_t0[g][e] + tid; t < t1; t += 256) {
                    const int r = (t - s_t0[g][e]) * TILE_ROWS;
                    tile_group[g * max_tiles + t] = e;
                    tile_row0[g * max_tiles + t] = s_off[ve] + r;
                    tile_nrows[g * max_tiles + t] = min(TILE_ROWS, s_off[ve + 1] - s_off[ve] - r);
                }
            }
    })) return;
    // ---- compaction (route_fill_k): workgroup-local cursors, one global reservation per (workgroup, expert) ----
    int lp[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int p = get_p(i);
        lp[i] = 0;
        if (p < 0) continue;
        const int le = ((pair_of(i) >> 1) >= gsplit ? MAXE : 0) + (p & (MAXE - 1));
        lp[i] = atomicAdd(&s_cnt[le], 1);
    }
    __syncthreads();
    if (tid < 2 * MAXE && s_cnt[tid]) s_base[tid] = ld_state(&state[ST_OFF + tid]) + atomicAdd(&state[ST_FILL + tid], s_cnt[tid]);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int p = get_p(i);
        if (p < 0) continue;
        const long a = pair_of(i);
        const int le = ((a >> 1) >= gsplit ? MAXE : 0) + (p & (MAXE - 1));
        const int slot = s_base[le] + lp[i];
        src_row[slot] = (int)(a >> 1);
        dst_row[slot] = (int)a;
    }
}

// ---------------------------------------------------------------------------------------
// Small batches (a few thousand tokens): the whole routing step -- problem setup, the 8 radix passes with their picks,
// keep / combine weights, slot ranges + tile map, compaction -- as ONE workgroup instead of 12 launches of a few
// microseconds of work each.  Same state block, same decisions (integer arithmetic on the same composite keys).
// ---------------------------------------------------------------------------------------
constexpr int SMALL_THREADS = 1024;
// PER = (token, choice) pairs per thread, held in registers for the whole kernel: problem id + composite key are built
// once, the radix passes and the keep / compaction phases touch no global memory for them.
// A selection problem is resolved early when the bin that holds the rank-th key is taken whole (all keys of the bin are
// kept: the threshold is the prefix with zero low bits) -- with distinct scores that happens after the 4 score bytes,
// the token-index bytes only ever split exact ties.
template <int PER>
__global__ __launch_bounds__(SMALL_THREADS) void route_small_k(const int* __restrict__ idx, const float* __restrict__ gate,
                                                              const uint32_t* __restrict__ key, long N, long Nsrc, long gsplit,
                                                              int E, int capacity, int cnt_mul, float* __restrict__ comb_w,
                                                              int* __restrict__ state, int* __restrict__ src_row,
                                                              int* __restrict__ dst_row, int* __restrict__ tile_group,
                                                              int* __restrict__ tile_row0, int* __restrict__ tile_nrows, int max_tiles,
                                                              uint32_t tie_xor) {
    __shared__ int h[MAXP * 256];
    __shared__ int s_act[MAXP], s_rank[MAXP];      // act: 1 selecting, 2 resolved (threshold in s_pre), 0 keep all, -1 keep none
    __shared__ unsigned long long s_pre[MAXP];
    __shared__ int s_kept[2 * MAXE], s_fill[2 * MAXE], s_off[2 * MAXE + 1], s_t0[2][MAXE + 1];
    __shared__ int s_any, s_split;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // pairs a = tid + i * SMALL_THREADS: importance key (1 register) and problem id (one byte, -1 past the end / no slot);
    // the composite key is rebuilt from them (the token index is a function of a)
    uint32_t kk[PER];
    int pp8[(PER + 3) / 4];
    auto get_p = [&](int i) { return (int)(pp8[i >> 2] << (24 - 8 * (i & 3))) >> 24; };
    auto set_p = [&](int i, int v) { pp8[i >> 2] = (pp8[i >> 2] & ~(0xFF << (8 * (i & 3)))) | ((v & 0xFF) << (8 * (i & 3))); };
    auto tok_of = [&](int i) { return (long)(tid + (long)i * SMALL_THREADS) >> 1; };
#pragma unroll
    for (int i = 0; i < (PER + 3) / 4; ++i) pp8[i] = -1;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const long a = tid + (long)i * SMALL_THREADS;
        kk[i] = 0u;
        if (a < 2 * N) {
            const long tok = a >> 1;
            const long ts = tok >= Nsrc ? tok - Nsrc : tok;
            set_p(i, (int)(a & 1) * MAXE + idx[2 * ts + (a & 1)]);
            kk[i] = key[ts];
        }
        if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);      // at most 8 pairs' loads in flight: enough to hide the latency, no spills
    }
    if (tid == 0) { s_any = 0; s_split = 0; }
    if (tid < 2 * MAXE) { s_kept[tid] = 0; s_fill[tid] = 0; }
    __syncthreads();
    if (tid < MAXP) {                                            // problem setup (route_init_k)
        const int choice = tid / MAXE, e = tid % MAXE;
        int act = 0, rank = 0;
        if (e < E) {
            const int c0 = state[ST_CNT + e] * cnt_mul;
            const int cnt = state[ST_CNT + tid] * cnt_mul;
            const int limit = choice == 0 ? capacity : capacity - c0;
            if (limit <= 0) act = cnt > 0 ? -1 : 0;
            else if (cnt > limit) { act = 1; rank = limit; }
        }
        s_act[tid] = act;
        s_rank[tid] = rank;
        s_pre[tid] = 0ull;
        if (act == 1) atomicOr(&s_any, 1);
    }
    __syncthreads();
    if (tid < MAXP) state[ST_CNT + tid] = 0;                     // counts consumed: left clean for the next layer's gate (no memset launch)
#pragma unroll 1
    for (int pass = 0; pass < 8 && s_any; ++pass) {              // radix select, one byte per pass (route_hist_k + pick)
        if (pass == 4 || pass == 5) {
            // token indices are < 2^16 here (<= 32768 pairs), so the two upper bytes of (token ^ tie_xor) are those of
            // tie_xor for every key: the pass cannot split the candidates, append the byte and go on (rank unchanged)
            if (tid < MAXP && s_act[tid] == 1) s_pre[tid] = (s_pre[tid] << 8) | (unsigned long long)((tie_xor >> (pass == 4 ? 24 : 16)) & 0xFFu);
            __syncthreads();
            continue;
        }
        for (int i = tid; i < MAXP * 256; i += SMALL_THREADS) h[i] = 0;
        __syncthreads();
        if (tid == 0) s_any = 0;                                 // (every thread read it in the loop condition before the barrier above)
        const int shift = 56 - 8 * pass;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int p = get_p(i);
            if (p < 0 || s_act[p] != 1) continue;
            const unsigned long long V = composite(kk[i], (uint32_t)tok_of(i), tie_xor);
            if (pass > 0 && (V >> (shift + 8)) != s_pre[p]) continue;
            atomicAdd(&h[p * 256 + (int)((V >> shift) & 255)], 1);
        }
        __syncthreads();
        for (int p = wave; p < MAXP; p += SMALL_THREADS / 64) {
            if (s_act[p] != 1) continue;
            int cb[4], t = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) { cb[j] = h[p * 256 + 4 * lane + j]; t += cb[j]; }
            int suf = t;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_down(suf, o, 64);
                if (lane + o < 64) suf += v;
            }
            const int rank = s_rank[p];
            int above = suf - t, found = -1, above_found = 0, incl_found = 0;
#pragma unroll
            for (int j = 3; j >= 0; --j) {
                const int incl = above + cb[j];
                if (found < 0 && incl >= rank && above < rank) { found = 4 * lane + j; above_found = above; incl_found = incl; }
                above = incl;
            }
            if (found >= 0) {                                // exactly one lane finds the bin
                const unsigned long long pre = (s_pre[p] << 8) | (unsigned long long)found;
                if (incl_found == rank && pass < 7) {        // the whole bin is kept: threshold = prefix, low bits zero
                    s_pre[p] = pre << shift;
                    s_act[p] = 2;
                } else {
                    s_pre[p] = pre;
                    s_rank[p] = rank - above_found;
                    s_any = 1;                               // (benign race: every writer stores 1)
                }
            }
        }
        __syncthreads();
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER; ++i) {                               // keep / drop, combine weights, kept counts (route_keep_k)
        const int p = get_p(i);
        if (p < 0) continue;
        const long a = tid + (long)i * SMALL_THREADS;
        const long tok = a >> 1;
        const long ts = tok >= Nsrc ? tok - Nsrc : tok;
        const int act = s_act[p];
        bool keep = true;
        if (act == -1) keep = false;
        else if (act > 0) keep = composite(kk[i], (uint32_t)tok, tie_xor) >= s_pre[p];
        comb_w[a] = keep ? gate[2 * ts + (a & 1)] : 0.f;
        if (tok >= Nsrc && act > 0) {
            const bool keep_orig = composite(kk[i], (uint32_t)ts, tie_xor) >= s_pre[p];
            if (keep_orig != keep) s_split = 1;
        }
        if (!(keep && tok < Nsrc)) set_p(i, -1);                  // from here on: p >= 0 marks a pair that gets an expert slot
        else atomicAdd(&s_kept[(tok >= gsplit ? MAXE : 0) + (p & (MAXE - 1))], 1);
        if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    if (tid == 0) {                                              // slot ranges + tile counts (route_plan_k)
        int off = 0;
        for (int g = 0; g < 2; ++g) {
            int nt = 0;
            for (int e = 0; e < E; ++e) {
                s_off[g * MAXE + e] = off;
                s_t0[g][e] = nt;
                state[ST_OFF + g * MAXE + e] = off;
                const int cnt = s_kept[g * MAXE + e];
                off += cnt;
                nt += (cnt + TILE_ROWS - 1) / TILE_ROWS;
            }
            for (int e = E; e < MAXE; ++e) { s_off[g * MAXE + e] = off; state[ST_OFF + g * MAXE + e] = off; }
            s_t0[g][E] = nt;
            state[ST_NTILES + g] = min(nt, max_tiles);
        }
        s_off[2 * MAXE] = off;
        state[ST_OFF + 2 * MAXE] = off;
        state[ST_SPLIT] = s_split;
        state[ST_DONE] = 0;
    }
    __syncthreads();
    for (int g = 0; g < 2; ++g)
        for (int e = 0; e < E; ++e) {
            const int ve = g * MAXE + e, t1 = min(s_t0[g][e + 1], max_tiles);
            for (int t = s_t0[g][e] + tid; t < t1; t += SMALL_THREADS) {
                const int r = (t - s_t0[g][e]) * TILE_ROWS;
                tile_group[g * max_tiles + t] = e;
                tile_row0[g * max_tiles + t] = s_off[ve] + r;
                tile_nrows[g * max_tiles + t] = min(TILE_ROWS, s_off[ve + 1] - s_off[ve] - r);
            }
        }
#pragma unroll
    for (int i = 0; i < PER; ++i) {                               // compaction (route_fill_k)
        const int p = get_p(i);
        if (p < 0) continue;
        const long a = tid + (long)i * SMALL_THREADS;
        const int le = ((a >> 1) >= gsplit ? MAXE : 0) + (p & (MAXE - 1));
        const int slot = s_off[le] + atomicAdd(&s_fill[le], 1);
        src_row[slot] = (int)(a >> 1);
        dst_row[slot] = (int)a;
        if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
}

// The same kernel for more pairs than fit in registers (20 < pairs per thread <= 32): idx / key are re-read from L2 in
// every phase.
__global__ __launch_bounds__(SMALL_THREADS) void route_small_stream_k(const int* __restrict__ idx, const float* __restrict__ gate,
                                                              const uint32_t* __restrict__ key, long N, long Nsrc, long gsplit,
                                                              int E, int capacity, int cnt_mul, float* __restrict__ comb_w,
                                                              int* __restrict__ state, int* __restrict__ src_row,
                                                              int* __restrict__ dst_row, int* __restrict__ tile_group,
                                                              int* __restrict__ tile_row0, int* __restrict__ tile_nrows, int max_tiles,
                                                              uint32_t tie_xor) {
    __shared__ int h[MAXP * 256];
    __shared__ int s_act[MAXP], s_rank[MAXP];
    __shared__ unsigned long long s_pre[MAXP];
    __shared__ int s_kept[2 * MAXE], s_fill[2 * MAXE], s_off[2 * MAXE + 1], s_t0[2][MAXE + 1];
    __shared__ int s_any, s_split;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { s_any = 0; s_split = 0; }
    if (tid < 2 * MAXE) { s_kept[tid] = 0; s_fill[tid] = 0; }
    __syncthreads();
    if (tid < MAXP) {                                            // problem setup (route_init_k)
        const int choice = tid / MAXE, e = tid % MAXE;
        int act = 0, rank = 0;
        if (e < E) {
            const int c0 = state[ST_CNT + e] * cnt_mul;
            const int cnt = state[ST_CNT + tid] * cnt_mul;
            const int limit = choice == 0 ? capacity : capacity - c0;
            if (limit <= 0) act = cnt > 0 ? -1 : 0;
            else if (cnt > limit) { act = 1; rank = limit; }
        }
        s_act[tid] = act;
        s_rank[tid] = rank;
        s_pre[tid] = 0ull;
        if (act == 1) atomicOr(&s_any, 1);
    }
    __syncthreads();
    if (tid < MAXP) state[ST_CNT + tid] = 0;                     // counts consumed: left clean for the next layer's gate (no memset launch)
    if (s_any) {
        for (int pass = 0; pass < 8; ++pass) {                   // radix select, one byte per pass (route_hist_k + pick)
            if (pass == 4 || pass == 5) {
                // token indices are < 2^16 here (<= 32768 pairs), so the two upper bytes of (token ^ tie_xor) are those of
                // tie_xor for every key: the pass cannot split the candidates, append the byte and go on (rank unchanged)
                if (tid < MAXP && s_act[tid] == 1) s_pre[tid] = (s_pre[tid] << 8) | (unsigned long long)((tie_xor >> (pass == 4 ? 24 : 16)) & 0xFFu);
                __syncthreads();
                continue;
            }
            for (int i = tid; i < MAXP * 256; i += SMALL_THREADS) h[i] = 0;
            __syncthreads();
            const int shift = 56 - 8 * pass;
            for (long a = tid; a < 2 * N; a += SMALL_THREADS) {
                const long tok = a >> 1;
                const long ts = tok >= Nsrc ? tok - Nsrc : tok;
                const int p = (int)(a & 1) * MAXE + idx[2 * ts + (a & 1)];
                if (s_act[p] != 1) continue;
                const unsigned long long V = composite(key[ts], (uint32_t)tok, tie_xor);
                if (pass > 0 && (V >> (shift + 8)) != s_pre[p]) continue;
                atomicAdd(&h[p * 256 + (int)((V >> shift) & 255)], 1);
            }
            __syncthreads();
            for (int p = wave; p < MAXP; p += SMALL_THREADS / 64) {
                if (s_act[p] != 1) continue;
                int cb[4], t = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) { cb[j] = h[p * 256 + 4 * lane + j]; t += cb[j]; }
                int suf = t;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int v = __shfl_down(suf, o, 64);
                    if (lane + o < 64) suf += v;
                }
                const int rank = s_rank[p];
                int above = suf - t, found = -1, above_found = 0;
#pragma unroll
                for (int j = 3; j >= 0; --j) {
                    const int incl = above + cb[j];
                    if (found < 0 && incl >= rank && above < rank) { found = 4 * lane + j; above_found = above; }
                    above = incl;
                }
                if (found >= 0) {                                // exactly one lane finds the bin
                    s_pre[p] = (s_pre[p] << 8) | (unsigned long long)found;
                    s_rank[p] = rank - above_found;
                }
            }
            __syncthreads();
        }
    }
    for (long a = tid; a < 2 * N; a += SMALL_THREADS) {           // keep / drop, combine weights, kept counts (route_keep_k)
        const long tok = a >> 1;
        const long ts = tok >= Nsrc ? tok - Nsrc : tok;
        const long as = 2 * ts + (a & 1);
        const int e = idx[as];
        const int p = (int)(a & 1) * MAXE + e;
        bool keep = true;
        if (s_act[p] == -1) keep = false;
        else if (s_act[p] == 1) keep = composite(key[ts], (uint32_t)tok, tie_xor) >= s_pre[p];
        comb_w[a] = keep ? gate[as] : 0.f;
        if (tok >= Nsrc && s_act[p] == 1) {
            const bool keep_orig = composite(key[ts], (uint32_t)ts, tie_xor) >= s_pre[p];
            if (keep_orig != keep) s_split = 1;
        }
        if (keep && tok < Nsrc) atomicAdd(&s_kept[(tok >= gsplit ? MAXE : 0) + e], 1);
    }
    __syncthreads();
    if (tid == 0) {                                              // slot ranges + tile counts (route_plan_k)
        int off = 0;
        for (int g = 0; g < 2; ++g) {
            int nt = 0;
            for (int e = 0; e < E; ++e) {
                s_off[g * MAXE + e] = off;
                s_t0[g][e] = nt;
                state[ST_OFF + g * MAXE + e] = off;
                const int cnt = s_kept[g * MAXE + e];
                off += cnt;
                nt += (cnt + TILE_ROWS - 1) / TILE_ROWS;
            }
            for (int e = E; e < MAXE; ++e) { s_off[g * MAXE + e] = off; state[ST_OFF + g * MAXE + e] = off; }
            s_t0[g][E] = nt;
            state[ST_NTILES + g] = min(nt, max_tiles);
        }
        s_off[2 * MAXE] = off;
        state[ST_OFF + 2 * MAXE] = off;
        state[ST_SPLIT] = s_split;
        state[ST_DONE] = 0;
    }
    __syncthreads();
    for (int g = 0; g < 2; ++g)
        for (int e = 0; e < E; ++e) {
            const int ve = g * MAXE + e, t1 = min(s_t0[g][e + 1], max_tiles);
            for (int t = s_t0[g][e] + tid; t < t1; t += SMALL_THREADS) {
                const int r = (t - s_t0[g][e]) * TILE_ROWS;
                tile_group[g * max_tiles + t] = e;
                tile_row0[g * max_tiles + t] = s_off[ve] + r;
                tile_nrows[g * max_tiles + t] = min(TILE_ROWS, s_off[ve + 1] - s_off[ve] - r);
            }
        }
    for (long a = tid; a < 2 * Nsrc; a += SMALL_THREADS) {        // compaction (route_fill_k): a -> thread mapping as in the keep loop
        if (comb_w[a] == 0.f) continue;
        const int le = ((a >> 1) >= gsplit ? MAXE : 0) + idx[a];
        const int slot = s_off[le] + atomicAdd(&s_fill[le], 1);
        src_row[slot] = (int)(a >> 1);
        dst_row[slot] = (int)a;
    }
}

}  // namespace

static long route_small_pairs() {
    static const long v = [] { const char* e = getenv("MC_ROUTE_SMALL"); const long x = e ? atol(e) : 20480L; return x > 131072L ? 131072L : x; }();   // the one-workgroup kernels assume token indices < 2^16; default = what fits the register kernels (beyond: route_coop_k; B=3: 97.5 -> 90.4 ms vs the streaming form)
    return v;
}

size_t mc_route_state_ints(int) { return ST_TOTAL; }
bool mc_route_is_small(long N) { return 2 * N <= route_small_pairs(); }
bool mc_route_cleans_counts(const RouteBufs& rb, long N) {
    return 2 * N <= (rb.small_pairs >= 0 ? rb.small_pairs : route_small_pairs()) || (rb.coop && 2 * N <= COOP_MAX_WG * 256 * COOP_PER);
}
size_t mc_route_barrier_offset() { return ST_BAR; }
size_t mc_route_barrier_ints() { return 17 * 32; }
size_t mc_route_error_offset() { return ST_BAR + 1; }

// ---- admission of the cooperative routing kernel ---------------------------------------------------------------
// route_coop_k meets at a hand-rolled grid barrier, so ALL its workgroups have to be resident at once.  A context launches at
// most one such kernel at a time (stream order), so the library reserves a context's workgroups out of what the device can
// hold -- occupancy query x compute units of the VISIBLE device (a CPX partition reports its own 32 CUs) -- when the context
// is created, and a context that does not fit runs the launch sequence instead (RouteBufs::coop = false).  Other kernels in
// flight only delay the barrier (they finish); only barrier kernels the library does not know of (another process on the same
// GPU) can still starve it, and then the barrier times out into an error flag (grid_sync) instead of hanging.
#include <atomic>
static std::atomic<int> g_coop_reserved[64];
int mc_route_coop_wgs(long N) { return (2 * N <= COOP_MAX_WG * 256 * COOP_PER) ? cdiv(2 * N, 256L * COOP_PER) : 0; }
int mc_route_coop_slots(int dev) {
    static std::atomic<int> cached[64];
    if (dev < 0 || dev >= 64) return 0;
    int v = cached[dev].load();
    if (v > 0) return v;
    int per_cu = 0, per_cu16 = 0;
    hipDeviceProp_t prop;
    // the launch picks route_coop_k<10> or <16> (another register footprint): admit against the smaller of the two occupancies
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, route_coop_k<COOP_PER>, 256, 0) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu16, route_coop_k<16>, 256, 0) != hipSuccess ||
        hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    if (per_cu16 < per_cu) per_cu = per_cu16;
    if (per_cu > 8) per_cu = 8;                    // (hardware admits at most 8 256-thread blocks per CU: MI355X_MICROARCH.md, residency)
    v = per_cu * prop.multiProcessorCount;
    if (const char* e = getenv("MC_ROUTE_COOP_SLOTS")) v = atoi(e) > 0 ? atoi(e) : v;      // (tests: a small capacity exercises the fallback)
    cached[dev].store(v);
    return v;
}
bool mc_route_coop_reserve(int dev, int nwg) {
    if (nwg <= 0 || dev < 0 || dev >= 64) return false;
    const int slots = mc_route_coop_slots(dev);
    int cur = g_coop_reserved[dev].load();
    while (cur + nwg <= slots)
        if (g_coop_reserved[dev].compare_exchange_weak(cur, cur + nwg)) return true;
    return false;
}
void mc_route_coop_release(int dev, int nwg) {      // dev = the device the reservation was taken on (not whatever is current at destroy time)
    if (nwg > 0 && dev >= 0 && dev < 64) g_coop_reserved[dev].fetch_sub(nwg);
}
const int* mc_route_num_tiles_ptr(const RouteBufs& rb, int group) { return rb.state + ST_NTILES + group; }
const int* mc_route_split_flag_ptr(const RouteBufs& rb) { return rb.state + ST_SPLIT; }

int mc_launch_gate_finish(const float* proj, const float* sim_n, const float* logit_scale, long N, int E,
                          RouteBufs rb, hipStream_t s) {
    MC_REQUIRE(E >= 2 && E <= MAXE, "gate: num_experts=%d unsupported (2..16)", E);
    MC_HIP(hipMemsetAsync(rb.state, 0, sizeof(int) * (ST_CNT + MAXP), s));
    int blocks = cdiv(N, 16);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gate_finish_k, dim3(blocks), dim3(256), 0, s, proj, sim_n, logit_scale, N, E, rb.idx, rb.gate,
                       rb.key, rb.state);
    MC_LAUNCH_CHECK();
    return MC_OK;
}

// Nsrc == N: the plain case.  Nsrc == N/2 ("twin" mode): tokens [N/2, N) are exact copies of [0, N/2) (the two CFG
// halves enter the first decoder layer with the same residual stream), gate outputs exist for the first half only;
// the capacity test still ranks all N tokens (a twin ranks right behind its original: same score, larger index, so
// "twin kept => original kept"), combine weights are produced for all N, expert slots only for the first half.
// gsplit: tokens >= gsplit form slot group 1 (own slot ranges and tile map at [max_tiles, 2 max_tiles)); >= N: one group.
int mc_launch_route(long N, long Nsrc, long gsplit, int E, int capacity, RouteBufs rb, hipStream_t s) {
    MC_REQUIRE(Nsrc == N || 2 * Nsrc == N, "route: Nsrc=%ld must be N or N/2 (N=%ld)", Nsrc, N);
    MC_REQUIRE(Nsrc == N || rb.tie_xor == 0xFFFFFFFFu, "route: the twin mode needs the stable tie order (a twin must rank right behind its original)");
    if (2 * N <= (rb.small_pairs >= 0 ? rb.small_pairs : route_small_pairs())) {
#define MC_ROUTE_SMALL(KERNEL)                                                                                                       \
    hipLaunchKernelGGL(KERNEL, dim3(1), dim3(SMALL_THREADS), 0, s, rb.idx, rb.gate, rb.key, N, Nsrc, gsplit, E, capacity,   \
                       (int)(N / Nsrc), rb.comb_w, rb.state, rb.src_row, rb.dst_row, rb.tile_group, rb.tile_row0, rb.tile_nrows,        \
                       rb.max_tiles, rb.tie_xor)
        if (rb.reg_kernel && 2 * N <= 10L * SMALL_THREADS) MC_ROUTE_SMALL(route_small_k<10>);
        else if (rb.reg_kernel && 2 * N <= 20L * SMALL_THREADS) MC_ROUTE_SMALL(route_small_k<20>);      // (25 registers spill to scratch: still -4 % per step at B=2 vs the streaming form)
        else MC_ROUTE_SMALL(route_small_stream_k);
#undef MC_ROUTE_SMALL
        MC_LAUNCH_CHECK();
        return MC_OK;
    }
    if (rb.coop && 2 * N <= COOP_MAX_WG * 256 * COOP_PER) {
        // pairs per thread: 10 while that is at most one workgroup per CU (<= 256; B=64 at 196 frames: 236 workgroups, 45 us either way --
        // the ~6 grid barriers set the time), 16 beyond (M2D at 160 windows per GPU: 360 -> 225 workgroups, 88 -> 80 us per routing)
        const int per = rb.coop_per == 10 || rb.coop_per == 16 ? rb.coop_per : (cdiv(2 * N, 256L * COOP_PER) > 256 ? 16 : 10);
        const int nwg = cdiv(2 * N, 256L * per);              // (<= the PER = 10 count the context reserved)
#define MC_ROUTE_COOP(P)                                                                                                             \
    hipLaunchKernelGGL(route_coop_k<P>, dim3(nwg), dim3(256), 0, s, rb.idx, rb.gate, rb.key, N, Nsrc, gsplit, E, capacity, (int)(N / Nsrc), \
                       rb.comb_w, rb.state, rb.src_row, rb.dst_row, rb.tile_group, rb.tile_row0, rb.tile_nrows, rb.max_tiles, rb.tie_xor,  \
                       N <= 65536 ? 1 : 0)
        if (per == 16) MC_ROUTE_COOP(16); else MC_ROUTE_COOP(10);
#undef MC_ROUTE_COOP
        MC_LAUNCH_CHECK();
        return MC_OK;
    }
    hipLaunchKernelGGL(route_init_k, dim3(1), dim3(256), 0, s, rb.state, E, capacity, (int)(N / Nsrc));
    int blocks = cdiv(2 * N, 256 * 8);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    for (int pass = 0; pass < 8; ++pass) {
        hipLaunchKernelGGL(route_hist_k, dim3(blocks), dim3(256), 0, s, rb.idx, rb.key, N, Nsrc, rb.state, pass, rb.tie_xor);
    }
    hipLaunchKernelGGL(route_keep_k, dim3(blocks), dim3(256), 0, s, rb.idx, rb.gate, rb.key, N, Nsrc, gsplit, rb.comb_w, rb.state, rb.tie_xor);
    // (planning inside the keep kernel's last workgroup was measured slower: 39 us vs 10 + 14 for the two launches)
    hipLaunchKernelGGL(route_plan_k, dim3(1), dim3(256), 0, s, rb.state, E, rb.tile_group, rb.tile_row0, rb.tile_nrows,
                       rb.max_tiles);
    hipLaunchKernelGGL(route_fill_k, dim3(cdiv(2 * Nsrc, 256 * 8)), dim3(256), 0, s, rb.idx, rb.comb_w, Nsrc, gsplit, rb.state,
                       rb.src_row, rb.dst_row);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
