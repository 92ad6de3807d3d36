// Single-launch, run-to-run deterministic exclusive scan of per-tile totals (float64 sums or integer counts).
//
// A workgroup takes its tile from a ticket counter, so all predecessors of a tile are resident or finished and every
// wait below ends.  Tile totals A[t] and group totals S[g] (TS_GROUP consecutive tiles; published by the workgroup of the
// group's last tile) travel as single 64-bit relaxed agent-scope atomics -- value and "published" in one word, the
// all-ones pattern (the workspace is filled with 0xFF before the launch) meaning "not yet".  The offset of tile t in
// group g is  tree(S[0..g-1]) + tree(A[64 g .. t-1])  with fixed summation trees: unlike a decoupled look-back, which
// adds whatever mix of aggregates and inclusive prefixes it finds, the float64 rounding does not depend on timing.
// Workspace: ts_words(ntiles) 8-byte words.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace caf {

constexpr int TS_GROUP = 64;
constexpr uint64_t TS_EMPTY = ~0ull;

// words: A[ntiles], S[ngroups], then -- a 4 KB page of its own, away from the words everybody polls -- the ticket: the
// tickets are same-address device-scope atomics, served by the memory side one at a time (~18 ns each, measured), and
// whatever shares their channel queues behind them
__host__ __device__ inline int64_t ts_ticket_word(int64_t ntiles) {
    return ((ntiles + (ntiles + TS_GROUP - 1) / TS_GROUP + 511) / 512 + 1) * 512;
}
__host__ __device__ inline int64_t ts_words(int64_t ntiles) { return ts_ticket_word(ntiles) + 512; }

__device__ __forceinline__ uint64_t ts_bits(double v) {
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    return b == TS_EMPTY ? 0x7ff8000000000000ull : b;  // (a NaN total never reads as "not yet")
}
__device__ __forceinline__ uint64_t ts_bits(int64_t v) { return (uint64_t)v; }  // counts: >= 0
__device__ __forceinline__ void ts_value(uint64_t b, double& v) { v = __longlong_as_double((long long)b); }
__device__ __forceinline__ void ts_value(uint64_t b, int64_t& v) { v = (int64_t)b; }

__device__ __forceinline__ uint64_t ts_wait(const uint64_t* p) {
    uint64_t v;
    int spins = 0;
    while ((v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == TS_EMPTY) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > (1 << 21)) return 0;  // cannot happen (ticket order); bounds the wait regardless
    }
    return v;
}

template <typename T>
__device__ __forceinline__ T ts_wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// thread 0 of the workgroup; returns the tile index of this workgroup (share it through LDS)
__device__ __forceinline__ uint32_t ts_ticket(uint64_t* ws, uint32_t ntiles) {
    return (uint32_t)(__hip_atomic_fetch_add(ws + ts_ticket_word(ntiles), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull);
}

// Called by all 64 lanes of ONE wave with the tile's total (wave-uniform): publishes it and returns the sum of the
// totals of tiles 0 .. tile - 1.
template <typename T>
__device__ __forceinline__ T ts_exclusive(uint64_t* ws, uint32_t tile, uint32_t ntiles, T total) {
    const int lane = threadIdx.x & 63;
    uint64_t* A = ws;
    uint64_t* S = ws + ntiles;
    const uint32_t g = tile / TS_GROUP, r = tile % TS_GROUP;
    if (lane == 0) __hip_atomic_store(&A[tile], ts_bits(total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // the polls of the tile totals and of the first 64 group totals are in flight together (one round trip, not two)
    const bool want_a = (uint32_t)lane < r, want_s = (uint32_t)lane < g;
    uint64_t ba = TS_EMPTY, bs = TS_EMPTY;
    for (int spins = 0; spins < (1 << 21); ++spins) {  // (bounded: cannot run out, see the ticket order)
        if (want_a && ba == TS_EMPTY) ba = __hip_atomic_load(&A[g * TS_GROUP + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (want_s && bs == TS_EMPTY) bs = __hip_atomic_load(&S[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!((want_a && ba == TS_EMPTY) || (want_s && bs == TS_EMPTY))) break;
        __builtin_amdgcn_s_sleep(2);
    }
    T a = 0, sg = 0;
    if (want_a) ts_value(ba, a);
    if (want_s) ts_value(bs, sg);
    const T before_in_group = ts_wave_sum(a);
    if (r == TS_GROUP - 1) {  // the group's total: same tree over all of its tiles
        const T s = ts_wave_sum(lane == TS_GROUP - 1 ? total : a);
        if (lane == 0) __hip_atomic_store(&S[g], ts_bits(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (uint32_t k = 64 + lane; k < g; k += 64) {
        T v;
        ts_value(ts_wait(&S[k]), v);
        sg += v;
    }
    return ts_wave_sum(sg) + before_in_group;
}

}  // namespace caf
