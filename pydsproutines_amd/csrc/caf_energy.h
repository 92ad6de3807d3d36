// Window energies from a float64 prefix of |x|^2, exact where a difference of prefix entries is not (caf_internal.h,
// CAF_ENERGY_RESOLVED): device helpers shared by every kernel that normalises by a window energy.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

namespace caf {

// (what the threshold is and why: caf_internal.h, "Window energies")
constexpr double CAF_ENERGY_RESOLVED = 9.313225746154785e-10;  // 2^-30

// where the chunk energies of a prefix over m samples start (launch_energy_prefix writes them there)
__device__ __forceinline__ const double* energy_chunks(const double* prefix, int64_t m) { return prefix + ((m + 2) & ~(int64_t)1); }

__device__ __forceinline__ double sample_energy(const float2 v) {
    return __builtin_fma((double)v.x, (double)v.x, (double)v.y * (double)v.y);
}

// sum_{a <= i < b} |x_i|^2 with no subtraction anywhere (0 <= a <= b <= the array's length); chunks[c] = the energy of
// samples 64 c .. 64 c + 63 relative to x
__device__ inline double window_energy_direct(const float2* __restrict__ x, const double* __restrict__ chunks, int64_t a, int64_t b) {
    double e = 0.0;
    const int64_t h = (a + 63) & ~(int64_t)63, t = b & ~(int64_t)63;
    if (h >= t) {  // no whole chunk inside
        for (int64_t i = a; i < b; ++i) e += sample_energy(x[i]);
        return e;
    }
    for (int64_t i = a; i < h; ++i) e += sample_energy(x[i]);
    for (int64_t c = h >> 6; c < (t >> 6); ++c) e += chunks[c];
    for (int64_t i = t; i < b; ++i) e += sample_energy(x[i]);
    return e;
}

// the window's energy: the prefix difference where it is resolved, the direct sum where it is not
__device__ __forceinline__ double window_energy(const double* __restrict__ prefix, const float2* __restrict__ x, int64_t xlen, int64_t a,
                                                int64_t b) {
    const double pb = prefix[b], e = pb - prefix[a];
    if (e > CAF_ENERGY_RESOLVED * pb) return e;
    return window_energy_direct(x, energy_chunks(prefix, xlen), a, b);
}

}  // namespace caf
