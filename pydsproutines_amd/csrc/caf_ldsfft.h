// In-LDS power-of-two complex FFT for one "row" of N = 2^LOGN points (64 <= N <= 16384) held by N/16 threads with
// 16 points each: Stockham decimation-in-time, radix 16 (+ one final radix 2/4/8 pass when log2 N is not a multiple
// of four), in place in a padded LDS image, inverse kernel e^{+j 2 pi m n / N} (a forward transform is
// conj(IDFT(conj x))).  Shared by the fused per-delay correlator (caf_perdelay.hip) and the overlap-save FIR
// (caf_firos.hip).  Input registers v[t] = x[l + t N/16]; output registers hold X[pd_out_index(l, r)], which is the
// same SET of positions {l + u N/16} (pd_out_slot gives u), so two transforms chain without an LDS exchange.
#pragma once
#include "caf_fft_dev.h"

namespace caf {

constexpr int PD_TWN = 16384;  // twiddle table: e^{+j 2 pi q / 16384}, q = 0 .. 16383

__device__ __forceinline__ int pd_pad(int a) { return a + (a >> 4); }

// synchronisation among the threads of ONE row: rows of up to 64 threads live inside a wave, whose LDS operations
// execute in order -- no s_barrier at all; wider rows use the workgroup barrier
template <int LOGN>
__device__ __forceinline__ void pd_row_sync() {
    if ((1 << LOGN) / 16 <= 64)
        __builtin_amdgcn_wave_barrier();
    else
        __syncthreads();
}

template <int R>
__device__ __forceinline__ void pd_butterfly(float2* v) {
    if (R == 16) idft16(*reinterpret_cast<float2(*)[16]>(v));
    if (R == 8) idft8(*reinterpret_cast<float2(*)[8]>(v));
    if (R == 4) idft4(v[0], v[1], v[2], v[3]);
    if (R == 2) idft2(v[0], v[1]);
}

// One Stockham pass of radix R over the row image `buf` (N padded elements) for the thread with row-local id l:
// butterflies j = l + q * (N/16), q < 16/R.  FIRST: inputs come from `v` (registers) instead of LDS and there
// are no twiddles (Ns = 1).  LAST: outputs stay in `v` (natural index j + t * N/R for register q*R + t).
template <int LOGN, int R, int NS, bool FIRST, bool LAST>
__device__ __forceinline__ void pd_pass(float2* __restrict__ buf, const float2* __restrict__ tw, int l, float2 (&v)[16]) {
    constexpr int N = 1 << LOGN, NTR = N / 16, NB = 16 / R, STR = N / R;
    // Padded addresses are affine in the register index: pad(a + 16 m) = pad(a) + 17 m, so every access of a
    // butterfly is one base register plus an immediate offset (all strides below are multiples of 16, or the
    // base itself is).
    static_assert(FIRST || STR % 16 == 0, "read stride must be a multiple of 16");
    static_assert(LAST || NS == 1 || NS % 16 == 0, "write stride must be 1 or a multiple of 16");
    if (!FIRST) {
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const float2* src = buf + pd_pad(l + q * NTR);
#pragma unroll
            for (int t = 0; t < R; ++t) v[q * R + t] = src[t * (STR + STR / 16)];
        }
    }
    // every butterfly of the row has its inputs: the image may now be overwritten.  (Not needed in front of the
    // first pass: the previous row's last pass ends with this barrier and writes nothing afterwards.)
    if (!FIRST) pd_row_sync<LOGN>();
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const int j = l + q * NTR;
        const int k = j & (NS - 1);
        if (NS > 1) {
            // v[t] *= W_{NS*R}^{k t}: base from the table, powers by recurrence
            const float2 w1 = tw[k * (PD_TWN / (NS * R))];
            float2 p = w1;
            v[q * R + 1] = cmul(v[q * R + 1], p);
#pragma unroll
            for (int t = 2; t < R; ++t) {
                p = cmul(p, w1);
                v[q * R + t] = cmul(v[q * R + t], p);
            }
        }
        pd_butterfly<R>(&v[q * R]);
        if (!LAST) {
            const int j0 = ((j - k) * R) + k;  // (j / NS) * NS * R + k
            float2* dst = buf + pd_pad(j0);     // NS == 1: j0 = R j is a multiple of 16 and t < 16
#pragma unroll
            for (int t = 0; t < R; ++t) dst[NS == 1 ? t : t * (NS + NS / 16)] = v[q * R + t];
        }
    }
    if (!LAST) pd_row_sync<LOGN>();
}

// all passes for N = 2^LOGN; v: in = pass-1 inputs v[t] = p[l + t N/16], out = spectrum values at
// index out_index<LOGN>(l, reg)
template <int LOGN>
__device__ __forceinline__ void pd_fft(float2* __restrict__ buf, const float2* __restrict__ tw, int l, float2 (&v)[16]) {
    constexpr int A = LOGN / 4, RL = 1 << (LOGN % 4);  // A radix-16 passes, then one radix-RL pass if RL > 1
    static_assert(A >= 1 && A <= 3, "64 <= N <= 16384");
    if constexpr (A == 1) {
        pd_pass<LOGN, 16, 1, true, RL == 1>(buf, tw, l, v);
        if constexpr (RL > 1) pd_pass<LOGN, RL, 16, false, true>(buf, tw, l, v);
    } else if constexpr (A == 2) {
        pd_pass<LOGN, 16, 1, true, false>(buf, tw, l, v);
        pd_pass<LOGN, 16, 16, false, RL == 1>(buf, tw, l, v);
        if constexpr (RL > 1) pd_pass<LOGN, RL, 256, false, true>(buf, tw, l, v);
    } else {
        pd_pass<LOGN, 16, 1, true, false>(buf, tw, l, v);
        pd_pass<LOGN, 16, 16, false, false>(buf, tw, l, v);
        pd_pass<LOGN, 16, 256, false, RL == 1>(buf, tw, l, v);
        if constexpr (RL > 1) pd_pass<LOGN, RL, 4096, false, true>(buf, tw, l, v);
    }
}
// spectrum index held in register r of row-local thread l after pd_fft
template <int LOGN>
__device__ __forceinline__ int pd_out_index(int l, int r) {
    constexpr int N = 1 << LOGN, NTR = N / 16, RL = (LOGN % 4) ? (1 << (LOGN % 4)) : 16;
    const int q = r / RL, t = r - q * RL;  // last pass: butterfly q of the thread, output t
    return l + q * NTR + t * (N / RL);
}

// position u (element l + u N/16) held in register r after pd_fft: the register layout of the NEXT transform's input
template <int LOGN>
__device__ __forceinline__ constexpr int pd_out_slot(int r) {
    constexpr int RL = (LOGN % 4) ? (1 << (LOGN % 4)) : 16;
    return (r / RL) + (r % RL) * (16 / RL);
}

}  // namespace caf
