// Direct (time-domain) hypothesis engine for templates with at most 64 non-zero samples -- composite templates
// (GroupXcorr, xcorrRoutines.py:852-954) whose groups cover only a few samples of a long span.
//
// Why it exists: the overlap-save engines form the correlation on whole blocks, so their absolute float32 error follows
// the BLOCK's energy; a template with one or two samples of support (QF^2 ~ 1 on anything, normalised by those few
// samples) then shows 3-6e-5 where ordinary templates show 1e-7.  With K <= 64 products per (delay, hypothesis) the
// definition itself is the cheaper and the exact way:
//     QF2[t][s][f] = | sum_k rx[s + n_k] * w_{t,f,k} |^2 / ( ||tmpl_t||^2 * sum_k |rx[s + n_k]|^2 ),
//     w_{t,f,k} = conj(u_t[n_k]) * exp(-j 2 pi nu_f n_k)   (float64 on the host, rounded once),
// which is what GroupXcorr.xcorr evaluates per delay (freqMat @ product, :917-954).  One thread per delay; the
// multipliers are wave-uniform (scalar loads), the samples run with the lane (coalesced, L1-resident across the
// hypothesis loop); sixteen hypotheses share each sample load.  The window energy is the float64 sum over the K
// samples themselves (no prefix differences), so a window of zeros has energy exactly 0.
#include "caf_internal.h"

namespace caf {

constexpr int DIR_FCHUNK = 16;  // hypotheses per sample load (and 64-byte surface segments per delay)

__global__ __launch_bounds__(256) void k_direct_caf(const float2* __restrict__ rx, int64_t shift_start, int64_t num_shifts,
                                                    int32_t ntmpl, int32_t nfreq, int32_t nk, const int32_t* __restrict__ pos,
                                                    const float2* __restrict__ w, const float* __restrict__ tscale,
                                                    float* __restrict__ surface, float* __restrict__ row_max,
                                                    int32_t* __restrict__ row_arg, PeakRec* __restrict__ partial,
                                                    int64_t partial_per_tmpl) {
    __shared__ PeakRec s_w[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < num_shifts;
    const float2* xs = rx + shift_start + (live ? i : 0);  // (dead lanes read delay 0 of the call: in range)
    double e = 0.0;
    for (int k = 0; k < nk; ++k) {
        const float2 x = xs[pos[k]];
        e += (double)x.x * (double)x.x + (double)x.y * (double)x.y;
    }
    const float inv = (float)(1.0 / e);
    for (int t = 0; t < ntmpl; ++t) {
        const float g = inv * tscale[t];  // the engines' rounding: value * (1/energy * 1/||t||^2)
        float bv = -1.f;
        int32_t bi = 0;
        const float2* wt = w + (int64_t)t * nfreq * nk;
        for (int f0 = 0; f0 < nfreq; f0 += DIR_FCHUNK) {
            float ar[DIR_FCHUNK], ai[DIR_FCHUNK];
#pragma unroll
            for (int j = 0; j < DIR_FCHUNK; ++j) ar[j] = ai[j] = 0.f;
            for (int k = 0; k < nk; ++k) {
                const float2 x = xs[pos[k]];
#pragma unroll
                for (int j = 0; j < DIR_FCHUNK; ++j) {
                    // (hypotheses past the end of the list reuse the last one: uniform, in range, discarded below)
                    const float2 c = wt[(int64_t)min(f0 + j, nfreq - 1) * nk + k];
                    ar[j] += x.x * c.x - x.y * c.y;
                    ai[j] += x.x * c.y + x.y * c.x;
                }
            }
#pragma unroll
            for (int j = 0; j < DIR_FCHUNK; ++j) {
                const int f = f0 + j;
                if (f < nfreq) {
                    const float val = (ar[j] * ar[j] + ai[j] * ai[j]) * g;
                    if (surface && live) surface[((int64_t)t * num_shifts + i) * nfreq + f] = val;
                    if (val > bv) {  // first maximum wins; NaN (zero-energy window) never does
                        bv = val;
                        bi = f;
                    }
                }
            }
        }
        if (live) {
            if (row_max) row_max[(int64_t)t * num_shifts + i] = bv;
            if (row_arg) row_arg[(int64_t)t * num_shifts + i] = bi;
        }
        if (partial) {
            PeakRec b;
            b.v = live ? bv : -2.f;
            b.delay = live ? (int32_t)(shift_start + i) : 0x7fffffff;
            b.f = bi;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                PeakRec r;
                r.v = __shfl_xor(b.v, o, 64);
                r.delay = __shfl_xor(b.delay, o, 64);
                r.f = __shfl_xor(b.f, o, 64);
                if (r.v > b.v || (r.v == b.v && r.delay < b.delay)) b = r;
            }
            __syncthreads();  // (s_w of the previous template has been read)
            if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = b;
            __syncthreads();
            if (threadIdx.x == 0) {
                for (int wv = 1; wv < 4; ++wv) {
                    const PeakRec r = s_w[wv];
                    if (r.v > b.v || (r.v == b.v && r.delay < b.delay)) b = r;
                }
                partial[(int64_t)t * partial_per_tmpl + blockIdx.x] = b;
            }
        }
    }
}

void launch_direct_caf(const float2* rx, int64_t shift_start, int64_t num_shifts, int32_t ntmpl, int32_t nfreq, int32_t nk,
                       const int32_t* pos, const float2* w, const float* tscale, float* surface, float* row_max,
                       int32_t* row_arg, PeakRec* partial, int64_t partial_per_tmpl, hipStream_t st) {
    const unsigned grid = (unsigned)((num_shifts + 255) / 256);
    hipLaunchKernelGGL(k_direct_caf, dim3(grid), dim3(256), 0, st, rx, shift_start, num_shifts, ntmpl, nfreq, nk, pos, w, tscale,
                       surface, row_max, row_arg, partial, partial_per_tmpl);
}

}  // namespace caf
