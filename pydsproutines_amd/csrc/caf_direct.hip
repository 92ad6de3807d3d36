// Direct (time-domain) hypothesis engine for templates with at most 64 non-zero samples -- composite templates
// (GroupXcorr, xcorrRoutines.py:852-954) whose groups cover only a few samples of a long span.
//
// Why it exists: the overlap-save engines form the correlation on whole blocks, so their absolute float32 error follows
// the BLOCK's energy; a template with one or two samples of support (QF^2 ~ 1 on anything, normalised by those few
// samples) then shows 3-6e-5 where ordinary templates show 1e-7.  With K <= 64 products per (delay, hypothesis) the
// definition itself is the cheaper and the exact way:
//     QF2[t][s][f] = | sum_k rx[s + n_k] * w_{t,f,k} |^2 / ( ||tmpl_t||^2 * sum_k |rx[s + n_k]|^2 ),
//     w_{t,f,k} = conj(u_t[n_k]) * exp(-j 2 pi nu_f n_k)   (float64 on the host, rounded once),
// which is what GroupXcorr.xcorr evaluates per delay (freqMat @ product, :917-954).  Four threads per delay, sixteen
// hypotheses each per pass (they share each sample load); multipliers from LDS; the window energy is the float64 sum over
// the K samples themselves (no prefix differences), so a window of zeros has energy exactly 0.
#include "caf_internal.h"

namespace caf {

constexpr int DIR_FCHUNK = 16;  // hypotheses per thread and pass
constexpr int DIR_ROWS = 128;   // delays per workgroup: (64 delay pairs) x (4 interleaved sub-chunks of 16 hypotheses)
constexpr int DIR_FPASS = 64;   // hypotheses per pass of a workgroup (= one 256-byte surface row segment per delay)
constexpr int DIR_TP = DIR_FPASS + 4;  // pitch of the value tile (floats)

// Workgroup = 128 consecutive delays; thread = (delay pair d, d + 64; sub-chunk cg of the hypotheses 4 j + cg), the four
// sub-chunks of a delay in adjacent lanes.  Per pass of 64 hypotheses: the multipliers of the pass are staged in LDS at an
// odd pitch (as VGPR operands an FMA costs 2.4 cycles; from SGPRs, which the wave-uniform scalar loads of the first
// version gave, 4.2; the four sub-chunks read four different banks), every thread accumulates 2 x 16 (delay,
// hypothesis) sums over the K samples -- one multiplier read per 8 FMAs: the LDS would bind at one per 4 -- and the
// 128 x 64 values leave through an LDS tile as WHOLE 256-byte row segments (the first version stored 64-byte pieces at a
// 256-byte pitch from each lane: 1.6 x the algorithmic HBM traffic).
__global__ __launch_bounds__(256) void k_direct_caf(const float2* __restrict__ rx, int64_t shift_start, int64_t num_shifts,
                                                    int32_t ntmpl, int32_t nfreq, int32_t nk, const int32_t* __restrict__ pos,
                                                    const float2* __restrict__ w, const float* __restrict__ tscale,
                                                    float* __restrict__ surface, float* __restrict__ row_max,
                                                    int32_t* __restrict__ row_arg, PeakRec* __restrict__ partial,
                                                    int64_t partial_per_tmpl) {
    __shared__ float2 s_c[DIR_FPASS * 65];          // multipliers of the pass: [hypothesis][k] at pitch nk | 1, nk <= 64
    __shared__ float s_t[DIR_ROWS][DIR_TP];         // the pass' values, delay-major
    __shared__ int32_t s_pos[64];
    __shared__ PeakRec s_w[4];
    const int tid = threadIdx.x;
    const int cg = tid & 3, dp = tid >> 2;  // sub-chunk, delay pair
    const int cp = nk | 1;                  // (odd pitch: the four sub-chunks' reads fall on different banks)
    const int64_t d0 = (int64_t)blockIdx.x * DIR_ROWS;
    const int64_t i0 = d0 + dp, i1 = d0 + dp + 64;
    const bool live0 = i0 < num_shifts, live1 = i1 < num_shifts;
    const float2* xs0 = rx + shift_start + (live0 ? i0 : 0);  // (dead delays read delay 0 of the call: in range)
    const float2* xs1 = rx + shift_start + (live1 ? i1 : 0);
    if (tid < nk) s_pos[tid] = pos[tid];
    __syncthreads();
    double e0 = 0.0, e1 = 0.0;
    for (int k0 = 0; k0 < nk; k0 += 8) {  // eight sample pairs in flight (one dependent load per iteration otherwise)
        float2 x0[8], x1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = s_pos[min(k0 + u, nk - 1)];
            x0[u] = xs0[p];
            x1[u] = xs1[p];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (k0 + u < nk) {
                e0 += (double)x0[u].x * (double)x0[u].x + (double)x0[u].y * (double)x0[u].y;
                e1 += (double)x1[u].x * (double)x1[u].x + (double)x1[u].y * (double)x1[u].y;
            }
    }
    const float inv0 = e0 > 0.0 ? (float)(1.0 / e0) : __builtin_nanf(""), inv1 = e1 > 0.0 ? (float)(1.0 / e1) : __builtin_nanf("");
    for (int t = 0; t < ntmpl; ++t) {
        const float ts = tscale[t];
        const float g0 = inv0 * ts, g1 = inv1 * ts;  // the engines' rounding: value * (1/energy * 1/||t||^2)
        float bv0 = -1.f, bv1 = -1.f;
        int32_t bi0 = 0, bi1 = 0;
        const float2* wt = w + (int64_t)t * nfreq * nk;
        for (int f0 = 0; f0 < nfreq; f0 += DIR_FPASS) {
            const int nf = min(DIR_FPASS, nfreq - f0);
            __syncthreads();  // the previous pass' tile and multipliers have been read
            for (int q = tid; q < nf * nk; q += 256) {
                const int fl = q / nk, k = q - fl * nk;
                s_c[fl * cp + k] = wt[(int64_t)f0 * nk + q];  // contiguous in memory: [f0 .. f0 + nf)[nk]
            }
            __syncthreads();
            float ar0[DIR_FCHUNK], ai0[DIR_FCHUNK], ar1[DIR_FCHUNK], ai1[DIR_FCHUNK];
#pragma unroll
            for (int j = 0; j < DIR_FCHUNK; ++j) ar0[j] = ai0[j] = ar1[j] = ai1[j] = 0.f;
            float2 nx0 = xs0[s_pos[0]], nx1 = xs1[s_pos[0]];
            for (int k = 0; k < nk; ++k) {
                const float2 x0 = nx0, x1 = nx1;
                const int kn = min(k + 1, nk - 1);  // the next pair of samples is in flight under this one's 128 FMAs
                nx0 = xs0[s_pos[kn]];
                nx1 = xs1[s_pos[kn]];
                // this thread's hypotheses of the pass: 4 j + cg (past the end: the last one, discarded below); all sixteen
                // multipliers are read before the first is used (one LDS latency per sample instead of sixteen), and every
                // product is ONE fused multiply-add into its sum
                float2 c[DIR_FCHUNK];
#pragma unroll
                for (int j = 0; j < DIR_FCHUNK; ++j) c[j] = s_c[min(4 * j + cg, nf - 1) * cp + k];
#pragma unroll
                for (int j = 0; j < DIR_FCHUNK; ++j) {
                    ar0[j] = __builtin_fmaf(-x0.y, c[j].y, __builtin_fmaf(x0.x, c[j].x, ar0[j]));
                    ai0[j] = __builtin_fmaf(x0.y, c[j].x, __builtin_fmaf(x0.x, c[j].y, ai0[j]));
                    ar1[j] = __builtin_fmaf(-x1.y, c[j].y, __builtin_fmaf(x1.x, c[j].x, ar1[j]));
                    ai1[j] = __builtin_fmaf(x1.y, c[j].x, __builtin_fmaf(x1.x, c[j].y, ai1[j]));
                }
            }
#pragma unroll
            for (int j = 0; j < DIR_FCHUNK; ++j) {
                const int fl = 4 * j + cg;
                if (fl < nf) {
                    const float v0 = (ar0[j] * ar0[j] + ai0[j] * ai0[j]) * g0, v1 = (ar1[j] * ar1[j] + ai1[j] * ai1[j]) * g1;
                    s_t[dp][fl] = v0;
                    s_t[dp + 64][fl] = v1;
                    if (v0 > bv0) {  // increasing hypothesis order: the first maximum wins; NaN never does
                        bv0 = v0;
                        bi0 = f0 + fl;
                    }
                    if (v1 > bv1) {
                        bv1 = v1;
                        bi1 = f0 + fl;
                    }
                }
            }
            if (surface) {
                __syncthreads();
                // 128 delays x nf values leave as rows: a wave writes the nf-float segment of one delay per instruction
                for (int r = tid >> 6; r < DIR_ROWS; r += 4) {
                    const int fl = tid & 63;
                    if (fl < nf && d0 + r < num_shifts) surface[((int64_t)t * num_shifts + d0 + r) * nfreq + f0 + fl] = s_t[r][fl];
                }
            }
        }
        // the four sub-chunks of a delay sit in adjacent lanes: highest value, lowest hypothesis on ties
#pragma unroll
        for (int o = 1; o <= 2; o <<= 1) {
            float ov = __shfl_xor(bv0, o, 64);
            int32_t oi = __shfl_xor(bi0, o, 64);
            if (ov > bv0 || (ov == bv0 && oi < bi0)) bv0 = ov, bi0 = oi;
            ov = __shfl_xor(bv1, o, 64);
            oi = __shfl_xor(bi1, o, 64);
            if (ov > bv1 || (ov == bv1 && oi < bi1)) bv1 = ov, bi1 = oi;
        }
        if (cg == 0) {
            if (live0 && row_max) row_max[(int64_t)t * num_shifts + i0] = bv0 < 0.f ? __builtin_nanf("") : bv0;  // (zero-energy window)
            if (live0 && row_arg) row_arg[(int64_t)t * num_shifts + i0] = bi0;
            if (live1 && row_max) row_max[(int64_t)t * num_shifts + i1] = bv1 < 0.f ? __builtin_nanf("") : bv1;
            if (live1 && row_arg) row_arg[(int64_t)t * num_shifts + i1] = bi1;
        }
        if (partial) {
            // the better of the thread's two delays (the lower one on ties), then the workgroup's
            PeakRec b;
            b.v = live0 ? bv0 : -2.f;
            b.delay = live0 ? (int32_t)(shift_start + i0) : 0x7fffffff;
            b.f = bi0;
            if (live1 && bv1 > b.v) {
                b.v = bv1;
                b.delay = (int32_t)(shift_start + i1);
                b.f = bi1;
            }
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
    const unsigned grid = (unsigned)((num_shifts + DIR_ROWS - 1) / DIR_ROWS);
    hipLaunchKernelGGL(k_direct_caf, dim3(grid), dim3(256), 0, st, rx, shift_start, num_shifts, ntmpl, nfreq, nk, pos, w, tscale,
                       surface, row_max, row_arg, partial, partial_per_tmpl);
}

}  // namespace caf
