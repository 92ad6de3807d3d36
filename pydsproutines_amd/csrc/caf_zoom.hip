// Device side of caf_zoom_czt (BASELINE config C5: coarse CAF -> top-k local maxima -> chirp-Z fine zoom), the
// single-call form of the reference's two-stage workflow (benchmarks/benchmark_czts.py:31-82 coarse xcorr then
// IppCZT32fc::runMany; pybinds/ippGroupXcorrCZT/GroupXcorrCZT.cpp:202-329; cztXcorr xcorrRoutines.py:413-457):
//   k_zoom_topk    k strongest candidates of the local-maxima list (value descending, index ascending), on the device
//   k_zoom_rows    per peak: rx[d:d+n] * conj(u) / (||window|| ||u||), rotated onto the common relative grid and
//                  multiplied by the Bluestein pre-chirp, zero-padded to nfft -- ONE launch for all peaks
//   (rocFFT forward, x fv, rocFFT inverse, x ww slice: caf_ops.hip)
//   k_zoom_finish  assembles the result table
// Nothing but the final k-row table ever needs to leave the device, and the host never waits in between.
#include "caf_internal.h"

namespace caf {

namespace {

// one workgroup: k rounds of "best remaining candidate"
__global__ __launch_bounds__(1024) void k_zoom_topk(const float* __restrict__ trace, const int32_t* __restrict__ cand,
                                                    const int32_t* __restrict__ cand_count, int32_t max_cand, int32_t k,
                                                    float* __restrict__ vals, int32_t* __restrict__ sel,
                                                    int32_t* __restrict__ sel_count) {
    __shared__ float s_v[16];
    __shared__ int32_t s_i[16];
    __shared__ int32_t s_win;
    const int tid = threadIdx.x;
    const int cnt = min(*cand_count, max_cand);
    for (int i = tid; i < cnt; i += 1024) vals[i] = trace[cand[i]];
    __syncthreads();
    int found = 0;
    for (int r = 0; r < k; ++r) {
        float bv = -INFINITY;
        int32_t bi = 0x7fffffff;
        for (int i = tid; i < cnt; i += 1024) {
            const float v = vals[i];
            if (v > bv) {  // candidates are in ascending index order: the first maximum wins
                bv = v;
                bi = i;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int32_t oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) {
                bv = ov;
                bi = oi;
            }
        }
        if ((tid & 63) == 0) {
            s_v[tid >> 6] = bv;
            s_i[tid >> 6] = bi;
        }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 16; ++w)
                if (s_v[w] > bv || (s_v[w] == bv && s_i[w] < bi)) {
                    bv = s_v[w];
                    bi = s_i[w];
                }
            s_win = (bi != 0x7fffffff && bv > -INFINITY) ? bi : -1;
            if (s_win >= 0) {
                sel[r] = cand[s_win];
                vals[s_win] = -INFINITY;
            } else {
                sel[r] = -1;
            }
        }
        __syncthreads();
        if (s_win >= 0) ++found;
        __syncthreads();
    }
    if (tid == 0) *sel_count = found;
}

// one workgroup per peak
__global__ __launch_bounds__(256) void k_zoom_rows(const float2* __restrict__ rx, const float2* __restrict__ uconj, int32_t n,
                                                   const float* __restrict__ tscale_t, const int32_t* __restrict__ row_arg,
                                                   const double* __restrict__ nu, const float2* __restrict__ aa,
                                                   const int32_t* __restrict__ sel, const int32_t* __restrict__ sel_count,
                                                   int64_t shift_start, int32_t nfft, float2* __restrict__ rows) {
    __shared__ double s_e[4];
    const int pk = blockIdx.x;
    float2* row = rows + (int64_t)pk * nfft;
    const int rel = pk < *sel_count ? sel[pk] : -1;
    if (rel < 0) {
        for (int t = threadIdx.x; t < nfft; t += 256) row[t] = make_float2(0.f, 0.f);
        return;
    }
    const float2* w = rx + shift_start + rel;
    double e = 0.0;
    for (int t = threadIdx.x; t < n; t += 256) {
        const float2 a = w[t];
        e += (double)a.x * a.x + (double)a.y * a.y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o, 64);
    if ((threadIdx.x & 63) == 0) s_e[threadIdx.x >> 6] = e;
    __syncthreads();
    e = s_e[0] + s_e[1] + s_e[2] + s_e[3];
    // 1 / (||window|| ||u||): the normalisation of multiplySlidesNormalised (multiplySlices.cu:190-211)
    const float inv = (float)(sqrt((double)tscale_t[0]) / sqrt(e));
    const double f0 = nu[row_arg[rel]];
    for (int t = threadIdx.x; t < nfft; t += 256) {
        float2 o = make_float2(0.f, 0.f);
        if (t < n) {
            const float2 a = w[t], b = uconj[t];
            const float pr = (a.x * b.x - a.y * b.y) * inv, pi = (a.x * b.y + a.y * b.x) * inv;
            // rotate by exp(-j 2 pi f0 t): the peak's grid f0 - span ... f0 + span becomes the shared relative grid
            double cyc = f0 * (double)t;
            cyc -= floor(cyc);
            double sn, cs;
            sincos(-2.0 * M_PI * cyc, &sn, &cs);
            const float rr = pr * (float)cs - pi * (float)sn, ri = pr * (float)sn + pi * (float)cs;
            const float2 c = aa[t];
            o = make_float2(rr * c.x - ri * c.y, rr * c.y + ri * c.x);
        }
        row[t] = o;
    }
}

__global__ void k_zoom_finish(const float* __restrict__ trace, const int32_t* __restrict__ row_arg,
                              const double* __restrict__ nu, const int32_t* __restrict__ sel,
                              const int32_t* __restrict__ sel_count, int32_t k, int64_t shift_start, double span, double step,
                              const uint32_t* __restrict__ fine_arg, const float* __restrict__ fine_max, int32_t cand_cap,
                              const int32_t* __restrict__ cand_count, int32_t* o_count, int32_t* o_delay, int32_t* o_cidx,
                              float* o_cqf2, int32_t* o_fidx, double* o_ffreq, float* o_fqf2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && o_count) *o_count = (*cand_count > cand_cap) ? -1 : *sel_count;  // -1: candidate list overflowed
    if (i >= k) return;
    const int rel = i < *sel_count ? sel[i] : -1;
    const bool ok = rel >= 0;
    const int ci = ok ? row_arg[rel] : 0;
    if (o_delay) o_delay[i] = ok ? (int32_t)(shift_start + rel) : -1;
    if (o_cidx) o_cidx[i] = ci;
    if (o_cqf2) o_cqf2[i] = ok ? trace[rel] : 0.f;
    if (o_fidx) o_fidx[i] = ok ? (int32_t)fine_arg[i] : 0;
    if (o_ffreq) o_ffreq[i] = ok ? (nu[ci] - span) + (double)fine_arg[i] * step : 0.0;
    if (o_fqf2) o_fqf2[i] = ok ? fine_max[i] : 0.f;
}

}  // namespace

void launch_zoom_topk(const float* trace, const int32_t* cand, const int32_t* cand_count, int32_t max_cand, int32_t k,
                      float* vals_scratch, int32_t* sel, int32_t* sel_count, hipStream_t st) {
    hipLaunchKernelGGL(k_zoom_topk, dim3(1), dim3(1024), 0, st, trace, cand, cand_count, max_cand, k, vals_scratch, sel,
                       sel_count);
}
void launch_zoom_rows(const float2* rx, const float2* uconj, int32_t n, const float* tscale_t, const int32_t* row_arg,
                      const double* nu, const float2* aa, const int32_t* sel, const int32_t* sel_count, int32_t k,
                      int64_t shift_start, int32_t nfft, float2* rows, hipStream_t st) {
    hipLaunchKernelGGL(k_zoom_rows, dim3((unsigned)k), dim3(256), 0, st, rx, uconj, n, tscale_t, row_arg, nu, aa, sel, sel_count,
                       shift_start, nfft, rows);
}
void launch_zoom_finish(const float* trace, const int32_t* row_arg, const double* nu, const int32_t* sel,
                        const int32_t* sel_count, int32_t k, int64_t shift_start, double span, double step,
                        const uint32_t* fine_arg, const float* fine_max, int32_t cand_overflow_cap, const int32_t* cand_count,
                        int32_t* o_count, int32_t* o_delay, int32_t* o_cidx, float* o_cqf2, int32_t* o_fidx, double* o_ffreq,
                        float* o_fqf2, hipStream_t st) {
    hipLaunchKernelGGL(k_zoom_finish, dim3((unsigned)((k + 63) / 64)), dim3(64), 0, st, trace, row_arg, nu, sel, sel_count, k,
                       shift_start, span, step, fine_arg, fine_max, cand_overflow_cap, cand_count, o_count, o_delay, o_cidx,
                       o_cqf2, o_fidx, o_ffreq, o_fqf2);
}

}  // namespace caf
