// gfx950 kernels of the per-delay (time-domain product) path and the stand-alone kernel-level ops.
// CDNA4 counterparts -- by semantics, not by code -- of the reference's
//   custom_kernels/multiplySlices.cu:113-216  slidingMultiplyNormalised
//   custom_kernels/multiplySlices.cu:251-399  multiTemplateSlidingDotProduct
//   custom_kernels/multiplySlices.cu:25-84    multiplySlicesWithIndexedRowsOptimistic
//   custom_kernels/complex_magn.cu:8-19       complex_magnSq_kernel<T,U>
//   custom_kernels/argmax.cu:93-153           multiArgmaxAbsRows_complex64
//   custom_kernels/filter.cu:196-347,374-438  movingAverage / multiMovingAverage / movingComplexSum
//   custom_kernels/filter.cu:9-181            filter_smtaps*  (lfilter semantics)
//   custom_kernels/upfirdn.cu:6-182           upfirdn_naive / upfirdn_sm
//   custom_kernels/peakfinding.cu:14-58       findLocalMaxima
//   custom_kernels/copying.cu:8-138, cupyExtensions.py:17-38   slice/group copies
// All are HBM-bound (or, for long FIRs, VALU-bound) elementwise / sliding-window work.
#include "caf_internal.h"
#include "caf_energy.h"

namespace caf {

// LDS staging loops: element t = tid, tid + 256, ... of `count`, value load(t), stored by store(t, value).  STG loads per
// thread are issued before the first of them is stored: written element by element (load, wait, store) the round trips of
// a thread's share stand one after the other in front of the workgroup's barrier.  Worth 7-10 % on the decimating filters
// (128 taps / 4 on 2^24 samples: 72 -> 65 us), nothing on the others -- several workgroups per CU cover one another -- and
// -4 % on k_upfirdn_poly, which keeps its plain loops (profiles/r04/ab_staging_loops.log).
template <int STG, typename Load, typename Store>
__device__ __forceinline__ void stage_batched(int count, Load&& load, Store&& store) {
    for (int t0 = threadIdx.x; t0 < count; t0 += 256 * STG) {
        decltype(load(0)) v[STG];
#pragma unroll
        for (int u = 0; u < STG; ++u)
            if (t0 + 256 * u < count) v[u] = load(t0 + 256 * u);
#pragma unroll
        for (int u = 0; u < STG; ++u)
            if (t0 + 256 * u < count) store(t0 + 256 * u, v[u]);
    }
}

// sum |x|^2 of a complex64 vector in float64 as NORM_PARTS partial sums (fixed assignment of elements to workgroups
// and a fixed summation order: the result does not depend on scheduling); the consumer adds the partials up.
// Replaces a blocking device-to-host copy + host loop in front of the per-delay path.
constexpr int NORM_PARTS = 128;
__global__ __launch_bounds__(256) void k_cutout_sumsq(const float2* __restrict__ x, int64_t n, double* __restrict__ parts) {
    __shared__ double s[4];
    double e = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)NORM_PARTS * 256) {
        const float2 a = x[i];
        e += (double)a.x * a.x + (double)a.y * a.y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o, 64);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = e;
    __syncthreads();
    if (threadIdx.x == 0) parts[blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}
__global__ void k_cutout_final(double* __restrict__ parts) {  // parts[NORM_PARTS] = sqrt(sum of the partials), fixed order
    double t = 0.0;
    for (int i = 0; i < NORM_PARTS; ++i) t += parts[i];
    parts[NORM_PARTS] = sqrt(t);
}
int cutout_norm_scratch_doubles() { return NORM_PARTS + 1; }
// returns the device address of ||x|| (valid once the two kernels have run on `st`)
const double* launch_cutout_norm(const float2* x, int64_t n, double* parts, hipStream_t st) {
    hipLaunchKernelGGL(k_cutout_sumsq, dim3(NORM_PARTS), dim3(256), 0, st, x, n, parts);
    hipLaunchKernelGGL(k_cutout_final, dim3(1), dim3(1), 0, st, parts);
    return parts + NORM_PARTS;
}

// ---------------------------------------------------------------------------------------
// z[i][t] = x[t] * y[s_i + t] / (sqrt(E_i) * coef),  s_i = start + i*step,
// E_i = sum_t |y[s_i + t]|^2 from the f64 prefix array (samples past the end of y read as 0).
// Rows whose window does not lie inside [0, ylen) are written as zeros when zero_oor != 0
// (IppXcorrFFT.cpp:125-130 semantics), otherwise they are computed with zero padding
// (multiplySlices.cu:147-163 semantics).
// ---------------------------------------------------------------------------------------
// A workgroup handles `rpw` consecutive rows (short rows: one 1/sqrt(E) in float64 per LANE instead of one per wave --
// for 1000-sample rows the float64 square root and division were three quarters of the instructions) and the
// blockIdx.x-th chunk of their samples.
constexpr int SM_MAX_RPW = 64;
__global__ __launch_bounds__(256) void k_sliding_multiply(const float2* __restrict__ x, int32_t xlen,
                                                          const float2* __restrict__ y, int64_t ylen,
                                                          const double* __restrict__ prefix, int64_t start,
                                                          int64_t step, double coef, int32_t zero_oor,
                                                          float2* __restrict__ z, const double* __restrict__ d_coef,
                                                          int64_t rows, int32_t rpw, int32_t rows_fastest) {
    __shared__ float s_inv[SM_MAX_RPW];
    // rows_fastest (long rows): the row groups are the fast grid dimension, so the workgroups in flight work on the SAME chunk of
    // x and of the (overlapping) windows of y for different rows and find it in the L2 / Infinity Cache -- with the chunks fastest
    // every row group streamed both arrays from HBM again (128 rows of 10^7 samples: 5.6 GB fetched for 0.16 GB of inputs)
    const uint32_t bx = rows_fastest ? blockIdx.y : blockIdx.x, by = rows_fastest ? blockIdx.x : blockIdx.y;
    const uint32_t gdx = rows_fastest ? gridDim.y : gridDim.x;
    const int64_t row0 = (int64_t)by * rpw;
    if ((int)threadIdx.x < rpw && row0 + threadIdx.x < rows) {
        const int64_t s = start + (row0 + threadIdx.x) * step;
        const bool oor = (s < 0) || (s + xlen > ylen);
        float inv = 0.f;
        if (!(oor && zero_oor)) {
            int64_t a = s < 0 ? 0 : (s > ylen ? ylen : s);
            int64_t b = s + xlen;
            b = b < 0 ? 0 : (b > ylen ? ylen : b);
            const double e = window_energy(prefix, y, ylen, a, b);  // (exact where the difference is not: caf_energy.h)
            inv = (float)(1.0 / (sqrt(e) * (d_coef ? coef * *d_coef : coef)));
        }
        s_inv[threadIdx.x] = inv;
    }
    __syncthreads();
    // four rows at a time: x[t] is loaded once for them, and with consecutive delays (step 1) the four y loads of a thread
    // are the neighbouring threads' lines -- per output 8 + 8 bytes came from L2 / the Infinity Cache, now about a quarter of
    // that (long rows, 128 x 10^7 outputs: 715 -> 665 us per batch; non-temporal stores and one contiguous chunk of t per
    // workgroup instead of the grid stride were measured too: no change / 10 % slower; round 4: two outputs per lane with 16-byte
    // stores: no change on 1430-sample rows, 10 % slower on 10^7-sample ones.  A write-only stream reaches 4.4 .. 6.1 TB/s on this
    // chip, scripts/ubench/stream_rw.hip: the 4.3 TB/s of the short rows are at that ceiling)
    for (int r = 0; r < rpw && row0 + r < rows; r += 4) {
        int64_t sq[4];
        bool zq[4], live[4];
        float iq[4];
        float2* zrow[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            live[q] = r + q < rpw && row0 + r + q < rows;
            sq[q] = start + (row0 + r + q) * step;
            zq[q] = ((sq[q] < 0) || (sq[q] + xlen > ylen)) && zero_oor;
            iq[q] = live[q] ? s_inv[r + q] : 0.f;
            zrow[q] = z + (row0 + r + q) * (int64_t)xlen;
        }
        for (int t = bx * 256 + threadIdx.x; t < xlen; t += gdx * 256) {
            const float2 a = x[t];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (!live[q]) continue;
                float2 o = make_float2(0.f, 0.f);
                const int64_t j = sq[q] + t;
                if (!zq[q] && j >= 0 && j < ylen) {
                    const float2 b = y[j];
                    o.x = (a.x * b.x - a.y * b.y) * iq[q];
                    o.y = (a.x * b.y + a.y * b.x) * iq[q];
                }
                zrow[q][t] = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Row post-processing after the row FFT: per row r of (rows, len) complex64
//   argmax[r] (first index of the maximum of |z|^2), max[r] (|z|^2 or |z|),
//   optional |z|^2 plane (float32), all scaled by `scale`.
// One workgroup per row.
// ---------------------------------------------------------------------------------------
// nan_empty: an all-NaN row reports (NaN, 0) -- the per-delay path's zero-energy window, the reference's pmax / norm / 0
// (xcorrRoutines.py:527-528) -- instead of the (0, 0) of the CUDA kernel's zero-initialised workspace
__global__ __launch_bounds__(256) void k_rows_argmax(const float2* __restrict__ z, int64_t len, int32_t use_normsq,
                                                     float scale, uint32_t* __restrict__ argmax,
                                                     float* __restrict__ maxv, float* __restrict__ plane, int32_t nan_empty) {
    __shared__ float s_v[4];
    __shared__ uint32_t s_i[4];
    const int64_t row = blockIdx.x;
    const float2* zr = z + row * len;
    float bv = -1.f;
    uint32_t bi = 0;
    for (int64_t t = threadIdx.x; t < len; t += 256) {
        const float2 a = zr[t];
        const float v = (a.x * a.x + a.y * a.y) * scale;
        if (plane) plane[row * len + t] = v;
        if (v > bv) {
            bv = v;
            bi = (uint32_t)t;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const uint32_t oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) {
            bv = ov;
            bi = oi;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        s_v[threadIdx.x >> 6] = bv;
        s_i[threadIdx.x >> 6] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (s_v[w] > bv || (s_v[w] == bv && s_i[w] < bi)) {
                bv = s_v[w];
                bi = s_i[w];
            }
        if (bv < 0.f) {  // empty or all-NaN row: the reference's zero-initialised workspace (argmax.cu:108-109)
            bv = (nan_empty && len > 0) ? __builtin_nanf("") : 0.f;
            bi = 0;
        }
        if (argmax) argmax[row] = bi;
        if (maxv) maxv[row] = use_normsq ? bv : sqrtf(bv);
    }
}

// Many short rows (the per-delay path's product rows: 10^5 rows of ~10^3 elements): one WAVE per row, four rows per
// workgroup -- no barrier, no LDS --, 16-byte loads (two elements per lane and instruction, four instructions in flight).
// A row that starts 8 bytes off a 16-byte boundary gives its first element to lane 0.  Per lane the indices are visited in
// increasing order, so the strict comparison keeps the first maximum; across lanes the lower index wins ties.
__global__ __launch_bounds__(256) void k_rows_argmax_wave(const float2* __restrict__ z, int64_t rows, int64_t len, int32_t use_normsq,
                                                          float scale, uint32_t* __restrict__ argmax, float* __restrict__ maxv,
                                                          float* __restrict__ plane, int32_t nan_empty) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;  // (wave-uniform)
    const float2* zr = z + row * len;
    float* pr = plane ? plane + row * len : nullptr;
    float bv = -1.f;
    uint32_t bi = 0;
    auto offer = [&](float2 a, int64_t t) {
        const float v = (a.x * a.x + a.y * a.y) * scale;
        if (pr) pr[t] = v;
        if (v > bv) {
            bv = v;
            bi = (uint32_t)t;
        }
    };
    const int head = (int)((reinterpret_cast<uintptr_t>(zr) >> 3) & 1);
    if (head && lane == 0 && len > 0) offer(zr[0], 0);
    const int64_t nv = len > head ? (len - head) >> 1 : 0;
    const float4* zv = reinterpret_cast<const float4*>(zr + head);
#pragma unroll 4
    for (int64_t p = lane; p < nv; p += 64) {
        const float4 q = zv[p];
        offer(make_float2(q.x, q.y), head + 2 * p);
        offer(make_float2(q.z, q.w), head + 2 * p + 1);
    }
    if (len > head && ((len - head) & 1) && lane == 0) offer(zr[len - 1], len - 1);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const uint32_t oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) {
            bv = ov;
            bi = oi;
        }
    }
    if (lane == 0) {
        if (bv < 0.f) {  // empty or all-NaN row: see k_rows_argmax
            bv = (nan_empty && len > 0) ? __builtin_nanf("") : 0.f;
            bi = 0;
        }
        if (argmax) argmax[row] = bi;
        if (maxv) maxv[row] = use_normsq ? bv : sqrtf(bv);
    }
}

// Rows too long for one workgroup each (cp_fastXcorr at N = 1e7: three workgroups scanning 1e7 elements took 17 ms per
// launch): the row is cut into chunks, one workgroup per (chunk, row) leaves a packed key
//   (float bits of the maximum) << 32 | (0xFFFFFFFF - index)      (values >= 0: the bits order like the floats; 0 = none)
// and a second small kernel takes the largest key per row: same result as k_rows_argmax (first index on ties).
__global__ __launch_bounds__(256) void k_rows_argmax_part(const float2* __restrict__ z, int64_t len, int64_t chunk, float scale,
                                                          unsigned long long* __restrict__ part, float* __restrict__ plane) {
    __shared__ unsigned long long s_k[4];
    const int64_t row = blockIdx.y, c = blockIdx.x;
    const float2* zr = z + row * len;
    const int64_t lo = c * chunk, hi = min(len, lo + chunk);
    float bv = -1.f;
    uint32_t bi = 0;
    for (int64_t t = lo + threadIdx.x; t < hi; t += 256) {
        const float2 a = zr[t];
        const float v = (a.x * a.x + a.y * a.y) * scale;
        if (plane) plane[row * len + t] = v;
        if (v > bv) {
            bv = v;
            bi = (uint32_t)t;
        }
    }
    unsigned long long key = bv < 0.f ? 0ull : (((unsigned long long)__float_as_uint(bv)) << 32) | (0xFFFFFFFFu - bi);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long ok = __shfl_xor(key, o, 64);
        key = ok > key ? ok : key;
    }
    if ((threadIdx.x & 63) == 0) s_k[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) key = s_k[w] > key ? s_k[w] : key;
        part[row * gridDim.x + c] = key;
    }
}
__global__ __launch_bounds__(64) void k_rows_argmax_fin(const unsigned long long* __restrict__ part, int32_t chunks,
                                                        int64_t rows, int32_t use_normsq, uint32_t* __restrict__ argmax,
                                                        float* __restrict__ maxv, int32_t nan_empty) {
    const int64_t row = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (row >= rows) return;
    unsigned long long key = 0ull;
    for (int c = 0; c < chunks; ++c) {
        const unsigned long long k = part[row * chunks + c];
        key = k > key ? k : key;
    }
    const float bv = key ? __uint_as_float((uint32_t)(key >> 32)) : nan_empty ? __builtin_nanf("") : 0.f;  // empty / all-NaN row: (0, 0) or (NaN, 0)
    if (argmax) argmax[row] = key ? 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull) : 0u;
    if (maxv) maxv[row] = use_normsq ? bv : sqrtf(bv);
}

// multiArgmax3d_uint32 (argmax.cu:11-81): per item, argmax over the last three dimensions of a
// (items, d1, d2, d3) uint32 array -> the three indices (+ the maximum).  First flat index on ties;
// an all-zero item reports (0, 0, 0) like the reference's zero-initialised workspace.
__global__ __launch_bounds__(256) void k_argmax3d_u32(const uint32_t* __restrict__ x, int32_t d1, int32_t d2, int32_t d3,
                                                      uint32_t* __restrict__ argmax, uint32_t* __restrict__ maxv) {
    __shared__ uint32_t s_v[4];
    __shared__ uint32_t s_i[4];
    const int64_t n = (int64_t)d1 * d2 * d3;
    const uint32_t* xi = x + (int64_t)blockIdx.x * n;
    uint32_t bv = 0, bi = 0;
    for (int64_t t = threadIdx.x; t < n; t += 256) {
        const uint32_t v = xi[t];
        if (v > bv) {
            bv = v;
            bi = (uint32_t)t;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t ov = __shfl_xor(bv, o, 64);
        const uint32_t oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) {
            bv = ov;
            bi = oi;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        s_v[threadIdx.x >> 6] = bv;
        s_i[threadIdx.x >> 6] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (s_v[w] > bv || (s_v[w] == bv && s_i[w] < bi)) {
                bv = s_v[w];
                bi = s_i[w];
            }
        argmax[blockIdx.x * 3 + 0] = bi / (uint32_t)(d2 * d3);
        argmax[blockIdx.x * 3 + 1] = (bi / (uint32_t)d3) % (uint32_t)d2;
        argmax[blockIdx.x * 3 + 2] = bi % (uint32_t)d3;
        if (maxv) maxv[blockIdx.x] = bv;
    }
}

// |x|^2, elementwise.  IN: 0 complex64, 1 complex128.  OUT: 0 float32, 1 float64.
template <typename TIn, typename TOut>
__global__ __launch_bounds__(256) void k_magnsq(const TIn* __restrict__ x, int64_t n, TOut* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const TIn v = x[i];
        out[i] = (TOut)(v.x * v.x + v.y * v.y);
    }
}

// ---------------------------------------------------------------------------------------
// Causal moving sum / mean of float32 (zeros in front), double accumulation (filter.cu:324-339).
// Each workgroup produces MA_TILE outputs from an LDS-staged window; a thread sums its first
// window directly and then slides, re-anchoring every MA_PER_THREAD outputs.
// ---------------------------------------------------------------------------------------
constexpr int MA_THREADS = 256;
constexpr int MA_PER_THREAD = 16;
constexpr int MA_TILE = MA_THREADS * MA_PER_THREAD;

__global__ __launch_bounds__(MA_THREADS) void k_moving_sum_prefix(const float* __restrict__ x, int64_t n,
                                                                  double* __restrict__ tile_sums) {
    // per-tile sums of x (float64) for the two-level prefix used by the moving sum
    __shared__ double s_part[MA_THREADS / 64];
    const int64_t base = (int64_t)blockIdx.x * MA_TILE + (int64_t)threadIdx.x * MA_PER_THREAD;
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < MA_PER_THREAD; ++j)
        if (base + j < n) acc += (double)x[base + j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < MA_THREADS / 64; ++w) t += s_part[w];
        tile_sums[blockIdx.x] = t;
    }
}

// prefix[i] = sum_{j<i} x[j] (float64), i in [0, n]
__global__ __launch_bounds__(MA_THREADS) void k_moving_prefix_write(const float* __restrict__ x, int64_t n,
                                                                    const double* __restrict__ tile_off,
                                                                    double* __restrict__ prefix) {
    __shared__ double s_wave[MA_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * MA_TILE + (int64_t)threadIdx.x * MA_PER_THREAD;
    double p[MA_PER_THREAD];
    double tot = 0.0;
#pragma unroll
    for (int j = 0; j < MA_PER_THREAD; ++j) {
        p[j] = tot;
        if (base + j < n) tot += (double)x[base + j];
    }
    double incl = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double u = __shfl_up(incl, o, 64);
        if (lane >= o) incl += u;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    double off = tile_off[blockIdx.x] + (incl - tot);
    for (int w = 0; w < wave; ++w) off += s_wave[w];
#pragma unroll
    for (int j = 0; j < MA_PER_THREAD; ++j)
        if (base + j <= n) prefix[base + j] = off + p[j];
}

// out[i] = (prefix[i+1] - prefix[max(0, i+1-L)]) [/ L]
__global__ __launch_bounds__(256) void k_moving_from_prefix(const double* __restrict__ prefix, int64_t n, int32_t L,
                                                            int32_t sum_instead, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t lo = i + 1 - L;
    const double s = prefix[i + 1] - prefix[lo < 0 ? 0 : lo];
    out[i] = sum_instead ? (float)s : (float)(s / (double)L);
}

// Causal moving sum / mean in ONE launch for windows up to MAT_MAXL: a workgroup covers MAT_SPAN consecutive samples
// (its outputs preceded by a halo of the window, zeros before the start), each thread 8 of them straight from two
// 16-byte loads; their float64 inclusive prefix is built in registers (thread, wave shuffle, wave totals) and only the
// prefix goes through LDS, once: out[i] = P[i] - P[i - L].  No global prefix array, no scratch; 4 B read + 4 B written
// per sample plus the halo.  (The form this replaces staged the samples in LDS as well and read them twice: six LDS
// operations per sample against three, 47 us against the time in profiles/ for 2^24 samples.)
constexpr int MAT_NT = 256, MAT_PER = 8;
constexpr int MAT_SPAN = MAT_NT * MAT_PER;
constexpr int MAT_MAXL = 1024;
// prefix through slot t lives at s_p[mat_slot(t + 1)]: one pad per 8 entries, so that the 8-consecutive writes of a
// thread (stride 9 doubles across lanes) and the consecutive reads of the output loop both spread over the banks
__device__ __forceinline__ int mat_slot(int t) { return t + (t >> 3); }
__host__ __device__ inline int mat_halo(int L) { return (L - 1 + 3) & ~3; }       // slots before the first output
__host__ __device__ inline int mat_outputs(int L) { return MAT_SPAN - mat_halo(L); }  // outputs per workgroup (multiple of 4)

__global__ __launch_bounds__(MAT_NT) void k_moving_tile(const float* __restrict__ x, int64_t n, int32_t L, int32_t sum_instead,
                                                        float* __restrict__ out) {
    __shared__ double s_p[MAT_SPAN + MAT_SPAN / 8 + 2];
    __shared__ double s_wave[MAT_NT / 64];
    const float* xr = x + (int64_t)blockIdx.y * n;
    float* outr = out + (int64_t)blockIdx.y * n;
    const int H = mat_halo(L), T = MAT_SPAN - H;
    const int64_t i0 = (int64_t)blockIdx.x * T;  // first output of the workgroup; slot t <-> sample i0 - H + t
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t0 = threadIdx.x * MAT_PER;
    const int64_t j0 = i0 - H + t0;
    float v[MAT_PER];
    if (j0 >= 0 && j0 + MAT_PER <= n && (reinterpret_cast<uintptr_t>(xr + j0) & 15) == 0) {
        const float4 a = *reinterpret_cast<const float4*>(xr + j0), b = *reinterpret_cast<const float4*>(xr + j0 + 4);
        v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
    } else {
#pragma unroll
        for (int k = 0; k < MAT_PER; ++k) v[k] = (j0 + k >= 0 && j0 + k < n) ? xr[j0 + k] : 0.f;
    }
    double pl[MAT_PER];
    double tot = 0.0;
#pragma unroll
    for (int k = 0; k < MAT_PER; ++k) pl[k] = (tot += (double)v[k]);
    double incl = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double u = __shfl_up(incl, o, 64);
        if (lane >= o) incl += u;
    }
    if (lane == 63) s_wave[wave] = incl;
    if (threadIdx.x == 0) s_p[0] = 0.0;
    __syncthreads();
    double off = incl - tot;
    for (int w = 0; w < wave; ++w) off += s_wave[w];
#pragma unroll
    for (int k = 0; k < MAT_PER; ++k) s_p[mat_slot(t0 + k + 1)] = off + pl[k];
    __syncthreads();
    for (int l = threadIdx.x; l < T; l += MAT_NT) {
        const int64_t i = i0 + l;
        if (i >= n) break;
        const int t = H + l;  // window of output i: slots t - L + 1 .. t
        const double s = s_p[mat_slot(t + 1)] - s_p[mat_slot(t + 1 - L)];
        outr[i] = sum_instead ? (float)s : (float)(s / (double)L);
    }
}

// valid-only forward moving complex sum -> |sum|^2 (filter.cu:374-438): direct O(L) per output in f64
// staged through LDS (L is small in the reference's use: symbol-length sums).
__global__ __launch_bounds__(256) void k_complex_moving_sum(const float2* __restrict__ x, int64_t n, int32_t L,
                                                            float* __restrict__ out) {
    extern __shared__ float2 s_x[];  // 256*CMS_PER + L - 1 samples
    constexpr int PER = 8;
    const int64_t o0 = (int64_t)blockIdx.x * 256 * PER;
    const int64_t nout = n - L + 1;
    const int span = 256 * PER + L - 1;
    stage_batched<8>(span, [&](int t) { const int64_t j = o0 + t; return (j < n) ? x[j] : make_float2(0.f, 0.f); },
                     [&](int t, float2 v) { s_x[t] = v; });
    __syncthreads();
    const int l0 = threadIdx.x * PER;
    double sr = 0.0, si = 0.0;
    for (int k = 0; k < L; ++k) {
        sr += (double)s_x[l0 + k].x;
        si += (double)s_x[l0 + k].y;
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int64_t o = o0 + l0 + j;
        if (o < nout) out[o] = (float)(sr * sr + si * si);
        sr += (double)s_x[l0 + j + L].x - (double)s_x[l0 + j].x;
        si += (double)s_x[l0 + j + L].y - (double)s_x[l0 + j].y;
    }
}

// ---------------------------------------------------------------------------------------
// multiTemplateSlidingDotProduct: per slide k, best template i of
//   |sum_t T_i[t] x[k+t]|^2 / E_i / ||x[k:k+L]||^2   (first template wins ties; all-zero -> (0, 0)).
// One workgroup owns MT_SLIDES consecutive slides; the x section and one template at a time live in
// LDS; each wave computes whole dot products (lanes stride over t, shuffle reduce), so no block-wide
// barrier per slide as in the reference.
// ---------------------------------------------------------------------------------------
constexpr int MT_SLIDES = 64;

__global__ __launch_bounds__(256) void k_multi_template_dot(const float2* __restrict__ tm, const float* __restrict__ te,
                                                            int32_t ntm, int32_t L, const float2* __restrict__ x,
                                                            int64_t xlen, const double* __restrict__ prefix,
                                                            int64_t start, int64_t nslides, int32_t* __restrict__ tidx,
                                                            float* __restrict__ qf2) {
    extern __shared__ float2 s_mem[];
    float2* s_t = s_mem;          // L
    float2* s_xs = s_mem + L;     // MT_SLIDES + L - 1
    const int64_t k0 = (int64_t)blockIdx.x * MT_SLIDES;
    const int span = MT_SLIDES + L - 1;
    stage_batched<8>(span, [&](int t) { const int64_t j = start + k0 + t; return (j < xlen) ? x[j] : make_float2(0.f, 0.f); },
                     [&](int t, float2 v) { s_xs[t] = v; });
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int PER_WAVE = MT_SLIDES / 4;
    float bv[PER_WAVE];
    int32_t bi[PER_WAVE];
#pragma unroll
    for (int r = 0; r < PER_WAVE; ++r) {
        bv[r] = 0.f;
        bi[r] = 0;
    }
    for (int i = 0; i < ntm; ++i) {
        __syncthreads();
        for (int t = threadIdx.x; t < L; t += 256) s_t[t] = tm[(int64_t)i * L + t];
        __syncthreads();
        const float inv_te = 1.0f / te[i];
#pragma unroll
        for (int r = 0; r < PER_WAVE; ++r) {
            const int k = wave * PER_WAVE + r;
            if (k0 + k >= nslides) break;
            float ar = 0.f, ai = 0.f;
            for (int t = lane; t < L; t += 64) {
                const float2 a = s_t[t], b = s_xs[k + t];
                ar += a.x * b.x - a.y * b.y;
                ai += a.x * b.y + a.y * b.x;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                ar += __shfl_xor(ar, o, 64);
                ai += __shfl_xor(ai, o, 64);
            }
            const int64_t s = start + k0 + k;
            int64_t e1 = s + L;
            if (e1 > xlen) e1 = xlen;
            const float e = (float)(prefix[e1] - prefix[s]);
            const float v = (ar * ar + ai * ai) * inv_te / e;
            if (v > bv[r]) {
                bv[r] = v;
                bi[r] = i;
            }
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < PER_WAVE; ++r) {
            const int64_t k = k0 + wave * PER_WAVE + r;
            if (k < nslides) {
                tidx[k] = bi[r];
                qf2[k] = bv[r];
            }
        }
    }
}

// Register-tiled form of the above for templates up to MTR_MAXL samples: a thread owns MTR_R consecutive slides and
// keeps their MTR_R-sample window of x in registers (it slides by one sample per template tap; the tap loop is
// unrolled by MTR_R so that the window rotates through fixed register names), so a tap costs one LDS sample read
// and one broadcast tap read for MTR_R complex MACs, where the kernel above reads both operands per MAC and
// shuffle-reduces every dot product.  x window stored transposed (e % MTR_R major) as in k_fir_fast; templates
// zero-padded to a multiple of MTR_R.
constexpr int MTR_R = 8;
constexpr int MTR_SLIDES = 256 * MTR_R;
constexpr int MTR_MAXL = 2048;

__global__ __launch_bounds__(256) void k_multi_template_dot_rt(const float2* __restrict__ tm, const float* __restrict__ te,
                                                               int32_t ntm, int32_t L, const float2* __restrict__ x,
                                                               int64_t xlen, const double* __restrict__ prefix,
                                                               int64_t start, int64_t nslides,
                                                               int32_t* __restrict__ tidx, float* __restrict__ qf2) {
    extern __shared__ float2 s_mtr[];
    const int Lp = (L + MTR_R - 1) / MTR_R * MTR_R;
    const int span = MTR_SLIDES + Lp;
    const int pitch = span / MTR_R + 1;
    float2* s_t = s_mtr;        // Lp
    float2* s_xs = s_mtr + Lp;  // MTR_R rows of `pitch`
    const int64_t k0 = (int64_t)blockIdx.x * MTR_SLIDES;
    stage_batched<8>(span, [&](int t) { const int64_t j = start + k0 + t; return (j < xlen) ? x[j] : make_float2(0.f, 0.f); },
                     [&](int t, float2 v) { s_xs[(t % MTR_R) * pitch + t / MTR_R] = v; });
    const int l0 = threadIdx.x * MTR_R;
    float ewin[MTR_R], bv[MTR_R];  // ewin: ||x[k:k+L]||^2 per slide (1 for slides past the end)
    int32_t bi[MTR_R];
#pragma unroll
    for (int r = 0; r < MTR_R; ++r) {
        const int64_t s = start + k0 + l0 + r;
        float e = 1.f;
        if (k0 + l0 + r < nslides) {
            int64_t e1 = s + L;
            if (e1 > xlen) e1 = xlen;
            e = (float)(prefix[e1] - prefix[s]);
        }
        ewin[r] = e;
        bv[r] = 0.f;
        bi[r] = 0;
    }
    for (int i = 0; i < ntm; ++i) {
        __syncthreads();  // previous template consumed (and, first time, the x window written)
        for (int t = threadIdx.x; t < Lp; t += 256) s_t[t] = t < L ? tm[(int64_t)i * L + t] : make_float2(0.f, 0.f);
        __syncthreads();
        float2 acc[MTR_R], win[MTR_R];
#pragma unroll
        for (int r = 0; r < MTR_R; ++r) {
            acc[r] = make_float2(0.f, 0.f);
            win[r] = s_xs[r * pitch + threadIdx.x];  // e = l0 + r
        }
        for (int t0 = 0; t0 < Lp; t0 += MTR_R) {
#pragma unroll
            for (int tt = 0; tt < MTR_R; ++tt) {
                const float2 a = s_t[t0 + tt];
                // slide r at tap t reads sample l0 + r + t, held in slot (r + tt) mod R
#pragma unroll
                for (int r = 0; r < MTR_R; ++r) {
                    const float2 b = win[(r + tt) % MTR_R];
                    acc[r].x += a.x * b.x - a.y * b.y;
                    acc[r].y += a.x * b.y + a.y * b.x;
                }
                // sample l0 + t is done; slot tt takes l0 + t + R  (row tt, column tid + (t0 + R) / R)
                win[tt] = s_xs[tt * pitch + threadIdx.x + t0 / MTR_R + 1];
            }
        }
        const float inv_te = 1.0f / te[i];
#pragma unroll
        for (int r = 0; r < MTR_R; ++r) {
            const float v = (acc[r].x * acc[r].x + acc[r].y * acc[r].y) * inv_te / ewin[r];
            if (v > bv[r]) {
                bv[r] = v;
                bi[r] = i;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < MTR_R; ++r) {
        const int64_t k = k0 + l0 + r;
        if (k < nslides) {
            tidx[k] = bi[r];
            qf2[k] = bv[r];
        }
    }
}

// out[i][t] = rows[row_idx[i]][t] * x[slice_start[i] + t] for t < slice_lens[i] (0 beyond), t < slice_len
__global__ __launch_bounds__(256) void k_multiply_indexed_rows(const float2* __restrict__ x, int64_t xlen,
                                                               const float2* __restrict__ rows, int32_t row_len,
                                                               const int32_t* __restrict__ slice_start,
                                                               const int32_t* __restrict__ slice_lens,
                                                               const int32_t* __restrict__ row_idx, int32_t slice_len,
                                                               float2* __restrict__ out) {
    const int64_t i = blockIdx.y;
    const float2* r = rows + (int64_t)row_idx[i] * row_len;
    const int64_t s0 = slice_start[i];
    const int li = slice_lens ? min(slice_lens[i], row_len) : min(slice_len, row_len);
    for (int t = blockIdx.x * 256 + threadIdx.x; t < slice_len; t += gridDim.x * 256) {
        const int64_t j = s0 + t;
        float2 v = make_float2(0.f, 0.f);
        if (t < li && j >= 0 && j < xlen) {
            const float2 a = r[t], b = x[j];
            v = make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
        }
        out[i * slice_len + t] = v;
    }
}

// generic gather of equal-length slices: out[i][t] = x[starts[i] + t]  (or start0 + i*inc when starts == NULL)
// starts_stride = 2 reads the start column of an (N, 2) [start, end) bounds array and limits row i to end-start.
__global__ __launch_bounds__(256) void k_copy_slices(const float2* __restrict__ x, int64_t xlen,
                                                     const int32_t* __restrict__ starts, int32_t starts_stride,
                                                     int64_t start0, int64_t inc, int32_t len,
                                                     float2* __restrict__ out) {
    const int64_t i = blockIdx.y;
    const int64_t s0 = starts ? (int64_t)starts[i * starts_stride] : start0 + i * inc;
    const int li = (starts && starts_stride == 2) ? min(len, starts[i * 2 + 1] - starts[i * 2]) : len;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < len; t += gridDim.x * 256) {
        const int64_t j = s0 + t;
        out[i * len + t] = (t < li && j >= 0 && j < xlen) ? x[j] : make_float2(0.f, 0.f);
    }
}

// copy groups: y[ys[b] + i] = x[xs[b] + i], i < len[b]   (cupyExtensions.py:17-38)
__global__ __launch_bounds__(256) void k_copy_groups(const float2* __restrict__ x, float2* __restrict__ y,
                                                     const int32_t* __restrict__ xs, const int32_t* __restrict__ ys,
                                                     const int32_t* __restrict__ lens) {
    const int b = blockIdx.x;
    const int64_t xo = xs[b], yo = ys[b];
    for (int i = threadIdx.x; i < lens[b]; i += 256) y[yo + i] = x[xo + i];
}

// findLocalMaxima (peakfinding.cu:14-58 predicate: above min_height and above both neighbours, zeros beyond the ends):
// ordered (ascending index) compaction that reads x ONCE.  Launch 1: a thread owns LM_PER consecutive samples
// (neighbours from the adjacent lanes), keeps its flags as a bit mask (n / 8 bytes of scratch) and the workgroup adds up
// the tile's count.  Launch 2 never touches x: a workgroup sums the counts of the tiles before its own (or reads the
// scanned counts when there are many tiles), ranks its masks with a workgroup scan and writes the indices; tiles
// without a maximum leave at once.  (A single-launch form with a ticketed look-back, scripts/ubench/tilescan_model.hip,
// costs more than this second launch: same-address tickets are served at ~10 ns each.)
constexpr int LM_NT = 1024, LM_PER = 16;
constexpr int LM_TILE = LM_NT * LM_PER;
constexpr int LM_DIRECT_TILES = 4096;  // up to here launch 2 adds the preceding counts itself (<= 16 KB per workgroup)

template <bool ALIGNED>
__global__ __launch_bounds__(LM_NT) void k_local_max_flags(const float* __restrict__ x, int64_t n, float min_height,
                                                           uint16_t* __restrict__ masks, int32_t* __restrict__ tile_count) {
    __shared__ int32_t s_w[LM_NT / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * LM_TILE + (int64_t)threadIdx.x * LM_PER;
    float v[LM_PER];
    if (ALIGNED && base + LM_PER <= n) {
#pragma unroll
        for (int j = 0; j < LM_PER; j += 4) {
            const float4 q = *reinterpret_cast<const float4*>(x + base + j);
            v[j] = q.x, v[j + 1] = q.y, v[j + 2] = q.z, v[j + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < LM_PER; ++j) v[j] = base + j < n ? x[base + j] : 0.f;
    }
    float left = __shfl_up(v[LM_PER - 1], 1, 64), right = __shfl_down(v[0], 1, 64);
    if (lane == 0) left = base > 0 && base - 1 < n ? x[base - 1] : 0.f;
    if (lane == 63) right = base + LM_PER < n ? x[base + LM_PER] : 0.f;
    uint32_t mask = 0;
#pragma unroll
    for (int j = 0; j < LM_PER; ++j) {
        const float l = j ? v[j - 1] : left, r = j + 1 < LM_PER ? v[j + 1] : right;
        if (base + j < n && v[j] > min_height && v[j] > l && v[j] > r) mask |= 1u << j;
    }
    masks[(int64_t)blockIdx.x * LM_NT + threadIdx.x] = (uint16_t)mask;
    int c = __popc(mask);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane == 0) s_w[wave] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
#pragma unroll
        for (int w = 0; w < LM_NT / 64; ++w) t += s_w[w];
        tile_count[blockIdx.x] = t;
    }
}

// in place: tile_count[t] -> number of maxima before tile t; tile_count[ntiles] = the total (many tiles only)
__global__ __launch_bounds__(1024) void k_local_max_scan(int32_t* __restrict__ tile_count, int64_t ntiles) {
    __shared__ int32_t s_wave[16];
    __shared__ int32_t s_base;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t c0 = 0; c0 < ntiles; c0 += 1024) {
        const int64_t t = c0 + threadIdx.x;
        const int v = t < ntiles ? tile_count[t] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(incl, o, 64);
            if (lane >= o) incl += u;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wave; ++w) off += s_wave[w];
        if (t < ntiles) tile_count[t] = off + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_base = off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_count[ntiles] = s_base;
}

__global__ __launch_bounds__(LM_NT) void k_local_max_write(const uint16_t* __restrict__ masks, const int32_t* __restrict__ tile_count,
                                                           int32_t scanned, int32_t max_out, int32_t* __restrict__ idx,
                                                           int32_t* __restrict__ count) {
    __shared__ int32_t s_w[LM_NT / 64];
    __shared__ int32_t s_before;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool last = blockIdx.x == gridDim.x - 1;
    int mine, before;
    if (scanned) {
        before = tile_count[blockIdx.x];
        mine = tile_count[blockIdx.x + 1] - before;
        if (last && threadIdx.x == 0) *count = before + mine;
        if (mine == 0) return;  // (uniform)
    } else {
        mine = tile_count[blockIdx.x];
        if (mine == 0 && !last) return;  // (uniform)
        int c = 0;
        for (int t = threadIdx.x; t < (int)blockIdx.x; t += LM_NT) c += tile_count[t];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if (lane == 0) s_w[wave] = c;
        __syncthreads();
        if (threadIdx.x == 0) {
            int t = 0;
#pragma unroll
            for (int w = 0; w < LM_NT / 64; ++w) t += s_w[w];
            s_before = t;
            if (last) *count = t + mine;
        }
        __syncthreads();
        before = s_before;
        if (mine == 0) return;
        __syncthreads();  // (s_w is reused below)
    }
    uint32_t mask = masks[(int64_t)blockIdx.x * LM_NT + threadIdx.x];
    const int c = __popc(mask);
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(incl, o, 64);
        if (lane >= o) incl += u;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    int off = before + (incl - c);
    for (int w = 0; w < wave; ++w) off += s_w[w];
    const int64_t base = (int64_t)blockIdx.x * LM_TILE + (int64_t)threadIdx.x * LM_PER;
    while (mask) {
        const int j = __ffs(mask) - 1;
        mask &= mask - 1;
        if (off < max_out) idx[off] = (int32_t)(base + j);
        ++off;
    }
}

// out[i] = x[idx[i]] for 4-byte elements (values / arguments of the candidate peaks without copying whole traces)
__global__ __launch_bounds__(256) void k_gather_b32(const uint32_t* __restrict__ x, int64_t xlen,
                                                    const int32_t* __restrict__ idx, int64_t n,
                                                    uint32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t j = idx[i];
    out[i] = (j >= 0 && j < xlen) ? x[j] : 0u;
}

// out[i] = (double) x[idx ? idx[i] : i]: float32 traces into the float64 device arrays the reference's GPU entry
// points return (xc = cp.zeros(shifts.size), xcorrRoutines.py:1198-1203) without a host round trip
__global__ __launch_bounds__(256) void k_gather_f32_f64(const float* __restrict__ x, int64_t xlen,
                                                        const int32_t* __restrict__ idx, int64_t n,
                                                        double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t j = idx ? (int64_t)idx[i] : i;
    out[i] = (j >= 0 && j < xlen) ? (double)x[j] : 0.0;
}

// ---------------------------------------------------------------------------------------
// FIR == scipy.signal.lfilter(taps, 1, x) on complex64 with real float32 taps, optional carried-in
// history (`delay` = the dlen samples preceding x) and decimation out[k] = y[k*dsr + phase].
// Taps and the input window of the tile are staged in LDS.
// ---------------------------------------------------------------------------------------
constexpr int FIR_TILE = 1024;  // outputs (before decimation) per workgroup

__global__ __launch_bounds__(256) void k_fir(const float2* __restrict__ x, int64_t n, const float* __restrict__ taps,
                                             int32_t ntaps, const float2* __restrict__ delay, int32_t dlen,
                                             int32_t dsr, int32_t phase, float2* __restrict__ out, int64_t nout) {
    extern __shared__ float s_fir[];
    float* s_taps = s_fir;                                              // ntaps
    float2* s_in = reinterpret_cast<float2*>(s_fir + ((ntaps + 1) & ~1));  // FIR_TILE + ntaps - 1
    const int64_t i0 = (int64_t)blockIdx.x * FIR_TILE;  // first un-decimated output index of the tile
    for (int t = threadIdx.x; t < ntaps; t += 256) s_taps[t] = taps[t];
    const int span = FIR_TILE + ntaps - 1;
    stage_batched<8>(
        span,
        [&](int t) {
            const int64_t j = i0 - (ntaps - 1) + t;  // input index
            float2 v = make_float2(0.f, 0.f);
            if (j >= 0) {
                if (j < n) v = x[j];
            } else if (delay && -j <= dlen) {
                v = delay[dlen + j];
            }
            return v;
        },
        [&](int t, float2 v) { s_in[t] = v; });
    __syncthreads();
    for (int l = threadIdx.x; l < FIR_TILE; l += 256) {
        const int64_t i = i0 + l;
        if (i >= n) break;
        if (dsr > 1 && ((i - phase) % dsr != 0 || i < phase)) continue;
        float ar = 0.f, ai = 0.f;
        // y[i] = sum_k taps[k] x[i-k];  x[i-k] sits at s_in[l + ntaps-1 - k]
        const float2* w = s_in + l + ntaps - 1;
        for (int k = 0; k < ntaps; ++k) {
            const float c = s_taps[k];
            ar += c * w[-k].x;
            ai += c * w[-k].y;
        }
        const int64_t o = (i - phase) / dsr;
        if (o < nout) out[o] = make_float2(ar, ai);
    }
}

// Undecimated FIR, register-tiled: a thread produces FIRF_R consecutive outputs from a sliding window that lives
// in registers, so every tap costs one LDS read of a new sample + one (broadcast) read of the tap for FIRF_R complex
// FMAs -- the kernel above reads a tap and a sample per FMA and is bound by the LDS instruction rate.  The tap loop is
// unrolled by FIRF_R so that the window rotates through fixed register names (no moves).  The input window of the
// workgroup is stored transposed, element e at (e % FIRF_R) * pitch + e / FIRF_R: lanes, whose windows start
// FIRF_R samples apart, then read consecutive addresses (no bank conflicts).
constexpr int FIRF_R = 8;
constexpr int FIRF_TILE = 256 * FIRF_R;  // outputs per workgroup

__global__ __launch_bounds__(256) void k_fir_fast(const float2* __restrict__ x, int64_t n, const float* __restrict__ taps,
                                                  int32_t ntaps, const float2* __restrict__ delay, int32_t dlen,
                                                  float2* __restrict__ out) {
    extern __shared__ float s_firf[];
    const int ntp = (ntaps + FIRF_R - 1) / FIRF_R * FIRF_R;  // taps padded with zeros to a multiple of FIRF_R
    float* s_taps = s_firf;                                  // ntp
    float2* s_in = reinterpret_cast<float2*>(s_firf + ntp);  // FIRF_R rows of `pitch`
    const int span = FIRF_TILE + ntp;                        // samples i0 - ntp .. i0 + FIRF_TILE - 1
    const int pitch = span / FIRF_R + 1;
    const int64_t i0 = (int64_t)blockIdx.x * FIRF_TILE;
    for (int t = threadIdx.x; t < ntp; t += 256) s_taps[t] = t < ntaps ? taps[t] : 0.f;
    stage_batched<8>(
        span,
        [&](int t) {
            const int64_t j = i0 - ntp + t;
            float2 v = make_float2(0.f, 0.f);
            if (j >= 0) {
                if (j < n) v = x[j];
            } else if (delay && -j <= dlen) {
                v = delay[dlen + j];
            }
            return v;
        },
        [&](int t, float2 v) { s_in[(t % FIRF_R) * pitch + t / FIRF_R] = v; });
    __syncthreads();
    // outputs l0 .. l0 + R - 1 of the tile; sample index (tile-local, offset ntp) of output l and tap k: ntp + l - k
    const int l0 = threadIdx.x * FIRF_R;
    float2 acc[FIRF_R], win[FIRF_R];
#pragma unroll
    for (int r = 0; r < FIRF_R; ++r) {
        acc[r] = make_float2(0.f, 0.f);
        const int e = ntp + l0 + r;  // tap 0
        win[r] = s_in[(e % FIRF_R) * pitch + e / FIRF_R];
    }
    for (int k0 = 0; k0 < ntp; k0 += FIRF_R) {
#pragma unroll
        for (int kk = 0; kk < FIRF_R; ++kk) {
            const float c = s_taps[k0 + kk];
            // at tap k = k0 + kk output r needs sample e = ntp + l0 + r - k, held in win[(r - kk) mod R]
#pragma unroll
            for (int r = 0; r < FIRF_R; ++r) {
                const float2 w = win[(r - kk + FIRF_R) % FIRF_R];
                acc[r].x += c * w.x;
                acc[r].y += c * w.y;
            }
            // the sample of output R-1 (slot (R-1-kk) mod R) is not needed again: replace it with the one output 0
            // needs at the next tap, e = ntp + l0 - (k + 1)
            const int e = ntp + l0 - (k0 + kk + 1);
            win[(FIRF_R - 1 - kk) % FIRF_R] = s_in[((e % FIRF_R + FIRF_R) % FIRF_R) * pitch + e / FIRF_R];
        }
    }
#pragma unroll
    for (int r = 0; r < FIRF_R; ++r) {
        const int64_t i = i0 + l0 + r;
        if (i < n) out[i] = acc[r];
    }
}

// Decimating FIR (2 <= dsr <= FIRD_MAXDSR), optionally fused with the int16 IQ ingest (the front-end
// filter/decimate folded into the rx load): out[o] = y[o*dsr + phase], y = lfilter(taps, 1, scale * x).  Only the
// kept outputs are computed -- k_fir evaluates the tile un-decimated and leaves (dsr-1)/dsr of its lanes idle.
// A thread owns `per` kept outputs o0 + lane + r*256; the workgroup's input window is stored in polyphase order
// (element e at (e % dsr) * pitch + e / dsr), so that at every tap the lanes, whose samples are dsr apart, read
// consecutive LDS words.  TIn = float2 (complex64) or short2 (interleaved int16 IQ: 4 B read per sample).
constexpr int FIRD_MAXDSR = 16;
constexpr int FIRD_MAXPER = 4;

__device__ __forceinline__ float2 fird_load(const float2* p, int64_t i, float) { return p[i]; }
__device__ __forceinline__ float2 fird_load(const short2* p, int64_t i, float scale) {
    const short2 v = p[i];
    return make_float2((float)v.x * scale, (float)v.y * scale);
}

template <typename TIn>
__global__ __launch_bounds__(256) void k_fir_decim(const TIn* __restrict__ x, int64_t n, float scale,
                                                   const float* __restrict__ taps, int32_t ntaps,
                                                   const TIn* __restrict__ delay, int32_t dlen, int32_t dsr, int32_t phase,
                                                   int32_t per, float2* __restrict__ out, int64_t nout) {
    extern __shared__ float s_fird[];
    float* s_taps = s_fird;                                              // ntaps
    float2* s_in = reinterpret_cast<float2*>(s_fird + ((ntaps + 1) & ~1));  // dsr rows of `pitch`
    const int tile = 256 * per;                                          // kept outputs per workgroup
    const int span = (tile - 1) * dsr + ntaps;                           // inputs i0 .. i0 + span - 1
    const int pitch = span / dsr + 1;
    const int64_t o0 = (int64_t)blockIdx.x * tile;
    const int64_t i0 = o0 * dsr + phase - (ntaps - 1);                   // input index of window element 0
    for (int t = threadIdx.x; t < ntaps; t += 256) s_taps[t] = taps[t];
    stage_batched<(sizeof(TIn) == 8 ? 8 : 1)>(
        span,
        [&](int t) {
            const int64_t j = i0 + t;
            float2 v = make_float2(0.f, 0.f);
            if (j >= 0) {
                if (j < n) v = fird_load(x, j, scale);
            } else if (delay && -j <= dlen) {
                v = fird_load(delay, dlen + j, scale);
            }
            return v;
        },
        [&](int t, float2 v) { s_in[(t % dsr) * pitch + t / dsr] = v; });
    __syncthreads();
    // output l of the tile at tap k reads window element e = l*dsr + (ntaps-1-k): row (ntaps-1-k) % dsr,
    // column l + (ntaps-1-k) / dsr
    float2 acc[FIRD_MAXPER];
#pragma unroll
    for (int r = 0; r < FIRD_MAXPER; ++r) acc[r] = make_float2(0.f, 0.f);
    int m = ntaps - 1;
    int row = m % dsr, col = m / dsr;
    for (int k = 0; k < ntaps; ++k) {
        const float c = s_taps[k];
        const float2* w = s_in + row * pitch + col + threadIdx.x;
#pragma unroll
        for (int r = 0; r < FIRD_MAXPER; ++r) {
            if (r < per) {
                const float2 v = w[r * 256];
                acc[r].x += c * v.x;
                acc[r].y += c * v.y;
            }
        }
        if (--row < 0) {
            row = dsr - 1;
            --col;
        }
    }
#pragma unroll
    for (int r = 0; r < FIRD_MAXPER; ++r) {
        const int64_t o = o0 + threadIdx.x + r * 256;
        if (r < per && o < nout && o * dsr + phase < n) out[o] = acc[r];
    }
}

// Register-tiled decimating FIR for small decimation factors (window of the tile below ~7000 samples).  Polyphase
// view: with m = ntaps-1-k = q*dsr + rho, output l reads sample (l + q)*dsr + rho, i.e. column l + q of branch rho,
// so per branch the filter is a sliding dot product over columns with the sub-filter g_rho[q] = taps[ntaps-1-m]
// -- the structure of k_fir_fast.  A thread owns FIRP_R consecutive outputs and keeps their FIRP_R-column window of
// the current branch in registers (the q loop is unrolled by FIRP_R, the window rotates through fixed names): one LDS
// sample read and one broadcast tap read per FIRP_R complex-by-real MACs, 2.5x fewer LDS reads than k_fir_decim.
// Branch rows are stored with the columns transposed (c % FIRP_R major) so that lanes read consecutive words.
constexpr int FIRP_R = 4;
constexpr int FIRP_TILE = 256 * FIRP_R;
constexpr int FIRP_MAXSPAN = 7400;  // samples of the tile window (+ the tap table: < 64 KB of LDS)

// A workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ...: the next tile's window is fetched into REGISTERS (raw
// TIn: one register per int16 IQ sample) right after the barrier that hands the current one to the sliding dot products, so its
// round trips to memory run under a tile's worth of arithmetic instead of in front of it (one tile per workgroup: stage ->
// barrier -> compute, 62 % of the wave cycles parked; the launch now keeps as many workgroups as are resident).
__device__ __forceinline__ float2 fird_cvt(float2 v, float) { return v; }
__device__ __forceinline__ float2 fird_cvt(short2 v, float scale) { return make_float2((float)v.x * scale, (float)v.y * scale); }
template <typename TIn>
__device__ __forceinline__ TIn fird_zero();
template <>
__device__ __forceinline__ float2 fird_zero<float2>() { return make_float2(0.f, 0.f); }
template <>
__device__ __forceinline__ short2 fird_zero<short2>() { return make_short2(0, 0); }

// EPT: window elements a thread stages per tile (instantiated for 8 / 16 / 24 / 32: the prefetch registers of the shape at hand)
template <typename TIn, int NT, int EPT>
__global__ __launch_bounds__(NT) void k_fir_poly(const TIn* __restrict__ x, int64_t n, float scale,
                                                  const float* __restrict__ taps, int32_t ntaps,
                                                  const TIn* __restrict__ delay, int32_t dlen, int32_t dsr, int32_t phase,
                                                  float2* __restrict__ out, int64_t nout, int64_t ntiles) {
    extern __shared__ float s_firp[];
    const int qmax = (ntaps + dsr - 1) / dsr;                      // sub-filter length of branch 0 (the longest)
    const int qpad = (qmax + FIRP_R - 1) / FIRP_R * FIRP_R;        // padded with zero taps
    const int ncols = NT * FIRP_R + qpad;                            // columns per branch row
    const int pitch2 = ncols / FIRP_R + 1;
    const int rowpitch = FIRP_R * pitch2;
    float* s_g = s_firp;                                           // dsr * qpad sub-filter taps
    float2* s_x = reinterpret_cast<float2*>(s_firp + ((dsr * qpad + 1) & ~1));  // dsr rows of rowpitch
    for (int t = threadIdx.x; t < dsr * qpad; t += NT) {
        const int rho = t / qpad, q = t - rho * qpad;
        const int m = q * dsr + rho;
        s_g[t] = m < ntaps ? taps[ntaps - 1 - m] : 0.f;
    }
    // window element e = c*dsr + rho -> row rho, column c; a thread stages the elements tid, tid + NT, ...
    constexpr int MAXE = EPT;
    const int total = ncols * dsr;
    TIn pre[MAXE];
    auto fetch = [&](int64_t tile) {
        const int64_t i0 = tile * (NT * FIRP_R) * dsr + phase - (ntaps - 1);  // input index of window element 0
        // elements lo <= e < hi come from x (uniform base + 32-bit offsets), dl <= e < lo from the delay line, the rest are zeros
        const int lo = (int)std::min<int64_t>(std::max<int64_t>(-i0, 0), total), hi = (int)std::min<int64_t>(std::max<int64_t>(n - i0, 0), total);
        const int dl = delay ? (int)std::min<int64_t>(std::max<int64_t>(-i0 - dlen, 0), total) : lo;
        const TIn* xb = x + i0;
        const TIn* db = delay + (dlen + i0);
#pragma unroll
        for (int u = 0; u < MAXE; ++u) {
            const int e = (int)threadIdx.x + NT * u;
            TIn v = fird_zero<TIn>();
            if (e >= lo) {
                if (e < hi) v = xb[e];
            } else if (e >= dl) {
                v = db[e];
            }
            pre[u] = v;
        }
    };
    int64_t tile = blockIdx.x;
    if (tile < ntiles) fetch(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        {   // (rho, c) advance without divisions
            int c = threadIdx.x / dsr, rho = threadIdx.x - c * dsr;
            const int dc = NT / dsr, dr = NT - dc * dsr;
#pragma unroll
            for (int u = 0; u < MAXE; ++u) {
                if ((int)threadIdx.x + NT * u < total) s_x[rho * rowpitch + (c % FIRP_R) * pitch2 + c / FIRP_R] = fird_cvt(pre[u], scale);
                c += dc;
                rho += dr;
                if (rho >= dsr) {
                    rho -= dsr;
                    ++c;
                }
            }
        }
        __syncthreads();
        if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x);
        const int64_t o0 = tile * (NT * FIRP_R);
        float2 acc[FIRP_R];
#pragma unroll
        for (int r = 0; r < FIRP_R; ++r) acc[r] = make_float2(0.f, 0.f);
        for (int rho = 0; rho < dsr; ++rho) {
            const float2* xr = s_x + rho * rowpitch + threadIdx.x;  // column l0 + r + q with l0 = R * tid
            const float* g = s_g + rho * qpad;
            float2 win[FIRP_R];
#pragma unroll
            for (int r = 0; r < FIRP_R; ++r) win[r] = xr[r * pitch2];  // columns l0 + r (q = 0)
            for (int q0 = 0; q0 < qpad; q0 += FIRP_R) {
#pragma unroll
                for (int qq = 0; qq < FIRP_R; ++qq) {
                    const float c = g[q0 + qq];
                    // output r at q reads column l0 + r + q, held in slot (r + qq) mod R
#pragma unroll
                    for (int r = 0; r < FIRP_R; ++r) {
                        const float2 w = win[(r + qq) % FIRP_R];
                        acc[r].x += c * w.x;
                        acc[r].y += c * w.y;
                    }
                    // column l0 + q is done; slot qq takes column l0 + q + R
                    win[qq] = xr[qq * pitch2 + q0 / FIRP_R + 1];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < FIRP_R; ++r) {
            const int64_t o = o0 + threadIdx.x * FIRP_R + r;
            if (o < nout && o * dsr + phase < n) out[o] = acc[r];
        }
        __syncthreads();  // the window is overwritten by the next tile
    }
}

// upfirdn == scipy.signal.upfirdn(taps, x, up, down) per row; out[r][o] = sum_k taps[k] xu[o*down - k],
// xu = x upsampled by `up` (zeros between samples).  Optional |.| output.
// The taps that meet a sample of x for output o are k = k0, k0 + up, ... with k0 = (o down) mod up, and they meet
// x[j0], x[j0 - 1], ... (j0 = (o down - k0) / up): one division per output, none per tap.
// STAGE: the workgroup's input window x[jlo .. jhi] (256 consecutive outputs: (255 down + ntaps) / up + 2 samples)
// is staged in LDS with coalesced loads (upfirdn.cu:68-182 does the same with its shared-memory window); without it
// every tap re-reads x from global memory.
template <bool STAGE>
__global__ __launch_bounds__(256) void k_upfirdn(const float2* __restrict__ x, int64_t n, const float* __restrict__ taps,
                                                 int32_t ntaps, int32_t up, int32_t down, int64_t nout, int32_t span,
                                                 float2* __restrict__ out, float* __restrict__ out_abs) {
    extern __shared__ float s_tp[];
    float2* s_x = reinterpret_cast<float2*>(s_tp + ((ntaps + 1) & ~1));
    for (int t = threadIdx.x; t < ntaps; t += 256) s_tp[t] = taps[t];
    const int64_t row = blockIdx.y;
    const float2* xr = x + row * n;
    const int64_t o0 = (int64_t)blockIdx.x * 256;
    // first sample any output of the workgroup can touch: floor((o0 down - (ntaps - 1)) / up), clipped below
    const int64_t plo = o0 * down - (ntaps - 1);
    const int64_t jlo = plo >= 0 ? plo / up : -((-plo + up - 1) / up);
    if (STAGE) {
        stage_batched<8>(span, [&](int i) { const int64_t j = jlo + i; return (j >= 0 && j < n) ? xr[j] : make_float2(0.f, 0.f); },
                         [&](int i, float2 v) { s_x[i] = v; });
    }
    __syncthreads();
    const int64_t o = o0 + threadIdx.x;
    if (o >= nout) return;
    const int64_t pos = o * down;  // index into the upsampled stream
    const int k0 = (int)(pos % up);
    int64_t j = (pos - k0) / up;
    float ar = 0.f, ai = 0.f;
    for (int k = k0; k < ntaps && j >= 0; k += up, --j) {
        if (j < n) {
            const float c = s_tp[k];
            const float2 v = STAGE ? s_x[j - jlo] : xr[j];
            ar += c * v.x;
            ai += c * v.y;
        }
    }
    if (out) out[row * nout + o] = make_float2(ar, ai);
    if (out_abs) out_abs[row * nout + o] = sqrtf(ar * ar + ai * ai);
}

// Polyphase form of the same filter for small interpolation factors (up <= 16).
// A thread owns one GROUP of `up` consecutive outputs o = g up + p, p = 0 .. up-1.  For phase p the taps are
// k0 + i up with k0 = (p down) mod up and the samples x[g down + c - i] with c = (p down) div up: within a wave the tap
// index is the same for every lane (one broadcast 16-byte LDS read serves four taps) and the sample index runs with the
// lane.  The window is stored by residue modulo `down` and the taps of a phase are walked residue by residue
// (i = i' down + rho), so that inside a residue both the sample column (tid + q0 - i') and the tap index (i') are
// linear: conflict-free reads at immediate offsets, no address arithmetic per tap.  Phases that share c (c is
// non-decreasing in p) share the samples, which are read once for up to four of them.  Against k_upfirdn this is ~3x
// fewer LDS reads per multiply-add, no per-lane tap addressing and no integer division per output.  The tile's
// 256 up outputs leave through LDS as whole rows.  (upfirdn.cu:68-182 keeps one output per thread.)
constexpr int UFP_MAXUP = 16;
template <int NP>
__device__ __forceinline__ void ufp_residue(const float2* __restrict__ xcol, const float* __restrict__ tp, int tp_phase_stride,
                                            int ntip, float (&ar)[4], float (&ai)[4]) {
    for (int i = 0; i < ntip; i += 4) {
        float2 xs[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) xs[u] = xcol[-(i + u)];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const float4 t4 = *reinterpret_cast<const float4*>(tp + q * tp_phase_stride + i);
            ar[q] += t4.x * xs[0].x + t4.y * xs[1].x + t4.z * xs[2].x + t4.w * xs[3].x;
            ai[q] += t4.x * xs[0].y + t4.y * xs[1].y + t4.z * xs[2].y + t4.w * xs[3].y;
        }
    }
}
__global__ __launch_bounds__(256) void k_upfirdn_poly(const float2* __restrict__ x, int64_t n, const float* __restrict__ taps,
                                                      int32_t ntaps, int32_t up, int32_t down, int64_t nout, int32_t ntip,
                                                      int32_t pitch, int32_t span, float2* __restrict__ out,
                                                      float* __restrict__ out_abs) {
    extern __shared__ __attribute__((aligned(16))) float s_ufp[];
    float* s_tp = s_ufp;                                             // [up][down][ntip] taps, zero-padded
    float2* s_x = reinterpret_cast<float2*>(s_tp + up * down * ntip);  // [down][pitch] window by residue
    float2* s_out = s_x + down * pitch;                              // [256][up] outputs of the tile
    const int tid = threadIdx.x;
    const int64_t row = blockIdx.y;
    const float2* xr = x + row * n;
    const int64_t gg0 = (int64_t)blockIdx.x * 256;                   // first group of the workgroup
    const int i_max = ntip * down - 1;                               // largest (padded) tap number of a phase
    for (int e = tid; e < up * down * ntip; e += 256) {
        const int ip = e % ntip, pr = e / ntip;                      // pr = p * down + rho
        const int rho = pr % down, p = pr / down;
        const int k = (p * down) % up + (ip * down + rho) * up;
        s_tp[e] = k < ntaps ? taps[k] : 0.f;
    }
    const int64_t jlo = gg0 * down - i_max;
    for (int idx = tid; idx < span; idx += 256) {
        const int64_t j = jlo + idx;
        const int q = idx / down, r = idx - q * down;
        s_x[r * pitch + q] = (j >= 0 && j < n) ? xr[j] : make_float2(0.f, 0.f);
    }
    __syncthreads();
    for (int p = 0; p < up;) {
        const int c = (p * down) / up;
        int np = 1;
        while (np < 4 && p + np < up && ((p + np) * down) / up == c) ++np;  // phases p .. p+np-1 share their samples
        float ar[4] = {0.f, 0.f, 0.f, 0.f}, ai[4] = {0.f, 0.f, 0.f, 0.f};
        for (int rho = 0; rho < down; ++rho) {
            // tap i = i' down + rho reads window index tid down + (c + i_max - rho) - i' down
            const int e = c + i_max - rho;
            const int eq = e / down, er = e - eq * down;
            const float2* xcol = s_x + er * pitch + eq + tid;
            const float* tp = s_tp + (p * down + rho) * ntip;
            if (np == 1) ufp_residue<1>(xcol, tp, down * ntip, ntip, ar, ai);
            else if (np == 2) ufp_residue<2>(xcol, tp, down * ntip, ntip, ar, ai);
            else if (np == 3) ufp_residue<3>(xcol, tp, down * ntip, ntip, ar, ai);
            else ufp_residue<4>(xcol, tp, down * ntip, ntip, ar, ai);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < np) s_out[tid * up + p + q] = make_float2(ar[q], ai[q]);
        p += np;
    }
    __syncthreads();
    const int64_t o0 = gg0 * up;
    for (int e = tid; e < 256 * up; e += 256) {
        const int64_t o = o0 + e;
        if (o < nout) {
            const float2 v = s_out[e];
            if (out) out[row * nout + o] = v;
            if (out_abs) out_abs[row * nout + o] = sqrtf(v.x * v.x + v.y * v.y);
        }
    }
}

// elementwise complex row-broadcast multiply: y[r][i] = x[r][i] * v[i]  (CZT pre/post chirps, spectra)
__global__ __launch_bounds__(256) void k_rows_mul_vec(const float2* __restrict__ x, int64_t in_pitch, int64_t in_off,
                                                      const float2* __restrict__ v, int64_t len,
                                                      float2* __restrict__ y, int64_t out_pitch, int64_t pad_to,
                                                      float scale) {
    const int64_t r = blockIdx.y;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < pad_to; i += (int64_t)gridDim.x * 256) {
        float2 o = make_float2(0.f, 0.f);
        if (i < len) {
            const float2 a = x[r * in_pitch + in_off + i], b = v[i];
            o = make_float2((a.x * b.x - a.y * b.y) * scale, (a.x * b.y + a.y * b.x) * scale);
        }
        y[r * out_pitch + i] = o;
    }
}

__global__ __launch_bounds__(256) void k_scale(float2* __restrict__ y, int64_t n, float scale) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        float2 v = y[i];
        y[i] = make_float2(v.x * scale, v.y * scale);
    }
}

// interleaved int16 IQ -> complex64 (usrpRoutines.simpleBinRead's .astype(float32).view(complex64),
// usrpRoutines.py:51-67, done on the device as in benchmarks/benchmark_cupyCopyAndConvert.py:17-25):
// 4 B read + 8 B write per sample; one thread converts 4 samples (16-B load, 2 x 16-B stores).
__global__ __launch_bounds__(256) void k_iq16_to_c64(const short* __restrict__ in, int64_t nsamp, float scale,
                                                     float2* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
    for (int64_t s = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; s < nsamp; s += stride) {
        if (s + 4 <= nsamp) {
            const short4 a = *reinterpret_cast<const short4*>(in + 2 * s);
            const short4 b = *reinterpret_cast<const short4*>(in + 2 * s + 4);
            float4 o0 = make_float4(a.x * scale, a.y * scale, a.z * scale, a.w * scale);
            float4 o1 = make_float4(b.x * scale, b.y * scale, b.z * scale, b.w * scale);
            *reinterpret_cast<float4*>(out + s) = o0;
            *reinterpret_cast<float4*>(out + s + 2) = o1;
        } else {
            for (int64_t k = s; k < nsamp; ++k) out[k] = make_float2(in[2 * k] * scale, in[2 * k + 1] * scale);
        }
    }
}

// ---- engine add-ons: complex QF output and the across-template maximum -------------------
// cqf[h][i] = P[h][i] * sqrt(tscale[t] * inv_e[i])   (TemplateCrossCorrelator layout, xcorrRoutines.py:352-357)
__global__ __launch_bounds__(256) void k_complex_norm(const float2* __restrict__ pbuf, int32_t pitch, int32_t nfreq,
                                                      const float* __restrict__ tscale,
                                                      const float* __restrict__ inv_e, int64_t num_shifts,
                                                      int32_t step, int32_t blk0, int32_t nhyp,
                                                      float2* __restrict__ cqf) {
    const int z = blockIdx.z, h = blockIdx.y;
    const int blk = blk0 + z;
    const int sl = blockIdx.x * 256 + threadIdx.x;
    const int64_t rel = (int64_t)blk * step + sl;
    if (sl >= step || rel >= num_shifts) return;
    const float2 p = pbuf[((int64_t)z * nhyp + h) * pitch + sl];
    const float g = sqrtf(tscale[h / nfreq]) * sqrtf(inv_e[rel]);
    cqf[(int64_t)h * num_shifts + rel] = make_float2(p.x * g, p.y * g);
}

// per column i of complex (rows, n): max_r |z[r][i]| and its first row index (int32, or int64 = cp.argmax's dtype)
template <typename TArg>
__global__ __launch_bounds__(256) void k_colmax_abs(const float2* __restrict__ z, int32_t rows, int64_t n,
                                                    float* __restrict__ maxv, TArg* __restrict__ arg) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float bv = -1.f;
    TArg bi = 0;
    for (int r = 0; r < rows; ++r) {
        const float2 a = z[(int64_t)r * n + i];
        // |z| via float64 so that the float32 result is the correctly rounded one (== numpy/hypotf)
        const float v = (float)sqrt((double)a.x * (double)a.x + (double)a.y * (double)a.y);
        if (v > bv) {
            bv = v;
            bi = r;
        }
    }
    maxv[i] = bv;
    arg[i] = bi;
}

// per column i of a real (rows, n) matrix of QF^2 values: max_r sqrt(q[r][i]) and its first row (int64, the dtype
// of cp.argmax) -- TemplateCrossCorrelator.correlate(returnMax=True) on per-template QF^2 traces; the comparison
// is made on the float32 square roots, like the reference's on |QF| (xcorrRoutines.py:361-371)
__global__ __launch_bounds__(256) void k_colmax_sqrt(const float* __restrict__ q, int32_t rows, int64_t n,
                                                     float* __restrict__ maxv, int64_t* __restrict__ arg) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float bv = -1.f;
    int64_t bi = 0;
    for (int r = 0; r < rows; ++r) {
        const float v = sqrtf(q[(int64_t)r * n + i]);
        if (v > bv) {
            bv = v;
            bi = r;
        }
    }
    maxv[i] = bv;
    arg[i] = bi;
}

// ---------------------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------------------
static inline unsigned cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

void launch_sliding_multiply(const float2* x, int32_t xlen, const float2* y, int64_t ylen, const double* prefix,
                             int64_t start, int64_t step, int64_t rows, double coef, int32_t zero_oor, float2* z,
                             hipStream_t st, const double* d_coef) {
    // rows of up to 8192 samples: ~64K elements per workgroup (up to 64 rows), one chunk; longer rows: one row per
    // workgroup row, up to 64 chunks
    int rpw = 1;
    int rows_fastest = 0;
    unsigned gx = std::min<unsigned>(cdiv(xlen, 256), 64);
    if (xlen > 8192 && rows >= 4) {
        rows_fastest = 1;
        // long rows: four rows per workgroup (one x load for the four), enough chunks to fill the chip
        rpw = 4;
        const int64_t groups = (rows + 3) / 4;
        gx = (unsigned)std::min<int64_t>(cdiv(xlen, 1024), std::max<int64_t>(64, 4096 / groups));
    }
    if (xlen <= 8192) {
        rpw = (int)std::max<int64_t>(1, std::min<int64_t>(SM_MAX_RPW, 65536 / std::max(xlen, 1)));
        // keep at least ~2048 workgroups in flight when there are that many rows
        while (rpw > 1 && (rows + rpw - 1) / rpw < 2048) rpw >>= 1;
        if (rpw > 1) gx = 1;
    }
    const int64_t rows_per_launch = (int64_t)65535 * rpw;
    for (int64_t r0 = 0; r0 < rows; r0 += rows_per_launch) {
        const int64_t nr = std::min<int64_t>(rows_per_launch, rows - r0);
        const unsigned gy = (unsigned)((nr + rpw - 1) / rpw);
        hipLaunchKernelGGL(k_sliding_multiply, rows_fastest ? dim3(gy, gx) : dim3(gx, gy), dim3(256), 0, st, x, xlen, y, ylen,
                           prefix, start + r0 * step, step, coef, zero_oor, z + r0 * (int64_t)xlen, d_coef, nr, rpw, rows_fastest);
    }
}

int rows_argmax_chunks(int64_t rows, int64_t len) {
    // one workgroup per row is fine while there are enough rows to fill the chip or the rows are short
    if (len <= 131072 || rows >= 2048) return 0;
    return (int)std::min<int64_t>(1024, (len + 32767) / 32768);
}

void launch_rows_argmax(const float2* z, int64_t rows, int64_t len, int32_t use_normsq, float scale, uint32_t* argmax,
                        float* maxv, float* plane, hipStream_t st, unsigned long long* part, int32_t nan_empty) {
    if (rows <= 0) return;
    const int chunks = part ? rows_argmax_chunks(rows, len) : 0;
    if (chunks > 1) {
        const int64_t chunk = ((len + chunks - 1) / chunks + 255) / 256 * 256;
        hipLaunchKernelGGL(k_rows_argmax_part, dim3((unsigned)chunks, (unsigned)rows), dim3(256), 0, st, z, len, chunk, scale, part,
                           plane);
        hipLaunchKernelGGL(k_rows_argmax_fin, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, st, part, chunks, rows, use_normsq,
                           argmax, maxv, nan_empty);
        return;
    }
    if (rows >= 1024 && len <= 32768) {
        hipLaunchKernelGGL(k_rows_argmax_wave, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, z, rows, len, use_normsq, scale,
                           argmax, maxv, plane, nan_empty);
        return;
    }
    hipLaunchKernelGGL(k_rows_argmax, dim3((unsigned)rows), dim3(256), 0, st, z, len, use_normsq, scale, argmax, maxv,
                       plane, nan_empty);
}

void launch_magnsq(const void* x, int64_t n, int in_c128, void* out, int out_f64, hipStream_t st) {
    const unsigned g = std::min<unsigned>(cdiv(n, 256), 256 * 16);
    if (!in_c128 && !out_f64)
        hipLaunchKernelGGL((k_magnsq<float2, float>), dim3(g), dim3(256), 0, st, (const float2*)x, n, (float*)out);
    else if (!in_c128 && out_f64)
        hipLaunchKernelGGL((k_magnsq<float2, double>), dim3(g), dim3(256), 0, st, (const float2*)x, n, (double*)out);
    else
        hipLaunchKernelGGL((k_magnsq<double2, double>), dim3(g), dim3(256), 0, st, (const double2*)x, n, (double*)out);
}

// Combination step of GroupXcorrCZT_Permutations.getCAF (xcorrRoutines.py:1454-1484, 1549-1585):
// out[i][k] = | sum_j planes[idx[j]][i][k] |^2 / (row_norm[i] * ynormsq), complex64 planes of rows x cols
constexpr int SUMPL_MAX = 64;
struct SumPlanesIdx {
    int32_t v[SUMPL_MAX];
};
__global__ __launch_bounds__(256) void k_sum_planes_qf2(const float2* __restrict__ planes, int64_t plane_elems,
                                                        int32_t cols, SumPlanesIdx idx, int32_t nsel,
                                                        const double* __restrict__ row_norm, double ynormsq,
                                                        double* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < plane_elems; e += stride) {
        float2 acc = make_float2(0.f, 0.f);
        for (int j = 0; j < nsel; ++j) {
            const float2 v = planes[(int64_t)idx.v[j] * plane_elems + e];
            acc.x += v.x;
            acc.y += v.y;
        }
        const float m = acc.x * acc.x + acc.y * acc.y;  // cp.abs(complex64)**2 is float32 upstream
        out[e] = (double)m / row_norm[e / cols] / ynormsq;
    }
}

void launch_sum_planes_qf2(const float2* planes, int64_t plane_elems, int32_t cols, const int32_t* h_idx, int32_t nsel,
                           const double* row_norm, double ynormsq, double* out, hipStream_t st) {
    SumPlanesIdx idx;
    for (int j = 0; j < SUMPL_MAX; ++j) idx.v[j] = j < nsel ? h_idx[j] : 0;
    const unsigned g = std::min<unsigned>(cdiv(plane_elems, 256), 256 * 16);
    hipLaunchKernelGGL(k_sum_planes_qf2, dim3(g), dim3(256), 0, st, planes, plane_elems, cols, idx, nsel, row_norm, ynormsq,
                       out);
}

// Coherent sum over the GROUPS of a composite template on the per-delay path (GroupXcorrCZT.xcorr, xcorrRoutines.py:996-1039;
// GroupXcorrCZT.cpp:106-329): planes[g][row][col] = the chirp-Z transform of group g's product row at delay `row`, evaluated
// as if the group began at sample 0; phase[g][col] = e^{-j 2 pi f_col start_g / fs} moves it to where the group lies.
//   out[row][col] = | sum_g phase[g][col] planes[g][row][col] |^2 / row_norm[row] / ynormsq
// (float32 products and sum like the upstream complex64 arithmetic, float64 normalisation).  Any number of groups.
__global__ __launch_bounds__(256) void k_sum_groups_qf2(const float2* __restrict__ planes, int32_t ngroups, int64_t plane_elems,
                                                        int32_t cols, const float2* __restrict__ phase,
                                                        const double* __restrict__ row_norm, double ynormsq,
                                                        double* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < plane_elems; e += stride) {
        const int64_t row = e / cols;
        const int col = (int)(e - row * cols);
        float2 acc = make_float2(0.f, 0.f);
        for (int g = 0; g < ngroups; ++g) {
            const float2 v = planes[(int64_t)g * plane_elems + e];
            const float2 p = phase ? phase[(int64_t)g * cols + col] : make_float2(1.f, 0.f);
            acc.x += v.x * p.x - v.y * p.y;
            acc.y += v.x * p.y + v.y * p.x;
        }
        const float m = acc.x * acc.x + acc.y * acc.y;
        out[e] = (double)m / row_norm[row] / ynormsq;
    }
}

void launch_sum_groups_qf2(const float2* planes, int32_t ngroups, int64_t plane_elems, int32_t cols, const float2* phase,
                           const double* row_norm, double ynormsq, double* out, hipStream_t st) {
    const unsigned g = std::min<unsigned>(cdiv(plane_elems, 256), 256 * 16);
    hipLaunchKernelGGL(k_sum_groups_qf2, dim3(g), dim3(256), 0, st, planes, ngroups, plane_elems, cols, phase, row_norm, ynormsq,
                       out);
}

// Sub-sample refinement after the peak (fineFreqTimeSearch / GenXcorr, xcorrRoutines.py:583-719):
//   k_mul_conj : out[i] = a[i] * conj(b[i])                       (x_fft * y_fft.conj(), y.conj() * x, masks)
//   k_steer_dot: out[r] = scale * sum_k vec[k] * conj(steer[r][k]) (np.dot(rx_vec, steeringvec.conj().T), np.vdot)
// The steering matrix is complex128 as upstream (phases 2 pi f tau need the precision); products and the sum
// are float64, the vector is the complex64 the device FFT produced.
__global__ __launch_bounds__(256) void k_mul_conj(const float2* __restrict__ a, const float2* __restrict__ b, int64_t n,
                                                  float2* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float2 x = a[i], y = b[i];
        out[i] = make_float2(x.x * y.x + x.y * y.y, x.y * y.x - x.x * y.y);
    }
}

__global__ __launch_bounds__(256) void k_steer_dot(const float2* __restrict__ vec, const double2* __restrict__ steer,
                                                   int64_t n, double scale, double2* __restrict__ out) {
    __shared__ double s_re[4], s_im[4];
    const double2* row = steer + (int64_t)blockIdx.x * n;
    double re = 0.0, im = 0.0;
    for (int64_t k = threadIdx.x; k < n; k += 256) {
        const float2 v = vec[k];
        const double2 s = row[k];
        re += (double)v.x * s.x + (double)v.y * s.y;  // v * conj(s)
        im += (double)v.y * s.x - (double)v.x * s.y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        re += __shfl_xor(re, o, 64);
        im += __shfl_xor(im, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        s_re[threadIdx.x >> 6] = re;
        s_im[threadIdx.x >> 6] = im;
    }
    __syncthreads();
    if (threadIdx.x == 0)
        out[blockIdx.x] = make_double2(scale * (s_re[0] + s_re[1] + s_re[2] + s_re[3]),
                                       scale * (s_im[0] + s_im[1] + s_im[2] + s_im[3]));
}

void launch_mul_conj(const float2* a, const float2* b, int64_t n, float2* out, hipStream_t st) {
    const unsigned g = std::min<unsigned>(cdiv(n, 256), 256 * 16);
    hipLaunchKernelGGL(k_mul_conj, dim3(g), dim3(256), 0, st, a, b, n, out);
}

void launch_steer_dot(const float2* vec, const double2* steer, int64_t rows, int64_t n, double scale, double2* out,
                      hipStream_t st) {
    hipLaunchKernelGGL(k_steer_dot, dim3((unsigned)rows), dim3(256), 0, st, vec, steer, n, scale, out);
}

// Tone-dot zoom (dotTonesScaling_32f, genTones.cu:165-283; cupyDotTonesScaling, spectralRoutines.py:580-630):
//   out[b][k] = sum_{i in 64-sample block b} src[i] * exp(j 2 pi (f0 + k fstep) i),  k < num_freqs
// One wave per block.  Every lane carries src[i] * tone and steps it by exp(j 2 pi fstep i) (complex64, as
// upstream), re-anchored with a float64 sincospi at every batch of 64 frequencies (upstream lets the float
// recurrence run over all frequencies); a 64 x 65 LDS patch turns 64 frequencies x 64 samples into row sums.
__global__ __launch_bounds__(64) void k_dot_tones(double f0, double fstep, int32_t num_freqs, int64_t len,
                                                  const float2* __restrict__ src, float2* __restrict__ out) {
    __shared__ float2 s_ws[64 * 65];
    const int lane = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * 64 + lane;
    const float2 v = i < len ? src[i] : make_float2(0.f, 0.f);
    double sr, cr;
    {
        double t = fstep * (double)i;
        t -= floor(t);  // whole cycles do not matter; keeps the argument of sincospi small
        sincospi(2.0 * t, &sr, &cr);
    }
    const float2 alpha = make_float2((float)cr, (float)sr);
    for (int k0 = 0; k0 < num_freqs; k0 += 64) {
        double t = (f0 + (double)k0 * fstep) * (double)i;
        t -= floor(t);
        sincospi(2.0 * t, &sr, &cr);
        float2 cur = make_float2(v.x * (float)cr - v.y * (float)sr, v.x * (float)sr + v.y * (float)cr);
        const int nk = min(64, num_freqs - k0);
        for (int r = 0; r < nk; ++r) {
            s_ws[r * 65 + lane] = cur;
            cur = make_float2(cur.x * alpha.x - cur.y * alpha.y, cur.x * alpha.y + cur.y * alpha.x);
        }
        __builtin_amdgcn_wave_barrier();  // one wave: LDS operations execute in order
        if (lane < nk) {
            float2 acc = make_float2(0.f, 0.f);
#pragma unroll 8
            for (int c = 0; c < 64; ++c) {
                const float2 w = s_ws[lane * 65 + c];
                acc.x += w.x;
                acc.y += w.y;
            }
            out[(int64_t)blockIdx.x * num_freqs + k0 + lane] = acc;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

void launch_dot_tones(double f0, double fstep, int32_t num_freqs, int64_t len, const float2* src, float2* out,
                      hipStream_t st) {
    hipLaunchKernelGGL(k_dot_tones, dim3((unsigned)((len + 63) / 64)), dim3(64), 0, st, f0, fstep, num_freqs, len, src,
                       out);
}

int64_t moving_num_tiles(int64_t n) { return (n + 1 + MA_TILE - 1) / MA_TILE; }

void scan_tiles(double* tile_sums, int64_t ntiles, hipStream_t st);  // caf_kernels.hip

void launch_moving_average(const float* x, int64_t n, int32_t L, int32_t sum_instead, double* tile_sums,
                           double* prefix, float* out, hipStream_t st) {
    const int64_t nt = moving_num_tiles(n);
    hipLaunchKernelGGL(k_moving_sum_prefix, dim3((unsigned)nt), dim3(MA_THREADS), 0, st, x, n, tile_sums);
    scan_tiles(tile_sums, nt, st);
    hipLaunchKernelGGL(k_moving_prefix_write, dim3((unsigned)nt), dim3(MA_THREADS), 0, st, x, n, tile_sums, prefix);
    hipLaunchKernelGGL(k_moving_from_prefix, dim3(cdiv(n, 256)), dim3(256), 0, st, prefix, n, L, sum_instead, out);
}

void launch_complex_moving_sum(const float2* x, int64_t n, int32_t L, float* out, hipStream_t st) {
    const int64_t nout = n - L + 1;
    const size_t sm = (size_t)(256 * 8 + L - 1 + 8) * sizeof(float2);
    hipLaunchKernelGGL(k_complex_moving_sum, dim3(cdiv(nout, 256 * 8)), dim3(256), sm, st, x, n, L, out);
}

void launch_multi_template_dot(const float2* tm, const float* te, int32_t ntm, int32_t L, const float2* x, int64_t xlen,
                               const double* prefix, int64_t start, int64_t nslides, int32_t* tidx, float* qf2,
                               hipStream_t st) {
    const int Lp = (L + MTR_R - 1) / MTR_R * MTR_R;
    if (Lp <= MTR_MAXL) {
        const int span = MTR_SLIDES + Lp;
        const size_t smr = (size_t)(Lp + MTR_R * (span / MTR_R + 1)) * sizeof(float2);
        hipLaunchKernelGGL(k_multi_template_dot_rt, dim3(cdiv(nslides, MTR_SLIDES)), dim3(256), smr, st, tm, te, ntm, L, x,
                           xlen, prefix, start, nslides, tidx, qf2);
        return;
    }
    const size_t sm = (size_t)(2 * L + MT_SLIDES) * sizeof(float2);
    hipLaunchKernelGGL(k_multi_template_dot, dim3(cdiv(nslides, MT_SLIDES)), dim3(256), sm, st, tm, te, ntm, L, x, xlen,
                       prefix, start, nslides, tidx, qf2);
}

void launch_multiply_indexed_rows(const float2* x, int64_t xlen, const float2* rows, int32_t row_len,
                                  const int32_t* slice_start, const int32_t* slice_lens, const int32_t* row_idx,
                                  int32_t slice_len, int64_t nslices, float2* out, hipStream_t st) {
    const unsigned gx = std::min<unsigned>(cdiv(slice_len, 256), 64);
    for (int64_t r0 = 0; r0 < nslices; r0 += 65535) {
        const int64_t nr = std::min<int64_t>(65535, nslices - r0);
        hipLaunchKernelGGL(k_multiply_indexed_rows, dim3(gx, (unsigned)nr), dim3(256), 0, st, x, xlen, rows, row_len,
                           slice_start + r0, slice_lens ? slice_lens + r0 : nullptr, row_idx + r0, slice_len,
                           out + r0 * (int64_t)slice_len);
    }
}

void launch_copy_slices(const float2* x, int64_t xlen, const int32_t* starts, int32_t starts_stride, int64_t start0,
                        int64_t inc, int32_t len, int64_t rows, float2* out, hipStream_t st) {
    const unsigned gx = std::min<unsigned>(cdiv(len, 256), 64);
    for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
        const int64_t nr = std::min<int64_t>(65535, rows - r0);
        hipLaunchKernelGGL(k_copy_slices, dim3(gx, (unsigned)nr), dim3(256), 0, st, x, xlen,
                           starts ? starts + r0 * starts_stride : nullptr, starts_stride, start0 + r0 * inc, inc, len,
                           out + r0 * (int64_t)len);
    }
}

void launch_copy_groups(const float2* x, float2* y, const int32_t* xs, const int32_t* ys, const int32_t* lens,
                        int32_t ngroups, hipStream_t st) {
    if (ngroups > 0) hipLaunchKernelGGL(k_copy_groups, dim3(ngroups), dim3(256), 0, st, x, y, xs, ys, lens);
}

// scratch: the tile counts (+ the total) followed by one 16-bit mask per thread of launch 1
int64_t local_maxima_scratch_ints(int64_t n) {
    const int64_t ntiles = (n + LM_TILE - 1) / LM_TILE;
    return ((ntiles + 1 + 3) & ~(int64_t)3) + ntiles * LM_NT / 2;
}

void launch_find_local_maxima(const float* x, int64_t n, float min_height, int32_t* tile_scratch, int32_t max_out,
                              int32_t* idx, int32_t* count, hipStream_t st) {
    const int64_t ntiles = (n + LM_TILE - 1) / LM_TILE;
    uint16_t* masks = reinterpret_cast<uint16_t*>(tile_scratch + ((ntiles + 1 + 3) & ~(int64_t)3));
    if ((reinterpret_cast<uintptr_t>(x) & 15) == 0)
        hipLaunchKernelGGL(k_local_max_flags<true>, dim3((unsigned)ntiles), dim3(LM_NT), 0, st, x, n, min_height, masks, tile_scratch);
    else
        hipLaunchKernelGGL(k_local_max_flags<false>, dim3((unsigned)ntiles), dim3(LM_NT), 0, st, x, n, min_height, masks,
                           tile_scratch);
    const int scanned = ntiles > LM_DIRECT_TILES;
    if (scanned) hipLaunchKernelGGL(k_local_max_scan, dim3(1), dim3(1024), 0, st, tile_scratch, ntiles);
    hipLaunchKernelGGL(k_local_max_write, dim3((unsigned)ntiles), dim3(LM_NT), 0, st, masks, tile_scratch, scanned, max_out, idx,
                       count);
}

void launch_gather_b32(const void* x, int64_t xlen, const int32_t* idx, int64_t n, void* out, hipStream_t st) {
    if (n > 0)
        hipLaunchKernelGGL(k_gather_b32, dim3(cdiv(n, 256)), dim3(256), 0, st, (const uint32_t*)x, xlen, idx, n,
                           (uint32_t*)out);
}

void launch_gather_f32_f64(const float* x, int64_t xlen, const int32_t* idx, int64_t n, double* out, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_gather_f32_f64, dim3(cdiv(n, 256)), dim3(256), 0, st, x, xlen, idx, n, out);
}

int moving_tile_max_window() { return MAT_MAXL; }

void launch_moving_tile(const float* x, int64_t rows, int64_t n, int32_t L, int32_t sum_instead, float* out,
                        hipStream_t st) {
    hipLaunchKernelGGL(k_moving_tile, dim3(cdiv(n, mat_outputs(L)), (unsigned)rows), dim3(MAT_NT), 0, st, x, n, L, sum_instead,
                       out);
}

bool fir_decim_ok(int32_t ntaps, int32_t dsr) { return dsr >= 1 && dsr <= FIRD_MAXDSR && ntaps <= 2048; }
// the register-tiled polyphase form (k_fir_poly) applies: its tile window of 1024 kept outputs fits the LDS
static size_t fir_poly_lds(int32_t ntaps, int32_t dsr, int* ncols_out) {
    const int qmax = (ntaps + dsr - 1) / dsr;
    const int qpad = (qmax + FIRP_R - 1) / FIRP_R * FIRP_R;
    const int ncols = FIRP_TILE + qpad;
    if (ncols_out) *ncols_out = ncols;
    return (size_t)((dsr * qpad + 1) & ~1) * sizeof(float) + (size_t)dsr * FIRP_R * (ncols / FIRP_R + 1) * sizeof(float2);
}
bool fir_poly_fits(int32_t ntaps, int32_t dsr) {
    int ncols = 0;
    const size_t smp = fir_poly_lds(ntaps, dsr, &ncols);
    return dsr >= 1 && smp <= 64 * 1024 && (size_t)ncols * dsr <= (size_t)FIRP_MAXSPAN + 4 * FIRP_R * dsr;
}

template <typename TIn>
static void launch_fir_decim(const TIn* x, int64_t n, float scale, const float* taps, int32_t ntaps, const TIn* delay,
                             int32_t dlen, int32_t dsr, int32_t phase, float2* out, int64_t nout, hipStream_t st) {
    if (nout <= 0) return;
    {   // register-tiled polyphase form when its tile window fits the LDS (small decimation factors)
        int ncols = 0;
        const size_t smp = fir_poly_lds(ntaps, dsr, &ncols);
        if (fir_poly_fits(ntaps, dsr)) {
            // (Tiles of 512 outputs on 128 threads -- half the LDS, twice the independent workgroups per CU, the same waves -- measured
            //  the same as 1024 on 256 with one tile per workgroup: 55.8 / 56.1 us for 2^24 int16 samples, 64 taps, dsr 4
            //  (profiles/r05/ab_fir_poly_nt.log).)
            // resident workgroups: what the LDS holds per CU, at most 16 waves' worth of registers (CAF_FIR_POLY_WGS: per CU, A/B)
            static const int wgs_env = [] {
                const char* e = getenv("CAF_FIR_POLY_WGS");
                return e ? atoi(e) : 0;
            }();
            static const int ncu = [] {
                int dev = 0, c = 256;
                (void)hipGetDevice(&dev);
                (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev);
                return c;
            }();
            const int64_t ntiles = cdiv(nout, FIRP_TILE);
            const int ept = (ncols * dsr + 255) / 256;
            // the grid is what is RESIDENT (tiles are dealt by striding: a workgroup that waits for a slot would start its share late)
            auto resident = [&](const void* kern) {
                thread_local std::map<std::pair<const void*, size_t>, int> cache;
                auto it = cache.find({kern, smp});
                if (it != cache.end()) return it->second;
                int nb = 1;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, smp) != hipSuccess || nb < 1) nb = 1;
                cache[{kern, smp}] = nb;
                return nb;
            };
#define CAF_FIR_POLY(E)                                                                                                            \
    do {                                                                                                                           \
        const int per_cu = wgs_env > 0 ? wgs_env : resident(reinterpret_cast<const void*>(&k_fir_poly<TIn, 256, E>));                \
        const dim3 grid((unsigned)std::min<int64_t>(ntiles, (int64_t)ncu * per_cu));                                               \
        hipLaunchKernelGGL((k_fir_poly<TIn, 256, E>), grid, dim3(256), smp, st, x, n, scale, taps, ntaps, delay, dlen, dsr, phase, \
                           out, nout, ntiles);                                                                                     \
    } while (0)
            if (ept <= 8) CAF_FIR_POLY(8); else if (ept <= 16) CAF_FIR_POLY(16); else if (ept <= 24) CAF_FIR_POLY(24); else CAF_FIR_POLY(32);
#undef CAF_FIR_POLY
            return;
        }
    }
    // kept outputs per thread: the window (tile - 1) * dsr + ntaps stays below ~6200 samples (LDS < 64 KB with the taps)
    const int per = dsr <= 4 ? 4 : (dsr <= 8 ? 2 : 1);
    const int tile = 256 * per;
    const int span = (tile - 1) * dsr + ntaps;
    const size_t sm = (size_t)((ntaps + 1) & ~1) * sizeof(float) + (size_t)dsr * (span / dsr + 1) * sizeof(float2);
    if (nout > 0)
        hipLaunchKernelGGL(k_fir_decim<TIn>, dim3(cdiv(nout, tile)), dim3(256), sm, st, x, n, scale, taps, ntaps, delay,
                           dlen, dsr, phase, per, out, nout);
}

void launch_fir(const float2* x, int64_t n, const float* taps, int32_t ntaps, const float2* delay, int32_t dlen,
                int32_t dsr, int32_t phase, float2* out, int64_t nout, hipStream_t st) {
    if (dsr == 1 && phase == 0 && nout >= n && ntaps <= 2048) {  // undecimated: the register-tiled kernel (LDS < 64 KB)
        const int ntp = (ntaps + FIRF_R - 1) / FIRF_R * FIRF_R;
        const int pitch = (FIRF_TILE + ntp) / FIRF_R + 1;
        const size_t smf = (size_t)ntp * sizeof(float) + (size_t)FIRF_R * pitch * sizeof(float2);
        hipLaunchKernelGGL(k_fir_fast, dim3(cdiv(n, FIRF_TILE)), dim3(256), smf, st, x, n, taps, ntaps, delay, dlen, out);
        return;
    }
    if (fir_decim_ok(ntaps, dsr)) {  // decimating: only the kept outputs are computed
        launch_fir_decim(x, n, 1.0f, taps, ntaps, delay, dlen, dsr, phase, out, nout, st);
        return;
    }
    const size_t sm = (size_t)((ntaps + 1) & ~1) * sizeof(float) + (size_t)(FIR_TILE + ntaps) * sizeof(float2);
    hipLaunchKernelGGL(k_fir, dim3(cdiv(n, FIR_TILE)), dim3(256), sm, st, x, n, taps, ntaps, delay, dlen, dsr, phase, out,
                       nout);
}

void launch_iq16_fir(const int16_t* iq, int64_t n, float scale, const float* taps, int32_t ntaps, const int16_t* delay,
                     int32_t dlen, int32_t dsr, int32_t phase, float2* out, int64_t nout, hipStream_t st) {
    launch_fir_decim(reinterpret_cast<const short2*>(iq), n, scale, taps, ntaps, reinterpret_cast<const short2*>(delay),
                     dlen, dsr, phase, out, nout, st);
}

void launch_upfirdn(const float2* x, int64_t rows, int64_t n, const float* taps, int32_t ntaps, int32_t up, int32_t down,
                    int64_t nout, float2* out, float* out_abs, hipStream_t st) {
    if (up <= UFP_MAXUP) {
        // polyphase form: 256 groups of `up` outputs per workgroup; taps of a phase split by residue of the tap number
        const int nt = (ntaps - 1) / up + 1;                          // taps per phase
        const int ntip = ((nt + down - 1) / down + 3) & ~3;           // per residue, padded to whole 16-byte reads
        const int cmax = (int)(((int64_t)(up - 1) * down) / up);
        const int64_t span = 255 * (int64_t)down + cmax + (int64_t)ntip * down;
        const int64_t pitch = span / down + 2;
        const size_t lds = (size_t)up * down * ntip * sizeof(float) + (size_t)down * pitch * sizeof(float2) +
                           (size_t)256 * up * sizeof(float2);
        if (lds <= 64 * 1024) {
            const int64_t ngroups = (nout + up - 1) / up;
            hipLaunchKernelGGL(k_upfirdn_poly, dim3(cdiv(ngroups, 256), (unsigned)rows), dim3(256), lds, st, x, n, taps, ntaps, up,
                               down, nout, ntip, (int32_t)pitch, (int32_t)span, out, out_abs);
            return;
        }
    }
    // input window of 256 consecutive outputs; staged in LDS when it fits beside the taps (<= 64 KB)
    const int64_t span = (255 * (int64_t)down + ntaps - 1) / up + 3;
    const size_t tap_bytes = (size_t)((ntaps + 1) & ~1) * sizeof(float);
    if (tap_bytes + (size_t)span * sizeof(float2) <= 64 * 1024)
        hipLaunchKernelGGL(k_upfirdn<true>, dim3(cdiv(nout, 256), (unsigned)rows), dim3(256),
                           tap_bytes + (size_t)span * sizeof(float2), st, x, n, taps, ntaps, up, down, nout, (int32_t)span, out,
                           out_abs);
    else
        hipLaunchKernelGGL(k_upfirdn<false>, dim3(cdiv(nout, 256), (unsigned)rows), dim3(256), tap_bytes, st, x, n, taps,
                           ntaps, up, down, nout, 0, out, out_abs);
}

void launch_rows_mul_vec(const float2* x, int64_t in_pitch, int64_t in_off, const float2* v, int64_t len, float2* y,
                         int64_t out_pitch, int64_t pad_to, int64_t rows, float scale, hipStream_t st) {
    const unsigned gx = std::min<unsigned>(cdiv(pad_to, 256), 256);
    for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
        const int64_t nr = std::min<int64_t>(65535, rows - r0);
        hipLaunchKernelGGL(k_rows_mul_vec, dim3(gx, (unsigned)nr), dim3(256), 0, st, x + r0 * in_pitch, in_pitch, in_off, v,
                           len, y + r0 * out_pitch, out_pitch, pad_to, scale);
    }
}

void launch_complex_norm(const float2* pbuf, int32_t pitch, int32_t nfreq, const float* tscale, const float* inv_e,
                         int64_t num_shifts, int32_t step, int32_t blk0, int32_t nblk, int32_t nhyp, float2* cqf,
                         hipStream_t st) {
    hipLaunchKernelGGL(k_complex_norm, dim3(cdiv(step, 256), nhyp, nblk), dim3(256), 0, st, pbuf, pitch, nfreq, tscale,
                       inv_e, num_shifts, step, blk0, nhyp, cqf);
}

void launch_argmax3d_u32(const uint32_t* x, int64_t items, int32_t d1, int32_t d2, int32_t d3, uint32_t* argmax,
                         uint32_t* maxv, hipStream_t st) {
    if (items > 0)
        hipLaunchKernelGGL(k_argmax3d_u32, dim3((unsigned)items), dim3(256), 0, st, x, d1, d2, d3, argmax, maxv);
}

void launch_iq16_to_c64(const short* in, int64_t nsamp, float scale, float2* out, hipStream_t st) {
    if (nsamp > 0)
        hipLaunchKernelGGL(k_iq16_to_c64, dim3(std::min<unsigned>(cdiv(nsamp, 1024), 8192)), dim3(256), 0, st, in, nsamp,
                           scale, out);
}

void launch_scale(float2* y, int64_t n, float scale, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_scale, dim3(std::min<unsigned>(cdiv(n, 256), 4096)), dim3(256), 0, st, y, n, scale);
}

void launch_colmax_abs(const float2* z, int32_t rows, int64_t n, float* maxv, void* arg, int32_t arg64, hipStream_t st) {
    if (arg64)
        hipLaunchKernelGGL(k_colmax_abs<int64_t>, dim3(cdiv(n, 256)), dim3(256), 0, st, z, rows, n, maxv, (int64_t*)arg);
    else
        hipLaunchKernelGGL(k_colmax_abs<int32_t>, dim3(cdiv(n, 256)), dim3(256), 0, st, z, rows, n, maxv, (int32_t*)arg);
}

void launch_colmax_sqrt(const float* q, int32_t rows, int64_t n, float* maxv, int64_t* arg, hipStream_t st) {
    hipLaunchKernelGGL(k_colmax_sqrt, dim3(cdiv(n, 256)), dim3(256), 0, st, q, rows, n, maxv, arg);
}

}  // namespace caf
