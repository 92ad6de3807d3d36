// Fused per-delay correlator for power-of-two cutouts (64 <= N <= 16384): ONE kernel does, per delay s,
//   p[t] = x[t] * y[s + t]   ->   N-point FFT held in LDS   ->   |.|^2 / (E_s * ||x||^2)   ->   (max, first argmax)
// with the window energy E_s = sum_t |y[s+t]|^2 and ||x||^2 summed (float64) from the samples the kernel reads
// anyway.  No (rows, N) product matrix, no energy-prefix pass over rx, no host synchronisation: the only HBM
// traffic that scales with rows x N is the optional |.|^2 / complex plane the caller asks for.
// This is the reference's literal branch-B/C algorithm (xcorrRoutines.py:511-566: slice * conj(cutout) -> fft ->
// abs^2 -> argmax -> two norms; threaded twin IppXcorrFFT.cpp:94-178; GPU v1/v2 xcorrRoutines.py:29-274 with
// multiplySlices.cu:113-216 + cuFFT + argmax.cu:93-153), minus its three matrix-sized HBM passes.
//
// Transform: Stockham decimation-in-time, radix 16 (+ one final radix 2/4/8 pass when log2 N is not a multiple of
// four), 16 points per thread, N/16 threads per row (4 ... 1024), in place in LDS with two workgroup barriers
// per pass (all butterflies read, then all write).  The forward DFT is evaluated as conj(IDFT(conj p)) so that
// the inverse butterflies of caf_fft_dev.h serve both engines; only |.|^2 and the optional complex plane see
// the conjugate.  Twiddles: one table lookup (W_16384^q, exact to f32) per butterfly and pass, powers by
// recurrence.  LDS addresses are padded by one element per sixteen, which makes the stride-16 stores of the
// first pass conflict-free.
#include <mutex>
#include <vector>
#include <complex>

#include "caf_internal.h"
#include "caf_fft_dev.h"

namespace caf {

namespace {

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

// WG threads = max(256, N/16); rows handled concurrently RPW = WG / (N/16); each workgroup walks `rows_per_wg`
// consecutive groups of RPW rows with the cutout resident in registers.
template <int LOGN>
// (4 waves per SIMD: 128 VGPRs, the budget that lets four 256-thread workgroups -- or one of 1024 -- share a CU)
__global__ __launch_bounds__((1 << LOGN) / 16 > 256 ? (1 << LOGN) / 16 : 256, 4) void k_perdelay_fused(
    const float2* __restrict__ x, const float2* __restrict__ y, int64_t ylen, const float2* __restrict__ tw, int64_t start,
    int64_t step, int64_t num, int32_t rows_per_wg, int32_t zero_oor, float* __restrict__ qf2, uint32_t* __restrict__ fidx,
    float* __restrict__ plane, float2* __restrict__ cplane) {
    constexpr int N = 1 << LOGN, NTR = N / 16, WG = NTR > 256 ? NTR : 256, RPW = WG / NTR;
    constexpr int NW = WG / 64;            // waves per workgroup
    constexpr int WPR = NTR > 64 ? NTR / 64 : 1;  // waves per row
    extern __shared__ __attribute__((aligned(16))) float2 s_buf[];  // RPW rows of N + N/16 elements
    __shared__ double s_e[2][NW];     // cross-wave partial sums / maxima of rows wider than a wave, double-buffered
    __shared__ float s_bv[2][NW];     // over consecutive rows so that a slot is rewritten only two barriers later
    __shared__ uint32_t s_bi[2][NW];
    const int tid = threadIdx.x;
    const int rl = tid / NTR, l = tid - rl * NTR;  // row slot of this thread, row-local id
    float2* buf = s_buf + rl * (N + N / 16);
    const int lane = tid & 63, wave = tid >> 6;

    // cutout: 16 points per thread (pass-1 positions), resident; ||x||^2 in float64
    float2 xr[16];
    double xs = 0.0;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        xr[t] = x[l + t * NTR];
        xs += (double)xr[t].x * xr[t].x + (double)xr[t].y * xr[t].y;
    }
    // sum over the lanes of the row slot that share this wave
    auto wave_sum = [&](double e) {
#pragma unroll
        for (int o = 1; o < 64; o <<= 1)
            if (o < NTR) e += __shfl_xor(e, o, 64);
        return e;
    };
    double xnorm2 = wave_sum(xs);
    if (WPR > 1) {
        if (lane == 0) s_e[0][wave] = xnorm2;
        __syncthreads();
        xnorm2 = 0.0;
#pragma unroll
        for (int w = 0; w < WPR; ++w) xnorm2 += s_e[0][(wave / WPR) * WPR + w];
        __syncthreads();
    }

    const int64_t row0 = (int64_t)blockIdx.x * rows_per_wg * RPW;
    for (int it = 0; it < rows_per_wg; ++it) {
        const int64_t row = row0 + (int64_t)it * RPW + rl;
        const bool live = row < num;  // (uniform per row slot; dead slots run the barriers with zeros)
        const int64_t s = start + row * step;
        const bool oor = (s < 0) || (s + N > ylen);
        const bool zero = !live || (oor && zero_oor);
        float2 v[16];
        double es = 0.0;
        // one 64-bit row pointer, 32-bit offsets; the bounds-checked form only for windows that leave rx
        const float2* yrow = y + s;
        if (!zero && !oor) {
#pragma unroll
            for (int t = 0; t < 16; ++t) v[t] = yrow[l + t * NTR];
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int64_t j = s + l + t * NTR;
                v[t] = (!zero && j >= 0 && j < ylen) ? y[j] : make_float2(0.f, 0.f);
            }
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float2 a = xr[t], b = v[t];
            es += (double)b.x * b.x + (double)b.y * b.y;
            // conj(x * y): the inverse butterflies then deliver conj(FFT(x * y))
            v[t] = make_float2(a.x * b.x - a.y * b.y, -(a.x * b.y + a.y * b.x));
        }
        // window energy: lanes of the wave by shuffles; across the waves of a wide row through LDS, published by the
        // barrier that ends the first pass
        double e = wave_sum(es);
        if (WPR > 1 && lane == 0) s_e[it & 1][wave] = e;
        // (opaque copy of the row-local id: otherwise the twiddle powers and LDS addresses of all passes are hoisted
        // out of the row loop as loop invariants -- ~150 registers)
        int lo = l;
        asm volatile("" : "+v"(lo));
        pd_fft<LOGN>(buf, tw, lo, v);
        if (WPR > 1) {
            e = 0.0;
#pragma unroll
            for (int w = 0; w < WPR; ++w) e += s_e[it & 1][(wave / WPR) * WPR + w];
        }
        // normalisation as the unfused path rounds it: inv = (float)(1 / (sqrt(E) * ||x||)), applied to the amplitude
        const float inv = zero ? 0.f : (float)(1.0 / (sqrt(e) * sqrt(xnorm2)));
        float bv = -1.f;
        uint32_t bi = 0;
        float* prow = (plane && live) ? plane + row * N : nullptr;
        float2* crow = (cplane && live) ? cplane + row * N : nullptr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int idx = pd_out_index<LOGN>(l, r);
            const float zr = v[r].x * inv, zi = v[r].y * inv;
            const float val = zr * zr + zi * zi;
            if (prow) prow[idx] = val;
            if (crow) crow[idx] = make_float2(zr, -zi);
            if (val > bv || (val == bv && (uint32_t)idx < bi)) {  // first index of the maximum; NaN never wins
                bv = val;
                bi = (uint32_t)idx;
            }
        }
        if (qf2 || fidx) {
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                if (o < NTR) {
                    const float ov = __shfl_xor(bv, o, 64);
                    const uint32_t oi = __shfl_xor(bi, o, 64);
                    if (ov > bv || (ov == bv && oi < bi)) {
                        bv = ov;
                        bi = oi;
                    }
                }
            }
            if (WPR > 1) {
                if (lane == 0) {
                    s_bv[it & 1][wave] = bv;
                    s_bi[it & 1][wave] = bi;
                }
                __syncthreads();
                bv = -1.f;
                bi = 0;
#pragma unroll
                for (int w = 0; w < WPR; ++w) {
                    const float ov = s_bv[it & 1][(wave / WPR) * WPR + w];
                    const uint32_t oi = s_bi[it & 1][(wave / WPR) * WPR + w];
                    if (ov > bv || (ov == bv && oi < bi)) {
                        bv = ov;
                        bi = oi;
                    }
                }
            }
            if (live && l == 0) {
                if (bv < 0.f) {  // all-NaN row (zero-energy window): the reference's zero-initialised workspace
                    bv = 0.f;
                    bi = 0;
                }
                if (qf2) qf2[row] = bv;
                if (fidx) fidx[row] = bi;
            }
        }
        // (no barrier here: the last pass wrote nothing after its barrier, so the next row may overwrite the image)
    }
}

int pd_twiddles(int device, const float2** out) {
    static std::mutex mu;
    static std::vector<float2*> per_dev;
    std::lock_guard<std::mutex> lk(mu);
    if ((int)per_dev.size() <= device) per_dev.resize(device + 1, nullptr);
    if (!per_dev[device]) {
        std::vector<std::complex<float>> t(PD_TWN);
        for (int q = 0; q < PD_TWN; ++q) {
            const double ph = 2.0 * M_PI * (double)q / (double)PD_TWN;
            t[q] = std::complex<float>((float)std::cos(ph), (float)std::sin(ph));
        }
        float2* d = nullptr;
        CAF_HIP_TRY(hipMalloc((void**)&d, (size_t)PD_TWN * 8));
        const hipError_t e = hipMemcpy(d, t.data(), (size_t)PD_TWN * 8, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(d);
            CAF_HIP_TRY(e);
        }
        per_dev[device] = d;
    }
    *out = per_dev[device];
    return CAF_OK;
}

template <int LOGN>
int pd_launch(const float2* x, const float2* y, int64_t ylen, const float2* tw, int64_t start, int64_t step, int64_t num,
              int32_t zero_oor, float* qf2, uint32_t* fidx, float* plane, float2* cplane, hipStream_t st) {
    constexpr int N = 1 << LOGN, NTR = N / 16, WG = NTR > 256 ? NTR : 256, RPW = WG / NTR;
    const size_t lds = (size_t)RPW * (N + N / 16) * sizeof(float2);
    static bool attr_set = false;  // (per instantiation; benign race)
    if (!attr_set) {
        CAF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_perdelay_fused<LOGN>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    // enough workgroups to fill the chip several times over, few enough that the cutout load is amortised
    const int64_t groups = (num + RPW - 1) / RPW;
    int32_t rows_per_wg = (int32_t)std::max<int64_t>(1, std::min<int64_t>(16, groups / 4096));
    const int64_t nwg = (groups + rows_per_wg - 1) / rows_per_wg;
    CAF_REQUIRE(nwg <= 0x7fffffff, "caf_xcorr_perdelay: too many delays for one launch");
    hipLaunchKernelGGL(k_perdelay_fused<LOGN>, dim3((unsigned)nwg), dim3(WG), lds, st, x, y, ylen, tw, start, step, num,
                       rows_per_wg, zero_oor, qf2, fidx, plane, cplane);
    return CAF_OK;
}

}  // namespace

bool perdelay_fused_ok(int32_t n) { return n >= 64 && n <= 16384 && (n & (n - 1)) == 0; }

int launch_perdelay_fused(const float2* x, int32_t n, const float2* y, int64_t ylen, int64_t start, int64_t step,
                          int64_t num, int32_t zero_oor, float* qf2, uint32_t* fidx, float* plane, float2* cplane,
                          hipStream_t st) {
    int dev = 0;
    CAF_HIP_TRY(hipGetDevice(&dev));
    const float2* tw = nullptr;
    int rc = pd_twiddles(dev, &tw);
    if (rc) return rc;
    switch (n) {
        case 64: return pd_launch<6>(x, y, ylen, tw, start, step, num, zero_oor, qf2, fidx, plane, cplane, st);
        case 128: return pd_launch<7>(x, y, ylen, tw, start, step, num, zero_oor, qf2, fidx, plane, cplane, st);
        case 256: return pd_launch<8>(x, y, ylen, tw, start, step, num, zero_oor, qf2, fidx, plane, cplane, st);
        case 512: return pd_launch<9>(x, y, ylen, tw, start, step, num, zero_oor, qf2, fidx, plane, cplane, st);
        case 1024: return pd_launch<10>(x, y, ylen, tw, start, step, num, zero_oor, qf2, fidx, plane, cplane, st);
        case 2048: return pd_launch<11>(x, y, ylen, tw, start, step, num, zero_oor, qf2, fidx, plane, cplane, st);
        case 4096: return pd_launch<12>(x, y, ylen, tw, start, step, num, zero_oor, qf2, fidx, plane, cplane, st);
        case 8192: return pd_launch<13>(x, y, ylen, tw, start, step, num, zero_oor, qf2, fidx, plane, cplane, st);
        case 16384: return pd_launch<14>(x, y, ylen, tw, start, step, num, zero_oor, qf2, fidx, plane, cplane, st);
    }
    set_error("launch_perdelay_fused: unsupported length");
    return CAF_ERR_INVALID;
}

}  // namespace caf
