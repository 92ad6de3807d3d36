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
#include <algorithm>
#include <mutex>
#include <vector>
#include <complex>

#include "caf_internal.h"
#include "caf_energy.h"
#include "caf_ldsfft.h"

namespace caf {

namespace {

// global load of element `elem` of a complex64 array: when `base` is wave-uniform this is the scalar-base form
// (s[base] + 32-bit VGPR byte offset), so no 64-bit per-thread address is kept alive across the row loop
#define PD_AS1 __attribute__((address_space(1)))
__device__ __forceinline__ float2 pd_ld2(const float2* base, uint32_t elem) {
    const uint64_t u = *reinterpret_cast<const PD_AS1 uint64_t*>((const PD_AS1 char*)base + (elem << 3));
    float2 r;
    __builtin_memcpy(&r, &u, 8);
    return r;
}
__device__ __forceinline__ const float2* pd_uniform(const float2* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const float2*>(((uint64_t)hi << 32) | lo);
}

// WG threads = max(256, N/16); rows handled concurrently RPW = WG / (N/16); each workgroup walks `rows_per_wg`
// consecutive groups of RPW rows (neighbouring rows share all but one sample of their rx windows: L1/L2 hits).
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
    __shared__ unsigned long long s_key[2][NW];  // over consecutive rows so that a slot is rewritten only two barriers later
    const int tid = threadIdx.x;
    const int rl = tid / NTR, l = tid - rl * NTR;  // row slot of this thread, row-local id
    float2* buf = s_buf + rl * (N + N / 16);
    const int lane = tid & 63, wave = tid >> 6;

    // ||x||^2 in float64 from the thread's 16 cutout points (pass-1 positions); the cutout stays in registers across
    // the workgroup's rows (since the row epilogue shrank it fits at every size: 4096 x 1e6 rows 7.1 -> 6.4 ms against
    // re-reading it per row from L1/L2; two sizes keep 28 / 40 bytes of scratch outside the row loop)
    float2 xr[16];
    double xs = 0.0;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const float2 a = x[l + t * NTR];
        xr[t] = a;
        xs += (double)a.x * a.x + (double)a.y * a.y;
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
        // (opaque copy of the row-local id: otherwise the load offsets, the twiddle powers and the LDS addresses of all
        // passes are hoisted out of the row loop as loop invariants -- ~150 registers, most of them spilled)
        int lo = l;
        asm volatile("" : "+v"(lo));
        // one row pointer (wave-uniform, i.e. scalar, when a row fills the workgroup), 32-bit offsets; the
        // bounds-checked form only for windows that leave rx
        const float2* yrow = y + s;
        if (RPW == 1) yrow = pd_uniform(yrow);
        if (!zero && !oor) {
#pragma unroll
            for (int t = 0; t < 16; ++t) v[t] = pd_ld2(yrow, (uint32_t)(lo + t * NTR));
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
        pd_fft<LOGN>(buf, tw, lo, v);
        if (WPR > 1) {
            e = 0.0;
#pragma unroll
            for (int w = 0; w < WPR; ++w) e += s_e[it & 1][(wave / WPR) * WPR + w];
        }
        // normalisation as the unfused path applies it: inv = (float)(1 / (sqrt(E) * ||x||)) on the amplitude.  Evaluated as
        // rsq(E ||x||^2) + one Newton step (2^-45 or better before the rounding to float32) instead of a float64 square root
        // and a float64 division per row and thread: those two were ~40 half-rate instructions of the ~1350 of a row.
        // E = 0 (all-zero window): rsq = inf, 0 * inf = NaN -> NaN row, as before.
        float inv = 0.f;
        if (!zero) {
            const double a = e * xnorm2;
            const double y0 = __builtin_amdgcn_rsq(a);
            inv = (float)__builtin_fma(__builtin_fma(-(a * y0), 0.5 * y0, 0.5), y0, y0);
        }
        // (explicit fma: the planes and the row maximum must see the same bits whatever the compiler contracts where)
        if (plane || cplane) {
            float* prow = (plane && live) ? plane + row * N : nullptr;
            float2* crow = (cplane && live) ? cplane + row * N : nullptr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int idx = pd_out_index<LOGN>(lo, r);  // (from the opaque copy: 16 loop-invariant indices otherwise)
                const float zr = v[r].x * inv, zi = v[r].y * inv;
                if (prow) prow[idx] = __builtin_fmaf(zr, zr, zi * zi);
                if (crow) crow[idx] = make_float2(zr, -zi);
            }
        }
        // First index of the row maximum; NaN never wins.  Within the thread the registers are visited in ascending
        // index order (pd_out_index: t-major), so a strict comparison keeps the first; across lanes and waves one
        // unsigned maximum of the key (value bits, ~index): values are >= +0, whose bit patterns order like the numbers.
        constexpr int RLE = (LOGN % 4) ? (1 << (LOGN % 4)) : 16, NQE = 16 / RLE;
        float bv = -1.f;
        uint32_t br = 0;
#pragma unroll
        for (int t = 0; t < RLE; ++t) {
#pragma unroll
            for (int q = 0; q < NQE; ++q) {
                const int r = q * RLE + t;
                const float zr = v[r].x * inv, zi = v[r].y * inv;
                const float val = __builtin_fmaf(zr, zr, zi * zi);
                if (val > bv) {
                    bv = val;
                    br = (uint32_t)r;
                }
            }
        }
        if (qf2 || fidx) {
            asm volatile("" : "+v"(br));  // (the selects above on 0 .. 15 -- inline constants -- and ONE index computation)
            const uint32_t bi = (uint32_t)lo + (br / RLE) * NTR + (br % RLE) * (N / RLE);  // = pd_out_index(lo, br)
            // (a thread that saw only NaNs offers key 0: it loses against every real value, and an all-NaN row -- a
            // zero-energy window -- reports (NaN, 0): the reference's pmax / ||cutout||^2 / 0, xcorrRoutines.py:527-528,
            // IppXcorrFFT.cpp:174; a zero ROW of the out-of-range rule has inv = 0, all values +0, key != 0: (0, 0))
            unsigned long long key = bv < 0.f ? 0ull : (((unsigned long long)__float_as_uint(bv) << 32) | (uint32_t)~bi);
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                if (o < NTR) {
                    const unsigned long long ok = __shfl_xor(key, o, 64);
                    key = ok > key ? ok : key;
                }
            }
            if (WPR > 1) {
                if (lane == 0) s_key[it & 1][wave] = key;
                __syncthreads();
                key = 0ull;
#pragma unroll
                for (int w = 0; w < WPR; ++w) {
                    const unsigned long long ok = s_key[it & 1][(wave / WPR) * WPR + w];
                    key = ok > key ? ok : key;
                }
            }
            if (live && l == 0) {
                if (qf2) qf2[row] = key ? __uint_as_float((uint32_t)(key >> 32)) : __builtin_nanf("");
                if (fidx) fidx[row] = key ? ~(uint32_t)key : 0u;
            }
        }
        // (no barrier here: the last pass wrote nothing after its barrier, so the next row may overwrite the image)
    }
}

}  // namespace

int lds_fft_twiddles(int device, const float2** out) {
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

// ----------------------------------------------------------------------------------------
// Forward spectra of the overlap-save blocks of the LDS engines (B = 16384): X[b][m] = sum_n rx[src0 + b step + n]
// e^{-j 2 pi m n / B} (zeros past the end of rx), unnormalised -- gather + transform in ONE launch instead of
// k_gather_blocks (8 B written + 8 B read back per point) followed by ~60 batched rocFFT launches.
// forward = conj(IDFT(conj x)) on the shared in-LDS transform.  Replaces cuFFT's forward plan of the reference's
// batched xcorr (xcorrRoutines.py:1221-1232) on the hot path; rocFFT keeps the one-off template spectra.
// The block's samples are exactly the ones the sliding energies of its delays need (step + N - 1 = B), so with
// inv_e != NULL the same launch also writes 1 / sum_g sum_{n in group g} |rx[d + n]|^2 for the block's delays
// (filter.cu:291-347 + multiplySlices.cu:190-204 in the reference): |x|^2 in float64 into the idle LDS image, a
// block-wide exclusive prefix, differences per group -- no pass over rx of its own and no float64 prefix in HBM.
// One workgroup per CU walks blocks b, b + grid, ...: the next block's samples are in flight while this one is
// transformed (a workgroup owns the CU's LDS, so nothing else could hide that latency).
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_block_spectra(const float2* __restrict__ rx, int64_t rx_len, int64_t src0,
                                                        int32_t step, int64_t nblk, const float2* __restrict__ tw,
                                                        float2* __restrict__ xb, float* __restrict__ inv_e,
                                                        int64_t num_shifts, const int32_t* __restrict__ gstart,
                                                        const int32_t* __restrict__ glen, int32_t ngroups) {
    constexpr int LOGN = 14, N = 1 << LOGN, NTR = N / 16;
    extern __shared__ __attribute__((aligned(16))) float2 s_bs[];
    __shared__ double s_wtot[16];
    __shared__ double s_total;
    __shared__ double s_chunk[N / 64];
    const int l = threadIdx.x;
    // conj(x) of the block's 16 points of this thread (zeros past the end of rx)
    auto load = [&](int64_t b, float2 (&d)[16], int lo) {
        const int64_t s0 = src0 + b * step;
        if (s0 + N <= rx_len) {
            const float2* p = rx + s0;  // whole block inside rx: one 64-bit base, 32-bit offsets
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float2 x = p[lo + t * NTR];
                d[t] = make_float2(x.x, -x.y);
            }
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int64_t i = s0 + lo + t * NTR;
                float2 x = make_float2(0.f, 0.f);
                if (i < rx_len) x = rx[i];
                d[t] = make_float2(x.x, -x.y);
            }
        }
    };
    float2 v[16], nx[16];
    int64_t b = blockIdx.x;
    if (b < nblk) load(b, v, l);
    for (; b < nblk; b += gridDim.x) {
        // (opaque copy of the thread id: otherwise the load offsets, twiddle powers and LDS addresses of all passes are
        // hoisted out of the block loop as loop invariants and spilled)
        int lo = l;
        asm volatile("" : "+v"(lo));
        if (inv_e) {
            // P[m] = sum_{i<m} |x_i|^2 over the block, element m at m + (m >> 4) (a thread's 16 consecutive values at an
            // odd pitch); the image is free here: the previous transform ended with a barrier and wrote nothing after it
            double* s_p = reinterpret_cast<double*>(s_bs);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int m = lo + t * NTR;
                s_p[m + (m >> 4)] = (double)v[t].x * (double)v[t].x + (double)v[t].y * (double)v[t].y;
            }
            __syncthreads();
            double tot = 0.0;
#pragma unroll
            for (int j = 0; j < 16; ++j) tot += s_p[17 * lo + j];
            const int lane = lo & 63, wave = lo >> 6;
            {   // energies of the block's 64-sample chunks (4 lanes x 16 samples), plain sums: for windows the prefix cannot resolve
                double c4 = tot;
                c4 += __shfl_xor(c4, 1, 64);
                c4 += __shfl_xor(c4, 2, 64);
                if ((lane & 3) == 0) s_chunk[lo >> 2] = c4;
            }
            double incl = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const double u = __shfl_up(incl, o, 64);
                if (lane >= o) incl += u;
            }
            if (lane == 63) s_wtot[wave] = incl;
            double base = __shfl_up(incl, 1, 64);  // exclusive value from the neighbour (no inclusive-minus-own)
            if (lane == 0) base = 0.0;
            __syncthreads();
            for (int w = 0; w < wave; ++w) base += s_wtot[w];
            // in place: value -> exclusive prefix (each thread touches only its own 16 elements)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const double e = s_p[17 * lo + j];
                s_p[17 * lo + j] = base;
                base += e;
            }
            if (lo == 1023) s_total = base;  // P[16384]
            __syncthreads();
            auto P = [&](int m) { return m >= N ? s_total : s_p[m + (m >> 4)]; };
            const int64_t i0 = b * step;
            const int64_t s0 = src0 + b * step;  // the block's first sample in rx
            // a window the block's prefix does not resolve (caf_internal.h, CAF_ENERGY_RESOLVED): summed again without any
            // subtraction -- edge samples re-read from rx (they are in the L2: this block has just loaded them), whole chunks
            // from s_chunk
            auto direct = [&](int a, int e_) {
                auto smp = [&](int m) {
                    const int64_t j = s0 + m;
                    return j < rx_len ? sample_energy(rx[j]) : 0.0;
                };
                double e = 0.0;
                const int h = (a + 63) & ~63, t = e_ & ~63;
                if (h >= t) {
                    for (int m = a; m < e_; ++m) e += smp(m);
                    return e;
                }
                for (int m = a; m < h; ++m) e += smp(m);
                for (int c = h >> 6; c < (t >> 6); ++c) e += s_chunk[c];
                for (int m = t; m < e_; ++m) e += smp(m);
                return e;
            };
            for (int k = lo; k < step; k += 1024) {
                const int64_t i = i0 + k;
                if (i < num_shifts) {
                    double e = 0.0;
                    for (int g = 0; g < ngroups; ++g) {
                        const int a = k + gstart[g];
                        const double pb = P(a + glen[g]), d = pb - P(a);
                        e += d > CAF_ENERGY_RESOLVED * pb ? d : direct(a, a + glen[g]);
                    }
                    // (a window of zeros -- a gap in a recording -- is the reference's 0 / 0 = NaN, and only that is)
                    inv_e[i] = e > 0.0 ? (float)(1.0 / e) : __builtin_nanf("");
                }
            }
            __syncthreads();  // the image is overwritten by the transform's first pass
        }
        const int64_t bn = b + gridDim.x;
        if (bn < nblk) load(bn, nx, lo);
        pd_fft<LOGN>(s_bs, tw, lo, v);
        float2* o = xb + b * N;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[pd_out_index<LOGN>(lo, r)] = make_float2(v[r].x, -v[r].y);
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = nx[t];
    }
}

int launch_block_spectra(const float2* rx, int64_t rx_len, int64_t src0, int32_t step, int64_t nblk, float2* xb,
                         hipStream_t st, float* inv_e, int64_t num_shifts, const int32_t* gstart, const int32_t* glen,
                         int32_t ngroups) {
    int dev = 0;
    CAF_HIP_TRY(hipGetDevice(&dev));
    const float2* tw = nullptr;
    const int rc = lds_fft_twiddles(dev, &tw);
    if (rc) return rc;
    const size_t lds = (size_t)(16384 + 1024) * sizeof(float2);
    static std::mutex mu;
    static std::vector<char> attr_set;
    {
        std::lock_guard<std::mutex> lk(mu);
        if ((int)attr_set.size() <= dev) attr_set.resize(dev + 1, 0);
        if (!attr_set[dev]) {
            CAF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_block_spectra),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set[dev] = 1;
        }
    }
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const unsigned grid = (unsigned)std::min<int64_t>(nblk, std::max(cus, 1));
    hipLaunchKernelGGL(k_block_spectra, dim3(grid), dim3(1024), lds, st, rx, rx_len, src0, step, nblk, tw, xb, inv_e,
                       num_shifts, gstart, glen, ngroups);
    CAF_HIP_TRY(hipGetLastError());
    return CAF_OK;
}

// ----------------------------------------------------------------------------------------
// The same for the 32768-point blocks of fused_item2q (templates of 8193 ... 16384 samples), written directly in the
// layout that role reads: [block][parity c][16384], element m of a half (= X[2 m + c]) at (m & ~1023) + fp_tid_of(m & 1023)
// ("butterfly order", caf_fft_dev.h).  One decimation-in-frequency step in registers, then the shared 16384-point
// transform per half:
//     X[2 m]     = DFT_16384( x[n] + x[n + 16384] )[m]
//     X[2 m + 1] = DFT_16384( (x[n] - x[n + 16384]) e^{-j 2 pi n / 32768} )[m]
// One block per turn, both parities from one read of its samples; thread tid takes the logical butterfly l = fp_m2(tid), so that its
// outputs m = l + 1024 q' land on positions 1024 q' + tid: 512 contiguous bytes per wave and store.  Its loads are the
// four whole 128-byte lines per wave of fp_m2's mapping.  Replaces k_gather_blocks + batched rocFFT + k_parity_major.
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_block_spectra32(const float2* __restrict__ rx, int64_t rx_len, int64_t src0,
                                                          int32_t step, int64_t nblk, const float2* __restrict__ tw,
                                                          float2* __restrict__ xb2) {
    constexpr int LOGN = 14, N = 1 << LOGN, NTR = N / 16;
    extern __shared__ __attribute__((aligned(16))) float2 s_bs[];
    constexpr float C32[16] = {1.0f, 0.98078528f, 0.923879533f, 0.831469612f, 0.707106781f, 0.555570233f, 0.382683432f, 0.195090322f, 0.0f, -0.195090322f, -0.382683432f, -0.555570233f, -0.707106781f, -0.831469612f, -0.923879533f, -0.98078528f};
    constexpr float S32[16] = {0.0f, 0.195090322f, 0.382683432f, 0.555570233f, 0.707106781f, 0.831469612f, 0.923879533f, 0.98078528f, 1.0f, 0.98078528f, 0.923879533f, 0.831469612f, 0.707106781f, 0.555570233f, 0.382683432f, 0.195090322f};
    const int l0 = (int)fp_m2(threadIdx.x);
    float sn, cs;
    sincospif((float)l0 * (1.0f / 16384.0f), &sn, &cs);  // e^{+j 2 pi l / 32768}
    const float2 wl = make_float2(cs, sn);
    for (int64_t b = blockIdx.x; b < nblk; b += gridDim.x) {
        int lo = l0;
        asm volatile("" : "+v"(lo));  // (as in k_block_spectra: keeps the passes' addresses out of the block loop's invariants)
        const int64_t s0 = src0 + b * step;
        float2 v[16], d[16];  // conj(x_lo + x_hi), conj(x_lo - x_hi): the block is read once for both parities
        if (s0 + 2 * N <= rx_len) {
            const float2* p = rx + s0;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float2 a = p[lo + t * NTR], h = p[N + lo + t * NTR];
                v[t] = make_float2(a.x + h.x, -(a.y + h.y));
                d[t] = make_float2(a.x - h.x, h.y - a.y);
            }
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int64_t i = s0 + lo + t * NTR;
                float2 a = make_float2(0.f, 0.f), h = a;
                if (i < rx_len) a = rx[i];
                if (i + N < rx_len) h = rx[i + N];
                v[t] = make_float2(a.x + h.x, -(a.y + h.y));
                d[t] = make_float2(a.x - h.x, h.y - a.y);
            }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (c) {
#pragma unroll
                for (int t = 0; t < 16; ++t) v[t] = cmul(d[t], cmul(wl, make_float2(C32[t], S32[t])));  // * e^{+j 2 pi (l + 1024 t) / 32768}
            }
            pd_fft<LOGN>(s_bs, tw, lo, v);
            float2* o = xb2 + b * (2 * N) + c * N + threadIdx.x;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[pd_out_index<LOGN>(0, r)] = make_float2(v[r].x, -v[r].y);
            // (pd_fft's last pass ends with a barrier and writes nothing after it: the next transform may overwrite the image)
        }
    }
}

int launch_block_spectra32(const float2* rx, int64_t rx_len, int64_t src0, int32_t step, int64_t nblk, float2* xb2, hipStream_t st) {
    int dev = 0;
    CAF_HIP_TRY(hipGetDevice(&dev));
    const float2* tw = nullptr;
    const int rc = lds_fft_twiddles(dev, &tw);
    if (rc) return rc;
    const size_t lds = (size_t)(16384 + 1024) * sizeof(float2);
    static std::mutex mu;
    static std::vector<char> attr_set;
    {
        std::lock_guard<std::mutex> lk(mu);
        if ((int)attr_set.size() <= dev) attr_set.resize(dev + 1, 0);
        if (!attr_set[dev]) {
            CAF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_block_spectra32),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set[dev] = 1;
        }
    }
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const unsigned grid = (unsigned)std::min<int64_t>(nblk, std::max(cus, 1));
    if (grid == 0) return CAF_OK;
    hipLaunchKernelGGL(k_block_spectra32, dim3(grid), dim3(1024), lds, st, rx, rx_len, src0, step, nblk, tw, xb2);
    CAF_HIP_TRY(hipGetLastError());
    return CAF_OK;
}

// ----------------------------------------------------------------------------------------
// ... and for the 65536-point blocks of the folded role (templates of 16385 ... 262144 samples), in the layout fused_item2q<FOLD>
// reads: [block][parity c][16384] PAIRS (P_c[m], P_c[m + 16384]), P_c[m] = X[2 m + c], pair m at (m & ~1023) + fp_tid_of(m & 1023).
// Two decimation-in-frequency steps in registers, four 16384-point transforms per block: with n < 16384, x_q = x[n + 16384 q],
//     X[4 m + 2 d + c] = DFT_16384( [ (x_0 + (-1)^c x_2) + (-1)^d (-j)^c (x_1 + (-1)^c x_3) ] e^{-j 2 pi (c + 2 d) n / 65536} )[m]
// and P_c[2 m + d] = X[4 m + 2 d + c]: the transform (c, d) holds both members of the pairs with m = 2 m'' + d, m'' < 8192 (its
// outputs m'' and m'' + 8192).  The samples of a block are read once per c (the second time from the L2).  Thread tid takes
// the logical butterfly l whose outputs' pair positions run with the lane: bits of l = a fixed permutation of the bits of tid
// (below), so that a wave writes four runs of 256 bytes per store.  Replaces k_gather_blocks + batched rocFFT + k_parity_pairs.
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_block_spectra64(const float2* __restrict__ rx, int64_t rx_len, int64_t src0,
                                                          int32_t step, int64_t nblk, const float2* __restrict__ tw,
                                                          float4* __restrict__ xb2) {
    constexpr int LOGN = 14, N = 1 << LOGN, NTR = N / 16;
    extern __shared__ __attribute__((aligned(16))) float2 s_bs[];
    // e^{+j 2 pi t / 64}, t = 0 .. 63 (the part of the input twiddle that runs with the register: n = l + 1024 t)
    const uint32_t u = threadIdx.x & 511u, hb = threadIdx.x >> 9;
    // position bits (without bit 4, which is d): pos[3:0] = u[3:0], pos[5] = u[4], pos[9:6] = u[8:5]; the pair at that position
    // has source m2 = fp_m2(pos) = 2 lambda + d:  lambda[6:5] = u[1:0], lambda[2:1] = u[3:2], lambda[0] = u[4], lambda[8:7] = u[6:5],
    // lambda[4:3] = u[8:7]
    const uint32_t lam = ((u & 3u) << 5) | (((u >> 2) & 3u) << 1) | ((u >> 4) & 1u) | (((u >> 5) & 3u) << 7) | (((u >> 7) & 3u) << 3);
    const int l0 = (int)(lam + 512u * hb);
    const uint32_t pos0 = (u & 15u) | (((u >> 4) & 1u) << 5) | ((u >> 5) << 6);  // + 16 d
    float sn, cs;
    sincospif((float)l0 * (1.0f / 32768.0f), &sn, &cs);  // e^{+j 2 pi l / 65536}
    const float2 w1_0 = make_float2(cs, sn), w2_0 = cmul(w1_0, w1_0), w3_0 = cmul(w2_0, w1_0);
    for (int64_t b = blockIdx.x; b < nblk; b += gridDim.x) {
        int lo = l0;
        asm volatile("" : "+v"(lo));  // (keeps the passes' addresses out of the block loop's invariants)
        float2 w1 = w1_0, w2 = w2_0, w3 = w3_0;  // (... and the 48 products with the per-register constants: 96 registers hoisted otherwise)
        asm volatile("" : "+v"(w1.x), "+v"(w1.y), "+v"(w2.x), "+v"(w2.y), "+v"(w3.x), "+v"(w3.y));
        const int64_t s0 = src0 + b * step;
        const bool inside = s0 + 4 * N <= rx_len;
        const float2* p = rx + s0;  // (whole block inside rx: one 64-bit pointer, 32-bit offsets)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            // the inputs of the transforms (c, 0) and (c, 1) from one read of the block's four quarters (conjugated: the inverse
            // butterflies deliver conj(DFT)); the second one waits in registers while the first is transformed
            float2 v0[16], v1[16];
            constexpr double TWO_PI = 6.283185307179586476925;
            auto combine = [&](int t, float2 x0, float2 x1, float2 x2, float2 x3) __attribute__((always_inline)) {
                float2 sv, rv;
                if (c == 0) {
                    sv = make_float2(x0.x + x2.x, -(x0.y + x2.y));
                    rv = make_float2(x1.x + x3.x, -(x1.y + x3.y));
                } else {
                    sv = make_float2(x0.x - x2.x, -(x0.y - x2.y));
                    const float2 q = make_float2(x1.x - x3.x, -(x1.y - x3.y));  // conj(x1 - x3)
                    rv = make_float2(-q.y, q.x);                                // * (+j) = conj((-j)(x1 - x3))
                }
                // * e^{+j 2 pi (c + 2 d) (l + 1024 t) / 65536}
                const float2 a0 = cadd(sv, rv), a1 = csub(sv, rv);
                const int e0 = (c * t) & 63, e1 = ((c + 2) * t) & 63;
                v0[t] = c == 0 ? a0 : cmul(a0, cmul(w1, make_float2((float)__builtin_cos(TWO_PI * e0 / 64.0), (float)__builtin_sin(TWO_PI * e0 / 64.0))));
                v1[t] = a1;  // (its twiddle waits until the first transform is done: fewer values alive beside it)
                (void)e1;
            };
            if (inside) {
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int n = lo + t * NTR;
                    combine(t, p[n], p[n + N], p[n + 2 * N], p[n + 3 * N]);
                    if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // sixteen loads in flight, not sixty-four (registers)
                }
            } else {
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    // (the block runs past the end of rx: loads clamped to the last sample and zeroed by a select -- no branch per load)
                    const int64_t i = s0 + lo + t * NTR, last = rx_len - 1;
                    auto at = [&](int64_t j) {
                        const float2 v = rx[j < last ? j : last];
                        return j <= last ? v : make_float2(0.f, 0.f);
                    };
                    combine(t, at(i), at(i + N), at(i + 2 * N), at(i + 3 * N));
                    if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            }
            // register (q, t) of a transformed array holds m'' = l + 1024 (q + 4 t); the pairs of transform (c, d): t = 0, 1 with t + 2
            auto xform = [&](float2(&v)[16], uint32_t d) __attribute__((always_inline)) {
                pd_fft<LOGN>(s_bs, tw, lo, v);
                float4* o = xb2 + b * (int64_t)(2 * N) + (int64_t)c * N + pos0 + 16u * d;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float2 a = v[q * 4 + t], h = v[q * 4 + t + 2];  // (pd_out_index: r = q * RL + t, RL = 4)
                        // m = 2 m'' + d = 2 lambda + d + 1024 hb + 2048 (q + 4 t): chunk hb + 2 (q + 4 t)
                        o[1024 * ((int)hb + 2 * (q + 4 * t))] = make_float4(a.x, -a.y, h.x, -h.y);
                    }
                // (pd_fft's last pass ends with a barrier and writes nothing after it: the next transform may overwrite the image)
            };
            xform(v0, 0u);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int e1 = ((c + 2) * t) & 63;
                v1[t] = cmul(v1[t], cmul(c == 0 ? w2 : w3, make_float2((float)__builtin_cos(TWO_PI * e1 / 64.0), (float)__builtin_sin(TWO_PI * e1 / 64.0))));
            }
            xform(v1, 1u);
        }
    }
}

int launch_block_spectra64(const float2* rx, int64_t rx_len, int64_t src0, int32_t step, int64_t nblk, float2* xb2, hipStream_t st) {
    int dev = 0;
    CAF_HIP_TRY(hipGetDevice(&dev));
    const float2* tw = nullptr;
    const int rc = lds_fft_twiddles(dev, &tw);
    if (rc) return rc;
    const size_t lds = (size_t)(16384 + 1024) * sizeof(float2);
    static std::mutex mu;
    static std::vector<char> attr_set;
    {
        std::lock_guard<std::mutex> lk(mu);
        if ((int)attr_set.size() <= dev) attr_set.resize(dev + 1, 0);
        if (!attr_set[dev]) {
            CAF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_block_spectra64),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set[dev] = 1;
        }
    }
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const unsigned grid = (unsigned)std::min<int64_t>(nblk, std::max(cus, 1));
    if (grid == 0) return CAF_OK;
    hipLaunchKernelGGL(k_block_spectra64, dim3(grid), dim3(1024), lds, st, rx, rx_len, src0, step, nblk, tw, reinterpret_cast<float4*>(xb2));
    CAF_HIP_TRY(hipGetLastError());
    return CAF_OK;
}

namespace {

template <int LOGN>
int pd_launch(const float2* x, const float2* y, int64_t ylen, const float2* tw, int64_t start, int64_t step, int64_t num,
              int32_t zero_oor, float* qf2, uint32_t* fidx, float* plane, float2* cplane, hipStream_t st) {
    constexpr int N = 1 << LOGN, NTR = N / 16, WG = NTR > 256 ? NTR : 256, RPW = WG / NTR;
    const size_t lds = (size_t)RPW * (N + N / 16) * sizeof(float2);
    {
        const int rc_lds = allow_dynamic_lds(reinterpret_cast<const void*>(k_perdelay_fused<LOGN>), lds);
        if (rc_lds) return rc_lds;
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

// ----------------------------------------------------------------------------------------
// Cutouts of 100 / 1000 / 10000 samples (benchmark_xcorrs.py's default cutout is 1000): the same per-delay algorithm with
// radix-10 Stockham passes -- N / 10 threads per row, ten points each, two / three / four passes through a padded LDS image
// (one extra element per ten: the stride-10 stores of the first pass fall into disjoint banks) -- instead of the product
// matrix -> rocFFT rows -> argmax chain (>= 32 B of HBM traffic per element).  Rows are not wave-aligned (100 threads), so
// the row-wide results go through LDS: the window energy comes from the float64 prefix the caller has anyway (as in the
// three-kernel form, same normalisation: 1 / (sqrt(E) ||x||)), the (value, first index) maximum through one 64-bit
// `ds_max` per thread into the row's slot (values are >= +0: the key orders like the pair).
// ----------------------------------------------------------------------------------------
template <int P>
struct R10 {
    static constexpr int N = P == 2 ? 100 : (P == 3 ? 1000 : 10000);
    static constexpr int NTR = N / 10;
    static constexpr int WG = P == 2 ? 256 : (P == 3 ? 512 : 1024);
    static constexpr int RPW = WG / NTR;            // 25 / 5 / 1 rows per workgroup (250 / 500 / 1000 active threads)
    static constexpr int IMG = N + N / 10;          // padded row image
};
__device__ __forceinline__ int r10_pad(int a) { return a + a / 10; }

// one Stockham pass of radix 10 with Ns = 10^(pass index): butterfly j = l of the row; tw10 = W_N^q, q < N
template <int P, int NS, bool FIRST, bool LAST>
__device__ __forceinline__ void r10_pass(float2* __restrict__ buf, const float2* __restrict__ tw10, int l, bool active, float2 (&v)[10]) {
    constexpr int N = R10<P>::N, NTR = R10<P>::NTR;
    if (!FIRST) {
        if (active) {
            const float2* src = buf + r10_pad(l);
#pragma unroll
            for (int t = 0; t < 10; ++t) v[t] = src[t * (NTR + NTR / 10)];  // pad(l + t NTR) = pad(l) + 11 t NTR / 10 (NTR is a multiple of 10)
        }
        __syncthreads();  // every butterfly has its inputs: the image may be overwritten
    }
    const int k = l % NS;
    if (NS > 1) {
        const float2 w1 = tw10[k * (N / (NS * 10))];  // W_{10 NS}^k
        float2 p = w1;
        v[1] = cmul(v[1], p);
#pragma unroll
        for (int t = 2; t < 10; ++t) {
            p = cmul(p, w1);
            v[t] = cmul(v[t], p);
        }
    }
    idft10(v);
    if (!LAST) {
        if (active) {
            float2* dst = buf + r10_pad((l - k) * 10 + k);
#pragma unroll
            for (int t = 0; t < 10; ++t) dst[NS == 1 ? t : t * (NS + NS / 10)] = v[t];  // (NS == 1: 10 l + t stays inside one padded decade)
        }
        __syncthreads();
    }
}
template <int P>
__device__ __forceinline__ void r10_fft(float2* __restrict__ buf, const float2* __restrict__ tw10, int l, bool active, float2 (&v)[10]) {
    if constexpr (P == 2) {
        r10_pass<P, 1, true, false>(buf, tw10, l, active, v);
        r10_pass<P, 10, false, true>(buf, tw10, l, active, v);
    } else if constexpr (P == 3) {
        r10_pass<P, 1, true, false>(buf, tw10, l, active, v);
        r10_pass<P, 10, false, false>(buf, tw10, l, active, v);
        r10_pass<P, 100, false, true>(buf, tw10, l, active, v);
    } else {
        r10_pass<P, 1, true, false>(buf, tw10, l, active, v);
        r10_pass<P, 10, false, false>(buf, tw10, l, active, v);
        r10_pass<P, 100, false, false>(buf, tw10, l, active, v);
        r10_pass<P, 1000, false, true>(buf, tw10, l, active, v);
    }
}

template <int P>
__global__ __launch_bounds__(R10<P>::WG, 4) void k_perdelay_r10(  // (4 waves per SIMD: 128 VGPRs)
    const float2* __restrict__ x, const float2* __restrict__ y, int64_t ylen, const float2* __restrict__ tw10,
    const double* __restrict__ prefix, const double* __restrict__ xnorm, int64_t start, int64_t step, int64_t num,
    int32_t rows_per_wg, int32_t zero_oor, float* __restrict__ qf2, uint32_t* __restrict__ fidx, float* __restrict__ plane,
    float2* __restrict__ cplane) {
    constexpr int N = R10<P>::N, NTR = R10<P>::NTR, RPW = R10<P>::RPW, IMG = R10<P>::IMG;
    extern __shared__ __attribute__((aligned(16))) float2 s_buf[];  // RPW row images
    __shared__ unsigned long long s_key[2][RPW];
    const int tid = threadIdx.x;
    const bool active = tid < RPW * NTR;            // (the last threads of the workgroup only keep the barriers company)
    const int rl = active ? tid / NTR : 0, l = active ? tid - rl * NTR : 0;
    float2* buf = s_buf + rl * IMG;
    float2 xr[10];
#pragma unroll
    for (int t = 0; t < 10; ++t) xr[t] = x[l + t * NTR];
    const double xn = *xnorm;
    if (tid < 2 * RPW) s_key[tid / RPW][tid % RPW] = 0ull;
    __syncthreads();
    const int64_t row0 = (int64_t)blockIdx.x * rows_per_wg * RPW;
    for (int it = 0; it < rows_per_wg; ++it) {
        const int64_t row = row0 + (int64_t)it * RPW + rl;
        const bool live = active && row < num;
        const int64_t s = start + row * step;
        const bool oor = (s < 0) || (s + N > ylen);
        const bool zero = !live || (oor && zero_oor);
        float2 v[10];
        if (!zero && !oor) {
            const float2* yrow = y + s;
#pragma unroll
            for (int t = 0; t < 10; ++t) v[t] = yrow[l + t * NTR];
        } else {
#pragma unroll
            for (int t = 0; t < 10; ++t) {
                const int64_t j = s + l + t * NTR;
                v[t] = (!zero && j >= 0 && j < ylen) ? y[j] : make_float2(0.f, 0.f);
            }
        }
#pragma unroll
        for (int t = 0; t < 10; ++t) {
            const float2 a = xr[t], b = v[t];
            v[t] = make_float2(a.x * b.x - a.y * b.y, -(a.x * b.y + a.y * b.x));  // conj(x y): see k_perdelay_fused
        }
        // normalisation exactly as the three-kernel form computes it (k_sliding_multiply): window energy from the prefix
        float inv = 0.f;
        if (!zero) {
            const int64_t a = s < 0 ? 0 : (s > ylen ? ylen : s);
            int64_t b = s + N;
            b = b < 0 ? 0 : (b > ylen ? ylen : b);
            // 1 / (sqrt(E) ||x||) as rsq(E ||x||^2) + one Newton step (2^-45 or better before the rounding to float32), as in
            // k_perdelay_fused: a float64 square root and a float64 division per row and thread are ~40 half-rate
            // instructions.  E = 0: rsq = inf, 0 * inf = NaN -> NaN row, as before.
            const double en = window_energy(prefix, y, ylen, a, b) * (xn * xn);  // (exact where the difference is not: caf_energy.h)
            const double y0 = __builtin_amdgcn_rsq(en);
            inv = (float)__builtin_fma(__builtin_fma(-(en * y0), 0.5 * y0, 0.5), y0, y0);
        }
        r10_fft<P>(buf, tw10, l, active, v);
        // outputs: register t <-> spectrum index l + t NTR (ascending in t)
        if ((plane || cplane) && live) {
            float* prow = plane ? plane + row * N : nullptr;
            float2* crow = cplane ? cplane + row * N : nullptr;
#pragma unroll
            for (int t = 0; t < 10; ++t) {
                const float zr = v[t].x * inv, zi = v[t].y * inv;
                if (prow) prow[l + t * NTR] = __builtin_fmaf(zr, zr, zi * zi);
                if (crow) crow[l + t * NTR] = make_float2(zr, -zi);
            }
        }
        if (qf2 || fidx) {
            float bv = -1.f;
            uint32_t bt = 0;
#pragma unroll
            for (int t = 0; t < 10; ++t) {
                const float zr = v[t].x * inv, zi = v[t].y * inv;
                const float val = __builtin_fmaf(zr, zr, zi * zi);
                if (val > bv) {
                    bv = val;
                    bt = (uint32_t)t;
                }
            }
            const uint32_t bi = (uint32_t)l + bt * NTR;
            const unsigned long long key = bv < 0.f ? 0ull : (((unsigned long long)__float_as_uint(bv) << 32) | (uint32_t)~bi);
            if (live && key) atomicMax(&s_key[it & 1][rl], key);
            __syncthreads();
            if (live && l == 0) {
                const unsigned long long kk = s_key[it & 1][rl];
                if (qf2) qf2[row] = kk ? __uint_as_float((uint32_t)(kk >> 32)) : __builtin_nanf("");  // (all-NaN row = zero-energy window: (NaN, 0))
                if (fidx) fidx[row] = kk ? ~(uint32_t)kk : 0u;
                s_key[it & 1][rl] = 0ull;  // (next used two rows from now, behind the barriers of the row in between)
            }
        }
    }
}

template <int P>
int r10_launch(const float2* x, const float2* y, int64_t ylen, const float2* tw10, const double* prefix, const double* xnorm,
               int64_t start, int64_t step, int64_t num, int32_t zero_oor, float* qf2, uint32_t* fidx, float* plane,
               float2* cplane, hipStream_t st) {
    constexpr int WG = R10<P>::WG, RPW = R10<P>::RPW;
    const size_t lds = (size_t)RPW * R10<P>::IMG * sizeof(float2);
    {
        const int rc_lds = allow_dynamic_lds(reinterpret_cast<const void*>(k_perdelay_r10<P>), lds);
        if (rc_lds) return rc_lds;
    }
    const int64_t groups = (num + RPW - 1) / RPW;
    const int32_t rows_per_wg = (int32_t)std::max<int64_t>(1, std::min<int64_t>(16, groups / 4096));
    const int64_t nwg = (groups + rows_per_wg - 1) / rows_per_wg;
    CAF_REQUIRE(nwg <= 0x7fffffff, "caf_xcorr_perdelay: too many delays for one launch");
    hipLaunchKernelGGL(k_perdelay_r10<P>, dim3((unsigned)nwg), dim3(WG), lds, st, x, y, ylen, tw10, prefix, xnorm, start, step, num,
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
    int rc = lds_fft_twiddles(dev, &tw);
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

bool perdelay_decimal_ok(int32_t n) { return n == 100 || n == 1000 || n == 10000; }

// e^{+j 2 pi q / n}, q < n, for the radix-10 transforms (built once per device and length)
static int r10_twiddles(int device, int32_t n, const float2** out) {
    static std::mutex mu;
    static std::vector<std::pair<std::pair<int, int32_t>, float2*>> tabs;
    std::lock_guard<std::mutex> lk(mu);
    for (auto& e : tabs)
        if (e.first.first == device && e.first.second == n) {
            *out = e.second;
            return CAF_OK;
        }
    std::vector<std::complex<float>> t(n);
    for (int q = 0; q < n; ++q) {
        const double ph = 2.0 * M_PI * (double)q / (double)n;
        t[q] = std::complex<float>((float)std::cos(ph), (float)std::sin(ph));
    }
    float2* d = nullptr;
    CAF_HIP_TRY(hipMalloc((void**)&d, (size_t)n * 8));
    const hipError_t e = hipMemcpy(d, t.data(), (size_t)n * 8, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(d);
        CAF_HIP_TRY(e);
    }
    tabs.push_back({{device, n}, d});
    *out = d;
    return CAF_OK;
}

int launch_perdelay_decimal(const float2* x, int32_t n, const float2* y, int64_t ylen, const double* prefix, const double* xnorm,
                            int64_t start, int64_t step, int64_t num, int32_t zero_oor, float* qf2, uint32_t* fidx, float* plane,
                            float2* cplane, hipStream_t st) {
    int dev = 0;
    CAF_HIP_TRY(hipGetDevice(&dev));
    const float2* tw10 = nullptr;
    const int rc = r10_twiddles(dev, n, &tw10);
    if (rc) return rc;
    switch (n) {
        case 100: return r10_launch<2>(x, y, ylen, tw10, prefix, xnorm, start, step, num, zero_oor, qf2, fidx, plane, cplane, st);
        case 1000: return r10_launch<3>(x, y, ylen, tw10, prefix, xnorm, start, step, num, zero_oor, qf2, fidx, plane, cplane, st);
        case 10000: return r10_launch<4>(x, y, ylen, tw10, prefix, xnorm, start, step, num, zero_oor, qf2, fidx, plane, cplane, st);
    }
    set_error("launch_perdelay_decimal: unsupported length");
    return CAF_ERR_INVALID;
}

}  // namespace caf
