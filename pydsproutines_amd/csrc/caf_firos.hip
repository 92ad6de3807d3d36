// Overlap-save FIR for long tap sets: the lfilter semantics of filter_smtaps* (custom_kernels/filter.cu:9-181,
// filterRoutines.py:417-575: y = lfilter(taps, 1, [delay; x]) restricted to x, kept outputs [dsPhase::dsr]) evaluated
// block-wise in the frequency domain,
//   y_blk = IDFT_B( DFT_B(segment) * DFT_B(taps) ),  L = B - T + 1 new outputs per block,
// instead of T multiply-adds per output: HBM-bound again (16 B per output at dsr = 1) where the direct form is
// VALU-bound beyond ~80 taps.  The reference only has an unfinished attempt at this
// (custom_kernels/cufftdxKernels.cu:120-175).
//   * T <= 8192: ONE fused kernel per call -- segment load (complex64 or raw int16 IQ, carried-in history included)
//     -> forward transform in LDS -> x taps spectrum (L2-resident table) -> inverse transform in LDS -> kept outputs.
//     The two transforms chain in registers (caf_ldsfft.h), so a block makes one trip through HBM in each direction.
//   * longer tap sets: the same algebra on rocFFT rows (gather / multiply / scatter kernels here, orchestration in
//     caf_ops.hip), any length.
#include "caf_internal.h"
#include "caf_ldsfft.h"

namespace caf {

namespace {

__device__ __forceinline__ float2 fos_cvt(float2 v, float) { return v; }
__device__ __forceinline__ float2 fos_cvt(short2 v, float scale) { return make_float2((float)v.x * scale, (float)v.y * scale); }

// extended input: i in [-dlen, 0) -> carried-in history, [0, n) -> x, anything else 0
template <typename TIn>
__device__ __forceinline__ float2 fos_xe(const TIn* __restrict__ x, int64_t n, const TIn* __restrict__ delay, int32_t dlen,
                                         float scale, int64_t i) {
    if (i >= 0) return i < n ? fos_cvt(x[i], scale) : make_float2(0.f, 0.f);
    return (i >= -(int64_t)dlen) ? fos_cvt(delay[dlen + i], scale) : make_float2(0.f, 0.f);
}

// Ht[m] = DFT_B(taps zero-padded)[m] / B, one workgroup (row) of B/16 threads
template <int LOGN>
__global__ __launch_bounds__((1 << LOGN) / 16 > 64 ? (1 << LOGN) / 16 : 64) void k_fos_taps(const float* __restrict__ taps,
                                                                                         int32_t ntaps,
                                                                                         const float2* __restrict__ tw,
                                                                                         float2* __restrict__ ht) {
    constexpr int N = 1 << LOGN, NTR = N / 16;
    extern __shared__ __attribute__((aligned(16))) float2 s_buf[];
    const int l = threadIdx.x;
    float2 v[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int i = l + t * NTR;
        v[t] = make_float2(i < ntaps ? taps[i] : 0.f, 0.f);
    }
    pd_fft<LOGN>(s_buf, tw, l, v);  // IDFT(taps) = conj(DFT(taps)) for real taps
    const float s = 1.0f / (float)N;
#pragma unroll
    for (int r = 0; r < 16; ++r) ht[pd_out_index<LOGN>(l, r)] = make_float2(v[r].x * s, -v[r].y * s);
}

// One overlap-save block per row of B/16 threads; 256-thread workgroups hold 256 / (B/16) rows (B <= 4096).
template <int LOGN, typename TIn>
__global__ __launch_bounds__((1 << LOGN) / 16 > 256 ? (1 << LOGN) / 16 : 256, 4) void k_fir_os(
    const TIn* __restrict__ x, int64_t n, const TIn* __restrict__ delay, int32_t dlen, float scale,
    const float2* __restrict__ ht, const float2* __restrict__ tw, int32_t ntaps, int32_t dsr, int32_t phase,
    float2* __restrict__ out, int64_t nout, int64_t nblk, int64_t x_row_stride, int64_t out_row_stride) {
    constexpr int N = 1 << LOGN, NTR = N / 16, WG = NTR > 256 ? NTR : 256, RPW = WG / NTR;
    extern __shared__ __attribute__((aligned(16))) float2 s_buf[];
    const int tid = threadIdx.x;
    // (blockIdx.y: independent signals of n samples each, x_row_stride / out_row_stride apart -- the rows of upfirdn)
    x += (int64_t)blockIdx.y * x_row_stride;
    out += (int64_t)blockIdx.y * out_row_stride;
    const int rl = tid / NTR, l = tid - rl * NTR;
    float2* buf = s_buf + rl * (N + N / 16);
    const int64_t b = (int64_t)blockIdx.x * RPW + rl;
    const bool live = b < nblk;  // dead row slots run the barriers on zeros
    const int L = N - ntaps + 1;
    const int64_t seg0 = b * L - (ntaps - 1);  // first sample of the segment in x coordinates

    float2 v[16];
    if (live && seg0 >= 0 && seg0 + N <= n) {
        const TIn* xs = x + seg0;  // whole segment inside x: one 64-bit pointer, 32-bit offsets
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = fos_cvt(xs[l + t * NTR], scale);
    } else {
#pragma unroll
        for (int t = 0; t < 16; ++t)
            v[t] = live ? fos_xe(x, n, delay, dlen, scale, seg0 + l + t * NTR) : make_float2(0.f, 0.f);
    }
    // the taps spectrum at this thread's sixteen output positions: it depends on nothing the forward transform produces, so
    // it is fetched in front of it -- behind it, the loads' L2 round trip stood between the two transforms of every block
    // (1024 taps on 2^24 samples: 93 -> 84 us).  (A workgroup that loops over its blocks with the spectrum resident and
    // the next segment fetched under the inverse transform measured no better: profiles/r04/ab_fir_os_prefetch.log.)
    float2 hh[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) hh[r] = ht[pd_out_index<LOGN>(l, r)];
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t].y = -v[t].y;  // conj: the inverse butterflies deliver conj(DFT(segment))
    pd_fft<LOGN>(buf, tw, l, v);
    // spectrum x taps spectrum, straight into the input registers of the inverse transform (same positions)
    float2 w[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float2 h = hh[r];
        const float2 g = v[r];  // conj(F)
        w[pd_out_slot<LOGN>(r)] = make_float2(g.x * h.x + g.y * h.y, g.x * h.y - g.y * h.x);  // conj(g) * h
    }
    pd_fft<LOGN>(buf, tw, l, w);
    if (!live) return;
    // valid outputs: segment positions T-1 .. B-1 -> full-rate output b L + (idx - (T-1)); kept ones [phase::dsr]
    const int64_t o0 = b * L;
    if (dsr == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rel = pd_out_index<LOGN>(l, r) - (ntaps - 1);
            const int64_t o = o0 + rel;
            if (rel >= 0 && o < nout) out[o] = w[r];
        }
    } else {
        // (o0 + rel - phase) divisible by dsr: one 64-bit remainder per row, 32-bit arithmetic per output
        const int base = (int)(((o0 - phase) % dsr + dsr) % dsr);
        const int64_t q0 = (o0 - phase - base) / dsr;  // exact
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rel = pd_out_index<LOGN>(l, r) - (ntaps - 1);
            if (rel < 0) continue;
            const int z = base + rel;
            const int qd = z / dsr;
            if (qd * dsr != z) continue;
            const int64_t o = q0 + qd;
            if (o >= 0 && o < nout) out[o] = w[r];
        }
    }
}

// ---- long tap sets: rocFFT rows ----
// rows[b][i] = xe[b L - (T-1) + i], i < B (zero beyond the data)
template <typename TIn>
__global__ __launch_bounds__(256) void k_fos_gather(const TIn* __restrict__ x, int64_t n, const TIn* __restrict__ delay,
                                                    int32_t dlen, float scale, int64_t b0, int64_t L, int64_t B, int32_t ntaps,
                                                    float2* __restrict__ rows) {
    const int64_t b = b0 + blockIdx.y;
    const int64_t seg0 = b * L - (ntaps - 1);
    float2* dst = rows + (int64_t)blockIdx.y * B;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < B; i += (int64_t)gridDim.x * 256)
        dst[i] = fos_xe(x, n, delay, dlen, scale, seg0 + i);
}
// padded complex taps row for the spectrum
__global__ __launch_bounds__(256) void k_fos_taps_pad(const float* __restrict__ taps, int32_t ntaps, int64_t B,
                                                      float2* __restrict__ row) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < B; i += (int64_t)gridDim.x * 256)
        row[i] = make_float2(i < ntaps ? taps[i] : 0.f, 0.f);
}
// kept outputs of the rows: positions T-1 .. B-1 of row b -> full-rate index b L + rel
__global__ __launch_bounds__(256) void k_fos_scatter(const float2* __restrict__ rows, int64_t b0, int64_t L, int64_t B,
                                                     int32_t ntaps, int32_t dsr, int32_t phase, float2* __restrict__ out,
                                                     int64_t nout) {
    const int64_t b = b0 + blockIdx.y;
    const float2* src = rows + (int64_t)blockIdx.y * B + (ntaps - 1);
    // kept outputs o with b L <= phase + o dsr < (b + 1) L
    const int64_t lo = b * L, hi = lo + L;
    int64_t ofirst = lo <= phase ? 0 : (lo - phase + dsr - 1) / dsr;
    for (int64_t o = ofirst + (int64_t)blockIdx.x * 256 + threadIdx.x; o < nout; o += (int64_t)gridDim.x * 256) {
        const int64_t pos = phase + o * dsr;
        if (pos >= hi) break;
        out[o] = src[pos - lo];
    }
}

template <int LOGN>
int fos_taps_launch(const float* taps, int32_t ntaps, const float2* tw, float2* ht, hipStream_t st) {
    constexpr int N = 1 << LOGN, NTR = N / 16;
    const size_t lds = (size_t)(N + N / 16) * sizeof(float2);
    {
        const int rc_lds = allow_dynamic_lds(reinterpret_cast<const void*>(k_fos_taps<LOGN>), lds);
        if (rc_lds) return rc_lds;
    }
    hipLaunchKernelGGL(k_fos_taps<LOGN>, dim3(1), dim3(NTR), lds, st, taps, ntaps, tw, ht);
    return CAF_OK;
}

template <int LOGN, typename TIn>
int fos_launch(const TIn* x, int64_t n, const TIn* delay, int32_t dlen, float scale, const float2* ht, const float2* tw,
               int32_t ntaps, int32_t dsr, int32_t phase, float2* out, int64_t nout, hipStream_t st, int64_t rows = 1,
               int64_t x_row_stride = 0, int64_t out_row_stride = 0) {
    constexpr int N = 1 << LOGN, NTR = N / 16, WG = NTR > 256 ? NTR : 256, RPW = WG / NTR;
    const size_t lds = (size_t)RPW * (N + N / 16) * sizeof(float2);
    {
        const int rc_lds = allow_dynamic_lds(reinterpret_cast<const void*>(k_fir_os<LOGN, TIn>), lds);
        if (rc_lds) return rc_lds;
    }
    const int64_t L = N - ntaps + 1;
    const int64_t last = phase + (nout - 1) * (int64_t)dsr;  // last full-rate output wanted
    const int64_t nblk = last / L + 1;
    const int64_t nwg = (nblk + RPW - 1) / RPW;
    CAF_REQUIRE(nwg <= 0x7fffffff, "overlap-save FIR: too many blocks for one launch");
    CAF_REQUIRE(rows >= 1 && rows <= 65535, "overlap-save FIR: too many rows for one launch");
    hipLaunchKernelGGL((k_fir_os<LOGN, TIn>), dim3((unsigned)nwg, (unsigned)rows), dim3(WG), lds, st, x, n, delay, dlen, scale, ht, tw,
                       ntaps, dsr, phase, out, nout, nblk, x_row_stride, out_row_stride);
    return CAF_OK;
}

}  // namespace

// block size of the fused form for a tap count (0: too long, use the rocFFT rows)
int fir_os_fused_block(int32_t ntaps) {
    if (ntaps <= 256) return 1024;
    if (ntaps <= 1024) return 4096;
    if (ntaps <= 8192) return 16384;
    return 0;
}

// ht: scratch of fir_os_fused_block(ntaps) complex values
template <typename TIn>
static int fir_os_fused_t(const TIn* x, int64_t n, const float* taps, int32_t ntaps, const TIn* delay, int32_t dlen, float scale,
                          int32_t dsr, int32_t phase, float2* out, int64_t nout, float2* ht, hipStream_t st, int64_t rows = 1,
                          int64_t xs = 0, int64_t os = 0) {
    if (nout <= 0) return CAF_OK;
    int dev = 0;
    CAF_HIP_TRY(hipGetDevice(&dev));
    const float2* tw = nullptr;
    int rc = lds_fft_twiddles(dev, &tw);
    if (rc) return rc;
    switch (fir_os_fused_block(ntaps)) {
        case 1024:
            if ((rc = fos_taps_launch<10>(taps, ntaps, tw, ht, st))) return rc;
            return fos_launch<10, TIn>(x, n, delay, dlen, scale, ht, tw, ntaps, dsr, phase, out, nout, st, rows, xs, os);
        case 4096:
            if ((rc = fos_taps_launch<12>(taps, ntaps, tw, ht, st))) return rc;
            return fos_launch<12, TIn>(x, n, delay, dlen, scale, ht, tw, ntaps, dsr, phase, out, nout, st, rows, xs, os);
        case 16384:
            if ((rc = fos_taps_launch<14>(taps, ntaps, tw, ht, st))) return rc;
            return fos_launch<14, TIn>(x, n, delay, dlen, scale, ht, tw, ntaps, dsr, phase, out, nout, st, rows, xs, os);
    }
    set_error("overlap-save FIR: tap set too long for the fused form");
    return CAF_ERR_INVALID;
}
int launch_fir_os_fused(const float2* x, int64_t n, const float* taps, int32_t ntaps, const float2* delay, int32_t dlen,
                        int32_t dsr, int32_t phase, float2* out, int64_t nout, float2* ht, hipStream_t st, int64_t rows,
                        int64_t x_row_stride, int64_t out_row_stride) {
    return fir_os_fused_t<float2>(x, n, taps, ntaps, delay, dlen, 1.0f, dsr, phase, out, nout, ht, st, rows, x_row_stride, out_row_stride);
}
int launch_iq16_fir_os_fused(const int16_t* iq, int64_t n, float scale, const float* taps, int32_t ntaps, const int16_t* delay,
                             int32_t dlen, int32_t dsr, int32_t phase, float2* out, int64_t nout, float2* ht, hipStream_t st) {
    return fir_os_fused_t<short2>(reinterpret_cast<const short2*>(iq), n, taps, ntaps, reinterpret_cast<const short2*>(delay),
                                  dlen, scale, dsr, phase, out, nout, ht, st);
}

void launch_fos_gather(const float2* x, int64_t n, const float2* delay, int32_t dlen, int64_t b0, int64_t nb, int64_t L,
                       int64_t B, int32_t ntaps, float2* rows, hipStream_t st) {
    const unsigned gx = (unsigned)std::min<int64_t>((B + 255) / 256, 1024);
    hipLaunchKernelGGL(k_fos_gather<float2>, dim3(gx, (unsigned)nb), dim3(256), 0, st, x, n, delay, dlen, 1.0f, b0, L, B, ntaps,
                       rows);
}
void launch_fos_gather_iq16(const int16_t* x, int64_t n, float scale, const int16_t* delay, int32_t dlen, int64_t b0,
                            int64_t nb, int64_t L, int64_t B, int32_t ntaps, float2* rows, hipStream_t st) {
    const unsigned gx = (unsigned)std::min<int64_t>((B + 255) / 256, 1024);
    hipLaunchKernelGGL(k_fos_gather<short2>, dim3(gx, (unsigned)nb), dim3(256), 0, st, reinterpret_cast<const short2*>(x), n,
                       reinterpret_cast<const short2*>(delay), dlen, scale, b0, L, B, ntaps, rows);
}
void launch_fos_taps_pad(const float* taps, int32_t ntaps, int64_t B, float2* row, hipStream_t st) {
    hipLaunchKernelGGL(k_fos_taps_pad, dim3((unsigned)std::min<int64_t>((B + 255) / 256, 1024)), dim3(256), 0, st, taps, ntaps,
                       B, row);
}
void launch_fos_scatter(const float2* rows, int64_t b0, int64_t nb, int64_t L, int64_t B, int32_t ntaps, int32_t dsr,
                        int32_t phase, float2* out, int64_t nout, hipStream_t st) {
    const unsigned gx = (unsigned)std::min<int64_t>((L / dsr + 256) / 256, 1024);
    hipLaunchKernelGGL(k_fos_scatter, dim3(gx, (unsigned)nb), dim3(256), 0, st, rows, b0, L, B, ntaps, dsr, phase, out, nout);
}

}  // namespace caf
