// libcaf C-ABI, part 2: the per-delay path and the stand-alone kernel-level entry points
// (include/caf.h).  Host-side C++ that validates arguments, manages scratch and launches the
// gfx950 kernels of caf_rows.hip / caf_kernels.hip and batched rocFFT rows.
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "caf_internal.h"

using namespace caf;

namespace {

// Small cache of row-FFT plans keyed by (device, len, batch, inverse, inplace); plans are reused
// across calls (create -> use many), destroyed at process exit by the OS.
struct PlanKey {
    int dev;
    int64_t len, batch;
    int inv, inplace;
    bool operator<(const PlanKey& o) const {
        return std::tie(dev, len, batch, inv, inplace) < std::tie(o.dev, o.len, o.batch, o.inv, o.inplace);
    }
};
std::mutex g_plan_mu;
std::map<PlanKey, FftPlan> g_plans;

int get_row_plan(int64_t len, int64_t batch, bool inverse, bool inplace, FftPlan** out) {
    int dev = 0;
    CAF_HIP_TRY(hipGetDevice(&dev));
    PlanKey k{dev, len, batch, inverse ? 1 : 0, inplace ? 1 : 0};
    std::lock_guard<std::mutex> lk(g_plan_mu);
    auto it = g_plans.find(k);
    if (it == g_plans.end()) {
        if (g_plans.size() > 64) {  // bound the cache
            for (auto& kv : g_plans) kv.second.destroy();
            g_plans.clear();
        }
        FftPlan p;
        int rc = p.create(inverse, (size_t)len, (size_t)batch, (size_t)len, inplace);
        if (rc) {
            p.destroy();
            return rc;
        }
        it = g_plans.emplace(k, p).first;
    }
    *out = &it->second;
    return CAF_OK;
}

// rows FFT of a (rows, len) matrix, chunked so that the plan batch is bounded
int fft_rows(const float2* in, float2* out, int64_t rows, int64_t len, bool inverse, hipStream_t st) {
    if (rows <= 0) return CAF_OK;
    const bool inplace = (out == in);
    int64_t done = 0;
    while (done < rows) {
        // largest power-of-two chunk <= remaining keeps the number of distinct plans small
        int64_t chunk = 1;
        while (chunk * 2 <= rows - done && chunk * 2 * len <= ((int64_t)1 << 27)) chunk *= 2;
        FftPlan* p = nullptr;
        int rc = get_row_plan(len, chunk, inverse, inplace, &p);
        if (rc) return rc;
        rc = p->exec((void*)(in + done * len), inplace ? nullptr : (void*)(out + done * len), st);
        if (rc) return rc;
        done += chunk;
    }
    return CAF_OK;
}

struct Scratch {
    std::vector<void*> ptrs;
    template <typename T>
    int get(T** p, int64_t count) {
        void* q = nullptr;
        const int rc = pool_alloc(&q, std::max<int64_t>(count * (int64_t)sizeof(T), 16));
        if (rc) return rc;
        ptrs.push_back(q);
        *p = (T*)q;
        return CAF_OK;
    }
    ~Scratch() {
        for (void* q : ptrs) (void)pool_free(q);
    }
};

int energy_prefix(const float2* x, int64_t n, Scratch& sc, double** prefix, hipStream_t st) {
    double* tiles = nullptr;
    int rc = sc.get(&tiles, prefix_num_tiles(n) + 1024);
    if (rc) return rc;
    if ((rc = sc.get(prefix, energy_prefix_doubles(n)))) return rc;
    launch_energy_prefix(x, n, tiles, *prefix, st);
    return CAF_OK;
}

// Overlap-save FIR (caf_firos.hip).  Direct form costs ntaps multiply-adds per KEPT output, overlap-save ~130 flops
// per full-rate output: measured on 2^24 samples it is level with the direct kernels at ~96 taps per unit of
// decimation (0.10 vs 0.14 ms at 128 taps, dsr 1; 0.10 vs 0.11 ms at 256 taps, dsr 4) and ahead beyond, and it is the
// only form for tap sets longer than the direct kernels' LDS windows.  CAF_FIR_OS_MIN_TAPS overrides the 96
// (A/B switch; 0 = always).
bool fir_use_overlap_save(int32_t ntaps, int32_t dsr, int32_t direct_limit) {
    static const int min_taps = [] {
        const char* e = getenv("CAF_FIR_OS_MIN_TAPS");
        return e ? atoi(e) : 96;
    }();
    // decimation factors beyond the register-tiled polyphase kernel's window run on k_fir_decim (a tap and a sample read from LDS
    // per multiply-add): level with overlap-save at 64 taps, half its speed at 128 (2^24 int16 samples, /8: 133 against 77 us) --
    // there the 96 taps count as such, not per unit of decimation
    if (dsr >= 2 && !fir_poly_fits(ntaps, dsr) && ntaps >= std::max(min_taps, 1) && fir_os_fused_block(ntaps)) return true;
    return ntaps > direct_limit || (int64_t)ntaps > (int64_t)min_taps * dsr;
}

// is_iq16: x / delay are interleaved int16 IQ pairs scaled by `scale`, else complex64
int fir_overlap_save(const void* x, int64_t n, bool is_iq16, float scale, const float* taps, int32_t ntaps, const void* delay,
                     int32_t dlen, int32_t dsr, int32_t phase, float2* out, int64_t nout, hipStream_t st) {
    if (nout <= 0) return CAF_OK;
    Scratch sc;
    int rc;
    if (const int fb = fir_os_fused_block(ntaps)) {
        float2* ht = nullptr;
        if ((rc = sc.get(&ht, fb))) return rc;
        rc = is_iq16 ? launch_iq16_fir_os_fused((const int16_t*)x, n, scale, taps, ntaps, (const int16_t*)delay, dlen, dsr, phase,
                                                out, nout, ht, st)
                     : launch_fir_os_fused((const float2*)x, n, taps, ntaps, (const float2*)delay, dlen, dsr, phase, out, nout,
                                           ht, st);
        if (rc) return rc;
    } else {
        // long tap sets: rocFFT rows of B >= 4 ntaps points (>= 75 % new outputs per block)
        int64_t B = 65536;
        while (B < 4 * (int64_t)ntaps) B <<= 1;
        CAF_REQUIRE(B <= ((int64_t)1 << 26), "overlap-save FIR: more than 2^24 taps");
        const int64_t L = B - ntaps + 1;
        const int64_t last = phase + (nout - 1) * (int64_t)dsr;
        const int64_t nblk = last / L + 1;
        const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(nblk, 65535), ((int64_t)1 << 25) / B));
        float2 *rows = nullptr, *hrow = nullptr;
        if ((rc = sc.get(&rows, chunk * B)) || (rc = sc.get(&hrow, B))) return rc;
        launch_fos_taps_pad(taps, ntaps, B, hrow, st);
        if ((rc = fft_rows(hrow, hrow, 1, B, false, st))) return rc;
        for (int64_t b0 = 0; b0 < nblk; b0 += chunk) {
            const int64_t nb = std::min(chunk, nblk - b0);
            if (is_iq16)
                launch_fos_gather_iq16((const int16_t*)x, n, scale, (const int16_t*)delay, dlen, b0, nb, L, B, ntaps, rows, st);
            else
                launch_fos_gather((const float2*)x, n, (const float2*)delay, dlen, b0, nb, L, B, ntaps, rows, st);
            if ((rc = fft_rows(rows, rows, nb, B, false, st))) return rc;
            launch_rows_mul_vec(rows, B, 0, hrow, B, rows, B, B, nb, 1.0f / (float)B, st);
            if ((rc = fft_rows(rows, rows, nb, B, true, st))) return rc;
            launch_fos_scatter(rows, b0, nb, L, B, ntaps, dsr, phase, out, nout, st);
        }
    }
    // The scratch goes back to the pool on return.  Reuse is stream-ordered (caf_pool.hip): on the default stream a
    // later user is ordered behind the kernels above, so the call stays asynchronous like the direct form; a caller's
    // own stream is synchronised, because the next user of the block may sit on another stream.
    if (st != nullptr) CAF_HIP_TRY(hipStreamSynchronize(st));
    CAF_HIP_TRY(hipGetLastError());
    return CAF_OK;
}

// ---- chirp-Z constants of the zoom, computed like CZTCached / IppCZT32fc (spectralRoutines.py:239-267,
// CZT.cpp:89-140): float64 on the host, stored as complex64; W exponent = step (the labelled grid IS the evaluated
// grid), nfft = next 7-smooth length >= m + k - 1.  Cached per (device, m, span, step).
typedef std::complex<double> cd;
void host_fft(std::vector<cd>& a) {  // in-place forward DFT, length with prime factors <= 7 (recursive mixed radix)
    const size_t n = a.size();
    if (n <= 1) return;
    size_t p = 0;
    for (size_t q : {2, 3, 5, 7})
        if (n % q == 0) {
            p = q;
            break;
        }
    if (!p) {  // (not reached for 7-smooth lengths) plain DFT
        std::vector<cd> o(n);
        for (size_t k = 0; k < n; ++k) {
            cd acc = 0;
            for (size_t j = 0; j < n; ++j) acc += a[j] * std::polar(1.0, -2.0 * M_PI * (double)((j * k) % n) / (double)n);
            o[k] = acc;
        }
        a.swap(o);
        return;
    }
    const size_t m = n / p;
    std::vector<std::vector<cd>> sub(p, std::vector<cd>(m));
    for (size_t j = 0; j < n; ++j) sub[j % p][j / p] = a[j];
    for (auto& v : sub) host_fft(v);
    for (size_t k = 0; k < n; ++k) {
        cd acc = 0;
        for (size_t r = 0; r < p; ++r) acc += sub[r][k % m] * std::polar(1.0, -2.0 * M_PI * (double)((r * k) % n) / (double)n);
        a[k] = acc;
    }
}
int64_t next_fast_len7(int64_t n) {
    for (;; ++n) {
        int64_t v = n;
        for (int q : {2, 3, 5, 7})
            while (v % q == 0) v /= q;
        if (v == 1) return n;
    }
}
struct ZoomCzt {
    int nfft = 0, k = 0;
    float2 *aa = nullptr, *fv = nullptr, *wws = nullptr;  // device, never freed (a handful of small arrays per process)
};
std::mutex g_zoom_mu;
std::map<std::tuple<int, int, double, double>, ZoomCzt> g_zoom_czt;

int zoom_num_bins(double span, double step) { return (int)std::floor(2.0 * span / step + 1.0 + 1e-9); }

int zoom_constants(int dev, int m, double span, double step, ZoomCzt* out) {
    std::lock_guard<std::mutex> lk(g_zoom_mu);
    const auto key = std::make_tuple(dev, m, span, step);
    auto it = g_zoom_czt.find(key);
    if (it != g_zoom_czt.end()) {
        *out = it->second;
        return CAF_OK;
    }
    ZoomCzt z;
    z.k = zoom_num_bins(span, step);
    z.nfft = (int)next_fast_len7((int64_t)m + z.k - 1);
    const int k = z.k, nfft = z.nfft;
    const int lo = -m + 1, hi = std::max(k - 1, m - 1);
    std::vector<cd> ww(hi - lo + 1);
    for (int i = lo; i <= hi; ++i) {
        double cyc = step * ((double)i * (double)i / 2.0);
        cyc -= std::floor(cyc);
        ww[i - lo] = std::polar(1.0, -2.0 * M_PI * cyc);
    }
    std::vector<cd> fv(nfft, cd(0, 0));
    for (int i = 0; i < k - 1 + m; ++i) fv[i] = 1.0 / ww[i];
    host_fft(fv);
    std::vector<std::complex<float>> aa(m), fvf(nfft), wws(k);
    for (int n = 0; n < m; ++n) {
        double cyc = span * (double)n;  // exp(-2 pi j f1 n) with f1 = -span
        cyc -= std::floor(cyc);
        const cd v = std::polar(1.0, 2.0 * M_PI * cyc) * ww[m - 1 + n];
        aa[n] = std::complex<float>((float)v.real(), (float)v.imag());
    }
    for (int i = 0; i < nfft; ++i) fvf[i] = std::complex<float>((float)fv[i].real(), (float)fv[i].imag());
    for (int i = 0; i < k; ++i) wws[i] = std::complex<float>((float)ww[m - 1 + i].real(), (float)ww[m - 1 + i].imag());
    CAF_HIP_TRY(hipMalloc((void**)&z.aa, (size_t)m * 8));
    CAF_HIP_TRY(hipMalloc((void**)&z.fv, (size_t)nfft * 8));
    CAF_HIP_TRY(hipMalloc((void**)&z.wws, (size_t)k * 8));
    CAF_H2D(z.aa, aa.data(), (size_t)m * 8);
    CAF_H2D(z.fv, fvf.data(), (size_t)nfft * 8);
    CAF_H2D(z.wws, wws.data(), (size_t)k * 8);
    if (g_zoom_czt.size() > 64) g_zoom_czt.clear();  // (leaks a few hundred KB at worst; bounded)
    g_zoom_czt[key] = z;
    *out = z;
    return CAF_OK;
}

}  // namespace

extern "C" {

int32_t caf_fft_rows(const float* d_in, float* d_out, int64_t rows, int64_t len, int32_t inverse, void* stream) {
    CAF_REQUIRE(d_in && d_out && rows >= 0 && len >= 1, "caf_fft_rows: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    int rc = fft_rows((const float2*)d_in, (float2*)d_out, rows, len, inverse != 0, st);
    if (rc) return rc;
    if (inverse) launch_scale((float2*)d_out, rows * len, 1.0f / (float)len, st);  // numpy/cupy ifft normalisation
    CAF_HIP_TRY(hipGetLastError());
    return CAF_OK;
}

int32_t caf_xcorr_perdelay_one_kernel(int32_t n) {
    return (perdelay_fused_ok(n) || perdelay_decimal_ok(n) || perdelay_mixed_ok(n) || perdelay_jit_ok(n)) ? 1 : 0;
}

int32_t caf_perdelay_jit_describe(int32_t n, const char* arch, const char* dump_path, char* buf, int32_t len) {
    CAF_REQUIRE(buf && len > 0, "caf_perdelay_jit_describe: no buffer");
    std::string text;
    const int rc = perdelay_jit_describe(n, arch, dump_path, &text);
    std::snprintf(buf, (size_t)len, "%s", text.c_str());
    return rc;
}

int32_t caf_xcorr_perdelay(const float* d_cutout, int32_t n, const float* d_rx, int64_t rx_len, int64_t start,
                           int64_t step, int64_t num, int32_t zero_oor, float* d_qf2, int32_t* d_fidx, float* d_caf,
                           float* d_ccaf, int64_t batch_rows, void* stream) {
    CAF_REQUIRE(d_cutout && d_rx && n >= 1 && rx_len >= 1 && num >= 0 && step != 0, "caf_xcorr_perdelay: bad arguments");
    if (!zero_oor) {
        const int64_t last = start + (num - 1) * step;
        const int64_t lo = std::min(start, last), hi = std::max(start, last);
        CAF_REQUIRE(num == 0 || (lo >= 0 && hi + n <= rx_len), "caf_xcorr_perdelay: delay window leaves rx");
    }
    if (num == 0) return CAF_OK;
    hipStream_t st = (hipStream_t)stream;
    // CAF_JIT_ALL=1 (measurements): every power of two and of ten through the run-time-compiled kernel as well.  By default those
    // where it measured faster than the prebuilt kernels, from the row count at which its per-call work (energy prefix, cutout
    // norm: ~25 us) is paid back (profiles/r05/timing_pow2_pow10_through_jit.log, ms per 1e5 delays): 10000 (2.0 against the
    // radix-10 kernel's 4.0), 16384 (3.1 / 3.8), 8192 (1.50 / 1.82), 2048 (0.29 / 0.32).
    const bool jit_all = [n, num] {
        const char* e = getenv("CAF_JIT_ALL");
        if (e) return atoi(e) != 0;
        return n == 10000 || ((n == 8192 || n == 16384) && num >= 8192) || (n == 2048 && num >= (1 << 17));
    }();
    // Power-of-two cutouts up to 16384 samples: one fused kernel (product -> LDS FFT -> |.|^2 -> argmax; window energies
    // and the cutout norm summed in the kernel): no product matrix, no prefix pass, no scratch, no synchronisation.
    // CAF_PERDELAY_UNFUSED=1 keeps the three-kernel form below (A/B switch; it also serves every other length).
    {
        static const bool unfused = [] {
            const char* e = getenv("CAF_PERDELAY_UNFUSED");
            return e && atoi(e) != 0;
        }();
        if (!unfused && perdelay_fused_ok(n) && !(jit_all && perdelay_jit_ok(n))) {
            const int rc1 = launch_perdelay_fused((const float2*)d_cutout, n, (const float2*)d_rx, rx_len, start, step, num,
                                                  zero_oor ? 1 : 0, d_qf2, (uint32_t*)d_fidx, d_caf, (float2*)d_ccaf, st);
            if (rc1) return rc1;
            CAF_HIP_TRY(hipGetLastError());
            return CAF_OK;
        }
    }
    Scratch sc;
    // energy prefix over the span of rx the delays touch (not the whole array: 128 delays of a 10^7-sample rx used
    // to cost a full pass), indices shifted accordingly; only when every window lies inside rx (otherwise the
    // out-of-range rules are stated against the whole array and the whole array is scanned)
    const int64_t s_last = start + (num - 1) * step;
    const int64_t span_lo = std::max<int64_t>(0, std::min(start, s_last));
    const int64_t span_hi = std::min<int64_t>(rx_len, std::max(start, s_last) + n);
    const bool clip = span_hi > span_lo && (span_lo > 0 || span_hi < rx_len) && span_lo == std::min(start, s_last) &&
                      span_hi == std::max(start, s_last) + n;  // only when no window leaves rx
    const float2* yv = (const float2*)d_rx + (clip ? span_lo : 0);
    const int64_t ylen_v = clip ? span_hi - span_lo : rx_len;
    const int64_t start_v = clip ? start - span_lo : start;
    double* prefix = nullptr;
    int rc = energy_prefix(yv, ylen_v, sc, &prefix, st);
    if (rc) return rc;
    // ||cutout|| in float64 on the device (no host round trip)
    double* d_cnorm = nullptr;
    if ((rc = sc.get(&d_cnorm, cutout_norm_scratch_doubles()))) return rc;
    const double* d_norm = launch_cutout_norm((const float2*)d_cutout, n, d_cnorm, st);
    // cutouts of 100 / 1000 / 10000 samples: one fused kernel with radix-10 passes in LDS (CAF_PERDELAY_UNFUSED=1: the chain below)
    {
        static const bool unfused10 = [] {
            const char* e = getenv("CAF_PERDELAY_UNFUSED");
            return e && atoi(e) != 0;
        }();
        // lengths with prime factors up to 23 that are neither a power of two nor of ten: a kernel compiled for the length at run
        // time (caf_jit.hip); a length / box without one, or a compilation that fails (reported once), keeps the plan-driven
        // kernel (7-smooth lengths) or the three-kernel form below
        if (!unfused10 && (jit_all || !(perdelay_fused_ok(n) || perdelay_decimal_ok(n))) && perdelay_jit_ok(n)) {
            rc = launch_perdelay_jit((const float2*)d_cutout, n, yv, ylen_v, prefix, d_norm, start_v, step, num, zero_oor ? 1 : 0, d_qf2,
                                     (uint32_t*)d_fidx, d_caf, (float2*)d_ccaf, st);
            if (rc == CAF_OK) {
                CAF_HIP_TRY(hipStreamSynchronize(st));  // scratch (prefix, norm) is freed on return
                CAF_HIP_TRY(hipGetLastError());
                return CAF_OK;
            }
            perdelay_jit_failed(n);
            static bool said = false;
            if (!said) {
                said = true;
                char msg[2048];
                caf_last_error(msg, sizeof(msg));
                std::fprintf(stderr, "[caf] run-time compilation unavailable, using the prebuilt per-delay kernels: %s\n", msg);
            }
        }
        if (!unfused10 && (perdelay_decimal_ok(n) || perdelay_mixed_ok(n))) {
            // (the radix-10 kernel of caf_perdelay.hip / the plan-driven mixed-radix kernel of caf_perdelay_mr.hip)
            rc = perdelay_decimal_ok(n)
                     ? launch_perdelay_decimal((const float2*)d_cutout, n, yv, ylen_v, prefix, d_norm, start_v, step, num,
                                               zero_oor ? 1 : 0, d_qf2, (uint32_t*)d_fidx, d_caf, (float2*)d_ccaf, st)
                     : launch_perdelay_mixed((const float2*)d_cutout, n, yv, ylen_v, prefix, d_norm, start_v, step, num,
                                             zero_oor ? 1 : 0, d_qf2, (uint32_t*)d_fidx, d_caf, (float2*)d_ccaf, st);
            if (rc) return rc;
            CAF_HIP_TRY(hipStreamSynchronize(st));  // scratch (prefix, norm) is freed on return
            CAF_HIP_TRY(hipGetLastError());
            return CAF_OK;
        }
    }
    // rows per batch: up to 2^28 product elements (2 GiB of the 288) in flight, so that even 1e7-sample cutouts go
    // through rocFFT and the argmax a few dozen rows at a time
    if (batch_rows <= 0) batch_rows = std::max<int64_t>(1, std::min<int64_t>(num, ((int64_t)1 << 28) / n));
    batch_rows = std::min(batch_rows, num);
    float2* rows = nullptr;
    float2* direct = (float2*)d_ccaf;  // when the complex plane is wanted, build it in place
    if (!direct && (rc = sc.get(&rows, batch_rows * n))) return rc;
    unsigned long long* part = nullptr;  // long rows: chunked argmax
    if (const int ch = rows_argmax_chunks(batch_rows, n))
        if ((rc = sc.get(&part, batch_rows * ch))) return rc;
    for (int64_t r0 = 0; r0 < num; r0 += batch_rows) {
        const int64_t nr = std::min(batch_rows, num - r0);
        float2* buf = direct ? direct + r0 * n : rows;
        launch_sliding_multiply((const float2*)d_cutout, n, yv, ylen_v, prefix, start_v + r0 * step, step, nr, 1.0,
                                zero_oor ? 1 : 0, buf, st, d_norm);
        if ((rc = fft_rows(buf, buf, nr, n, false, st))) return rc;
        if (d_qf2 || d_fidx || d_caf)
            launch_rows_argmax(buf, nr, n, 1, 1.0f, (uint32_t*)(d_fidx ? d_fidx + r0 : nullptr), d_qf2 ? d_qf2 + r0 : nullptr,
                               d_caf ? d_caf + r0 * n : nullptr, st, part, 1);
    }
    CAF_HIP_TRY(hipStreamSynchronize(st));  // scratch is freed on return
    CAF_HIP_TRY(hipGetLastError());
    return CAF_OK;
}

int32_t caf_sliding_multiply_normalised(const float* d_x, int32_t xlen, const float* d_y, int64_t ylen,
                                        int64_t start_idx, int64_t idxlen, double coefficient, float* d_z,
                                        void* stream) {
    CAF_REQUIRE(d_x && d_y && d_z && xlen >= 1 && ylen >= 1, "caf_sliding_multiply_normalised: bad arguments");
    CAF_REQUIRE(start_idx >= 0 && idxlen >= 0 && start_idx + idxlen <= ylen,
                "startIdx and idxlen should be within the bounds of d_y.");
    if (idxlen == 0) return CAF_OK;
    hipStream_t st = (hipStream_t)stream;
    Scratch sc;
    double* prefix = nullptr;
    int rc = energy_prefix((const float2*)d_y, ylen, sc, &prefix, st);
    if (rc) return rc;
    launch_sliding_multiply((const float2*)d_x, xlen, (const float2*)d_y, ylen, prefix, start_idx, 1, idxlen, coefficient,
                            0, (float2*)d_z, st);
    CAF_HIP_TRY(hipStreamSynchronize(st));
    CAF_HIP_TRY(hipGetLastError());
    return CAF_OK;
}

int32_t caf_multi_template_sliding_dot(const float* d_templates, const float* d_energies, int32_t num_templates,
                                       int32_t template_len, const float* d_x, int64_t xlen, int64_t start_idx,
                                       int64_t idxlen, int32_t* d_template_idx, float* d_qf2, void* stream) {
    CAF_REQUIRE(d_templates && d_energies && d_x && d_template_idx && d_qf2, "caf_multi_template_sliding_dot: NULL");
    CAF_REQUIRE(num_templates >= 1 && template_len >= 1, "need >= 1 template");
    CAF_REQUIRE(template_len <= 8192, "template too long for the LDS-resident kernel (use the hypothesis engine)");
    CAF_REQUIRE(start_idx >= 0 && idxlen >= 0 && start_idx + idxlen - 1 + template_len - 1 < xlen,
                "final slide index should be within the bounds of d_x");
    if (idxlen == 0) return CAF_OK;
    hipStream_t st = (hipStream_t)stream;
    Scratch sc;
    double* prefix = nullptr;
    int rc = energy_prefix((const float2*)d_x, xlen, sc, &prefix, st);
    if (rc) return rc;
    launch_multi_template_dot((const float2*)d_templates, d_energies, num_templates, template_len, (const float2*)d_x,
                              xlen, prefix, start_idx, idxlen, d_template_idx, d_qf2, st);
    CAF_HIP_TRY(hipStreamSynchronize(st));
    CAF_HIP_TRY(hipGetLastError());
    return CAF_OK;
}

int32_t caf_multiply_slices_indexed_rows(const float* d_x, int64_t xlen, const float* d_rows, int32_t num_rows,
                                         int32_t row_len, const int32_t* d_slice_starts, const int32_t* d_slice_lens,
                                         const int32_t* d_row_idx, int32_t out_len, int64_t num_slices, float* d_out,
                                         void* stream) {
    CAF_REQUIRE(d_x && d_rows && d_slice_starts && d_row_idx && d_out, "caf_multiply_slices_indexed_rows: NULL");
    CAF_REQUIRE(out_len >= 1 && num_rows >= 1 && num_slices >= 0, "bad slice/row lengths");
    launch_multiply_indexed_rows((const float2*)d_x, xlen, (const float2*)d_rows, row_len, d_slice_starts, d_slice_lens,
                                 d_row_idx, out_len, num_slices, (float2*)d_out, (hipStream_t)stream);
    CAF_HIP_TRY(hipGetLastError());
    return CAF_OK;
}

int32_t caf_complex_magnsq(const void* d_x, int64_t n, int32_t in_c128, void* d_out, int32_t out_f64, void* stream) {
    CAF_REQUIRE(d_x && d_out && n >= 0, "caf_complex_magnsq: bad arguments");
    CAF_REQUIRE(!(in_c128 && !out_f64), "complex128 input needs float64 output");
    if (n) launch_magnsq(d_x, n, in_c128, d_out, out_f64, (hipStream_t)stream);
    CAF_HIP_TRY(hipGetLastError());
    return CAF_OK;
}

int32_t caf_argmax_abs_rows(const float* d_x, int64_t rows, int64_t len, uint32_t* d_argmax, float* d_max,
                            int32_t use_normsq, void* stream) {
    CAF_REQUIRE(d_x && d_argmax && rows >= 0 && len >= 1, "caf_argmax_abs_rows: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    Scratch sc;
    unsigned long long* part = nullptr;
    const int ch = rows > 0 ? rows_argmax_chunks(rows, len) : 0;
    if (ch) {
        const int rc = sc.get(&part, rows * ch);
        if (rc) return rc;
    }
    for (int64_t r0 = 0; r0 < rows; r0 += ((int64_t)1 << 30))
        launch_rows_argmax((const float2*)d_x + r0 * len, std::min<int64_t>(rows - r0, (int64_t)1 << 30), len, use_normsq,
                           1.0f, d_argmax + r0, d_max ? d_max + r0 : nullptr, nullptr, st, part ? part + r0 * ch : nullptr);
    if (ch && st != nullptr) CAF_HIP_TRY(hipStreamSynchronize(st));  // (scratch: stream-ordered on the default stream)
    CAF_HIP_TRY(hipGetLastError());
    return CAF_OK;
}

int32_t caf_moving_average(const float* d_x, int64_t rows, int64_t n, int32_t avg_length, int32_t sum_instead,
                           float* d_out, void* stream) {
    CAF_REQUIRE(d_x && d_out && rows >= 1 && n >= 1 && avg_length >= 1, "caf_moving_average: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (avg_length <= moving_tile_max_window() && rows <= 65535) {  // one launch, no scratch, asynchronous
        launch_moving_tile(d_x, rows, n, avg_length, sum_instead, d_out, st);
        CAF_HIP_TRY(hipGetLastError());
        return CAF_OK;
    }
    Scratch sc;
    double *tiles = nullptr, *prefix = nullptr;
    int rc = sc.get(&tiles, moving_num_tiles(n) + 1024);
    if (rc) return rc;
    if ((rc = sc.get(&prefix, n + 1))) return rc;
    for (int64_t r = 0; r < rows; ++r)
        launch_moving_average(d_x + r * n, n, avg_length, sum_instead, tiles, prefix, d_out + r * n, st);
    CAF_HIP_TRY(hipStreamSynchronize(st));
    CAF_HIP_TRY(hipGetLastError());
    return CAF_OK;
}

int32_t caf_complex_moving_sum(const float* d_x, int64_t n, int32_t sum_length, float* d_out, void* stream) {
    CAF_REQUIRE(d_x && d_out && sum_length >= 1 && n >= sum_length, "caf_complex_moving_sum: bad arguments");
    CAF_REQUIRE(sum_length <= 4096, "sum_length too long for the LDS-resident kernel");
    launch_complex_moving_sum((const float2*)d_x, n, sum_length, d_out, (hipStream_t)stream);
    CAF_HIP_TRY(hipGetLastError());
    return CAF_OK;
}

int32_t caf_copy_slices_to_matrix(const float* d_x, int64_t xlen, const int32_t* d_starts, int32_t starts_stride,
                                  int64_t start0, int64_t increment, int32_t len, int64_t rows, float* d_out,
                                  void* stream) {
    CAF_REQUIRE(d_x && d_out && len >= 1 && rows >= 0, "caf_copy_slices_to_matrix: bad arguments");
    CAF_REQUIRE(!d_starts || starts_stride == 1 || starts_stride == 2, "starts_stride must be 1 or 2");
    if (rows)
        launch_copy_slices((const float2*)d_x, xlen, d_starts, starts_stride, start0, increment, len, rows,
                           (float2*)d_out, (hipStream_t)stream);
    CAF_HIP_TRY(hipGetLastError());
    return CAF_OK;
}

int32_t caf_copy_groups(const float* d_x, float* d_y, const int32_t* d_x_starts, const int32_t* d_y_starts,
                        const int32_t* d_lengths, int32_t num_groups, void* stream) {
    CAF_REQUIRE(d_x && d_y && d_x_starts && d_y_starts && d_lengths && num_groups >= 0, "caf_copy_groups: bad arguments");
    launch_copy_groups((const float2*)d_x, (float2*)d_y, d_x_starts, d_y_starts, d_lengths, num_groups,
                       (hipStream_t)stream);
    CAF_HIP_TRY(hipGetLastError());
    return CAF_OK;
}

int32_t caf_find_local_maxima(const float* d_x, int64_t n, float min_height, int32_t max_peaks, int32_t* d_peak_index,
                              int32_t* d_count, void* stream) {
    CAF_REQUIRE(d_x && d_peak_index && d_count && n >= 1 && max_peaks >= 1, "caf_find_local_maxima: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    Scratch sc;
    int32_t* tiles = nullptr;
    int rc = sc.get(&tiles, local_maxima_scratch_ints(n));
    if (rc) return rc;
    launch_find_local_maxima(d_x, n, min_height, tiles, max_peaks, d_peak_index, d_count, st);
    CAF_HIP_TRY(hipStreamSynchronize(st));
    CAF_HIP_TRY(hipGetLastError());
    return CAF_OK;
}

int32_t caf_gather_b32(const void* d_x, int64_t xlen, const int32_t* d_index, int64_t n, void* d_out, void* stream) {
    CAF_REQUIRE(d_x && d_index && d_out && xlen >= 1 && n >= 0, "caf_gather_b32: bad arguments");
    launch_gather_b32(d_x, xlen, d_index, n, d_out, (hipStream_t)stream);
    CAF_HIP_TRY(hipGetLastError());
    return CAF_OK;
}

int32_t caf_gather_f32_f64(const float* d_x, int64_t xlen, const int32_t* d_index, int64_t n, double* d_out, void* stream) {
    CAF_REQUIRE(d_x && d_out && xlen >= 1 && n >= 0, "caf_gather_f32_f64: bad arguments");
    launch_gather_f32_f64(d_x, xlen, d_index, n, d_out, (hipStream_t)stream);
    CAF_HIP_TRY(hipGetLastError());
    return CAF_OK;
}

int32_t caf_fir_lfilter(const float* d_x, int64_t n, const float* d_taps, int32_t num_taps, const float* d_delay,
                        int32_t delay_len, int32_t dsr, int32_t ds_phase, float* d_out, int64_t out_len, void* stream) {
    CAF_REQUIRE(d_x && d_taps && d_out && n >= 1 && num_taps >= 1, "caf_fir_lfilter: bad arguments");
    CAF_REQUIRE(dsr >= 1 && ds_phase >= 0 && ds_phase < dsr, "dsPhase must be between in the range [0,dsr-1].");
    CAF_REQUIRE(delay_len >= 0 && (delay_len == 0 || d_delay), "delay_len > 0 needs d_delay");
    CAF_REQUIRE(out_len >= 0 && out_len <= (n - ds_phase + dsr - 1) / dsr, "caf_fir_lfilter: out_len exceeds len(x[dsPhase::dsr])");
    if (fir_use_overlap_save(num_taps, dsr, 4096))  // long tap sets: frequency-domain blocks (any length)
        return fir_overlap_save(d_x, n, false, 1.0f, d_taps, num_taps, d_delay, delay_len, dsr, ds_phase, (float2*)d_out, out_len,
                                (hipStream_t)stream);
    launch_fir((const float2*)d_x, n, d_taps, num_taps, (const float2*)d_delay, delay_len, dsr, ds_phase, (float2*)d_out,
               out_len, (hipStream_t)stream);
    CAF_HIP_TRY(hipGetLastError());
    return CAF_OK;
}

int32_t caf_iq16_fir_decimate(const int16_t* d_iq, int64_t num_samples, float scale, const float* d_taps, int32_t num_taps,
                              const int16_t* d_delay, int32_t delay_len, int32_t dsr, int32_t ds_phase, float* d_out,
                              int64_t out_len, void* stream) {
    CAF_REQUIRE(d_iq && d_taps && d_out && num_samples >= 1 && num_taps >= 1, "caf_iq16_fir_decimate: bad arguments");
    CAF_REQUIRE(dsr >= 1 && ds_phase >= 0 && ds_phase < dsr, "dsPhase must be between in the range [0,dsr-1].");
    CAF_REQUIRE(delay_len >= 0 && (delay_len == 0 || d_delay), "delay_len > 0 needs d_delay");
    CAF_REQUIRE(((uintptr_t)d_iq & 3) == 0 && ((uintptr_t)d_delay & 3) == 0, "caf_iq16_fir_decimate: IQ pairs must be 4-byte aligned");
    CAF_REQUIRE(out_len >= 0 && out_len <= (num_samples - ds_phase + dsr - 1) / dsr,
                "caf_iq16_fir_decimate: out_len exceeds len(x[dsPhase::dsr])");
    // direct polyphase form: <= 2048 taps and dsr <= 16; anything else (and long tap sets) goes overlap-save
    if (!fir_decim_ok(num_taps, dsr) || fir_use_overlap_save(num_taps, dsr, 2048))
        return fir_overlap_save(d_iq, num_samples, true, scale, d_taps, num_taps, d_delay, delay_len, dsr, ds_phase,
                                (float2*)d_out, out_len, (hipStream_t)stream);
    launch_iq16_fir(d_iq, num_samples, scale, d_taps, num_taps, d_delay, delay_len, dsr, ds_phase, (float2*)d_out, out_len,
                    (hipStream_t)stream);
    CAF_HIP_TRY(hipGetLastError());
    return CAF_OK;
}

int32_t caf_upfirdn(const float* d_x, int64_t rows, int64_t n, const float* d_taps, int32_t num_taps, int32_t up,
                    int32_t down, float* d_out, float* d_out_abs, int64_t out_len, void* stream) {
    CAF_REQUIRE(d_x && d_taps && (d_out || d_out_abs) && rows >= 1 && rows <= 65535 && n >= 1, "caf_upfirdn: bad arguments");
    CAF_REQUIRE(num_taps >= 1 && num_taps <= 16384 && up >= 1 && down >= 1, "caf_upfirdn: bad taps/up/down");
    const int64_t full = ((n - 1) * up + num_taps + down - 1) / down;
    CAF_REQUIRE(out_len >= 1 && out_len <= full, "caf_upfirdn: out_len larger than the full upfirdn length");
    // up == 1 is a FIR with decimation: full-convolution outputs [0 :: down] (zeros beyond the input).  From 96 taps per unit of
    // decimation on, the overlap-save form (caf_firos.hip: the rows are blockIdx.y of ONE launch) -- 64 x 262144 samples, 128
    // taps, up = down = 1: 0.38 ms through the polyphase kernel (0.09 of the HBM bound), the same job as caf_fir_lfilter otherwise
    if (up == 1 && d_out && !d_out_abs && fir_os_fused_block(num_taps) && fir_use_overlap_save(num_taps, down, 1 << 30)) {
        hipStream_t st = (hipStream_t)stream;
        Scratch sc;
        float2* ht = nullptr;
        int rc = sc.get(&ht, fir_os_fused_block(num_taps));
        if (rc) return rc;
        rc = launch_fir_os_fused((const float2*)d_x, n, d_taps, num_taps, nullptr, 0, down, 0, (float2*)d_out, out_len, ht, st, rows, n,
                                 out_len);
        if (rc) return rc;
        if (st != nullptr) CAF_HIP_TRY(hipStreamSynchronize(st));  // (scratch: stream-ordered on the default stream, see fir_overlap_save)
        CAF_HIP_TRY(hipGetLastError());
        return CAF_OK;
    }
    launch_upfirdn((const float2*)d_x, rows, n, d_taps, num_taps, up, down, out_len, (float2*)d_out, d_out_abs,
                   (hipStream_t)stream);
    CAF_HIP_TRY(hipGetLastError());
    return CAF_OK;
}

int32_t caf_czt_run_many(const float* d_x, int64_t rows, int32_t m, int32_t k, int32_t nfft, const float* d_aa,
                         const float* d_fv, const float* d_ww, float* d_out, void* stream) {
    CAF_REQUIRE(d_x && d_aa && d_fv && d_ww && d_out, "caf_czt_run_many: NULL");
    CAF_REQUIRE(rows >= 0 && m >= 1 && k >= 1 && nfft >= m + k - 1, "caf_czt_run_many: need nfft >= m + k - 1");
    if (rows == 0) return CAF_OK;
    hipStream_t st = (hipStream_t)stream;
    Scratch sc;
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(rows, ((int64_t)1 << 25) / nfft));
    float2* buf = nullptr;
    int rc = sc.get(&buf, chunk * nfft);
    if (rc) return rc;
    for (int64_t r0 = 0; r0 < rows; r0 += chunk) {
        const int64_t nr = std::min(chunk, rows - r0);
        // y = x * aa, zero-padded to nfft
        launch_rows_mul_vec((const float2*)d_x + r0 * m, m, 0, (const float2*)d_aa, m, buf, nfft, nfft, nr, 1.0f, st);
        if ((rc = fft_rows(buf, buf, nr, nfft, false, st))) return rc;
        launch_rows_mul_vec(buf, nfft, 0, (const float2*)d_fv, nfft, buf, nfft, nfft, nr, 1.0f, st);
        if ((rc = fft_rows(buf, buf, nr, nfft, true, st))) return rc;
        // g[m-1 : m+k-1] * ww / nfft
        launch_rows_mul_vec(buf, nfft, m - 1, (const float2*)d_ww, k, (float2*)d_out + r0 * k, k, k, nr,
                            1.0f / (float)nfft, st);
    }
    CAF_HIP_TRY(hipStreamSynchronize(st));
    CAF_HIP_TRY(hipGetLastError());
    return CAF_OK;
}

int32_t caf_argmax3d_u32(const uint32_t* d_x, int64_t num_items, int32_t dim1, int32_t dim2, int32_t dim3,
                         uint32_t* d_argmax, uint32_t* d_max, void* stream) {
    CAF_REQUIRE(d_x && d_argmax && num_items >= 0 && dim1 >= 1 && dim2 >= 1 && dim3 >= 1, "caf_argmax3d_u32: bad arguments");
    CAF_REQUIRE((int64_t)dim1 * dim2 * dim3 < ((int64_t)1 << 32) && num_items < ((int64_t)1 << 31),
                "caf_argmax3d_u32: item too large");
    launch_argmax3d_u32(d_x, num_items, dim1, dim2, dim3, d_argmax, d_max, (hipStream_t)stream);
    CAF_HIP_TRY(hipGetLastError());
    return CAF_OK;
}

int32_t caf_iq16_to_c64(const int16_t* d_iq, int64_t num_samples, float scale, float* d_out, void* stream) {
    CAF_REQUIRE(d_iq && d_out && num_samples >= 0, "caf_iq16_to_c64: bad arguments");
    CAF_REQUIRE((reinterpret_cast<uintptr_t>(d_iq) & 7) == 0 && (reinterpret_cast<uintptr_t>(d_out) & 15) == 0,
                "caf_iq16_to_c64: d_iq must be 8-byte and d_out 16-byte aligned");
    launch_iq16_to_c64(d_iq, num_samples, scale, (float2*)d_out, (hipStream_t)stream);
    CAF_HIP_TRY(hipGetLastError());
    return CAF_OK;
}

int32_t caf_colmax_abs(const float* d_z, int32_t rows, int64_t n, float* d_max, void* d_arg, int32_t arg_int64,
                       void* stream) {
    CAF_REQUIRE(d_z && d_max && d_arg && rows >= 1 && n >= 1, "caf_colmax_abs: bad arguments");
    launch_colmax_abs((const float2*)d_z, rows, n, d_max, d_arg, arg_int64, (hipStream_t)stream);
    CAF_HIP_TRY(hipGetLastError());
    return CAF_OK;
}

int32_t caf_colmax_sqrt(const float* d_q2, int32_t rows, int64_t n, float* d_max, int64_t* d_arg, void* stream) {
    CAF_REQUIRE(d_q2 && d_max && d_arg && rows >= 1 && n >= 1, "caf_colmax_sqrt: bad arguments");
    launch_colmax_sqrt(d_q2, rows, n, d_max, d_arg, (hipStream_t)stream);
    CAF_HIP_TRY(hipGetLastError());
    return CAF_OK;
}

int32_t caf_dot_tones(const float* d_src, int64_t len, double f0, double fstep, int32_t num_freqs, float* d_out,
                      void* stream) {
    CAF_REQUIRE(d_src && d_out && len >= 1 && num_freqs >= 1, "caf_dot_tones: bad arguments");
    CAF_REQUIRE((len + 63) / 64 < ((int64_t)1 << 31), "caf_dot_tones: source too long");
    launch_dot_tones(f0, fstep, num_freqs, len, (const float2*)d_src, (float2*)d_out, (hipStream_t)stream);
    CAF_HIP_TRY(hipGetLastError());
    return CAF_OK;
}

int32_t caf_mul_conj(const float* d_a, const float* d_b, int64_t n, float* d_out, void* stream) {
    CAF_REQUIRE(d_a && d_b && d_out && n >= 0, "caf_mul_conj: bad arguments");
    if (n) launch_mul_conj((const float2*)d_a, (const float2*)d_b, n, (float2*)d_out, (hipStream_t)stream);
    CAF_HIP_TRY(hipGetLastError());
    return CAF_OK;
}

int32_t caf_steer_dot(const float* d_vec, const double* d_steer, int64_t rows, int64_t n, double scale, double* d_out,
                      void* stream) {
    CAF_REQUIRE(d_vec && d_steer && d_out && rows >= 1 && n >= 1, "caf_steer_dot: bad arguments");
    CAF_REQUIRE(rows < ((int64_t)1 << 31), "caf_steer_dot: too many rows");
    launch_steer_dot((const float2*)d_vec, (const double2*)d_steer, rows, n, scale, (double2*)d_out, (hipStream_t)stream);
    CAF_HIP_TRY(hipGetLastError());
    return CAF_OK;
}

int32_t caf_sum_planes_qf2(const float* d_planes, int32_t num_planes, int64_t rows, int32_t cols, const int32_t* h_sel,
                           int32_t num_sel, const double* d_row_norm, double ynormsq, double* d_out, void* stream) {
    CAF_REQUIRE(d_planes && h_sel && d_row_norm && d_out && num_planes >= 1 && rows >= 1 && cols >= 1,
                "caf_sum_planes_qf2: bad arguments");
    CAF_REQUIRE(num_sel >= 1 && num_sel <= 64, "caf_sum_planes_qf2: between 1 and 64 planes can be summed");
    for (int j = 0; j < num_sel; ++j)
        CAF_REQUIRE(h_sel[j] >= 0 && h_sel[j] < num_planes, "caf_sum_planes_qf2: plane number out of range");
    CAF_REQUIRE(ynormsq > 0.0, "caf_sum_planes_qf2: ynormsq must be positive");
    launch_sum_planes_qf2((const float2*)d_planes, rows * cols, cols, h_sel, num_sel, d_row_norm, ynormsq, d_out,
                          (hipStream_t)stream);
    CAF_HIP_TRY(hipGetLastError());
    return CAF_OK;
}

int32_t caf_sum_groups_qf2(const float* d_planes, int32_t num_groups, int64_t rows, int32_t cols, const float* d_phase,
                           const double* d_row_norm, double ynormsq, double* d_out, void* stream) {
    CAF_REQUIRE(d_planes && d_row_norm && d_out && num_groups >= 1 && rows >= 1 && cols >= 1, "caf_sum_groups_qf2: bad arguments");
    CAF_REQUIRE(ynormsq > 0.0, "caf_sum_groups_qf2: ynormsq must be positive");
    launch_sum_groups_qf2((const float2*)d_planes, num_groups, rows * cols, cols, (const float2*)d_phase, d_row_norm, ynormsq, d_out,
                          (hipStream_t)stream);
    CAF_HIP_TRY(hipGetLastError());
    return CAF_OK;
}

int32_t caf_zoom_num_bins(double span, double step, int32_t* num_bins) {
    CAF_REQUIRE(num_bins && span > 0.0 && step > 0.0 && span / step < 1e6, "caf_zoom_num_bins: bad span/step");
    *num_bins = zoom_num_bins(span, step);
    return CAF_OK;
}

int32_t caf_zoom_czt(caf_plan plan, int32_t template_index, const float* d_rx, int64_t rx_len, const float* d_row_max,
                     const int32_t* d_row_arg, int64_t shift_start, int64_t num_shifts, int32_t k, float min_height,
                     double span, double step, const caf_zoom_outputs* out, void* stream) {
    CAF_REQUIRE(plan && d_rx && d_row_max && d_row_arg && out, "caf_zoom_czt: NULL argument");
    PlanZoomView v;
    int rc = plan_zoom_view(plan, &v);
    if (rc) return rc;
    CAF_REQUIRE(template_index >= 0 && template_index < v.T, "caf_zoom_czt: template_index outside the plan");
    CAF_REQUIRE(v.G == 1, "caf_zoom_czt: composite (multi-group) templates are not supported");
    CAF_REQUIRE(k >= 1 && k <= 4096, "caf_zoom_czt: need 1 <= k <= 4096");
    CAF_REQUIRE(span > 0.0 && step > 0.0 && span / step < 1e5, "caf_zoom_czt: bad span/step");
    CAF_REQUIRE(shift_start >= 0 && num_shifts >= 1 && shift_start + num_shifts - 1 + v.N <= rx_len,
                "caf_zoom_czt: delays run past the end of rx");
    CAF_REQUIRE(num_shifts < ((int64_t)1 << 31), "caf_zoom_czt: trace too long");
    int cur = -1;
    CAF_HIP_TRY(hipGetDevice(&cur));
    CAF_REQUIRE(cur == v.device, "caf_zoom_czt: the plan was created on another device than the current one");
    hipStream_t st = (hipStream_t)stream;
    ZoomCzt z;
    if ((rc = zoom_constants(v.device, v.N, span, step, &z))) return rc;
    Scratch sc;
    const int32_t max_cand = (int32_t)std::min<int64_t>(num_shifts, (int64_t)1 << 20);
    int32_t *tiles = nullptr, *cand = nullptr, *cnt = nullptr, *sel = nullptr, *selcnt = nullptr;
    float *vals = nullptr, *fmax = nullptr;
    uint32_t* farg = nullptr;
    float2 *rows = nullptr, *zk = nullptr;
    if ((rc = sc.get(&tiles, local_maxima_scratch_ints(num_shifts))) || (rc = sc.get(&cand, max_cand)) ||
        (rc = sc.get(&cnt, 1)) || (rc = sc.get(&sel, k)) || (rc = sc.get(&selcnt, 1)) || (rc = sc.get(&vals, max_cand)) ||
        (rc = sc.get(&fmax, k)) || (rc = sc.get(&farg, k)) || (rc = sc.get(&rows, (int64_t)k * z.nfft)) ||
        (rc = sc.get(&zk, (int64_t)k * z.k)))
        return rc;
    // 1. local maxima above min_height (peakfinding.cu:52 predicate), ascending index order, count on the device
    launch_find_local_maxima(d_row_max, num_shifts, min_height, tiles, max_cand, cand, cnt, st);
    // 2. the k strongest: value descending, index ascending
    launch_zoom_topk(d_row_max, cand, cnt, max_cand, k, vals, sel, selcnt, st);
    // 3. all product rows in one launch, already rotated, pre-chirped and padded
    launch_zoom_rows((const float2*)d_rx, v.d_uconj + (int64_t)template_index * v.N, v.N, v.d_tscale + template_index, d_row_arg,
                     v.d_nu, z.aa, sel, selcnt, k, shift_start, z.nfft, rows, st);
    // 4. one batched Bluestein transform
    if ((rc = fft_rows(rows, rows, k, z.nfft, false, st))) return rc;
    launch_rows_mul_vec(rows, z.nfft, 0, z.fv, z.nfft, rows, z.nfft, z.nfft, k, 1.0f, st);
    if ((rc = fft_rows(rows, rows, k, z.nfft, true, st))) return rc;
    launch_rows_mul_vec(rows, z.nfft, v.N - 1, z.wws, z.k, zk, z.k, z.k, k, 1.0f / (float)z.nfft, st);
    // 5. |.|^2, fine argmax (first index), optional planes
    launch_rows_argmax(zk, k, z.k, 1, 1.0f, farg, fmax, out->d_planes, st);
    launch_zoom_finish(d_row_max, d_row_arg, v.d_nu, sel, selcnt, k, shift_start, span, step, farg, fmax, max_cand, cnt,
                       out->d_count, out->d_delay, out->d_coarse_freq_index, out->d_coarse_qf2, out->d_fine_index,
                       out->d_fine_freq, out->d_fine_qf2, st);
    if (st != nullptr) CAF_HIP_TRY(hipStreamSynchronize(st));  // scratch: see fir_overlap_save
    CAF_HIP_TRY(hipGetLastError());
    return CAF_OK;
}

}  // extern "C"
