// Host <-> device transfers of pageable host memory through the library's OWN pinned staging buffers.
//
// Why the library does not hand a caller's pointer to hipMemcpy: from 1 MB up (GPU_PINNED_MIN_XFER_SIZE) the HIP runtime
// does not stage a pageable copy, it PINS the caller's pages ("Locking to pool ... HSA Copy Using Pinned resource" in
// AMD_LOG_LEVEL=4) and lets the DMA engine write into them.  The pages belong to a NumPy array; when NumPy hands that
// memory back to the operating system (munmap of an array above malloc's mmap threshold, a heap trim) while the range is
// still registered with the driver, the MMU notifier evicts the process's GPU queues, and the driver restores them a fixed
// 100 ms later: the NEXT submission of the process -- typically the 1 MB upload at the start of the next call -- waits for
// that.  Round 4 saw it as "cztXcorr's per-delay form settles at exactly 100.0 ms per call" and blamed long rocFFT rows;
// what those calls had in common was a 3.2 MB result array (profiles/r05/stall_variants.log: the same process with the
// host arrays kept on malloc's heap, or with the runtime's pin threshold raised, runs 1.8 - 2.5 ms per call).
// A staging copy never registers user memory, so nothing the caller does with its arrays afterwards reaches the driver.
//
// Layout: up to 8 transfer lanes per device, each a non-blocking stream with two pinned 4 MB slots (64 MB of pinned memory
// at most, allocated on first use).  A transfer is cut into contiguous slices, one lane + one host thread per slice for
// large transfers (a single core moves ~10 GB/s between staging and user memory, the link ~55 GB/s); inside a lane the
// DMA of chunk i overlaps the host copy of chunk i - 1.  Transfers are ordered behind the work already queued on the
// caller's stream (an event) and are complete when the call returns, like the hipMemcpyAsync + hipStreamSynchronize pair
// they replace.  d2h_transposed additionally turns [rows][pitch] float32 device rows into a [cols][rows] host array of
// float32 or float64 on the way (8 x 8 register transposes): the hypothesis-major CAF surface leaves the device as the
// delay-major array the reference returns (xcorrRoutines.py:553-566, 1028-1039) without a transposition on the GPU and
// without a second pass on the host for the float64 the CPU signatures return.
#include <hip/hip_runtime_api.h>
#include <immintrin.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "caf.h"

namespace caf {
void set_error(const std::string& msg);

namespace {

constexpr size_t SLOT_BYTES = (size_t)4 << 20;
constexpr int MAX_LANES = 8;
constexpr int64_t DIRECT_BELOW = (int64_t)256 << 10;  // smaller copies: the runtime's own staged path (it pins from 1 MB)
constexpr int64_t BYTES_PER_EXTRA_LANE = (int64_t)8 << 20;

struct Lane {
    hipStream_t s = nullptr;
    char* slot[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
};
struct DevLanes {
    std::vector<Lane> lanes;
    hipEvent_t start = nullptr;
};

std::mutex g_mu;  // one transfer at a time (the lanes are shared)
std::map<int, DevLanes> g_dev;

int max_lanes() {
    static const int n = [] {
        if (const char* e = std::getenv("CAF_XFER_THREADS")) return std::max(1, std::min(MAX_LANES, std::atoi(e)));
        const unsigned hc = std::thread::hardware_concurrency();
        return (int)std::max(1u, std::min<unsigned>(MAX_LANES, hc / 2));
    }();
    return n;
}

#define XFER_TRY(expr)                                                      \
    do {                                                                    \
        hipError_t _e = (expr);                                             \
        if (_e != hipSuccess) {                                             \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e));   \
            return (_e == hipErrorOutOfMemory) ? CAF_ERR_NOMEM : CAF_ERR_HIP; \
        }                                                                   \
    } while (0)

// caller holds g_mu
int get_lanes(int dev, int want, DevLanes** out) {
    DevLanes& d = g_dev[dev];
    if (!d.start) XFER_TRY(hipEventCreateWithFlags(&d.start, hipEventDisableTiming));
    while ((int)d.lanes.size() < want) {
        Lane l;
        XFER_TRY(hipStreamCreateWithFlags(&l.s, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            XFER_TRY(hipHostMalloc((void**)&l.slot[i], SLOT_BYTES, hipHostMallocDefault));
            XFER_TRY(hipEventCreateWithFlags(&l.ev[i], hipEventDisableTiming));
        }
        d.lanes.push_back(l);
    }
    *out = &d;
    return CAF_OK;
}

// ---- 8 x 8 float transposes ----------------------------------------------------------------------------------------
// src: 8 rows of 8 floats, row pitch sp (floats); dst: 8 rows (the source's columns), row pitch dp (elements)
__attribute__((target("avx2"))) inline void tr8x8_load(const float* src, int64_t sp, __m256 r[8]) {
    __m256 a[8], b[8];
    for (int i = 0; i < 8; ++i) a[i] = _mm256_loadu_ps(src + i * sp);
    for (int i = 0; i < 4; ++i) {
        b[2 * i] = _mm256_unpacklo_ps(a[2 * i], a[2 * i + 1]);
        b[2 * i + 1] = _mm256_unpackhi_ps(a[2 * i], a[2 * i + 1]);
    }
    for (int i = 0; i < 2; ++i) {
        a[4 * i + 0] = _mm256_shuffle_ps(b[4 * i], b[4 * i + 2], 0x44);
        a[4 * i + 1] = _mm256_shuffle_ps(b[4 * i], b[4 * i + 2], 0xee);
        a[4 * i + 2] = _mm256_shuffle_ps(b[4 * i + 1], b[4 * i + 3], 0x44);
        a[4 * i + 3] = _mm256_shuffle_ps(b[4 * i + 1], b[4 * i + 3], 0xee);
    }
    for (int i = 0; i < 4; ++i) {
        r[i] = _mm256_permute2f128_ps(a[i], a[i + 4], 0x20);
        r[i + 4] = _mm256_permute2f128_ps(a[i], a[i + 4], 0x31);
    }
}
__attribute__((target("avx2"))) void tr_block_avx2(const float* src, int64_t sp, int64_t nr, int64_t nc, void* dst, int64_t dp,
                                                   bool f64) {
    // src [nr][sp] (nc columns used) -> dst [nc][dp] (nr entries used per row)
    const int64_t nr8 = nr & ~(int64_t)7, nc8 = nc & ~(int64_t)7;
    for (int64_t c = 0; c < nc8; c += 8) {
        for (int64_t r0 = 0; r0 < nr8; r0 += 8) {
            __m256 v[8];
            tr8x8_load(src + r0 * sp + c, sp, v);
            if (f64) {
                double* d = (double*)dst + c * dp + r0;
                for (int i = 0; i < 8; ++i) {
                    _mm256_storeu_pd(d + i * dp, _mm256_cvtps_pd(_mm256_castps256_ps128(v[i])));
                    _mm256_storeu_pd(d + i * dp + 4, _mm256_cvtps_pd(_mm256_extractf128_ps(v[i], 1)));
                }
            } else {
                float* d = (float*)dst + c * dp + r0;
                for (int i = 0; i < 8; ++i) _mm256_storeu_ps(d + i * dp, v[i]);
            }
        }
    }
    // edges
    for (int64_t c = 0; c < nc; ++c)
        for (int64_t r = (c < nc8 ? nr8 : 0); r < nr; ++r) {
            const float x = src[r * sp + c];
            if (f64)
                ((double*)dst)[c * dp + r] = (double)x;
            else
                ((float*)dst)[c * dp + r] = x;
        }
}
void tr_block_scalar(const float* src, int64_t sp, int64_t nr, int64_t nc, void* dst, int64_t dp, bool f64) {
    for (int64_t c0 = 0; c0 < nc; c0 += 16)
        for (int64_t r0 = 0; r0 < nr; r0 += 16)
            for (int64_t c = c0; c < std::min(nc, c0 + 16); ++c)
                for (int64_t r = r0; r < std::min(nr, r0 + 16); ++r) {
                    const float x = src[r * sp + c];
                    if (f64)
                        ((double*)dst)[c * dp + r] = (double)x;
                    else
                        ((float*)dst)[c * dp + r] = x;
                }
}
void tr_block(const float* src, int64_t sp, int64_t nr, int64_t nc, void* dst, int64_t dp, bool f64) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2)
        tr_block_avx2(src, sp, nr, nc, dst, dp, f64);
    else
        tr_block_scalar(src, sp, nr, nc, dst, dp, f64);
}

// float32 -> float64, n values
__attribute__((target("avx2"))) void widen_block_avx2(const float* src, double* dst, int64_t n) {
    int64_t i = 0;
    for (; i + 8 <= n; i += 8) {
        const __m256 v = _mm256_loadu_ps(src + i);
        _mm256_storeu_pd(dst + i, _mm256_cvtps_pd(_mm256_castps256_ps128(v)));
        _mm256_storeu_pd(dst + i + 4, _mm256_cvtps_pd(_mm256_extractf128_ps(v, 1)));
    }
    for (; i < n; ++i) dst[i] = (double)src[i];
}
void widen_block(const float* src, double* dst, int64_t n) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2)
        widen_block_avx2(src, dst, n);
    else
        for (int64_t i = 0; i < n; ++i) dst[i] = (double)src[i];
}

// ---- one lane's share of a transfer ----------------------------------------------------------------------------------
enum Kind { H2D, D2H, D2H_T, D2H_W };  // D2H_W: float32 device values arrive as float64 (lo / hi: byte range of the source)
struct Job {
    Kind kind;
    char* h;        // host base (D2H_T: the [cols][rows] array)
    const char* d;  // device base (H2D: destination)
    int64_t lo, hi;  // H2D / D2H: byte range; D2H_T: column range
    // D2H_T
    int64_t rows, pitch, col0;
    bool f64;
};

int run_lane(int dev, Lane& l, hipEvent_t start, const Job& j) {
    XFER_TRY(hipSetDevice(dev));
    XFER_TRY(hipStreamWaitEvent(l.s, start, 0));
    if (j.kind == H2D) {
        int64_t off = j.lo;
        for (int i = 0; off < j.hi; ++i) {
            const int k = i & 1;
            const int64_t n = std::min<int64_t>(SLOT_BYTES, j.hi - off);
            if (i >= 2) XFER_TRY(hipEventSynchronize(l.ev[k]));
            std::memcpy(l.slot[k], j.h + off, (size_t)n);
            XFER_TRY(hipMemcpyAsync((void*)(j.d + off), l.slot[k], (size_t)n, hipMemcpyHostToDevice, l.s));
            XFER_TRY(hipEventRecord(l.ev[k], l.s));
            off += n;
        }
        XFER_TRY(hipStreamSynchronize(l.s));
        return CAF_OK;
    }
    // D2H (plain or transposed): chunk i is in flight while chunk i - 1 is moved out of its slot
    const int64_t unit = j.kind != D2H_T ? (int64_t)SLOT_BYTES : std::max<int64_t>(16, (int64_t)(SLOT_BYTES / 4 / j.rows) & ~(int64_t)15);
    int64_t pend_off = 0, pend_n = 0;
    int pend_k = -1;
    auto drain = [&]() -> int {
        if (pend_k < 0) return CAF_OK;
        XFER_TRY(hipEventSynchronize(l.ev[pend_k]));
        if (j.kind == D2H)
            std::memcpy(j.h + pend_off, l.slot[pend_k], (size_t)pend_n);
        else if (j.kind == D2H_W)
            widen_block((const float*)l.slot[pend_k], (double*)(j.h + 2 * pend_off), pend_n / 4);
        else  // slot holds [rows][pend_n] -> host rows pend_off - col0 ...
            tr_block((const float*)l.slot[pend_k], pend_n, j.rows, pend_n,
                     j.h + (pend_off - j.col0) * j.rows * (j.f64 ? 8 : 4), j.rows, j.f64);
        pend_k = -1;
        return CAF_OK;
    };
    int64_t off = j.lo;
    for (int i = 0; off < j.hi; ++i) {
        const int k = i & 1;
        const int64_t n = std::min<int64_t>(unit, j.hi - off);
        if (j.kind != D2H_T)
            XFER_TRY(hipMemcpyAsync(l.slot[k], j.d + off, (size_t)n, hipMemcpyDeviceToHost, l.s));
        else
            XFER_TRY(hipMemcpy2DAsync(l.slot[k], (size_t)n * 4, j.d + off * 4, (size_t)j.pitch * 4, (size_t)n * 4, (size_t)j.rows,
                                      hipMemcpyDeviceToHost, l.s));
        XFER_TRY(hipEventRecord(l.ev[k], l.s));
        const int rc = drain();
        if (rc) return rc;
        pend_k = k, pend_off = off, pend_n = n;
        off += n;
    }
    return drain();
}

int transfer(Job j, int64_t weight_bytes, int64_t align, hipStream_t st) {
    int dev = 0;
    XFER_TRY(hipGetDevice(&dev));
    const int want = (int)std::max<int64_t>(1, std::min<int64_t>(max_lanes(), weight_bytes / BYTES_PER_EXTRA_LANE));
    std::lock_guard<std::mutex> lk(g_mu);
    DevLanes* dl = nullptr;
    int rc = get_lanes(dev, want, &dl);
    if (rc) return rc;
    XFER_TRY(hipEventRecord(dl->start, st));
    if (want == 1) return run_lane(dev, dl->lanes[0], dl->start, j);
    // contiguous slices, boundaries rounded to `align` units
    const int64_t total = j.hi - j.lo;
    int64_t per = (total + want - 1) / want;
    per = (per + align - 1) / align * align;
    std::vector<std::thread> th;
    std::vector<int> rcs(want, CAF_OK);
    std::vector<std::string> msgs(want);
    int used = 0;
    for (int w = 0; w < want; ++w) {
        Job s = j;
        s.lo = j.lo + w * per;
        s.hi = std::min(j.hi, s.lo + per);
        if (s.lo >= s.hi) break;
        ++used;
        if (w == 0) continue;  // the calling thread's own slice, below
        th.emplace_back([&, s, w] {
            rcs[w] = run_lane(dev, dl->lanes[w], dl->start, s);
            if (rcs[w]) {
                char buf[512];
                caf_last_error(buf, sizeof(buf));  // (thread-local message of the worker)
                msgs[w] = buf;
            }
        });
    }
    {
        Job s = j;
        s.hi = std::min(j.hi, j.lo + per);
        rcs[0] = run_lane(dev, dl->lanes[0], dl->start, s);
    }
    for (auto& t : th) t.join();
    for (int w = 0; w < used; ++w)
        if (rcs[w]) {
            if (w) set_error(msgs[w]);
            return rcs[w];
        }
    return CAF_OK;
}

}  // namespace

int host_h2d(void* d_dst, const void* h_src, int64_t bytes, hipStream_t st) {
    if (bytes <= 0) return CAF_OK;
    if (bytes < DIRECT_BELOW) {
        XFER_TRY(hipMemcpyAsync(d_dst, h_src, (size_t)bytes, hipMemcpyHostToDevice, st));
        XFER_TRY(hipStreamSynchronize(st));
        return CAF_OK;
    }
    Job j{H2D, (char*)const_cast<void*>(h_src), (const char*)d_dst, 0, bytes, 0, 0, 0, false};
    return transfer(j, bytes, 4096, st);
}

int host_d2h(void* h_dst, const void* d_src, int64_t bytes, hipStream_t st) {
    if (bytes <= 0) return CAF_OK;
    if (bytes < DIRECT_BELOW) {
        XFER_TRY(hipMemcpyAsync(h_dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost, st));
        XFER_TRY(hipStreamSynchronize(st));
        return CAF_OK;
    }
    Job j{D2H, (char*)h_dst, (const char*)d_src, 0, bytes, 0, 0, 0, false};
    return transfer(j, bytes, 4096, st);
}

int host_d2h_f64(double* h_dst, const float* d_src, int64_t count, hipStream_t st) {
    if (count <= 0) return CAF_OK;
    Job j{D2H_W, (char*)h_dst, (const char*)d_src, 0, count * 4, 0, 0, 0, true};
    return transfer(j, count * 8, 4096, st);
}

int host_d2h_transposed(void* h_dst, bool dst_f64, const float* d_src, int64_t rows, int64_t pitch, int64_t col0, int64_t ncols,
                        hipStream_t st) {
    if (rows <= 0 || ncols <= 0) return CAF_OK;
    Job j{D2H_T, (char*)h_dst, (const char*)d_src, col0, col0 + ncols, rows, pitch, col0, dst_f64};
    return transfer(j, rows * ncols * (dst_f64 ? 8 : 4), 16, st);
}

}  // namespace caf
