// rocFFT plumbing shared by the hypothesis-engine plan and the kernel-level ops:
// batched 1-D complex64 transforms (in place or out of place) with an explicit batch distance.
#include <rocfft/rocfft.h>

#include <map>
#include <mutex>
#include <tuple>

#include "caf_internal.h"

namespace caf {

static std::once_flag g_rocfft_once;

#define CAF_FFT_TRY(expr)                                                                 \
    do {                                                                                  \
        rocfft_status _s = (expr);                                                        \
        if (_s != rocfft_status_success) {                                                \
            caf::set_error(std::string(#expr) + ": rocfft status " + std::to_string(_s)); \
            return CAF_ERR_ROCFFT;                                                        \
        }                                                                                 \
    } while (0)

int FftPlan::create(bool inverse, size_t len, size_t batch, size_t dist, bool inplace) {
    std::call_once(g_rocfft_once, [] { rocfft_setup(); });
    rocfft_plan_description desc = nullptr;
    CAF_FFT_TRY(rocfft_plan_description_create(&desc));
    size_t stride = 1;
    CAF_FFT_TRY(rocfft_plan_description_set_data_layout(desc, rocfft_array_type_complex_interleaved,
                                                        rocfft_array_type_complex_interleaved, nullptr, nullptr, 1,
                                                        &stride, dist, 1, &stride, dist));
    rocfft_plan p = nullptr;
    rocfft_status s = rocfft_plan_create(&p, inplace ? rocfft_placement_inplace : rocfft_placement_notinplace,
                                         inverse ? rocfft_transform_type_complex_inverse
                                                 : rocfft_transform_type_complex_forward,
                                         rocfft_precision_single, 1, &len, batch, desc);
    rocfft_plan_description_destroy(desc);
    CAF_FFT_TRY(s);
    plan = p;
    CAF_FFT_TRY(rocfft_plan_get_work_buffer_size(p, &work_bytes));
    rocfft_execution_info ei = nullptr;
    CAF_FFT_TRY(rocfft_execution_info_create(&ei));
    info = ei;
    if (work_bytes) {
        CAF_HIP_TRY(hipMalloc(&work, work_bytes));
        CAF_FFT_TRY(rocfft_execution_info_set_work_buffer(ei, work, work_bytes));
    }
    return CAF_OK;
}

int FftPlan::exec(void* in, void* out, hipStream_t st) {
    CAF_FFT_TRY(rocfft_execution_info_set_stream((rocfft_execution_info)info, st));
    void* ib[1] = {in};
    void* ob[1] = {out};
    CAF_FFT_TRY(rocfft_execute((rocfft_plan)plan, ib, (out && out != in) ? ob : nullptr, (rocfft_execution_info)info));
    return CAF_OK;
}

void FftPlan::destroy() {
    if (info) rocfft_execution_info_destroy((rocfft_execution_info)info);
    if (plan) rocfft_plan_destroy((rocfft_plan)plan);
    if (work) (void)hipFree(work);
    info = nullptr;
    plan = nullptr;
    work = nullptr;
}

// ---- checkout cache of rocFFT plans --------------------------------------------------------------------------
// rocfft_plan_create costs milliseconds (kernel selection / code-object loading), more than a small CAF job
// itself; the per-call entry points of the host layer (fastXcorr, cztXcorr, ...) build a CAF plan per call.
// Plans are therefore parked on release and handed out again for the same (device, direction, length, batch,
// distance, placement).  A parked plan has exactly one owner after acquire (its work buffer is not shared).
namespace {
struct FftKey {
    int dev, inverse, inplace;
    size_t len, batch, dist;
    bool operator<(const FftKey& o) const {
        return std::tie(dev, inverse, inplace, len, batch, dist) < std::tie(o.dev, o.inverse, o.inplace, o.len, o.batch, o.dist);
    }
};
std::mutex g_fft_mu;
std::multimap<FftKey, FftPlan> g_fft_parked;
constexpr size_t FFT_PARK_MAX_PLANS = 32;
constexpr size_t FFT_PARK_MAX_WORK = (size_t)64 << 20;  // plans with larger work buffers are destroyed (<= 2 GiB parked)
}  // namespace

int fft_plan_acquire(FftPlan* out, bool inverse, size_t len, size_t batch, size_t dist, bool inplace) {
    int dev = 0;
    CAF_HIP_TRY(hipGetDevice(&dev));
    const FftKey k{dev, inverse ? 1 : 0, inplace ? 1 : 0, len, batch, dist};
    {
        std::lock_guard<std::mutex> lk(g_fft_mu);
        auto it = g_fft_parked.find(k);
        if (it != g_fft_parked.end()) {
            *out = it->second;
            g_fft_parked.erase(it);
            return CAF_OK;
        }
    }
    *out = FftPlan();
    const int rc = out->create(inverse, len, batch, dist, inplace);
    if (rc) out->destroy();
    out->key_dev = dev;
    out->key_inverse = inverse;
    out->key_inplace = inplace;
    out->key_len = len;
    out->key_batch = batch;
    out->key_dist = dist;
    return rc;
}

void fft_plan_release(FftPlan* p) {
    if (!p->plan) return;
    {
        std::lock_guard<std::mutex> lk(g_fft_mu);
        if (p->key_len && p->work_bytes <= FFT_PARK_MAX_WORK && g_fft_parked.size() < FFT_PARK_MAX_PLANS) {
            g_fft_parked.emplace(FftKey{p->key_dev, p->key_inverse ? 1 : 0, p->key_inplace ? 1 : 0, p->key_len, p->key_batch,
                                        p->key_dist},
                                 *p);
            *p = FftPlan();
            return;
        }
    }
    p->destroy();
}

}  // namespace caf
