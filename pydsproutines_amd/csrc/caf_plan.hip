// libcaf: plan object of the hypothesis engine + the C-ABI of include/caf.h.
// Host-side C++ around the gfx950 kernels of caf_kernels.hip and batched rocFFT.
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstring>
#include <mutex>
#include <vector>

#include <unistd.h>

#include "caf_internal.h"

namespace caf {

static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }

}  // namespace caf

using namespace caf;

// Auxiliary stream + fork/join events of a plan, checked out of a small per-device list: creating and destroying
// a HIP stream costs ~1 ms, as much as a whole small CAF job, and the per-call entry points of the host layer
// build a plan per call.  A released set is idle (the plan synchronises the device before releasing).
namespace {
struct AuxSet {
    int dev;
    hipStream_t s;
    hipEvent_t fork, join;
};
std::mutex g_aux_mu;
std::vector<AuxSet> g_aux_idle;

int aux_acquire(int dev, hipStream_t* s, hipEvent_t* fork, hipEvent_t* join) {
    {
        std::lock_guard<std::mutex> lk(g_aux_mu);
        for (size_t i = 0; i < g_aux_idle.size(); ++i)
            if (g_aux_idle[i].dev == dev) {
                *s = g_aux_idle[i].s;
                *fork = g_aux_idle[i].fork;
                *join = g_aux_idle[i].join;
                g_aux_idle.erase(g_aux_idle.begin() + (long)i);
                return CAF_OK;
            }
    }
    CAF_HIP_TRY(hipStreamCreateWithFlags(s, hipStreamNonBlocking));
    CAF_HIP_TRY(hipEventCreateWithFlags(fork, hipEventDisableTiming));
    CAF_HIP_TRY(hipEventCreateWithFlags(join, hipEventDisableTiming));
    return CAF_OK;
}

void aux_release(int dev, hipStream_t s, hipEvent_t fork, hipEvent_t join) {
    if (s && fork && join) {
        std::lock_guard<std::mutex> lk(g_aux_mu);
        if (g_aux_idle.size() < 16) {
            g_aux_idle.push_back(AuxSet{dev, s, fork, join});
            return;
        }
    }
    if (fork) (void)hipEventDestroy(fork);
    if (join) (void)hipEventDestroy(join);
    if (s) (void)hipStreamDestroy(s);
}
}  // namespace

struct caf_plan_t {
    int T = 0, N = 0, F = 0, G = 0;
    int freq_mode = 0, mul_mode = 0;
    int B = 0, step = 0, pitch = 0, nb = 0, tiles_per_blk = 0, hyp_per_wg = 16, fwd_chunk = 1;
    int npart = 1;  // 65536-point in-LDS engine: template partitions of 32768 samples (templates longer than 32768)
    int nb_nosurf = 0;  // persistent engine, no-surface mode: blocks per launch (the pair arrays are ~1/32 of the tiles)
    int64_t max_rx = 0, max_blocks = 0, partial_per_tmpl = 0;
    int device = 0;
    float2* d_hc = nullptr;
    int32_t* d_shifts = nullptr;
    float* d_tscale = nullptr;
    int32_t* d_gstart = nullptr;
    int32_t* d_glen = nullptr;
    double* d_tile_sums = nullptr;
    double* d_prefix = nullptr;
    float* d_inv_e = nullptr;
    float2* d_xb = nullptr;
    float2* d_xb2 = nullptr;  // 32768-point blocks: the block spectra parity-major (fused_item2q); 65536-point blocks: as parity pairs
    float2* d_pbuf = nullptr;
    PeakRec* d_partial = nullptr;
    bool fused = false;
    bool persistent = false;  // fused stages as one work-queue launch (k_caf_persistent)
    bool direct = false;      // time-domain evaluation over <= 64 non-zero template samples (caf_direct.hip)
    int dir_k = 0;
    int32_t* d_dir_pos = nullptr;
    float2* d_dir_w = nullptr;
    int n_cus = 0, tr_slots = 0;
    PersistParams* d_params = nullptr;
    int32_t* d_pq = nullptr;
    float* d_vt = nullptr;
    float2* d_tw1 = nullptr;
    float2* d_tw23 = nullptr;
    float2* d_uconj = nullptr;  // conj(u) per template in the time domain (T x N): product rows of the zoom (caf_zoom_czt)
    double* d_nu = nullptr;     // coarse frequency of hypothesis f in cycles per sample (F)
    FftPlan fwd, inv;
    hipStream_t s_aux = nullptr;  // sliding-energy pass beside gather + forward FFTs
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int64_t workspace_bytes = 0;
    // profiling
    bool prof = false;
    struct Rec {
        int stage;
        hipEvent_t a, b;
    };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    double acc_ms[CAF_NUM_STAGES] = {0};
    int64_t acc_n[CAF_NUM_STAGES] = {0};

    hipEvent_t get_event() {
        if (!pool.empty()) {
            hipEvent_t e = pool.back();
            pool.pop_back();
            return e;
        }
        hipEvent_t e;
        (void)hipEventCreate(&e);
        return e;
    }
    void stage_begin(int stage, hipStream_t st) {
        if (!prof) return;
        Rec r;
        r.stage = stage;
        r.a = get_event();
        r.b = get_event();
        (void)hipEventRecord(r.a, st);
        recs.push_back(r);
    }
    void stage_end(hipStream_t st) {
        if (!prof) return;
        (void)hipEventRecord(recs.back().b, st);
    }
    void drain() {
        for (auto& r : recs) {
            (void)hipEventSynchronize(r.b);
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
                acc_ms[r.stage] += ms;
                acc_n[r.stage] += 1;
            }
            pool.push_back(r.a);
            pool.push_back(r.b);
        }
        recs.clear();
    }
    template <typename Tp>
    int alloc(Tp** p, int64_t count) {
        const int64_t bytes = count * (int64_t)sizeof(Tp);
        // through the caching allocator: the per-call entry points of the host layer build a plan per call
        const int rc = pool_alloc((void**)p, std::max<int64_t>(bytes, 16));
        if (rc) return rc;
        workspace_bytes += bytes;
        return CAF_OK;
    }
    void release() {
        drain();
        // nothing of this plan may still be running when its buffers and FFT plans go back to the caches
        (void)hipDeviceSynchronize();
        for (auto e : pool) (void)hipEventDestroy(e);
        pool.clear();
        fft_plan_release(&fwd);
        fft_plan_release(&inv);
        aux_release(device, s_aux, ev_fork, ev_join);
        s_aux = nullptr;
        ev_fork = ev_join = nullptr;
        void* ptrs[] = {d_dir_pos, d_dir_w, d_hc,  d_shifts, d_tscale, d_gstart,  d_glen, d_tile_sums, d_prefix, d_inv_e,
                        d_xb,  d_pbuf,   d_partial, d_vt,     d_params, d_pq,     d_uconj, d_nu,    d_xb2};  // (d_tw1 / d_tw23 are shared, per device)
        for (void* p : ptrs)
            if (p) (void)pool_free(p);
    }
};

// inter-pass twiddles of the 16*16*16*4 decomposition (caf_fused.hip), computed in f64 once per device and kept
// for the life of the process: every fused plan reads the same two small tables
static int fused_twiddles(int device, float2** tw1_out, float2** tw23_out) {
    static std::mutex mu;
    static std::vector<std::pair<float2*, float2*>> per_dev;
    std::lock_guard<std::mutex> lk(mu);
    if ((int)per_dev.size() <= device) per_dev.resize(device + 1, {nullptr, nullptr});
    if (!per_dev[device].first) {
        std::vector<std::complex<float>> tw1(16 * 1024), tw23(2 * (16 * 64 + 16 * 4));
        auto cis = [](double num, double den) {
            const double ph = 2.0 * M_PI * std::fmod(num, den) / den;
            return std::complex<float>((float)std::cos(ph), (float)std::sin(ph));
        };
        for (int n1 = 0; n1 < 16; ++n1)
            for (int m2 = 0; m2 < 1024; ++m2) tw1[n1 * 1024 + m2] = cis((double)n1 * m2, 16384.0);
        for (int n2 = 0; n2 < 16; ++n2)
            for (int c = 0; c < 64; ++c) tw23[n2 * 64 + c] = cis((double)n2 * c, 1024.0);
        for (int n3 = 0; n3 < 16; ++n3)
            for (int dd = 0; dd < 4; ++dd) tw23[1024 + n3 * 4 + dd] = cis((double)n3 * dd, 64.0);
        // the same two tables for the ODD half-transform of a 32768-point block (fused_item2q): its outputs carry the
        // combination twiddle W_32768^{n'}, factor by factor in the pass that produces each output digit
        for (int n2 = 0; n2 < 16; ++n2)
            for (int c = 0; c < 64; ++c) tw23[1088 + n2 * 64 + c] = cis((double)n2 * (2 * c + 1), 2048.0);
        for (int n3 = 0; n3 < 16; ++n3)
            for (int dd = 0; dd < 4; ++dd) tw23[1088 + 1024 + n3 * 4 + dd] = cis((double)n3 * (2 * dd + 1), 128.0);
        float2 *a = nullptr, *b = nullptr;
        CAF_HIP_TRY(hipMalloc((void**)&a, tw1.size() * 8));
        hipError_t e = hipMalloc((void**)&b, tw23.size() * 8);
        if (e == hipSuccess) e = hipMemcpy(a, tw1.data(), tw1.size() * 8, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(b, tw23.data(), tw23.size() * 8, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(a);
            (void)hipFree(b);
            CAF_HIP_TRY(e);
        }
        per_dev[device] = {a, b};
    }
    *tw1_out = per_dev[device].first;
    *tw23_out = per_dev[device].second;
    return CAF_OK;
}

static int ilog2_ceil(int64_t v) {
    int l = 0;
    while (((int64_t)1 << l) < v) ++l;
    return l;
}

namespace caf {
int plan_zoom_view(caf_plan p, PlanZoomView* v) {
    CAF_REQUIRE(p && v, "NULL plan");
    v->T = p->T;
    v->N = p->N;
    v->F = p->F;
    v->G = p->G;
    v->device = p->device;
    v->d_uconj = p->d_uconj;
    v->d_nu = p->d_nu;
    v->d_tscale = p->d_tscale;
    return CAF_OK;
}
}  // namespace caf

extern "C" {

int32_t caf_last_error(char* buf, int32_t len) {
    if (!buf || len <= 0) return CAF_ERR_INVALID;
    std::strncpy(buf, g_last_error.c_str(), (size_t)len - 1);
    buf[len - 1] = 0;
    return CAF_OK;
}

int32_t caf_abi_version(void) { return (1 << 16) | 9; }  // minor: +1 per batch of added entry points

int32_t caf_device_count(int32_t* count) {
    CAF_REQUIRE(count, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        set_error(std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
        return CAF_ERR_NODEVICE;
    }
    *count = n;
    return CAF_OK;
}

int32_t caf_set_device(int32_t device) {
    CAF_HIP_TRY(hipSetDevice(device));
    return CAF_OK;
}

int32_t caf_device_info(int32_t device, char* name, int32_t name_len, int64_t* total_mem, int32_t* compute_units) {
    hipDeviceProp_t prop;
    CAF_HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (name && name_len > 0) {
        std::snprintf(name, (size_t)name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (total_mem) *total_mem = (int64_t)prop.totalGlobalMem;
    if (compute_units) *compute_units = prop.multiProcessorCount;
    return CAF_OK;
}

int32_t caf_malloc(void** d_ptr, int64_t bytes) {
    CAF_REQUIRE(d_ptr && bytes >= 0, "caf_malloc: bad arguments");
    return pool_alloc(d_ptr, bytes);
}
int32_t caf_free(void* d_ptr) { return pool_free(d_ptr); }
int32_t caf_pool_trim(void) {
    pool_trim();
    return CAF_OK;
}
int32_t caf_pool_stats(int64_t* cached_bytes, int64_t* in_use_bytes, int64_t* hits, int64_t* misses) {
    pool_stats(cached_bytes, in_use_bytes, hits, misses);
    return CAF_OK;
}
int32_t caf_memset(void* d_ptr, int32_t value, int64_t bytes, void* stream) {
    CAF_HIP_TRY(hipMemsetAsync(d_ptr, value, (size_t)bytes, (hipStream_t)stream));
    return CAF_OK;
}
int32_t caf_h2d(void* d_dst, const void* h_src, int64_t bytes, void* stream) {
    CAF_REQUIRE(bytes >= 0 && (bytes == 0 || (d_dst && h_src)), "caf_h2d: bad arguments");
    return host_h2d(d_dst, h_src, bytes, (hipStream_t)stream);
}
int32_t caf_d2h(void* h_dst, const void* d_src, int64_t bytes, void* stream) {
    CAF_REQUIRE(bytes >= 0 && (bytes == 0 || (h_dst && d_src)), "caf_d2h: bad arguments");
    return host_d2h(h_dst, d_src, bytes, (hipStream_t)stream);
}
int32_t caf_d2h_f64(double* h_dst, const float* d_src, int64_t count, void* stream) {
    CAF_REQUIRE(count >= 0 && (count == 0 || (h_dst && d_src)), "caf_d2h_f64: bad arguments");
    return host_d2h_f64(h_dst, d_src, count, (hipStream_t)stream);
}
int32_t caf_d2h_transposed(void* h_dst, int32_t dst_f64, const float* d_src, int64_t rows, int64_t pitch, int64_t col0,
                           int64_t ncols, void* stream) {
    CAF_REQUIRE(h_dst && d_src, "caf_d2h_transposed: NULL");
    CAF_REQUIRE(rows >= 1 && rows <= 65536 && ncols >= 0 && col0 >= 0 && col0 + ncols <= pitch, "caf_d2h_transposed: bad shape");
    return host_d2h_transposed(h_dst, dst_f64 != 0, d_src, rows, pitch, col0, ncols, (hipStream_t)stream);
}
int32_t caf_d2d(void* d_dst, const void* d_src, int64_t bytes, void* stream) {
    CAF_HIP_TRY(hipMemcpyAsync(d_dst, d_src, (size_t)bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return CAF_OK;
}
int32_t caf_stream_sync(void* stream) {
    CAF_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return CAF_OK;
}
int32_t caf_stream_create(void** stream) {
    CAF_REQUIRE(stream, "caf_stream_create: NULL");
    hipStream_t s = nullptr;
    CAF_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void*)s;
    return CAF_OK;
}
int32_t caf_stream_destroy(void* stream) {
    if (stream) CAF_HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return CAF_OK;
}

int32_t caf_plan_destroy(caf_plan plan) {
    if (!plan) return CAF_OK;
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != plan->device) (void)hipSetDevice(plan->device);
    plan->release();
    if (cur >= 0 && cur != plan->device) (void)hipSetDevice(cur);  // leave the caller's current device as it was
    delete plan;
    return CAF_OK;
}

// Direct engine (caf_direct.hip): everything the plan needs is the list of non-zero template positions, one complex
// multiplier per (template, hypothesis, position), the template scales, the peak records -- and what caf_zoom_czt reads.
static int32_t plan_build_direct(caf_plan p, const caf_plan_desc* d, const std::vector<int32_t>& gs,
                                 const std::vector<int32_t>& gl, const std::vector<int32_t>& nz) {
    const int T = p->T, N = p->N, F = p->F, K = (int)nz.size();
    const std::complex<float>* tm = reinterpret_cast<const std::complex<float>*>(d->h_templates);
    p->direct = true;
    p->dir_k = K;
    p->B = p->step = p->nb = 0;
    p->partial_per_tmpl = (d->max_rx_len - N + 1 + 127) / 128;  // one record per workgroup of 128 delays (caf_direct.hip)
    std::vector<double> nu(F);
    for (int f = 0; f < F; ++f) {
        if (d->freq_mode == CAF_FREQ_BINS) {
            CAF_REQUIRE(d->h_bins && d->grid >= 1, "CAF_FREQ_BINS needs bins and grid");
            nu[f] = (double)d->h_bins[f] / (double)d->grid;
        } else {
            CAF_REQUIRE(d->h_freqs_norm, "CAF_FREQ_NORM needs freqs_norm");
            nu[f] = d->h_freqs_norm[f];
        }
    }
    // w[t][f][k] = conj(u_t[n_k]) exp(-j 2 pi nu_f n_k),  u = auto_conj ? tmpl : conj(tmpl)
    std::vector<std::complex<float>> w((size_t)T * F * K);
    std::vector<float> tscale(T);
    for (int t = 0; t < T; ++t) {
        double e = 0.0;
        for (int n = 0; n < N; ++n) e += std::norm(std::complex<double>(tm[(size_t)t * N + n]));
        tscale[t] = (float)(1.0 / e);
        for (int f = 0; f < F; ++f)
            for (int k = 0; k < K; ++k) {
                std::complex<double> u(tm[(size_t)t * N + nz[k]]);
                if (d->auto_conj) u = std::conj(u);  // conj(u) with u = tmpl
                double cyc = nu[f] * (double)nz[k];
                cyc -= std::floor(cyc);  // phase reduced in cycles before the trig call
                const double ph = -2.0 * M_PI * cyc;
                u *= std::complex<double>(std::cos(ph), std::sin(ph));
                w[((size_t)t * F + f) * K + k] = std::complex<float>((float)u.real(), (float)u.imag());
            }
    }
    int rc;
    if ((rc = p->alloc(&p->d_dir_pos, K)) || (rc = p->alloc(&p->d_dir_w, (int64_t)T * F * K)) || (rc = p->alloc(&p->d_tscale, T)) ||
        (rc = p->alloc(&p->d_partial, (int64_t)T * p->partial_per_tmpl + (int64_t)T * PEAK_PARTS)) ||
        (rc = p->alloc(&p->d_gstart, p->G)) || (rc = p->alloc(&p->d_glen, p->G)) || (rc = p->alloc(&p->d_uconj, (int64_t)T * N)) ||
        (rc = p->alloc(&p->d_nu, F)))
        return rc;
    std::vector<std::complex<float>> uc((size_t)T * N);
    for (size_t i = 0; i < uc.size(); ++i) uc[i] = d->auto_conj ? std::conj(tm[i]) : tm[i];
    CAF_H2D(p->d_dir_pos, nz.data(), (size_t)K * 4);
    CAF_H2D(p->d_dir_w, w.data(), w.size() * 8);
    CAF_H2D(p->d_tscale, tscale.data(), (size_t)T * 4);
    CAF_H2D(p->d_gstart, gs.data(), gs.size() * 4);
    CAF_H2D(p->d_glen, gl.data(), gl.size() * 4);
    CAF_H2D(p->d_uconj, uc.data(), uc.size() * 8);
    CAF_H2D(p->d_nu, nu.data(), nu.size() * 8);
    return CAF_OK;
}

static int32_t plan_build(caf_plan p, const caf_plan_desc* d) {
    CAF_REQUIRE(d->num_templates >= 1 && d->template_len >= 1 && d->h_templates, "need >= 1 template");
    CAF_REQUIRE(d->num_freqs >= 1, "need >= 1 frequency hypothesis");
    CAF_REQUIRE(d->max_rx_len >= d->template_len, "max_rx_len shorter than the template");
    CAF_REQUIRE(d->max_rx_len < ((int64_t)1 << 31) - (1 << 20), "rx longer than 2^31 samples is not supported");
    CAF_REQUIRE(d->freq_mode == CAF_FREQ_BINS || d->freq_mode == CAF_FREQ_NORM, "bad freq_mode");
    CAF_HIP_TRY(hipGetDevice(&p->device));
    p->T = d->num_templates;
    p->N = d->template_len;
    p->F = d->num_freqs;
    p->freq_mode = d->freq_mode;
    p->max_rx = d->max_rx_len;
    const int T = p->T, N = p->N, F = p->F;

    // groups (support of the template for the energy normalisation)
    std::vector<int32_t> gs, gl;
    if (d->num_groups >= 1 && d->h_group_start && d->h_group_len) {
        gs.assign(d->h_group_start, d->h_group_start + d->num_groups);
        gl.assign(d->h_group_len, d->h_group_len + d->num_groups);
    } else {
        gs = {0};
        gl = {N};
    }
    p->G = (int)gs.size();
    for (int g = 0; g < p->G; ++g)
        CAF_REQUIRE(gs[g] >= 0 && gl[g] >= 1 && (int64_t)gs[g] + gl[g] <= N, "group outside the template span");

    // engine: the fused LDS-resident kernel works on 16384-point blocks
    CAF_REQUIRE(d->engine >= CAF_ENGINE_AUTO && d->engine <= CAF_ENGINE_DIRECT && d->reserved == 0, "bad engine field");
    {   // the direct engine: asked for, or chosen for composite templates whose groups cover fewer than 64 samples
        int64_t support = 0;
        for (int g = 0; g < p->G; ++g) support += gl[g];
        static const bool auto_direct = [] {
            const char* e = getenv("CAF_DIRECT");  // A/B switch for AUTO plans, default on
            return !e || atoi(e);
        }();
        const bool composite = d->num_groups >= 1 && d->h_group_start && d->h_group_len;
        if (d->engine == CAF_ENGINE_DIRECT || (d->engine == CAF_ENGINE_AUTO && auto_direct && composite && support < 64)) {
            const std::complex<float>* tm0 = reinterpret_cast<const std::complex<float>*>(d->h_templates);
            std::vector<int32_t> nz;  // union over the templates of their non-zero positions
            for (int n = 0; n < N && nz.size() <= 64; ++n)
                for (int t = 0; t < T; ++t)
                    if (tm0[(size_t)t * N + n] != std::complex<float>(0.f, 0.f)) {
                        nz.push_back(n);
                        break;
                    }
            const bool ok = !nz.empty() && nz.size() <= 64;
            CAF_REQUIRE(d->engine != CAF_ENGINE_DIRECT || ok, "the direct engine needs 1 .. 64 non-zero template samples");
            if (ok) return plan_build_direct(p, d, gs, gl, nz);
        }
    }
    // the LDS-resident engines: 16384-point blocks for templates up to 8192 samples, 32768-point blocks (two chained
    // 16384-point transforms per hypothesis, persistent engine only) up to 16384
    // (CAF_FUSED_LB15=1: 32768-point blocks for the shorter templates too -- A/B switch: 87.5 % valid outputs per block at
    //  N = 4096 instead of 75 %, against the dearer half-transforms of the 32768-point role)
    static const bool lb15_env = [] {
        const char* e = getenv("CAF_FUSED_LB15");
        return e && atoi(e);
    }();
    // (templates of 16385 .. 32768 samples: 65536-point blocks in the folded form, two chained transforms per output residue,
    //  fused_item2q<FOLD>; CAF_FUSED_LB16=0 sends them to the rocfft engine as before -- A/B switch)
    static const bool lb16_env = [] {
        const char* e = getenv("CAF_FUSED_LB16");
        return !e || atoi(e);
    }();
    // (templates of 32769 .. 262144 samples: the same 65536-point blocks with the template cut into partitions of 32768
    //  samples -- Z_b = sum_p X_{b + p} . Hc_p, the frequency-domain delay line of partitioned convolution: the products of the
    //  item's block and of the blocks that follow it with the partitions' spectra are summed before the one inverse transform,
    //  fused_item2q<PART>; CAF_FUSED_PARTS = the largest number of partitions taken, 1 sends them to the rocfft engine -- A/B switch)
    static const int max_parts = [] {
        const char* e = getenv("CAF_FUSED_PARTS");
        return e ? std::min(std::max(atoi(e), 1), 16) : 8;
    }();
    const int fused_lb = (N <= 8192 && !lb15_env) ? 14 : N <= 16384 ? 15 : 16;
    const int fused_parts = fused_lb == 16 ? (N + 32767) / 32768 : 1;
    // on-grid hypotheses are circular shifts of one template spectrum: every bin must land on a whole element of the engine's
    // block (any grid that divides 16384 does; so does bin 0 on any grid), an even one where the rows are read parity-major
    const bool bins_fit = [&] {
        if (d->freq_mode != CAF_FREQ_BINS) return true;
        if (d->grid < 1 || !d->h_bins) return false;
        const int64_t Bf = (int64_t)1 << fused_lb;
        for (int f = 0; f < F; ++f) {
            const int64_t num = (int64_t)d->h_bins[f] * Bf;
            if (num % d->grid != 0 || (fused_lb >= 15 && ((num / d->grid) & 1))) return false;
        }
        return true;
    }();
    const bool fused_ok = N <= (lb16_env ? 32768 * max_parts : 16384) && bins_fit && (d->log2_block == 0 || d->log2_block == fused_lb);
    CAF_REQUIRE(d->engine != CAF_ENGINE_PERSISTENT || fused_ok,
                "the persistent engine needs template_len <= 262144, bins on whole (even, beyond 8192 samples) elements of its block and log2_block 0, 14 (<= 8192 samples), 15 (<= 16384) or 16");
    CAF_REQUIRE(d->engine != CAF_ENGINE_FUSED || (fused_ok && N <= 8192),
                "the two-launch fused engine needs template_len <= 8192, bins on whole elements of its 16384-point block and log2_block 0 or 14");
    p->fused = (d->engine == CAF_ENGINE_FUSED) || (d->engine == CAF_ENGINE_PERSISTENT) ||
               (d->engine == CAF_ENGINE_AUTO && fused_ok);
    // one launch for both stages whenever the fused FFT applies: its tile role handles any F (steps of 32
    // hypotheses with hardware bounds handling, a streaming path for F == 1) and was faster than the two-launch
    // form on every measured shape (C2 1.15x, C4 share 1.25x, C3 with 64 templates and no frequency scan ~10x)
    p->persistent = d->engine == CAF_ENGINE_PERSISTENT || (d->engine == CAF_ENGINE_AUTO && fused_ok);
    if (const char* e = getenv("CAF_PERSISTENT"))  // A/B switch for AUTO plans
        if (d->engine == CAF_ENGINE_AUTO && fused_ok && N <= 8192) p->persistent = atoi(e) != 0;

    // block size: B = 2^k, B >= 2N (>= 50 % valid outputs); default 16 N clipped to [2^12, 2^18]
    int lb = p->fused ? fused_lb : d->log2_block;
    const int lmin = ilog2_ceil(2 * (int64_t)N);
    if (lb <= 0) {
        lb = std::min(std::max(ilog2_ceil(16 * (int64_t)N), 12), 18);
        // no point in blocks much longer than the data
        lb = std::min(lb, std::max(ilog2_ceil(d->max_rx_len), 12));
        lb = std::max(lb, lmin);
    }
    p->npart = p->fused ? fused_parts : 1;
    CAF_REQUIRE((lb >= lmin || p->npart > 1) && lb <= 24, "log2_block must satisfy 2N <= 2^log2_block <= 2^24");
    lb = std::max(lb, 9);  // the multiply kernel tiles 512 points per workgroup
    p->B = 1 << lb;
    p->step = p->B - N + 1;
    // In-LDS engines: delays leave a block in tiles of 64.  A step that is a few delays over a multiple of 64 (N = 4096:
    // 12289 = 192 * 64 + 1) costs a 193rd tile per block holding one delay -- a tile item of its own for the tile role
    // (1365 of 17745 items at C2) and, in the FFT role, the whole last quarter of the radix-4 pass for one output.
    // Giving up those few delays per block (< 0.33 % more blocks) removes both.
    if (p->fused && p->step % 64 != 0 && (p->step % 64) * 300 <= p->step) p->step -= p->step % 64;
    // 65536-point blocks: only the first half of a block's outputs is used -- the folded form computes exactly the two output
    // residues 2 k + r, k < 16384, each as the lower half of a 32768-point transform (fused_item2q<FOLD>) -- so a block yields
    // 32768 delays whatever the template length in (16384, 32768]
    if (p->fused && lb == 16) p->step = 32768;
    p->pitch = p->B + 64;  // break the power-of-two stride between hypothesis rows
    const int B = p->B;

    // frequency hypotheses
    std::vector<int32_t> shifts;
    bool all_even = true;
    if (d->freq_mode == CAF_FREQ_BINS) {
        CAF_REQUIRE(d->h_bins && d->grid >= 1, "CAF_FREQ_BINS needs bins and grid");
        shifts.resize(F);
        for (int f = 0; f < F; ++f) {
            const int64_t num = (int64_t)d->h_bins[f] * B;
            CAF_REQUIRE(num % d->grid == 0, "every bin must be a whole number of elements of the block: bins * block_size / grid (use CAF_FREQ_NORM otherwise)");
            int64_t s = (num / d->grid) % B;
            if (s < 0) s += B;
            shifts[f] = (int32_t)s;
            if (s & 1) all_even = false;
        }
        p->mul_mode = all_even ? 0 : 1;
    } else {
        CAF_REQUIRE(d->h_freqs_norm, "CAF_FREQ_NORM needs freqs_norm");
        p->mul_mode = 2;
    }
    const int NP = p->npart, PLEN = NP > 1 ? 32768 : N;  // partitions per template and their length (one partition: the template)
    const int64_t nspec = ((p->mul_mode == 2) ? (int64_t)T * F : T) * NP;  // template spectra held (one row per partition)

    // batch: aim at ~128 MiB of hypothesis products in flight
    const int64_t total_blocks = (d->max_rx_len - N + 1 + p->step - 1) / p->step;
    p->tiles_per_blk = (p->step + MAG_S - 1) / MAG_S;
    int nb = d->blocks_per_batch;
    if (nb <= 0) {
        if (p->fused) {
            // |y|^2 tiles of a batch: up to 18 GiB of the 288 GB HBM, so that config C2 (17.3 GB) is ONE
            // launch of 5460 workgroups (21.3 rounds over 256 CUs: small tail) instead of many short ones
            const int64_t per_block = (int64_t)p->tiles_per_blk * T * F * 64 * 4;
            nb = (int)std::max<int64_t>(1, std::min<int64_t>(65535, ((int64_t)18 << 30) / per_block));
        } else {
            const int64_t per_block = (int64_t)T * F * p->pitch * 8;
            nb = (int)std::max<int64_t>(1, std::min<int64_t>(64, ((int64_t)128 << 20) / per_block));
        }
    }
    nb = (int)std::min<int64_t>(nb, std::max<int64_t>(1, total_blocks));
    p->nb = nb;
    p->max_blocks = (total_blocks + nb - 1) / nb * nb;
    p->partial_per_tmpl = p->max_blocks * p->tiles_per_blk;
    p->hyp_per_wg = (int)std::min<int64_t>(p->fused ? 64 : 16, (int64_t)T * F);
    if (const char* e = getenv("CAF_HYP_PER_WG"))  // A/B switch: hypotheses per FFT work item
        if (atoi(e) >= 1) p->hyp_per_wg = (int)std::min<int64_t>(std::min(atoi(e), 256), (int64_t)T * F);
    if (p->fused) {
        // small jobs: fewer hypotheses per FFT work item, so that there are about two items per CU when the job
        // allows it (one block x 32 hypotheses as ONE item kept 255 CUs idle for 0.25 ms)
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, p->device);
        const int64_t want = ((int64_t)T * F * total_blocks + 2 * cus - 1) / (2 * cus);
        p->hyp_per_wg = (int)std::min<int64_t>(p->hyp_per_wg, std::max<int64_t>(4, want));
        // whole rounds: 1365 items of 64 hypotheses on 256 CUs are 5.33 rounds, i.e. 6 with a third of the chip idle in
        // the last one (config C3); halve the items while that buys more than the extra block-spectrum loads cost
        if (!getenv("CAF_HYP_PER_WG")) {
            auto eff = [&](int hp) {
                const int64_t items = (((int64_t)T * F + hp - 1) / hp) * total_blocks;
                const int64_t rounds = (items + cus - 1) / cus;
                return (double)items / (double)(rounds * cus);
            };
            int best = p->hyp_per_wg;
            for (int hp = best / 2; hp >= 16; hp /= 2)
                if (eff(hp) > eff(best) + 0.03) best = hp;
            p->hyp_per_wg = best;
        }
        // 65536-point blocks: the role re-reads its block spectrum per sub-transform whatever the item size, and smaller items
        // keep fewer blocks (512 KB of spectrum each) in flight per XCD: 44.8 -> 43.5 ms at the C2 shape with 32 (16: 43.3)
        // (partitioned templates read npart block spectra and npart template rows per sub-transform: smaller items again -- 16
        //  hypotheses at 2 .. 5 partitions (35.7 -> 34.9 ms at two, 59.1 -> 58.1 at four), 8 beyond (127.9 -> 120.8 ms at eight))
        if (lb == 16 && !getenv("CAF_HYP_PER_WG")) p->hyp_per_wg = std::min(p->hyp_per_wg, p->npart >= 6 ? 8 : p->npart > 1 ? 16 : 32);
        // groups that do not straddle templates allow the no-surface mode (running maxima instead of tiles):
        // prefer the largest divisor of F that is not much smaller than the group size chosen above
        if (F % p->hyp_per_wg != 0)
            for (int dv = p->hyp_per_wg; dv >= std::max(4, p->hyp_per_wg / 3); --dv)
                if (F % dv == 0) {
                    p->hyp_per_wg = dv;
                    break;
                }
    }

    // No-surface mode of the persistent engine: one (value, hypothesis) pair per delay and hypothesis GROUP lives in
    // the tile buffer instead of one value per delay and hypothesis, so the same buffer holds hyp_per_wg / 2 times
    // as many blocks per launch (config C4's 512 templates x 512 bins on one GPU: 32 blocks instead of 1; fewer,
    // longer launches = fewer drain tails where the last blocks' reductions run on a handful of CUs).
    p->nb_nosurf = nb;
    if (p->persistent && p->B == 16384 && F >= p->hyp_per_wg) {
        const int64_t gpt = (F + p->hyp_per_wg - 1) / p->hyp_per_wg;
        if (2 * gpt <= F) {
            const int64_t cap = (int64_t)nb * F / (2 * gpt);  // nb_nosurf * 2 * T * gpt <= nb * T * F pairs
            p->nb_nosurf = (int)std::max<int64_t>(nb, std::min<int64_t>(std::min<int64_t>(cap, 4096), std::max<int64_t>(1, total_blocks)));
        }
    }

    // device buffers
    int rc;
    if ((rc = p->alloc(&p->d_hc, nspec * B))) return rc;
    if ((rc = p->alloc(&p->d_shifts, std::max(F, 1)))) return rc;
    if ((rc = p->alloc(&p->d_tscale, T))) return rc;
    if ((rc = p->alloc(&p->d_gstart, p->G))) return rc;
    if ((rc = p->alloc(&p->d_glen, p->G))) return rc;
    if ((rc = p->alloc(&p->d_tile_sums, prefix_num_tiles(d->max_rx_len) + 1024))) return rc;
    if ((rc = p->alloc(&p->d_prefix, energy_prefix_doubles(d->max_rx_len)))) return rc;
    if ((rc = p->alloc(&p->d_inv_e, d->max_rx_len))) return rc;
    // all rx block spectra are produced up front, fwd_chunk blocks per rocFFT launch
    // (~32 MiB of spectra per forward launch: 256 blocks of 16384, 64 blocks of 65536, ...)
    p->fwd_chunk = (int)std::min<int64_t>(std::max<int64_t>(1, ((int64_t)1 << 22) / B), p->max_blocks);
    if ((rc = p->alloc(&p->d_xb, (p->max_blocks + NP - 1 + p->fwd_chunk) * B))) return rc;
    if (p->fused && p->B >= 32768 && (rc = p->alloc(&p->d_xb2, (p->max_blocks + NP - 1 + p->fwd_chunk) * B))) return rc;
    if (p->fused) {
        if ((rc = p->alloc(&p->d_vt, (int64_t)nb * p->tiles_per_blk * T * F * 64))) return rc;
        if ((rc = fused_twiddles(p->device, &p->d_tw1, &p->d_tw23))) return rc;  // per device, shared by all plans
        if (p->persistent) {
            int ncu = 0;  // (hipGetDeviceProperties costs ~1 ms per call; the attribute query does not)
            CAF_HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, p->device));
            p->n_cus = ncu;  // one resident 1024-thread workgroup per CU
            p->tr_slots = 12;                     // 96 CUs look at the tile queue first (measured optimum on C2: 10..14)
            if (const char* e = getenv("CAF_PERSIST_WGS")) p->n_cus = std::max(1, atoi(e));
            if (const char* e = getenv("CAF_PERSIST_TR_SLOTS")) p->tr_slots = std::max(0, atoi(e));
            p->tr_slots = std::min(p->tr_slots, 31);  // slot 0 of every XCD never prefers tiles (termination argument)
            if ((rc = p->alloc(&p->d_params, 1))) return rc;
            if ((rc = p->alloc(&p->d_pq, 12 + std::max(nb, p->nb_nosurf)))) return rc;
        }
    } else {
        if ((rc = p->alloc(&p->d_pbuf, (int64_t)nb * T * F * p->pitch))) return rc;
    }
    // tile records + PEAK_PARTS records per template for the two-stage reduction
    if ((rc = p->alloc(&p->d_partial, (int64_t)T * p->partial_per_tmpl + (int64_t)T * PEAK_PARTS))) return rc;

    // template spectra: u = auto_conj ? tmpl : conj(tmpl);  u_f[n] = u[n] exp(+j 2 pi nu_f n);
    // Hc = conj(FFT_B(u_f)) / B  (rocFFT's inverse is unnormalised, 1/B is folded in here)
    std::vector<float> tscale(T);
    const std::complex<float>* tm = reinterpret_cast<const std::complex<float>*>(d->h_templates);
    for (int t = 0; t < T; ++t) {
        double e = 0.0;
        for (int n = 0; n < N; ++n) e += std::norm(std::complex<double>(tm[(size_t)t * N + n]));
        tscale[t] = (float)(1.0 / e);
    }
    void *tmp_tm = nullptr, *tmp_nu = nullptr;  // inputs of the device-side table build, freed after the sync below
    auto free_tmp = [&]() {
        (void)pool_free(tmp_tm);
        (void)pool_free(tmp_nu);
        tmp_tm = tmp_nu = nullptr;
    };
    if (p->mul_mode == 2 && T <= 65535) {
        // explicit frequencies: the T*F modulated templates are generated on the device (k_build_hyp_time)
        if ((rc = pool_alloc(&tmp_tm, (int64_t)T * N * 8)) || (rc = pool_alloc(&tmp_nu, (int64_t)F * 8))) {
            free_tmp();
            return rc;
        }
        if ((rc = host_h2d(tmp_tm, tm, (int64_t)T * N * 8, nullptr)) || (rc = host_h2d(tmp_nu, d->h_freqs_norm, (int64_t)F * 8, nullptr))) {
            free_tmp();
            return rc;
        }
        launch_build_hyp_time((const float2*)tmp_tm, (const double*)tmp_nu, N, B, F, T, d->auto_conj ? 0 : 1, p->d_hc,
                              nullptr, NP, PLEN);
    } else {
        std::vector<std::complex<float>> host((size_t)nspec * B, std::complex<float>(0.f, 0.f));
        for (int t = 0; t < T; ++t) {
            // (row of partition q of a spectrum: the samples q * PLEN .. of the template at the row's start; the on-grid shifts
            //  need no phase per partition: an even shift at 65536 points makes nu * 32768 q a whole number of cycles)
            if (p->mul_mode != 2) {
                for (int n = 0; n < N; ++n) {
                    std::complex<float> u = tm[(size_t)t * N + n];
                    host[((size_t)t * NP + n / PLEN) * B + n % PLEN] = d->auto_conj ? u : std::conj(u);
                }
            } else {
                for (int f = 0; f < F; ++f) {
                    const double nu = d->h_freqs_norm[f];
                    std::complex<float>* dst0 = &host[((size_t)t * F + f) * NP * B];
                    for (int n = 0; n < N; ++n) {
                        std::complex<float>* dst = dst0 + (size_t)(n / PLEN) * B - (size_t)(n / PLEN) * PLEN;
                        std::complex<double> u(tm[(size_t)t * N + n]);
                        if (!d->auto_conj) u = std::conj(u);
                        // reduce the phase in cycles before the trig call to keep full f64 accuracy
                        double cyc = nu * (double)n;
                        cyc -= std::floor(cyc);
                        const double ph = 2.0 * M_PI * cyc;
                        u *= std::complex<double>(std::cos(ph), std::sin(ph));
                        dst[n] = std::complex<float>((float)u.real(), (float)u.imag());
                    }
                }
            }
        }
        CAF_H2D(p->d_hc, host.data(), host.size() * sizeof(std::complex<float>));
    }
    {
        FftPlan tmp;
        rc = fft_plan_acquire(&tmp, false, (size_t)B, (size_t)nspec, (size_t)B);
        if (rc == CAF_OK) rc = tmp.exec(p->d_hc, nullptr, nullptr);
        if (rc == CAF_OK) launch_conj_scale(p->d_hc, nspec * B, 1.0f / (float)B, nullptr);
        if (rc == CAF_OK && p->fused && B == 65536) {
            // the 65536-point engine reads its template-spectrum rows as pairs of the two halves of each parity, every
            // 1024-chunk in butterfly order, which this one pass produces as well (the permutation below is skipped)
            float2* tmp = nullptr;
            rc = pool_alloc((void**)&tmp, nspec * (int64_t)B * 8);
            if (rc == CAF_OK) {
                launch_parity_pairs(p->d_hc, tmp, nspec, B / 4, nullptr);
                if (hipMemcpyAsync(p->d_hc, tmp, (size_t)nspec * B * 8, hipMemcpyDeviceToDevice, nullptr) != hipSuccess) rc = CAF_ERR_HIP;
                (void)hipStreamSynchronize(nullptr);
                (void)pool_free(tmp);
            }
        }
        if (rc == CAF_OK && p->fused && B == 32768) {
            // the 32768-point engine reads its template-spectrum rows parity-major (even samples, then odd samples)
            float2* tmp = nullptr;
            rc = pool_alloc((void**)&tmp, nspec * (int64_t)B * 8);
            if (rc == CAF_OK) {
                launch_parity_major(p->d_hc, tmp, nspec, B / 2, nullptr);
                if (hipMemcpyAsync(p->d_hc, tmp, (size_t)nspec * B * 8, hipMemcpyDeviceToDevice, nullptr) != hipSuccess) rc = CAF_ERR_HIP;
                (void)hipStreamSynchronize(nullptr);
                (void)pool_free(tmp);
            }
        }
        if (rc == CAF_OK && p->fused && B != 65536) {
            // the in-LDS engines read their rows in butterfly order (caf_fused.hip, fp_tid_of): permuted once, here
            float2* tmpb = nullptr;
            rc = pool_alloc((void**)&tmpb, nspec * (int64_t)B * 8);
            if (rc == CAF_OK) {
                launch_butterfly_order(p->d_hc, tmpb, nspec * (int64_t)B / 1024, nullptr);
                if (hipMemcpyAsync(p->d_hc, tmpb, (size_t)nspec * B * 8, hipMemcpyDeviceToDevice, nullptr) != hipSuccess) rc = CAF_ERR_HIP;
                (void)hipStreamSynchronize(nullptr);
                (void)pool_free(tmpb);
            }
        }
        hipError_t e = hipStreamSynchronize(nullptr);
        fft_plan_release(&tmp);
        free_tmp();
        if (rc) return rc;
        CAF_HIP_TRY(e);
    }
    if (!shifts.empty())
        CAF_H2D(p->d_shifts, shifts.data(), shifts.size() * 4);
    CAF_H2D(p->d_tscale, tscale.data(), (size_t)T * 4);
    {   // what the zoom needs beside the coarse result: the time-domain multiplier of a product row,
        // rx[d + n] * conj(u[n]), and the frequency each hypothesis index stands for
        std::vector<std::complex<float>> uc((size_t)T * N);
        for (size_t i = 0; i < uc.size(); ++i) uc[i] = d->auto_conj ? std::conj(tm[i]) : tm[i];
        std::vector<double> nu(F);
        for (int f = 0; f < F; ++f)
            nu[f] = d->freq_mode == CAF_FREQ_BINS ? (double)d->h_bins[f] / (double)d->grid : d->h_freqs_norm[f];
        if ((rc = p->alloc(&p->d_uconj, (int64_t)T * N)) || (rc = p->alloc(&p->d_nu, F))) return rc;
        CAF_H2D(p->d_uconj, uc.data(), uc.size() * 8);
        CAF_H2D(p->d_nu, nu.data(), nu.size() * 8);
    }
    CAF_H2D(p->d_gstart, gs.data(), gs.size() * 4);
    CAF_H2D(p->d_glen, gl.data(), gl.size() * 4);

    if ((rc = fft_plan_acquire(&p->fwd, false, (size_t)B, (size_t)p->fwd_chunk, (size_t)B))) return rc;
    if (!p->fused && (rc = fft_plan_acquire(&p->inv, true, (size_t)B, (size_t)nb * T * F, (size_t)p->pitch))) return rc;
    p->workspace_bytes += (int64_t)p->fwd.work_bytes + (int64_t)p->inv.work_bytes;
    const char* aux = getenv("CAF_AUX_STREAM");  // A/B switch, default on
    if (!aux || atoi(aux)) {
        if ((rc = aux_acquire(p->device, &p->s_aux, &p->ev_fork, &p->ev_join))) return rc;
    }
    return CAF_OK;
}

int32_t caf_plan_create(caf_plan* plan, const caf_plan_desc* desc) {
    CAF_REQUIRE(plan && desc, "caf_plan_create: NULL argument");
    *plan = nullptr;
    caf_plan p = new (std::nothrow) caf_plan_t();
    if (!p) {
        set_error("out of host memory");
        return CAF_ERR_NOMEM;
    }
    int32_t rc;
    try {
        rc = plan_build(p, desc);
    } catch (const std::bad_alloc&) {
        set_error("out of host memory while building template spectra");
        rc = CAF_ERR_NOMEM;
    }
    if (rc != CAF_OK) {
        p->release();
        delete p;
        return rc;
    }
    *plan = p;
    return CAF_OK;
}

int32_t caf_plan_info(caf_plan plan, int32_t* block, int32_t* step, int32_t* blocks_per_batch,
                      int64_t* workspace_bytes) {
    CAF_REQUIRE(plan, "NULL plan");
    if (block) *block = plan->B;
    if (step) *step = plan->step;
    if (blocks_per_batch) *blocks_per_batch = plan->nb;
    if (workspace_bytes) *workspace_bytes = plan->workspace_bytes;
    return CAF_OK;
}

int32_t caf_plan_engine(caf_plan plan, int32_t* engine) {
    CAF_REQUIRE(plan && engine, "NULL argument");
    *engine = plan->direct ? CAF_ENGINE_DIRECT
              : plan->persistent ? CAF_ENGINE_PERSISTENT
              : plan->fused      ? CAF_ENGINE_FUSED
                                 : CAF_ENGINE_ROCFFT;
    return CAF_OK;
}

int32_t caf_plan_watchdog(caf_plan plan, int32_t* marks) {
    CAF_REQUIRE(plan && marks, "NULL argument");
    marks[0] = marks[1] = 0;
    if (!plan->persistent || !plan->d_pq) return CAF_OK;
    plan->drain();
    CAF_HIP_TRY(hipDeviceSynchronize());
    CAF_HIP_TRY(hipMemcpy(marks, plan->d_pq + 2, 2 * sizeof(int32_t), hipMemcpyDeviceToHost));
    return CAF_OK;
}

int32_t caf_plan_profile(caf_plan plan, int32_t enable) {
    CAF_REQUIRE(plan, "NULL plan");
    plan->drain();
    plan->prof = enable != 0;
    for (int i = 0; i < CAF_NUM_STAGES; ++i) {
        plan->acc_ms[i] = 0.0;
        plan->acc_n[i] = 0;
    }
    return CAF_OK;
}

int32_t caf_plan_profile_get(caf_plan plan, double* ms, int64_t* launches) {
    CAF_REQUIRE(plan, "NULL plan");
    plan->drain();
    for (int i = 0; i < CAF_NUM_STAGES; ++i) {
        if (ms) ms[i] = plan->acc_ms[i];
        if (launches) launches[i] = plan->acc_n[i];
    }
    return CAF_OK;
}

int32_t caf_plan_execute(caf_plan p, const float* d_rx, int64_t rx_len, int64_t shift_start, int64_t num_shifts,
                         const caf_outputs* out, void* stream) {
    CAF_REQUIRE(out, "caf_plan_execute: NULL argument");
    caf_outputs2 o2;
    std::memset(&o2, 0, sizeof(o2));
    o2.base = *out;
    return caf_plan_execute2(p, d_rx, rx_len, shift_start, num_shifts, &o2, stream);
}

int32_t caf_plan_execute2(caf_plan p, const float* d_rx, int64_t rx_len, int64_t shift_start, int64_t num_shifts,
                          const caf_outputs2* out2, void* stream) {
    CAF_REQUIRE(p && d_rx && out2, "caf_plan_execute: NULL argument");
    CAF_REQUIRE(!out2->reserved[0] && !out2->reserved[1] && !out2->reserved[2], "caf_outputs2.reserved must be NULL");
    // (the engine below works on one flat view of the outputs)
    struct AllOutputs : caf_outputs {
        float* d_surface_t;
    } all;
    static_cast<caf_outputs&>(all) = out2->base;
    all.d_surface_t = out2->d_surface_t;
    const AllOutputs* out = &all;
    CAF_REQUIRE(rx_len >= p->N && rx_len <= p->max_rx, "rx_len outside [template_len, max_rx_len]");
    CAF_REQUIRE(shift_start >= 0 && num_shifts >= 1, "need shift_start >= 0 and num_shifts >= 1");
    CAF_REQUIRE(shift_start + num_shifts - 1 + p->N <= rx_len, "delays run past the end of rx");
    CAF_REQUIRE((reinterpret_cast<uintptr_t>(d_rx) & 7) == 0, "d_rx must be 8-byte aligned");
    int cur_dev = -1;
    CAF_HIP_TRY(hipGetDevice(&cur_dev));
    CAF_REQUIRE(cur_dev == p->device, "caf_plan_execute: the plan was created on another device than the current one");
    hipStream_t st = (hipStream_t)stream;
    const float2* rx = reinterpret_cast<const float2*>(d_rx);
    const int T = p->T, F = p->F;
    const bool want_peak = out->d_peak_val || out->d_peak_delay || out->d_peak_freq;
    // hypothesis-major surface [T][F][S]: with one hypothesis per template it IS the delay-major one
    if (all.d_surface_t && F == 1) {
        CAF_REQUIRE(!all.d_surface, "d_surface and d_surface_t cannot both be given");
        all.d_surface = all.d_surface_t;
        all.d_surface_t = nullptr;
    }
    const bool surf_t = out->d_surface_t != nullptr;
    CAF_REQUIRE(!surf_t || (p->persistent && p->B == 16384 && !out->d_surface && !out->d_cqf),
                "the hypothesis-major surface (d_surface_t) is written by the persistent engine with 16384-point blocks (templates of "
                "at most 8192 samples), and not together with d_surface or d_cqf");

    if (p->direct) {
        CAF_REQUIRE(!out->d_cqf, "the direct engine has no complex-QF output (create the plan with CAF_ENGINE_ROCFFT)");
        p->stage_begin(3, st);
        launch_direct_caf(rx, shift_start, num_shifts, T, F, p->dir_k, p->d_dir_pos, p->d_dir_w, p->d_tscale, out->d_surface,
                          out->d_row_max, out->d_row_arg, want_peak ? p->d_partial : nullptr, p->partial_per_tmpl, st);
        p->stage_end(st);
        if (want_peak) {
            p->stage_begin(6, st);
            launch_peak_reduce(p->d_partial, (num_shifts + 127) / 128, p->partial_per_tmpl, T,
                               p->d_partial + (int64_t)T * p->partial_per_tmpl, out->d_peak_val, out->d_peak_delay,
                               out->d_peak_freq, st);
            p->stage_end(st);
        }
        CAF_HIP_TRY(hipGetLastError());
        return CAF_OK;
    }

    const int64_t nblk = (num_shifts + p->step - 1) / p->step;
    const int64_t nblk_pad = (nblk + p->nb - 1) / p->nb * p->nb;
    // overlap-save blocks of rx -> spectra X[b] for every block of this call
    // (partitioned templates: block b's products also take the spectra of blocks b + 1 .. b + npart - 1)
    const int64_t nfwd = ((p->fused ? nblk + p->npart - 1 : nblk_pad) + p->fwd_chunk - 1) / p->fwd_chunk;
    // LDS engines with 16384-point blocks: gather + forward transform in one launch of the in-LDS FFT, which also
    // writes the sliding energies of each block's delays from the samples it holds anyway.
    // (CAF_FWD_ROCFFT=1: the gather kernel + batched rocFFT transforms that every other block size uses;
    //  CAF_ENERGY_PREFIX=1: the separate float64-prefix pass for the energies)
    static const bool fwd_rocfft = [] {
        const char* e = getenv("CAF_FWD_ROCFFT");
        return e && atoi(e);
    }();
    static const bool energy_prefix = [] {
        const char* e = getenv("CAF_ENERGY_PREFIX");
        return e && atoi(e);
    }();
    const bool lds_fwd = p->fused && p->B == 16384 && !fwd_rocfft;
    const bool energy_in_fwd = lds_fwd && !energy_prefix;
    // Otherwise the sliding-energy pass (float64 prefix of |rx|^2, then differences) is independent of the block
    // spectra: it runs on the plan's auxiliary stream beside gather + forward FFTs (small memory-bound kernels that do
    // not fill the chip one at a time) and is joined before the first consumer of inv_e.
    const bool aux = p->s_aux && !energy_in_fwd;
    if (!energy_in_fwd) {
        hipStream_t se = aux ? p->s_aux : st;
        if (aux) {
            CAF_HIP_TRY(hipEventRecord(p->ev_fork, st));
            CAF_HIP_TRY(hipStreamWaitEvent(p->s_aux, p->ev_fork, 0));
        }
        p->stage_begin(0, se);
        launch_energy_prefix(rx, rx_len, p->d_tile_sums, p->d_prefix, se);
        launch_inv_energy(rx, rx_len, p->d_prefix, shift_start, num_shifts, p->d_gstart, p->d_glen, p->G, p->d_inv_e, se);
        p->stage_end(se);
        if (aux) CAF_HIP_TRY(hipEventRecord(p->ev_join, p->s_aux));
    }
    const bool lds_fwd32 = p->fused && p->B == 32768 && !fwd_rocfft;
    const bool lds_fwd64 = p->fused && p->B == 65536 && !fwd_rocfft;
    if (lds_fwd32 || lds_fwd64) {
        p->stage_begin(2, st);
        const int rc = lds_fwd32 ? launch_block_spectra32(rx, rx_len, shift_start, p->step, nfwd * p->fwd_chunk, p->d_xb2, st)
                                 : launch_block_spectra64(rx, rx_len, shift_start, p->step, nfwd * p->fwd_chunk, p->d_xb2, st);
        p->stage_end(st);
        if (rc) return rc;
    } else if (lds_fwd) {
        p->stage_begin(2, st);
        const int rc = energy_in_fwd
                           ? launch_block_spectra(rx, rx_len, shift_start, p->step, nfwd * p->fwd_chunk, p->d_xb, st, p->d_inv_e,
                                                  num_shifts, p->d_gstart, p->d_glen, p->G)
                           : launch_block_spectra(rx, rx_len, shift_start, p->step, nfwd * p->fwd_chunk, p->d_xb, st);
        p->stage_end(st);
        if (rc) return rc;
    } else {
        p->stage_begin(1, st);
        launch_gather_blocks(rx, rx_len, shift_start, p->step, p->B, (int32_t)(nfwd * p->fwd_chunk), p->d_xb, st);
        p->stage_end(st);
        for (int64_t c = 0; c < nfwd; ++c) {
            p->stage_begin(2, st);
            int rc = p->fwd.exec(p->d_xb + c * p->fwd_chunk * (int64_t)p->B, nullptr, st);
            p->stage_end(st);
            if (rc) return rc;
        }
    }
    if (p->fused && p->B == 32768 && !lds_fwd32)  // block spectra parity-major for the two chained half-transforms
        launch_parity_major(p->d_xb, p->d_xb2, nfwd * p->fwd_chunk, p->B / 2, st, true);
    if (p->fused && p->B == 65536 && !lds_fwd64)  // ... as pairs of the two halves of each parity for the folded form
        launch_parity_pairs(p->d_xb, p->d_xb2, nfwd * p->fwd_chunk, p->B / 4, st);
    if (aux) CAF_HIP_TRY(hipStreamWaitEvent(st, p->ev_join, 0));
    bool f1_direct = false, f1_item_peaks = false;
    if (p->fused) {
        // complex QF rows: written by the FFT items of the one-launch engine themselves (fused_item MODE 4), as the only
        // output of the call
        const bool cqf_rows = out->d_cqf != nullptr;
        CAF_REQUIRE(!cqf_rows || (p->persistent && !out->d_surface && !out->d_row_max && !out->d_row_arg && !want_peak),
                    "the in-LDS engines write the complex-QF plane only from the persistent engine and only as the "
                    "sole output of a call (create the plan with CAF_ENGINE_ROCFFT for the other combinations)");
        // No surface wanted (per-delay traces / peaks only): the FFT items keep running per-delay maxima and
        // write one (value, hypothesis) pair per delay and group instead of the |y|^2 tiles (1/32 of the
        // bytes at 64 hypotheses per group).  Needs groups that do not straddle templates: with at least one
        // group per template they are formed per template, evenly sized (F = 201: 4 groups of 51/51/51/48).
        // The pairs live in the tile buffer: vmax [block][group][tile][64] f32, then imax (same shape, i32).
        int ns_gpt = 0;
        {
            const char* e = getenv("CAF_PERSIST_NOSURF");  // A/B switch, default on
            if (p->persistent && p->B == 16384 && !out->d_surface && !cqf_rows && F >= p->hyp_per_wg && (!e || atoi(e))) {
                const int gpt = (F + p->hyp_per_wg - 1) / p->hyp_per_wg;
                // the two pair arrays must fit the tile buffer they replace (true for >= 2 hypotheses per group)
                if (2 * (int64_t)T * gpt <= (int64_t)T * F) ns_gpt = gpt;
            }
            // A work item costs ~13 us beside its hypotheses (claim, parameters, block spectrum, first row:
            // profiles/r04/c3_items_and_xcd_groups.log), 3 % of an item of 64: groups of up to 256 hypotheses (the running
            // maxima pack the index into 8 bits) while the launch keeps >= 40 items per CU -- config C4's share of one GPU
            // (64 templates x 512 bins) 1010 -> 979 ms; C2 (one template, 1365 x 4 items) stays at 64
            if (ns_gpt && !getenv("CAF_HYP_PER_WG")) {
                const int64_t blocks_launch = std::min<int64_t>(p->nb_nosurf, nblk);
                while (ns_gpt > 1) {
                    const int g2 = (ns_gpt + 1) / 2;
                    if ((F + g2 - 1) / g2 > 256 || (int64_t)T * g2 * blocks_launch < (int64_t)40 * p->n_cus) break;
                    ns_gpt = g2;
                }
            }
            // the hypothesis-major surface rides on the same per-template groups (F >= 2 here; fewer hypotheses than a
            // group: one group per template)
            if (surf_t) ns_gpt = (F + p->hyp_per_wg - 1) / p->hyp_per_wg;
        }
        // the |y|^2 tiles of one block are addressed with 32-bit byte offsets (descriptor + SGPR + VGPR offset)
        CAF_REQUIRE(ns_gpt || (int64_t)(p->B / 64) * T * F * 256 < ((int64_t)1 << 32),
                    "too many hypotheses (templates x frequencies) for one launch with |y|^2 tiles: ask for no surface, "
                    "or split the templates over several plans");
        // No frequency scan (F == 1) with per-delay rows wanted: the FFT items write the finished rows themselves
        // (fused_item MODE 3) and there are no tile items at all; row_arg is all zeros and the peaks come from the rows.
        static const bool f1_env = [] {
            const char* e = getenv("CAF_PERSIST_F1DIRECT");  // A/B switch, default on
            return !e || atoi(e);
        }();
        // (CAF_F1_ITEM_PEAKS=0: the peak records from a pass over the finished rows instead of from the items -- A/B switch)
        static const bool f1_pk_env = [] {
            const char* e = getenv("CAF_F1_ITEM_PEAKS");
            return !e || atoi(e);
        }();
        f1_direct = p->persistent && p->B == 16384 && F == 1 && f1_env && !cqf_rows &&
                    (out->d_row_max || out->d_surface || (want_peak && f1_pk_env));
        f1_item_peaks = f1_direct && want_peak && f1_pk_env;
        const int nb_launch = ns_gpt ? p->nb_nosurf : p->nb;  // blocks per launch
        for (int64_t b0 = 0; p->persistent && b0 < nblk; b0 += nb_launch) {
            const int32_t nbk = (int32_t)std::min<int64_t>(nb_launch, nblk - b0);
            PersistParams h;
            std::memset(&h, 0, sizeof(h));
            h.xb = (p->B >= 32768 ? p->d_xb2 : p->d_xb) + b0 * (int64_t)p->B;
            h.hc = p->d_hc;
            h.shifts = p->d_shifts;
            h.tw1 = p->d_tw1;
            h.tw23 = p->d_tw23;
            h.vt = p->d_vt;
            h.table_mode = p->mul_mode == 2 ? 1 : 0;
            h.nfreq = F;
            h.nhyp = T * F;
            h.hyp_per_wg = p->hyp_per_wg;
            h.nblk = nbk;
            h.tiles_per_blk = p->tiles_per_blk;
            h.block_log2 = p->B == 65536 ? 16 : p->B == 32768 ? 15 : 14;
            h.dstride = p->B == 65536 ? 2 : 1;
            h.npart = p->npart;
            h.ntmpl = T;
            h.step = p->step;
            h.blk0 = (int32_t)b0;
            h.tscale = p->d_tscale;
            h.inv_e = p->d_inv_e;
            h.num_shifts = num_shifts;
            h.shift_start = shift_start;
            h.surface = out->d_surface;
            h.row_max = out->d_row_max;
            h.row_arg = out->d_row_arg;
            h.partial = want_peak ? p->d_partial : nullptr;  // (f1_direct: set only if the items write the records)
            h.partial_per_tmpl = p->partial_per_tmpl;
            h.pq = p->d_pq;
            h.tr_slots = p->tr_slots;
            h.ngroups = (T * F + p->hyp_per_wg - 1) / p->hyp_per_wg;
            h.surface_t = out->d_surface_t;
            if (ns_gpt) {
                h.nosurf = surf_t ? 2 : 1;
                h.gpt = ns_gpt;
                h.hyp_per_wg = (F + ns_gpt - 1) / ns_gpt;
                h.ngroups = T * ns_gpt;
            }
            h.vmax = p->d_vt;
            h.imax = reinterpret_cast<int32_t*>(p->d_vt + (int64_t)nb_launch * h.ngroups * p->tiles_per_blk * 64);
            if (h.block_log2 == 16) h.ngroups *= 2;  // one work item per (hypothesis group, output residue): fused_item2q<FOLD>
            h.n_fft = nbk * h.ngroups;
            h.ipb = (p->tiles_per_blk + 15) / 16;  // 16 tiles per item (PQ_TILES, caf_fused.hip)
            h.n_tr = nbk * h.ipb;
            if (f1_direct) {
                h.f1_direct = 1;
                h.n_tr = 0;
                if (!f1_item_peaks) h.partial = nullptr;
            }
            if (cqf_rows) {
                h.cqf = out->d_cqf;
                h.n_tr = 0;
            }
            // both stages are one kernel: its time is booked on the multiply/FFT stage
            int32_t* h_dbg = nullptr;
            // CAF_PERSIST_DEBUG=1: role statistics of every launch; =2: of every 10th launch only, so that the nine before it
            // run back to back with the plain kernel (the report waits for the launch, which idles the GPU)
            static int dbg_launches = 0;
            const char* dbg_env = getenv("CAF_PERSIST_DEBUG");
            if (dbg_env && (atoi(dbg_env) < 2 || ++dbg_launches % 10 == 0)) {  // host-mapped, 8 ints per workgroup
                (void)hipHostMalloc((void**)&h_dbg, sizeof(int32_t) * 8 * (size_t)p->n_cus, hipHostMallocMapped);
                std::memset(h_dbg, 0, sizeof(int32_t) * 8 * (size_t)p->n_cus);
                (void)hipHostGetDevicePointer((void**)&h.dbg, h_dbg, 0);
            }
            p->stage_begin(3, st);
            launch_caf_persistent(&h, p->d_params, p->n_cus, st);
            p->stage_end(st);
            if (h_dbg) {
                // wait (bounded) and report what the workgroups did: time per role in us (100 MHz device clock)
                hipEvent_t ev;
                (void)hipEventCreate(&ev);
                (void)hipEventRecord(ev, st);
                bool done = false;
                for (int i = 0; i < 80 && !done; ++i) {
                    done = hipEventQuery(ev) == hipSuccess;
                    if (!done) usleep(100000);
                }
                if (!done) {
                    fprintf(stderr, "[caf persistent] kernel still running after 8 s: aborting the process\n");
                    _exit(3);
                }
                std::vector<int32_t> q(4 + nbk);
                (void)hipMemcpy(q.data(), p->d_pq, q.size() * 4, hipMemcpyDeviceToHost);
                double sum[2][5] = {{0}};
                int cnt[2] = {0, 0};
                for (int w = 0; w < p->n_cus; ++w) {
                    const int pref = ((w >> 3) & 31) >= 32 - p->tr_slots;
                    ++cnt[pref];
                    for (int k = 0; k < 5; ++k) sum[pref][k] += h_dbg[8 * w + k];
                }
                uint32_t t0 = 0xffffffffu, t1 = 0, tl = 0;
                for (int w = 0; w < p->n_cus; ++w) {
                    t0 = std::min(t0, (uint32_t)h_dbg[8 * w + 5]);
                    tl = std::max(tl, (uint32_t)h_dbg[8 * w + 5]);
                    t1 = std::max(t1, (uint32_t)h_dbg[8 * w + 6]);
                }
                fprintf(stderr, "[caf persistent] n_fft=%d n_tr=%d | fft_next=%d tr_next=%d watchdog=%d,%d | first start -> last end "
                        "%.0f us, last start +%.0f us\n", h.n_fft, h.n_tr, q[0], q[1], q[2], q[3], (t1 - t0) / 100.0, (tl - t0) / 100.0);
                for (int r = 0; r < 2; ++r)
                    if (cnt[r])
                        fprintf(stderr,
                                "  %s workgroups (%d): claim+wait %.0f us, fft %.0f us in %.1f items (%.1f us each), tiles %.0f us "
                                "in %.1f items (%.1f us each)\n",
                                r ? "tile-first" : "fft-first", cnt[r], sum[r][0] / cnt[r] / 100.0, sum[r][1] / cnt[r] / 100.0,
                                sum[r][3] / cnt[r], sum[r][3] > 0 ? sum[r][1] / sum[r][3] / 100.0 : 0.0,
                                sum[r][2] / cnt[r] / 100.0, sum[r][4] / cnt[r],
                                sum[r][4] > 0 ? sum[r][2] / sum[r][4] / 100.0 : 0.0);
                // CAF_PERSIST_TILE_ONLY="16,64,256": re-run only the tile role on k workgroups over the tiles just
                // produced (per-CU streaming rate of the role at different levels of HBM concurrency)
                if (const char* lst = getenv("CAF_PERSIST_TILE_ONLY")) {
                    PersistParams h2 = h;
                    h2.n_fft = 0;
                    h2.ngroups = 0;  // every block counts as published
                    h2.dbg = nullptr;
                    hipEvent_t e0, e1;
                    (void)hipEventCreate(&e0);
                    (void)hipEventCreate(&e1);
                    for (const char* c = lst; *c;) {
                        const int k = atoi(c);
                        while (*c && *c != ',') ++c;
                        if (*c == ',') ++c;
                        if (k < 1) continue;
                        (void)hipEventRecord(e0, st);
                        launch_caf_persistent(&h2, p->d_params, k, st);
                        (void)hipEventRecord(e1, st);
                        (void)hipEventSynchronize(e1);
                        float ms = 0.f;
                        (void)hipEventElapsedTime(&ms, e0, e1);
                        const double bytes = (double)h2.n_tr * 16.0 * 64.0 * F * 4.0 * (out->d_surface ? 2.0 : 1.0);
                        fprintf(stderr, "  tile role alone on %3d workgroups: %.2f ms, %.1f us per item, %.1f GB/s per CU, %.2f TB/s\n",
                                k, ms, ms * 1e3 * k / h2.n_tr, bytes / (ms * 1e-3) / k / 1e9, bytes / (ms * 1e-3) / 1e12);
                    }
                }
                (void)hipHostFree(h_dbg);
            }
        }
        for (int64_t b0 = 0; !p->persistent && b0 < nblk; b0 += p->nb) {
            const int32_t nbk = (int32_t)std::min<int64_t>(p->nb, nblk - b0);
            p->stage_begin(3, st);
            launch_fused_caf(p->d_xb + b0 * (int64_t)p->B, p->d_hc, p->d_shifts, p->d_tw1, p->d_tw23,
                             p->mul_mode == 2 ? 1 : 0, F, T * F, p->hyp_per_wg, nbk, p->tiles_per_blk, p->d_vt, st);
            p->stage_end(st);
            p->stage_begin(5, st);
            launch_transpose_norm_argmax(p->d_vt, T, F, p->d_tscale, p->d_inv_e, num_shifts, shift_start, p->step,
                                         (int32_t)b0, nbk, p->tiles_per_blk, out->d_surface, out->d_row_max,
                                         out->d_row_arg, want_peak ? p->d_partial : nullptr, p->partial_per_tmpl, st);
            p->stage_end(st);
        }
    }
    for (int64_t b0 = 0; !p->fused && b0 < nblk; b0 += p->nb) {
        int rc;
        p->stage_begin(3, st);
        launch_spectral_mul(p->mul_mode, p->d_xb + b0 * (int64_t)p->B, p->d_hc, p->d_shifts, p->B, p->pitch, F, T * F, p->hyp_per_wg, p->nb,
                            p->d_pbuf, st);
        p->stage_end(st);

        p->stage_begin(4, st);
        rc = p->inv.exec(p->d_pbuf, nullptr, st);
        p->stage_end(st);
        if (rc) return rc;

        if (out->d_cqf)
            launch_complex_norm(p->d_pbuf, p->pitch, F, p->d_tscale, p->d_inv_e, num_shifts, p->step, (int32_t)b0, p->nb,
                                T * F, reinterpret_cast<float2*>(out->d_cqf), st);
        const bool want_mag = out->d_surface || out->d_row_max || out->d_row_arg || want_peak;
        p->stage_begin(5, st);
        if (want_mag)
            launch_magsq(p->d_pbuf, p->pitch, T, F, p->d_tscale, p->d_inv_e, num_shifts, shift_start, p->step, (int32_t)b0,
                     p->nb, p->tiles_per_blk, out->d_surface, out->d_row_max, out->d_row_arg,
                     want_peak ? p->d_partial : nullptr, p->partial_per_tmpl, st);
        p->stage_end(st);
    }
    // one hypothesis per template: its index is 0 everywhere.  (On this stream, after the launch: a fill running beside
    // the persistent kernel on another stream takes CUs from its resident workgroups -- measured 0.6 ms slower.)
    if (f1_direct && out->d_row_arg) CAF_HIP_TRY(hipMemsetAsync(out->d_row_arg, 0, (size_t)T * (size_t)num_shifts * 4, st));
    if (want_peak) {
        p->stage_begin(6, st);
        int64_t nrec = (p->fused ? nblk : nblk_pad) * p->tiles_per_blk;  // only the records of the blocks touched by this call
        if (f1_direct && f1_item_peaks) {
            nrec = nblk * 16;  // one record per (template, block, wave), written by the FFT items (fused_item PK)
        } else if (f1_direct) {
            launch_rows_peak(out->d_row_max ? out->d_row_max : out->d_surface, T, num_shifts, shift_start, p->d_partial,
                             p->partial_per_tmpl, st);
            nrec = rows_peak_chunks(num_shifts);
        }
        launch_peak_reduce(p->d_partial, nrec, p->partial_per_tmpl, T, p->d_partial + (int64_t)T * p->partial_per_tmpl,
                           out->d_peak_val, out->d_peak_delay, out->d_peak_freq, st);
        p->stage_end(st);
    }
    CAF_HIP_TRY(hipGetLastError());
    return CAF_OK;
}

int32_t caf_plan_execute_host(caf_plan p, const float* h_rx, int64_t rx_len, int64_t shift_start, int64_t num_shifts,
                              float* h_surface, float* h_row_max, int32_t* h_row_arg, float* h_peak_val,
                              int32_t* h_peak_delay, int32_t* h_peak_freq) {
    CAF_REQUIRE(p && h_rx, "caf_plan_execute_host: NULL argument");
    CAF_REQUIRE(num_shifts >= 1, "num_shifts must be >= 1");
    const int T = p->T, F = p->F;
    float2* d_rx = nullptr;
    caf_outputs2 o2;
    std::memset(&o2, 0, sizeof(o2));
    caf_outputs& o = o2.base;
    // The surface goes to the host anyway: where the engine can write it hypothesis-major itself (persistent engine, 16384-point
    // blocks: no |y|^2 tiles, no tile role -- 10.5 instead of 13.2 ms at config C2) that launch runs, and the transposition to the
    // reference's (delays, frequencies) layout happens in the download (host_d2h_transposed): the same numbers bit for bit
    // (tests/test_gpu_fullsize.py::test_c2_hypothesis_major_surface).  CAF_HOST_SURFACE_DELAY_MAJOR=1: the delay-major launch (A/B).
    static const bool host_delay_major = [] {
        const char* e = getenv("CAF_HOST_SURFACE_DELAY_MAJOR");
        return e && atoi(e);
    }();
    const bool surf_t = h_surface && p->persistent && p->B == 16384 && F > 1 && F <= 65536 && !host_delay_major;
    std::vector<void*> owned;
    auto dalloc = [&](void** ptr, int64_t bytes) -> int {
        const int prc = pool_alloc(ptr, std::max<int64_t>(bytes, 16));  // cached across calls (caf_pool.hip)
        if (prc) return prc;
        owned.push_back(*ptr);
        return CAF_OK;
    };
    int rc = dalloc((void**)&d_rx, rx_len * 8);
    if (!rc && h_surface) rc = dalloc((void**)(surf_t ? &o2.d_surface_t : &o.d_surface), (int64_t)T * num_shifts * F * 4);
    if (!rc && h_row_max) rc = dalloc((void**)&o.d_row_max, (int64_t)T * num_shifts * 4);
    if (!rc && h_row_arg) rc = dalloc((void**)&o.d_row_arg, (int64_t)T * num_shifts * 4);
    if (!rc && h_peak_val) rc = dalloc((void**)&o.d_peak_val, T * 4);
    if (!rc && h_peak_delay) rc = dalloc((void**)&o.d_peak_delay, T * 4);
    if (!rc && h_peak_freq) rc = dalloc((void**)&o.d_peak_freq, T * 4);
    auto cleanup = [&]() {
        for (void* q : owned) (void)pool_free(q);
    };
    if (rc) {
        cleanup();
        return rc;
    }
    // (host arrays travel through the library's pinned staging lanes, never as pinned user pages: caf_host.cpp)
    rc = host_h2d(d_rx, h_rx, rx_len * 8, nullptr);
    if (!rc) rc = caf_plan_execute2(p, reinterpret_cast<const float*>(d_rx), rx_len, shift_start, num_shifts, &o2, nullptr);
    if (rc) {
        cleanup();
        return rc;
    }
    hipError_t e = hipSuccess;
    auto back = [&](void* h, const void* dptr, int64_t bytes) {
        if (!rc && h) rc = host_d2h(h, dptr, bytes, nullptr);
    };
    if (surf_t) {
        for (int t = 0; t < T && !rc; ++t)
            rc = host_d2h_transposed(h_surface + (int64_t)t * num_shifts * F, false, o2.d_surface_t + (int64_t)t * F * num_shifts, F,
                                     num_shifts, 0, num_shifts, nullptr);
    } else {
        back(h_surface, o.d_surface, (int64_t)T * num_shifts * F * 4);
    }
    back(h_row_max, o.d_row_max, (int64_t)T * num_shifts * 4);
    back(h_row_arg, o.d_row_arg, (int64_t)T * num_shifts * 4);
    back(h_peak_val, o.d_peak_val, T * 4);
    back(h_peak_delay, o.d_peak_delay, T * 4);
    back(h_peak_freq, o.d_peak_freq, T * 4);
    cleanup();
    CAF_HIP_TRY(e);
    return rc;
}

}  // extern "C"
