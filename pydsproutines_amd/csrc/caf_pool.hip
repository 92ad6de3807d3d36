// Caching device allocator behind caf_malloc / caf_free and the library's own scratch buffers.
//
// The reference leans on cupy's memory pool: every cp.empty / cp.zeros inside its wrappers is a pool hit, not a
// cudaMalloc.  hipMalloc + hipFree cost 100-300 us per pair and hipFree synchronises the device, which is more than
// most of the kernel-level entry points take, so freed blocks are kept per (device, rounded size) and handed out
// again.  Reuse is safe in stream order: a block freed while a kernel on stream S still uses it is only ever touched
// again by later work, so callers that stay on one stream (the host layer uses the null stream) need no
// synchronisation; callers mixing streams synchronise before freeing, as with any stream-ordered pool.
//
//   CAF_POOL_MB   upper bound of cached (free) bytes per process, default 8192; 0 disables caching.
#include <cstdlib>
#include <map>
#include <mutex>
#include <unordered_map>

#include "caf_internal.h"

namespace caf {
namespace {

struct Live {
    int64_t bytes;
    int dev;
};

std::mutex g_mu;
struct Cached {
    void* p;
    uint64_t seq;  // when it was freed (larger = more recent)
};
std::multimap<std::pair<int, int64_t>, Cached> g_free;  // (device, rounded bytes) -> block
uint64_t g_seq = 0;
std::unordered_map<void*, Live> g_live;                // blocks handed out
int64_t g_cached = 0, g_in_use = 0, g_hits = 0, g_misses = 0;

int64_t pool_limit() {
    static const int64_t lim = [] {
        const char* e = std::getenv("CAF_POOL_MB");
        const int64_t mb = e ? std::atoll(e) : 8192;
        return (mb < 0 ? 0 : mb) << 20;
    }();
    return lim;
}

// <= 1 MiB: next power of two (>= 512 B); above: next multiple of 1 MiB
int64_t round_size(int64_t bytes) {
    if (bytes <= 512) return 512;
    if (bytes <= ((int64_t)1 << 20)) {
        int64_t r = 512;
        while (r < bytes) r <<= 1;
        return r;
    }
    return (bytes + ((int64_t)1 << 20) - 1) & ~(((int64_t)1 << 20) - 1);
}

// caller holds g_mu; hipFree of every cached block of `dev` (or of all devices when dev < 0)
void trim_locked(int dev) {
    for (auto it = g_free.begin(); it != g_free.end();) {
        if (dev < 0 || it->first.first == dev) {
            (void)hipFree(it->second.p);
            g_cached -= it->first.second;
            it = g_free.erase(it);
        } else {
            ++it;
        }
    }
}

}  // namespace

int pool_alloc(void** out, int64_t bytes) {
    int dev = 0;
    CAF_HIP_TRY(hipGetDevice(&dev));
    const int64_t r = round_size(bytes);
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_free.find({dev, r});
    void* p = nullptr;
    if (it != g_free.end()) {
        p = it->second.p;
        g_free.erase(it);
        g_cached -= r;
        ++g_hits;
    } else {
        ++g_misses;
        hipError_t e = hipMalloc(&p, (size_t)r);
        if (e == hipErrorOutOfMemory) {  // give the cache back to the driver and try once more
            (void)hipGetLastError();
            trim_locked(dev);
            e = hipMalloc(&p, (size_t)r);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            set_error(std::string("hipMalloc(") + std::to_string(r) + " bytes): " + hipGetErrorString(e));
            return e == hipErrorOutOfMemory ? CAF_ERR_NOMEM : CAF_ERR_HIP;
        }
    }
    g_live[p] = Live{r, dev};
    g_in_use += r;
    *out = p;
    return CAF_OK;
}

int pool_free(void* p) {
    if (!p) return CAF_OK;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_live.find(p);
    if (it == g_live.end()) {  // not ours (allocated before the pool existed / by the caller): plain free
        CAF_HIP_TRY(hipFree(p));
        return CAF_OK;
    }
    const Live l = it->second;
    g_live.erase(it);
    g_in_use -= l.bytes;
    if (l.bytes > pool_limit()) {  // (larger than the whole cache: straight back to the driver)
        CAF_HIP_TRY(hipFree(p));
        return CAF_OK;
    }
    // A full cache gives up its OLDEST blocks for the one that has just been in use -- kept the other way round (the new block
    // dropped), a process whose cache had filled with sizes it no longer asks for paid a hipMalloc + hipFree pair (10-20 ms
    // each at 160 MB, and a device synchronisation) for every scratch buffer of every later call.
    while (g_cached + l.bytes > pool_limit() && !g_free.empty()) {
        auto old = g_free.begin();
        for (auto it = g_free.begin(); it != g_free.end(); ++it)
            if (it->second.seq < old->second.seq) old = it;
        (void)hipFree(old->second.p);
        g_cached -= old->first.second;
        g_free.erase(old);
    }
    g_free.emplace(std::make_pair(l.dev, l.bytes), Cached{p, ++g_seq});
    g_cached += l.bytes;
    return CAF_OK;
}

void pool_trim() {
    std::lock_guard<std::mutex> lk(g_mu);
    trim_locked(-1);
}

void pool_stats(int64_t* cached, int64_t* in_use, int64_t* hits, int64_t* misses) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (cached) *cached = g_cached;
    if (in_use) *in_use = g_in_use;
    if (hits) *hits = g_hits;
    if (misses) *misses = g_misses;
}

}  // namespace caf
