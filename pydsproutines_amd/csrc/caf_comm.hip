// Multi-GPU exchange of the path behind the C-ABI: one process per GPU, templates block-sharded, and the
// ONLY collective is the all-gather of the per-template peak rows (int32 delay, int32 frequency index, float32
// |peak|^2 as bits).  RCCL (ncclAllGather over xGMI) directly, so that a reference-side binder of include/caf.h has
// the multi-GPU step without torch; bench.py and the Python host use torch.distributed (backend "nccl" == RCCL) for
// the same exchange.  The reference itself has no multi-GPU code (its nearest analogue is the thread-strided split
// of IppXcorrFFT.cpp:117), so nothing here mirrors an upstream call pattern.
// RCCL is bound at first use (dlopen), not at link time: libcaf.so then loads on hosts without RCCL, and a process
// that also runs torch.distributed keeps the single RCCL its torch build brought along (same SONAME).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <new>

#include "caf_internal.h"

struct caf_comm_t {
    ncclComm_t comm = nullptr;
    int world = 0, rank = 0, device = 0;
};

namespace {
struct Rccl {
    decltype(&::ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&::ncclCommInitRank) CommInitRank = nullptr;
    decltype(&::ncclCommDestroy) CommDestroy = nullptr;
    decltype(&::ncclAllGather) AllGather = nullptr;
    decltype(&::ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};
const Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // RTLD_LOCAL on purpose: a global RCCL (and its librocm_smi64) interposes the amd::smi statics that
        // libamd_smi repeats, and both then run their destructors over one object at exit ("double free").
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(h, "ncclAllGather"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.GetErrorString;
    });
    return r.ok ? &r : nullptr;
}
}  // namespace
#define CAF_NEED_RCCL()                                                              \
    const Rccl* R = rccl();                                                          \
    if (!R) {                                                                        \
        caf::set_error("RCCL (librccl.so.1) could not be loaded: " + std::string(dlerror() ? dlerror() : "missing symbols")); \
        return CAF_ERR_HIP;                                                          \
    }

#define CAF_NCCL_TRY(expr)                                                                       \
    do {                                                                                         \
        ncclResult_t _r = (expr);                                                                \
        if (_r != ncclSuccess) {                                                                 \
            caf::set_error(std::string(#expr) + ": " + R->GetErrorString(_r));                  \
            return CAF_ERR_HIP;                                                                  \
        }                                                                                        \
    } while (0)

extern "C" {

int32_t caf_comm_unique_id(void* id128) {
    CAF_REQUIRE(id128, "caf_comm_unique_id: NULL");
    CAF_NEED_RCCL();
    static_assert(sizeof(ncclUniqueId) == 128, "the id crosses the ABI as 128 opaque bytes");
    ncclUniqueId id;
    CAF_NCCL_TRY(R->GetUniqueId(&id));
    std::memcpy(id128, &id, sizeof(id));
    return CAF_OK;
}

int32_t caf_comm_create(caf_comm* comm, int32_t world, int32_t rank, const void* id128) {
    CAF_REQUIRE(comm && id128 && world >= 1 && rank >= 0 && rank < world, "caf_comm_create: bad arguments");
    *comm = nullptr;
    CAF_NEED_RCCL();
    caf_comm c = new (std::nothrow) caf_comm_t();
    CAF_REQUIRE(c, "out of host memory");
    hipError_t e = hipGetDevice(&c->device);
    if (e != hipSuccess) {
        delete c;
        CAF_HIP_TRY(e);
    }
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    const ncclResult_t r = R->CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        delete c;
        caf::set_error(std::string("ncclCommInitRank: ") + R->GetErrorString(r));
        return CAF_ERR_HIP;
    }
    c->world = world;
    c->rank = rank;
    *comm = c;
    return CAF_OK;
}

int32_t caf_comm_destroy(caf_comm comm) {
    if (!comm) return CAF_OK;
    if (comm->comm)
        if (const Rccl* R = rccl()) (void)R->CommDestroy(comm->comm);
    delete comm;
    return CAF_OK;
}

int32_t caf_peak_table_allgather(caf_comm comm, const int32_t* d_local, int32_t rows_per_rank, int32_t* d_table,
                                 void* stream) {
    CAF_REQUIRE(comm && d_local && d_table && rows_per_rank >= 1, "caf_peak_table_allgather: bad arguments");
    int cur = -1;
    CAF_HIP_TRY(hipGetDevice(&cur));
    CAF_REQUIRE(cur == comm->device, "caf_peak_table_allgather: the communicator belongs to another device");
    CAF_NEED_RCCL();
    CAF_NCCL_TRY(R->AllGather(d_local, d_table, (size_t)3 * rows_per_rank, ncclInt32, comm->comm, (hipStream_t)stream));
    return CAF_OK;
}

}  // extern "C"
