// Multi-GPU collective of the path through the C-ABI: the ONE data-path exchange of the row-sharded dense path (a sum
// all-reduce of the D*D + D doubles of [H | g]) and of the CG route (the D-vector, or K x D block, of partial HVPs).
// The Python mirror uses torch.distributed (backend "nccl" = RCCL); a C / C++ host has no torch, so these entry
// points bind RCCL directly.  librccl.so is loaded lazily (dlopen) at the first call: hosts that never shard do not
// need it, and a box without it gets VT_ERR_UNSUPPORTED instead of a load failure of the whole library.
// No counterpart in the reference (single-process NumPy); SURVEY.md section 8b / 8e name the hook (`vt_set_comm`).
#include "vt_common.h"
#include <dlfcn.h>
#include <cstring>
#include <mutex>

namespace {

// the few RCCL declarations needed (rccl.h: NCCL 2 API), kept local so that the build does not depend on the header
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;                       // ncclSuccess = 0
constexpr int kNcclFloat64 = 8, kNcclSum = 0;   // ncclDataType_t::ncclFloat64, ncclRedOp_t::ncclSum

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    bool ok = false;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names) {
            r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) return;
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
        r.AllReduce = (decltype(r.AllReduce))dlsym(r.handle, "ncclAllReduce");
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce;
    });
    return r;
}

}  // namespace

extern "C" int vt_comm_unique_id(void* id128) {
    if (!id128) return VT_ERR_BAD_ARG;
    Rccl& r = rccl();
    if (!r.ok) return VT_ERR_UNSUPPORTED;
    ncclUniqueId id;
    if (r.GetUniqueId(&id) != 0) return VT_ERR_HIP;
    memcpy(id128, &id, sizeof(id));
    return VT_OK;
}

extern "C" int vt_comm_init(void** comm, int nranks, int rank, const void* id128) {
    if (!comm || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return VT_ERR_BAD_ARG;
    Rccl& r = rccl();
    if (!r.ok) return VT_ERR_UNSUPPORTED;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    if (r.CommInitRank(&c, nranks, id, rank) != 0) return VT_ERR_HIP;   // binds the CURRENT device
    *comm = (void*)c;
    return VT_OK;
}

extern "C" int vt_comm_destroy(void* comm) {
    if (!comm) return VT_ERR_BAD_ARG;
    Rccl& r = rccl();
    if (!r.ok) return VT_ERR_UNSUPPORTED;
    return r.CommDestroy((ncclComm_t)comm) == 0 ? VT_OK : VT_ERR_HIP;
}

extern "C" int vt_allreduce_sum(void* comm, double* buf, int64_t count, void* stream) {
    if (!comm || count < 0 || (count > 0 && !buf)) return VT_ERR_BAD_ARG;
    if (count == 0) return VT_OK;
    Rccl& r = rccl();
    if (!r.ok) return VT_ERR_UNSUPPORTED;
    return r.AllReduce(buf, buf, (size_t)count, kNcclFloat64, kNcclSum, (ncclComm_t)comm, (hipStream_t)stream) == 0
               ? VT_OK
               : VT_ERR_HIP;
}
