// rccl_hook.cpp — a native all-reduce hook for the landmark-sharded BA (include/svi_hot.h svi_rccl_*).
//
// The one exchange step of the path - the sum of the reduced camera normal equations over the landmark shards, plus the few
// scalars of an LM decision - as ncclAllReduce( ..., ncclDouble, ncclSum, comm, stream ) on the handle's own HIP stream: one
// process per GPU, RCCL over xGMI, no Python and no torch in between.  RCCL is not linked: like the HIP runtime it is taken
// from the process (a host that already loaded librccl - PyTorch-ROCm does - shares it; otherwise $ROCM_PATH/lib/librccl.so).
// The reference has no counterpart (it is a single process); this is north_star's addition.
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "common.h"

namespace {

struct ncclUniqueId { char internal[128]; };
typedef void* ncclComm_t;
typedef int ncclResult_t;
enum { kNcclSum = 0, kNcclDouble = 8 }; // rccl.h: ncclSum = 0, ncclFloat64 = ncclDouble = 8

struct Api {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

// loaded once, by whichever thread asks first (several shard threads of one process may initialise side by side: a function-local
// static is initialised exactly once and the others wait for it); load_error keeps the loader's message for the callers
std::string g_load_error;

Api load_api()
{
    Api a;
    std::string rocm = getenv("ROCM_PATH") ? getenv("ROCM_PATH") : "/opt/rocm";
    const std::string cands[] = {"librccl.so", "librccl.so.1", rocm + "/lib/librccl.so"};
    for (const std::string& c : cands) {
        void* h = dlopen(c.c_str(), RTLD_NOW | RTLD_GLOBAL);
        if (!h) { const char* e = dlerror(); g_load_error = e ? e : "not found"; continue; } // (dlerror clears itself: read once)
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(h, "ncclAllReduce"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        if (a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce && a.GetErrorString) { a.lib = h; return a; }
        g_load_error = c + ": an ncclXxx entry point is missing";
        dlclose(h);
    }
    return Api{};
}

Api* api()
{
    static Api a = load_api();
    return a.lib ? &a : nullptr;
}

int nccl_fail(Api* a, const char* what, ncclResult_t r)
{
    return svi::fail(SVI_ERR_COMM, "%s: %s", what, a->GetErrorString ? a->GetErrorString(r) : "RCCL error");
}

} // namespace

struct svi_rccl {
    ncclComm_t comm = nullptr;
    int rank = 0, n_ranks = 1, device = 0;
};

extern "C" {

// 1 if librccl resolved in this process (nothing collective happens here): the ranks exchange this answer BEFORE any of them
// enters ncclCommInitRank, which blocks until every rank has arrived - a rank that cannot load the library must not leave the
// others waiting inside it
int svi_rccl_available(void) { return api() ? 1 : 0; }

int svi_rccl_unique_id(void* id_out)
{
    if (!id_out) return svi::fail(SVI_ERR_INVALID, "svi_rccl_unique_id: null argument");
    Api* a = api();
    if (!a) return svi::fail(SVI_ERR_COMM, "librccl.so could not be loaded: %s", g_load_error.c_str());
    ncclUniqueId id;
    const ncclResult_t r = a->GetUniqueId(&id);
    if (r != 0) return nccl_fail(a, "ncclGetUniqueId", r);
    memcpy(id_out, &id, sizeof(id));
    return SVI_OK;
}

int svi_rccl_create(const void* unique_id, int rank, int n_ranks, int device, svi_rccl** out)
{
    if (!unique_id || !out) return svi::fail(SVI_ERR_INVALID, "svi_rccl_create: null argument");
    *out = nullptr;
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return svi::fail(SVI_ERR_INVALID, "svi_rccl_create: bad rank %d / n_ranks %d", rank, n_ranks);
    Api* a = api();
    if (!a) return svi::fail(SVI_ERR_COMM, "librccl.so could not be loaded: %s", g_load_error.c_str());
    if (int rc = svi::use_device(device)) return rc;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm = nullptr;
    const ncclResult_t r = a->CommInitRank(&comm, n_ranks, id, rank);
    if (r != 0) return nccl_fail(a, "ncclCommInitRank", r);
    svi_rccl* c = new svi_rccl();
    c->comm = comm; c->rank = rank; c->n_ranks = n_ranks; c->device = device;
    *out = c;
    return SVI_OK;
}

int svi_rccl_destroy(svi_rccl* c)
{
    if (!c) return SVI_OK;
    Api* a = api();
    if (a && c->comm) (void)a->CommDestroy(c->comm);
    delete c;
    return SVI_OK;
}

// svi_allreduce_fn: user = the svi_rccl*; in-place sum of `count` doubles, ordered on `stream`
int svi_rccl_allreduce(void* user, void* buf, size_t count, void* stream)
{
    svi_rccl* c = static_cast<svi_rccl*>(user);
    Api* a = api();
    if (!c || !a || !buf) return 1;
    const ncclResult_t r = a->AllReduce(buf, buf, count, kNcclDouble, kNcclSum, c->comm, static_cast<hipStream_t>(stream));
    if (r != 0) { nccl_fail(a, "ncclAllReduce", r); return (int)r; }
    return 0;
}

} // extern "C"
