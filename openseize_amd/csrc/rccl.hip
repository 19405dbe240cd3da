// rccl.hip -- the one collective of the path behind the C ABI.
//
// The reference averages periodograms over the whole stream with a running
// mean (src/openseize/spectra/estimators.py:149-152).  When the stream of a
// channel block is split in TIME across GPUs, every rank holds the sum of its
// own periodograms and its segment count; one RCCL all-reduce(sum) of the
// (nch x nfreq) float64 accumulator and of the count makes every rank's handle
// hold the global sum -- osz_welch_reduce (spec.hip).
//
// RCCL is bound at run time (dlopen), not at link time: a PyTorch host already
// has its own librccl.so mapped (same soname -> the same instance is returned,
// so communicators and calls agree), a C host gets /opt/rocm/lib/librccl.so.1;
// and libosz_hip.so keeps loading on machines without RCCL.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "common.h"

namespace osz {

namespace {
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*get_unique_id)(ncclUniqueId *) = nullptr;
    ncclResult_t (*comm_init_rank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    ncclResult_t (*comm_count)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*all_reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t,
                               ncclComm_t, hipStream_t) = nullptr;
    const char *(*error_string)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;
std::mutex g_rccl_mu;

int rccl_fail(const char *what, ncclResult_t r) {
    return fail(OSZ_ERR_HIP, "%s: %s", what,
                g_rccl.error_string ? g_rccl.error_string(r) : "RCCL error");
}
}  // namespace

int rccl_bind(const char *path) {
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (g_rccl.lib) return OSZ_OK;
    const char *candidates[] = {path, "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    void *lib = nullptr;
    for (const char *c : candidates) {
        if (!c) continue;
        lib = dlopen(c, RTLD_NOW | RTLD_LOCAL);
        if (lib) break;
    }
    if (!lib) return fail(OSZ_ERR_UNSUPPORTED, "osz_rccl_bind: cannot load librccl (%s)", dlerror());
    RcclApi api;
    api.lib = lib;
#define OSZ_SYM(field, name)                                                        \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(lib, name));           \
    if (!api.field) return fail(OSZ_ERR_UNSUPPORTED, "osz_rccl_bind: %s not found", name)
    OSZ_SYM(get_unique_id, "ncclGetUniqueId");
    OSZ_SYM(comm_init_rank, "ncclCommInitRank");
    OSZ_SYM(comm_destroy, "ncclCommDestroy");
    OSZ_SYM(comm_count, "ncclCommCount");
    OSZ_SYM(all_reduce, "ncclAllReduce");
    OSZ_SYM(error_string, "ncclGetErrorString");
#undef OSZ_SYM
    g_rccl = api;
    return OSZ_OK;
}

// in-place sum over the ranks of `comm`; count elements of float64 or int64
int rccl_allreduce_sum(void *buf, size_t count, bool is_f64, void *comm, hipStream_t st) {
    int rc = rccl_bind(nullptr);
    if (rc) return rc;
    ncclResult_t r = g_rccl.all_reduce(buf, buf, count, is_f64 ? ncclFloat64 : ncclInt64, ncclSum,
                                       reinterpret_cast<ncclComm_t>(comm), st);
    if (r != ncclSuccess) return rccl_fail("ncclAllReduce", r);
    return OSZ_OK;
}

}  // namespace osz

using namespace osz;

extern "C" {

int osz_rccl_bind(const char *path) { return rccl_bind(path); }

int osz_rccl_unique_id(char *id128) {
    OSZ_REQUIRE(id128, "osz_rccl_unique_id: null argument");
    static_assert(sizeof(ncclUniqueId) == OSZ_RCCL_ID_BYTES, "ncclUniqueId is 128 bytes");
    int rc = rccl_bind(nullptr);
    if (rc) return rc;
    ncclUniqueId id;
    ncclResult_t r = g_rccl.get_unique_id(&id);
    if (r != ncclSuccess) return rccl_fail("ncclGetUniqueId", r);
    memcpy(id128, &id, sizeof id);
    return OSZ_OK;
}

int osz_rccl_comm_create(void **comm, int nranks, int rank, const char *id128) {
    OSZ_REQUIRE(comm && id128, "osz_rccl_comm_create: null argument");
    OSZ_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "osz_rccl_comm_create: rank %d of %d",
                rank, nranks);
    int rc = rccl_bind(nullptr);
    if (rc) return rc;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t c = nullptr;
    ncclResult_t r = g_rccl.comm_init_rank(&c, nranks, id, rank);
    if (r != ncclSuccess) return rccl_fail("ncclCommInitRank", r);
    *comm = c;
    return OSZ_OK;
}

int osz_rccl_comm_destroy(void *comm) {
    if (!comm) return OSZ_OK;
    int rc = rccl_bind(nullptr);
    if (rc) return rc;
    ncclResult_t r = g_rccl.comm_destroy(reinterpret_cast<ncclComm_t>(comm));
    if (r != ncclSuccess) return rccl_fail("ncclCommDestroy", r);
    return OSZ_OK;
}

int osz_rccl_comm_size(void *comm, int *nranks) {
    OSZ_REQUIRE(comm && nranks, "osz_rccl_comm_size: null argument");
    int rc = rccl_bind(nullptr);
    if (rc) return rc;
    ncclResult_t r = g_rccl.comm_count(reinterpret_cast<ncclComm_t>(comm), nranks);
    if (r != ncclSuccess) return rccl_fail("ncclCommCount", r);
    return OSZ_OK;
}

}  // extern "C"
