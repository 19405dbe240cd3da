// lib.hip -- library-level entry points: errors, device info, memory/stream
// helpers, HIP-event timing.
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"

namespace osz {

namespace {
struct ProfRec {
    int id;
    hipEvent_t a, b;
};
bool g_prof = false;
std::vector<std::string> g_names;
std::vector<ProfRec> g_recs;       // records of the current window
std::vector<hipEvent_t> g_pool;    // recycled events
std::map<std::string, std::pair<int64_t, double>> g_totals;

hipEvent_t prof_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

void prof_collect() {
    for (auto &r : g_recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess &&
            hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            auto &t = g_totals[g_names[r.id]];
            t.first += 1;
            t.second += ms;
        }
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    g_recs.clear();
}
}  // namespace

bool profile_on() { return g_prof; }

int profile_begin(const char *name, hipStream_t st) {
    int id = -1;
    for (size_t i = 0; i < g_names.size(); ++i)
        if (g_names[i] == name) id = (int)i;
    if (id < 0) {
        g_names.push_back(name);
        id = (int)g_names.size() - 1;
    }
    ProfRec r{id, prof_event(), prof_event()};
    (void)hipEventRecord(r.a, st);
    g_recs.push_back(r);
    return (int)g_recs.size() - 1;
}

// rec: what profile_begin returned (timers nest: osz_chain_step brackets two launches
// that carry timers of their own, on two streams)
void profile_end(int rec, hipStream_t st) {
    if (rec >= 0 && rec < (int)g_recs.size()) (void)hipEventRecord(g_recs[rec].b, st);
}

int current_device(int *dev) {
    OSZ_HIP(hipGetDevice(dev));
    return OSZ_OK;
}

int ensure_dyn_lds(const void *kern, size_t bytes) {
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, size_t> done;
    int dev = 0;
    OSZ_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    auto key = std::make_pair(kern, dev);
    auto it = done.find(key);
    if (it != done.end() && it->second >= bytes) return OSZ_OK;
    OSZ_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    done[key] = bytes;
    return OSZ_OK;
}

char *err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace osz

using namespace osz;

extern "C" {

int osz_version(void) { return 100; }

const char *osz_last_error(void) { return err_buf(); }

int osz_device_info(int *cu_count, size_t *hbm_bytes, char *name, int name_len) {
    int dev = 0;
    OSZ_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    OSZ_HIP(hipGetDeviceProperties(&prop, dev));
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    if (name && name_len > 0) {
        snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    return OSZ_OK;
}

int osz_profile_enable(int on) {
    g_prof = on != 0;
    return OSZ_OK;
}

int osz_profile_reset(void) {
    prof_collect();
    g_totals.clear();
    return OSZ_OK;
}

int osz_profile_query(const char *name, int64_t *launches, double *total_ms) {
    OSZ_REQUIRE(name && launches && total_ms, "osz_profile_query: null argument");
    prof_collect();
    auto it = g_totals.find(name);
    *launches = it == g_totals.end() ? 0 : it->second.first;
    *total_ms = it == g_totals.end() ? 0.0 : it->second.second;
    return OSZ_OK;
}

int osz_malloc(void **dptr, size_t bytes) {
    OSZ_REQUIRE(dptr != nullptr, "osz_malloc: null out pointer");
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
    if (e == hipErrorOutOfMemory)
        return fail(OSZ_ERR_NOMEM, "osz_malloc: out of device memory (%zu B)", bytes);
    OSZ_HIP(e);
    return OSZ_OK;
}

int osz_free(void *dptr) {
    OSZ_HIP(hipFree(dptr));
    return OSZ_OK;
}

int osz_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream) {
    OSZ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    return OSZ_OK;
}

int osz_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream) {
    OSZ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    return OSZ_OK;
}

int osz_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream) {
    OSZ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    return OSZ_OK;
}

int osz_memset(void *dst, int value, size_t bytes, void *stream) {
    OSZ_HIP(hipMemsetAsync(dst, value, bytes, as_stream(stream)));
    return OSZ_OK;
}

int osz_stream_sync(void *stream) {
    OSZ_HIP(hipStreamSynchronize(as_stream(stream)));
    return OSZ_OK;
}

int osz_event_create(void **ev) {
    OSZ_REQUIRE(ev != nullptr, "osz_event_create: null out pointer");
    hipEvent_t e;
    OSZ_HIP(hipEventCreate(&e));
    *ev = e;
    return OSZ_OK;
}

int osz_event_destroy(void *ev) {
    OSZ_HIP(hipEventDestroy(reinterpret_cast<hipEvent_t>(ev)));
    return OSZ_OK;
}

int osz_event_record(void *ev, void *stream) {
    OSZ_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(ev), as_stream(stream)));
    return OSZ_OK;
}

int osz_event_elapsed_ms(void *start, void *stop, float *ms) {
    OSZ_HIP(hipEventSynchronize(reinterpret_cast<hipEvent_t>(stop)));
    OSZ_HIP(hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(start),
                                reinterpret_cast<hipEvent_t>(stop)));
    return OSZ_OK;
}

}  // extern "C"
