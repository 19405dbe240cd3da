// common.h -- shared host-side plumbing for libosz_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/osz_hip.h"

namespace osz {

// thread-local last-error text behind osz_last_error()
char *err_buf();
int fail(int code, const char *fmt, ...);

#define OSZ_HIP(call)                                                        \
    do {                                                                     \
        hipError_t e_ = (call);                                              \
        if (e_ != hipSuccess)                                                \
            return osz::fail(OSZ_ERR_HIP, "%s: %s (%s:%d)", #call,           \
                             hipGetErrorString(e_), __FILE__, __LINE__);     \
    } while (0)

#define OSZ_REQUIRE(cond, ...)                                               \
    do {                                                                     \
        if (!(cond)) return osz::fail(OSZ_ERR_INVALID, __VA_ARGS__);         \
    } while (0)

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute: this
// sets it once per (kernel, device) pair (mutex-guarded map in lib.hip).
int ensure_dyn_lds(const void *kern, size_t bytes);
#define OSZ_DYN_LDS(kern, bytes)                                             \
    do {                                                                     \
        int rc_ = osz::ensure_dyn_lds(reinterpret_cast<const void *>(kern), (bytes)); \
        if (rc_) return rc_;                                                 \
    } while (0)

// Handles own device buffers: every entry point checks that the calling
// thread's current device is the one the handle was created on.
int current_device(int *dev);
#define OSZ_SAME_DEVICE(h, fn)                                               \
    do {                                                                     \
        int dev_ = -1;                                                       \
        int rc_ = osz::current_device(&dev_);                                \
        if (rc_) return rc_;                                                 \
        if (dev_ != (h)->device)                                             \
            return osz::fail(OSZ_ERR_STATE, "%s: handle belongs to device %d, current device is %d", \
                             fn, (h)->device, dev_);                         \
    } while (0)

constexpr int kWave = 64;  // CDNA wavefront width

// rccl.hip: in-place all-reduce(sum) of `count` float64 / int64 elements over
// the ranks of an ncclComm_t; RCCL is bound at run time (dlopen)
int rccl_allreduce_sum(void *buf, size_t count, bool is_f64, void *comm, hipStream_t st);

// Optional per-kernel timing with HIP events on the launch stream
// (osz_profile_enable / osz_profile_query): bench.py uses it to get each
// kernel's average duration live, inside the timed region.
bool profile_on();
int profile_begin(const char *name, hipStream_t st);
void profile_end(int rec, hipStream_t st);
struct KernelTimer {
    hipStream_t st;
    bool on;
    int rec = -1;
    KernelTimer(const char *name, hipStream_t s) : st(s), on(profile_on()) {
        if (on) rec = profile_begin(name, st);
    }
    ~KernelTimer() {
        if (on) profile_end(rec, st);
    }
};

}  // namespace osz
