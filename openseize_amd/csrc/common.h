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

// Where a caller's state array lives: the get/set_state entry points take host arrays (and wait
// for the copy) or device arrays (the copy is ordered on the stream, nothing waits).
inline bool on_device(const void *p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();               // plain host memory: not an error of ours
        return false;
    }
    return at.type == hipMemoryTypeDevice;
}

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

// DPP moves with bound_ctrl (lanes without a source read 0) and no `old`
// operand: nothing to initialise in front of them.
template <int CTRL>
__device__ __forceinline__ double dpp_mov0(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// Sum over the 64 lanes of a wave, valid in LANE 63 (the other lanes hold partial
// sums): four row_shr steps inside the 16-lane rows, then row_bcast:15 (a row takes the
// total of the row before it) and row_bcast:31 (lanes 32..63 take lane 31) -- vector
// moves only, where a __shfl_down chain goes through the LDS crossbar (ds_bpermute) and
// waits for it six times.  (benchmarks/dpp_probe.hip pins what these controls deliver.)
__device__ __forceinline__ double wave_sum63(double v) {
    v += dpp_mov0<0x111>(v);
    v += dpp_mov0<0x112>(v);
    v += dpp_mov0<0x114>(v);
    v += dpp_mov0<0x118>(v);
    v += dpp_mov0<0x142>(v);
    v += dpp_mov0<0x143>(v);
    return v;
}

// wave-private LDS hand-offs need no workgroup barrier: LDS executes one
// wave's instructions in order; this only stops the compiler reordering them
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The two workgroups of a CU (the 256-thread kernels that fill it with two: one wave of each
// per SIMD) do not get equal shares by themselves: the issue arbiter serves the older wave
// first, the workgroup placed second runs ~20 % behind and then finishes alone on a half-empty
// CU (benchmarks/zp_timeline.hip).  Each wave therefore raises and drops its priority in turn
// with the other wave of its SIMD (they differ in the lowest bit of their wave slot),
// switching on bit 18 of the shader clock (~130 us; 12 ... 20 measured).  Long launches at
// full width gain nothing (the chip runs at the clock its power allows either way), short and
// narrow ones do: the zero-phase chain at 32 channels 297 -> 266 us (the overlap-add FIR and
// the Welch kernel measured no different with it and stay without).
__device__ __forceinline__ void take_turns() {
#ifdef OSZ_NO_TURNS   // diagnostic builds only: the arbiter's own order
    return;
#endif
    unsigned slot;
    unsigned long long now;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 1)" : "=s"(slot));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
    if ((((unsigned)(now >> 18)) ^ slot) & 1) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
}

// Buffer addressing for rows of one channel: the base (a uniform pointer) and the row's byte
// offset travel in scalar registers, a lane contributes one 32-bit offset -- no 64-bit vector
// arithmetic per access, which is what `p[256 * j + t]` costs once the immediate field (4 KB)
// is exceeded.  Raw buffer, stride 0; word 3 as gfx90a / gfx942 / gfx950 want it.
typedef unsigned buf_u2 __attribute__((ext_vector_type(2)));
typedef unsigned buf_u4 __attribute__((ext_vector_type(4)));
typedef double buf_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void *base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ double buf_load(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned row_bytes) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, lane_bytes, row_bytes, 0));
}
// (non-temporal: the last read of a line that other lines are wanted in the cache after)
__device__ __forceinline__ double buf_load_nt(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned row_bytes) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, lane_bytes, row_bytes, 2));
}
__device__ __forceinline__ buf_d2 buf_load2(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned row_bytes) {
    return __builtin_bit_cast(buf_d2, __builtin_amdgcn_raw_buffer_load_b128(r, lane_bytes, row_bytes, 0));
}
// (output rows are written once and not read again by the launch: stored non-temporal -- aux
// bit 1 = `nt` -- they do not displace the spectrum in L2 / the Infinity Cache; with the same hint
// on the rows' LDS-DMA requests the zero-phase chain runs 2 % faster, profiles/README.md round 5)
__device__ __forceinline__ void buf_store(double v, __amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned row_bytes) {
#ifdef OSZ_NO_NT      // (A/B builds only)
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(buf_u2, v), r, lane_bytes, row_bytes, 0);
#else
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(buf_u2, v), r, lane_bytes, row_bytes, 2);
#endif
}

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
