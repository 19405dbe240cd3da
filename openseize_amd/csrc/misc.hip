// misc.hip -- K7 mask compaction (gather along samples), the synthetic
// device-resident source used by benchmarks/tests, and a block checksum.
#include "common.h"

namespace osz {

// y[c, j] = x[c, idx[j]] : np.take(arr, np.flatnonzero(mask), axis) of
// MaskedProducer.__iter__ (reference core/producer.py:432).  Consecutive lanes
// write consecutive outputs; reads are as coalesced as the mask allows.
__global__ void take_kernel(const double *x, int64_t ldx, const int64_t *idx, int64_t nidx,
                            double *y, int64_t ldy) {
    const int c = blockIdx.y;
    const double *xr = x + (int64_t)c * ldx;
    double *yr = y + (int64_t)c * ldy;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nidx;
         j += (int64_t)gridDim.x * blockDim.x)
        yr[j] = xr[idx[j]];
}

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    // splitmix64 finaliser
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Counter-based N(0,1): value depends only on (seed, channel, sample index).
__global__ void synth_normal_kernel(double *x, int64_t ldx, int64_t n, uint64_t seed, int64_t ch0,
                                    int64_t n0) {
    const int c = blockIdx.y;
    double *xr = x + (int64_t)c * ldx;
    const uint64_t key = mix64(seed ^ (0x9E3779B97F4A7C15ull * (uint64_t)(ch0 + c + 1)));
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n;
         j += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t ctr = (uint64_t)(n0 + j);
        const uint64_t r1 = mix64(key + 2 * ctr);
        const uint64_t r2 = mix64(key + 2 * ctr + 1);
        const double u1 = ((double)(r1 >> 11) + 0.5) * (1.0 / 9007199254740992.0);
        const double u2 = ((double)(r2 >> 11) + 0.5) * (1.0 / 9007199254740992.0);
        xr[j] = sqrt(-2.0 * log(u1)) * cospi(2.0 * u2);
    }
}

__global__ void checksum_kernel(const double *x, int64_t ldx, int64_t n,
                                unsigned long long *bits, double *fsum) {
    const int c = blockIdx.y;
    const double *xr = x + (int64_t)c * ldx;
    unsigned long long b = 0;
    double f = 0.0;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n;
         j += (int64_t)gridDim.x * blockDim.x) {
        const double v = xr[j];
        b += (unsigned long long)__double_as_longlong(v);
        f += v;
    }
    for (int off = 32; off > 0; off >>= 1) {
        b += __shfl_down(b, off, 64);
        f += __shfl_down(f, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(bits, b);
        atomicAdd(fsum, f);
    }
}

}  // namespace osz

using namespace osz;

extern "C" {

int osz_take(const double *x, int64_t ldx, int nch, const int64_t *idx, int64_t nidx, double *y,
             int64_t ldy, void *stream) {
    OSZ_REQUIRE(x && idx && y, "osz_take: null argument");
    OSZ_REQUIRE(nch >= 1 && nch <= 65535 && nidx >= 0 && ldy >= nidx, "osz_take: bad sizes");
    if (nidx == 0) return OSZ_OK;
    int64_t bx = (nidx + 255) / 256;
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(take_kernel, dim3((unsigned)bx, nch), dim3(256), 0, as_stream(stream), x,
                       ldx, idx, nidx, y, ldy);
    OSZ_HIP(hipGetLastError());
    return OSZ_OK;
}

int osz_synth_normal(double *x, int64_t ldx, int nch, int64_t n, uint64_t seed, int64_t ch0,
                     int64_t n0, void *stream) {
    OSZ_REQUIRE(x && nch >= 1 && n >= 0 && ldx >= n, "osz_synth_normal: bad arguments");
    if (n == 0) return OSZ_OK;
    int64_t bx = (n + 255) / 256;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(synth_normal_kernel, dim3((unsigned)bx, nch), dim3(256), 0,
                       as_stream(stream), x, ldx, n, seed, ch0, n0);
    OSZ_HIP(hipGetLastError());
    return OSZ_OK;
}

int osz_checksum(const double *x, int64_t ldx, int nch, int64_t n, uint64_t *bits, double *fsum,
                 void *stream) {
    OSZ_REQUIRE(x && bits && fsum && nch >= 1 && n >= 0, "osz_checksum: bad arguments");
    hipStream_t st = as_stream(stream);
    struct Acc {
        unsigned long long bits;
        double fsum;
    };
    Acc *d = nullptr;
    OSZ_HIP(hipMalloc(&d, sizeof(Acc)));
    OSZ_HIP(hipMemsetAsync(d, 0, sizeof(Acc), st));
    if (n > 0) {
        int64_t bx = (n + 255) / 256;
        if (bx > 16) bx = 16;   // few blocks per row: the two atomics per wave stay cheap
        hipLaunchKernelGGL(checksum_kernel, dim3((unsigned)bx, nch), dim3(256), 0, st, x, ldx, n,
                           &d->bits, &d->fsum);
    }
    Acc hacc;
    OSZ_HIP(hipMemcpyAsync(&hacc, d, sizeof(Acc), hipMemcpyDeviceToHost, st));
    OSZ_HIP(hipStreamSynchronize(st));
    OSZ_HIP(hipFree(d));
    *bits = hacc.bits;
    *fsum = hacc.fsum;
    return OSZ_OK;
}

}  // extern "C"
