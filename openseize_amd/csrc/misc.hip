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

// EDF records -> physical float64 samples (reference file_io/edf.py:452-556):
// a record holds spr[c] little-endian int16 samples of every signal one after
// the other; out[c, i] = raw * slope[c] + offset[c] for sample start + i of
// channel c (two roundings, as the reference's `arr * slopes; += offsets`);
// positions a channel cannot fill get padvalue pushed through the same map
// (the reference pads BEFORE deciphering, edf.py:553-556).
struct EdfArgs {
    const int16_t *raw;      // records [rec0, rec0 + nrec) of the file
    const int32_t *choff;    // offset of channel c inside a record (samples)
    const int32_t *spr;      // samples per record of channel c
    const double *slope, *offset;
    const int64_t *len;      // valid output samples of channel c
    double *out;
    int64_t ldo, rec0, start, width;
    int reclen;
    double padvalue;
};

__global__ void edf_decode_kernel(EdfArgs a) {
#pragma clang fp contract(off)   // keep the reference's two roundings: no fused multiply-add
    const int c = blockIdx.y;
    const int spr = a.spr[c];
    const double slope = a.slope[c], offset = a.offset[c];
    const int64_t len = a.len[c];
    const int16_t *raw = a.raw + a.choff[c];
    double *o = a.out + (int64_t)c * a.ldo;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.width;
         i += (int64_t)gridDim.x * blockDim.x) {
        double d = a.padvalue;
        if (i < len) {
            const int64_t s = a.start + i;
            const int64_t rec = s / spr - a.rec0;
            d = (double)raw[rec * a.reclen + (s % spr)];
        }
        const double scaled = d * slope;
        o[i] = scaled + offset;
    }
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

int osz_edf_decode(const int16_t *raw, int reclen, int nch, const int32_t *choff,
                   const int32_t *spr, const double *slope, const double *offset,
                   const int64_t *len, int64_t rec0, int64_t start, int64_t width, double padvalue,
                   double *out, int64_t ldo, void *stream) {
    OSZ_REQUIRE(raw && choff && spr && slope && offset && len && out, "osz_edf_decode: null argument");
    OSZ_REQUIRE(nch >= 1 && nch <= 65535 && reclen >= 1 && width >= 0 && ldo >= width,
                "osz_edf_decode: bad sizes");
    if (width == 0) return OSZ_OK;
    EdfArgs a{};
    a.raw = raw;
    a.choff = choff;
    a.spr = spr;
    a.slope = slope;
    a.offset = offset;
    a.len = len;
    a.out = out;
    a.ldo = ldo;
    a.rec0 = rec0;
    a.start = start;
    a.width = width;
    a.reclen = reclen;
    a.padvalue = padvalue;
    int64_t bx = (width + 255) / 256;
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(edf_decode_kernel, dim3((unsigned)bx, nch), dim3(256), 0, as_stream(stream),
                       a);
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
