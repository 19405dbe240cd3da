// glue.hip -- the rank-2 / rank-4 arithmetic that sits between the DSP kernels
// in a device-resident producer chain: streaming per-channel
// moments, elementwise broadcast arithmetic, complex join / magnitude / phase,
// and Simpson band power.  All HBM-bound stream kernels: consecutive lanes on
// consecutive samples of one row, 8 bytes per lane (rows of a chunk view are
// not 16-byte aligned in general), deterministic reductions (fixed partition,
// fixed fold order -- no atomics).
//
// Reference call sites replaced (src/openseize/):
//   core/protools.py:500-545 mean, :547-592 std, :594-671 standardize,
//   :72-125 add, :127-180 multiply, :334-384 multiply_along_axis;
//   experimental/coupling/transforms.py:153-192 (x + i H(x), |z|, angle);
//   spectra/metrics.py:25-87 power (scipy.integrate.simpson), :90-141 power_norm.
#include <cmath>

#include "common.h"

namespace osz {

constexpr int kMomBlk = 256;

// deterministic block sum of three doubles (wave shuffles, then the 4 wave
// totals in wave order through LDS); result valid in thread 0
__device__ __forceinline__ void block_sum3(double &a, double &b, double &c, double *lds) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        a += __shfl_down(a, off, 64);
        b += __shfl_down(b, off, 64);
        c += __shfl_down(c, off, 64);
    }
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) {
        lds[3 * w + 0] = a;
        lds[3 * w + 1] = b;
        lds[3 * w + 2] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = b = c = 0.0;
        for (int q = 0; q < (int)(blockDim.x >> 6); ++q) {
            a += lds[3 * q + 0];
            b += lds[3 * q + 1];
            c += lds[3 * q + 2];
        }
    }
}

// partial[(c * nblk + blk) * 3 + {0, 1, 2}] = sum x, sum x^2, count over the
// columns [blk * span, (blk + 1) * span) of row c
__global__ __launch_bounds__(kMomBlk) void moments_partial_kernel(
    const double *__restrict__ x, int64_t ldx, int64_t n, int64_t span, int ignore_nan,
    double *__restrict__ partial) {
    __shared__ double lds[3 * (kMomBlk / 64)];
    const int c = blockIdx.y, blk = blockIdx.x;
    const double *row = x + (int64_t)c * ldx;
    const int64_t lo = (int64_t)blk * span;
    int64_t hi = lo + span;
    if (hi > n) hi = n;
    double s1 = 0.0, s2 = 0.0, k = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += kMomBlk) {
        const double v = row[i];
        if (ignore_nan && v != v) continue;
        s1 += v;
        s2 = fma(v, v, s2);
        k += 1.0;
    }
    block_sum3(s1, s2, k, lds);
    if (threadIdx.x == 0) {
        double *p = partial + ((int64_t)c * gridDim.x + blk) * 3;
        p[0] = s1;
        p[1] = s2;
        p[2] = k;
    }
}

// acc (3, nch): A += n * mean(x), B += n * mean(x^2), L += n  -- the
// chunk-length weighting of protools.mean / std (:533-536, :579-584)
__global__ void moments_fold_kernel(const double *__restrict__ partial, int nblk, int nch, double n,
                                    double *__restrict__ acc) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nch) return;
    double s1 = 0.0, s2 = 0.0, k = 0.0;
    for (int b = 0; b < nblk; ++b) {
        const double *p = partial + ((int64_t)c * nblk + b) * 3;
        s1 += p[0];
        s2 += p[1];
        k += p[2];
    }
    acc[c] += n * (s1 / k);
    acc[nch + c] += n * (s2 / k);
    acc[2 * nch + c] += n;
}

__global__ void moments_finish_kernel(const double *__restrict__ acc, int nch, double *mean,
                                      double *sd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nch) return;
    const double m = acc[c] / acc[2 * nch + c];
    if (mean) mean[c] = m;
    if (sd) sd[c] = sqrt(acc[nch + c] / acc[2 * nch + c] - m * m);
}

// Moments along the FIRST axis of a (nred, ncols) matrix, one thread per column
// (consecutive lanes on consecutive columns): numpy.(nan)mean and the two-pass
// numpy.(nan)std the reference applies per chunk when the reduced axis is not
// the production axis (core/protools.py:538-545, :586-592).
__global__ __launch_bounds__(256) void col_moments_kernel(const double *__restrict__ x, int64_t ldx,
                                                          int nred, int64_t ncols, int ignore_nan,
                                                          double *mean, double *sd) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ncols) return;
    double s = 0.0, k = 0.0;
    for (int c = 0; c < nred; ++c) {
        const double v = x[(int64_t)c * ldx + j];
        if (ignore_nan && v != v) continue;
        s += v;
        k += 1.0;
    }
    const double m = s / k;
    if (mean) mean[j] = m;
    if (sd) {
        double q = 0.0;
        for (int c = 0; c < nred; ++c) {
            const double v = x[(int64_t)c * ldx + j];
            if (ignore_nan && v != v) continue;
            q = fma(v - m, v - m, q);
        }
        sd[j] = sqrt(q / k);
    }
}

// y = x (op) operand(s); KIND: how a (and b) are indexed
template <int OP, int KIND>
__global__ __launch_bounds__(256) void ew_kernel(const double *__restrict__ x, int64_t ldx, int64_t n,
                                                 const double *__restrict__ a,
                                                 const double *__restrict__ b, int64_t ldab,
                                                 double *__restrict__ y, int64_t ldy) {
    const int c = blockIdx.y;
    const double *xr = x + (int64_t)c * ldx;
    double *yr = y + (int64_t)c * ldy;
    double a0 = 0.0, b0 = 1.0;
    if (KIND == OSZ_BCAST_SCALAR) {
        a0 = a[0];
        if (OP == OSZ_EW_STANDARDIZE) b0 = b[0];
    } else if (KIND == OSZ_BCAST_ROW) {
        a0 = a[c];
        if (OP == OSZ_EW_STANDARDIZE) b0 = b[c];
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        double av = a0, bv = b0;
        if (KIND == OSZ_BCAST_COL) {
            av = a[i];
            if (OP == OSZ_EW_STANDARDIZE) bv = b[i];
        } else if (KIND == OSZ_BCAST_FULL) {
            av = a[(int64_t)c * ldab + i];
            if (OP == OSZ_EW_STANDARDIZE) bv = b[(int64_t)c * ldab + i];
        }
        const double v = xr[i];
        double r;
        if (OP == OSZ_EW_ADD) r = v + av;
        else if (OP == OSZ_EW_MUL) r = v * av;
        else if (OP == OSZ_EW_DIV) r = v / av;
        else r = (v - av) / bv;
        yr[i] = r;
    }
}

__global__ __launch_bounds__(256) void complex_join_kernel(const double *__restrict__ re, int64_t ldre,
                                                           const double *__restrict__ im, int64_t ldim,
                                                           int64_t n, double2 *__restrict__ z,
                                                           int64_t ldz) {
    const int c = blockIdx.y;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        z[(int64_t)c * ldz + i] = make_double2(re[(int64_t)c * ldre + i], im[(int64_t)c * ldim + i]);
}

// |z| as numpy.abs (hypot) and the phase of numpy.angle mapped to [0, 2 pi)
__global__ __launch_bounds__(256) void magphase_kernel(const double2 *__restrict__ z, int64_t ldz,
                                                       int64_t n, double *__restrict__ mag,
                                                       double *__restrict__ phase, int64_t ldo) {
    const int c = blockIdx.y;
    constexpr double kTwoPi = 6.283185307179586476925286766559;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const double2 v = z[(int64_t)c * ldz + i];
        if (mag) mag[(int64_t)c * ldo + i] = hypot(v.x, v.y);
        if (phase) {
            double p = atan2(v.y, v.x);
            if (p < 0.0) p += kTwoPi;
            phase[(int64_t)c * ldo + i] = p;
        }
    }
}

// out[c] = scipy.integrate.simpson(p[c, a : a + m], dx): composite rule on the
// first m (odd) or m - 1 samples; for an even count the last interval comes
// from the parabola through the last three points (SciPy >= 1.11)
__global__ __launch_bounds__(256) void simpson_kernel(const double *__restrict__ p, int64_t ldp,
                                                      int64_t a, int64_t m, double dx,
                                                      double *__restrict__ out) {
    __shared__ double lds[3 * 4];
    const int c = blockIdx.x;
    const double *row = p + (int64_t)c * ldp + a;
    if (m == 1) {
        if (threadIdx.x == 0) out[c] = 0.0;
        return;
    }
    if (m == 2) {
        if (threadIdx.x == 0) out[c] = 0.5 * dx * (row[0] + row[1]);
        return;
    }
    const int64_t mo = (m & 1) ? m : m - 1;        // odd count of the composite rule
    double s4 = 0.0, s2 = 0.0, unused = 0.0;
    for (int64_t i = 1 + threadIdx.x; i < mo - 1; i += blockDim.x) {
        if (i & 1) s4 += row[i];
        else s2 += row[i];
    }
    block_sum3(s4, s2, unused, lds);
    if (threadIdx.x == 0) {
        double r = dx / 3.0 * (row[0] + row[mo - 1] + 4.0 * s4 + 2.0 * s2);
        if (!(m & 1)) r += dx * (5.0 * row[m - 1] + 8.0 * row[m - 2] - row[m - 3]) / 12.0;
        out[c] = r;
    }
}

}  // namespace osz

using namespace osz;

struct osz_moments_s {
    int device, nch;
    double *dacc;        // (3, nch): sum n*mean, sum n*mean(x^2), sum n
    double *dpartial;    // (nch, kMaxBlk, 3)
};

static constexpr int kMomMaxBlk = 64;

static dim3 row_grid(int64_t n, int nch) {
    int64_t bx = (n + 256 * 8 - 1) / (256 * 8);      // ~8 elements per thread
    if (bx < 1) bx = 1;
    if (bx > 256) bx = 256;
    return dim3((unsigned)bx, (unsigned)nch);
}

extern "C" {

int osz_moments_create(osz_moments_t *h, int nch) {
    OSZ_REQUIRE(h && nch >= 1 && nch <= 65535, "osz_moments_create: nch=%d not in [1, 65535]", nch);
    osz_moments_s *p = new osz_moments_s();
    p->nch = nch;
    p->device = 0;
    (void)hipGetDevice(&p->device);
    OSZ_HIP(hipMalloc(&p->dacc, sizeof(double) * 3 * nch));
    OSZ_HIP(hipMalloc(&p->dpartial, sizeof(double) * 3 * (size_t)nch * kMomMaxBlk));
    OSZ_HIP(hipMemset(p->dacc, 0, sizeof(double) * 3 * nch));
    *h = p;
    return OSZ_OK;
}

int osz_moments_destroy(osz_moments_t h) {
    if (!h) return OSZ_OK;
    (void)hipFree(h->dacc);
    (void)hipFree(h->dpartial);
    delete h;
    return OSZ_OK;
}

int osz_moments_reset(osz_moments_t h, void *stream) {
    OSZ_REQUIRE(h, "osz_moments_reset: null handle");
    OSZ_HIP(hipMemsetAsync(h->dacc, 0, sizeof(double) * 3 * h->nch, as_stream(stream)));
    return OSZ_OK;
}

int osz_moments_push(osz_moments_t h, const double *x, int64_t ldx, int64_t n, int ignore_nan,
                     void *stream) {
    OSZ_REQUIRE(h && (x || n == 0), "osz_moments_push: null argument");
    OSZ_REQUIRE(n >= 0 && ldx >= n, "osz_moments_push: n=%lld ldx=%lld", (long long)n, (long long)ldx);
    if (n == 0) return OSZ_OK;
    OSZ_SAME_DEVICE(h, "osz_moments_push");
    hipStream_t st = as_stream(stream);
    int64_t nblk = (n + 8191) / 8192;
    if (nblk > kMomMaxBlk) nblk = kMomMaxBlk;
    const int64_t span = (n + nblk - 1) / nblk;
    {
        KernelTimer kt("moments", st);
        hipLaunchKernelGGL(moments_partial_kernel, dim3((unsigned)nblk, h->nch), dim3(kMomBlk), 0, st,
                           x, ldx, n, span, ignore_nan, h->dpartial);
    }
    OSZ_HIP(hipGetLastError());
    hipLaunchKernelGGL(moments_fold_kernel, dim3((h->nch + 255) / 256), dim3(256), 0, st,
                       h->dpartial, (int)nblk, h->nch, (double)n, h->dacc);
    OSZ_HIP(hipGetLastError());
    return OSZ_OK;
}

int osz_moments_finish(osz_moments_t h, double *dmean, double *dstd, void *stream) {
    OSZ_REQUIRE(h, "osz_moments_finish: null handle");
    OSZ_SAME_DEVICE(h, "osz_moments_finish");
    hipLaunchKernelGGL(moments_finish_kernel, dim3((h->nch + 255) / 256), dim3(256), 0,
                       as_stream(stream), h->dacc, h->nch, dmean, dstd);
    OSZ_HIP(hipGetLastError());
    return OSZ_OK;
}

int osz_col_moments(const double *x, int64_t ldx, int nred, int64_t ncols, int ignore_nan,
                    double *dmean, double *dstd, void *stream) {
    OSZ_REQUIRE(x && (dmean || dstd), "osz_col_moments: null argument");
    OSZ_REQUIRE(nred >= 1 && ncols >= 0 && ldx >= ncols, "osz_col_moments: bad shape");
    if (ncols == 0) return OSZ_OK;
    hipLaunchKernelGGL(col_moments_kernel, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0,
                       as_stream(stream), x, ldx, nred, ncols, ignore_nan, dmean, dstd);
    OSZ_HIP(hipGetLastError());
    return OSZ_OK;
}

int osz_ew(int op, const double *x, int64_t ldx, int nch, int64_t n, const double *a,
           const double *b, int kind, int64_t ldab, double *y, int64_t ldy, void *stream) {
    OSZ_REQUIRE(x && a && y, "osz_ew: null argument");
    OSZ_REQUIRE(op >= OSZ_EW_ADD && op <= OSZ_EW_STANDARDIZE, "osz_ew: unknown op %d", op);
    OSZ_REQUIRE(kind >= OSZ_BCAST_SCALAR && kind <= OSZ_BCAST_FULL, "osz_ew: unknown operand kind %d",
                kind);
    OSZ_REQUIRE(op != OSZ_EW_STANDARDIZE || b, "osz_ew: standardize needs two operands");
    OSZ_REQUIRE(nch >= 1 && nch <= 65535 && n >= 0 && ldx >= n && ldy >= n &&
                    (kind != OSZ_BCAST_FULL || ldab >= n),
                "osz_ew: bad shape (nch=%d n=%lld)", nch, (long long)n);
    if (n == 0) return OSZ_OK;
    hipStream_t st = as_stream(stream);
    using kern_t = void (*)(const double *, int64_t, int64_t, const double *, const double *, int64_t,
                            double *, int64_t);
#define OSZ_EW_ROW(OP)                                                                   \
    {ew_kernel<OP, OSZ_BCAST_SCALAR>, ew_kernel<OP, OSZ_BCAST_ROW>, ew_kernel<OP, OSZ_BCAST_COL>, \
     ew_kernel<OP, OSZ_BCAST_FULL>}
    static const kern_t kerns[4][4] = {OSZ_EW_ROW(OSZ_EW_ADD), OSZ_EW_ROW(OSZ_EW_MUL),
                                       OSZ_EW_ROW(OSZ_EW_DIV), OSZ_EW_ROW(OSZ_EW_STANDARDIZE)};
#undef OSZ_EW_ROW
    KernelTimer kt("ew", st);
    hipLaunchKernelGGL(kerns[op][kind], row_grid(n, nch), dim3(256), 0, st, x, ldx, n, a, b, ldab, y,
                       ldy);
    OSZ_HIP(hipGetLastError());
    return OSZ_OK;
}

int osz_complex_join(const double *re, int64_t ldre, const double *im, int64_t ldim, int nch,
                     int64_t n, double *z, int64_t ldz, void *stream) {
    OSZ_REQUIRE(re && im && z, "osz_complex_join: null argument");
    OSZ_REQUIRE(nch >= 1 && nch <= 65535 && n >= 0 && ldre >= n && ldim >= n && ldz >= n,
                "osz_complex_join: bad shape");
    if (n == 0) return OSZ_OK;
    hipLaunchKernelGGL(complex_join_kernel, row_grid(n, nch), dim3(256), 0, as_stream(stream), re, ldre,
                       im, ldim, n, reinterpret_cast<double2 *>(z), ldz);
    OSZ_HIP(hipGetLastError());
    return OSZ_OK;
}

int osz_magphase(const double *z, int64_t ldz, int nch, int64_t n, double *mag, double *phase,
                 int64_t ldo, void *stream) {
    OSZ_REQUIRE(z && (mag || phase), "osz_magphase: null argument");
    OSZ_REQUIRE(nch >= 1 && nch <= 65535 && n >= 0 && ldz >= n && ldo >= n, "osz_magphase: bad shape");
    if (n == 0) return OSZ_OK;
    KernelTimer kt("magphase", as_stream(stream));
    hipLaunchKernelGGL(magphase_kernel, row_grid(n, nch), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const double2 *>(z), ldz, n, mag, phase, ldo);
    OSZ_HIP(hipGetLastError());
    return OSZ_OK;
}

int osz_simpson(const double *p, int64_t ldp, int nch, int64_t a, int64_t m, double dx, double *out,
                void *stream) {
    OSZ_REQUIRE(p && out, "osz_simpson: null argument");
    OSZ_REQUIRE(nch >= 1 && a >= 0 && m >= 1 && ldp >= a + m, "osz_simpson: bad range a=%lld m=%lld",
                (long long)a, (long long)m);
    hipLaunchKernelGGL(simpson_kernel, dim3(nch), dim3(256), 0, as_stream(stream), p, ldp, a, m, dx, out);
    OSZ_HIP(hipGetLastError());
    return OSZ_OK;
}

}  // extern "C"
