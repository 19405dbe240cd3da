// fir.hip -- K1: streaming overlap-add FFT convolution on gfx950.
//
// Replaces the hot loop of oaconvolve (reference
// src/openseize/core/numerical.py:254-298): zero-pad (:234), np.fft.rfft
// (:235), multiply by H (:238), np.fft.irfft (:241), add the previous
// overlap (:243-251, :268), keep the new one (:269).
//
// Design for MI355X:
//   - one fixed transform length, NFFT = 4096, independent of the reference's
//     nfft (65 536 / 262 144 for 256 / 1024 taps, far beyond LDS): the linear
//     convolution is segmentation independent, so the segment is chosen to
//     keep r2c -> xH -> c2r entirely on chip (fft4096.h: registers + 68 KB LDS);
//   - two consecutive real blocks ride one complex transform (a + i b): since
//     h is real, Re/Im of the inverse are the two blocks' results -- no
//     real-FFT split/merge pass and no bit reversal;
//   - a workgroup owns a RUN of consecutive blocks of one channel and keeps
//     the (ntaps-1) overlap tail in LDS from pair to pair; runs are
//     independent: each starts with a zero tail and publishes its last tail,
//     and a small seam kernel adds tails across run boundaries and across
//     pushes (the carried state of the iterator);
//   - HBM: every sample is read once (8 B, consecutive lanes -> consecutive
//     samples) and written once (8 B): 16 B per channel-sample; twiddles and
//     H (64 KB each) stay L2 resident.
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "common.h"
#include "fft4096.h"

namespace osz {

constexpr int kFirMaxTaps = 2049;  // NFFT - ntaps + 1 >= ntaps - 1

struct FirArgs {
    const double *x;
    double *y;
    int64_t ldx, ldy, n, skip;
    int wlen, step, R, nruns;
    int64_t nblocks;
    const double *H;  // [4096][2], already divided by 4096
    fft::Tables tb;
    double *tails;    // [nch][nruns][wlen-1]
};

template <bool POW>
__global__ __launch_bounds__(256) void fir_oa_kernel(FirArgs a) {
    extern __shared__ double lds[];
    double *pr = lds;
    double *pi = lds + fft::PLANE;
    double *carry = lds + 2 * fft::PLANE;  // wlen - 1 doubles

    const int t = threadIdx.x;
    const int run = blockIdx.x;
    const int c = blockIdx.y;
    const int wm1 = a.wlen - 1;
    const double *xr = a.x + (int64_t)c * a.ldx;
    double *yr = a.y + (int64_t)c * a.ldy;

    const int64_t blk0 = (int64_t)run * a.R;
    const int64_t blk1 = (run == a.nruns - 1) ? a.nblocks : blk0 + a.R;

    for (int i = t; i < wm1; i += 256) carry[i] = 0.0;
    __syncthreads();

    double re[16], im[16];
    for (int64_t blk = blk0; blk < blk1; blk += 2) {
        const int64_t start_a = blk * a.step;
        const int64_t rem_a = a.n - start_a;
        const int len_a = rem_a < a.step ? (int)rem_a : a.step;
        const int64_t start_b = start_a + len_a;
        int len_b = 0;
        if (blk + 1 < blk1) {
            const int64_t rem_b = a.n - start_b;
            len_b = rem_b < a.step ? (int)rem_b : a.step;
        }
        // ---- load two zero-padded real blocks as one complex block
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int p = 256 * j + t;
            re[j] = p < len_a ? xr[start_a + p] : 0.0;
            im[j] = p < len_b ? xr[start_b + p] : 0.0;
        }
        // ---- forward transform
        fft::f1<POW>(t, re, im, a.tb, pr, pi);
        __syncthreads();
        fft::f2_load(t, re, im, pr, pi);
        fft::f2_compute(t, re, im, a.tb);
        __syncthreads();
        fft::f2_store(t, re, im, pr, pi);
        __syncthreads();
        fft::f3(t, re, im, pr, pi);
        // ---- multiply by the filter spectrum
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = t + 256 * fft::dr(r);
            const double hr = a.H[2 * k], hi = a.H[2 * k + 1];
            const double u = re[r], v = im[r];
            re[r] = u * hr - v * hi;
            im[r] = u * hi + v * hr;
        }
        // ---- inverse transform (each thread overwrites only the slots it read)
        fft::i3(t, re, im, pr, pi);
        __syncthreads();
        fft::i2_load(t, re, im, a.tb, pr, pi);
        __syncthreads();
        fft::i2_compute_store(t, re, im, pr, pi, true);
        __syncthreads();
        fft::i1<POW>(t, re, im, a.tb, pr, pi);
        __syncthreads();
        // ---- overlap add.  re[j] = a[256 j + t], im[j] = b[256 j + t]
        double *xb = pr;  // a's tail handed to b's head
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int p = 256 * j + t;
            if (p < wm1) re[j] += carry[p];
            const int q = p - len_a;
            if (q >= 0 && q < wm1) xb[q] = re[j];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int p = 256 * j + t;
            if (p < wm1) im[j] += xb[p];
            const int q = p - len_b;
            if (q >= 0 && q < wm1) carry[q] = im[j];
        }
        // ---- write the finished samples (full-convolution positions start_a + p)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int p = 256 * j + t;
            const int64_t oa = start_a + p - a.skip;
            if (p < len_a && oa >= 0) yr[oa] = re[j];
            const int64_t ob = start_b + p - a.skip;
            if (p < len_b && ob >= 0) yr[ob] = im[j];
        }
        __syncthreads();
    }
    double *tl = a.tails + ((int64_t)c * a.nruns + run) * wm1;
    for (int i = t; i < wm1; i += 256) tl[i] = carry[i];
}

// Adds every run's published tail into the head of the next run, the carried
// state of the previous push into the head of this push, and forms the new
// carried state.  Host guarantees that, when nruns > 1, every run is at least
// wlen-1 samples long, so sources never overlap inside y.
struct SeamArgs {
    double *y;
    int64_t ldy, n, skip;
    int wlen, step, R, nruns;
    const double *tails;      // [nch][nruns][wlen-1]
    const double *state_old;  // [nch][wlen-1]
    double *state_new;        // [nch][wlen-1]
};

__global__ void fir_seam_kernel(SeamArgs a) {
    const int c = blockIdx.y;
    const int src = blockIdx.z;  // 0: carried state; s >= 1: tails of run s-1
    const int wm1 = a.wlen - 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= wm1) return;
    double *yr = a.y + (int64_t)c * a.ldy;
    if (src == a.nruns) {
        // new carried state = last run's tail (+ what is left of the old state)
        double v = a.tails[((int64_t)c * a.nruns + a.nruns - 1) * wm1 + i];
        if (a.nruns == 1 && a.n + i < wm1) v += a.state_old[(int64_t)c * wm1 + a.n + i];
        a.state_new[(int64_t)c * wm1 + i] = v;
        return;
    }
    double v;
    int64_t off;
    if (src == 0) {
        v = a.state_old[(int64_t)c * wm1 + i];
        off = 0;
    } else {
        v = a.tails[((int64_t)c * a.nruns + src - 1) * wm1 + i];
        off = (int64_t)src * a.R * a.step;
    }
    const int64_t pos = off + i;
    if (pos < a.n && pos >= a.skip) yr[pos - a.skip] += v;
}

// Process-wide twiddle tables on the device (one per process per device).
struct FftTablesDev {
    double *t1 = nullptr, *t2 = nullptr;
    int device = -1;
};

int get_fft_tables(fft::Tables &out) {
    static std::mutex mu;
    static FftTablesDev tabs;
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    OSZ_HIP(hipGetDevice(&dev));
    if (tabs.t1 == nullptr || tabs.device != dev) {
        const long double PI = acosl(-1.0L);
        std::vector<double> t1(16 * 256 * 2), t2(16 * 16 * 2);
        for (int k0 = 0; k0 < 16; ++k0)
            for (int t = 0; t < 256; ++t) {
                const long double ang = -2.0L * PI * (long double)(t * k0) / 4096.0L;
                t1[(k0 * 256 + t) * 2] = (double)cosl(ang);
                t1[(k0 * 256 + t) * 2 + 1] = (double)sinl(ang);
            }
        for (int n0 = 0; n0 < 16; ++n0)
            for (int k1 = 0; k1 < 16; ++k1) {
                const long double ang = -2.0L * PI * (long double)(n0 * k1) / 256.0L;
                t2[(n0 * 16 + k1) * 2] = (double)cosl(ang);
                t2[(n0 * 16 + k1) * 2 + 1] = (double)sinl(ang);
            }
        OSZ_HIP(hipMalloc(&tabs.t1, t1.size() * sizeof(double)));
        OSZ_HIP(hipMalloc(&tabs.t2, t2.size() * sizeof(double)));
        OSZ_HIP(hipMemcpy(tabs.t1, t1.data(), t1.size() * sizeof(double), hipMemcpyHostToDevice));
        OSZ_HIP(hipMemcpy(tabs.t2, t2.data(), t2.size() * sizeof(double), hipMemcpyHostToDevice));
        tabs.device = dev;
    }
    out.t1 = tabs.t1;
    out.t2 = tabs.t2;
    return OSZ_OK;
}

}  // namespace osz

using namespace osz;

struct osz_fir_s {
    int ntaps, nch, step;
    double *dH;        // [4096][2]
    double *dstate[2]; // ping-pong carried tails [nch][ntaps-1]
    int cur;           // which dstate holds the live tail
    double *dtails;    // workspace [nch][nruns_cap][ntaps-1]
    int nruns_cap;
    fft::Tables tb;
};

extern "C" {

int osz_fir_create(osz_fir_t *h, const double *taps, int ntaps, int nch) {
    OSZ_REQUIRE(h && taps, "osz_fir_create: null argument");
    OSZ_REQUIRE(nch >= 1, "osz_fir_create: nch=%d must be positive", nch);
    OSZ_REQUIRE(ntaps >= 1, "osz_fir_create: ntaps=%d must be positive", ntaps);
    if (ntaps > kFirMaxTaps)
        return fail(OSZ_ERR_UNSUPPORTED, "osz_fir_create: %d taps > %d supported by the on-chip "
                    "4096-point transform", ntaps, kFirMaxTaps);
    osz_fir_s *p = new osz_fir_s();
    p->ntaps = ntaps;
    p->nch = nch;
    p->step = fft::N - ntaps + 1;
    p->cur = 0;
    p->dtails = nullptr;
    p->nruns_cap = 0;
    int rc = get_fft_tables(p->tb);
    if (rc) { delete p; return rc; }
    // H[k] = sum_m h[m] W4096^(k m) / 4096, long double accumulation
    const long double PI = acosl(-1.0L);
    std::vector<long double> wc(fft::N), ws(fft::N);
    for (int j = 0; j < fft::N; ++j) {
        const long double ang = -2.0L * PI * (long double)j / (long double)fft::N;
        wc[j] = cosl(ang);
        ws[j] = sinl(ang);
    }
    std::vector<double> H(2 * fft::N);
    for (int k = 0; k < fft::N; ++k) {
        long double sr = 0, si = 0;
        for (int m = 0; m < ntaps; ++m) {
            const int j = (int)(((int64_t)k * m) & (fft::N - 1));
            sr += (long double)taps[m] * wc[j];
            si += (long double)taps[m] * ws[j];
        }
        H[2 * k] = (double)(sr / fft::N);
        H[2 * k + 1] = (double)(si / fft::N);
    }
    const size_t sb = sizeof(double) * (size_t)nch * (ntaps > 1 ? ntaps - 1 : 1);
    OSZ_HIP(hipMalloc(&p->dH, H.size() * sizeof(double)));
    OSZ_HIP(hipMalloc(&p->dstate[0], sb));
    OSZ_HIP(hipMalloc(&p->dstate[1], sb));
    OSZ_HIP(hipMemcpy(p->dH, H.data(), H.size() * sizeof(double), hipMemcpyHostToDevice));
    OSZ_HIP(hipMemset(p->dstate[0], 0, sb));
    OSZ_HIP(hipMemset(p->dstate[1], 0, sb));
    *h = p;
    return OSZ_OK;
}

int osz_fir_destroy(osz_fir_t h) {
    if (!h) return OSZ_OK;
    (void)hipFree(h->dH);
    (void)hipFree(h->dstate[0]);
    (void)hipFree(h->dstate[1]);
    (void)hipFree(h->dtails);
    delete h;
    return OSZ_OK;
}

int osz_fir_reset(osz_fir_t h, void *stream) {
    OSZ_REQUIRE(h, "osz_fir_reset: null handle");
    const size_t sb = sizeof(double) * (size_t)h->nch * (h->ntaps > 1 ? h->ntaps - 1 : 1);
    OSZ_HIP(hipMemsetAsync(h->dstate[h->cur], 0, sb, as_stream(stream)));
    return OSZ_OK;
}

int osz_fir_push(osz_fir_t h, const double *x, int64_t ldx, int64_t n, double *y, int64_t ldy,
                 int64_t skip, void *stream) {
    OSZ_REQUIRE(h && x, "osz_fir_push: null argument");
    OSZ_REQUIRE(n >= 0 && ldx >= n, "osz_fir_push: n=%lld ldx=%lld", (long long)n, (long long)ldx);
    OSZ_REQUIRE(skip >= 0 && skip <= n, "osz_fir_push: skip=%lld not in [0, n]", (long long)skip);
    OSZ_REQUIRE(skip == n || (y && ldy >= n - skip), "osz_fir_push: bad output");
    if (n == 0) return OSZ_OK;
    hipStream_t st = as_stream(stream);
    const int wm1 = h->ntaps - 1;
    const int64_t nblocks = (n + h->step - 1) / h->step;
    // run length: enough workgroups to fill 256 CUs x 2, runs of an even number of blocks
    int64_t R = (nblocks * h->nch) / 2048;
    if (R > 32) R = 32;
    if (R < 2) R = 2;
    R &= ~1LL;
    int64_t nruns = nblocks / R;
    if (nruns < 1) nruns = 1;
    if (nruns > h->nruns_cap) {
        // grow-only workspace; freeing waits for work that may still use it
        if (h->dtails) {
            OSZ_HIP(hipStreamSynchronize(st));
            OSZ_HIP(hipFree(h->dtails));
            h->dtails = nullptr;
        }
        const size_t tb = sizeof(double) * (size_t)h->nch * nruns * (wm1 > 0 ? wm1 : 1);
        hipError_t e = hipMalloc(&h->dtails, tb);
        if (e != hipSuccess) return fail(OSZ_ERR_NOMEM, "osz_fir_push: tails workspace %zu B", tb);
        h->nruns_cap = (int)nruns;
    }
    static bool attr_set = false;
    static bool pow_tw = true;    // pass-1 twiddles as products of 4 loaded powers (OSZ_FIR_T1POW=0: table)
    const size_t lds = sizeof(double) * (2 * fft::PLANE + 2048);
    if (!attr_set) {
        OSZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fir_oa_kernel<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        OSZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fir_oa_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const char *e = getenv("OSZ_FIR_T1POW");
        pow_tw = !(e && atoi(e) == 0);   // default on: measured 2.4 % faster
        attr_set = true;
    }
    FirArgs a{};
    a.x = x;
    a.y = y;
    a.ldx = ldx;
    a.ldy = ldy;
    a.n = n;
    a.skip = skip;
    a.wlen = h->ntaps;
    a.step = h->step;
    a.R = (int)R;
    a.nruns = (int)nruns;
    a.nblocks = nblocks;
    a.H = h->dH;
    a.tb = h->tb;
    a.tails = h->dtails;
    const size_t lds_used = sizeof(double) * (2 * fft::PLANE + (wm1 > 0 ? wm1 : 1));
    {
        KernelTimer kt("fir_oa", st);
        if (pow_tw)
            hipLaunchKernelGGL(fir_oa_kernel<true>, dim3((unsigned)nruns, h->nch), dim3(256),
                               lds_used, st, a);
        else
            hipLaunchKernelGGL(fir_oa_kernel<false>, dim3((unsigned)nruns, h->nch), dim3(256),
                               lds_used, st, a);
    }
    OSZ_HIP(hipGetLastError());
    if (wm1 > 0) {
        SeamArgs s{};
        s.y = y;
        s.ldy = ldy;
        s.n = n;
        s.skip = skip;
        s.wlen = h->ntaps;
        s.step = h->step;
        s.R = (int)R;
        s.nruns = (int)nruns;
        s.tails = h->dtails;
        s.state_old = h->dstate[h->cur];
        s.state_new = h->dstate[h->cur ^ 1];
        {
            KernelTimer kt("fir_seam", st);
            hipLaunchKernelGGL(fir_seam_kernel,
                               dim3((wm1 + 255) / 256, h->nch, (unsigned)nruns + 1), dim3(256), 0,
                               st, s);
        }
        OSZ_HIP(hipGetLastError());
        h->cur ^= 1;
    }
    return OSZ_OK;
}

int osz_fir_flush(osz_fir_t h, double *y, int64_t ldy, int64_t skip, int64_t drop, void *stream) {
    OSZ_REQUIRE(h, "osz_fir_flush: null handle");
    const int64_t wm1 = h->ntaps - 1;
    OSZ_REQUIRE(skip >= 0 && drop >= 0 && skip + drop <= wm1, "osz_fir_flush: skip=%lld drop=%lld",
                (long long)skip, (long long)drop);
    const int64_t cnt = wm1 - skip - drop;
    if (cnt == 0) return OSZ_OK;
    OSZ_REQUIRE(y && ldy >= cnt, "osz_fir_flush: bad output");
    OSZ_HIP(hipMemcpy2DAsync(y, ldy * sizeof(double), h->dstate[h->cur] + skip,
                             wm1 * sizeof(double), cnt * sizeof(double), h->nch,
                             hipMemcpyDeviceToDevice, as_stream(stream)));
    return OSZ_OK;
}

}  // extern "C"
