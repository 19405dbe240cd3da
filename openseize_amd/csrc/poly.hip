// poly.hip -- K4: streaming polyphase rational resampler on gfx950.
//
// Replaces scipy.signal.resample_poly(padded, L, M, window=h) as the reference
// calls it per chunk (src/openseize/core/numerical.py:610, :631) together
// with the prior/next-chunk overhang padding (:590-632).  The chunk machinery
// reproduces the global definition
//     out[j] = sum_k L*h[k] * xup[j*M + half - k],   half = (ntaps-1)/2,
// xup = x zero-stuffed by L, zeros outside, ceil(N*L/M) outputs
// (resampling/resampling.py:91), which is what this kernel evaluates directly:
// only the taps k = (j*M + half) mod L, +L, +2L, ... hit non-zero samples.
// One thread per output sample; consecutive lanes read input windows that
// overlap by all but M/L samples, so the reads are L1/L2 hits and HBM sees
// each input once (8 B) and each output once (8*L/M B).
// The handle carries the last `hist` input samples across pushes.
#include <vector>

#include "common.h"

namespace osz {

struct PolyArgs {
    const double *x;      // current chunk (nch, n)
    const double *hist;   // (nch, H): samples [nin - H, nin)
    double *y;
    const double *hL;     // L * h[k]
    int64_t ldx, ldy;
    int64_t nin;          // samples consumed before this push
    int64_t navail;       // nin + n
    int64_t j0, j1;       // outputs [j0, j1) are produced
    int m, L, M, H, half;
};

__global__ __launch_bounds__(256) void poly_kernel(PolyArgs a) {
    const int c = blockIdx.y;
    const double *xr = a.x + (int64_t)c * a.ldx;
    const double *hr = a.hist + (int64_t)c * a.H;
    double *yr = a.y + (int64_t)c * a.ldy;
    for (int64_t j = a.j0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < a.j1;
         j += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = j * a.M + a.half;
        int k = (int)(t % a.L);
        int64_t i = (t - k) / a.L;  // input index for tap k; decreases by 1 per L taps
        double acc = 0.0;
        for (; k < a.m && i >= 0; k += a.L, --i) {
            if (i >= a.navail) continue;
            const double v = i >= a.nin ? xr[i - a.nin] : (i >= a.nin - a.H ? hr[i - (a.nin - a.H)] : 0.0);
            acc = fma(a.hL[k], v, acc);
        }
        yr[j - a.j0] = acc;
    }
}

// LDS-tiled polyphase kernel.  Output j only meets the taps k = phi + L k',
// phi = (j M + half) mod L, and phi depends on j mod L alone, so the outputs
// of one residue class r = j mod L form a plain decimating FIR
//     out[jf + L q] = sum_k' hL[phi + L k'] * x[itop + M q - k'],
//     itop = (jf M + half - phi) / L,
// with its own sub-filter.  Grid (tiles, nch, L): a 256-thread workgroup
// produces 256*R consecutive outputs of ONE residue class of one channel: the
// input window (256 R M + ceil(m/L) - M samples) is staged once in LDS with
// coalesced loads; thread t accumulates R outputs reading its taps from LDS
// (lane stride M doubles: conflict free for odd M) while the coefficient of
// each tap is wave-uniform (scalar load).  For decimation (L = 1, the EEG case)
// HBM sees every input once and every output once; for L > 1 the L residue
// classes re-read the same window through L2.
template <int R>
__global__ __launch_bounds__(256) void poly_phase_kernel(PolyArgs a,
                                                         const double *__restrict__ hL) {
    extern __shared__ double win[];
    const int c = blockIdx.y;
    const int t = threadIdx.x;
    const int r = blockIdx.z;                           // residue class j mod L
    const double *xr = a.x + (int64_t)c * a.ldx;
    const double *hr = a.hist + (int64_t)c * a.H;
    double *yr = a.y + (int64_t)c * a.ldy;
    // first output of this class at or after j0, and the class' sub-filter
    int64_t jf = a.j0 + (((int64_t)r - a.j0) % a.L + a.L) % a.L;
    const int phi = (int)(((int64_t)r * a.M + a.half) % a.L);
    const int msub = phi < a.m ? (a.m - phi + a.L - 1) / a.L : 0;
    const int64_t nq_all = jf < a.j1 ? (a.j1 - jf + a.L - 1) / a.L : 0;
    const int64_t qt = (int64_t)blockIdx.x * (256 * R);
    if (qt >= nq_all) return;
    const int nj = (int)((nq_all - qt) < (256 * R) ? (nq_all - qt) : (256 * R));
    jf += qt * a.L;                                     // first output of this tile
    const int64_t itop = (jf * a.M + a.half - phi) / a.L;
    // window: inputs [i0, i0 + wlen)
    const int64_t i0 = itop - (msub - 1);
    const int wlen = (nj - 1) * a.M + msub;
    for (int q = t; q < wlen; q += 256) {
        const int64_t i = i0 + q;
        double v = 0.0;
        if (i >= 0 && i < a.navail)
            v = i >= a.nin ? xr[i - a.nin] : (i >= a.nin - a.H ? hr[i - (a.nin - a.H)] : 0.0);
        win[q] = v;
    }
    __syncthreads();
    // output q = t + 256 s reads win[q M + (msub - 1) - k']
    double acc[R];
    const double *base[R];
#pragma unroll
    for (int s = 0; s < R; ++s) {
        acc[s] = 0.0;
        const int o = t + 256 * s;
        base[s] = win + (o < nj ? o : 0) * a.M + (msub - 1);
    }
    const double *hs = hL + phi;
#pragma unroll 4
    for (int k = 0; k < msub; ++k) {
        const double ck = hs[(int64_t)k * a.L];
#pragma unroll
        for (int s = 0; s < R; ++s) acc[s] = fma(ck, base[s][-k], acc[s]);
    }
#pragma unroll
    for (int s = 0; s < R; ++s) {
        const int o = t + 256 * s;
        if (o < nj) yr[(jf - a.j0) + (int64_t)o * a.L] = acc[s];
    }
}

// newhist = last H samples of (hist ++ x[0:n])
__global__ void poly_hist_kernel(const double *x, int64_t ldx, int64_t n, const double *hist,
                                 double *newhist, int H) {
    const int c = blockIdx.y;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= H) return;
    const int64_t rel = n - H + q;  // index relative to the start of x
    double v;
    if (rel >= 0)
        v = x[(int64_t)c * ldx + rel];
    else
        v = (rel + H >= 0) ? hist[(int64_t)c * H + rel + H] : 0.0;
    newhist[(int64_t)c * H + q] = v;
}

}  // namespace osz

using namespace osz;

struct osz_poly_s {
    int m, L, M, nch, H, half;
    double *dhL;
    double *dhist[2];
    int cur;
    int64_t nin, nout;
};

static int64_t ceil_div(int64_t a, int64_t b) { return a >= 0 ? (a + b - 1) / b : -((-a) / b); }

static int64_t poly_end(const osz_poly_s *h, int64_t navail, int final_) {
    // outputs j with all needed inputs available: j*M + half < navail*L
    int64_t e = final_ ? ceil_div(navail * h->L, h->M) : ceil_div(navail * h->L - h->half, h->M);
    if (e < h->nout) e = h->nout;
    return e;
}

extern "C" {

int osz_poly_create(osz_poly_t *h, const double *taps, int ntaps, int L, int M, int nch) {
    OSZ_REQUIRE(h && taps, "osz_poly_create: null argument");
    OSZ_REQUIRE(ntaps >= 1 && L >= 1 && M >= 1 && nch >= 1 && nch <= 65535,
                "osz_poly_create: bad sizes (ntaps=%d L=%d M=%d nch=%d; nch <= 65535)", ntaps, L, M, nch);
    osz_poly_s *p = new osz_poly_s();
    p->m = ntaps;
    p->L = L;
    p->M = M;
    p->nch = nch;
    p->half = (ntaps - 1) / 2;
    p->H = (ntaps - 1 + L - 1) / L + 1;
    p->cur = 0;
    p->nin = p->nout = 0;
    std::vector<double> hL(ntaps);
    for (int k = 0; k < ntaps; ++k) hL[k] = (double)L * taps[k];
    const size_t hb = sizeof(double) * (size_t)nch * p->H;
    OSZ_HIP(hipMalloc(&p->dhL, sizeof(double) * ntaps));
    OSZ_HIP(hipMalloc(&p->dhist[0], hb));
    OSZ_HIP(hipMalloc(&p->dhist[1], hb));
    OSZ_HIP(hipMemcpy(p->dhL, hL.data(), sizeof(double) * ntaps, hipMemcpyHostToDevice));
    OSZ_HIP(hipMemset(p->dhist[0], 0, hb));
    OSZ_HIP(hipMemset(p->dhist[1], 0, hb));
    *h = p;
    return OSZ_OK;
}

int osz_poly_destroy(osz_poly_t h) {
    if (!h) return OSZ_OK;
    (void)hipFree(h->dhL);
    (void)hipFree(h->dhist[0]);
    (void)hipFree(h->dhist[1]);
    delete h;
    return OSZ_OK;
}

int osz_poly_reset(osz_poly_t h, void *stream) {
    OSZ_REQUIRE(h, "osz_poly_reset: null handle");
    h->nin = h->nout = 0;
    OSZ_HIP(hipMemsetAsync(h->dhist[h->cur], 0, sizeof(double) * (size_t)h->nch * h->H,
                           as_stream(stream)));
    return OSZ_OK;
}

int64_t osz_poly_out_count(osz_poly_t h, int64_t n, int final_) {
    if (!h || n < 0) return -1;
    return poly_end(h, h->nin + n, final_) - h->nout;
}

int osz_poly_push(osz_poly_t h, const double *x, int64_t ldx, int64_t n, int final_, double *y,
                  int64_t ldy, int64_t *n_out, void *stream) {
    OSZ_REQUIRE(h, "osz_poly_push: null handle");
    OSZ_REQUIRE(n >= 0 && (n == 0 || (x && ldx >= n)), "osz_poly_push: bad input");
    hipStream_t st = as_stream(stream);
    const int64_t navail = h->nin + n;
    const int64_t j1 = poly_end(h, navail, final_);
    const int64_t cnt = j1 - h->nout;
    OSZ_REQUIRE(cnt == 0 || (y && ldy >= cnt), "osz_poly_push: output too small for %lld samples",
                (long long)cnt);
    if (cnt > 0) {
        PolyArgs a{};
        a.x = x ? x : h->dhist[h->cur];
        a.hist = h->dhist[h->cur];
        a.y = y;
        a.hL = h->dhL;
        a.ldx = ldx;
        a.ldy = ldy;
        a.nin = h->nin;
        a.navail = navail;
        a.j0 = h->nout;
        a.j1 = j1;
        a.m = h->m;
        a.L = h->L;
        a.M = h->M;
        a.H = h->H;
        a.half = h->half;
        constexpr int R = 4;
        const int msub = (h->m + h->L - 1) / h->L;
        const size_t lds = sizeof(double) * ((size_t)(256 * R - 1) * h->M + msub);
        if (lds <= 64 * 1024 && h->L <= 64) {
            // LDS-tiled path: 256*R outputs of one residue class per workgroup
            const int64_t nq = (cnt + h->L - 1) / h->L + 1;
            const int64_t bx = (nq + 256 * R - 1) / (256 * R);
            KernelTimer kt("poly_phase", st);
            hipLaunchKernelGGL(poly_phase_kernel<R>, dim3((unsigned)bx, h->nch, h->L), dim3(256),
                               lds, st, a, h->dhL);
        } else {
            int64_t bx = (cnt + 255) / 256;
            if (bx > 4096) bx = 4096;
            KernelTimer kt("poly", st);
            hipLaunchKernelGGL(poly_kernel, dim3((unsigned)bx, h->nch), dim3(256), 0, st, a);
        }
        OSZ_HIP(hipGetLastError());
    }
    if (n > 0) {
        hipLaunchKernelGGL(poly_hist_kernel, dim3((h->H + 255) / 256, h->nch), dim3(256), 0, st, x,
                           ldx, n, h->dhist[h->cur], h->dhist[h->cur ^ 1], h->H);
        OSZ_HIP(hipGetLastError());
        h->cur ^= 1;
    }
    h->nin = navail;
    h->nout = j1;
    if (n_out) *n_out = cnt;
    return OSZ_OK;
}

}  // extern "C"
