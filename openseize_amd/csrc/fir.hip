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
//     keep r2c -> xH -> c2r entirely on chip (fft4096.h: registers + 64 KB LDS);
//   - two consecutive real blocks ride one complex transform (a + i b): since
//     h is real, Re/Im of the inverse are the two blocks' results -- no
//     real-FFT split/merge pass and no bit reversal;
//   - a workgroup owns a RUN of consecutive blocks of one channel and keeps
//     the (ntaps-1) overlap tail in registers from pair to pair; runs are
//     independent: each starts with a zero tail and publishes its last tail,
//     and a small seam kernel adds tails across run boundaries and across
//     pushes (the carried state of the iterator);
//   - HBM: every sample is read once (8 B, consecutive lanes -> consecutive
//     samples) and written once (8 B): 16 B per channel-sample; the filter
//     spectrum H (64 KB) and the twiddles live in registers for a whole run.
#include <cmath>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include "common.h"
#include "fft4096.h"
#include "fir_pair.h"
#include "fir_pf_table.h"
#include "handles.h"
#include "nega_window.h"

namespace osz {

// HPRE < 0 (shipped): the 16 filter-spectrum bins of this thread stay in
// registers for the whole run -- the passes other than pass 3 have the room,
// and the hot loop then issues no table load at all (-3.5 % time against
// loading 14 of them per pair before the second barrier, HPRE = 14; HPRE = 0
// loads them where they are used).
template <int NR, int HPRE = -1, int PF = 0>
__global__ __launch_bounds__(256, 2) void fir_oa_kernel(FirArgs a) {
    extern __shared__ fft::cube::C2 cube_lds[];
    const int t = threadIdx.x, run = blockIdx.x, c = blockIdx.y;
    const int64_t blk0 = fir_run_start(run, a.nblocks, a.nruns);
    FirPair<NR, HPRE, PF> P{a, t, a.wlen - 1, a.x + (int64_t)c * a.ldx, a.y + (int64_t)c * a.ldy,
                  fir_run_start(run + 1, a.nblocks, a.nruns), cube_lds};
    fft::cube::tw_load(t, a.tb, P.tw1, P.tw2);
#pragma unroll
    for (int j = 0; j < 16 - NR; ++j) P.cr[j] = 0.0;
    if (HPRE < 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = t + 256 * fft::dr(r);
            P.Hr[HPRE < 0 ? r : 0] = a.H[2 * k];
            P.Hi[HPRE < 0 ? r : 0] = a.H[2 * k + 1];
        }
    }

    int64_t blk = blk0;
    for (; blk < P.blk1 && !P.whole(blk); blk += 2) P.any_pair(blk);
#ifdef OSZ_FIR_STAMPS
    unsigned long long rt_begin, mt_begin;
    asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_begin), "=s"(mt_begin)::"memory");
    for (int q = 0; q < 12; ++q) P.stamp_acc[q] = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(P.stamp_last)::"memory");
#endif
    if (PF != 0) {
        // the last whole pair of the run requests itself again (unused, in bounds)
        if (blk < P.blk1 && P.whole(blk)) {
            P.next_blk = (blk + 2 < P.blk1 && P.whole(blk + 2)) ? blk + 2 : blk;
            P.template fast_pair<false>(blk);
            blk += 2;
        }
        for (; blk < P.blk1 && P.whole(blk); blk += 2) {
            P.next_blk = (blk + 2 < P.blk1 && P.whole(blk + 2)) ? blk + 2 : blk;
            P.template fast_pair<true>(blk);
        }
        P.drain_spectrum();
    } else {
        for (; blk < P.blk1 && P.whole(blk); blk += 2) P.fast_pair(blk);
    }
#ifdef OSZ_FIR_STAMPS
    if (g_fir_stamps && (t & 63) == 0) {
        unsigned long long *o = g_fir_stamps + (((int64_t)c * a.nruns + run) * 4 + (t >> 6)) * 12;
        for (int q = 0; q < 12; ++q) o[q] = P.stamp_acc[q];
        // clock check: the same interval in shader-clock ticks (s_memtime) and in
        // constant 100 MHz ticks (s_memrealtime), first workgroup only
        if (c == 0 && run == 0 && t == 0) {
            unsigned long long rt1, mt1;
            asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1), "=s"(mt1)::"memory");
            g_fir_clock[0] = mt1 - mt_begin;
            g_fir_clock[1] = rt1 - rt_begin;
        }
    }
#endif
    for (; blk < P.blk1; blk += 2) P.any_pair(blk);

    double *tl = a.tails + ((int64_t)c * a.nruns + run) * P.wm1;
#pragma unroll
    for (int j = 0; j < 16 - NR; ++j) {
        const int p = 256 * j + t;
        if (p < P.wm1) tl[p] = P.cr[j];
    }
}


// ---- one real block per transform (fft::nega, fft4096.h) -------------------------------
// The pair kernel above pays the filter's tail (ntaps - 1 samples) once per 4096-sample window:
// 12 of 16 rows carry samples at 1024 taps.  Here a window is 8192 samples of ONE real block on
// the same 4096-point transform (odd frequencies, negacyclic wrap -- nothing wraps, the block is
// NB <= (8193 - ntaps) / 256 rows): 28 of 32 rows at 1024 taps.  A block's rows come in by
// LDS-DMA behind the previous block's transform (zp_request_rows, nega_window.h: into the
// wave's own pieces of the cube, read back behind the wave's own vmcnt wait), the spectrum is
// requested per block from L2, the W256 twiddle rows live in LDS.  Runs, tails and the seam
// kernel are the pair kernel's: a run starts with a zero tail and publishes its last one.
// Blocks start at multiples of 256 samples, so a block's tail rows are the next block's first
// rows in the SAME thread: ragged and cut blocks need no staging through LDS either.
template <int NB>
__global__ __launch_bounds__(256, 2) void fir_nega_kernel(FirArgs a) {
    constexpr int D = 32 - NB, S = 256 * NB, NHI = NB - 16, NP = (NB + 1) / 2;
    static_assert(NB >= 24 && NB <= 31, "rows per block");
    extern __shared__ fft::cube::C2 cube_lds[];
    fft::cube::C2 *tw2l = cube_lds + fft::cube::SLOTS;                 // [4 q][16 n0]
    const int t = threadIdx.x, run = blockIdx.x, c = blockIdx.y;
    const int wm1 = a.wlen - 1;
    const double *xr = a.x + (int64_t)c * a.ldx;
    double *yr = a.y + (int64_t)c * a.ldy;
    const int blk0 = (int)fir_run_start(run, a.nblocks, a.nruns);
    const int blk1 = (int)fir_run_start(run + 1, a.nblocks, a.nruns);
    NegaWindow<NHI> P{a, cube_lds, tw2l};
    {
        fft::cube::TwPow w2;
        fft::nega::tw_load(t, a.tb, P.tw1, w2);
        if (t < 16) {
#pragma unroll
            for (int q = 0; q < 4; ++q) tw2l[16 * q + t] = fft::cube::C2{w2.r[q], w2.i[q]};
        }
    }
    double cr[D];
#pragma unroll
    for (int j = 0; j < D; ++j) cr[j] = 0.0;
    auto whole = [&](int blk) { return (int64_t)(blk + 1) * S <= a.n && (int64_t)blk * S >= a.skip && !a.accum; };
    bool requested = false;          // this block's rows are on their way into the cube
    int younger = -1;                // vector-memory operations issued behind that request
    bool tail_in_cr = true;
    if (blk0 < blk1 && whole(blk0)) {
        zp_request_rows<NP>(xr + (int64_t)blk0 * S, NB, t, cube_lds);
        requested = true;
        younger = 0;
    }
    __syncthreads();
    for (int blk = blk0; blk < blk1; ++blk) {
        const int64_t o = (int64_t)blk * S;
        double re[16], im[16];
        int tq = t;
        asm volatile("" : "+v"(tq));
        const bool fast = whole(blk);
        const int64_t rem = a.n - o;
        const int len = rem < S ? (int)rem : S;
        if (requested) {
            if (younger == NB) asm volatile("s_waitcnt vmcnt(%0) ; osz:dma" ::"n"(NB) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) ; osz:dma" ::: "memory");
            const double *xs = reinterpret_cast<const double *>(cube_lds) + 128 * (tq >> 6) + (tq & 63);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                re[j] = xs[512 * (j >> 1) + 64 * (j & 1)];
                im[j] = j < NHI ? xs[512 * ((j + 16) >> 1) + 64 * ((j + 16) & 1)] : 0.0;
            }
        } else {
            const double *xc = xr + o + tq;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                re[j] = 256 * j + tq < len ? xc[256 * j] : 0.0;
                im[j] = (j < NHI && 4096 + 256 * j + tq < len) ? xc[4096 + 256 * j] : 0.0;
            }
        }
        requested = blk + 1 < blk1 && whole(blk + 1);
        P.template transform<NP>(t, re, im, requested ? xr + o + S : nullptr, NB);
        int tt = t;
        asm volatile("" : "+v"(tt));
        // the previous block's tail: rows NB .. 31 of its window are rows 0 .. D-1 of this one
#pragma unroll
        for (int j = 0; j < D; ++j)
            if (256 * j + tt < wm1) re[j] += cr[j];
        if (fast) {
            const unsigned lane8 = 8u * (unsigned)tt;
            const __amdgpu_buffer_rsrc_t ry = buf_rsrc(yr + (o - a.skip));
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                buf_store(re[j], ry, lane8, 2048u * j);
                if (j < NHI) buf_store(im[j], ry, lane8, 2048u * (j + 16));
            }
            younger = NB;
        } else {
            // ragged (the push's last block), cut on the left ('same' / 'valid') or accumulating
            // (a piece of a partitioned filter)
            younger = -1;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const int pp = 256 * j + tt;
                const double v = j < 16 ? re[j & 15] : im[j & 15];
                const int64_t oq = o + pp - a.skip;
                if (pp < len && oq >= 0) yr[oq] = a.accum ? yr[oq] + v : v;
            }
        }
        if (len == S) {
#pragma unroll
            for (int j = 0; j < D; ++j) cr[j] = im[NHI + j];
        } else {
            // the push ends inside this block: its tail starts at sample `len`, off the rows
            tail_in_cr = false;
            double *tl = a.tails + ((int64_t)c * a.nruns + run) * wm1;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const int q = 256 * j + tt - len;
                if (q >= 0 && q < wm1) tl[q] = j < 16 ? re[j & 15] : im[j & 15];
            }
        }
    }
    if (tail_in_cr) {
        double *tl = a.tails + ((int64_t)c * a.nruns + run) * wm1;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const int p = 256 * j + t;
            if (p < wm1) tl[p] = cr[j];
        }
    }
}

// Adds every run's published tail into the head of the next run, the carried
// state of the previous push into the head of this push, and forms the new
// carried state.  Host guarantees that, when nruns > 1, every run is at least
// wlen-1 samples long, so sources never overlap inside y.
struct SeamArgs {
    double *y;
    int64_t ldy, n, skip;
    int wlen, step, nruns;
    int64_t nblocks;
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
        off = fir_run_start(src, a.nblocks, a.nruns) * a.step;
    }
    const int64_t pos = off + i;
    if (pos < a.n && pos >= a.skip) yr[pos - a.skip] += v;
}

// ---- partitioned filters (ntaps > 2049): the taps are cut into P pieces of
// kFirPart taps; piece p is an ordinary overlap-add stream whose output is
// delayed by p*kFirPart samples.  A push accumulates all pieces into a work
// row W = [deferred sums | n new positions]; the first n columns are the
// finished samples, the rest is deferred to the next push.
__global__ void fir_w_init_kernel(double *W, int64_t ldw, const double *D, int64_t dlen) {
    const int c = blockIdx.y;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ldw;
         i += (int64_t)gridDim.x * blockDim.x)
        W[(int64_t)c * ldw + i] = i < dlen ? D[(int64_t)c * dlen + i] : 0.0;
}

__global__ void fir_w_drain_kernel(const double *W, int64_t ldw, int64_t n, int64_t skip,
                                   double *y, int64_t ldy, double *D, int64_t dlen) {
    const int c = blockIdx.y;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ldw;
         i += (int64_t)gridDim.x * blockDim.x) {
        const double v = W[(int64_t)c * ldw + i];
        if (i < n) {
            if (i >= skip) y[(int64_t)c * ldy + i - skip] = v;
        } else {
            D[(int64_t)c * dlen + (i - n)] = v;
        }
    }
}

struct FlushArgs {
    double *y;
    int64_t ldy, skip, cnt;
    const double *D;
    int64_t dlen;
    int nparts, part;          // part = taps per piece (offset unit)
    const double *state[16];   // live tail of each piece
    int wm1[16];               // its length
};

// y[c, i - skip] = deferred[i] + sum_p tail_p[i - p*part]
__global__ void fir_flush_kernel(FlushArgs a) {
    const int c = blockIdx.y;
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= a.cnt) return;
    const int64_t i = o + a.skip;
    double v = i < a.dlen ? a.D[(int64_t)c * a.dlen + i] : 0.0;
    for (int p = 0; p < a.nparts; ++p) {
        const int64_t q = i - (int64_t)p * a.part;
        if (q >= 0 && q < a.wm1[p]) v += a.state[p][(int64_t)c * a.wm1[p] + q];
    }
    a.y[(int64_t)c * a.ldy + o] = v;
}

// Twiddle tables on the device: one set per device, built on first use and
// kept for the life of the process.
struct FftTablesDev {
    double *t1 = nullptr, *t2 = nullptr, *t0 = nullptr;
};

int get_fft_tables(fft::Tables &out) {
    static std::mutex mu;
    static std::map<int, FftTablesDev> per_device;
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    OSZ_HIP(hipGetDevice(&dev));
    FftTablesDev &tabs = per_device[dev];
    if (tabs.t1 == nullptr) {
        const long double PI = acosl(-1.0L);
        std::vector<double> t1(16 * 256 * 2), t2(16 * 16 * 2), t0(256 * 2);
        for (int t = 0; t < 256; ++t) {          // W16384^t: the thread part of fft::nega's twist
            const long double ang = -PI * (long double)t / 8192.0L;
            t0[2 * t] = (double)cosl(ang);
            t0[2 * t + 1] = (double)sinl(ang);
        }
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
        OSZ_HIP(hipMalloc(&tabs.t0, t0.size() * sizeof(double)));
        OSZ_HIP(hipMemcpy(tabs.t0, t0.data(), t0.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    out.t1 = tabs.t1;
    out.t2 = tabs.t2;
    out.t0 = tabs.t0;
    return OSZ_OK;
}

}  // namespace osz

using namespace osz;

constexpr int kFirPart = 2048;   // taps per piece of a partitioned filter
constexpr int kFirMaxParts = 16; // => up to 32768 taps

static int fir_build_part(FirPart &pt, const double *taps, int ntaps, int nch) {
    pt.ntaps = ntaps;
    pt.step = ((fft::N - ntaps + 1) / 256) * 256;   // whole rows of 256: see fir_oa_kernel
    if (pt.step > 15 * 256) pt.step = 15 * 256;
    pt.cur = 0;
    pt.dH = pt.dstate[0] = pt.dstate[1] = nullptr;
    // H[k] = sum_m h[m] W4096^(k m) / 4096 in long double: an in-place radix-2
    // decimation-in-time FFT of the zero-padded taps (a direct sum costs
    // 4096 * ntaps long-double multiply-adds, ~8 ms per filter at 1024 taps)
    const long double PI = acosl(-1.0L);
    std::vector<long double> wc(fft::N / 2), ws(fft::N / 2);
    for (int j = 0; j < fft::N / 2; ++j) {
        const long double ang = -2.0L * PI * (long double)j / (long double)fft::N;
        wc[j] = cosl(ang);
        ws[j] = sinl(ang);
    }
    std::vector<long double> fr(fft::N, 0.0L), fi(fft::N, 0.0L);
    for (int m = 0; m < ntaps; ++m) {           // bit-reversed load
        unsigned r = 0;
        for (int b = 0; b < 12; ++b) r |= ((unsigned)(m >> b) & 1u) << (11 - b);
        fr[r] = (long double)taps[m];
    }
    for (int len = 2; len <= fft::N; len <<= 1) {
        const int half = len >> 1, tstep = fft::N / len;
        for (int base = 0; base < fft::N; base += len)
            for (int j = 0; j < half; ++j) {
                const long double c = wc[j * tstep], sn = ws[j * tstep];
                const int u = base + j, v = u + half;
                const long double tr = fr[v] * c - fi[v] * sn, ti = fr[v] * sn + fi[v] * c;
                fr[v] = fr[u] - tr;
                fi[v] = fi[u] - ti;
                fr[u] += tr;
                fi[u] += ti;
            }
    }
    std::vector<double> H(2 * fft::N);
    for (int k = 0; k < fft::N; ++k) {
        H[2 * k] = (double)(fr[k] / fft::N);
        H[2 * k + 1] = (double)(fi[k] / fft::N);
    }
    // the same response at the quarter-shifted bins 2 pi (j + 1/4) / 4096 for fir_nega_kernel: the
    // taps twisted by e^{-i 2 pi m / 16384}, the same transform; stored in the order fft::cube2
    // leaves the bins in ([r][t])
    pt.dHn = nullptr;
    pt.stepn = 0;
    if (ntaps >= 2 && ntaps <= kFirMaxTaps) {
        int nb = (2 * fft::N + 1 - ntaps) / 256;
        if (nb > 31) nb = 31;
        pt.stepn = 256 * nb;
        std::fill(fr.begin(), fr.end(), 0.0L);
        std::fill(fi.begin(), fi.end(), 0.0L);
        for (int m = 0; m < ntaps; ++m) {
            unsigned r = 0;
            for (int b = 0; b < 12; ++b) r |= ((unsigned)(m >> b) & 1u) << (11 - b);
            const long double ang = -2.0L * PI * (long double)m / (4.0L * fft::N);
            fr[r] = (long double)taps[m] * cosl(ang);
            fi[r] = (long double)taps[m] * sinl(ang);
        }
        for (int len = 2; len <= fft::N; len <<= 1) {
            const int half = len >> 1, tstep = fft::N / len;
            for (int base = 0; base < fft::N; base += len)
                for (int j = 0; j < half; ++j) {
                    const long double c = wc[j * tstep], sn = ws[j * tstep];
                    const int u = base + j, v = u + half;
                    const long double tr = fr[v] * c - fi[v] * sn, ti = fr[v] * sn + fi[v] * c;
                    fr[v] = fr[u] - tr;
                    fi[v] = fi[u] - ti;
                    fr[u] += tr;
                    fi[u] += ti;
                }
        }
        std::vector<double> Hn(2 * fft::N);
        for (int r = 0; r < 16; ++r)
            for (int t = 0; t < 256; ++t) {
                const int k = fft::cube2::bin(t, r);
                Hn[2 * (256 * r + t)] = (double)(fr[k] / fft::N);
                Hn[2 * (256 * r + t) + 1] = (double)(fi[k] / fft::N);
            }
        OSZ_HIP(hipMalloc(&pt.dHn, Hn.size() * sizeof(double)));
        OSZ_HIP(hipMemcpy(pt.dHn, Hn.data(), Hn.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    const size_t sb = sizeof(double) * (size_t)nch * (ntaps > 1 ? ntaps - 1 : 1);
    OSZ_HIP(hipMalloc(&pt.dH, H.size() * sizeof(double)));
    OSZ_HIP(hipMalloc(&pt.dstate[0], sb));
    OSZ_HIP(hipMalloc(&pt.dstate[1], sb));
    OSZ_HIP(hipMemcpy(pt.dH, H.data(), H.size() * sizeof(double), hipMemcpyHostToDevice));
    OSZ_HIP(hipMemset(pt.dstate[0], 0, sb));
    OSZ_HIP(hipMemset(pt.dstate[1], 0, sb));
    return OSZ_OK;
}

// one overlap-add stream: main kernel + seam kernel
static int fir_part_push(osz_fir_s *h, FirPart &pt, const double *x, int64_t ldx, int64_t n,
                         double *y, int64_t ldy, int64_t skip, int accum, hipStream_t st) {
    const int wm1 = pt.ntaps - 1;
    // one real block per transform (fir_nega_kernel) where its tables exist; OSZ_FIR_NEGA=0: the
    // pair kernel, for comparison
    static const bool nega_on = [] {
        const char *e = getenv("OSZ_FIR_NEGA");
        return !(e && e[0] == '0');
    }();
    const bool nega = nega_on && pt.dHn != nullptr && wm1 >= 1;
    const int step = nega ? pt.stepn : pt.step;
    const int64_t nblocks = (n + step - 1) / step;
    // run length: long runs (every workgroup pays for its twiddle loads, its tail
    // and its seam) as long as about 3/4 of the 512 workgroup slots stay busy --
    // measured at 16..256 channels, runs of an even number of blocks
    int64_t R = (nblocks * h->nch) / 384;
    if (R > 32) R = 32;
    if (R < 2) R = 2;
    R &= ~1LL;
    int64_t nruns = (nblocks + R - 1) / R;       // balanced runs of <= R blocks
    {
        // whole "rounds" of resident workgroups: with W workgroups on S slots the
        // launch takes ceil(W / S) rounds, so prefer a run count that wastes
        // little of the last round (slots = CUs x workgroups per CU by LDS)
        static int cus = 0;
        if (!cus) {
            int dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess &&
                hipGetDeviceProperties(&prop, dev) == hipSuccess)
                cus = prop.multiProcessorCount;
            if (cus <= 0) cus = 256;
        }
        const int64_t slots = (int64_t)cus * 2;   // 64 KB of LDS per workgroup
        int64_t best = nruns;
        double best_waste = 2.0;
        for (int64_t cand = nruns; cand < nruns + 8; ++cand) {
            const int64_t w = cand * h->nch, rounds = (w + slots - 1) / slots;
            const double waste = 1.0 - (double)w / (double)(rounds * slots);
            if (waste < best_waste - 1e-9) {
                best_waste = waste;
                best = cand;
            }
        }
        nruns = best;
    }
    // every run must hold at least one whole block (>= ntaps - 1 samples) so that
    // tails never reach past the run that follows: at most one run per pair of
    // blocks whose first block is whole
    if (nruns > nblocks / 2) nruns = nblocks / 2;
    if (nruns < 1) nruns = 1;
    const int64_t need = (int64_t)h->nch * nruns * (wm1 > 0 ? wm1 : 1);
    if (need > h->tails_cap) {
        // grow-only workspace; freeing waits for work that may still use it
        if (h->dtails) {
            OSZ_HIP(hipStreamSynchronize(st));
            OSZ_HIP(hipFree(h->dtails));
            h->dtails = nullptr;
        }
        if (hipMalloc(&h->dtails, sizeof(double) * need) != hipSuccess)
            return fail(OSZ_ERR_NOMEM, "osz_fir_push: tails workspace %lld doubles", (long long)need);
        h->tails_cap = need;
    }
    FirArgs a{};
    a.x = x;
    a.y = y;
    a.ldx = ldx;
    a.ldy = ldy;
    a.n = n;
    a.skip = skip;
    a.wlen = pt.ntaps;
    a.step = step;
    a.R = (int)R;
    a.nruns = (int)nruns;
    a.accum = accum;
    a.nblocks = nblocks;
    a.H = nega ? pt.dHn : pt.dH;
    a.tb = h->tb;
    a.tails = h->dtails;
    if (nega) {
        using kern_t = void (*)(FirArgs);
        static const kern_t kn[8] = {fir_nega_kernel<24>, fir_nega_kernel<25>, fir_nega_kernel<26>,
                                     fir_nega_kernel<27>, fir_nega_kernel<28>, fir_nega_kernel<29>,
                                     fir_nega_kernel<30>, fir_nega_kernel<31>};
        const int nb = step / 256;
        const size_t lds = sizeof(fft::cube::C2) * fft::cube::SLOTS + 1024;
        OSZ_DYN_LDS(kn[nb - 24], lds);
        KernelTimer kt("fir_oa", st);
        hipLaunchKernelGGL(kn[nb - 24], dim3((unsigned)nruns, h->nch), dim3(256), lds, st, a);
    } else {
        using kern_t = void (*)(FirArgs);
        // rows per block: 8 (2049 taps) .. 15 (<= 257 taps)
        static const kern_t kerns0[8] = {fir_oa_kernel<8>,  fir_oa_kernel<9>,  fir_oa_kernel<10>,
                                         fir_oa_kernel<11>, fir_oa_kernel<12>, fir_oa_kernel<13>,
                                         fir_oa_kernel<14>, fir_oa_kernel<15>};
        // 1: next pair's samples requested behind the spectrum multiply (FirPair::nx);
        // 2: and the next pair's spectrum ahead of this pair's stores (FirPair::Hn)
        static const kern_t kerns1[8] = {
            fir_oa_kernel<8, 16, 1>,  fir_oa_kernel<9, 16, 1>,  fir_oa_kernel<10, 16, 1>,
            fir_oa_kernel<11, 16, 1>, fir_oa_kernel<12, 16, 1>, fir_oa_kernel<13, 16, 1>,
            fir_oa_kernel<14, 16, 1>, fir_oa_kernel<15, 16, 1>};
        static const kern_t kerns2[8] = {
            fir_oa_kernel<8, 16, 2>,  fir_oa_kernel<9, 16, 2>,  fir_oa_kernel<10, 16, 2>,
            fir_oa_kernel<11, 16, 2>, fir_oa_kernel<12, 16, 2>, fir_oa_kernel<13, 16, 2>,
            fir_oa_kernel<14, 16, 2>, fir_oa_kernel<15, 16, 2>};
        // Which variant a block height runs by default: the deepest one whose
        // registers with a load in flight the compiler neither spills nor reloads
        // (benchmarks/check_async_regions.py on the assembly of THIS build;
        // tests/test_fir_async.py fails when the table and the assembly disagree).
        static const int kDefaultPf[8] = OSZ_FIR_PF_TABLE;
        const int nr = pt.step / 256;
        const int pf = kDefaultPf[nr - 8];
        const kern_t *kerns = pf == 2 ? kerns2 : pf == 1 ? kerns1 : kerns0;
        size_t lds = sizeof(fft::cube::C2) * fft::cube::SLOTS;
        OSZ_DYN_LDS(kerns[nr - 8], lds);
        KernelTimer kt("fir_oa", st);
        hipLaunchKernelGGL(kerns[nr - 8], dim3((unsigned)nruns, h->nch), dim3(256), lds, st, a);
    }
    OSZ_HIP(hipGetLastError());
    if (wm1 > 0) {
        SeamArgs s{};
        s.y = y;
        s.ldy = ldy;
        s.n = n;
        s.skip = skip;
        s.wlen = pt.ntaps;
        s.step = step;
        s.nruns = (int)nruns;
        s.nblocks = nblocks;
        s.tails = h->dtails;
        s.state_old = pt.dstate[pt.cur];
        s.state_new = pt.dstate[pt.cur ^ 1];
        {
            KernelTimer kt("fir_seam", st);
            hipLaunchKernelGGL(fir_seam_kernel,
                               dim3((wm1 + 255) / 256, h->nch, (unsigned)nruns + 1), dim3(256), 0,
                               st, s);
        }
        OSZ_HIP(hipGetLastError());
        pt.cur ^= 1;
    }
    return OSZ_OK;
}

extern "C" {

int osz_fir_create(osz_fir_t *h, const double *taps, int ntaps, int nch) {
    OSZ_REQUIRE(h && taps, "osz_fir_create: null argument");
    OSZ_REQUIRE(nch >= 1 && nch <= 65535, "osz_fir_create: nch=%d not in [1, 65535]", nch);
    OSZ_REQUIRE(ntaps >= 1, "osz_fir_create: ntaps=%d must be positive", ntaps);
    const int nparts = ntaps <= kFirMaxTaps ? 1 : (ntaps + kFirPart - 1) / kFirPart;
    if (nparts > kFirMaxParts)
        return fail(OSZ_ERR_UNSUPPORTED, "osz_fir_create: %d taps > %d supported", ntaps,
                    kFirMaxParts * kFirPart);
    osz_fir_s *p = new osz_fir_s();
    p->ntaps = ntaps;
    p->nch = nch;
    p->device = 0;
    (void)hipGetDevice(&p->device);
    p->dtails = p->dD = p->dW = nullptr;
    p->spec = nullptr;
    p->zp = nullptr;
    p->htaps.assign(taps, taps + ntaps);
    p->tails_cap = p->w_cap = 0;
    p->dlen = 0;
    int rc = get_fft_tables(p->tb);
    if (rc) { delete p; return rc; }
    p->parts.resize(nparts);
    for (int q = 0; q < nparts; ++q) {
        const int off = nparts == 1 ? 0 : q * kFirPart;
        const int len = nparts == 1 ? ntaps : (ntaps - off < kFirPart ? ntaps - off : kFirPart);
        rc = fir_build_part(p->parts[q], taps + off, len, nch);
        if (rc) {
            p->parts.resize(q + 1);   // the parts built so far (null pointers are fine to free)
            (void)osz_fir_destroy(p);
            return rc;
        }
    }
    if (nparts > 1) {
        p->dlen = (int64_t)(nparts - 1) * kFirPart;
        if (hipMalloc(&p->dD, sizeof(double) * (size_t)nch * p->dlen) != hipSuccess) {
            (void)osz_fir_destroy(p);
            return fail(OSZ_ERR_NOMEM, "osz_fir_create: deferred sums (%lld doubles)",
                        (long long)nch * p->dlen);
        }
        OSZ_HIP(hipMemset(p->dD, 0, sizeof(double) * (size_t)nch * p->dlen));
    }
    *h = p;
    return OSZ_OK;
}

int osz_fir_destroy(osz_fir_t h) {
    if (!h) return OSZ_OK;
    spec_unlink(h->spec);
    zp_unlink(h->zp);
    for (auto &pt : h->parts) {
        (void)hipFree(pt.dH);
        (void)hipFree(pt.dHn);
        (void)hipFree(pt.dstate[0]);
        (void)hipFree(pt.dstate[1]);
    }
    (void)hipFree(h->dtails);
    (void)hipFree(h->dD);
    (void)hipFree(h->dW);
    delete h;
    return OSZ_OK;
}

int osz_fir_reset(osz_fir_t h, void *stream) {
    OSZ_REQUIRE(h, "osz_fir_reset: null handle");
    hipStream_t st = as_stream(stream);
    {
        int rc = spec_touch(h->spec, st);
        if (rc) return rc;
    }
    for (auto &pt : h->parts)
        OSZ_HIP(hipMemsetAsync(pt.dstate[pt.cur], 0,
                               sizeof(double) * (size_t)h->nch * (pt.ntaps > 1 ? pt.ntaps - 1 : 1),
                               st));
    if (h->dD) OSZ_HIP(hipMemsetAsync(h->dD, 0, sizeof(double) * (size_t)h->nch * h->dlen, st));
    return OSZ_OK;
}

// ---- checkpoint / resume: the carried tail of every piece, then the deferred sums
int64_t osz_fir_state_size(osz_fir_t h) {
    if (!h) return -1;
    int64_t n = (int64_t)h->nch * h->dlen;
    for (auto &pt : h->parts) n += (int64_t)h->nch * (pt.ntaps - 1);
    return n;
}

static int fir_state_copy(osz_fir_t h, double *state, bool out, hipStream_t st) {
    const bool dev = on_device(state);
    double *p = state;
    auto copy = [&](double *own, size_t n) -> int {
        if (out)
            OSZ_HIP(hipMemcpyAsync(p, own, sizeof(double) * n, dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
        else
            OSZ_HIP(hipMemcpyAsync(own, p, sizeof(double) * n, dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
        p += n;
        return OSZ_OK;
    };
    for (auto &pt : h->parts) {
        const size_t n = (size_t)h->nch * (pt.ntaps - 1);
        if (n == 0) continue;
        int rc = copy(pt.dstate[pt.cur], n);
        if (rc) return rc;
    }
    if (h->dlen > 0) {
        int rc = copy(h->dD, (size_t)h->nch * h->dlen);
        if (rc) return rc;
    }
    if (!dev) OSZ_HIP(hipStreamSynchronize(st));
    return OSZ_OK;
}

int osz_fir_get_state(osz_fir_t h, double *state, void *stream) {
    OSZ_REQUIRE(h && state, "osz_fir_get_state: null argument");
    OSZ_SAME_DEVICE(h, "osz_fir_get_state");
    {
        int rc = spec_settle(h->spec, as_stream(stream));
        if (rc) return rc;
    }
    return fir_state_copy(h, state, true, as_stream(stream));
}

int osz_fir_set_state(osz_fir_t h, const double *state, void *stream) {
    OSZ_REQUIRE(h && state, "osz_fir_set_state: null argument");
    OSZ_SAME_DEVICE(h, "osz_fir_set_state");
    {
        int rc = spec_touch(h->spec, as_stream(stream));
        if (rc) return rc;
    }
    return fir_state_copy(h, const_cast<double *>(state), false, as_stream(stream));
}

int osz_fir_push(osz_fir_t h, const double *x, int64_t ldx, int64_t n, double *y, int64_t ldy,
                 int64_t skip, void *stream) {
    OSZ_REQUIRE(h && x, "osz_fir_push: null argument");
    OSZ_REQUIRE(n >= 0 && ldx >= n, "osz_fir_push: n=%lld ldx=%lld", (long long)n, (long long)ldx);
    OSZ_REQUIRE(skip >= 0 && skip <= n, "osz_fir_push: skip=%lld not in [0, n]", (long long)skip);
    OSZ_REQUIRE(skip == n || (y && ldy >= n - skip), "osz_fir_push: bad output");
    if (n == 0) return OSZ_OK;
    OSZ_SAME_DEVICE(h, "osz_fir_push");
    {
        int rc = spec_touch(h->spec, as_stream(stream));
        if (rc) return rc;
    }
    return fir_push_raw(h, x, ldx, n, y, ldy, skip, as_stream(stream));
}

}  // extern "C"

int osz::fir_push_raw(osz_fir_s *h, const double *x, int64_t ldx, int64_t n, double *y, int64_t ldy,
                      int64_t skip, hipStream_t st) {
    if (h->parts.size() == 1)
        return fir_part_push(h, h->parts[0], x, ldx, n, y, ldy, skip, 0, st);
    // partitioned: accumulate every piece into W = [deferred | n new positions]
    const int64_t ldw = n + h->dlen;
    const int64_t need = (int64_t)h->nch * ldw;
    if (need > h->w_cap) {
        if (h->dW) {
            OSZ_HIP(hipStreamSynchronize(st));
            OSZ_HIP(hipFree(h->dW));
            h->dW = nullptr;
        }
        if (hipMalloc(&h->dW, sizeof(double) * need) != hipSuccess)
            return fail(OSZ_ERR_NOMEM, "osz_fir_push: work rows %lld doubles", (long long)need);
        h->w_cap = need;
    }
    int64_t bx = (ldw + 255) / 256;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(fir_w_init_kernel, dim3((unsigned)bx, h->nch), dim3(256), 0, st, h->dW, ldw,
                       h->dD, h->dlen);
    OSZ_HIP(hipGetLastError());
    for (size_t q = 0; q < h->parts.size(); ++q) {
        int rc = fir_part_push(h, h->parts[q], x, ldx, n, h->dW + (int64_t)q * kFirPart, ldw, 0, 1,
                               st);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(fir_w_drain_kernel, dim3((unsigned)bx, h->nch), dim3(256), 0, st, h->dW, ldw,
                       n, skip, y, ldy, h->dD, h->dlen);
    OSZ_HIP(hipGetLastError());
    return OSZ_OK;
}

extern "C" {

int osz_fir_flush(osz_fir_t h, double *y, int64_t ldy, int64_t skip, int64_t drop, void *stream) {
    OSZ_REQUIRE(h, "osz_fir_flush: null handle");
    {
        int rc = spec_settle(h->spec, as_stream(stream));
        if (rc) return rc;
    }
    const int64_t wm1 = h->ntaps - 1;
    OSZ_REQUIRE(skip >= 0 && drop >= 0 && skip + drop <= wm1, "osz_fir_flush: skip=%lld drop=%lld",
                (long long)skip, (long long)drop);
    const int64_t cnt = wm1 - skip - drop;
    if (cnt == 0) return OSZ_OK;
    OSZ_REQUIRE(y && ldy >= cnt, "osz_fir_flush: bad output");
    FlushArgs a{};
    a.y = y;
    a.ldy = ldy;
    a.skip = skip;
    a.cnt = cnt;
    a.D = h->dD;
    a.dlen = h->dlen;
    a.nparts = (int)h->parts.size();
    a.part = kFirPart;
    for (int q = 0; q < a.nparts; ++q) {
        a.state[q] = h->parts[q].dstate[h->parts[q].cur];
        a.wm1[q] = h->parts[q].ntaps - 1;
    }
    hipLaunchKernelGGL(fir_flush_kernel, dim3((unsigned)((cnt + 255) / 256), h->nch), dim3(256), 0,
                       as_stream(stream), a);
    OSZ_HIP(hipGetLastError());
    return OSZ_OK;
}

}  // extern "C"
