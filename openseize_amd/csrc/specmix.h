// specmix.h -- on-chip spectra for the transform lengths fft8.h does not cover.
//
// nfft = int(fs / resolution) in the reference (spectra/estimators.py:144, default
// resolution 0.5 -> nfft = 2 fs): 500, 1000, 2000, 5000, 10000 ... are products
// of 2, 3 and 5 far more often than powers of two.  One workgroup walks a run of
// segments of one channel; a segment of nfft real samples is packed into
// M = nfft / 2 complex points z[j] = y[2j] + i y[2j+1] in LDS (16 B a point,
// 80 KB at nfft = 10000), transformed IN PLACE by decimation-in-frequency passes
// of radix 10, 4, 2, 3, 5, 7 (in that order: the even radices run while the butterfly
// stride is long, the last passes -- stride 1 ... 25 -- are the odd ones, whose
// 48- and 80-byte lane strides spread over all banks), one barrier per pass, and
// untangled into the nfft / 2 + 1 bins of the real transform on the way out:
//   X[k] = E[k] + W_nfft^k O[k],  E = (Z[k] + conj Z[M-k]) / 2,  O = -i (Z[k] - conj Z[M-k]) / 2.
// In-place DIF leaves Z[k] at the digit-reversed slot pos[k] (host table).
// Twiddles: one table W_nfft^j, j < nfft, in global memory (L2); a butterfly
// loads W_B^inner once and squares / multiplies its way to the other powers.
//
// HBM traffic: the samples once (overlapping halves of consecutive segments of a
// run come back from L2), the PSD partial sums once per run -- against the
// rocFFT route's staging rows (prep -> r2c -> post: 5-10x the algorithmic bytes).
#pragma once

#include "common.h"

namespace osz {
namespace mix {

struct C2 {
    double re, im;
};

constexpr int kMaxPass = 14;   // 2^13 * ... : M <= 10240 needs at most 7 radix-4/2 passes
constexpr int kAcc = 12;       // PSD sums per thread: bins t + NT m, m < kAcc
constexpr int kBatch = 4;      // sample trips whose loads are in flight together (divides kAcc)
constexpr int kMaxM = 10240;   // 160 KB of LDS

struct Args {
    const double *x;        // one contiguous source: segment s starts at column s * stride
    const double *window;   // nwin
    void *out;              // SEGMENTS modes: (nseg, nch, nfreq) f64 / c128
    double *partial;        // PSD_MEAN: (nch, nruns, nfreq) sums of this launch
    const double *tw;       // W_N^j = exp(-2 pi i j / N), j < N, (re, im) pairs
    const int *pos;         // slot of Z[k] after the passes, k < M
    int64_t ldx, nseg;
    int stride, nwin, nch, nruns, N, M, npass;
    int radix[kMaxPass];
    // per pass: how butterflies map to lanes (host: the cheaper of the two under the
    // ds_read_b128 bank model) and the divisor of that map with its 2^32 reciprocal
    int blkfast[kMaxPass];      // 0: consecutive lanes walk a block (inner fastest); 1: walk the blocks
    int div[kMaxPass];          // inner fastest: S; blocks fastest: M / B
    unsigned inv[kMaxPass];     // ceil(2^32 / div), 0 when div == 1
    double scale;
};

__device__ __forceinline__ C2 cmul(C2 a, C2 b) {
    return C2{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
__device__ __forceinline__ C2 cadd(C2 a, C2 b) { return C2{a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ C2 csub(C2 a, C2 b) { return C2{a.re - b.re, a.im - b.im}; }
// a - i b, a + i b
__device__ __forceinline__ C2 sub_i(C2 a, C2 b) { return C2{a.re + b.im, a.im - b.re}; }
__device__ __forceinline__ C2 add_i(C2 a, C2 b) { return C2{a.re - b.im, a.im + b.re}; }

// forward DFT (e^{-2 pi i / R}) of R points in registers
template <int R>
__device__ __forceinline__ void dft(C2 *v);

template <>
__device__ __forceinline__ void dft<2>(C2 *v) {
    const C2 a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
}
template <>
__device__ __forceinline__ void dft<3>(C2 *v) {
    const double h = 0.86602540378443864676;   // sin(2 pi / 3)
    const C2 t1 = cadd(v[1], v[2]);
    const C2 d = csub(v[1], v[2]);
    const C2 m1 = C2{v[0].re - 0.5 * t1.re, v[0].im - 0.5 * t1.im};
    const C2 s = C2{h * d.re, h * d.im};
    v[0] = cadd(v[0], t1);
    v[1] = sub_i(m1, s);
    v[2] = add_i(m1, s);
}
template <>
__device__ __forceinline__ void dft<4>(C2 *v) {
    const C2 a = cadd(v[0], v[2]), b = csub(v[0], v[2]);
    const C2 c = cadd(v[1], v[3]), d = csub(v[1], v[3]);
    v[0] = cadd(a, c);
    v[2] = csub(a, c);
    v[1] = sub_i(b, d);
    v[3] = add_i(b, d);
}
template <>
__device__ __forceinline__ void dft<5>(C2 *v) {
    const double c1 = 0.30901699437494742410, c2 = -0.80901699437494742410;   // cos(2 pi/5), cos(4 pi/5)
    const double s1 = 0.95105651629515357212, s2 = 0.58778525229247312917;    // sin(2 pi/5), sin(4 pi/5)
    const C2 t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]);
    const C2 t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
    const C2 m1 = C2{v[0].re + c1 * t1.re + c2 * t2.re, v[0].im + c1 * t1.im + c2 * t2.im};
    const C2 m2 = C2{v[0].re + c2 * t1.re + c1 * t2.re, v[0].im + c2 * t1.im + c1 * t2.im};
    const C2 n1 = C2{s1 * t3.re + s2 * t4.re, s1 * t3.im + s2 * t4.im};
    const C2 n2 = C2{s2 * t3.re - s1 * t4.re, s2 * t3.im - s1 * t4.im};
    v[0] = C2{v[0].re + t1.re + t2.re, v[0].im + t1.im + t2.im};
    v[1] = sub_i(m1, n1);
    v[4] = add_i(m1, n1);
    v[2] = sub_i(m2, n2);
    v[3] = add_i(m2, n2);
}

// Radix 7 (fs = 350, 700, 1400 ... Hz at the default resolution), any odd prime in fact: the
// symmetric form, t_k = v[k] + v[R-k], d_k = v[k] - v[R-k],
//   X[j], X[R-j] = (v[0] + sum_k cos(2 pi j k / R) t_k)  -/+  i sum_k sin(2 pi j k / R) d_k,
// with the table index j k mod R folded at compile time (cos is even, sin odd about R / 2).
// (Radix 11 and 13 passes were tried: 52 + 48 registers of operands spill under the kernel's
// cap of 128; those lengths stay on the rocFFT route.)
template <int R>
struct OddTw;
template <> struct OddTw<7> {
    static constexpr double c[3] = {0.6234898018587335305250049, -0.2225209339563144042889026, -0.9009688679024191262361023};
    static constexpr double s[3] = {0.7818314824680298087084445, 0.9749279121818236070181317, 0.4338837391175581204757683};
};

template <int R>
__device__ __forceinline__ void dft_odd(C2 *v) {
    constexpr int H = (R - 1) / 2;
    C2 t[H], d[H];
#pragma unroll
    for (int k = 1; k <= H; ++k) {
        t[k - 1] = cadd(v[k], v[R - k]);
        d[k - 1] = csub(v[k], v[R - k]);
    }
    const C2 v0 = v[0];
    C2 sum = v0;
#pragma unroll
    for (int k = 0; k < H; ++k) sum = cadd(sum, t[k]);
    v[0] = sum;
#pragma unroll
    for (int j = 1; j <= H; ++j) {
        C2 m = v0, n = C2{0.0, 0.0};
#pragma unroll
        for (int k = 1; k <= H; ++k) {
            const int q = (j * k) % R;                   // compile time
            const double cq = OddTw<R>::c[(q <= H ? q : R - q) - 1];
            const double sq = q <= H ? OddTw<R>::s[q - 1] : -OddTw<R>::s[R - q - 1];
            m.re = fma(cq, t[k - 1].re, m.re);
            m.im = fma(cq, t[k - 1].im, m.im);
            n.re = fma(sq, d[k - 1].re, n.re);
            n.im = fma(sq, d[k - 1].im, n.im);
        }
        v[j] = sub_i(m, n);
        v[R - j] = add_i(m, n);
    }
}
template <>
__device__ __forceinline__ void dft<7>(C2 *v) { dft_odd<7>(v); }

// 10 = 2 x 5 by the prime-factor mapping: no twiddles inside the butterfly.
// Inputs n = (5 n1 + 2 n2) mod 10, outputs k = (5 k1 + 6 k2) mod 10.
template <>
__device__ __forceinline__ void dft<10>(C2 *v) {
    C2 a[5] = {v[0], v[2], v[4], v[6], v[8]};
    C2 b[5] = {v[5], v[7], v[9], v[1], v[3]};
    dft<5>(a);
    dft<5>(b);
    v[0] = cadd(a[0], b[0]);
    v[5] = csub(a[0], b[0]);
    v[6] = cadd(a[1], b[1]);
    v[1] = csub(a[1], b[1]);
    v[2] = cadd(a[2], b[2]);
    v[7] = csub(a[2], b[2]);
    v[8] = cadd(a[3], b[3]);
    v[3] = csub(a[3], b[3]);
    v[4] = cadd(a[4], b[4]);
    v[9] = csub(a[4], b[4]);
}

// One in-place DIF pass of radix R over the M points: blocks of B = R S points,
// butterfly (blk, inner) on the slots blk B + inner + q S, output q times W_B^(inner q).
template <int R, int NT>
__device__ __forceinline__ void pass(C2 *z, int t, int M, int S, int tstep, const double *tw,
                                     int blkfast, int div, unsigned inv) {
    const int nb = M / R;
    const int B = R * S;
#pragma unroll 1
    for (int b = t; b < nb; b += NT) {
        // b = hi * div + lo by multiplication with ceil(2^32 / div) (exact: b, div < 2^16)
        const int hi = div == 1 ? b : (int)__umulhi((unsigned)b, inv);
        const int lo = b - hi * div;
        const int blk = blkfast ? lo : hi;
        const int inner = blkfast ? hi : lo;
        C2 *p = z + blk * B + inner;
        C2 v[R];
#pragma unroll
        for (int q = 0; q < R; ++q) v[q] = p[q * S];
        dft<R>(v);
        if (S > 1) {
            const int j = inner * tstep;
            const C2 w1 = C2{tw[2 * j], tw[2 * j + 1]};
            C2 wq = w1;
            v[1] = cmul(v[1], wq);
#pragma unroll
            for (int q = 2; q < R; ++q) {
                wq = cmul(wq, w1);
                v[q] = cmul(v[q], wq);
            }
        }
#pragma unroll
        for (int q = 0; q < R; ++q) p[q * S] = v[q];
    }
}

template <int MODE, bool LINEAR, int NT>
__global__ __launch_bounds__(NT, 4) void specmix_kernel(Args a) {
    extern __shared__ C2 zmix[];
    constexpr int NWV = (NT + 63) / 64;
    __shared__ double red[NWV][2];
    C2 *z = zmix;
    const int t = threadIdx.x;
    const int run = blockIdx.x;
    const int c = blockIdx.y;
    const int M = a.M, N = a.N, NF = a.M + 1;
    const double *xr = a.x + (int64_t)c * a.ldx;
    const int64_t s0 = ((int64_t)run * a.nseg) / a.nruns;
    const int64_t s1 = ((int64_t)(run + 1) * a.nseg) / a.nruns;
    const double mid = 0.5 * (a.nwin - 1);
    const double s2 = a.scale * a.scale;
    double acc[kAcc];
#pragma unroll
    for (int m = 0; m < kAcc; ++m) acc[m] = 0.0;

    for (int64_t s = s0; s < s1; ++s) {
        const double *xs = xr + s * (int64_t)a.stride;
        // ---- samples into LDS (raw), block sums for the trend.  The trip count
        // is at most kAcc (the launch picks NT so): unrolled in batches whose
        // loads are in flight together -- a rolled loop is a chain of HBM latencies.
        double sum = 0.0, lin = 0.0;
        __syncthreads();   // the bin reads of the previous segment are done
        // opaque thread index, again before every phase: hoisted out of the segment
        // loop, the addresses / masks / ramp values of all kAcc trips would spill
        int tt = t;
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int h = 0; h < kAcc; h += kBatch) {
            if (NT * h >= M) break;   // uniform
            double v0[kBatch], v1[kBatch];
            asm volatile("" : "+v"(tt));
#pragma unroll
            for (int m = 0; m < kBatch; ++m) {
                const unsigned j = tt + NT * (h + m);
                const unsigned i0 = 2 * j, i1 = 2 * j + 1;
                // clamped addresses, zeros by select: no branch around the loads
                const double a0 = xs[i0 < (unsigned)a.nwin ? i0 : 0u], a1 = xs[i1 < (unsigned)a.nwin ? i1 : 0u];
                v0[m] = i0 < (unsigned)a.nwin ? a0 : 0.0;
                v1[m] = i1 < (unsigned)a.nwin ? a1 : 0.0;
            }
#pragma unroll
            for (int m = 0; m < kBatch; ++m) {
                const int j = tt + NT * (h + m);
                const int i0 = 2 * j, i1 = 2 * j + 1;
                if (j < M) {
                    sum += v0[m] + v1[m];
                    if (LINEAR) lin += (i0 - mid) * v0[m] + (i1 - mid) * v1[m];
                    z[j] = C2{v0[m], v1[m]};
                }
            }
        }
        sum = wave_sum63(sum);
        if (LINEAR) lin = wave_sum63(lin);
        if ((t & 63) == 63) {
            red[t >> 6][0] = sum;
            red[t >> 6][1] = lin;
        }
        __syncthreads();
        double tot = 0.0, tlin = 0.0;
#pragma unroll
        for (int q = 0; q < NWV; ++q) {
            tot += red[q][0];
            if (LINEAR) tlin += red[q][1];
        }
        const double mean = tot / a.nwin;
        double slope = 0.0;
        if (LINEAR) {
            const double nn = (double)a.nwin;
            const double sxx = nn * (nn * nn - 1.0) / 12.0;
            slope = sxx > 0.0 ? tlin / sxx : 0.0;
        }
        // ---- detrend and window in place (a thread rewrites the slots it filled)
#pragma unroll
        for (int h = 0; h < kAcc; h += kBatch) {
            if (NT * h >= M) break;   // uniform
            double w0[kBatch], w1[kBatch];
            asm volatile("" : "+v"(tt));
#pragma unroll
            for (int m = 0; m < kBatch; ++m) {
                const unsigned j = tt + NT * (h + m);
                const unsigned i0 = 2 * j, i1 = 2 * j + 1;
                const double a0 = a.window[i0 < (unsigned)a.nwin ? i0 : 0u];
                const double a1 = a.window[i1 < (unsigned)a.nwin ? i1 : 0u];
                w0[m] = i0 < (unsigned)a.nwin ? a0 : 0.0;
                w1[m] = i1 < (unsigned)a.nwin ? a1 : 0.0;
            }
#pragma unroll
            for (int m = 0; m < kBatch; ++m) {
                const int j = tt + NT * (h + m);
                const int i0 = 2 * j, i1 = 2 * j + 1;
                if (j < M) {
                    C2 v = z[j];
                    if (LINEAR) {
                        v.re = (v.re - mean - slope * (i0 - mid)) * w0[m];
                        v.im = (v.im - mean - slope * (i1 - mid)) * w1[m];
                    } else {
                        v.re = (v.re - mean) * w0[m];
                        v.im = (v.im - mean) * w1[m];
                    }
                    z[j] = v;
                }
            }
        }
        // ---- M-point transform, in place
        int B = M;
        for (int p = 0; p < a.npass; ++p) {
            __syncthreads();
            const int r = a.radix[p];
            const int S = B / r;
            const int tstep = N / B;
            if (r == 10) pass<10, NT>(z, t, M, S, tstep, a.tw, a.blkfast[p], a.div[p], a.inv[p]);
            else if (r == 4) pass<4, NT>(z, t, M, S, tstep, a.tw, a.blkfast[p], a.div[p], a.inv[p]);
            else if (r == 5) pass<5, NT>(z, t, M, S, tstep, a.tw, a.blkfast[p], a.div[p], a.inv[p]);
            else if (r == 2) pass<2, NT>(z, t, M, S, tstep, a.tw, a.blkfast[p], a.div[p], a.inv[p]);
            else if (r == 3) pass<3, NT>(z, t, M, S, tstep, a.tw, a.blkfast[p], a.div[p], a.inv[p]);
            else pass<7, NT>(z, t, M, S, tstep, a.tw, a.blkfast[p], a.div[p], a.inv[p]);
            B = S;
        }
        __syncthreads();
        // ---- bins of the real transform, two per thread and trip: k and M - k come
        // out of the same two slots, X[k] = E + W^k O and X[M-k] = conj(E - W^k O)
        // (k = 0: DC and Nyquist out of Z[0]).  Opaque thread index: hoisted out of
        // the segment loop, the table addresses of all trips would spill.
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int m = 0; m < kAcc / 2; ++m) {
            const int k = tt + NT * m;
            asm volatile("" ::: "memory");   // one pair at a time: hoisted, the loads of all trips spill
            if (2 * k <= M) {
                const int kb = k == 0 ? 0 : M - k;
                const C2 za = z[a.pos[k]], zb = z[a.pos[kb]];
                const double er = 0.5 * (za.re + zb.re), ei = 0.5 * (za.im - zb.im);
                const double dr = 0.5 * (za.re - zb.re), di = 0.5 * (za.im + zb.im);
                const double orr = di, oi = -dr;                    // O = -i D
                const double wr = a.tw[2 * k], wi = a.tw[2 * k + 1];
                const double pr = wr * orr - wi * oi, pi = wr * oi + wi * orr;   // W^k O
                const double xr_ = er + pr, xi_ = ei + pi;          // X[k]
                const double yr_ = er - pr, yi_ = -(ei - pi);       // X[M - k]
                const int k2 = M - k;                               // second bin (== k when 2 k == M)
                if (MODE == OSZ_SPEC_DFT_SEGMENTS) {
                    double *o = (double *)a.out + ((s * a.nch + c) * (int64_t)NF) * 2;
                    o[2 * k] = xr_ * a.scale;
                    o[2 * k + 1] = xi_ * a.scale;
                    if (k2 != k) {
                        o[2 * k2] = yr_ * a.scale;
                        o[2 * k2 + 1] = yi_ * a.scale;
                    }
                } else {
                    const double f = k != 0 ? 2.0 * s2 : s2;        // DC and Nyquist (k = 0) are not doubled
                    const double pw = (xr_ * xr_ + xi_ * xi_) * f;
                    const double qw = k2 != k ? (yr_ * yr_ + yi_ * yi_) * f : 0.0;
                    if (MODE == OSZ_SPEC_PSD_SEGMENTS) {
                        double *o = (double *)a.out + (s * a.nch + c) * (int64_t)NF;
                        o[k] = pw;
                        if (k2 != k) o[k2] = qw;
                    } else {
                        acc[2 * m] += pw;
                        acc[2 * m + 1] += qw;
                    }
                }
            }
        }
    }
    if (MODE == OSZ_SPEC_PSD_MEAN) {
        double *o = a.partial + ((int64_t)c * a.nruns + run) * NF;
#pragma unroll
        for (int m = 0; m < kAcc / 2; ++m) {
            const int k = t + NT * m;
            if (2 * k <= M) {
                o[k] = acc[2 * m];
                if (M - k != k) o[M - k] = acc[2 * m + 1];
            }
        }
    }
}

}  // namespace mix
}  // namespace osz
