// specmix.h -- on-chip spectra for the transform lengths fft8.h does not cover.
//
// nfft = int(fs / resolution) in the reference (spectra/estimators.py:144, default
// resolution 0.5 -> nfft = 2 fs): 500, 1000, 2000, 5000, 10000 ... are products
// of 2, 3 and 5 far more often than powers of two.  One workgroup walks a run of
// segments of one channel; a segment of nfft real samples is packed into
// M = nfft / 2 complex points z[j] = y[2j] + i y[2j+1] (16 B a point, 80 KB of LDS
// at nfft = 10000), transformed by decimation-in-frequency passes
// of radix 10, 4, 2, 3, 5, 7 (in that order: the even radices run while the butterfly
// stride is long, the last passes -- stride 1 ... 25 -- are the odd ones, whose
// 48- and 80-byte lane strides spread over all banks) -- the FIRST pass in registers,
// on the detrended and windowed samples as they come from memory (head_sums /
// head_finish), the others in place in LDS, one barrier per pass -- and
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

struct alignas(16) C2 {
    double re, im;
};

constexpr int kMaxPass = 14;   // 2^13 * ... : M <= 10240 needs at most 7 radix-4/2 passes
constexpr int kAcc = 12;       // PSD sums per thread: bins t + NT m, m < kAcc
constexpr int kMaxM = 10208;   // 16 B a point and the reduction's 512 B in 160 KB of LDS (nfft <= 20 412 = 2^2 3^6 7)

struct Args {
    const double *x;        // one contiguous source: segment s starts at column s * stride
    const double *window;   // nwin
    void *out;              // SEGMENTS modes: (nseg, nch, nfreq) f64 / c128
    double *partial;        // PSD_MEAN: (nch, nruns, nfreq) sums of this launch
    const double *tw;       // W_N^j = exp(-2 pi i j / N), j < N, (re, im) pairs
    const int *pos;         // slot of Z[k] after the passes, k < M
    int64_t ldx, nseg;
    int stride, nwin, nch, nruns, N, M, npass;
    int halfcarry;          // 50 % overlap of unpadded segments, even first radix: a segment's second half
                            // is the next one's first half in the same threads (its sums are carried)
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
// cap of 128; those lengths stay on the rocFFT route.  So does a radix-16 pass for the powers
// of two -- 16 points and their 15 twiddle powers: 44-86 spilled registers in the PSD kernels.)
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
// A thread has at most ceil(10 / R) <= kMaxBf butterflies (M <= 10 NT, specmix_threads).  Their
// twiddles W_B^inner -- 16-byte loads from the table in L2 -- are requested by the kernel BEFORE
// the barrier in front of the pass (pass_twiddles: one routine for all radices, or the arrays of
// every radix's branch stay live across the barrier the compiler merges them behind), so that the
// table's latency runs beside the wait for the other waves instead of behind every butterfly's LDS
// reads (a rolled loop was a chain of L2 round trips, one per butterfly).
constexpr int kMaxBf = 5;     // butterflies of a thread per pass
constexpr int kPreBf = 3;     // of them with their twiddle requested ahead (radix >= 4: all)

struct Split {      // butterfly b = hi * div + lo by multiplication with ceil(2^32 / div) (exact: b, div < 2^16)
    int blkfast, div;
    unsigned inv;
    __device__ __forceinline__ void operator()(int b, int &blk, int &inner) const {
        const int hi = div == 1 ? b : (int)__umulhi((unsigned)b, inv);
        const int lo = b - hi * div;
        blk = blkfast ? lo : hi;
        inner = blkfast ? hi : lo;
    }
};

template <int NT>
__device__ __forceinline__ void pass_twiddles(C2 *w1, int t, int nb, int tstep, const double *tw, Split sp) {
#pragma unroll
    for (int i = 0; i < kPreBf; ++i) {
        const int b = t + i * NT;
        int blk, inner;
        sp(b < nb ? b : 0, blk, inner);
        w1[i] = *reinterpret_cast<const C2 *>(tw + 2 * (inner * tstep));   // (S = 1: inner = 0, unused)
    }
}

template <int R, int NT>
__device__ __forceinline__ void pass(C2 *z, int t, int M, int S, const C2 *w1, Split sp, int tstep, const double *tw) {
    constexpr int NI = (10 + R - 1) / R;
    static_assert(NI <= kMaxBf, "butterflies per thread");
    const int nb = M / R;
    const int B = R * S;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        asm volatile("" ::: "memory");   // one butterfly's slots in registers at a time
        if (t + i * NT < nb) {
            int blk, inner;
            sp(t + i * NT, blk, inner);
            C2 *p = z + blk * B + inner;
            C2 v[R];
#pragma unroll
            for (int q = 0; q < R; ++q) v[q] = p[q * S];
            dft<R>(v);
            if (S > 1) {
                const C2 w = i < kPreBf ? w1[i] : *reinterpret_cast<const C2 *>(tw + 2 * (inner * tstep));
                C2 wq = w;
                v[1] = cmul(v[1], wq);
#pragma unroll
                for (int q = 2; q < R; ++q) {
                    wq = cmul(wq, w);
                    v[q] = cmul(v[q], wq);
                }
            }
#pragma unroll
            for (int q = 0; q < R; ++q) p[q * S] = v[q];
        }
    }
}

// The head of a segment: samples -> trend -> window -> FIRST pass (radix R0, S0 = M / R0, one
// block), all in registers: a thread loads the R0 points b + q S0 of its butterflies b = t + i NT
// (consecutive lanes still read consecutive samples), the block sums of the trend go through the
// one barrier, and what reaches LDS is the output of the first pass -- ONE store per point where
// the raw samples, the detrended ones and the pass each made one (a 16-byte LDS store moves at
// ~80 B per clock and CU, a third of a load: the stores, not the reads or the table, are what this
// kernel's time is made of, profiles/r04_specmix_phases.txt).  Two halves around the kernel's
// barrier, the points in ONE array for all radices (with the barrier inside a radix's branch the
// compiler merges the branches behind it and keeps every radix's array alive).
constexpr int kPreBf0 = 2;     // first-pass butterflies with their twiddle requested ahead (radix >= 5: all)

// a row of nwin doubles as a raw buffer whose range check returns 0.0 beyond it: the zero padding
// of a segment (nwin < nfft) and the lanes without a butterfly cost no compare and no select, and an
// address is ONE register (a 64-bit address per load is what made this phase spill)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const double *base, int nwin) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(base), 0, nwin * 8, 0x00020000);
}
constexpr unsigned kOffRange = 0x7ffffff0u;   // a byte offset beyond every row

// first half: the block sums of the trend over the thread's points (the samples themselves are not
// kept: second half), and the twiddles of the pass requested.  The points q < R0 / 2 of a butterfly
// are the segment's first half, the others its second half (R0 even): summed apart, and with
// `full` false the first half is not read at all -- its sums came with the previous segment.
template <int R0, bool LINEAR, int NT, int HALF>
__device__ __forceinline__ void head_sums(C2 *w1, double &sum, double &lin, const double *xs,
                                          const Args &a, int t) {
    constexpr int NI = (10 + R0 - 1) / R0;
    static_assert(NI <= kMaxBf, "butterflies of a thread");
    const int S0 = a.M / R0;
    const double mid = 0.5 * (a.nwin - 1);
    const __amdgpu_buffer_rsrc_t rx = row_rsrc(xs, a.nwin);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int b = t + i * NT;
        const bool on = b < S0;
#pragma unroll
        for (int q = HALF ? R0 / 2 : 0; q < (HALF ? R0 : R0 / 2); ++q) {
            // samples 2 j and 2 j + 1 of point j = b + q S0
            const int j = b + q * S0;
            const unsigned at = on ? 16u * (unsigned)j : kOffRange;
            const double x0 = buf_load(rx, at, 0), x1 = buf_load(rx, at + 8, 0);
            sum += x0 + x1;
            if (LINEAR) lin += (2 * j - mid) * x0 + (2 * j + 1 - mid) * x1;
        }
        // W_M^b of the pass, back by the time the barrier is behind us (tstep = N / M = 2)
        if (HALF && i < kPreBf0) w1[i] = *reinterpret_cast<const C2 *>(a.tw + 4 * (on ? b : 0));
    }
}

template <bool LINEAR, int NT, int HALF>
__device__ __forceinline__ void head_sums_any(int r0, C2 *w1, double &sum, double &lin, const double *xs,
                                              const Args &a, int t) {
    if (r0 == 10) head_sums<10, LINEAR, NT, HALF>(w1, sum, lin, xs, a, t);
    else if (r0 == 4) head_sums<4, LINEAR, NT, HALF>(w1, sum, lin, xs, a, t);
    else if (r0 == 5) head_sums<5, LINEAR, NT, HALF>(w1, sum, lin, xs, a, t);
    else if (r0 == 2) head_sums<2, LINEAR, NT, HALF>(w1, sum, lin, xs, a, t);
    else if (r0 == 3) head_sums<3, LINEAR, NT, HALF>(w1, sum, lin, xs, a, t);
    else head_sums<7, LINEAR, NT, HALF>(w1, sum, lin, xs, a, t);
}

// second half, behind the barrier: the samples once more (from L2: they came by a microsecond ago;
// kept in registers across the barrier they push the PSD sums of the mean mode out into scratch),
// trend off, window on, the butterfly, one store per point
template <int R0, bool LINEAR, int NT>
__device__ __forceinline__ void head_finish(C2 *z, const C2 *w1, double mean, double slope, const double *xs,
                                            const Args &a, int t) {
    constexpr int NI = (10 + R0 - 1) / R0;
    const int S0 = a.M / R0;
    const double mid = 0.5 * (a.nwin - 1);
    const __amdgpu_buffer_rsrc_t rx = row_rsrc(xs, a.nwin);
    const __amdgpu_buffer_rsrc_t rw = row_rsrc(a.window, a.nwin);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int b = t + i * NT;
        asm volatile("" ::: "memory");   // one butterfly's points in registers at a time
        if (b < S0) {
            C2 u[R0];
#pragma unroll
            for (int q = 0; q < R0; ++q) {
                const int j = b + q * S0;
                const unsigned at = 16u * (unsigned)j;
                // (with the usual 50 % overlap a first half is read for the last time: it shall not
                // displace the halves still to come back; a hint, and one in the instruction's encoding:
                // a run-time choice between the two loads spills)
#ifdef OSZ_MIX_NO_NT     // (A/B builds only)
                const bool last_use = false;
#else
                const bool last_use = q < R0 / 2;
#endif
                const double x0 = last_use ? buf_load_nt(rx, at, 0) : buf_load(rx, at, 0);
                const double x1 = last_use ? buf_load_nt(rx, at + 8, 0) : buf_load(rx, at + 8, 0);
                const double g0 = buf_load(rw, at, 0), g1 = buf_load(rw, at + 8, 0);   // 0 in the padding
                if (LINEAR) {
                    u[q].re = (x0 - mean - slope * (2 * j - mid)) * g0;
                    u[q].im = (x1 - mean - slope * (2 * j + 1 - mid)) * g1;
                } else {
                    u[q].re = (x0 - mean) * g0;
                    u[q].im = (x1 - mean) * g1;
                }
            }
            dft<R0>(u);
            if (S0 > 1) {
                const C2 w = i < kPreBf0 ? w1[i] : *reinterpret_cast<const C2 *>(a.tw + 4 * b);
                C2 wq = w;
                u[1] = cmul(u[1], wq);
#pragma unroll
                for (int q = 2; q < R0; ++q) {
                    wq = cmul(wq, w);
                    u[q] = cmul(u[q], wq);
                }
            }
#pragma unroll
            for (int q = 0; q < R0; ++q) z[b + q * S0] = u[q];
        }
    }
}

// a value every thread of the workgroup holds alike, into scalar registers
__device__ __forceinline__ double uniform(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

template <int MODE, bool LINEAR, int NT>
__global__ __launch_bounds__(NT, 4) void specmix_kernel(Args a) {
    extern __shared__ C2 zmix[];
    constexpr int NWV = (NT + 63) / 64;
    __shared__ double red[NWV][4];
    C2 *z = zmix;
    const int t = threadIdx.x;
    const int run = blockIdx.x;
    const int c = blockIdx.y;
    const int M = a.M, N = a.N, NF = a.M + 1;
    const double *xr = a.x + (int64_t)c * a.ldx;
    const int64_t s0 = ((int64_t)run * a.nseg) / a.nruns;
    const int64_t s1 = ((int64_t)(run + 1) * a.nseg) / a.nruns;
    const double s2 = a.scale * a.scale;
    double acc[kAcc];
#pragma unroll
    for (int m = 0; m < kAcc; ++m) acc[m] = 0.0;

    double carry_sum = 0.0, carry_lin = 0.0;     // sums over the previous segment's second half
    for (int64_t s = s0; s < s1; ++s) {
        const double *xs = xr + s * (int64_t)a.stride;
        // ---- samples, trend, window and the first pass: one LDS store per point (head_sums / head_finish)
        {
            const int r0 = a.radix[0];
            C2 w0[kPreBf0];
            const bool full = !a.halfcarry || s == s0;
            // opaque thread index, again before every phase: hoisted out of the segment loop, the
            // addresses / masks / ramp values of all a thread's points would spill
            int tt = t;
            asm volatile("" : "+v"(tt));
            if (full) {
                double sum = 0.0, lin = 0.0;
                head_sums_any<LINEAR, NT, 0>(r0, w0, sum, lin, xs, a, tt);
                sum = wave_sum63(sum);
                if (LINEAR) lin = wave_sum63(lin);
                if ((t & 63) == 63) {
                    red[t >> 6][0] = sum;
                    red[t >> 6][1] = lin;
                }
            }
            {
                double sum = 0.0, lin = 0.0;
                head_sums_any<LINEAR, NT, 1>(r0, w0, sum, lin, xs, a, tt);
                sum = wave_sum63(sum);
                if (LINEAR) lin = wave_sum63(lin);
                if ((t & 63) == 63) {
                    red[t >> 6][2] = sum;
                    red[t >> 6][3] = lin;
                }
            }
            __syncthreads();   // (also: the bin reads of the previous segment are done)
            double totA = 0.0, linA = 0.0, totB = 0.0, linB = 0.0;
#pragma unroll
            for (int q = 0; q < NWV; ++q) {
                totA += red[q][0];
                totB += red[q][2];
                if (LINEAR) {
                    linA += red[q][1];
                    linB += red[q][3];
                }
            }
            if (!full) {
                // the previous segment's second half, nwin / 2 samples further left in this one
                totA = carry_sum;
                linA = carry_lin - 0.5 * a.nwin * carry_sum;
            }
            carry_sum = uniform(totB);      // (the same in every thread: scalar registers)
            if (LINEAR) carry_lin = uniform(linB);
            const double tot = totA + totB, tlin = linA + linB;
            const double mean = tot / a.nwin;
            double slope = 0.0;
            if (LINEAR) {
                const double nn = (double)a.nwin;
                const double sxx = nn * (nn * nn - 1.0) / 12.0;
                slope = sxx > 0.0 ? tlin / sxx : 0.0;
            }
            asm volatile("" : "+v"(tt));
            if (r0 == 10) head_finish<10, LINEAR, NT>(z, w0, mean, slope, xs, a, tt);
            else if (r0 == 4) head_finish<4, LINEAR, NT>(z, w0, mean, slope, xs, a, tt);
            else if (r0 == 5) head_finish<5, LINEAR, NT>(z, w0, mean, slope, xs, a, tt);
            else if (r0 == 2) head_finish<2, LINEAR, NT>(z, w0, mean, slope, xs, a, tt);
            else if (r0 == 3) head_finish<3, LINEAR, NT>(z, w0, mean, slope, xs, a, tt);
            else head_finish<7, LINEAR, NT>(z, w0, mean, slope, xs, a, tt);
        }
        int tt = t;
        asm volatile("" : "+v"(tt));
        // ---- the other passes of the M-point transform, in place
        int B = M / a.radix[0];
        for (int p = 1; p < a.npass; ++p) {
            const int r = a.radix[p];
            const int S = B / r;
            const Split sp{a.blkfast[p], a.div[p], a.inv[p]};
            C2 w1[kPreBf];
            pass_twiddles<NT>(w1, t, M / r, N / B, a.tw, sp);
            __syncthreads();
            if (r == 10) pass<10, NT>(z, t, M, S, w1, sp, N / B, a.tw);
            else if (r == 4) pass<4, NT>(z, t, M, S, w1, sp, N / B, a.tw);
            else if (r == 5) pass<5, NT>(z, t, M, S, w1, sp, N / B, a.tw);
            else if (r == 2) pass<2, NT>(z, t, M, S, w1, sp, N / B, a.tw);
            else if (r == 3) pass<3, NT>(z, t, M, S, w1, sp, N / B, a.tw);
            else pass<7, NT>(z, t, M, S, w1, sp, N / B, a.tw);
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
