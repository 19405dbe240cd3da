// fft8.h -- power-of-two complex float64 FFTs of N = 512 ... 8192 points for ONE
// workgroup of N / 8 threads, 8 points per thread in registers, radix-8 passes
// with in-place LDS exchanges.  Used by the generic on-chip spectra kernel
// (spec.hip: Welch / STFT / periodogram at nfft = 512 ... 8192) -- the
// companion of fft4096.h (256 threads x 16 points), at half the registers per
// thread so that twice the waves share a CU.
//
// Decimation in frequency, in place.  Index bits (L = log2 N) are consumed three
// at a time from the top: stage s transforms the bit group [P, P + 3) with
// P = L - 3, L - 6, ...; when L is not a multiple of 3 the last stage is a
// radix-2 or radix-4 butterfly over the lowest 1 or 2 bits (its thread still
// holds the 8 elements of bits [0, 3)).  In stage P thread `tid` holds, at
// register r, the element
//     idx(tid, r, P) = ((tid >> P) << (P + 3)) | (r << P) | (tid & (2^P - 1)),
// reads its 8 slots, transforms, multiplies output digit k by
// W_{2^(P+3)}^(k * low), low = tid & (2^P - 1), and writes the SAME slots back:
// one workgroup barrier per exchange, nothing else.  The first stage takes its
// input from registers -- x[(N/8) r + tid], what a coalesced global load
// delivers -- and the last one leaves its output in registers: position
// idx(tid, r, 0) then holds X[k] for k = revdigits(idx) (digit d_s of stage s
// becomes digit s of k, least significant first).  The inverse runs the stages
// backwards (decimation in time, conjugated twiddles, unnormalised) from that
// order back to x[(N/8) r + tid].
//
// LDS: N interleaved complex slots (16 N bytes), slot swz(idx) with
//     swz(i) = i ^ ((i >> 3) & 7) ^ (((i >> 6) & 1) << 3)
// which makes every stage's reads (ds_read_b128, 16-lane groups of the guide)
// and writes (ds_write_b128, 8-lane groups) conflict free for all five sizes;
// checked by brute force in tests/host/fft8_host_check.cpp together with the
// arithmetic (the stage functions are __host__ __device__).
//
// Twiddles: one table W_8192^j, j < 1024, serves every N (W_N^j = W_8192^(j
// 8192 / N)); a thread keeps one base W^low per stage and forms W^2 .. W^7 on
// the spot (squarings and products: 6 complex multiplies).
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define OSZ8_HD __host__ __device__ __forceinline__
#else
#define OSZ8_HD inline
#endif

namespace osz {
namespace fft8 {

constexpr int kTabN = 8192;              // the twiddle table is W_8192^j
constexpr int kTabLen = kTabN / 8;       // j < 1024

struct alignas(16) C2 {
    double re, im;
};

constexpr int ilog2(int n) { return n <= 1 ? 0 : 1 + ilog2(n >> 1); }

OSZ8_HD int swz(int i) { return i ^ ((i >> 3) & 7) ^ (((i >> 6) & 1) << 3); }

// second layout, for the natural-order exchange after a forward transform
// (writes at k = revdigits(idx), reads at k and N - k with k lane-contiguous)
OSZ8_HD int swz_nat(int i) { return i ^ ((i >> 3) & 7) ^ ((i >> 6) & 15); }

template <int P>
OSZ8_HD int idx_of(int tid, int r) {
    return ((tid >> P) << (P + 3)) | (r << P) | (tid & ((1 << P) - 1));
}

// frequency held at position i after the forward transform of 2^L points
template <int L>
OSZ8_HD int revdigits(int i) {
    int k = 0, sh = 0;
    for (int pos = L; pos > 0;) {
        const int g = pos >= 3 ? 3 : pos;
        pos -= g;
        k |= ((i >> pos) & ((1 << g) - 1)) << sh;
        sh += g;
    }
    return k;
}

OSZ8_HD void cmul(double &re, double &im, double wr, double wi) {
    const double a = re, b = im;
    re = a * wr - b * wi;
    im = a * wi + b * wr;
}

constexpr double kR2 = 0.70710678118654752440;   // sqrt(1/2)

// 8-point DFT in place, natural order in and out (register k = output k).
// INV: conjugated twiddles (unnormalised inverse).
template <bool INV>
OSZ8_HD void dft8(double *re, double *im) {
    // first split: a_j = x_j + x_{j+4}, b_j = (x_j - x_{j+4}) W8^j
    double ar[4], ai[4], br[4], bi[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ar[j] = re[j] + re[j + 4];
        ai[j] = im[j] + im[j + 4];
        br[j] = re[j] - re[j + 4];
        bi[j] = im[j] - im[j + 4];
    }
    // W8^1 = (1 - i)/sqrt2, W8^2 = -i, W8^3 = (-1 - i)/sqrt2  (conjugates for INV)
    {
        const double r1 = br[1], i1 = bi[1], r3 = br[3], i3 = bi[3], r2 = br[2], i2 = bi[2];
        if (!INV) {
            br[1] = (r1 + i1) * kR2;  bi[1] = (i1 - r1) * kR2;
            br[2] = i2;               bi[2] = -r2;
            br[3] = (i3 - r3) * kR2;  bi[3] = -(r3 + i3) * kR2;
        } else {
            br[1] = (r1 - i1) * kR2;  bi[1] = (i1 + r1) * kR2;
            br[2] = -i2;              bi[2] = r2;
            br[3] = -(r3 + i3) * kR2; bi[3] = (r3 - i3) * kR2;
        }
    }
    // 4-point DFTs: a -> X[0, 2, 4, 6], b -> X[1, 3, 5, 7]
#define OSZ8_DFT4(xr, xi, o0, o1, o2, o3)                                         \
    {                                                                             \
        const double c0r = xr[0] + xr[2], c0i = xi[0] + xi[2];                    \
        const double c1r = xr[1] + xr[3], c1i = xi[1] + xi[3];                    \
        const double d0r = xr[0] - xr[2], d0i = xi[0] - xi[2];                    \
        const double er = xr[1] - xr[3], ei = xi[1] - xi[3];                      \
        const double d1r = INV ? -ei : ei, d1i = INV ? er : -er; /* (x1 - x3) (-/+ i) */ \
        re[o0] = c0r + c1r; im[o0] = c0i + c1i;                                   \
        re[o2] = c0r - c1r; im[o2] = c0i - c1i;                                   \
        re[o1] = d0r + d1r; im[o1] = d0i + d1i;                                   \
        re[o3] = d0r - d1r; im[o3] = d0i - d1i;                                   \
    }
    OSZ8_DFT4(ar, ai, 0, 2, 4, 6)
    OSZ8_DFT4(br, bi, 1, 3, 5, 7)
#undef OSZ8_DFT4
}

// radix-2^Q butterflies over the low Q bits of the register index (Q = 1, 2),
// the upper register bits being independent instances: the last stage of sizes
// that are not a power of 8
template <int Q, bool INV>
OSZ8_HD void dft_low(double *re, double *im) {
    if (Q == 1) {
#pragma unroll
        for (int h = 0; h < 8; h += 2) {
            const double r0 = re[h], i0 = im[h], r1 = re[h + 1], i1 = im[h + 1];
            re[h] = r0 + r1; im[h] = i0 + i1;
            re[h + 1] = r0 - r1; im[h + 1] = i0 - i1;
        }
    } else {
#pragma unroll
        for (int h = 0; h < 8; h += 4) {
            const double c0r = re[h] + re[h + 2], c0i = im[h] + im[h + 2];
            const double c1r = re[h + 1] + re[h + 3], c1i = im[h + 1] + im[h + 3];
            const double d0r = re[h] - re[h + 2], d0i = im[h] - im[h + 2];
            const double er = re[h + 1] - re[h + 3], ei = im[h + 1] - im[h + 3];
            const double d1r = INV ? -ei : ei, d1i = INV ? er : -er;
            re[h] = c0r + c1r; im[h] = c0i + c1i;
            re[h + 2] = c0r - c1r; im[h + 2] = c0i - c1i;
            re[h + 1] = d0r + d1r; im[h + 1] = d0i + d1i;
            re[h + 3] = d0r - d1r; im[h + 3] = d0i - d1i;
        }
    }
}

// re/im[k] *= W^k (CONJ: conj(W)^k), k = 1 .. 7, from the base W = (wr, wi).
// The powers are formed as a chain W^(k+1) = W^k W, each used as soon as it
// exists: two complex values live instead of seven (the kernels built on this
// run at 128 registers per thread).
template <bool CONJ>
OSZ8_HD void twiddle8(double *re, double *im, double wr, double wi) {
    if (CONJ) wi = -wi;
    double pr = wr, pi = wi;
    cmul(re[1], im[1], pr, pi);
#pragma unroll
    for (int k = 2; k < 8; ++k) {
        cmul(pr, pi, wr, wi);
        cmul(re[k], im[k], pr, pi);
    }
}

// Per-thread twiddle bases, one per stage with P > 0: W_{2^(P+3)}^low
template <int N>
struct Plan {
    static constexpr int L = ilog2(N);
    static constexpr int NT = N / 8;
    static constexpr int Q = L % 3;                         // bits of the short last stage (0: none)
    static constexpr int NS = (L + 2) / 3;                  // stages
    // bit position of stage s (0-based from the top)
    static constexpr int pos(int s) { return (L - 3 * (s + 1)) > 0 ? (L - 3 * (s + 1)) : 0; }
};

template <int N>
struct Twid {
    double wr[Plan<N>::NS], wi[Plan<N>::NS];   // entry s unused when pos(s) == 0
};

template <int N>
OSZ8_HD void twid_load(int tid, const double *tab /* [1024][2] = W_8192^j */, Twid<N> &tw) {
    using PL = Plan<N>;
#pragma unroll
    for (int s = 0; s < PL::NS; ++s) {
        const int P = PL::pos(s);
        if (P > 0) {
            const int low = tid & ((1 << P) - 1);
            const int j = low << (ilog2(kTabN) - P - 3);    // low * 8192 / 2^(P+3)
            tw.wr[s] = tab[2 * j];
            tw.wi[s] = tab[2 * j + 1];
        } else {
            tw.wr[s] = 1.0;
            tw.wi[s] = 0.0;
        }
    }
}

// One forward stage.  S = stage number, (wr, wi) = this thread's twiddle base
// of the stage (Twid); the caller places the workgroup barrier between stages.
// In a loop over transforms the caller should pass an OPAQUE copy of the base
// (asm volatile("" : "+v"(wr), "+v"(wi))): otherwise the compiler hoists the
// power chain W^2 .. W^7 of every stage out of the loop -- 40 live doubles.  FIRST: input already in registers (x[(N/8) r + tid]).
// LAST: output stays in registers.
template <int N, int S>
OSZ8_HD void fwd_stage(int tid, double *re, double *im, double wr, double wi, C2 *lds) {
    using PL = Plan<N>;
    constexpr int P = PL::pos(S);
    constexpr bool FIRST = S == 0, LAST = S == PL::NS - 1;
    constexpr bool SHORT = LAST && PL::Q != 0;               // radix-2 / radix-4 tail
    if (!FIRST) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const C2 v = lds[swz(idx_of<P>(tid, r))];
            re[r] = v.re;
            im[r] = v.im;
        }
    }
    if (SHORT) {
        dft_low<PL::Q == 0 ? 1 : PL::Q, false>(re, im);
    } else {
        dft8<false>(re, im);
        if (P > 0) twiddle8<false>(re, im, wr, wi);
    }
    if (!LAST) {
#pragma unroll
        for (int r = 0; r < 8; ++r) lds[swz(idx_of<P>(tid, r))] = C2{re[r], im[r]};
    }
}

// One inverse stage (run S = NS-1 down to 0).  The stage that is LAST in the
// forward direction takes its input from registers, stage 0 leaves its output
// there: y[(N/8) r + tid], times N.
template <int N, int S>
OSZ8_HD void inv_stage(int tid, double *re, double *im, double wr, double wi, C2 *lds) {
    using PL = Plan<N>;
    constexpr int P = PL::pos(S);
    constexpr bool FIRST = S == 0, LAST = S == PL::NS - 1;
    constexpr bool SHORT = LAST && PL::Q != 0;
    if (!LAST) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const C2 v = lds[swz(idx_of<P>(tid, r))];
            re[r] = v.re;
            im[r] = v.im;
        }
    }
    if (SHORT) {
        dft_low<PL::Q == 0 ? 1 : PL::Q, true>(re, im);
    } else {
        if (P > 0) twiddle8<true>(re, im, wr, wi);
        dft8<true>(re, im);
    }
    if (!FIRST) {
#pragma unroll
        for (int r = 0; r < 8; ++r) lds[swz(idx_of<P>(tid, r))] = C2{re[r], im[r]};
    }
}

}  // namespace fft8
}  // namespace osz
