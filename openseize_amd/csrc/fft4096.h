// fft4096.h -- a 4096-point complex float64 FFT for one 256-thread workgroup,
// 16 points per thread in registers, three radix-16 passes with two LDS
// exchanges.  Used by K1 (overlap-add FIR: two real blocks ride one complex
// transform) and K5 (Welch/STFT: two real segments per transform).
//
// Index algebra (n = 256 n2 + 16 n1 + n0,  k = k0 + 16 k1 + 256 k2):
//   X[k] = sum_n0 W16^(n0 k2) W256^(n0 k1) W4096^(n0 k0)
//          sum_n1 W16^(n1 k1) W256^(n1 k0)  sum_n2 W16^(n2 k0) x[n]
//   pass 1  layout A: thread t = n0 + 16 n1, register j = n2 -> k0
//           twiddle T1[k0][t] = W4096^(t k0)          (= W256^(n1 k0) W4096^(n0 k0))
//   pass 2  layout B: thread t = k0 + 16 n0, register j = n1 -> k1
//           twiddle T2[n0][k1] = W256^(n0 k1)
//   pass 3  layout C: thread t = k0 + 16 k1, register j = n0 -> k2
//   so input x[256 j + t] and output X[256 j + t] are both read/written with
//   consecutive lanes on consecutive elements (coalesced), no bit reversal.
// The inverse runs the same passes backwards with conjugated twiddles
// (1/4096 is folded into the caller's spectrum).
//
// LDS: one array of 4096 interleaved complex slots (64 KB), the "cube" below:
// every pass reads its 16 slots, transforms and writes the SAME slots back,
// so an exchange costs one barrier, and re/im travel as 16-byte accesses.
//
// The phase functions are __host__ __device__ so tests/host/fft_host_check.cpp
// can replay the 256 threads on the CPU and pin the index math without a GPU.
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define OSZ_HD __host__ __device__ __forceinline__
#else
#define OSZ_HD inline
#endif

namespace osz {
namespace fft {

constexpr int N = 4096;
constexpr int NT = 256;          // threads per transform

// after fwd16 register r holds logical index dr(r); inv16 expects the same
OSZ_HD constexpr int dr(int r) { return (r >> 2) | ((r & 3) << 2); }

constexpr double C1 = 0.92387953251128673848;  // cos(pi/8)
constexpr double S1_ = 0.38268343236508978178; // sin(pi/8)
constexpr double R2 = 0.70710678118654752440;  // sqrt(1/2)

// multiply (re, im) by W16^E (forward) or its conjugate (INV)
template <int E, bool INV>
OSZ_HD void tw16(double &re, double &im) {
    constexpr int e = E % 16;
    if constexpr (e == 0) {
        return;
    } else if constexpr (e == 4) {  // -i (fwd) / +i (inv)
        const double r = re;
        if constexpr (!INV) { re = im; im = -r; } else { re = -im; im = r; }
    } else if constexpr (e == 2) {  // (1 - i)/sqrt2  /  (1 + i)/sqrt2
        const double r = re, i = im;
        if constexpr (!INV) { re = (r + i) * R2; im = (i - r) * R2; }
        else { re = (r - i) * R2; im = (i + r) * R2; }
    } else if constexpr (e == 6) {  // (-1 - i)/sqrt2  /  (-1 + i)/sqrt2
        const double r = re, i = im;
        if constexpr (!INV) { re = (i - r) * R2; im = -(r + i) * R2; }
        else { re = -(r + i) * R2; im = (r - i) * R2; }
    } else {
        // generic: W16^e = (c, -s) forward
        constexpr double c = (e == 1) ? C1 : (e == 3) ? S1_ : (e == 9) ? -C1 : 0.0;
        constexpr double s = (e == 1) ? S1_ : (e == 3) ? C1 : (e == 9) ? -S1_ : 0.0;
        static_assert(e == 1 || e == 3 || e == 9, "unexpected radix-16 twiddle");
        const double r = re, i = im;
        if constexpr (!INV) { re = r * c + i * s; im = i * c - r * s; }
        else { re = r * c - i * s; im = i * c + r * s; }
    }
}

template <bool INV>
OSZ_HD void dft4(double &r0, double &i0, double &r1, double &i1, double &r2, double &i2,
                 double &r3, double &i3) {
    const double t0r = r0 + r2, t0i = i0 + i2;
    const double t1r = r0 - r2, t1i = i0 - i2;
    const double t2r = r1 + r3, t2i = i1 + i3;
    const double t3r = r1 - r3, t3i = i1 - i3;
    r0 = t0r + t2r; i0 = t0i + t2i;
    r2 = t0r - t2r; i2 = t0i - t2i;
    if constexpr (!INV) {  // X1 = t1 - i t3, X3 = t1 + i t3
        r1 = t1r + t3i; i1 = t1i - t3r;
        r3 = t1r - t3i; i3 = t1i + t3r;
    } else {
        r1 = t1r - t3i; i1 = t1i + t3r;
        r3 = t1r + t3i; i3 = t1i - t3r;
    }
}

// Four of the nine inner twiddles of a radix-16 butterfly are W16^2 / W16^6 = (+-1 +- i) / sqrt 2: a
// sum or difference of the two parts, then a scale by sqrt(1/2).  The scale rides the FMAs of the
// radix-4 butterfly that follows instead of costing two multiplications per twiddle (round 5: 8
// instructions less per 16-point transform, 160 -> 152).  tw16u leaves the value UNSCALED:
template <int E, bool INV>
OSZ_HD void tw16u(double &re, double &im) {
    static_assert(E == 2 || E == 6, "the sqrt(1/2) twiddles");
    const double r = re, i = im;
    if constexpr (E == 2) {
        if constexpr (!INV) { re = r + i; im = i - r; } else { re = r - i; im = i + r; }
    } else {
        if constexpr (!INV) { re = i - r; im = -r - i; } else { re = -r - i; im = r - i; }
    }
}
// ... the butterfly whose THIRD input (r2, i2) is such an unscaled value:
template <bool INV>
OSZ_HD void dft4_c(double &r0, double &i0, double &r1, double &i1, double &r2, double &i2,
                   double &r3, double &i3) {
    const double t0r = __builtin_fma(R2, r2, r0), t0i = __builtin_fma(R2, i2, i0);
    const double t1r = __builtin_fma(-R2, r2, r0), t1i = __builtin_fma(-R2, i2, i0);
    const double t2r = r1 + r3, t2i = i1 + i3;
    const double t3r = r1 - r3, t3i = i1 - i3;
    r0 = t0r + t2r; i0 = t0i + t2i;
    r2 = t0r - t2r; i2 = t0i - t2i;
    if constexpr (!INV) {
        r1 = t1r + t3i; i1 = t1i - t3r;
        r3 = t1r - t3i; i3 = t1i + t3r;
    } else {
        r1 = t1r - t3i; i1 = t1i + t3r;
        r3 = t1r + t3i; i3 = t1i - t3r;
    }
}
// ... and the one whose SECOND and FOURTH inputs are:
template <bool INV>
OSZ_HD void dft4_bd(double &r0, double &i0, double &r1, double &i1, double &r2, double &i2,
                    double &r3, double &i3) {
    const double t0r = r0 + r2, t0i = i0 + i2;
    const double t1r = r0 - r2, t1i = i0 - i2;
    const double t2r = r1 + r3, t2i = i1 + i3;      // (unscaled)
    const double t3r = r1 - r3, t3i = i1 - i3;
    r0 = __builtin_fma(R2, t2r, t0r); i0 = __builtin_fma(R2, t2i, t0i);
    r2 = __builtin_fma(-R2, t2r, t0r); i2 = __builtin_fma(-R2, t2i, t0i);
    if constexpr (!INV) {
        r1 = __builtin_fma(R2, t3i, t1r); i1 = __builtin_fma(-R2, t3r, t1i);
        r3 = __builtin_fma(-R2, t3i, t1r); i3 = __builtin_fma(R2, t3r, t1i);
    } else {
        r1 = __builtin_fma(-R2, t3i, t1r); i1 = __builtin_fma(R2, t3r, t1i);
        r3 = __builtin_fma(R2, t3i, t1r); i3 = __builtin_fma(-R2, t3r, t1i);
    }
}

// Forward 16-point DFT in place: input logical n at register n, output
// logical k at register dr(k).
OSZ_HD void fwd16(double *re, double *im) {
#define OSZ_D4(a, b, c, d) dft4<false>(re[a], im[a], re[b], im[b], re[c], im[c], re[d], im[d])
#define OSZ_D4C(a, b, c, d) dft4_c<false>(re[a], im[a], re[b], im[b], re[c], im[c], re[d], im[d])
#define OSZ_D4BD(a, b, c, d) dft4_bd<false>(re[a], im[a], re[b], im[b], re[c], im[c], re[d], im[d])
    // stage 1: over a' (n = 4a' + b): registers {b, b+4, b+8, b+12}, output c at b + 4c
    OSZ_D4(0, 4, 8, 12); OSZ_D4(1, 5, 9, 13); OSZ_D4(2, 6, 10, 14); OSZ_D4(3, 7, 11, 15);
    // twiddle W16^(b c) on register b + 4c (registers 6, 9, 11, 14: unscaled, see tw16u)
    tw16<1, false>(re[5], im[5]);   tw16u<2, false>(re[9], im[9]);  tw16<3, false>(re[13], im[13]);
    tw16u<2, false>(re[6], im[6]);  tw16<4, false>(re[10], im[10]); tw16u<6, false>(re[14], im[14]);
    tw16<3, false>(re[7], im[7]);   tw16u<6, false>(re[11], im[11]); tw16<9, false>(re[15], im[15]);
    // stage 2: over b for fixed c: registers {4c..4c+3}, output d at 4c + d (k = c + 4d)
    OSZ_D4(0, 1, 2, 3); OSZ_D4C(4, 5, 6, 7); OSZ_D4BD(8, 9, 10, 11); OSZ_D4C(12, 13, 14, 15);
#undef OSZ_D4
#undef OSZ_D4C
#undef OSZ_D4BD
}

// Inverse (unnormalised) 16-point DFT in place: input logical k at register
// dr(k), output logical n at register n.
OSZ_HD void inv16(double *re, double *im) {
#define OSZ_D4(a, b, c, d) dft4<true>(re[a], im[a], re[b], im[b], re[c], im[c], re[d], im[d])
#define OSZ_D4C(a, b, c, d) dft4_c<true>(re[a], im[a], re[b], im[b], re[c], im[c], re[d], im[d])
#define OSZ_D4BD(a, b, c, d) dft4_bd<true>(re[a], im[a], re[b], im[b], re[c], im[c], re[d], im[d])
    OSZ_D4(0, 1, 2, 3); OSZ_D4(4, 5, 6, 7); OSZ_D4(8, 9, 10, 11); OSZ_D4(12, 13, 14, 15);
    tw16<1, true>(re[5], im[5]);   tw16u<2, true>(re[9], im[9]);   tw16<3, true>(re[13], im[13]);
    tw16u<2, true>(re[6], im[6]);  tw16<4, true>(re[10], im[10]);  tw16u<6, true>(re[14], im[14]);
    tw16<3, true>(re[7], im[7]);   tw16u<6, true>(re[11], im[11]); tw16<9, true>(re[15], im[15]);
    OSZ_D4(0, 4, 8, 12); OSZ_D4C(1, 5, 9, 13); OSZ_D4BD(2, 6, 10, 14); OSZ_D4C(3, 7, 11, 15);
#undef OSZ_D4
#undef OSZ_D4C
#undef OSZ_D4BD
}

struct Tables {
    const double *t1;  // [16][256][2]  W4096^(t k0)  (re, im)
    const double *t2;  // [16][16][2]   W256^(n0 k1)
    const double *t0 = nullptr;  // [256][2]  W16384^t (the negacyclic transform's per-thread twist, nega below)
};

// ---- cube layout: interleaved complex, in-place exchanges ------------------
// One array of 4096 complex slots (64 KB, no padding) viewed as a 16x16x16 cube
//   slot(k0, m, l) = 256 k0 + 16 m + (l ^ k0)
// and three ownership views, one per pass:
//   view A  thread (l = n0, m = n1) owns the 16 slots k0 = 0..15   (pass 1)
//   view B  thread (k0, l = n0)     owns the 16 slots m = n1 | k1  (pass 2)
//   view C  thread (k0, m = k1)     owns the 16 slots l = n0 | k2  (pass 3)
// Every pass reads its 16 slots, transforms, and writes the SAME slots back,
// so the only hazards are between passes: one barrier per exchange (four per
// forward + inverse) instead of a load / barrier / store sandwich, and the
// slots a thread reads last (view A, inverse pass 1) are the ones it writes
// first for the next transform -- no barrier between transforms at all.
// re and im travel together as 16-byte accesses (ds_read_b128 / ds_write_b128:
// full LDS rate already at one or two waves per SIMD, MI355X_MICROARCH.md LDS
// table).  The XOR with k0 makes all three views conflict free for the b128
// lane groups (read: 16 lanes x 16 B over 64 banks, write: 8 lanes x 16 B
// over 32 banks): view A lanes walk l (16 distinct columns), views B and C
// lanes walk k0 (column (l ^ k0), again 16 distinct); checked by brute force
// over the guide's lane groups in tests/test_fft_host.py.
//
// Twiddles stay off the vector-memory path: a wave's global loads return in
// order, so a twiddle load (an L2 hit) issued after the sample loads of an HBM
// stream inherits their latency.  Each thread keeps the powers {1, 2, 4, 8} of
// its two twiddle bases resident (W4096^t for pass 1, W256^n0 for pass 2:
// 32 registers, exact table values) and applies W^k, k = b + 4a, as the
// product W^b (W^4)^a with W^3 = W W^2 and W^12 = W^4 W^8 formed on the spot:
// 26 complex multiplies per pass, as many as expanding all 15 powers would
// take, but only 6 twiddles live instead of 15.  After fwd16 (and before
// inv16) register r holds k = dr(r) = (r >> 2) + 4 (r & 3): b = r >> 2,
// a = r & 3.
namespace cube {

struct alignas(16) C2 {
    double re, im;
};

constexpr int SLOTS = 4096;

OSZ_HD int slot_a(int t, int k0) { return 256 * k0 + (t ^ k0); }            // t = 16 m + l
OSZ_HD int base_b(int t) { return 256 * (t & 15) + ((t >> 4) ^ (t & 15)); } // + 16 m
OSZ_HD int slot_c(int t, int l) { return 256 * (t & 15) + (t & ~15) + (l ^ (t & 15)); }

OSZ_HD void cmul(double &re, double &im, double wr, double wi) {
    const double a = re, b = im;
    re = a * wr - b * wi;
    im = a * wi + b * wr;
}

// (a + ib)(wr - i wi): conjugate multiply without negating the twiddle first
OSZ_HD void cmulc(double &re, double &im, double wr, double wi) {
    const double a = re, b = im;
    re = a * wr + b * wi;
    im = b * wr - a * wi;
}

struct TwPow {
    double r[4], i[4];   // base^{1,2,4,8}
};

OSZ_HD void tw_load(int t, const Tables &tb, TwPow &w1, TwPow &w2) {
    const int n0 = t >> 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = 1 << q;
        w1.r[q] = tb.t1[(k * 256 + t) * 2];
        w1.i[q] = tb.t1[(k * 256 + t) * 2 + 1];
        w2.r[q] = tb.t2[(n0 * 16 + k) * 2];
        w2.i[q] = tb.t2[(n0 * 16 + k) * 2 + 1];
    }
}

template <bool CONJ>
OSZ_HD void tw_mul(double *re, double *im, const TwPow &w) {
    double br[4], bi[4], ar[4], ai[4];   // W^b, (W^4)^a
    br[1] = w.r[0]; bi[1] = w.i[0];
    br[2] = w.r[1]; bi[2] = w.i[1];
    ar[1] = w.r[2]; ai[1] = w.i[2];
    ar[2] = w.r[3]; ai[2] = w.i[3];
    br[3] = br[1] * br[2] - bi[1] * bi[2]; bi[3] = br[1] * bi[2] + bi[1] * br[2];
    ar[3] = ar[1] * ar[2] - ai[1] * ai[2]; ai[3] = ar[1] * ai[2] + ai[1] * ar[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int b = r >> 2, a = r & 3;
        if (CONJ) {
            if (b != 0) cmulc(re[r], im[r], br[b], bi[b]);
            if (a != 0) cmulc(re[r], im[r], ar[a], ai[a]);
        } else {
            if (b != 0) cmul(re[r], im[r], br[b], bi[b]);
            if (a != 0) cmul(re[r], im[r], ar[a], ai[a]);
        }
    }
}

// F1: registers hold x[256 j + t] at register j.  Pass 1, twiddle W4096^(t k0),
// store view A.
OSZ_HD void f1(int t, double *re, double *im, const TwPow &w1, C2 *L) {
    fwd16(re, im);
    tw_mul<false>(re, im, w1);
#pragma unroll
    for (int r = 0; r < 16; ++r) L[slot_a(t, dr(r))] = C2{re[r], im[r]};
}

// F2: view B in place: load n1, pass 2, twiddle W256^(n0 k1), store k1.
OSZ_HD void f2(int t, double *re, double *im, const TwPow &w2, C2 *L) {
    const int base = base_b(t);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const C2 v = L[base + 16 * j];
        re[j] = v.re;
        im[j] = v.im;
    }
    fwd16(re, im);
    tw_mul<false>(re, im, w2);
#pragma unroll
    for (int r = 0; r < 16; ++r) L[base + 16 * dr(r)] = C2{re[r], im[r]};
}

// F3: view C: load n0, pass 3.  Afterwards register r holds X[t + 256 dr(r)].
OSZ_HD void f3(int t, double *re, double *im, const C2 *L) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const C2 v = L[slot_c(t, j)];
        re[j] = v.re;
        im[j] = v.im;
    }
    fwd16(re, im);
}

// I3: registers hold Y[t + 256 dr(r)] at register r.  Inverse pass 3, store
// view C (the slots F3 read).
OSZ_HD void i3(int t, double *re, double *im, C2 *L) {
    inv16(re, im);
#pragma unroll
    for (int j = 0; j < 16; ++j) L[slot_c(t, j)] = C2{re[j], im[j]};
}

// I2: view B in place: load k1 (at register dr(k1)), conj twiddle, inverse
// pass 2, store n1.
OSZ_HD void i2(int t, double *re, double *im, const TwPow &w2, C2 *L) {
    const int base = base_b(t);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const C2 v = L[base + 16 * dr(r)];
        re[r] = v.re;
        im[r] = v.im;
    }
    tw_mul<true>(re, im, w2);
    inv16(re, im);
#pragma unroll
    for (int j = 0; j < 16; ++j) L[base + 16 * j] = C2{re[j], im[j]};
}

// I1: view A: load k0 (at register dr(k0)), conj twiddle, inverse pass 1.
// Afterwards register j holds y[256 j + t] (times 4096).
OSZ_HD void i1(int t, double *re, double *im, const TwPow &w1, const C2 *L) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const C2 v = L[slot_a(t, dr(r))];
        re[r] = v.re;
        im[r] = v.im;
    }
    tw_mul<true>(re, im, w1);
    inv16(re, im);
}

}  // namespace cube

// ---- cube2: the same cube with the SECOND exchange inside a 16-lane row --------------
// In `cube` a thread changes both of its coordinates at each exchange, so both exchanges
// cross waves and cost a workgroup barrier.  Here the thread index keeps one coordinate
// per exchange:
//   pass 1  t = n0 + 16 n1   registers n2 -> k0      (as in cube: coalesced loads)
//   pass 2  t = n0 + 16 k0   registers n1 -> k1      exchange 1: register <-> HIGH half of t
//   pass 3  t = k1 + 16 k0   registers n0 -> k2      exchange 2: register <-> LOW half of t
// Exchange 2 moves data among the 16 lanes that share t >> 4: one DPP row of one wave.  A wave
// executes in lockstep, so its LDS writes are done for all its lanes once it has waited for
// them: no workgroup barrier, and the four waves of a workgroup stop marching in step.  The
// inverse mirrors it (exchange 2 first).  Two barriers per forward + inverse instead of four.
// After pass 3 register r of thread t holds bin  k = (t >> 4) + 16 (t & 15) + 256 dr(r):
// a caller that multiplies by a spectrum stores it in that order ([r][t], see bin()).
//   slot(k0, m, l) = 256 k0 + 16 m + (l ^ m)
// with views A (l = n0, m = n1 | k0 = 0..15), B (k0, l = n0 | m = n1, k1), C (k0, m = k1 | l = n0,
// k2): lanes walk l in A and B (16 consecutive slots), m in C (l ^ m distinct): conflict free for
// the b128 lane groups (tests/host/fft_host_check.cpp).
// (diagnostic builds of benchmarks/zpn_variant.hip only: OSZ_ABL_NOLDS drops the cube's loads and
// stores -- what the LDS exchanges cost a kernel, its results then meaningless)
#ifdef OSZ_ABL_NOLDS
#define OSZ_LDS_ST(slot_, r_, i_) do { } while (0)
#define OSZ_LDS_LD(slot_, r_, i_) do { } while (0)
#else
#define OSZ_LDS_ST(slot_, r_, i_) slot_ = C2{r_, i_}
#define OSZ_LDS_LD(slot_, r_, i_) do { const C2 v_ = slot_; r_ = v_.re; i_ = v_.im; } while (0)
#endif

namespace cube2 {

using cube::C2;
using cube::TwPow;
using cube::tw_mul;
constexpr int SLOTS = 4096;

OSZ_HD int slot(int k0, int m, int l) { return 256 * k0 + 16 * m + (l ^ m); }
OSZ_HD int slot_a(int t, int k0) { return slot(k0, t >> 4, t & 15); }          // t = n0 + 16 n1
OSZ_HD int slot_b(int t, int m) { return slot(t >> 4, m, t & 15); }            // t = n0 + 16 k0
OSZ_HD int slot_c(int t, int l) { return slot(t >> 4, t & 15, l); }            // t = k1 + 16 k0
// bin held by register r of thread t after f3 (and expected there by i3)
OSZ_HD int bin(int t, int r) { return (t >> 4) + 16 * (t & 15) + 256 * dr(r); }

OSZ_HD void tw_load(int t, const Tables &tb, TwPow &w1, TwPow &w2) {
    const int n0 = t & 15;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = 1 << q;
        w1.r[q] = tb.t1[(k * 256 + t) * 2];
        w1.i[q] = tb.t1[(k * 256 + t) * 2 + 1];
        w2.r[q] = tb.t2[(n0 * 16 + k) * 2];
        w2.i[q] = tb.t2[(n0 * 16 + k) * 2 + 1];
    }
}

OSZ_HD void f1(int t, double *re, double *im, const TwPow &w1, C2 *L) {
    fwd16(re, im);
    tw_mul<false>(re, im, w1);
#pragma unroll
    for (int r = 0; r < 16; ++r) OSZ_LDS_ST(L[slot_a(t, dr(r))], re[r], im[r]);
}

OSZ_HD void f2(int t, double *re, double *im, const TwPow &w2, C2 *L) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        OSZ_LDS_LD(L[slot_b(t, j)], re[j], im[j]);
    }
    fwd16(re, im);
    tw_mul<false>(re, im, w2);
#pragma unroll
    for (int r = 0; r < 16; ++r) OSZ_LDS_ST(L[slot_b(t, dr(r))], re[r], im[r]);
}

OSZ_HD void f3(int t, double *re, double *im, const C2 *L) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        OSZ_LDS_LD(L[slot_c(t, j)], re[j], im[j]);
    }
    fwd16(re, im);
}

OSZ_HD void i3(int t, double *re, double *im, C2 *L) {
    inv16(re, im);
#pragma unroll
    for (int j = 0; j < 16; ++j) OSZ_LDS_ST(L[slot_c(t, j)], re[j], im[j]);
}

OSZ_HD void i2(int t, double *re, double *im, const TwPow &w2, C2 *L) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        OSZ_LDS_LD(L[slot_b(t, dr(r))], re[r], im[r]);
    }
    tw_mul<true>(re, im, w2);
    inv16(re, im);
#pragma unroll
    for (int j = 0; j < 16; ++j) OSZ_LDS_ST(L[slot_b(t, j)], re[j], im[j]);
}

OSZ_HD void i1(int t, double *re, double *im, const TwPow &w1, const C2 *L) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        OSZ_LDS_LD(L[slot_a(t, dr(r))], re[r], im[r]);
    }
    tw_mul<true>(re, im, w1);
    inv16(re, im);
}

}  // namespace cube2


// ---- nega: ONE real block of 8192 samples on the 4096-point transform -------------------
// The pair trick (two real blocks as the real and imaginary part of one complex transform)
// pays the filter's tail and a guard row once per 4096-sample window.  Here the window is 8192
// samples of ONE real block and the transform is the DFT at the ODD frequencies
//   w_k = 2 pi (k + 1/2) / 8192,  k = 0 .. 8191,
// whose even-numbered half X_{2j} (the odd-numbered half is its conjugate mirror, x being real)
// is a 4096-point complex transform of
//   z[n] = (x[n] - i x[n + 4096]) e^{-i pi n / 8192},   n = 0 .. 4095:   X_{2j} = FFT4096(z)[j].
// No untangling pass: the filter multiplies bin j by its response at 2 pi (j + 1/4) / 4096, and
//   y[n] - i y[n + 4096] = IFFT4096(Y)[n] e^{+i pi n / 8192}.
// Multiplication at the odd frequencies is the NEGACYCLIC convolution: what leaves the window on
// one side comes back on the other with its sign changed -- as good as the cyclic one for
// overlap-add, where nothing but a cascade's ringing (fitted and removed, chain_zp.hip) wraps.
// Registers: re[j] = x[256 j + t], im[j] = x[4096 + 256 j + t] (rows 0-15 and 16-31 of the window),
// the same on the way out.  The twist e^{-i pi n / 8192}, n = 256 j + t, splits into a register
// part e^{-i pi j / 32} (constants, applied with the packing / unpacking) and a thread part
// beta = e^{-i pi t / 8192} = W16384^t that rides the pass-1 twiddle: W4096^(t k0) beta =
// beta^(4 k0 + 1).
namespace nega {

using cube::C2;
using cube::cmul;
using cube::cmulc;

constexpr double kCos[16] = {1.0,
                             0.995184726672196886244837,
                             0.9807852804032304491261822,
                             0.9569403357322088649357979,
                             0.9238795325112867561281832,
                             0.8819212643483550297127569,
                             0.8314696123025452370787884,
                             0.7730104533627369608109066,
                             0.7071067811865475244008444,
                             0.6343932841636454982151716,
                             0.5555702330196022247428308,
                             0.4713967368259976485563876,
                             0.38268343236508977172846,
                             0.2902846772544623676361924,
                             0.1950903220161282678482849,
                             0.09801714032956060199419556};
constexpr double kSin[16] = {0.0,
                             0.09801714032956060199419556,
                             0.1950903220161282678482849,
                             0.2902846772544623676361924,
                             0.38268343236508977172846,
                             0.4713967368259976485563876,
                             0.5555702330196022247428308,
                             0.6343932841636454982151716,
                             0.7071067811865475244008444,
                             0.7730104533627369608109066,
                             0.8314696123025452370787884,
                             0.8819212643483550297127569,
                             0.9238795325112867561281832,
                             0.9569403357322088649357979,
                             0.9807852804032304491261822,
                             0.995184726672196886244837};

struct TwPowN {
    double r[5], i[5];   // beta, W, W^2, W^4, W^8 with W = beta^4 = W4096^t
};

OSZ_HD void tw_load(int t, const Tables &tb, TwPowN &w1, cube::TwPow &w2) {
    const int n0 = t & 15;
    w1.r[0] = tb.t0[2 * t];
    w1.i[0] = tb.t0[2 * t + 1];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = 1 << q;
        w1.r[q + 1] = tb.t1[(k * 256 + t) * 2];
        w1.i[q + 1] = tb.t1[(k * 256 + t) * 2 + 1];
        w2.r[q] = tb.t2[(n0 * 16 + k) * 2];
        w2.i[q] = tb.t2[(n0 * 16 + k) * 2 + 1];
    }
}

// register r (bin digit k0 = dr(r) = b + 4 a, b = r >> 2, a = r & 3) times beta^(4 k0 + 1) =
// (beta W^b) (W^4)^a, or its conjugate
template <bool CONJ>
OSZ_HD void tw_mul(double *re, double *im, const TwPowN &w) {
    double br[4], bi[4], ar[4], ai[4];
    br[0] = w.r[0]; bi[0] = w.i[0];
    br[1] = w.r[0] * w.r[1] - w.i[0] * w.i[1]; bi[1] = w.r[0] * w.i[1] + w.i[0] * w.r[1];   // beta W
    br[2] = w.r[0] * w.r[2] - w.i[0] * w.i[2]; bi[2] = w.r[0] * w.i[2] + w.i[0] * w.r[2];   // beta W^2
    br[3] = br[1] * w.r[2] - bi[1] * w.i[2];   bi[3] = br[1] * w.i[2] + bi[1] * w.r[2];     // beta W^3
    ar[1] = w.r[3]; ai[1] = w.i[3];
    ar[2] = w.r[4]; ai[2] = w.i[4];
    ar[3] = ar[1] * ar[2] - ai[1] * ai[2]; ai[3] = ar[1] * ai[2] + ai[1] * ar[2];           // W^12
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int b = r >> 2, a = r & 3;
        if (CONJ) {
            cmulc(re[r], im[r], br[b], bi[b]);
            if (a != 0) cmulc(re[r], im[r], ar[a], ai[a]);
        } else {
            cmul(re[r], im[r], br[b], bi[b]);
            if (a != 0) cmul(re[r], im[r], ar[a], ai[a]);
        }
    }
}

// Pack: lo[j] = x[256 j + t], hi[j] = x[4096 + 256 j + t] (in re / im) -> z[256 j + t] without its
// thread factor.  NHI: rows of the upper half that hold samples (the rest of hi is zero by
// construction -- the block is shorter than the window).
template <int NHI>
OSZ_HD void pack(double *re, double *im) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double a = re[j], b = j < NHI ? im[j] : 0.0;
        if (j == 0) {
            im[j] = -b;
        } else if (j < NHI) {
            re[j] = a * kCos[j] - b * kSin[j];
            im[j] = -(a * kSin[j]) - b * kCos[j];
        } else {
            re[j] = a * kCos[j];
            im[j] = -(a * kSin[j]);
        }
    }
}

// Unpack: v = w e^{+i pi j / 32}; y[256 j + t] = Re v (re), y[4096 + 256 j + t] = -Im v (im)
OSZ_HD void unpack(double *re, double *im) {
#pragma unroll
    for (int j = 1; j < 16; ++j) {
        const double a = re[j], b = im[j];
        re[j] = a * kCos[j] - b * kSin[j];
        im[j] = -(a * kSin[j]) - b * kCos[j];
    }
    im[0] = -im[0];
}

template <int NHI>
OSZ_HD void f1(int t, double *re, double *im, const TwPowN &w1, C2 *L) {
    pack<NHI>(re, im);
    fwd16(re, im);
    tw_mul<false>(re, im, w1);
#pragma unroll
    for (int r = 0; r < 16; ++r) OSZ_LDS_ST(L[cube2::slot_a(t, dr(r))], re[r], im[r]);
}

// (in two halves: once the loads have returned the thread's slots of the cube are free -- the
// kernels request the next block's samples into them before the arithmetic below)
OSZ_HD void i1_load(int t, double *re, double *im, const C2 *L) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        OSZ_LDS_LD(L[cube2::slot_a(t, dr(r))], re[r], im[r]);
    }
}

OSZ_HD void i1_finish(double *re, double *im, const TwPowN &w1) {
    tw_mul<true>(re, im, w1);
    inv16(re, im);
    unpack(re, im);
}

OSZ_HD void i1(int t, double *re, double *im, const TwPowN &w1, const C2 *L) {
    i1_load(t, re, im, L);
    i1_finish(re, im, w1);
}

}  // namespace nega

}  // namespace fft
}  // namespace osz
