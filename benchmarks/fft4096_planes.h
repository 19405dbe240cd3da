// fft4096_planes.h -- the first LDS layout of the 4096-point transform
// (separate re/im planes of doubles, load / barrier / store exchanges), kept
// only for benchmarks/fir_ablate.hip, the ablation study that led to the cube
// layout now in openseize_amd/csrc/fft4096.h.  Not part of the library.
//
// Exchange 1 is a 16x16 transpose between lanes (n0 fast) and lanes (k0 fast):
//   slot1 = k0*272 + n1*16 + ((n0 + k0) & 15)
// exchange 2 is lane-contiguous on both sides:
//   slot2 = n0*272 + k1*16 + k0
#pragma once
#include "../openseize_amd/csrc/fft4096.h"

namespace osz {
namespace fft {

constexpr int S1 = 272;          // exchange-1 row stride (doubles); columns rotated by k0
constexpr int S2 = 272;          // exchange-2 row stride (doubles), = 16 mod 32
constexpr int PLANE = 16 * S2;   // doubles per plane (>= 16*S1)

// Pass-1 twiddles W4096^(t k0), k0 = 1..15, for this thread: only the four
// power-of-two ones are loaded (64 B instead of 240 B of L2 traffic per
// thread and pass); the others are products of at most three of them.
OSZ_HD void t1_powers(int t, const Tables &tb, double *wr, double *wi) {
#define OSZ_LD(K) wr[K] = tb.t1[((K) * 256 + t) * 2]; wi[K] = tb.t1[((K) * 256 + t) * 2 + 1];
#define OSZ_MUL(C, A, B) wr[C] = wr[A] * wr[B] - wi[A] * wi[B]; wi[C] = wr[A] * wi[B] + wi[A] * wr[B];
    OSZ_LD(1) OSZ_LD(2) OSZ_LD(4) OSZ_LD(8)
    OSZ_MUL(3, 1, 2) OSZ_MUL(5, 1, 4) OSZ_MUL(6, 2, 4) OSZ_MUL(7, 3, 4)
    OSZ_MUL(9, 1, 8) OSZ_MUL(10, 2, 8) OSZ_MUL(11, 3, 8) OSZ_MUL(12, 4, 8)
    OSZ_MUL(13, 5, 8) OSZ_MUL(14, 6, 8) OSZ_MUL(15, 7, 8)
#undef OSZ_LD
#undef OSZ_MUL
}

// ---- forward phases ----------------------------------------------------
// F1: registers hold x[256 j + t] at register j (layout A).  Pass 1, twiddle,
// store to exchange 1.
template <bool POW = true>
OSZ_HD void f1(int t, double *re, double *im, const Tables &tb, double *pr, double *pi) {
    double twr[16], twi[16];
    if constexpr (POW) {
        t1_powers(t, tb, twr, twi);
    } else {
#pragma unroll
        for (int k = 1; k < 16; ++k) {
            twr[k] = tb.t1[(k * 256 + t) * 2];
            twi[k] = tb.t1[(k * 256 + t) * 2 + 1];
        }
    }
    fwd16(re, im);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k0 = dr(r);
        if (k0 != 0) {
            const double wr = twr[k0], wi = twi[k0];
            const double a = re[r], b = im[r];
            re[r] = a * wr - b * wi;
            im[r] = a * wi + b * wr;
        }
        const int slot = k0 * S1 + (t & ~15) + ((t + k0) & 15);
        pr[slot] = re[r];
        pi[slot] = im[r];
    }
}

// F2: load layout B (register j = n1), pass 2, twiddle, (caller barriers), store exchange 2.
OSZ_HD void f2_load(int t, double *re, double *im, const double *pr, const double *pi) {
    const int k0 = t & 15, n0 = t >> 4;
    const int base = k0 * S1 + ((n0 + k0) & 15);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        re[j] = pr[base + j * 16];
        im[j] = pi[base + j * 16];
    }
}

OSZ_HD void f2_compute(int t, double *re, double *im, const Tables &tb) {
    const int n0 = t >> 4;
    fwd16(re, im);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k1 = dr(r);
        if (k1 != 0) {
            const double wr = tb.t2[(n0 * 16 + k1) * 2], wi = tb.t2[(n0 * 16 + k1) * 2 + 1];
            const double a = re[r], b = im[r];
            re[r] = a * wr - b * wi;
            im[r] = a * wi + b * wr;
        }
    }
}

OSZ_HD void f2_store(int t, const double *re, const double *im, double *pr, double *pi) {
    const int k0 = t & 15, n0 = t >> 4;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k1 = dr(r);
        pr[n0 * S2 + k1 * 16 + k0] = re[r];
        pi[n0 * S2 + k1 * 16 + k0] = im[r];
    }
}

// F3: load layout C (register j = n0), pass 3.  Afterwards register r holds
// X[t + 256 dr(r)].
OSZ_HD void f3(int t, double *re, double *im, const double *pr, const double *pi) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        re[j] = pr[j * S2 + t];
        im[j] = pi[j * S2 + t];
    }
    fwd16(re, im);
}

// ---- inverse phases ----------------------------------------------------
// I3: registers hold Y[t + 256 dr(r)] at register r.  Inverse pass 3, store.
OSZ_HD void i3(int t, double *re, double *im, double *pr, double *pi) {
    inv16(re, im);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        pr[j * S2 + t] = re[j];
        pi[j * S2 + t] = im[j];
    }
}

// I2: load layout B with logical k1 at register dr(k1), conj twiddle, inverse pass 2.
OSZ_HD void i2_load(int t, double *re, double *im, const Tables &tb, const double *pr,
                    const double *pi) {
    const int k0 = t & 15, n0 = t >> 4;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k1 = dr(r);
        double a = pr[n0 * S2 + k1 * 16 + k0], b = pi[n0 * S2 + k1 * 16 + k0];
        if (k1 != 0) {
            const double wr = tb.t2[(n0 * 16 + k1) * 2], wi = -tb.t2[(n0 * 16 + k1) * 2 + 1];
            const double a2 = a * wr - b * wi;
            b = a * wi + b * wr;
            a = a2;
        }
        re[r] = a;
        im[r] = b;
    }
}

OSZ_HD void i2_compute_store(int t, double *re, double *im, double *pr, double *pi) {
    inv16(re, im);
    const int k0 = t & 15, n0 = t >> 4;
    const int base = k0 * S1 + ((n0 + k0) & 15);
#pragma unroll
    for (int j = 0; j < 16; ++j) {  // register j = n1
        pr[base + j * 16] = re[j];
        pi[base + j * 16] = im[j];
    }
}

// I1: load layout A with logical k0 at register dr(k0), conj twiddle, inverse
// pass 1.  Afterwards register j holds y[256 j + t] (times 4096).
template <bool POW = true>
OSZ_HD void i1(int t, double *re, double *im, const Tables &tb, const double *pr,
               const double *pi) {
    double twr[16], twi[16];
    if constexpr (POW) {
        t1_powers(t, tb, twr, twi);
    } else {
#pragma unroll
        for (int k = 1; k < 16; ++k) {
            twr[k] = tb.t1[(k * 256 + t) * 2];
            twi[k] = tb.t1[(k * 256 + t) * 2 + 1];
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k0 = dr(r);
        const int slot = k0 * S1 + (t & ~15) + ((t + k0) & 15);
        double a = pr[slot], b = pi[slot];
        if (k0 != 0) {
            const double wr = twr[k0], wi = -twi[k0];
            const double a2 = a * wr - b * wi;
            b = a * wi + b * wr;
            a = a2;
        }
        re[r] = a;
        im[r] = b;
    }
    inv16(re, im);
}

}  // namespace fft
}  // namespace osz
