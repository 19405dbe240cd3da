// sos_tile.h -- the pieces of the SOS time-parallel recurrence shared by the
// stand-alone kernels (sos.hip) and the fused FIR -> forward-SOS kernel
// (chain.hip): the per-section constant tables, the cross-lane helpers and the
// per-tile computation on a lane's block of T samples held in registers.
#pragma once

#include "common.h"

namespace osz {

constexpr int kSosT = 32;       // samples per lane per tile
constexpr int kSosNW = 4;       // waves per workgroup (one channel), one per SIMD
constexpr int kSosPad = 1;      // LDS row padding (doubles): odd row stride, conflict-free b64
constexpr int kSosMaxSec = 32;  // sections supported per handle

// Per-section constants, built on the host in osz_sos_create.
struct SosSection {
    double b0, b1, b2, a1, a2;
    double pad_[3];
    double G8[8][2];          // row 0 of A^r, r = 0..7: homogeneous y response inside an octet
    double A8[4];             // A^8: octet-to-octet step of the homogeneous response
    double P[4][4];           // A^(T*2^k), k = 0..3: scan steps inside a 16-lane row (DPP)
    double B[4];              // A^(16*T): row-to-row step
    double Q[4];              // A^(64*T): wave-to-wave step
    double PL16[16][4];       // A^(T*j), j = 0..15: row start state -> lane start state
    double AJ[kSosT + 1][4];  // A^j (per-lane lookup for the final state of a chunk)
};

__device__ __forceinline__ void mat2_apply(const double *M, double u0, double u1,
                                           double &r0, double &r1) {
    r0 = fma(M[0], u0, M[1] * u1);
    r1 = fma(M[2], u0, M[3] * u1);
}

// lane i <- lane i-D inside its 16-lane row, 0 for the first D lanes (DPP row_shr)
template <int D>
__device__ __forceinline__ double row_shr(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x110 + D, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x110 + D, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double lane_bcast(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// One tile of one forward pass on a whole tile (every lane holds T valid
// samples): v[0..T) of lane l of wave w are samples [(64 w + l) T, +T) of the
// tile, in place input -> output, section after section.  sst[2][kSosMaxSec][2]
// (LDS) carries the tile start state of every section, `parity` says which
// half is current and flips; agg[2][NW][2] (LDS) holds the wave aggregates.
// One workgroup barrier per section.  The same algorithm as sos_body in
// sos.hip (zero-state pass, DPP scan with A^(T 2^k), wave replay, octet
// fix-up), without its chunk-edge handling; T need not be a multiple of 8.
template <int T, int NW>
__device__ __forceinline__ void sos_tile_full(double *v, const SosSection *__restrict__ sec,
                                              int nsec, double *sst, double *agg, int &parity,
                                              int &aggbuf, int w, int l) {
    for (int s = 0; s < nsec; ++s) {
        const SosSection *__restrict__ S = sec + s;
        const double b0 = S->b0, b1 = S->b1, b2 = S->b2, na1 = -S->a1, na2 = -S->a2;
        double z0 = 0.0, z1 = 0.0;
#pragma unroll
        for (int j = 0; j < T; ++j) {
            const double xin = v[j];
            const double y = fma(b0, xin, z0);
            z0 = fma(na1, y, fma(b1, xin, z1));
            z1 = fma(na2, y, b2 * xin);
            v[j] = y;
        }
        const double pl0 = S->PL16[l & 15][0], pl1 = S->PL16[l & 15][1];
        const double pl2 = S->PL16[l & 15][2], pl3 = S->PL16[l & 15][3];
        double e0 = z0, e1 = z1;
#define OSZ_SCAN_STEP(K, D)                                   \
    {                                                         \
        const double u0 = row_shr<D>(e0), u1 = row_shr<D>(e1); \
        double r0, r1;                                        \
        mat2_apply(S->P[K], u0, u1, r0, r1);                  \
        e0 += r0;                                             \
        e1 += r1;                                             \
    }
        OSZ_SCAN_STEP(0, 1)
        OSZ_SCAN_STEP(1, 2)
        OSZ_SCAN_STEP(2, 4)
        OSZ_SCAN_STEP(3, 8)
#undef OSZ_SCAN_STEP
        const double p0 = row_shr<1>(e0), p1 = row_shr<1>(e1);  // exclusive, 0 at row start
        const double R00 = lane_bcast(e0, 15), R01 = lane_bcast(e1, 15);
        const double R10 = lane_bcast(e0, 31), R11 = lane_bcast(e1, 31);
        const double R20 = lane_bcast(e0, 47), R21 = lane_bcast(e1, 47);
        const double R30 = lane_bcast(e0, 63), R31 = lane_bcast(e1, 63);
        {   // wave aggregate from a zero start: E = B(B(B R0 + R1) + R2) + R3
            double y0 = R00, y1 = R01, t0, t1;
            mat2_apply(S->B, y0, y1, t0, t1); y0 = t0 + R10; y1 = t1 + R11;
            mat2_apply(S->B, y0, y1, t0, t1); y0 = t0 + R20; y1 = t1 + R21;
            mat2_apply(S->B, y0, y1, t0, t1); y0 = t0 + R30; y1 = t1 + R31;
            if (l == 0) {
                agg[(aggbuf * NW + w) * 2 + 0] = y0;
                agg[(aggbuf * NW + w) * 2 + 1] = y1;
            }
        }
        __syncthreads();
        double s0 = sst[(parity * kSosMaxSec + s) * 2 + 0];
        double s1 = sst[(parity * kSosMaxSec + s) * 2 + 1];
        double sw0 = s0, sw1 = s1;
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            if (q == w) {
                sw0 = s0;
                sw1 = s1;
            }
            double r0, r1;
            mat2_apply(S->Q, s0, s1, r0, r1);
            s0 = r0 + agg[(aggbuf * NW + q) * 2 + 0];
            s1 = r1 + agg[(aggbuf * NW + q) * 2 + 1];
        }
        if (threadIdx.x == 0) {
            sst[((parity ^ 1) * kSosMaxSec + s) * 2 + 0] = s0;
            sst[((parity ^ 1) * kSosMaxSec + s) * 2 + 1] = s1;
        }
        aggbuf ^= 1;
        double x10, x11, x20, x21, x30, x31;
        mat2_apply(S->B, sw0, sw1, x10, x11); x10 += R00; x11 += R01;
        mat2_apply(S->B, x10, x11, x20, x21); x20 += R10; x21 += R11;
        mat2_apply(S->B, x20, x21, x30, x31); x30 += R20; x31 += R21;
        const int row = l >> 4;
        const double xr0 = row == 0 ? sw0 : (row == 1 ? x10 : (row == 2 ? x20 : x30));
        const double xr1 = row == 0 ? sw1 : (row == 1 ? x11 : (row == 2 ? x21 : x31));
        double h0 = fma(pl0, xr0, fma(pl1, xr1, p0));
        double h1 = fma(pl2, xr0, fma(pl3, xr1, p1));
        // homogeneous fix-up, octet by octet: y[8q + r] += row0(A^r) (A^8)^q s
#pragma unroll
        for (int q = 0; q < (T + 7) / 8; ++q) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (8 * q + r < T)
                    v[8 * q + r] = fma(S->G8[r][0], h0, fma(S->G8[r][1], h1, v[8 * q + r]));
            if (8 * (q + 1) < T) {
                double n0, n1;
                mat2_apply(S->A8, h0, h1, n0, n1);
                h0 = n0;
                h1 = n1;
            }
        }
    }
    parity ^= 1;
}

constexpr int kSos2MaxSec = 8;                 // sections whose tables fit beside 3 workgroups per CU
constexpr int kSos2Tab = 66;                   // A^(T k), k = 0 .. 64, then one all-zero entry
constexpr int kSos2Zero = 65;


// (dpp_mov0: common.h)
// lane i <- lane i - D inside its 16-lane row, 0 for the first D lanes
template <int D>
__device__ __forceinline__ double dpp_row_shr0(double v) { return dpp_mov0<0x110 + D>(v); }
// lane i <- lane 15 of the previous row.  Measured (benchmarks/dpp_probe.hip):
// the lanes of row 0 have no source and keep THEIR OWN value, bound_ctrl or
// not -- the scan multiplies what rows 0 and 2 receive by an all-zero matrix.
__device__ __forceinline__ double dpp_bcast15(double v) { return dpp_mov0<0x142>(v); }
// lanes 32..63 <- lane 31; lanes 0..31 keep their own value (zero matrix again)
__device__ __forceinline__ double dpp_bcast31(double v) { return dpp_mov0<0x143>(v); }
// lane i <- lane i - 1 across the whole wave, 0 into lane 0
__device__ __forceinline__ double dpp_wave_shr1(double v) { return dpp_mov0<0x138>(v); }

// The same tile computation with the trimmed scan of sos_body2 (sos.hip): the
// inclusive prefix over the 64 lane blocks of a wave is formed lane-parallel
// (row_shr x 4, row_bcast:15, row_bcast:31) with per-lane matrices A^(T k) from
// a table in LDS (`tab`: [nsec][4][kSos2Tab], built for THIS T), wave_shr:1
// makes it exclusive; FMA-chained steps; b2 == 1 sections skip a multiply.
template <int T, int NW>
__device__ __forceinline__ void sos_tile_full2(double *v, const SosSection *__restrict__ sec,
                                               const double *tab, int nsec, double *sst,
                                               double *agg, int &parity, int &aggbuf, int w,
                                               int l) {
    for (int s = 0; s < nsec; ++s) {
        const SosSection *__restrict__ S = sec + s;
        const double b0 = S->b0, b1 = S->b1, b2 = S->b2, na1 = -S->a1, na2 = -S->a2;
        double z0 = 0.0, z1 = 0.0;
        if (b2 == 1.0) {
#pragma unroll
            for (int j = 0; j < T; ++j) {
                const double xin = v[j];
                const double y = fma(b0, xin, z0);
                z0 = fma(na1, y, fma(b1, xin, z1));
                z1 = fma(na2, y, xin);
                v[j] = y;
            }
        } else {
#pragma unroll
            for (int j = 0; j < T; ++j) {
                const double xin = v[j];
                const double y = fma(b0, xin, z0);
                z0 = fma(na1, y, fma(b1, xin, z1));
                z1 = fma(na2, y, b2 * xin);
                v[j] = y;
            }
        }
        int ll = l;
        asm volatile("" : "+v"(ll));
        const int ka = (ll & 16) ? (ll & 15) + 1 : kSos2Zero, kb = ll >= 32 ? ll - 31 : kSos2Zero, kc = ll;
        const double *ts = tab + s * 4 * kSos2Tab;
        const double ma0 = ts[0 * kSos2Tab + ka], ma1 = ts[1 * kSos2Tab + ka];
        const double ma2 = ts[2 * kSos2Tab + ka], ma3 = ts[3 * kSos2Tab + ka];
        double e0 = z0, e1 = z1;
#define OSZ_SCAN_STEP(K, D)                                             \
    {                                                                   \
        const double u0 = dpp_row_shr0<D>(e0), u1 = dpp_row_shr0<D>(e1); \
        e0 = fma(S->P[K][0], u0, fma(S->P[K][1], u1, e0));              \
        e1 = fma(S->P[K][2], u0, fma(S->P[K][3], u1, e1));              \
    }
        OSZ_SCAN_STEP(0, 1)
        OSZ_SCAN_STEP(1, 2)
        OSZ_SCAN_STEP(2, 4)
        OSZ_SCAN_STEP(3, 8)
#undef OSZ_SCAN_STEP
        {
            const double u0 = dpp_bcast15(e0), u1 = dpp_bcast15(e1);
            e0 = fma(ma0, u0, fma(ma1, u1, e0));
            e1 = fma(ma2, u0, fma(ma3, u1, e1));
        }
        {
            const double mb0 = ts[0 * kSos2Tab + kb], mb1 = ts[1 * kSos2Tab + kb];
            const double mb2 = ts[2 * kSos2Tab + kb], mb3 = ts[3 * kSos2Tab + kb];
            const double u0 = dpp_bcast31(e0), u1 = dpp_bcast31(e1);
            e0 = fma(mb0, u0, fma(mb1, u1, e0));
            e1 = fma(mb2, u0, fma(mb3, u1, e1));
        }
        if (l == 63) {
            agg[(aggbuf * NW + w) * 2 + 0] = e0;
            agg[(aggbuf * NW + w) * 2 + 1] = e1;
        }
        const double p0 = dpp_wave_shr1(e0), p1 = dpp_wave_shr1(e1);
        __syncthreads();
        const double mc0 = ts[0 * kSos2Tab + kc], mc1 = ts[1 * kSos2Tab + kc];
        const double mc2 = ts[2 * kSos2Tab + kc], mc3 = ts[3 * kSos2Tab + kc];
        double s0 = sst[(parity * kSosMaxSec + s) * 2 + 0];
        double s1 = sst[(parity * kSosMaxSec + s) * 2 + 1];
        double sw0 = s0, sw1 = s1;
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            if (q == w) {
                sw0 = s0;
                sw1 = s1;
            }
            const double g0 = agg[(aggbuf * NW + q) * 2 + 0], g1 = agg[(aggbuf * NW + q) * 2 + 1];
            const double r0 = fma(S->Q[0], s0, fma(S->Q[1], s1, g0));
            const double r1 = fma(S->Q[2], s0, fma(S->Q[3], s1, g1));
            s0 = r0;
            s1 = r1;
        }
        if (threadIdx.x == 0) {
            sst[((parity ^ 1) * kSosMaxSec + s) * 2 + 0] = s0;
            sst[((parity ^ 1) * kSosMaxSec + s) * 2 + 1] = s1;
        }
        aggbuf ^= 1;
        double h0 = fma(mc0, sw0, fma(mc1, sw1, p0));
        double h1 = fma(mc2, sw0, fma(mc3, sw1, p1));
#pragma unroll
        for (int q = 0; q < (T + 7) / 8; ++q) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (8 * q + r < T)
                    v[8 * q + r] = fma(S->G8[r][0], h0, fma(S->G8[r][1], h1, v[8 * q + r]));
            if (8 * (q + 1) < T) {
                double n0, n1;
                mat2_apply(S->A8, h0, h1, n0, n1);
                h0 = n0;
                h1 = n1;
            }
        }
    }
    parity ^= 1;
}

// ---- NaN reach ----------------------------------------------------------------
// A NaN (or an Inf, which the recurrence turns into a NaN within a sample or two)
// never leaves the state of a section: scipy.signal.sosfilt, and with it the
// reference (core/numerical.py:334, :399-410), yields NaN from that sample to the
// end of the STREAM in a forward pass (the carried zi stays NaN) and to the start of
// the CHUNK in a backward pass.  A time segment that starts from a zero state behind
// a pre-roll knows nothing of what lies before the pre-roll, so:
//  * backward pass of chunk i: it is NaN throughout exactly when the LAST sample of
//    what it is initialised from -- forward chunk i + 1, or chunk i itself for the
//    last one -- is not finite (the forward output is NaN from its first NaN on,
//    and the reference warms up over the whole of chunk i + 1).  Every workgroup
//    reads that one sample (`probe`) first and, if so, writes NaN instead of
//    filtering;
//  * forward pass: a small launch behind the pass (`sos_seal_launch`, sos.hip) looks at
//    the final output sample of every segment but the last; from the first one that is
//    not finite, the rest of the chunk and the carried state are overwritten with NaN.
//    (It used to be the workgroup of a channel that finished last, behind an arrival
//    counter: every workgroup then pays an agent-scope release -- a write-back of its
//    XCD's L2 -- on its way out, 5 % of a dual launch.)
__device__ __forceinline__ bool sos_not_finite(double v) { return !(fabs(v) <= 1.79769313486231570815e308); }

__device__ __forceinline__ void sos_fill_nan(double *__restrict__ y, int64_t n) {
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) y[i] = qnan;
}

__device__ __forceinline__ void sos_state_nan(double *__restrict__ state_out, int nsec, int nch, int c) {
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);
    if (state_out && (int)threadIdx.x < nsec) {
        state_out[((int64_t)threadIdx.x * nch + c) * 2 + 0] = qnan;
        state_out[((int64_t)threadIdx.x * nch + c) * 2 + 1] = qnan;
    }
}

}  // namespace osz
