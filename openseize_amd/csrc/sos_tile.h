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

}  // namespace osz
