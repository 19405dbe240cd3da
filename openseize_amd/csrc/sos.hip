// sos.hip -- K2/K3: cascaded second-order sections (DF2T biquads) on gfx950.
//
// Replaces scipy.signal.sosfilt as called by the reference at
// src/openseize/core/numerical.py:334 (forward, carried zi) and
// :399/:402/:410 (backward sweeps of sosfiltfilt on flipped chunks).
//
// Parallel decomposition (time-parallel linear recurrence, exact):
//   one workgroup per channel, NW waves; a "tile" is NW*64*T consecutive
//   samples (in processing order), every lane owns T consecutive samples.
//   Per section s (state z in R^2, z' = A z + B x, y = b0 x + z0):
//     1. each lane filters its T samples from ZERO state, in registers;
//        this yields y_zs[0..T) and the end state e_l;
//     2. in-wave Kogge-Stone scan over lanes with the constant 2x2 matrices
//        A^(T*2^k) gives the state at the end of every lane's block;
//        wave aggregates go through LDS and every wave replays the (<= NW)
//        wave-level steps with A^(64T);
//     3. each lane adds the homogeneous response of its true start state:
//        y[j] += (A^j s_l)[0]  (table of A^j rows, scalar loads).
//   The tile start state per section lives in LDS ("zi resident in LDS") and
//   is carried from tile to tile; chunk-to-chunk state is in the handle.
//   HBM traffic: each sample is read once and written once (16 B / sample).
//   Global<->lane-block transposition is staged through a padded LDS tile so
//   both the HBM side (consecutive lanes -> consecutive samples) and the LDS
//   side (row stride T+2 doubles) are conflict free.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "common.h"
#include "sos_tile.h"
#include "handles.h"

namespace osz {

struct SosArgs {
    const double *x;
    double *y;  // may be null: state-only pass (sosfiltfilt warm-up)
    int64_t ldx, ldy, n;
    const SosSection *sec;
    const double *state_in;  // (nsec, nch, 2) or null
    const double *zi_unit;   // (nsec, 2); used when state_in is null
    double *state_out;       // (nsec, nch, 2) or null
    int nsec, nch;
    const double *tab2;      // per-lane scan matrices for sos_body2 ([nsec][4][66]) or null
    int touch;               // sos_body2: touch-prefetch the next tile's window
    // sos_body2: rows processed BEFORE x, outputs discarded -- the chunk-local
    // warm-up of sosfiltfilt (numerical.py:397-399) as a pre-roll of the
    // backward pass instead of a launch of its own; npre a whole number of tiles
    const double *prex;
    int64_t ldprex, npre;
    // Non-finite samples (see "NaN reach" below).  probe: backward passes, one sample per
    // channel (row pitch ldprobe) whose being non-finite makes the whole pass NaN
    // (forward passes in time segments: sos_seal_launch behind the pass).
    const double *probe;
    int64_t ldprobe;
};

// In-kernel phase stamps for the diagnostic build only
// (benchmarks/sos_stamps.hip defines OSZ_SOS_STAMPS); the library build has none.
#ifdef OSZ_SOS_STAMPS
__device__ unsigned long long *g_sos_stamps = nullptr;   // [waves][8] cycle sums
#define OSZ_STAMP(slot)                                                              \
    do {                                                                             \
        unsigned long long now_;                                                     \
        __builtin_amdgcn_sched_barrier(0);                                           \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                           \
        stamp_acc[slot] += now_ - stamp_last;                                        \
        stamp_last = now_;                                                           \
    } while (0)
#else
#define OSZ_STAMP(slot) do { } while (0)
#endif


// NaN reach across time segments: see sos_tile.h
// backward pass: is channel c's pass NaN throughout?  (workgroup-uniform)
__device__ __forceinline__ bool sos_bwd_poisoned(const SosArgs &a, int c) {
    return a.probe != nullptr && sos_not_finite(a.probe[(int64_t)c * a.ldprobe]);
}


// GUARD = false: n is a whole number of tiles (the hot kernel: no bounds
// checks, no predication); GUARD = true handles a ragged remainder.
// LEAN = true: no register prefetch and the HBM<->lane-block transposition is
// staged through LDS in two halves of 32 rows: ~150 VGPRs and 34 KB of LDS per
// workgroup, so three to four workgroups share a CU (a single wave issues one
// float64 op per ~9 cycles; the f64 pipe needs several waves per SIMD).
template <int T, int NW, bool REV, bool GUARD, bool LEAN = false>
__device__ __forceinline__ void sos_body(const SosArgs &a, const SosSection *__restrict__ sec,
                                         const int c, const bool zero_init = false,
                                         const int64_t skip_store_tiles = 0) {
    constexpr int ROW = T + kSosPad;
    constexpr int WAVE_ELEMS = 64 * T;
    constexpr int STAGE_ROWS = LEAN ? 32 : 64;       // rows of the staging tile per wave
    extern __shared__ double lds[];
    double *tile = lds;                              // NW * STAGE_ROWS * ROW
    double *agg = tile + NW * STAGE_ROWS * ROW;      // [2][NW][2]
    double *sst = agg + 2 * NW * 2;                  // [2][kSosMaxSec][2]

    const int w = threadIdx.x >> 6;
    const int l = threadIdx.x & 63;
    const int64_t n = a.n;
    const double *xrow = a.x + (int64_t)c * a.ldx;
    double *yrow = a.y ? a.y + (int64_t)c * a.ldy : nullptr;
    double *wl = tile + w * STAGE_ROWS * ROW;        // this wave's private staging rows

    // ---- initial state of every section -> LDS slot 0
    if (threadIdx.x < a.nsec) {
        const int s = threadIdx.x;
        double z0, z1;
        if (zero_init) {
            z0 = z1 = 0.0;
        } else if (a.state_in) {
            z0 = a.state_in[((int64_t)s * a.nch + c) * 2 + 0];
            z1 = a.state_in[((int64_t)s * a.nch + c) * 2 + 1];
        } else {
            const double x0 = xrow[REV ? n - 1 : 0];
            z0 = a.zi_unit[2 * s + 0] * x0;
            z1 = a.zi_unit[2 * s + 1] * x0;
        }
        sst[(0 * kSosMaxSec + s) * 2 + 0] = z0;
        sst[(0 * kSosMaxSec + s) * 2 + 1] = z1;
    }
    __syncthreads();

    const int64_t tile_elems = (int64_t)NW * WAVE_ELEMS;
    const int64_t ntiles = (n + tile_elems - 1) / tile_elems;
    int parity = 0;  // which sst slot holds the current tile's start states
    int aggbuf = 0;

    // HBM -> registers for one tile: consecutive lanes, consecutive samples.
    // Issued one tile ahead so the loads fly while the current tile computes.
    double nx[LEAN ? 1 : T];
    auto fetch = [&](int64_t t) {
        if constexpr (LEAN) return;
        const int64_t pw = t * tile_elems + (int64_t)w * WAVE_ELEMS;
        const int64_t mem_base = REV ? (n - pw - WAVE_ELEMS) : pw;
        const bool full = !GUARD || (REV ? (mem_base >= 0) : (pw + WAVE_ELEMS <= n));
        if constexpr (!LEAN) {
            if (full) {
                const double *p = xrow + mem_base + l;   // one base, immediate offsets
#pragma unroll
                for (int i = 0; i < T; ++i) nx[i] = p[i * 64];
            } else {
#pragma unroll
                for (int i = 0; i < T; ++i) {
                    const int64_t g = mem_base + i * 64 + l;
                    nx[i] = (g >= 0 && g < n) ? xrow[g] : 0.0;
                }
            }
        }
    };
    fetch(0);
#ifdef OSZ_SOS_STAMPS
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory");
#endif

    for (int64_t t = 0; t < ntiles; ++t) {
        const int64_t pw = t * tile_elems + (int64_t)w * WAVE_ELEMS;  // processing-order start of this wave
        const int64_t mem_base = REV ? (n - pw - WAVE_ELEMS) : pw;
        const bool full = !GUARD || (REV ? (mem_base >= 0) : (pw + WAVE_ELEMS <= n));

        // ---- registers -> LDS rows (transpose), then lane block -> registers
        // lane l, step i <-> element m = 64 i + l of the wave window: row m / T, col m % T
        double *stage = wl + (l / T) * ROW + (l % T);
        constexpr int STEP = (64 / T) * ROW;
        double v[T];
        const int myrow = REV ? (63 - l) : l;        // which row of the wave window is mine
        if constexpr (!LEAN) {
#pragma unroll
            for (int i = 0; i < T; ++i) stage[i * STEP] = nx[i];
            wave_lds_fence();
            const double *blk = wl + myrow * ROW;
#pragma unroll
            for (int j = 0; j < T; ++j) v[j] = blk[REV ? (T - 1 - j) : j];
        } else {
            constexpr int HALF = 32 * T;             // samples per half window
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                double tmp[T / 2];
                const double *src = xrow + mem_base + hh * HALF;
                if (full && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
                    // 16 bytes per lane: half as many vector-memory instructions.
                    // Lane l, step i holds elements m = 128 i + 2 l + {0, 1} of the
                    // half window: LDS row m / T, column m % T (T = 32: row 4 i + l / 16)
                    const double2 *p2 = reinterpret_cast<const double2 *>(src) + l;
                    double2 t2[T / 4];
#pragma unroll
                    for (int i = 0; i < T / 4; ++i) t2[i] = p2[i * 64];
                    double *st2 = wl + ((2 * l) / T) * ROW + (2 * l) % T;
#pragma unroll
                    for (int i = 0; i < T / 4; ++i) {
                        st2[i * (128 / T) * ROW] = t2[i].x;
                        st2[i * (128 / T) * ROW + 1] = t2[i].y;
                    }
                    wave_lds_fence();
                    if ((myrow >> 5) == hh) {
                        const double *blk = wl + (myrow & 31) * ROW;
#pragma unroll
                        for (int j = 0; j < T; ++j) v[j] = blk[REV ? (T - 1 - j) : j];
                    }
                    wave_lds_fence();
                    continue;
                }
                if (full) {
                    const double *p = xrow + mem_base + hh * HALF + l;
#pragma unroll
                    for (int i = 0; i < T / 2; ++i) tmp[i] = p[i * 64];
                } else {
#pragma unroll
                    for (int i = 0; i < T / 2; ++i) {
                        const int64_t g = mem_base + hh * HALF + i * 64 + l;
                        tmp[i] = (g >= 0 && g < n) ? xrow[g] : 0.0;
                    }
                }
#pragma unroll
                for (int i = 0; i < T / 2; ++i) stage[i * STEP] = tmp[i];
                wave_lds_fence();
                if ((myrow >> 5) == hh) {
                    const double *blk = wl + (myrow & 31) * ROW;
#pragma unroll
                    for (int j = 0; j < T; ++j) v[j] = blk[REV ? (T - 1 - j) : j];
                }
                wave_lds_fence();
            }
        }
        OSZ_STAMP(0);   // stage in: registers -> LDS -> lane blocks (waits for the prefetch)
        if (t + 1 < ntiles) fetch(t + 1);

        // valid samples in this lane (prefix of its block)
        const int64_t pl = pw + (int64_t)l * T;
        const int64_t left = n - pl;
        const int cnt = (!GUARD || left >= T) ? T : (left > 0 ? (int)left : 0);
        const bool wave_full = !GUARD || (pw + WAVE_ELEMS <= n);
        const bool has_last = (cnt > 0) && (pl + cnt == n);

        for (int s = 0; s < a.nsec; ++s) {
            const SosSection *__restrict__ S = sec + s;
            const double b0 = S->b0, b1 = S->b1, b2 = S->b2, na1 = -S->a1, na2 = -S->a2;
            // 1. zero-state pass.  Samples past the end of the chunk were
            // loaded as zeros, so the pass needs no predication; only the lane
            // that owns the last sample also needs the state after `cnt`
            // samples (rare path, last tile only).
            double zc0 = 0.0, zc1 = 0.0;
            if (!wave_full) {
                double q0 = 0.0, q1 = 0.0;
                for (int j = 0; j < T; ++j) {
                    if (j == cnt) {
                        zc0 = q0;
                        zc1 = q1;
                    }
                    const double xin = v[j];
                    const double y = fma(b0, xin, q0);
                    q0 = fma(na1, y, fma(b1, xin, q1));
                    q1 = fma(na2, y, b2 * xin);
                }
                if (cnt == T) {
                    zc0 = q0;
                    zc1 = q1;
                }
            }
            double z0 = 0.0, z1 = 0.0;
            if (b2 == 1.0) {
                // b2 = 1 (every section but the first of a Butterworth /
                // Chebyshev / elliptic design): one multiply less per sample,
                // bit-identical (1.0 * x == x); the branch is wave-uniform
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
            const double e0raw = wave_full ? z0 : zc0, e1raw = wave_full ? z1 : zc1;
            // per-lane constants for later (issued early: the load flies meanwhile)
            const double pl0 = S->PL16[l & 15][0], pl1 = S->PL16[l & 15][1];
            const double pl2 = S->PL16[l & 15][2], pl3 = S->PL16[l & 15][3];
            OSZ_STAMP(1);   // zero-state pass
            // 2a. inclusive scan of end states inside each 16-lane row (DPP)
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
            // row totals, wave-uniform
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
            OSZ_STAMP(2);   // row scan + row totals + aggregate
            __syncthreads();
            OSZ_STAMP(3);   // workgroup barrier
            // 2b. wave-level replay (every wave, uniform values)
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
            // 2c. start state of each row, then of each lane
            double x10, x11, x20, x21, x30, x31;
            mat2_apply(S->B, sw0, sw1, x10, x11); x10 += R00; x11 += R01;
            mat2_apply(S->B, x10, x11, x20, x21); x20 += R10; x21 += R11;
            mat2_apply(S->B, x20, x21, x30, x31); x30 += R20; x31 += R21;
            const int row = l >> 4;
            const double xr0 = row == 0 ? sw0 : (row == 1 ? x10 : (row == 2 ? x20 : x30));
            const double xr1 = row == 0 ? sw1 : (row == 1 ? x11 : (row == 2 ? x21 : x31));
            double ls0 = fma(pl0, xr0, fma(pl1, xr1, p0));
            double ls1 = fma(pl2, xr0, fma(pl3, xr1, p1));
            OSZ_STAMP(4);   // wave replay + lane start state
            // 3. homogeneous fix-up, octet by octet: y[8q + r] += row0(A^r) (A^8)^q s
            {
                double h0 = ls0, h1 = ls1;
#pragma unroll
                for (int q = 0; q < T / 8; ++q) {
#pragma unroll
                    for (int r = 0; r < 8; ++r)
                        v[8 * q + r] = fma(S->G8[r][0], h0, fma(S->G8[r][1], h1, v[8 * q + r]));
                    if (q + 1 < T / 8) {
                        double n0, n1;
                        mat2_apply(S->A8, h0, h1, n0, n1);
                        h0 = n0;
                        h1 = n1;
                    }
                }
            }
            // final state of the chunk: the lane that owns the last sample
            if (has_last && a.state_out) {
                double f0, f1;
                mat2_apply(S->AJ[cnt], ls0, ls1, f0, f1);
                a.state_out[((int64_t)s * a.nch + c) * 2 + 0] = f0 + e0raw;
                a.state_out[((int64_t)s * a.nch + c) * 2 + 1] = f1 + e1raw;
            }
            OSZ_STAMP(5);   // fix-up (the final-state store included)
        }
        parity ^= 1;

        // ---- registers -> LDS rows -> HBM (wave private: no workgroup barrier)
        if constexpr (LEAN) {
            if (yrow && t >= skip_store_tiles) {
                constexpr int HALF = 32 * T;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    if ((myrow >> 5) == hh) {
                        double *blk = wl + (myrow & 31) * ROW;
#pragma unroll
                        for (int j = 0; j < T; ++j) blk[REV ? (T - 1 - j) : j] = v[j];
                    }
                    wave_lds_fence();
                    double *dst = yrow + mem_base + hh * HALF;
                    if (full && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
                        double2 *q2 = reinterpret_cast<double2 *>(dst) + l;
                        const double *st2 = wl + ((2 * l) / T) * ROW + (2 * l) % T;
#pragma unroll
                        for (int i = 0; i < T / 4; ++i) {
                            double2 o;
                            o.x = st2[i * (128 / T) * ROW];
                            o.y = st2[i * (128 / T) * ROW + 1];
                            q2[i * 64] = o;
                        }
                    } else if (full) {
                        double *q = yrow + mem_base + hh * HALF + l;
#pragma unroll
                        for (int i = 0; i < T / 2; ++i) q[i * 64] = stage[i * STEP];
                    } else {
#pragma unroll
                        for (int i = 0; i < T / 2; ++i) {
                            const int64_t g = mem_base + hh * HALF + i * 64 + l;
                            if (g >= 0 && g < n) yrow[g] = stage[i * STEP];
                        }
                    }
                    wave_lds_fence();
                }
            }
        } else if (yrow && t >= skip_store_tiles) {
            double *blk = wl + (REV ? (63 - l) : l) * ROW;
#pragma unroll
            for (int j = 0; j < T; ++j) blk[REV ? (T - 1 - j) : j] = v[j];
            wave_lds_fence();
            if (full) {
                double *q = yrow + mem_base + l;
#pragma unroll
                for (int i = 0; i < T; ++i) q[i * 64] = stage[i * STEP];
            } else {
#pragma unroll
                for (int i = 0; i < T; ++i) {
                    const int m = i * 64 + l;
                    const int64_t g = mem_base + m;
                    if (g >= 0 && g < n) yrow[g] = wl[(m / T) * ROW + (m % T)];
                }
            }
            wave_lds_fence();
        }
        OSZ_STAMP(6);   // stage out: lane blocks -> LDS -> HBM stores issued
    }
#ifdef OSZ_SOS_STAMPS
    if (l == 0 && g_sos_stamps) {
        unsigned long long *o = g_sos_stamps + ((size_t)c * NW + w) * 8;
        for (int i = 0; i < 8; ++i) o[i] = stamp_acc[i];
    }
#endif
}

// ---------------------------------------------------------------------------
// Second body of the same algorithm, trimmed for instruction count: the lean
// kernels were measured at ~75 % VALU utilisation (profiles/r02_pmc_bench.json:
// 89 vector instructions per sample against 42 for the bare recurrence + fix-up),
// so vector instructions, not bytes, bound them.  Differences from sos_body:
//   * the scan over the 64 lane blocks of a wave is lane-parallel end to end:
//     after the four row_shr steps inside 16-lane rows, two more DPP steps
//     (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3) extend
//     the inclusive prefix to the whole wave, and wave_shr:1 makes it
//     exclusive.  The per-lane matrices A^(T k) they need (k = lane position
//     inside the row / half wave / wave) come from a table in LDS, 65 x 4
//     doubles per section, loaded once per workgroup.  Gone: 8 readlanes per
//     state pair, the wave-uniform aggregate and row-start chains every lane
//     recomputed, and the cndmask row selection;
//   * the HBM <-> lane-block transposition is staged by COLUMN halves (all 64
//     rows x 16 columns at a time) instead of row halves: every lane reads
//     its own row in both halves, so there is no predicated half and no
//     register copies behind it; the staging buffer keeps its size;
//   * whole tiles only (n is a multiple of NW * 64 * T), state_out from LDS.
// Same arithmetic per sample; results agree with sos_body to rounding.
struct Sos2Lds {
    // doubles: staging NW * 64 * (T/2 + 1) | tables nsec * 4 * 66 | agg 2 * NW * 2 | sst 2 * 32 * 2
    static __host__ __device__ constexpr int stage(int T, int NW) { return NW * 64 * (T / 2 + 1); }
    static __host__ __device__ constexpr size_t bytes(int T, int NW, int nsec) {
        return sizeof(double) * ((size_t)stage(T, NW) + (size_t)nsec * 4 * kSos2Tab + 2 * NW * 2 +
                                 2 * kSosMaxSec * 2) + sizeof(int) * NW * 64;
    }
};

template <int T, int NW, bool REV, bool AL16, bool PF>
__device__ __forceinline__ void sos_body2(const SosArgs &a, const SosSection *__restrict__ sec,
                                          const double *__restrict__ gtab, const int c,
                                          const bool zero_init, const int64_t skip_store_tiles) {
    static_assert(T == 32, "column-half staging is laid out for T = 32");
    constexpr int HC = T / 2;                  // columns per staged half
    constexpr int ROWH = HC + 1;               // LDS row stride (doubles): odd, conflict-free b64
    constexpr int WAVE_ELEMS = 64 * T;
    extern __shared__ double lds[];
    double *tile = lds;                                        // NW * 64 * ROWH
    double *tab = tile + Sos2Lds::stage(T, NW);                // [nsec][4][66]
    double *agg = tab + a.nsec * 4 * kSos2Tab;                 // [2][NW][2]
    double *sst = agg + 2 * NW * 2;                            // [2][kSosMaxSec][2]
    int *dump = reinterpret_cast<int *>(sst + 2 * kSosMaxSec * 2);   // [NW][64] touch-prefetch sink

    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: an SGPR
    const int l = threadIdx.x & 63;
    const int64_t n = a.n;
    const double *xrow = a.x + (int64_t)c * a.ldx;
    double *yrow = a.y ? a.y + (int64_t)c * a.ldy : nullptr;
    double *wl = tile + w * 64 * ROWH;                         // this wave's private staging rows

    for (int i = threadIdx.x; i < a.nsec * 4 * kSos2Tab; i += NW * 64) tab[i] = gtab[i];
    if (threadIdx.x < a.nsec) {
        const int s = threadIdx.x;
        double z0, z1;
        if (zero_init) {
            z0 = z1 = 0.0;
        } else if (a.state_in) {
            z0 = a.state_in[((int64_t)s * a.nch + c) * 2 + 0];
            z1 = a.state_in[((int64_t)s * a.nch + c) * 2 + 1];
        } else {
            const double x0 = a.prex ? a.prex[(int64_t)c * a.ldprex + (REV ? a.npre - 1 : 0)]
                                     : xrow[REV ? n - 1 : 0];
            z0 = a.zi_unit[2 * s + 0] * x0;
            z1 = a.zi_unit[2 * s + 1] * x0;
        }
        sst[(0 * kSosMaxSec + s) * 2 + 0] = z0;
        sst[(0 * kSosMaxSec + s) * 2 + 1] = z1;
    }
    __syncthreads();

    const int64_t tile_elems = (int64_t)NW * WAVE_ELEMS;
    const int64_t ntiles = n / tile_elems;
    int parity = 0, aggbuf = 0;
    const int myrow = REV ? (63 - l) : l;                      // row of the wave window this lane owns
    // AL16: every row of x and y starts 16-byte aligned (decided by the host per
    // launch): 16 bytes per lane on the HBM side, otherwise 8
    constexpr bool al16 = AL16;
    // 16 doubles of one column half of a wave window, HBM -> registers
    double pf[HC];
    auto request_half = [&](double *dstv, const double *rowp, int64_t base, int hh) {
        const double *src = rowp + base;
        if (al16) {
            const double2 *p2 = reinterpret_cast<const double2 *>(src + (l >> 3) * T + HC * hh) + (l & 7);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const double2 q = p2[i * (8 * T / 2)];
                dstv[2 * i] = q.x;
                dstv[2 * i + 1] = q.y;
            }
        } else {
            const double *p1 = src + (l >> 4) * T + HC * hh + (l & 15);
#pragma unroll
            for (int i = 0; i < 16; ++i) dstv[i] = p1[i * 4 * T];
        }
    };
    if (PF && n >= (int64_t)NW * WAVE_ELEMS)
        request_half(pf, xrow, REV ? (n - (int64_t)w * WAVE_ELEMS - WAVE_ELEMS) : (int64_t)w * WAVE_ELEMS, 0);
    const double *prow = (!PF && a.prex) ? a.prex + (int64_t)c * a.ldprex : nullptr;
    const int64_t pre_tiles = prow ? a.npre / tile_elems : 0;

    for (int64_t tall = 0; tall < pre_tiles + ntiles; ++tall) {
        const bool pre = tall < pre_tiles;                      // a pre-roll tile: no output
        const int64_t t = pre ? tall : tall - pre_tiles;
        const int64_t pw = t * tile_elems + (int64_t)w * WAVE_ELEMS;
        const int64_t mem_base = REV ? ((pre ? a.npre : n) - pw - WAVE_ELEMS) : pw;
        const double *srow = pre ? prow : xrow;
        double v[T];
        // ---- HBM -> LDS -> lane blocks, two column halves; the first half was
        // requested a tile ago (below) and is in `pf` by now
        // both halves are requested before the first one is staged: the second
        // half's latency hides behind the first half's trip through LDS (the
        // lane blocks are still empty here, the registers are there)
        double pf1[HC];
        if (!PF) request_half(pf, srow, mem_base, 0);
        request_half(pf1, srow, mem_base, 1);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const double *src = hh == 0 ? pf : pf1;
            if (al16) {
                // lane l, step i: row 8 i + (l >> 3), columns 16 hh + 2 (l & 7) + {0, 1}
                double *st = wl + (l >> 3) * ROWH + 2 * (l & 7);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    st[i * 8 * ROWH] = src[2 * i];
                    st[i * 8 * ROWH + 1] = src[2 * i + 1];
                }
            } else {
                // 8 bytes per lane: row 4 i + (l >> 4), column 16 hh + (l & 15)
                double *st = wl + (l >> 4) * ROWH + (l & 15);
#pragma unroll
                for (int i = 0; i < 16; ++i) st[i * 4 * ROWH] = src[i];
            }
            wave_lds_fence();
            const double *blk = wl + myrow * ROWH;
#pragma unroll
            for (int j = 0; j < HC; ++j) v[REV ? (T - 1 - (HC * hh + j)) : (HC * hh + j)] = blk[j];
            wave_lds_fence();
        }

        // Touch the next tile's window (one dword per 128-byte line, two loads per
        // lane = all 128 lines of the 16 KB) so that it is on its way into L2 /
        // the Infinity Cache while this tile computes.
        if (a.touch && !pre && t + 1 < ntiles) {
            const int64_t nb = REV ? (mem_base - tile_elems) : (mem_base + tile_elems);
            const int *pn = reinterpret_cast<const int *>(xrow + nb) + l * 32;
            // LDS-DMA into a 256-byte dump area of this wave: no destination
            // registers, nothing to keep alive; the bytes are never read.  As an
            // asm statement: through the builtin the compiler orders every later
            // workgroup barrier behind these loads (s_waitcnt vmcnt(0)), which
            // is the stall the prefetch is there to remove.  Loads it does not
            // know about only make its own counted vmcnt waits stricter (they
            // are older than whatever it counts), never wrong.
            const unsigned lds_off = __builtin_amdgcn_readfirstlane(
                (unsigned)reinterpret_cast<uintptr_t>(dump) + (unsigned)w * 256u);
            unsigned m0_saved;
            asm volatile("s_mov_b32 %0, m0\n\t"
                         "s_mov_b32 m0, %1\n\t"
                         "s_nop 0\n\t"
                         "global_load_lds_dword %2, off\n\t"
                         "global_load_lds_dword %3, off\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(m0_saved)
                         : "s"(lds_off), "v"(pn), "v"(pn + 64 * 32)
                         : "memory");
        }

        for (int s = 0; s < a.nsec; ++s) {
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
            // per-lane matrices A^(T k) of the later scan steps, k = this lane's
            // position in its row (+1), in its half wave (+1), in the wave; the
            // indices are recomputed per section from an opaque copy of the lane
            // number (hoisted out of the tile loop they would cost registers the
            // kernel does not have at three workgroups per CU)
            int ll = l;
            asm volatile("" : "+v"(ll));
            const int ka = (ll & 16) ? (ll & 15) + 1 : kSos2Zero, kb = ll >= 32 ? ll - 31 : kSos2Zero, kc = ll;
            const double *ts = tab + s * 4 * kSos2Tab;
            const double ma0 = ts[0 * kSos2Tab + ka], ma1 = ts[1 * kSos2Tab + ka];
            const double ma2 = ts[2 * kSos2Tab + ka], ma3 = ts[3 * kSos2Tab + ka];
            // 1. inclusive scan inside 16-lane rows (constant matrices A^(T 2^k)),
            // every step e += M u as two chained FMAs per component
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
            // 2. rows 1, 3 take the total of the row before them (rows 0 and 2 read
            // the all-zero table entry: the broadcast reaches them too) ...
            {
                const double u0 = dpp_bcast15(e0), u1 = dpp_bcast15(e1);
                e0 = fma(ma0, u0, fma(ma1, u1, e0));
                e1 = fma(ma2, u0, fma(ma3, u1, e1));
            }
            // ... and lanes 32..63 the total of lanes 0..31: inclusive over the wave
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
            const double p0 = dpp_wave_shr1(e0), p1 = dpp_wave_shr1(e1);   // exclusive, 0 in lane 0
            __syncthreads();
            const double mc0 = ts[0 * kSos2Tab + kc], mc1 = ts[1 * kSos2Tab + kc];
            const double mc2 = ts[2 * kSos2Tab + kc], mc3 = ts[3 * kSos2Tab + kc];
            // 3. wave-level replay (uniform): start state of this wave, end state of the tile
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
            // 4. this lane's true start state and the homogeneous fix-up
            double h0 = fma(mc0, sw0, fma(mc1, sw1, p0));
            double h1 = fma(mc2, sw0, fma(mc3, sw1, p1));
#pragma unroll
            for (int q = 0; q < T / 8; ++q) {
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    v[8 * q + r] = fma(S->G8[r][0], h0, fma(S->G8[r][1], h1, v[8 * q + r]));
                if (q + 1 < T / 8) {
                    double n0, n1;
                    mat2_apply(S->A8, h0, h1, n0, n1);
                    h0 = n0;
                    h1 = n1;
                }
            }
        }
        parity ^= 1;

        // ---- lane blocks -> LDS -> HBM, two column halves.  Between them the
        // next tile's first half is requested (the 32 registers the first half
        // of the outputs just left): it lands while the second half is staged
        // out -- the section loop has no registers to spare, this code does.
        const bool store = yrow && !pre && t >= skip_store_tiles;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            if (store) {
                double *dst = yrow + mem_base;
                double *blk = wl + myrow * ROWH;
#pragma unroll
                for (int j = 0; j < HC; ++j) blk[j] = v[REV ? (T - 1 - (HC * hh + j)) : (HC * hh + j)];
                wave_lds_fence();
                if (al16) {
                    double2 *q2 = reinterpret_cast<double2 *>(dst + (l >> 3) * T + HC * hh) + (l & 7);
                    const double *st = wl + (l >> 3) * ROWH + 2 * (l & 7);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        double2 o;
                        o.x = st[i * 8 * ROWH];
                        o.y = st[i * 8 * ROWH + 1];
                        q2[i * (8 * T / 2)] = o;
                    }
                } else {
                    double *q1 = dst + (l >> 4) * T + HC * hh + (l & 15);
                    const double *st = wl + (l >> 4) * ROWH + (l & 15);
#pragma unroll
                    for (int i = 0; i < 16; ++i) q1[i * 4 * T] = st[i * 4 * ROWH];
                }
                wave_lds_fence();
            }
            __builtin_amdgcn_sched_barrier(0);
            if (PF && hh == 0 && t + 1 < ntiles)
                request_half(pf, xrow, REV ? (mem_base - tile_elems) : (mem_base + tile_elems), 0);
        }
    }
    // carried state of the chunk: the end state of its last tile
    if (a.state_out && threadIdx.x < a.nsec) {
        const int s = threadIdx.x;
        a.state_out[((int64_t)s * a.nch + c) * 2 + 0] = sst[(parity * kSosMaxSec + s) * 2 + 0];
        a.state_out[((int64_t)s * a.nch + c) * 2 + 1] = sst[(parity * kSosMaxSec + s) * 2 + 1];
    }
}

template <int T, int NW, bool REV, bool GUARD>
__global__ __launch_bounds__(NW * 64) void sos_kernel(SosArgs a,
                                                      const SosSection *__restrict__ sec) {
    if (REV && sos_bwd_poisoned(a, blockIdx.x)) {
        if (a.y) sos_fill_nan(a.y + (int64_t)blockIdx.x * a.ldy, a.n);
        sos_state_nan(a.state_out, a.nsec, a.nch, blockIdx.x);
        return;
    }
    sos_body<T, NW, REV, GUARD>(a, sec, blockIdx.x);
}

// Time segments of one pass as separate workgroups.  Segment s > 0 starts from
// a ZERO state `pre` samples early and discards those outputs; `pre` = the
// handle's warm_len, the point where the cascade's transition matrix has
// decayed below 1e-18, so the state it reaches at its first kept sample equals
// the carried one to float64 (see sos_warmup_len).  Segment 0 starts from the
// true state, the last segment writes the carried state.  Whole tiles only.
template <int T, int NW, bool REV, bool LEAN>
__device__ __forceinline__ void sos_segment(const SosArgs &a, const SosSection *__restrict__ sec,
                                            int c, int s, int nseg, int64_t seglen, int64_t pre) {
    const int64_t begin = (int64_t)s * seglen;               // processing-order start
    const int64_t len = (s == nseg - 1) ? a.n - begin : seglen;
    const int64_t p = s > 0 ? pre : 0;
    SosArgs b = a;
    b.n = len + p;
    if (REV) {
        // processing order runs down from a.n: kept samples are memory
        // [a.n - begin - len, a.n - begin), the pre-roll lies just above
        const int64_t lo = a.n - begin - len;
        b.x = a.x + lo;
        if (a.y) b.y = a.y + lo;
    } else {
        b.x = a.x + begin - p;
        if (a.y) b.y = a.y + begin - p;
    }
    if (s != nseg - 1) b.state_out = nullptr;
    if (REV && sos_bwd_poisoned(a, c)) {
        if (a.y) sos_fill_nan(a.y + (int64_t)c * a.ldy + (a.n - begin - len), len);
        sos_state_nan(b.state_out, a.nsec, a.nch, c);
        return;
    }
    sos_body<T, NW, REV, false, LEAN>(b, sec, c, s > 0, p / ((int64_t)NW * 64 * T));
}

// One pass, grid (nch, nseg): fills the chip when there are few channels.
template <int T, int NW, bool REV>
__global__ __launch_bounds__(NW * 64) void sos_split_kernel(SosArgs a,
                                                            const SosSection *__restrict__ sec,
                                                            int nseg, int64_t seglen,
                                                            int64_t pre) {
    sos_segment<T, NW, REV, false>(a, sec, blockIdx.x, blockIdx.y, nseg, seglen, pre);
}

// Lean variant (see sos_body): <= 168 VGPRs, 34 KB LDS -> three workgroups per CU.
template <int T, int NW, bool REV>
__global__ __launch_bounds__(NW * 64, 3) void sos_split_lean_kernel(
    SosArgs a, const SosSection *__restrict__ sec, int nseg, int64_t seglen, int64_t pre) {
    sos_segment<T, NW, REV, true>(a, sec, blockIdx.x, blockIdx.y, nseg, seglen, pre);
}

// Time segment of one pass on the trimmed body (sos_body2).
template <int T, int NW, bool REV, bool AL16, bool PF>
__device__ __forceinline__ void sos_segment2(const SosArgs &a, const SosSection *__restrict__ sec,
                                             const double *__restrict__ gtab, int c, int s, int nseg,
                                             int64_t seglen, int64_t pre) {
    const int64_t begin = (int64_t)s * seglen;
    const int64_t len = (s == nseg - 1) ? a.n - begin : seglen;
    const int64_t p = s > 0 ? pre : 0;
    SosArgs b = a;
    b.n = len + p;
    if (REV) {
        const int64_t lo = a.n - begin - len;
        b.x = a.x + lo;
        if (a.y) b.y = a.y + lo;
    } else {
        b.x = a.x + begin - p;
        if (a.y) b.y = a.y + begin - p;
    }
    if (s != nseg - 1) b.state_out = nullptr;
    if (s != 0) b.prex = nullptr;
    if (REV && sos_bwd_poisoned(a, c)) {
        if (a.y) sos_fill_nan(a.y + (int64_t)c * a.ldy + (a.n - begin - len), len);
        sos_state_nan(b.state_out, a.nsec, a.nch, c);
        return;
    }
    sos_body2<T, NW, REV, AL16, PF>(b, sec, gtab, c, s > 0, p / ((int64_t)NW * 64 * T));
}

template <int T, int NW, bool REV, bool AL16, bool PF>
__global__ __launch_bounds__(NW * 64, 3) void sos_split2_kernel(
    SosArgs a, const SosSection *__restrict__ sec, const double *__restrict__ gtab, int nseg,
    int64_t seglen, int64_t pre) {
    sos_segment2<T, NW, REV, AL16, PF>(a, sec, gtab, blockIdx.x, blockIdx.y, nseg, seglen, pre);
}

template <int T, int NW, bool AL16, bool PF>
__global__ __launch_bounds__(NW * 64, 3) void sos_dual2_kernel(
    SosArgs f, SosArgs b, const SosSection *__restrict__ sec, const double *__restrict__ gtab,
    int nseg, int64_t seglen_f, int64_t seglen_b, int64_t pre) {
    const int pass = blockIdx.y / nseg, s = blockIdx.y % nseg;
    if (pass == 0)
        sos_segment2<T, NW, false, AL16, PF>(f, sec, gtab, blockIdx.x, s, nseg, seglen_f, pre);
    else
        sos_segment2<T, NW, true, AL16, PF>(b, sec, gtab, blockIdx.x, s, nseg, seglen_b, pre);
}

// every row of the pass starts 16-byte aligned (segments and tiles keep it)
// OSZ_SOS_PF=1: request the next tile's first half into registers between the two
// output halves (A/B knob; measured SLOWER, 1.88 ms against 1.76 ms for the dual
// launch: the 32 extra live registers push the kernel into scratch spills)
static constexpr bool sos_pf() { return false; }

// (sos_tile.h) -- A/B knob for tests/test_gpu_nonfinite.py, which fails with it
bool sos_nanfix() { return true; }

// NaN reach of a forward pass cut into runs (sos_tile.h): run q ends before sample
// ((q + 1) A / B) C of the channel's row; from the first run whose last output is not finite
// the rest of the row, the carried section states and / or a carried row become NaN.
struct SealArgs {
    double *y;
    int64_t ldy, n;
    int nseg;
    int64_t A, B, C;
    double *state;       // (nsec, nch, 2) or null
    int nsec, nch;
    double *carry;       // (nch, ldcarry) or null
    int64_t ldcarry, ncarry;
};

__global__ void sos_seal_kernel(SealArgs g) {
    __shared__ int sbad;
    const int c = blockIdx.x, t = threadIdx.x;
    double *yr = g.y + (int64_t)c * g.ldy;
    if (t == 0) sbad = g.nseg;
    __syncthreads();
    for (int s = t; s < g.nseg - 1; s += blockDim.x) {
        const int64_t e = (((int64_t)(s + 1) * g.A) / g.B) * g.C;
        if (sos_not_finite(yr[e - 1])) {
            atomicMin(&sbad, s);
            break;
        }
    }
    __syncthreads();
    const int bad = sbad;
    if (bad == g.nseg) return;
    const int64_t from = (((int64_t)(bad + 1) * g.A) / g.B) * g.C;
    sos_fill_nan(yr + from, g.n - from);
    sos_state_nan(g.state, g.nsec, g.nch, c);
    if (g.carry) sos_fill_nan(g.carry + (int64_t)c * g.ldcarry, g.ncarry);
}

int sos_seal_launch(double *y, int64_t ldy, int64_t n, int nseg, int64_t A, int64_t B, int64_t C, double *state,
                    int nsec, int nch, double *carry, int64_t ldcarry, int64_t ncarry, hipStream_t st) {
    if (!sos_nanfix() || nseg < 2 || !y) return OSZ_OK;
    SealArgs g{y, ldy, n, nseg, A, B, C, state, nsec, nch, carry, ldcarry, ncarry};
    hipLaunchKernelGGL(sos_seal_kernel, dim3(nch), dim3(nseg > 65 ? 256 : 64), 0, st, g);
    OSZ_HIP(hipGetLastError());
    return OSZ_OK;
}

static bool sos_rows_aligned16(const SosArgs &a) {
    auto ok = [](const void *p, int64_t ld) {
        return p == nullptr || ((reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld & 1) == 0);
    };
    return ok(a.x, a.ldx) && ok(a.y, a.ldy) && ok(a.prex, a.ldprex) && (a.n & 1) == 0;
}

// Forward pass of one chunk and backward pass of another in ONE launch:
// grid (nch, 2).  The two passes are independent; two workgroups per CU
// (<= 256 VGPRs, 67 KB LDS each) fill each other's latency gaps, which one
// wave per SIMD cannot do alone.
template <int T, int NW>
__global__ __launch_bounds__(NW * 64, 2) void sos_dual_kernel(SosArgs f, SosArgs b,
                                                             const SosSection *__restrict__ sec) {
    if (blockIdx.y == 0)
        sos_body<T, NW, false, false>(f, sec, blockIdx.x);
    else if (sos_bwd_poisoned(b, blockIdx.x)) {
        if (b.y) sos_fill_nan(b.y + (int64_t)blockIdx.x * b.ldy, b.n);
        sos_state_nan(b.state_out, b.nsec, b.nch, blockIdx.x);
    } else
        sos_body<T, NW, true, false>(b, sec, blockIdx.x);
}

// Same with every pass cut into nseg time segments, lean bodies: grid
// (nch, 2 nseg) -> three workgroups per CU share the float64 pipe.
template <int T, int NW>
__global__ __launch_bounds__(NW * 64, 3) void sos_dual_lean_kernel(
    SosArgs f, SosArgs b, const SosSection *__restrict__ sec, int nseg, int64_t seglen_f,
    int64_t seglen_b, int64_t pre) {
    const int pass = blockIdx.y / nseg, s = blockIdx.y % nseg;
    if (pass == 0)
        sos_segment<T, NW, false, true>(f, sec, blockIdx.x, s, nseg, seglen_f, pre);
    else
        sos_segment<T, NW, true, true>(b, sec, blockIdx.x, s, nseg, seglen_b, pre);
}

// The constant matrices are powers of the companion matrix A.  For poles close
// to the unit circle A is nearly defective and products formed in float64 lose
// up to ~1e-9 relative accuracy at A^2048; they are therefore built in 80-bit
// long double on the host and rounded once.
typedef long double ld_t;

static void mat2_mul(const ld_t *X, const ld_t *Y, ld_t *Z) {
    ld_t r[4] = {X[0] * Y[0] + X[1] * Y[2], X[0] * Y[1] + X[1] * Y[3],
                 X[2] * Y[0] + X[3] * Y[2], X[2] * Y[1] + X[3] * Y[3]};
    for (int i = 0; i < 4; ++i) Z[i] = r[i];
}

static void mat2_store(const ld_t *M, double *out) {
    for (int i = 0; i < 4; ++i) out[i] = (double)M[i];
}

static void build_section(const double *c, SosSection &S, int T) {
    memset(&S, 0, sizeof S);
    const double a0 = c[3];
    S.b0 = c[0] / a0;
    S.b1 = c[1] / a0;
    S.b2 = c[2] / a0;
    S.a1 = c[4] / a0;
    S.a2 = c[5] / a0;
    const ld_t A[4] = {-(ld_t)S.a1, 1.0L, -(ld_t)S.a2, 0.0L};
    ld_t M[4] = {1, 0, 0, 1}, AT[4];
    for (int j = 0; j <= T; ++j) {
        mat2_store(M, S.AJ[j]);
        if (j < 8) {
            S.G8[j][0] = (double)M[0];
            S.G8[j][1] = (double)M[1];
        }
        if (j == 8) mat2_store(M, S.A8);
        if (j == T) memcpy(AT, M, sizeof M);
        mat2_mul(A, M, M);
    }
    // A^T, then squarings: A^(T 2^k)
    ld_t Pk[4];
    memcpy(Pk, AT, sizeof Pk);
    for (int k = 0; k < 4; ++k) {
        mat2_store(Pk, S.P[k]);
        mat2_mul(Pk, Pk, Pk);
    }
    mat2_store(Pk, S.B);      // A^(16 T)
    mat2_mul(Pk, Pk, Pk);
    mat2_mul(Pk, Pk, Pk);
    mat2_store(Pk, S.Q);      // A^(64 T)
    ld_t L[4] = {1, 0, 0, 1};
    for (int j = 0; j < 16; ++j) {
        mat2_store(L, S.PL16[j]);
        mat2_mul(AT, L, L);
    }
}

// Length of the chunk-local backward warm-up that is numerically complete.
// sosfiltfilt back-filters the whole next chunk only to obtain a start state
// (numerical.py:397-399); the influence of a sample that lies k samples away
// on that state is bounded by ||M^k|| (M: state-transition matrix of the
// whole cascade, transient growth included).  Once ||M^k||_inf < 1e-18 the
// remaining samples cannot change a float64 state, so the warm-up stops there.
// Returns a multiple of `quantum` samples, or `cap` when the cascade decays
// too slowly (the caller then warms up over the full chunk).
static int64_t sos_warmup_len(const std::vector<SosSection> &secs, int64_t quantum, int64_t cap) {
    const int ns = (int)secs.size(), d = 2 * ns;
    std::vector<ld_t> M((size_t)d * d, 0.0L);
    for (int col = 0; col < d; ++col) {  // one homogeneous step applied to e_col
        std::vector<ld_t> z(d, 0.0L), zn(d, 0.0L);
        z[col] = 1.0L;
        ld_t u = 0.0L;  // input of the current section (x = 0)
        for (int s = 0; s < ns; ++s) {
            const SosSection &S = secs[s];
            const ld_t y = (ld_t)S.b0 * u + z[2 * s];
            zn[2 * s] = (ld_t)S.b1 * u - (ld_t)S.a1 * y + z[2 * s + 1];
            zn[2 * s + 1] = (ld_t)S.b2 * u - (ld_t)S.a2 * y;
            u = y;
        }
        for (int r = 0; r < d; ++r) M[(size_t)r * d + col] = zn[r];
    }
    auto matmul = [d](const std::vector<ld_t> &X, const std::vector<ld_t> &Y) {
        std::vector<ld_t> Z((size_t)d * d, 0.0L);
        for (int i = 0; i < d; ++i)
            for (int k = 0; k < d; ++k) {
                const ld_t x = X[(size_t)i * d + k];
                if (x == 0.0L) continue;
                for (int j = 0; j < d; ++j) Z[(size_t)i * d + j] += x * Y[(size_t)k * d + j];
            }
        return Z;
    };
    auto norm = [d](const std::vector<ld_t> &X) {
        ld_t m = 0.0L;
        for (int i = 0; i < d; ++i) {
            ld_t r = 0.0L;
            for (int j = 0; j < d; ++j) r += fabsl(X[(size_t)i * d + j]);
            if (r > m) m = r;
        }
        return m;
    };
    // Mq = M^quantum by square-and-multiply
    std::vector<ld_t> Mq((size_t)d * d, 0.0L), base = M;
    for (int i = 0; i < d; ++i) Mq[(size_t)i * d + i] = 1.0L;
    for (int64_t e = quantum; e > 0; e >>= 1) {
        if (e & 1) Mq = matmul(Mq, base);
        base = matmul(base, base);
    }
    std::vector<ld_t> P = Mq;
    for (int64_t len = quantum; len < cap; len += quantum) {
        const ld_t nm = norm(P);
        if (!(nm == nm)) break;        // NaN: unstable cascade, no truncation
        if (nm < 1e-18L) return len;
        P = matmul(P, Mq);
    }
    return cap;
}

// A^(T k), k = 0 .. 64, per section, element-major ([sec][4][66]) for sos_body2
static void build_lane_table(const double *c, int T, double *out) {
    const double a0 = c[3];
    const ld_t A[4] = {-(ld_t)(c[4] / a0), 1.0L, -(ld_t)(c[5] / a0), 0.0L};
    ld_t AT[4] = {1, 0, 0, 1};
    for (int j = 0; j < T; ++j) mat2_mul(A, AT, AT);
    ld_t M[4] = {1, 0, 0, 1};
    for (int k = 0; k < kSos2Zero; ++k) {
        for (int e = 0; e < 4; ++e) out[e * kSos2Tab + k] = (double)M[e];
        mat2_mul(AT, M, M);
    }
    for (int e = 0; e < 4; ++e) out[e * kSos2Tab + kSos2Zero] = 0.0;
}

int sos_tables_for(osz_sos_s *h, int T, const SosSection **dsec) {
    if (T == h->T) {
        *dsec = h->dsec;
        return OSZ_OK;
    }
    if (T < 2 || T > 32) return fail(OSZ_ERR_INVALID, "sos_tables_for: T=%d not in [2, 32]", T);
    if (!h->dsec_t[T]) {
        std::vector<SosSection> secs(h->nsec);
        for (int s = 0; s < h->nsec; ++s) build_section(h->coef + 6 * s, secs[s], T);
        OSZ_HIP(hipMalloc(&h->dsec_t[T], sizeof(SosSection) * h->nsec));
        OSZ_HIP(hipMemcpy(h->dsec_t[T], secs.data(), sizeof(SosSection) * h->nsec,
                          hipMemcpyHostToDevice));
    }
    *dsec = h->dsec_t[T];
    return OSZ_OK;
}

int sos_lane_table_for(osz_sos_s *h, int T, const double **dtab) {
    *dtab = nullptr;
    if (h->nsec > kSos2MaxSec) return OSZ_OK;
    if (T < 2 || T > 32) return fail(OSZ_ERR_INVALID, "sos_lane_table_for: T=%d not in [2, 32]", T);
    if (!h->dtab2_t[T]) {
        std::vector<double> tab((size_t)h->nsec * 4 * kSos2Tab);
        for (int s = 0; s < h->nsec; ++s)
            build_lane_table(h->coef + 6 * s, T, tab.data() + (size_t)s * 4 * kSos2Tab);
        OSZ_HIP(hipMalloc(&h->dtab2_t[T], tab.size() * sizeof(double)));
        OSZ_HIP(hipMemcpy(h->dtab2_t[T], tab.data(), tab.size() * sizeof(double),
                          hipMemcpyHostToDevice));
    }
    *dtab = h->dtab2_t[T];
    return OSZ_OK;
}

}  // namespace osz

using namespace osz;

template <int T, int NW, bool REV, bool GUARD>
static int sos_launch_one(const SosArgs &a, hipStream_t st) {
    auto kern = sos_kernel<T, NW, REV, GUARD>;
    const size_t lds = sizeof(double) * ((size_t)NW * 64 * (T + kSosPad) + 2 * NW * 2 +
                                         2 * kSosMaxSec * 2);
    OSZ_DYN_LDS(kern, lds);
    {
        KernelTimer kt(REV ? (a.y ? (GUARD ? "sos_bwd_tail" : "sos_bwd") : "sos_warmup")
                           : (GUARD ? "sos_fwd_tail" : "sos_fwd"), st);
        hipLaunchKernelGGL(kern, dim3(a.nch), dim3(NW * 64), lds, st, a, a.sec);
    }
    OSZ_HIP(hipGetLastError());
    return OSZ_OK;
}

// One logical pass = the hot kernel over the whole tiles + a guarded kernel
// over the ragged remainder; the cascade state travels through `carry`
// (nsec, nch, 2) between the two launches.
// lean bodies (no register prefetch, half-tile staging) at three workgroups
// per CU (measured 3-5 % faster than the prefetching bodies at two; the switch
// between them is gone)
static constexpr bool sos_lean() { return true; }

// whole-tile pass, cut into time segments when there are too few channels to
// fill the chip (see sos_split_kernel)
// time segments a whole-tile pass of `a` is cut into (1: one workgroup per channel)
static int64_t sos_plan_segments(const SosArgs &a, int64_t tile, int64_t warm_len) {
    const int64_t ntiles = a.n / tile, pre_tiles = warm_len / tile;
    int64_t nseg = 1;
    const int target_wgs = sos_lean() ? 768 : 512;   // three (lean bodies) or two workgroups per CU
    // in place (y aliases x) the pre-roll of segment s would read what segment
    // s - 1 is writing: one workgroup per channel then (it reads a tile before
    // it writes it)
    const bool in_place = a.y != nullptr && a.y == a.x;
    if (!in_place && a.nch < target_wgs && pre_tiles >= 1 && pre_tiles * tile == warm_len) {
        nseg = (target_wgs + a.nch - 1) / a.nch;
        const int64_t max_seg = ntiles / (4 * pre_tiles);    // pre-roll <= 25 % extra work
        if (nseg > max_seg) nseg = max_seg;
    }
    return nseg;
}

template <int T, int NW, bool REV>
static int sos_launch_main(const SosArgs &a, int64_t warm_len, hipStream_t st) {
    const int64_t tile = (int64_t)NW * 64 * T;
    const int64_t ntiles = a.n / tile;
    int64_t nseg = sos_plan_segments(a, tile, warm_len);
    if (nseg <= 1) return sos_launch_one<T, NW, REV, false>(a, st);
    const int64_t seg_tiles = (ntiles + nseg - 1) / nseg;
    nseg = (ntiles + seg_tiles - 1) / seg_tiles;
    const bool lean = sos_lean();
    if constexpr (T == 32 && NW == 4) {
        if (lean && a.tab2) {
            const bool al = sos_rows_aligned16(a);
            auto k2 = sos_pf() ? (al ? sos_split2_kernel<T, NW, REV, true, true>
                                     : sos_split2_kernel<T, NW, REV, false, true>)
                               : (al ? sos_split2_kernel<T, NW, REV, true, false>
                                     : sos_split2_kernel<T, NW, REV, false, false>);
            const size_t lds2 = Sos2Lds::bytes(T, NW, a.nsec);
            OSZ_DYN_LDS(k2, lds2);
            {
                KernelTimer kt(REV ? (a.y ? "sos_bwd_split" : "sos_warmup") : "sos_fwd_split", st);
                hipLaunchKernelGGL(k2, dim3(a.nch, (unsigned)nseg), dim3(NW * 64), lds2, st, a, a.sec,
                                   a.tab2, (int)nseg, seg_tiles * tile, warm_len);
            }
            OSZ_HIP(hipGetLastError());
            if (!REV)
                return sos_seal_launch(a.y, a.ldy, a.n, (int)nseg, 1, 1, seg_tiles * tile, a.state_out, a.nsec,
                                       a.nch, nullptr, 0, 0, st);
            return OSZ_OK;
        }
    }
    auto kern = lean ? sos_split_lean_kernel<T, NW, REV> : sos_split_kernel<T, NW, REV>;
    const size_t lds = sizeof(double) * ((size_t)NW * (lean ? 32 : 64) * (T + kSosPad) +
                                         2 * NW * 2 + 2 * kSosMaxSec * 2);
    OSZ_DYN_LDS(kern, lds);
    {
        KernelTimer kt(REV ? (a.y ? "sos_bwd_split" : "sos_warmup") : "sos_fwd_split", st);
        hipLaunchKernelGGL(kern, dim3(a.nch, (unsigned)nseg), dim3(NW * 64), lds, st, a, a.sec,
                           (int)nseg, seg_tiles * tile, warm_len);
    }
    OSZ_HIP(hipGetLastError());
    if (!REV)
        return sos_seal_launch(a.y, a.ldy, a.n, (int)nseg, 1, 1, seg_tiles * tile, a.state_out, a.nsec, a.nch,
                               nullptr, 0, 0, st);
    return OSZ_OK;
}

template <int T, int NW, bool REV>
static int sos_launch_tn(const SosArgs &a0, double *carry, int64_t warm_len, hipStream_t st) {
    const int64_t tile = (int64_t)NW * 64 * T;
    const int64_t nfull = (a0.n / tile) * tile, rem = a0.n - nfull;
    if (nfull == 0 || rem == 0) {
        return rem ? sos_launch_one<T, NW, REV, true>(a0, st)
                   : sos_launch_main<T, NW, REV>(a0, warm_len, st);
    }
    SosArgs m = a0, r = a0;  // main (whole tiles) first in processing order, then remainder
    m.n = nfull;
    r.n = rem;
    if (REV) {
        m.x = a0.x + rem;
        if (a0.y) m.y = a0.y + rem;
    } else {
        r.x = a0.x + nfull;
        if (a0.y) r.y = a0.y + nfull;
    }
    m.state_out = carry;
    r.state_in = carry;
    r.prex = nullptr;   // a pre-roll belongs to the launch that runs first
    int rc = sos_launch_main<T, NW, REV>(m, warm_len, st);
    if (rc) return rc;
    return sos_launch_one<T, NW, REV, true>(r, st);
}

template <bool REV>
static int sos_launch(const SosArgs &a, double *carry, int T, int NW, int64_t warm_len,
                      hipStream_t st) {
    if (T == 32 && NW == 4) return sos_launch_tn<32, 4, REV>(a, carry, warm_len, st);
    if (T == 32 && NW == 8) return sos_launch_tn<32, 8, REV>(a, carry, warm_len, st);
    if (T == 16 && NW == 8) return sos_launch_tn<16, 8, REV>(a, carry, warm_len, st);
    if (T == 16 && NW == 4) return sos_launch_tn<16, 4, REV>(a, carry, warm_len, st);
    return fail(OSZ_ERR_INVALID, "sos: unsupported geometry T=%d NW=%d", T, NW);
}

template <int T, int NW>
static int sos_launch_dual(const SosArgs &f, const SosArgs &b, hipStream_t st) {
    auto kern = sos_dual_kernel<T, NW>;
    const size_t lds = sizeof(double) * ((size_t)NW * 64 * (T + kSosPad) + 2 * NW * 2 +
                                         2 * kSosMaxSec * 2);
    OSZ_DYN_LDS(kern, lds);
    {
        KernelTimer kt("sos_dual", st);
        hipLaunchKernelGGL(kern, dim3(f.nch, 2), dim3(NW * 64), lds, st, f, b, f.sec);
    }
    OSZ_HIP(hipGetLastError());
    return OSZ_OK;
}

namespace osz {
int sosfiltfilt_chunk_on(osz_sos_s *h, const double *fa, int64_t ldfa, int64_t na, const double *fb,
                         int64_t ldfb, int64_t nb, double *y, int64_t ldy, double *tmp, double *carry,
                         hipStream_t st) {
    SosArgs a{};
    a.sec = h->dsec;
    a.nsec = h->nsec;
    a.nch = h->nch;
    a.zi_unit = h->dzi;
    a.tab2 = h->dtab2;
    a.touch = h->touch;
    // NaN reach of the backward pass: the last sample of what it is initialised from
    a.probe = !sos_nanfix() ? nullptr : fb ? fb + (nb - 1) : fa + (na - 1);
    a.ldprobe = fb ? ldfb : ldfa;
    // The warm-up rides the backward launch as a pre-roll of its first time segment
    // (sos_body2) when that launch is the trimmed split kernel on whole tiles; a
    // launch of its own otherwise.  (Beside a kernel that fills the chip --
    // osz_chain_step -- a tiny separate launch waits a millisecond for a free CU.)
    const int64_t tile = (int64_t)h->NW * 64 * h->T;
    bool preroll = false;
    if (fb && h->T == 32 && h->NW == 4 && sos_lean() && h->dtab2 && !sos_pf() && h->warm_len == tile &&
        nb >= h->warm_len && na >= 8 * tile) {
        SosArgs m{};
        m.nch = h->nch;
        m.n = (na / tile) * tile;
        m.x = fa;
        m.y = y;
        preroll = sos_plan_segments(m, tile, h->warm_len) >= 2;
    }
    if (preroll) {
        a.prex = fb;
        a.ldprex = ldfb;
        a.npre = h->warm_len;
        a.state_in = nullptr;
    } else if (fb) {
        // warm-up over the next chunk: state only (numerical.py:397-399)
        // only the warm_len samples next to chunk a can influence the state
        a.x = fb;
        a.ldx = ldfb;
        a.n = nb < h->warm_len ? nb : h->warm_len;
        a.y = nullptr;
        a.ldy = 0;
        a.state_in = nullptr;
        a.state_out = tmp;
        int rc = sos_launch<true>(a, carry, h->T, h->NW, h->warm_len, st);
        if (rc) return rc;
        a.state_in = tmp;
    } else {
        a.state_in = nullptr;  // zi_unit * fa[:, na-1]  (numerical.py:408-410)
    }
    a.x = fa;
    a.ldx = ldfa;
    a.n = na;
    a.y = y;
    a.ldy = ldy;
    a.state_out = nullptr;
    return sos_launch<true>(a, carry, h->T, h->NW, h->warm_len, st);
}
}  // namespace osz

extern "C" {

int osz_sos_create(osz_sos_t *h, const double *sos, int nsec, int nch) {
    OSZ_REQUIRE(h && sos, "osz_sos_create: null argument");
    OSZ_REQUIRE(nsec >= 1 && nsec <= kSosMaxSec, "osz_sos_create: nsec=%d not in [1, %d]",
                nsec, kSosMaxSec);
    OSZ_REQUIRE(nch >= 1, "osz_sos_create: nch=%d must be positive", nch);
    for (int s = 0; s < nsec; ++s)
        OSZ_REQUIRE(sos[6 * s + 3] == 1.0, "sos[:, 3] should be all ones (section %d)", s);
    // kernel geometry: 32 samples per lane, four waves per workgroup
    const int T = kSosT, NW = kSosNW;
    std::vector<SosSection> secs(nsec);
    for (int s = 0; s < nsec; ++s) build_section(sos + 6 * s, secs[s], T);
    osz_sos_s *p = new osz_sos_s();
    p->T = T;
    p->NW = NW;
    p->nsec = nsec;
    p->nch = nch;
    for (int i = 0; i < 33; ++i) p->dsec_t[i] = nullptr;
    for (int i = 0; i < 33; ++i) p->dtab2_t[i] = nullptr;
    for (int i = 0; i < 6 * nsec; ++i) p->coef[i] = sos[i];
    const size_t sb = sizeof(double) * (size_t)nsec * nch * 2;
    OSZ_HIP(hipMalloc(&p->dsec, sizeof(SosSection) * nsec));
    OSZ_HIP(hipGetDevice(&p->device));
    OSZ_HIP(hipMalloc(&p->dstate, sb));
    OSZ_HIP(hipMalloc(&p->dstate_alt, sb));
    OSZ_HIP(hipMalloc(&p->dtmp, sb));
    OSZ_HIP(hipMalloc(&p->dcarry, sb));
    OSZ_HIP(hipMalloc(&p->dzi, sizeof(double) * nsec * 2));
    OSZ_HIP(hipMemcpy(p->dsec, secs.data(), sizeof(SosSection) * nsec, hipMemcpyHostToDevice));
    OSZ_HIP(hipMemset(p->dstate, 0, sb));
    p->dtab2 = nullptr;
    p->side = nullptr;
    p->side_go = p->side_done[0] = p->side_done[1] = nullptr;
    p->side_cur = 0;
    p->dtmp_side = p->dcarry_side = nullptr;
    p->side_busy = false;
    p->spec = nullptr;
    p->zp = nullptr;
    p->zp_tol = 0.0;
    p->touch = 0;
    {
        // the trimmed lean body (sos_body2): T = 32, NW = 4, up to 8 sections
        if (T == 32 && NW == 4 && nsec <= kSos2MaxSec) {
            std::vector<double> tab((size_t)nsec * 4 * kSos2Tab);
            for (int s = 0; s < nsec; ++s) build_lane_table(sos + 6 * s, T, tab.data() + (size_t)s * 4 * kSos2Tab);
            OSZ_HIP(hipMalloc(&p->dtab2, tab.size() * sizeof(double)));
            OSZ_HIP(hipMemcpy(p->dtab2, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
        }
    }
    // steady-state unit-step state of each section (what scipy.signal.sosfilt_zi
    // returns at numerical.py:378): with section DC gain g and unit input,
    // z0 = g - b0, z1 = b2 - a2*g; the next section sees the input scaled by g.
    std::vector<double> zi(2 * nsec);
    double scale = 1.0;
    for (int s = 0; s < nsec; ++s) {
        const SosSection &S = secs[s];
        const double g = (S.b0 + S.b1 + S.b2) / (1.0 + S.a1 + S.a2);
        zi[2 * s + 0] = scale * (g - S.b0);
        zi[2 * s + 1] = scale * (S.b2 - S.a2 * g);
        scale *= g;
    }
    OSZ_HIP(hipMemcpy(p->dzi, zi.data(), sizeof(double) * 2 * nsec, hipMemcpyHostToDevice));
    {
        // not converged within the cap (slowly decaying or unstable cascade):
        // no truncated warm-up and no zero-start time segments for this filter
        const int64_t cap = (int64_t)1 << 24;
        const int64_t wl = sos_warmup_len(secs, (int64_t)NW * 64 * T, cap);
        p->warm_len = wl >= cap ? ((int64_t)1 << 62) : wl;
    }
    *h = p;
    return OSZ_OK;
}

int64_t osz_sos_warmup_len(osz_sos_t h) { return h ? h->warm_len : -1; }

int osz_sos_set_warmup_len(osz_sos_t h, int64_t len) {
    OSZ_REQUIRE(h && len >= 0, "osz_sos_set_warmup_len: bad argument");
    h->warm_len = len == 0 ? ((int64_t)1 << 62) : len;
    return OSZ_OK;
}

int osz_sos_set_zi_unit(osz_sos_t h, const double *zi_unit) {
    OSZ_REQUIRE(h && zi_unit, "osz_sos_set_zi_unit: null argument");
    OSZ_HIP(hipMemcpy(h->dzi, zi_unit, sizeof(double) * 2 * h->nsec, hipMemcpyHostToDevice));
    return OSZ_OK;
}

/* samples of the next chunk the chunk-local backward warm-up reads (1 << 62: all of it) */
int64_t osz_sos_warm_len(osz_sos_t h) { return h ? h->warm_len : -1; }

int osz_sos_destroy(osz_sos_t h) {
    if (!h) return OSZ_OK;
    spec_unlink(h->spec);
    zp_unlink(h->zp);
    if (h->side) (void)hipStreamSynchronize(h->side);   // a deferred backward pass of osz_chain_step
    (void)hipFree(h->dsec);
    for (int i = 0; i < 33; ++i) (void)hipFree(h->dsec_t[i]);
    for (int i = 0; i < 33; ++i) (void)hipFree(h->dtab2_t[i]);
    (void)hipFree(h->dstate);
    (void)hipFree(h->dstate_alt);
    (void)hipFree(h->dtmp);
    (void)hipFree(h->dcarry);
    (void)hipFree(h->dzi);
    (void)hipFree(h->dtab2);
    (void)hipFree(h->dtmp_side);
    (void)hipFree(h->dcarry_side);
    if (h->side_go) (void)hipEventDestroy(h->side_go);
    for (int q = 0; q < 2; ++q)
        if (h->side_done[q]) (void)hipEventDestroy(h->side_done[q]);
    if (h->side) (void)hipStreamDestroy(h->side);
    delete h;
    return OSZ_OK;
}

int osz_sos_set_state(osz_sos_t h, const double *zi, void *stream) {
    OSZ_REQUIRE(h, "osz_sos_set_state: null handle");
    const size_t sb = sizeof(double) * (size_t)h->nsec * h->nch * 2;
    hipStream_t st = as_stream(stream);
    {
        int rc = spec_touch(h->spec, st);
        if (rc) return rc;
    }
    const bool dev = zi && on_device(zi);
    if (zi) {
        OSZ_HIP(hipMemcpyAsync(h->dstate, zi, sb, dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    } else {
        OSZ_HIP(hipMemsetAsync(h->dstate, 0, sb, st));
    }
    if (!dev) OSZ_HIP(hipStreamSynchronize(st));
    return OSZ_OK;
}

int osz_sos_get_state(osz_sos_t h, double *zf, void *stream) {
    OSZ_REQUIRE(h && zf, "osz_sos_get_state: null argument");
    const size_t sb = sizeof(double) * (size_t)h->nsec * h->nch * 2;
    hipStream_t st = as_stream(stream);
    {
        int rc = spec_settle(h->spec, st);
        if (rc) return rc;
    }
    const bool dev = on_device(zf);
    OSZ_HIP(hipMemcpyAsync(zf, h->dstate, sb, dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
    if (!dev) OSZ_HIP(hipStreamSynchronize(st));
    return OSZ_OK;
}

__global__ void sos_scale_state_kernel(double *state, const double *zi_unit, const double *x,
                                       int64_t ldx, int64_t col, int nsec, int nch) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nsec * nch) return;
    const int s = i / nch, c = i % nch;
    const double x0 = x[(int64_t)c * ldx + col];
    state[2 * i + 0] = zi_unit[2 * s + 0] * x0;
    state[2 * i + 1] = zi_unit[2 * s + 1] * x0;
}

int osz_sos_set_state_scaled(osz_sos_t h, const double *x, int64_t ldx, int64_t col,
                             void *stream) {
    OSZ_REQUIRE(h && x, "osz_sos_set_state_scaled: null argument");
    hipStream_t st = as_stream(stream);
    {
        int rc = spec_touch(h->spec, st);
        if (rc) return rc;
    }
    const int total = h->nsec * h->nch;
    hipLaunchKernelGGL(sos_scale_state_kernel, dim3((total + 255) / 256), dim3(256), 0, st,
                       h->dstate, h->dzi, x, ldx, col, h->nsec, h->nch);
    OSZ_HIP(hipGetLastError());
    return OSZ_OK;
}

int osz_sos_forward(osz_sos_t h, const double *x, int64_t ldx, double *y, int64_t ldy, int64_t n,
                    void *stream) {
    OSZ_REQUIRE(h && x && y, "osz_sos_forward: null argument");
    OSZ_REQUIRE(n >= 0 && ldx >= n && ldy >= n, "osz_sos_forward: n=%lld ldx=%lld ldy=%lld",
                (long long)n, (long long)ldx, (long long)ldy);
    if (n == 0) return OSZ_OK;
    OSZ_SAME_DEVICE(h, "osz_sos_forward");
    {
        int rc = spec_touch(h->spec, as_stream(stream));
        if (rc) return rc;
    }
    return sos_forward_raw(h, x, ldx, y, ldy, n, as_stream(stream));
}

}  // extern "C"

int osz::sos_forward_raw(osz_sos_s *h, const double *x, int64_t ldx, double *y, int64_t ldy, int64_t n,
                         hipStream_t st_) {
    void *stream = st_;
    SosArgs a{};
    a.x = x;
    a.y = y;
    a.ldx = ldx;
    a.ldy = ldy;
    a.n = n;
    a.sec = h->dsec;
    a.state_in = h->dstate;
    a.zi_unit = nullptr;
    a.state_out = h->dstate_alt;
    a.nsec = h->nsec;
    a.nch = h->nch;
    a.tab2 = h->dtab2;
    a.touch = h->touch;
    int rc = sos_launch<false>(a, h->dcarry, h->T, h->NW, h->warm_len, as_stream(stream));
    if (rc) return rc;
    std::swap(h->dstate, h->dstate_alt);
    return OSZ_OK;
}

int osz::sos_backward_raw(osz_sos_s *h, const double *x, int64_t ldx, double *y, int64_t ldy, int64_t n,
                          const double *state, hipStream_t st) {
    SosArgs a{};
    a.x = x;
    a.y = y;
    a.ldx = ldx;
    a.ldy = ldy;
    a.n = n;
    a.sec = h->dsec;
    a.state_in = state;
    a.zi_unit = nullptr;
    a.state_out = nullptr;
    a.nsec = h->nsec;
    a.nch = h->nch;
    a.tab2 = h->dtab2;
    a.touch = h->touch;
    return sos_launch<true>(a, h->dcarry, h->T, h->NW, h->warm_len, st);
}

extern "C" {

int osz_sosfiltfilt_step(osz_sos_t h, const double *x, int64_t ldx, int64_t nx, double *f,
                         int64_t ldf, const double *fa, int64_t ldfa, int64_t na,
                         const double *fb, int64_t ldfb, int64_t nb, double *y, int64_t ldy,
                         void *stream) {
    OSZ_REQUIRE(h && x && f && fa && y, "osz_sosfiltfilt_step: null argument");
    OSZ_REQUIRE(nx >= 1 && ldx >= nx && ldf >= nx, "osz_sosfiltfilt_step: bad forward chunk");
    OSZ_REQUIRE(na >= 1 && ldfa >= na && ldy >= na, "osz_sosfiltfilt_step: bad chunk a");
    OSZ_REQUIRE(!fb || (nb >= 1 && ldfb >= nb), "osz_sosfiltfilt_step: bad chunk b");
    OSZ_SAME_DEVICE(h, "osz_sosfiltfilt_step");
    {
        int rc = spec_touch(h->spec, as_stream(stream));
        if (rc) return rc;
    }
    const int64_t tile = (int64_t)h->NW * 64 * h->T;
    const bool fusable = h->T == 32 && h->NW == 4 && nx % tile == 0 && na % tile == 0 &&
                         h->nch >= 96;   // fewer channels: time-split launches fill the chip better
    if (!fusable) {
        int rc = osz_sos_forward(h, x, ldx, f, ldf, nx, stream);
        if (rc) return rc;
        return osz_sosfiltfilt_chunk(h, fa, ldfa, na, fb, ldfb, nb, y, ldy, stream);
    }
    hipStream_t st = as_stream(stream);
    SosArgs w{}, fw{}, bw{};
    w.sec = fw.sec = bw.sec = h->dsec;
    w.nsec = fw.nsec = bw.nsec = h->nsec;
    w.nch = fw.nch = bw.nch = h->nch;
    w.zi_unit = fw.zi_unit = bw.zi_unit = h->dzi;
    w.tab2 = fw.tab2 = bw.tab2 = h->dtab2;
    w.touch = fw.touch = bw.touch = h->touch;
    // NaN reach of the backward pass: the last sample of what it is initialised from
    w.probe = bw.probe = !sos_nanfix() ? nullptr : fb ? fb + (nb - 1) : fa + (na - 1);
    w.ldprobe = bw.ldprobe = fb ? ldfb : ldfa;
    // The chunk-local warm-up (numerical.py:397-399) rides the dual launch as a
    // pre-roll of the backward pass's first segment when it is exactly whole
    // tiles of the trimmed body; otherwise it is a small launch of its own.
    const bool dual2 = sos_lean() && h->dtab2 && !sos_pf() && h->warm_len == tile &&
                       (nx / tile) >= 8 && (na / tile) >= 8;
    if (fb && dual2 && nb >= h->warm_len) {
        bw.prex = fb;
        bw.ldprex = ldfb;
        bw.npre = h->warm_len;
    } else if (fb) {  // warm-up over the head of the next forward chunk: state only
        w.x = fb;
        w.ldx = ldfb;
        w.n = nb < h->warm_len ? nb : h->warm_len;
        w.state_out = h->dtmp;
        int rc = sos_launch<true>(w, h->dcarry, h->T, h->NW, h->warm_len, st);
        if (rc) return rc;
        bw.state_in = h->dtmp;
    }
    fw.x = x;
    fw.y = f;
    fw.ldx = ldx;
    fw.ldy = ldf;
    fw.n = nx;
    fw.state_in = h->dstate;
    fw.state_out = h->dstate_alt;
    bw.x = fa;
    bw.y = y;
    bw.ldx = ldfa;
    bw.ldy = ldy;
    bw.n = na;
    if (sos_lean() && h->warm_len == tile && (nx / tile) >= 8 && (na / tile) >= 8) {
        // lean bodies: both passes cut into time segments, 3 workgroups per CU
        // at least four segments per pass: several rounds of the 768 resident
        // workgroups even out the tail (measured 96..512 channels; with one
        // segment per pass, >= 384 channels, the launch fell back to the
        // two-workgroups-per-CU body and lost 7 %)
        int nseg = (768 + 2 * h->nch - 1) / (2 * h->nch);
        if (nseg < 4) nseg = 4;
        const int64_t tmin = (nx < na ? nx : na) / tile;
        if (nseg > tmin / 4) nseg = (int)(tmin / 4);
        if (nseg >= 2) {
            const int64_t sf = ((nx / tile + nseg - 1) / nseg) * tile;
            const int64_t sb = ((na / tile + nseg - 1) / nseg) * tile;
            if (h->dtab2) {
                const bool al = sos_rows_aligned16(fw) && sos_rows_aligned16(bw);
                auto k2 = sos_pf() ? (al ? sos_dual2_kernel<32, 4, true, true>
                                         : sos_dual2_kernel<32, 4, false, true>)
                                   : (al ? sos_dual2_kernel<32, 4, true, false>
                                         : sos_dual2_kernel<32, 4, false, false>);
                const size_t lds2 = Sos2Lds::bytes(32, 4, h->nsec);
                OSZ_DYN_LDS(k2, lds2);
                KernelTimer kt("sos_dual", st);
                hipLaunchKernelGGL(k2, dim3(h->nch, 2 * nseg), dim3(256), lds2, st, fw, bw, h->dsec,
                                   h->dtab2, nseg, sf, sb, h->warm_len);
            } else {
                auto kern = sos_dual_lean_kernel<32, 4>;
                const size_t lds = sizeof(double) * ((size_t)4 * 32 * (32 + kSosPad) + 2 * 4 * 2 +
                                                     2 * kSosMaxSec * 2);
                OSZ_DYN_LDS(kern, lds);
                KernelTimer kt("sos_dual", st);
                hipLaunchKernelGGL(kern, dim3(h->nch, 2 * nseg), dim3(256), lds, st, fw, bw,
                                   h->dsec, nseg, sf, sb, h->warm_len);
            }
            OSZ_HIP(hipGetLastError());
            {
                int rcs = sos_seal_launch(fw.y, fw.ldy, fw.n, nseg, 1, 1, sf, fw.state_out, fw.nsec, fw.nch, nullptr,
                                          0, 0, st);
                if (rcs) return rcs;
            }
            std::swap(h->dstate, h->dstate_alt);
            return OSZ_OK;
        }
    }
    if (bw.prex) {   // not the trimmed dual launch after all: the warm-up as its own launch
        w.x = fb;
        w.ldx = ldfb;
        w.n = bw.npre;
        w.state_out = h->dtmp;
        int rcw = sos_launch<true>(w, h->dcarry, h->T, h->NW, h->warm_len, st);
        if (rcw) return rcw;
        bw.state_in = h->dtmp;
        bw.prex = nullptr;
    }
    int rc = sos_launch_dual<32, 4>(fw, bw, st);
    if (rc) return rc;
    std::swap(h->dstate, h->dstate_alt);
    return OSZ_OK;
}

int osz_sosfiltfilt_chunk(osz_sos_t h, const double *fa, int64_t ldfa, int64_t na,
                          const double *fb, int64_t ldfb, int64_t nb, double *y, int64_t ldy,
                          void *stream) {
    OSZ_REQUIRE(h && fa && y, "osz_sosfiltfilt_chunk: null argument");
    OSZ_REQUIRE(na >= 1 && ldfa >= na && ldy >= na, "osz_sosfiltfilt_chunk: bad chunk a");
    OSZ_REQUIRE(!fb || (nb >= 1 && ldfb >= nb), "osz_sosfiltfilt_chunk: bad chunk b");
    OSZ_SAME_DEVICE(h, "osz_sosfiltfilt_chunk");
    return sosfiltfilt_chunk_on(h, fa, ldfa, na, fb, ldfb, nb, y, ldy, h->dtmp, h->dcarry,
                                as_stream(stream));
}

}  // extern "C"
