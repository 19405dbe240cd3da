// chain_zp.hip -- FIR -> sosfiltfilt of a long stream as ONE multiplication per bin.
//
// The reference chains oaconvolve (core/numerical.py:158-298) into sosfiltfilt
// (:338-411): a forward pass of the cascade over the FIR output and, chunk by chunk, a
// backward pass started from what back-filtering the NEXT chunk leaves (:397-403).  When a
// chunk is much longer than the cascade's memory (warm_len samples, sos.hip) that start
// state is, to 1e-18, what back-filtering the whole rest of the stream leaves: the two
// passes are the zero-phase filter |H_iir|^2 applied to the FIR output, everywhere except
// within warm_len samples of the stream's two ends, which the caller finishes with the
// separate kernels (openseize_amd/core/numerical.py, _sosfiltfilt_after_fir).
//
// chain_spec.hip folds the FORWARD cascade into the FIR's spectrum; here the spectrum is
// H_fir |H_iir|^2 and a block rings on both sides of its window: behind its FIR output
// (causal modes, amplitudes mu, as there) and BEFORE its first sample (the backward pass:
// the same modes running towards the past, amplitudes nu).  In the circular 4096-sample
// window the second lot sits at the end of row 15, running down from sample 4095; one
// joint least-squares fit on the first and last nh samples of that row gives both.  Per
// block four bursts of R rows put things where they belong:
//     -mu forwards from the block's first sample      (the wrapped right tail leaves the window)
//     +mu forwards from its first sample + 4096       (and continues behind it)
//     -nu backwards from its first sample + 4095      (the wrapped left tail leaves the window)
//     +nu backwards from its first sample - 1         (and lands in front of the block)
// The last one reaches into rows the PREVIOUS block has produced: the last R rows of
// block b of a pair wait in registers for the next pair's fit; a run starts one pair early
// (as in chain_spec.hip) and stores the last R rows of that pair -- its block b depends on
// nothing before it -- for the run before it; a chunk's last 256 R samples wait in `held`
// for the next chunk: the output stream runs L = 256 R samples late
// (osz_chain_zp_lag).  Nothing else crosses a chunk: `carry` is the same kind of sequence
// as in chain_spec.hip (what the outputs behind the chunk would be if the input stopped).
//
// HBM traffic: 8 B read + 8 B written per channel-sample for the whole chain (the forward
// stream never exists); SURVEY 8d books the chain at 48.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <utility>
#include <vector>

#include "chain_zp.h"

namespace osz {

// The fit and the burst amplitudes of a pair in one stage.  Eight consecutive lanes share one
// amplitude (block, causal / anticausal, mode): each takes an eighth of the 2 nh fit samples
// for the real AND the imaginary row of M, three DPP steps add the parts up in the octet's
// last lane, and that lane writes kappa[r] = amplitude * lambda^(256 r), r < R:
//   kapA[amp][r][q]   amp = 0 mu_a, 1 mu_b, 2 nu_a, 3 nu_b              (this pair)
//   kapN[k][r][q]     k = 0 mu_b, 1 nu_b: what the NEXT pair meets as mu_pb, nu_pb
// (Four lanes per amplitude, two waves at work: the stage is a string of dependent LDS round
// trips that the other waves wait out at the barrier; eight lanes halve the string.)
template <int NM, int PER>
__device__ __forceinline__ void zp_fit_kappa_n(int tt, int R, const double *fitbuf, const double *mtab,
                                               const double *lrow, double *kapA, double *kapN) {
    constexpr int ns = 8 * PER;
    // 32 NM lanes have an amplitude to work on: the waves behind them skip the stage (whole
    // waves: the DPP steps below want their rows complete)
    if ((tt & ~63) >= 32 * NM) return;
    const int qd = tt >> 3, p8 = tt & 7;
    const bool valid = qd < 4 * NM;
    const int blk = valid ? qd / (2 * NM) : 0, kind = valid ? (qd / NM) % 2 : 0, q = valid ? qd % NM : 0;
    const double *yb = fitbuf + ns * blk + PER * p8;
    const double *mr = mtab + ns * ((2 * kind) * NM + q) + PER * p8;
    const double *mi = mtab + ns * ((2 * kind + 1) * NM + q) + PER * p8;
    // a batch of operands first, then independent chains (all of them at once and the data
    // registers spill)
    constexpr int B = PER % 3 == 0 ? 3 : 4;
    double sr0 = 0.0, sr1 = 0.0, si0 = 0.0, si1 = 0.0;
#pragma unroll
    for (int k0 = 0; k0 < PER; k0 += B) {
        double y[B], a[B], b[B];
#pragma unroll
        for (int k = 0; k < B; ++k) {
            y[k] = yb[k0 + k];
            a[k] = mr[k0 + k];
            b[k] = mi[k0 + k];
        }
#pragma unroll
        for (int k = 0; k < B; ++k) {
            if (k & 1) {
                sr1 = fma(a[k], y[k], sr1);
                si1 = fma(b[k], y[k], si1);
            } else {
                sr0 = fma(a[k], y[k], sr0);
                si0 = fma(b[k], y[k], si0);
            }
        }
    }
    double sr = valid ? sr0 + sr1 : 0.0, si = valid ? si0 + si1 : 0.0;
    sr += dpp_row_shr0<1>(sr);
    si += dpp_row_shr0<1>(si);
    sr += dpp_row_shr0<2>(sr);
    si += dpp_row_shr0<2>(si);
    sr += dpp_row_shr0<4>(sr);
    si += dpp_row_shr0<4>(si);
    if (p8 == 7 && valid) {
        const int amp = kind * 2 + blk;
        double lr[kSpecRMax], li[kSpecRMax];
#pragma unroll
        for (int r = 0; r < kSpecRMax; ++r) {
            lr[r] = r < R ? lrow[(r * NM + q) * 2 + 0] : 0.0;
            li[r] = r < R ? lrow[(r * NM + q) * 2 + 1] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < kSpecRMax; ++r) {
            if (r < R) {
                const double kr = sr * lr[r] - si * li[r], ki = sr * li[r] + si * lr[r];
                kapA[((amp * R + r) * NM + q) * 2 + 0] = kr;
                kapA[((amp * R + r) * NM + q) * 2 + 1] = ki;
                if (blk == 1) {
                    kapN[((kind * R + r) * NM + q) * 2 + 0] = kr;
                    kapN[((kind * R + r) * NM + q) * 2 + 1] = ki;
                }
            }
        }
    }
}

// nh is 16, 24 or 32 (spec::build_zp)
template <int NM>
__device__ __forceinline__ void zp_fit_kappa(int tt, int nh, int R, const double *fitbuf, const double *mtab,
                                             const double *lrow, double *kapA, double *kapN) {
    if (nh == 24) zp_fit_kappa_n<NM, 6>(tt, R, fitbuf, mtab, lrow, kapA, kapN);
    else if (nh == 32) zp_fit_kappa_n<NM, 8>(tt, R, fitbuf, mtab, lrow, kapA, kapN);
    else zp_fit_kappa_n<NM, 4>(tt, R, fitbuf, mtab, lrow, kapA, kapN);
}

// Three bursts of one row at once: every operand first, two chains per burst.
template <int NM>
__device__ __forceinline__ void zp_dot3(const double *k0, const double *k1, const double *k2, const double *pr,
                                        const double *pi, double &d0, double &d1, double &d2) {
    // (one mode at a time: six operands in flight; more at once and the data registers spill)
    double ar = 0.0, ai = 0.0, br = 0.0, bi = 0.0, cr = 0.0, ci = 0.0;
#pragma unroll
    for (int q = 0; q < NM; ++q) {
        ar = fma(k0[2 * q], pr[q], ar);
        ai = fma(k0[2 * q + 1], pi[q], ai);
        br = fma(k1[2 * q], pr[q], br);
        bi = fma(k1[2 * q + 1], pi[q], bi);
        cr = fma(k2[2 * q], pr[q], cr);
        ci = fma(k2[2 * q + 1], pi[q], ci);
    }
    d0 = ar - ai;
    d1 = br - bi;
    d2 = cr - ci;
}

// The forward bursts of a pair (RF rows): mu_a leaves row r of block a and arrives in row
// D + r of block b; mu_b leaves row r of block b; mu_pb arrives in row D + r of block a.
template <int NR, int NM, int RF>
__device__ __forceinline__ void zp_fwd_bursts(double *re, double *im, const double *kapA, const double *kpb, int R,
                                              const double *Pr, const double *Pi, double &first_a, double &first_b) {
    constexpr int D = 16 - NR;
#pragma unroll
    for (int r = 0; r < RF; ++r) {
        if (r < D) {
            double ca, cb, cp;
            zp_dot3<NM>(kapA + ((0 * R + r) * NM) * 2, kapA + ((1 * R + r) * NM) * 2, kpb + ((0 * R + r) * NM) * 2,
                        Pr, Pi, ca, cb, cp);
            re[r] -= ca;
            im[(D + r) & 15] += ca;
            im[r] -= cb;
            re[(D + r) & 15] += cp;
            if (r == 0) {
                first_a = ca;
                first_b = cb;
            }
        }
    }
}

// The backward bursts (RB rows): nu_a leaves row D-1-r of block b and arrives in the rows the
// previous block b holds back (c7); nu_b arrives in row NR-1-r of block a; nu_pb leaves row
// D-1-r of block a.
template <int NR, int NM, int RB>
__device__ __forceinline__ void zp_bwd_bursts(double *re, double *im, double *c7, const double *kapA,
                                              const double *kpb, int R, const double *Pr, const double *Pi) {
    constexpr int D = 16 - NR;
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        if (r < D) {
            double na, nb, np;
            zp_dot3<NM>(kapA + ((2 * R + r) * NM) * 2, kapA + ((3 * R + r) * NM) * 2, kpb + ((1 * R + r) * NM) * 2,
                        Pr, Pi, na, nb, np);
            im[(D - 1 - r) & 15] -= na;
            c7[r] = na;
            re[(NR - 1 - r) & 15] += nb;
            re[(D - 1 - r) & 15] -= np;
        }
    }
}

// In-kernel phase stamps for the diagnostic build only (benchmarks/zp_stamps.hip defines
// OSZ_FIR_STAMPS); the library build has none.  Slots 0-11: FirPair's; 12-15: this kernel's.
#ifdef OSZ_FIR_STAMPS
#define OSZ_ZSTAMP(slot)                                                             \
    do {                                                                             \
        unsigned long long now_;                                                     \
        __builtin_amdgcn_sched_barrier(0);                                           \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                           \
        zst[slot] += now_ - P.stamp_last;                                            \
        P.stamp_last = now_;                                                         \
    } while (0)
#else
#define OSZ_ZSTAMP(slot) do { } while (0)
#endif

// Absolute time marks of a workgroup (diagnostic build only: benchmarks/zp_timeline.hip defines
// OSZ_ZP_MARKS): entry, tables in LDS, whole pairs done, chunk closed, exit.
#ifdef OSZ_ZP_MARKS
__device__ unsigned long long *g_zp_marks = nullptr;   // [nch][nruns][8]
#define OSZ_ZMARK(k)                                                                            \
    do {                                                                                        \
        if (g_zp_marks && threadIdx.x == 0) {                                                   \
            unsigned long long now_;                                                            \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");        \
            g_zp_marks[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (k)] = now_;        \
        }                                                                                       \
    } while (0)
#else
#define OSZ_ZMARK(k) do { } while (0)
#endif

template <int NR, int NM, bool DMA = false>
__global__ __launch_bounds__(256, 2) void chain_zp_kernel(ZpArgs g) {
    OSZ_ZMARK(0);
#ifdef OSZ_ZP_MARKS
    if (g_zp_marks && threadIdx.x == 0) {
        unsigned long long rt;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt)::"memory");
        g_zp_marks[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 6] = rt;
    }
#endif
    constexpr int D = 16 - NR, S = 256 * NR;
    extern __shared__ fft::cube::C2 cube_lds[];
    const int R = g.R, Rf = g.Rf, nh = g.nh, ns = 2 * nh;
    double *xl = reinterpret_cast<double *>(cube_lds) + 2 * fft::cube::SLOTS;   // behind the cube
    double *fitbuf = xl;                               // [2 blk][2 nh]
    double *kapA = fitbuf + 2 * ns;                    // [4 amp][R][NM][2]: this pair's mu_a mu_b nu_a nu_b
    double *kapP = kapA + 4 * R * NM * 2;              // [2 parity][2][R][NM][2]: mu_pb, nu_pb
    double *lrow = kapP + 4 * R * NM * 2;              // [R][NM][2]
    double *ptab = lrow + R * NM * 2;                  // [20][NM][2]
    double *mtab = ptab + 20 * NM * 2;                 // [4 NM][2 nh]
    const FirArgs &a = g.f;
    const int t = threadIdx.x, run = blockIdx.x, c = blockIdx.y;
    const int L = 256 * R;
    const double *xr = a.x + (int64_t)c * a.ldx;
    // output q of this chunk: q < n0 -> y0r[q], else yr[q] (yr: the second buffer, shifted by n0)
    const int64_t n = g.n, n0 = g.n0;
    double *y0r = g.y0 ? g.y0 + (int64_t)c * g.ldy0 : nullptr;
    double *yr = a.y + (int64_t)c * a.ldy - n0;
    double *ho = g.held_out + (int64_t)c * L;
    // (the partition's 64-bit divisions run on the vector unit: their results, uniform, go back to
    // scalar registers -- a chunk has far fewer than 2^31 pairs)
    const int p0 = __builtin_amdgcn_readfirstlane((int)zp_run_start(run, g.W, g.nruns, g.wclose));
    const int p1 = __builtin_amdgcn_readfirstlane((int)zp_run_start(run + 1, g.W, g.nruns, g.wclose));
    const int first = run == 0 ? 0 : p0 - 1;
    const int lastf = p1 - 1;

    FirPair<NR, 16, 0, true> P{a, t, a.wlen - 1, xr, yr, 0, cube_lds};
    fft::cube2::tw_load(t, a.tb, P.tw1, P.tw2);
#pragma unroll
    for (int j = 0; j < D; ++j) P.cr[j] = 0.0;
    if (DMA && first <= lastf) zp_request_rows<NR>(xr + (int64_t)first * (2 * S), 2 * NR, t, cube_lds);
    {
        // lrow | ptab | mtab are one table on the device too (g.Lrow): every request of the
        // sweep is out before the first answer is needed
        const int ntab = R * NM * 2 + 20 * NM * 2 + 4 * NM * ns;
#pragma unroll 8
        for (int i = t; i < ntab; i += 256) lrow[i] = g.Lrow[i];
    }
    for (int i = t; i < 4 * R * NM * 2; i += 256) kapP[i] = 0.0;
    double held[kSpecRMax];          // rows NR-1-r of the previous pair's block b, one burst short
#pragma unroll
    for (int r = 0; r < kSpecRMax; ++r) held[r] = 0.0;
    // `bad` is uniform (a scalar), and sticky: the stream went bad in an earlier chunk, or a pair
    // of this run held non-finite samples -- behind the transform they are everywhere, every
    // amplitude of the fit and with it every lane's burst values are non-finite
    bool bad = g.nanpos[c] != 0x7fffffffffffffffLL;
    int64_t bad_at = 0;              // chunk position of the pair that went bad
    int par = 0;
    int younger = -1;                // DMA kernels: vector-memory operations behind the pending requests
    const unsigned lane8_entry = 8u * (unsigned)t;   // a lane's byte offset inside a row of 256 samples
    __syncthreads();
    OSZ_ZMARK(1);

    // a sample of the chunk (position i, value v) goes to the output, L samples late, or,
    // the chunk's last L samples, to `held`
#define OSZ_ZP_PUT(i_, v_)                      \
    do {                                        \
        const int64_t q_ = (i_) + L;            \
        if (q_ < n0) y0r[q_] = (v_);            \
        else if (q_ < n) yr[q_] = (v_);         \
        else ho[q_ - n] = (v_);                 \
    } while (0)

#ifdef OSZ_FIR_STAMPS
    unsigned long long zst[16];
    for (int q = 0; q < 16; ++q) zst[q] = 0;
    for (int q = 0; q < 12; ++q) P.stamp_acc[q] = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(P.stamp_last)::"memory");
#endif
    for (int p = first; p <= lastf; ++p) {
        const int64_t o = (int64_t)p * (2 * S);
        double re[16], im[16];
        take_turns();
        if (DMA) {
            // the pair's samples were requested behind the previous pair's transform: younger
            // than they are only that pair's stores (requests and stores retire in order on
            // one counter), `younger` of them when every one went through the row stores
            if (younger == 2 * NR) asm volatile("s_waitcnt vmcnt(%0) ; osz:dma" ::"n"(2 * NR) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) ; osz:dma" ::: "memory");
            int tq = t;
            asm volatile("" : "+v"(tq));
            const double *xs = reinterpret_cast<const double *>(cube_lds) + 128 * (tq >> 6) + (tq & 63);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                re[j] = j < NR ? xs[512 * (j >> 1) + 64 * (j & 1)] : 0.0;
                im[j] = j < NR ? xs[512 * ((j + NR) >> 1) + 64 * ((j + NR) & 1)] : 0.0;
            }
        } else {
            // rows of the pair: base and row offsets in scalar registers (buf_rsrc, common.h)
            const __amdgpu_buffer_rsrc_t rx = buf_rsrc(xr + o);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                re[j] = j < NR ? buf_load(rx, lane8_entry, 2048u * j) : 0.0;
                im[j] = j < NR ? buf_load(rx, lane8_entry, 2048u * (j + NR)) : 0.0;
            }
        }
        OSZ_ZSTAMP(0);    // previous pair's stores + this pair's loads issued
        P.transform(re, im);
        if (DMA && p < lastf) {
            // inverse pass 1 has read this wave's pieces of the cube (its values are in use
            // below): the next pair's samples can land there while fit, bursts and stores run
            asm volatile("s_waitcnt lgkmcnt(0) ; osz:dma" ::: "memory");
            zp_request_rows<NR>(xr + o + 2 * S, 2 * NR, t, cube_lds);
        }
        int nst = 0;             // row stores of this pair, -1: some went another way
        OSZ_ZSTAMP(11);   // inverse pass 1
        // thread -> role indices, recomputed per pair from an opaque copy of t
        int tt = t;
        asm volatile("" : "+v"(tt));
        const unsigned lane8 = DMA ? 8u * (unsigned)tt : lane8_entry;   // (DMA: not kept over the transform)
        if (tt < nh) {
            fitbuf[tt] = re[15];
            fitbuf[ns + tt] = im[15];
        } else if (tt >= 256 - nh) {
            fitbuf[tt - 256 + ns] = re[15];
            fitbuf[ns + tt - 256 + ns] = im[15];
        }
#pragma unroll
        for (int j = 0; j < D; ++j) {
            re[j] += P.cr[j];
            im[j] += re[j + NR];
            P.cr[j] = im[j + NR];
        }
        OSZ_ZSTAMP(12);   // fit samples to LDS, overlap add
        __syncthreads();
        OSZ_ZSTAMP(13);   // barrier
        // fit and amplitudes: this pair's into kapA, block b's also into the other half of
        // kapP for the next pair
        zp_fit_kappa<NM>(tt, nh, R, fitbuf, mtab, lrow, kapA, kapP + (par ^ 1) * (2 * R * NM * 2));
        __syncthreads();
        OSZ_ZSTAMP(14);   // fit + amplitudes + barrier
        const double *kap = kapA, *kpb = kapP + par * (2 * R * NM * 2);
        // Six burst evaluations per row index r serve the eight places a burst lands: the
        // wrapped right tail of block a leaves its row r and arrives, one window on, in row
        // D + r of block b with the same values (mu_a), and so do the left tail of block a
        // in row D-1-r of block b and in the rows the previous block b holds back (nu_a).
        double c7[kSpecRMax];
#pragma unroll
        for (int r = 0; r < kSpecRMax; ++r) c7[r] = 0.0;
        {
            // forward bursts with lambda^t, then (the same registers) backward ones with
            // lambda^(255 - t): both sets of powers at once do not fit beside the data.  The
            // row counts are compile-time inside (a switch): every operand of a row is
            // requested before the first product, three bursts run as six chains.
            double Pr[NM], Pi[NM];
            zp_powers<NM>(ptab, tt, Pr, Pi);
            double ca = 0.0, cb = 0.0;
            switch (Rf) {
                case 1: zp_fwd_bursts<NR, NM, 1>(re, im, kap, kpb, R, Pr, Pi, ca, cb); break;
                case 2: zp_fwd_bursts<NR, NM, 2>(re, im, kap, kpb, R, Pr, Pi, ca, cb); break;
                case 3: zp_fwd_bursts<NR, NM, 3>(re, im, kap, kpb, R, Pr, Pi, ca, cb); break;
                case 4: zp_fwd_bursts<NR, NM, 4>(re, im, kap, kpb, R, Pr, Pi, ca, cb); break;
                default: zp_fwd_bursts<NR, NM, 5>(re, im, kap, kpb, R, Pr, Pi, ca, cb); break;
            }
            if (!bad && __builtin_amdgcn_readfirstlane((int)(sos_not_finite(ca) || sos_not_finite(cb)))) {
                bad = true;
                bad_at = o;
                if (zp_exact_nanpos(xr + o, 2 * S, tt, reinterpret_cast<long long *>(g.nanpos + c), g.pos + o)) bad_at = -1;
            }
            __builtin_amdgcn_sched_barrier(0);
            zp_powers<NM>(ptab, 255 - tt, Pr, Pi);
            switch (R) {
                case 1: zp_bwd_bursts<NR, NM, 1>(re, im, c7, kap, kpb, R, Pr, Pi); break;
                case 2: zp_bwd_bursts<NR, NM, 2>(re, im, c7, kap, kpb, R, Pr, Pi); break;
                case 3: zp_bwd_bursts<NR, NM, 3>(re, im, c7, kap, kpb, R, Pr, Pi); break;
                case 4: zp_bwd_bursts<NR, NM, 4>(re, im, c7, kap, kpb, R, Pr, Pi); break;
                default: zp_bwd_bursts<NR, NM, 5>(re, im, c7, kap, kpb, R, Pr, Pi); break;
            }
        }
        OSZ_ZSTAMP(15);   // bursts
        const double qn = spec_qnan();
        const bool own = p >= p0 && p < p1;
        const bool edge = p == g.W - 1;        // its samples may be among the chunk's last L
        if (run == 0 && p == 0) {
            // the chunk opens: what the stream so far still owes these samples, and the
            // previous chunk's last L samples, complete with this block's +nu
            nst = -1;
            const double *ci = g.carry_in + (int64_t)c * kSpecLdc + tt;
            if (!bad &&
                __builtin_amdgcn_readfirstlane((int)sos_not_finite(g.carry_in[(int64_t)c * kSpecLdc + 4095 + 256 * R]))) {
                bad = true;
                bad_at = 0;
            }
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                re[j] += ci[256 * j];
                im[j] += ci[S + 256 * j];
            }
            const double *hi = g.held_in + (int64_t)c * L + tt;
#pragma unroll
            for (int r = 0; r < kSpecRMax; ++r)
                if (r < R) {
                    const int64_t q = 256 * (R - 1 - r) + tt;
                    (q < n0 ? y0r : yr)[q] = bad ? qn : hi[256 * (R - 1 - r)] + c7[r];
                }
        } else if (p > first) {
            // the previous pair's last R rows of block b, complete now
            // (also those of the pair a run starts early with: that pair's block b depends on
            // nothing before it, and the run before this one leaves its last R rows to us)
            if (!bad && o - S + L >= n0) {
                const __amdgpu_buffer_rsrc_t ry = buf_rsrc(yr + (o - S + L));
#pragma unroll
                for (int r = 0; r < kSpecRMax; ++r)
                    if (r < R) buf_store(held[r] + c7[r], ry, lane8, 2048u * (NR - 1 - r));
                if (nst >= 0) nst += R;
            } else {
                nst = -1;
                const int64_t ob = o - S + tt;
#pragma unroll
                for (int r = 0; r < kSpecRMax; ++r)
                    if (r < R) OSZ_ZP_PUT(ob + 256 * (NR - 1 - r), bad ? qn : held[r] + c7[r]);
            }
        }
        if (own) {
            if (!bad && !edge && o + L >= n0) {
                // the common case: whole rows into the current output, nothing to decide per sample
                const __amdgpu_buffer_rsrc_t ry = buf_rsrc(yr + (o + L));
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    buf_store(re[j], ry, lane8, 2048u * j);
                    if (j < NR - R) buf_store(im[j], ry, lane8, 2048u * (j + NR));
                }
                if (nst >= 0) nst += 2 * NR - R;
            } else {
                nst = -1;
                int64_t off = o + tt;
                asm volatile("" : "+v"(off));
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    OSZ_ZP_PUT(off + 256 * j, bad ? qn : re[j]);
                    if (j < NR - R) OSZ_ZP_PUT(off + 256 * (j + NR), bad ? qn : im[j]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < kSpecRMax; ++r)
            if (r < R) held[r] = im[(NR - 1 - r) & 15];
        par ^= 1;
        younger = nst;
    }

#ifdef OSZ_FIR_STAMPS
    if (g_fir_stamps && (t & 63) == 0) {
        unsigned long long *so = g_fir_stamps + (((int64_t)c * g.nruns + run) * 4 + (t >> 6)) * 16;
        for (int q = 1; q <= 10; ++q) so[q] = P.stamp_acc[q];
        so[0] = zst[0];
        so[11] = zst[11];
        for (int q = 12; q < 16; ++q) so[q] = zst[q];
    }
#endif
    OSZ_ZMARK(2);
    if (run == g.nruns - 1) {
        // ---- the closing pair: blocks of la and lb samples (lb > 0 only behind a whole
        // block a), accumulated in LDS over the idle cube: acc[i], i = samples from its start
        double *acc = reinterpret_cast<double *>(cube_lds);     // 8192 doubles
        const int64_t o = g.W * (2 * S);
        const int la = g.la, lb = g.lb;
        double re[16], im[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int pp = 256 * j + t;
            re[j] = pp < la ? xr[o + pp] : 0.0;
            im[j] = pp < lb ? xr[o + la + pp] : 0.0;
        }
        P.transform(re, im);
        int tt = t;
        asm volatile("" : "+v"(tt));
        if (tt < nh) {
            fitbuf[tt] = re[15];
            fitbuf[ns + tt] = im[15];
        } else if (tt >= 256 - nh) {
            fitbuf[tt - 256 + ns] = re[15];
            fitbuf[ns + tt - 256 + ns] = im[15];
        }
        __syncthreads();     // also: every thread is done reading the cube
        zp_fit_kappa<NM>(tt, nh, R, fitbuf, mtab, lrow, kapA, kapP + (par ^ 1) * (2 * R * NM * 2));
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            acc[256 * j + t] = re[j] + (j < D ? P.cr[j < D ? j : 0] : 0.0);
            acc[4096 + 256 * j + t] = 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[la + 256 * j + t] += im[j];
        {
            // non-finite amplitudes poison the rest of the stream
            bool nf = false;
            for (int i = 0; i < 4 * NM * 2; ++i) nf = nf || sos_not_finite(kapA[((i / (NM * 2)) * R) * NM * 2 + i % (NM * 2)]);
            if (!bad && __builtin_amdgcn_readfirstlane((int)nf)) {
                bad = true;
                bad_at = o;
                if (zp_exact_nanpos(xr + o, la + lb, t, reinterpret_cast<long long *>(g.nanpos + c), g.pos + o)) bad_at = -1;
            }
        }
        double Pfr[NM], Pfi[NM], Pbr[NM], Pbi[NM];
        zp_powers<NM>(ptab, tt, Pfr, Pfi);
        zp_powers<NM>(ptab, 255 - tt, Pbr, Pbi);
        // ten bursts: (amplitude table, sign, forwards?, first sample of a forward burst /
        // last sample of a backward one); the previous block b's from kapP, block b's own
        // "next" entries from the other half of kapP
        const double *kpb = kapP + par * (2 * R * NM * 2), *kpn = kapP + (par ^ 1) * (2 * R * NM * 2);
        __syncthreads();
        for (int s = 0; s < 10; ++s) {
            const double *tab = s == 0 ? kpb : s == 1 ? kapA : s == 2 ? kpn : s == 3 ? kapA : s == 4 ? kpn
                              : s == 5 ? kpb + R * NM * 2 : s == 6 ? kapA + 2 * R * NM * 2 : s == 7 ? kapA + 2 * R * NM * 2
                              : kpn + R * NM * 2;
            //              0 +mu_pb  1 -mu_a  2 -mu_b  3 +mu_a  4 +mu_b | 5 -nu_pb  6 -nu_a  7 +nu_a (held)  8 -nu_b  9 +nu_b
            const double sg = (s == 1 || s == 2 || s == 5 || s == 6 || s == 8) ? -1.0 : 1.0;
            const bool fwd = s < 5;
            const int off = s == 0 ? 256 * D : s == 1 ? 0 : s == 2 ? la : s == 3 ? 4096 : s == 4 ? la + 4096
                          : s == 5 ? 256 * D - 1 : s == 6 ? 4095 : s == 7 ? -1 : s == 8 ? la + 4095 : la - 1;
            if (s == 7) {
                // the previous pair's last R rows of block b: complete with this block's +nu
                const int64_t ob = o - S + tt;
#pragma unroll
                for (int r = 0; r < kSpecRMax; ++r)
                    if (r < R)
                        OSZ_ZP_PUT(ob + 256 * (NR - 1 - r),
                                   bad ? spec_qnan() : held[r] + zp_dot<NM>(tab + (r * NM) * 2, Pbr, Pbi));
                continue;
            }
            for (int r = 0; r < (fwd ? Rf : R); ++r) {
                const double cs = sg * (fwd ? zp_dot<NM>(tab + (r * NM) * 2, Pfr, Pfi)
                                            : zp_dot<NM>(tab + (r * NM) * 2, Pbr, Pbi));
                const int i = fwd ? off + 256 * r + tt : off - 256 * r - (255 - tt);
                if (i >= 0 && i < 8192) acc[i] += cs;
            }
            __syncthreads();
        }
        const double qn = spec_qnan();
        const int ltot = la + lb;
        for (int i = t; i < ltot; i += 256) OSZ_ZP_PUT(o + i, bad ? qn : acc[i]);
        double *co = g.carry_out + (int64_t)c * kSpecLdc;
        for (int i = t; i < kSpecLdc; i += 256) {
            const int src = ltot + i;
            co[i] = bad ? qn : (src < 8192 ? acc[src] : 0.0);
        }
        if (g.hist) {
            double *hr = g.hist + (int64_t)c * g.hist_len;
            const double *src = xr + n - g.hist_len;
            for (int i = t; i < g.hist_len; i += 256) hr[i] = src[i];
        }
    }
#undef OSZ_ZP_PUT
    OSZ_ZMARK(3);
    // where the forward stream of this channel first went bad: every later launch starts bad
    // (above), and osz_chain_zp_seal makes the chunks the reference loses NaN as a whole --
    // this chunk (also the runs behind this one, which do not see it) and the one before.
    // (No sealing of the later runs in here: overwriting what workgroups on other XCDs have
    // written wants an agent-scope release from each of them, a write-back of the XCD's L2 --
    // 12 us per workgroup on average, 50 at the worst, benchmarks/zp_timeline.hip.)
    if (bad && bad_at >= 0 && t == 0) atomicMin(reinterpret_cast<long long *>(g.nanpos + c), g.pos + bad_at);
    OSZ_ZMARK(4);
#ifdef OSZ_ZP_MARKS
    if (g_zp_marks && threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
        g_zp_marks[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 5] = ((unsigned long long)xcc << 32) | hw;
        unsigned long long rt;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt)::"memory");
        g_zp_marks[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 7] = rt;
    }
#endif
}

// NaN reach of sosfiltfilt (sos_tile.h): a chunk of the reference's output is NaN as a
// whole when the forward stream is NaN anywhere in it or in the chunk after it.  y holds
// output samples [s0, s0 + n) of channel c; chunks are cs samples from `origin` on.
// step > 0: the reference's FIR in front of the cascade works in segments of `step` input
// samples (core/numerical.py:202-217, :258-283: one FFT per segment), and a non-finite sample
// makes the whole segment's output non-finite -- the forward stream is bad from the START of the
// segment that holds the sample (osz_chain_zp_reach).
__global__ void zp_seal_kernel(double *y, int64_t ldy, int64_t n, long long s0, long long origin, long long cs,
                               const long long *nanpos, long long step) {
    const int c = blockIdx.y;
    long long np = nanpos[c];
    if (np == 0x7fffffffffffffffLL) return;
    if (step > 0) np = (np / step) * step;
    long long k = (np - origin) / cs;
    if (np < origin) k = 0;
    const long long from = origin + (k - 1) * cs;       // the chunk before the one that holds the NaN
    double *yr = y + (int64_t)c * ldy;
    const double qn = spec_qnan();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (s0 + i >= from) yr[i] = qn;
}

__global__ void zp_poison_kernel(const long long *nanpos, double *state, int nsec, int nch) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nch || nanpos[c] == 0x7fffffffffffffffLL) return;
    const double qn = spec_qnan();
    for (int q = 0; q < nsec; ++q) {
        state[((int64_t)q * nch + c) * 2 + 0] = qn;
        state[((int64_t)q * nch + c) * 2 + 1] = qn;
    }
}

__global__ void zp_fill_ll_kernel(long long *p, int n, long long v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---------------------------------------------------------------- host side
struct ChainZp {
    osz_fir_s *fir = nullptr;
    osz_sos_s *sos = nullptr;
    bool eligible = false, open = false;
    bool nega = false;             // one real block per transform (chain_zpn.hip); NR is then its rows per block
    int NR = 0, NM = 0, R = 0, Rf = 0, nh = 0;
    int NS = 0;                    // nega: the slow modes (spec::build_zpn)
    double *dH = nullptr, *dT = nullptr;   // dT: burst rows | mode powers | fit matrix, as the kernel's LDS holds them
    double *dM = nullptr, *dP = nullptr, *dL = nullptr;   // (views into dT)
    double *dcarry[2] = {nullptr, nullptr}, *dheld[2] = {nullptr, nullptr};
    int cur = 0;
    int64_t pos = 0;               // stream position of the next chunk's first sample
    long long *dnanpos = nullptr;
    int hist_cap = 0;              // wlen - 1 + warm_len
    double *dhist[2] = {nullptr, nullptr};
    int hcur = 0;
    int64_t hist_n = 0;
    double *dscratch = nullptr;
    double *dfin = nullptr;        // osz_chain_zp_finish: the kernel's outputs for the head of what follows
    size_t fin_cap = 0;            // doubles
    hipEvent_t fin_done = nullptr; // behind the last copy out of dfin: a later finish may come on another stream
    double *dzero = nullptr;       // (nsec, nch, 2) zeros: start state of the opening's backward pass
    int64_t ref_step = 0;          // osz_chain_zp_reach: the reference FIR's segment length (0: none)
};

// OSZ_ZP_NEGA=0: the pair kernel of this file instead of chain_zpn.hip's (one real block per
// transform), for comparison
static bool zp_nega() {
    static const bool on = [] {
        const char *e = getenv("OSZ_ZP_NEGA");
        return !(e && e[0] == '0');
    }();
    return on;
}
// the shortest chunk a step takes: two pairs of blocks / two blocks
static int64_t zp_min_chunk(const ChainZp *s) { return (s->nega ? 2 : 4) * 256 * (int64_t)s->NR; }

static void zp_free(ChainZp *s) {
    (void)hipFree(s->dH);
    (void)hipFree(s->dT);
    for (int q = 0; q < 2; ++q) {
        (void)hipFree(s->dcarry[q]);
        (void)hipFree(s->dheld[q]);
        (void)hipFree(s->dhist[q]);
    }
    (void)hipFree(s->dnanpos);
    (void)hipFree(s->dscratch);
    (void)hipFree(s->dfin);
    if (s->fin_done) (void)hipEventDestroy(s->fin_done);
    (void)hipFree(s->dzero);
    delete s;
}

void zp_unlink(ChainZp *s) {
    if (!s) return;
    if (s->fir) s->fir->zp = nullptr;
    if (s->sos) s->sos->zp = nullptr;
    zp_free(s);
}

static size_t zp_lds_bytes(const ChainZp *s) {
    const int NM = s->NM, R = s->R, ns = 2 * s->nh;
    if (s->nega)
        return sizeof(fft::cube::C2) * fft::cube::SLOTS +
               sizeof(double) * (ns + R * (2 * s->NS + NM) * 2 + NM * 2 + 20 * NM * 2 + (2 * s->NS + 2 * NM) * ns) +
               1024;   // + W256 rows
    return sizeof(fft::cube::C2) * fft::cube::SLOTS +
           sizeof(double) * (2 * ns + 8 * R * NM * 2 + R * NM * 2 + 20 * NM * 2 + 4 * NM * ns);
}

static int zp_get(osz_fir_s *fir, osz_sos_s *sos, ChainZp **out) {
    ChainZp *s = sos->zp;
    if (s && s->fir != fir) {
        zp_unlink(s);
        s = nullptr;
    }
    if (!s && fir->zp) zp_unlink(fir->zp);
    if (!s) {
        s = new ChainZp();
        s->fir = fir;
        s->sos = sos;
        fir->zp = sos->zp = s;
        if (fir->parts.size() == 1 && fir->nch == sos->nch) {
            spec::TablesZp T;
            const double tol = sos->zp_tol > 0.0 ? sos->zp_tol : (double)spec::kTailTol;
            if (zp_nega()) {
                const bool forgets = sos->warm_len <= (1 << 20);
                T = spec::kept_tables(spec::kKeptZpn, fir->htaps, sos->coef, sos->nsec, tol, forgets, [&] {
                    return spec::build_zpn(fir->htaps.data(), fir->ntaps, sos->coef, sos->nsec, forgets, 15360 - 1024,
                                           (spec::ld_t)tol);
                });
                s->nega = T.eligible;
            }
            if (!T.eligible)
                T = spec::build_zp(fir->htaps.data(), fir->ntaps, sos->coef, sos->nsec, sos->warm_len <= (1 << 20),
                                   15360, (spec::ld_t)tol);
            if (T.eligible) {
                auto up = [](double **d, const std::vector<double> &v) -> int {
                    OSZ_HIP(hipMalloc(d, v.size() * sizeof(double)));
                    OSZ_HIP(hipMemcpy(*d, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice));
                    return OSZ_OK;
                };
                int rc;
                // one table, in the order of the kernel's LDS: a workgroup fetches it in one sweep
                // (the one-block kernel keeps lambda^256 alone -- row 1 -- and forms the burst rows by products)
                const size_t nl = s->nega ? (size_t)T.NM * 2 : (size_t)T.R * T.NM * 2, np = (size_t)20 * T.NM * 2,
                             nm = (size_t)(s->nega ? 2 * T.NS + 2 * T.NM : 4 * T.NM) * 2 * T.nh;
                const size_t l0 = s->nega ? (size_t)T.NM * 2 : 0;
                if (T.L.size() < l0 + nl || T.P.size() != np || T.M.size() != nm)
                    return fail(OSZ_ERR_STATE, "zero-phase tables: %zu %zu %zu", T.L.size(), T.P.size(), T.M.size());
                std::vector<double> cat(T.L.begin() + l0, T.L.begin() + l0 + nl);
                cat.insert(cat.end(), T.P.begin(), T.P.end());
                cat.insert(cat.end(), T.M.begin(), T.M.end());
                if ((rc = up(&s->dH, spec_permuted_spectrum(T.H))) || (rc = up(&s->dT, cat))) return rc;
                s->dL = s->dT;
                s->dP = s->dL + nl;
                s->dM = s->dP + np;
                const int nch = fir->nch;
                for (int q = 0; q < 2; ++q) {
                    OSZ_HIP(hipMalloc(&s->dcarry[q], sizeof(double) * (size_t)nch * kSpecLdc));
                    OSZ_HIP(hipMalloc(&s->dheld[q], sizeof(double) * (size_t)nch * 256 * T.R));
                }
                s->hist_cap = (fir->ntaps - 1) + (int)sos->warm_len;
                for (int q = 0; q < 2; ++q)
                    OSZ_HIP(hipMalloc(&s->dhist[q], sizeof(double) * (size_t)nch * s->hist_cap));
                OSZ_HIP(hipMalloc(&s->dnanpos, sizeof(long long) * (size_t)nch));
                const size_t sb = sizeof(double) * (size_t)sos->nsec * nch * 2;
                OSZ_HIP(hipMalloc(&s->dzero, sb));
                OSZ_HIP(hipMemset(s->dzero, 0, sb));
                s->NR = T.NR;
                s->NM = T.NM;
                s->R = T.R;
                s->Rf = T.Rf;
                s->nh = T.nh;
                s->NS = T.NS;
                s->eligible = true;
            }
        }
    }
    *out = s;
    return OSZ_OK;
}

template <int NM, bool DMA>
static zp_kern_t zp_kernel_for(int nr) {
    static const zp_kern_t k[8] = {chain_zp_kernel<8, NM, DMA>,  chain_zp_kernel<9, NM, DMA>,
                                   chain_zp_kernel<10, NM, DMA>, chain_zp_kernel<11, NM, DMA>,
                                   chain_zp_kernel<12, NM, DMA>, chain_zp_kernel<13, NM, DMA>,
                                   chain_zp_kernel<14, NM, DMA>, chain_zp_kernel<15, NM, DMA>};
    return k[nr - 8];
}
// one chunk through the kernel; hist: keep the input a later osz_chain_zp_finish replays
static int zp_launch(ChainZp *s, const double *x, int64_t ldx, int64_t n, double *y0, int64_t ldy0,
                     int64_t n0, double *y, int64_t ldy, hipStream_t st) {
    osz_fir_s *fir = s->fir;
    const int NR = s->NR, S = 256 * NR;
    const int64_t pair = 2 * (int64_t)S;
    const int64_t npw = n / pair, rem = n - npw * pair;
    // whole pairs (blocks) on the fast path; the closing pair has 1 .. 2 S samples, the new
    // kernel's closing block 1 .. S
    const int64_t W = s->nega ? (n - 1) / S : rem == 0 ? npw - 1 : npw;
    const int64_t nlast = s->nega ? n - W * S : n - W * pair;
    // one round of resident workgroups (two per CU); a run has a pair of its own
    // (256, 384, 768 and 1024 workgroups measured slower at 16 - 64 channels, profiles/README.md)
    int64_t nruns = 512 / fir->nch;
    if (nruns > W) nruns = W;
    if (nruns < 1) nruns = 1;
    ZpArgs g{};
    g.f.x = x;
    g.f.y = y;
    g.f.ldx = ldx;
    g.f.ldy = ldy;
    g.f.n = n;
    g.f.skip = 0;
    g.f.wlen = fir->ntaps;
    g.f.step = S;
    g.f.H = s->dH;
    g.f.tb = fir->tb;
    g.y0 = y0;
    g.ldy0 = ldy0;
    g.n0 = n0;
    g.n = n;
    g.W = W;
    g.nruns = (int)nruns;
    g.la = (int)std::min<int64_t>(nlast, S);
    g.lb = (int)(nlast - g.la);
    g.R = s->R;
    g.Rf = s->Rf;
    g.nh = s->nh;
    g.M = s->dM;
    g.P = s->dP;
    g.Lrow = s->dL;
    g.carry_in = s->dcarry[s->cur];
    g.carry_out = s->dcarry[s->cur ^ 1];
    g.held_in = s->dheld[s->cur];
    g.held_out = s->dheld[s->cur ^ 1];
    g.nanpos = s->dnanpos;
    g.pos = s->pos;
    g.wclose = s->nega ? 1 : 4;   // pairs: measured, 2 ... 12 at 32 and 256 channels (profiles/README.md); a closing block is one more block
    if (n >= s->hist_cap) {
        g.hist = s->dhist[s->hcur];
        g.hist_len = s->hist_cap;
        s->hist_n = s->hist_cap;
    } else {
        const int64_t keep = std::min<int64_t>(s->hist_n, s->hist_cap - n);
        double *dst = s->dhist[s->hcur ^ 1];
        if (keep > 0)
            OSZ_HIP(hipMemcpy2DAsync(dst, sizeof(double) * s->hist_cap, s->dhist[s->hcur] + (s->hist_n - keep),
                                     sizeof(double) * s->hist_cap, sizeof(double) * keep, fir->nch,
                                     hipMemcpyDeviceToDevice, st));
        OSZ_HIP(hipMemcpy2DAsync(dst + keep, sizeof(double) * s->hist_cap, x, sizeof(double) * ldx,
                                 sizeof(double) * n, fir->nch, hipMemcpyDeviceToDevice, st));
        s->hcur ^= 1;
        s->hist_n = keep + n;
        g.hist = nullptr;
        g.hist_len = 0;
    }
    zp_kern_t kern = s->nega    ? zpn_kernel_for(NR, s->NM, s->NS, s->R)
                     : s->NM == 2 ? zp_kernel_for<2, true>(NR)
                     : s->NM == 4 ? zp_kernel_for<4, true>(NR)
                                  : zp_kernel_for<6, true>(NR);     // (rows by LDS-DMA, as the one-block kernel's)
    if (!kern) return fail(OSZ_ERR_STATE, "zero-phase kernel: no instance for %d rows, %d modes (%d slow)", NR, s->NM, s->NS);
    const size_t lds = zp_lds_bytes(s);
    OSZ_DYN_LDS(kern, lds);
    {
        KernelTimer kt("chain_zp", st);
        hipLaunchKernelGGL(kern, dim3((unsigned)nruns, fir->nch), dim3(256), lds, st, g);
    }
    OSZ_HIP(hipGetLastError());
    s->cur ^= 1;
    s->pos += n;
    return OSZ_OK;
}

}  // namespace osz

using namespace osz;

extern "C" {

int64_t osz_chain_zp_lag(osz_fir_t fir, osz_sos_t sos) {
    if (!fir || !sos) return -1;
    ChainZp *s = nullptr;
    if (zp_get(fir, sos, &s) != OSZ_OK || !s->eligible) return -1;
    return 256 * (int64_t)s->R;
}

int osz_chain_zp_tolerance(osz_fir_t fir, osz_sos_t sos, double tol) {
    OSZ_REQUIRE(fir && sos, "osz_chain_zp_tolerance: null handle");
    OSZ_REQUIRE(tol == 0.0 || (tol >= 1e-18 && tol <= 1e-6), "osz_chain_zp_tolerance: tol=%g", tol);
    if (sos->zp && sos->zp->open) return fail(OSZ_ERR_STATE, "osz_chain_zp_tolerance: a zero-phase stream is open");
    if (sos->zp_tol != tol) {
        // the forward link (osz_chain_forward) cuts its right tail at the same tolerance: the
        // handles' own states are brought up to date before its tables go
        if (sos->spec) {
            int rc = spec_settle(sos->spec, nullptr);
            if (rc) return rc;
            OSZ_HIP(hipStreamSynchronize(nullptr));
            spec_unlink(sos->spec);
        }
        sos->zp_tol = tol;
        if (sos->zp) zp_unlink(sos->zp);        // the tables are rebuilt at the next use
    }
    return OSZ_OK;
}

int osz_chain_zp_reach(osz_fir_t fir, osz_sos_t sos, int64_t step) {
    OSZ_REQUIRE(fir && sos && step >= 0, "osz_chain_zp_reach: step=%lld", (long long)step);
    ChainZp *s = nullptr;
    int rc = zp_get(fir, sos, &s);
    if (rc) return rc;
    if (!s->eligible)
        return fail(OSZ_ERR_UNSUPPORTED, "osz_chain_zp_reach: this filter pair does not take the zero-phase kernel");
    s->ref_step = step;
    return OSZ_OK;
}

int64_t osz_chain_zp_min_chunk(osz_fir_t fir, osz_sos_t sos) {
    if (!fir || !sos) return -1;
    ChainZp *s = nullptr;
    if (zp_get(fir, sos, &s) != OSZ_OK || !s->eligible) return -1;
    return zp_min_chunk(s);
}

int osz_chain_zp_open(osz_fir_t fir, osz_sos_t sos, int64_t skip, void *stream) {
    OSZ_REQUIRE(fir && sos, "osz_chain_zp_open: null handle");
    OSZ_SAME_DEVICE(fir, "osz_chain_zp_open");
    OSZ_SAME_DEVICE(sos, "osz_chain_zp_open");
    hipStream_t st = as_stream(stream);
    ChainZp *s = nullptr;
    int rc = zp_get(fir, sos, &s);
    if (rc) return rc;
    if (!s->eligible)
        return fail(OSZ_ERR_UNSUPPORTED, "osz_chain_zp_open: this filter pair does not take the zero-phase kernel");
    const int cl = 4096 + 256 * s->R;
    OSZ_REQUIRE(skip >= 0 && skip + cl <= kSpecLdc, "osz_chain_zp_open: skip=%lld", (long long)skip);
    // the handles' own states must be current (the cascade's start state is read from the SOS handle)
    rc = spec_touch(sos->spec, st);
    if (rc) return rc;
    const int nch = fir->nch;
    // carry[skip + i] = what the cascade's start state rings, forwards, then filtered backwards
    double *cb = s->dcarry[s->cur];
    OSZ_HIP(hipMemsetAsync(cb, 0, sizeof(double) * (size_t)nch * kSpecLdc, st));
    rc = sos_forward_raw(sos, cb + skip, kSpecLdc, cb + skip, kSpecLdc, cl, st);
    if (rc) return rc;
    rc = sos_backward_raw(sos, cb + skip, kSpecLdc, cb + skip, kSpecLdc, cl, s->dzero, st);
    if (rc) return rc;
    OSZ_HIP(hipMemsetAsync(s->dheld[s->cur], 0, sizeof(double) * (size_t)nch * 256 * s->R, st));
    hipLaunchKernelGGL(zp_fill_ll_kernel, dim3((nch + 255) / 256), dim3(256), 0, st, s->dnanpos, nch,
                       0x7fffffffffffffffLL);
    OSZ_HIP(hipGetLastError());
    s->pos = 0;
    s->hist_n = 0;
    s->open = true;
    return OSZ_OK;
}

int osz_chain_zp_step(osz_fir_t fir, osz_sos_t sos, const double *x, int64_t ldx, int64_t n, double *y0,
                      int64_t ldy0, int64_t n0, double *y, int64_t ldy, void *stream) {
    OSZ_REQUIRE(fir && sos && x, "osz_chain_zp_step: null argument");
    OSZ_REQUIRE(n >= 1 && ldx >= n, "osz_chain_zp_step: n=%lld ldx=%lld", (long long)n, (long long)ldx);
    OSZ_REQUIRE(n0 >= 0 && n0 <= n && (n0 == 0 || (y0 && ldy0 >= n0)) && (n0 == n || (y && ldy >= n - n0)),
                "osz_chain_zp_step: outputs n0=%lld of n=%lld", (long long)n0, (long long)n);
    OSZ_SAME_DEVICE(fir, "osz_chain_zp_step");
    OSZ_SAME_DEVICE(sos, "osz_chain_zp_step");
    ChainZp *s = sos->zp;
    OSZ_REQUIRE(s && s->fir == fir && s->open, "osz_chain_zp_step: osz_chain_zp_open first");
    OSZ_REQUIRE(n >= zp_min_chunk(s), "osz_chain_zp_step: a chunk of %lld samples is shorter than two (pairs of) blocks (%lld)",
                (long long)n, (long long)zp_min_chunk(s));
    return zp_launch(s, x, ldx, n, y0, ldy0, n0, y, ldy, as_stream(stream));
}

int osz_chain_zp_finish(osz_fir_t fir, osz_sos_t sos, const double *x, int64_t ldx, int64_t m, double *y,
                        int64_t ldy, int64_t ny, void *stream) {
    OSZ_REQUIRE(fir && sos, "osz_chain_zp_finish: null handle");
    OSZ_SAME_DEVICE(fir, "osz_chain_zp_finish");
    OSZ_SAME_DEVICE(sos, "osz_chain_zp_finish");
    hipStream_t st = as_stream(stream);
    ChainZp *s = sos->zp;
    OSZ_REQUIRE(s && s->fir == fir && s->open, "osz_chain_zp_finish: no zero-phase stream is open");
    const int nch = fir->nch;
    FirPart &pt = fir->parts[0];
    const int wm1 = pt.ntaps - 1;
    // 1. the handles' own states at the end of the samples stepped so far: the plain
    // kernels over the last hist_cap of them (everything, when fewer have gone by: the
    // stream started from a zero FIR tail; the cascade's state then is approximate only
    // if fewer than warm_len samples were stepped, which the callers exclude)
    {
        int rc = spec_touch(sos->spec, st);
        if (rc) return rc;
    }
    OSZ_HIP(hipMemsetAsync(pt.dstate[pt.cur], 0, sizeof(double) * (size_t)nch * wm1, st));
    OSZ_HIP(hipMemsetAsync(sos->dstate, 0, sizeof(double) * (size_t)sos->nsec * nch * 2, st));
    if (s->hist_n > 0) {
        if (!s->dscratch) OSZ_HIP(hipMalloc(&s->dscratch, sizeof(double) * (size_t)nch * s->hist_cap));
        int rc = fir_push_raw(fir, s->dhist[s->hcur], s->hist_cap, s->hist_n, s->dscratch, s->hist_cap, 0, st);
        if (rc) return rc;
        rc = sos_forward_raw(sos, s->dscratch, s->hist_cap, s->dscratch, s->hist_cap, s->hist_n, st);
        if (rc) return rc;
    }
    // (a NaN never leaves the cascade: the replay knows nothing of one before its samples)
    hipLaunchKernelGGL(zp_poison_kernel, dim3((nch + 255) / 256), dim3(256), 0, st, s->dnanpos, sos->dstate,
                       sos->nsec, nch);
    OSZ_HIP(hipGetLastError());
    // 2. the output samples the stream is still short of need the head of what follows
    if (ny > 0) {
        OSZ_REQUIRE(x && y && m >= zp_min_chunk(s) && ldx >= m && ny <= m - zp_min_chunk(s) / 2 &&
                        ldy >= ny,
                    "osz_chain_zp_finish: %lld output samples from %lld input samples", (long long)ny, (long long)m);
        // through a scratch output of the link's own (the kernel writes all m of them), grown on
        // demand and kept: nothing here waits for the stream
        const size_t want = (size_t)nch * (size_t)m;
        if (want > s->fin_cap) {
            if (s->dfin) {
                // an earlier finish may still read it -- on this stream or on the one it came on
                if (s->fin_done) OSZ_HIP(hipEventSynchronize(s->fin_done));
                OSZ_HIP(hipStreamSynchronize(st));
                OSZ_HIP(hipFree(s->dfin));
                s->dfin = nullptr;
                s->fin_cap = 0;
            }
            if (hipMalloc(&s->dfin, sizeof(double) * want) != hipSuccess)
                return fail(OSZ_ERR_NOMEM, "osz_chain_zp_finish: scratch of %zu doubles", want);
            s->fin_cap = want;
        }
        // (the scratch is ordered by events, not by the caller's choice of stream: the kernel
        // below must not overwrite what the previous finish's copy is still reading)
        if (s->fin_done) OSZ_HIP(hipStreamWaitEvent(st, s->fin_done, 0));
        int rc = zp_launch(s, x, ldx, m, nullptr, 0, 0, s->dfin, m, st);
        if (rc) return rc;
        OSZ_HIP(hipMemcpy2DAsync(y, sizeof(double) * ldy, s->dfin, sizeof(double) * m, sizeof(double) * ny, nch,
                                 hipMemcpyDeviceToDevice, st));
        if (!s->fin_done) OSZ_HIP(hipEventCreateWithFlags(&s->fin_done, hipEventDisableTiming));
        OSZ_HIP(hipEventRecord(s->fin_done, st));
    }
    s->open = false;
    return OSZ_OK;
}

int osz_chain_zp_seal(osz_fir_t fir, osz_sos_t sos, double *y, int64_t ldy, int64_t n, int64_t s0,
                      int64_t origin, int64_t cs, void *stream) {
    OSZ_REQUIRE(fir && sos && y, "osz_chain_zp_seal: null argument");
    OSZ_REQUIRE(n >= 0 && ldy >= n && cs >= 1, "osz_chain_zp_seal: n=%lld cs=%lld", (long long)n, (long long)cs);
    ChainZp *s = sos->zp;
    OSZ_REQUIRE(s && s->fir == fir, "osz_chain_zp_seal: no zero-phase stream");
    if (n == 0) return OSZ_OK;
    hipLaunchKernelGGL(zp_seal_kernel, dim3(4, fir->nch), dim3(256), 0, as_stream(stream), y, ldy, n,
                       (long long)s0, (long long)origin, (long long)cs, s->dnanpos, (long long)s->ref_step);
    OSZ_HIP(hipGetLastError());
    return OSZ_OK;
}

}  // extern "C"
