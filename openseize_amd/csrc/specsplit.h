// specsplit.h -- on-chip spectra for even transform lengths whose half does not fit the LDS
// (nfft = int(fs / resolution), spectra/estimators.py:143-144: 50 000 at 5 kHz and 0.1 Hz,
// 60 000 at 30 kHz and the default resolution).
//
// M = nfft / 2 = R0 S0.  The first decimation-in-frequency pass of the M-point transform leaves
// R0 INDEPENDENT S0-point transforms: bin k = q + R0 k' is output k' of
//     u_q[b] = W_M^(b q) * sum_n z[b + n S0] W_R0^(n q),          b < S0,
// and the bins k and M - k the real transform is untangled from (specmix.h) sit in the
// sub-transforms q and R0 - q.  So a workgroup takes ONE pair (q, R0 - q): it reads the whole
// segment -- every workgroup of a segment does; they are dispatched onto the same XCD one behind
// the other, so that the segment comes from HBM once and from that XCD's L2 the other times --
// forms its two u_q as it reads (R0 complex multiply-adds per point and output, constants of
// the pair in LDS: nothing of the pass is held in registers, R0 is any divisor), and runs the
// S0-point passes of specmix.h on its 2 S0 points of LDS.  No intermediate array in HBM: the
// staging route (spec_prep -> rocFFT's two or four kernels -> spec_post) moves 40-100 B per
// new sample, this one the samples and the partial sums.
//   workgroups of a segment: the pairs (q, R0 - q), q = 1 .. ceil(R0 / 2) - 1, and one for the
//   sub-transforms that pair with themselves: q = 0 and, R0 even, q = R0 / 2 -- the two together
//   (2 S0 points again); q = 0 alone when R0 is odd (S0 points, lane maps of their own).
#pragma once

#include "specmix.h"

#ifndef OSZ_SPLIT_ABL      // (timing builds with one phase switched off)
#define OSZ_SPLIT_ABL 0
#endif

namespace osz {
namespace mix {

constexpr int kMaxSplitLocal = 10176;   // 2 S0 points of 16 B, the reduction's and the pair's words: 160 KB
constexpr int kMaxSplitR0 = 32;

struct SplitArgs {
    Args a;            // N, M = N / 2; radix[1 .. npass) = the plan of S0 (radix[0] = R0); lane maps of the 2 S0 local points;
                       // pos = slots of the S0-point sub-transform's outputs
    int R0, S0, NW;    // NW workgroups per (run, channel)
    int nunits;        // nruns * nch
    // the lane maps of a self-paired workgroup's S0 local points (a.blkfast / div / inv: of a pair's 2 S0)
    int blkfast1[kMaxPass], div1[kMaxPass];
    unsigned inv1[kMaxPass];
};

template <int MODE, bool LINEAR, int NT>
__global__ __launch_bounds__(NT) void specsplit_kernel(SplitArgs g) {
    extern __shared__ C2 zmix[];
    constexpr int NWV = NT / 64;
    __shared__ double red[NWV][4];
    __shared__ C2 cst[2][kMaxSplitR0];  // W_R0^(qa n), W_R0^(qb n)
    const Args &a = g.a;
    C2 *z = zmix;
    const int t = threadIdx.x;
    // blockIdx -> (unit, pair): consecutive workgroup ids go round the eight XCDs, so the NW
    // workgroups of a unit are the ids  8 (grp NW + w) + xcd,  w < NW -- one XCD, one behind the other
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int grp = slot / g.NW, w = slot - grp * g.NW;
    const int unit = grp * 8 + xcd;
    if (unit >= g.nunits) return;
    const int run = unit % a.nruns, c = unit / a.nruns;
    const int R0 = g.R0, S0 = g.S0, M = a.M, N = a.N, NF = a.M + 1;
    // the workgroup's two sub-transforms: a pair (w, R0 - w); w = 0: the self-paired ones, 0 and
    // (R0 even) R0 / 2 -- `twin` -- or 0 alone -- `lone`
    const bool pair = w != 0, twin = w == 0 && (R0 & 1) == 0, lone = w == 0 && (R0 & 1) != 0;
    const int qa = w, qb = pair ? R0 - w : twin ? R0 / 2 : 0;
    const int Ml = lone ? S0 : 2 * S0;              // local points
    const double *xr = a.x + (int64_t)c * a.ldx;
    const int64_t s0 = ((int64_t)run * a.nseg) / a.nruns;
    const int64_t s1 = ((int64_t)(run + 1) * a.nseg) / a.nruns;
    const double s2 = a.scale * a.scale;
    // pairs of bins: (i, S0 - 1 - i) across a pair's two halves; within sub-transform 0 (i, S0 - i),
    // i <= S0 / 2; within sub-transform R0 / 2 (i, S0 - 1 - i), i < (S0 + 1) / 2
    const int np0 = S0 / 2 + 1;
    const int npairs = pair ? S0 : twin ? np0 + (S0 + 1) / 2 : np0;
    double acc[kAcc];
#pragma unroll
    for (int m = 0; m < kAcc; ++m) acc[m] = 0.0;
    if (t < R0) {
        cst[0][t] = *reinterpret_cast<const C2 *>(a.tw + 2 * (((qa * t) % R0) * (N / R0)));
        cst[1][t] = *reinterpret_cast<const C2 *>(a.tw + 2 * (((qb * t) % R0) * (N / R0)));
    }
    const double mid = 0.5 * (a.nwin - 1);
    const bool wpair = (a.nwin & 1) == 0;
    const int jh = a.halfcarry ? M / 2 : 0;       // points of a segment's first half, when its sums are carried
    double carry_sum = 0.0, carry_lin = 0.0;

    for (int64_t s = s0; s < s1; ++s) {
        const double *xs = xr + s * (int64_t)a.stride;
        const __amdgpu_buffer_rsrc_t rx = row_rsrc(xs, a.nwin);
        const __amdgpu_buffer_rsrc_t rw = row_rsrc(a.window, a.nwin);
        // ---- the trend's sums (a first half's come with the previous segment, specmix.h)
        const bool full = !a.halfcarry || s == s0;
        {
            double sA = 0.0, lA = 0.0, sB = 0.0, lB = 0.0;
            if (full) {
#pragma unroll 4
                for (int j = t; j < jh; j += NT) {
                    const unsigned at = 16u * (unsigned)j;
                    const double x0 = buf_load(rx, at, 0), x1 = buf_load(rx, at + 8, 0);
                    sA += x0 + x1;
                    if (LINEAR) lA += (2 * j - mid) * x0 + (2 * j + 1 - mid) * x1;
                }
            }
#pragma unroll 4
#if OSZ_SPLIT_ABL == 4
            for (int j = jh + t; j < 0; j += NT) {
#else
            for (int j = jh + t; j < M; j += NT) {
#endif
                const unsigned at = 16u * (unsigned)j;
                const double x0 = buf_load(rx, at, 0), x1 = buf_load(rx, at + 8, 0);
                sB += x0 + x1;
                if (LINEAR) lB += (2 * j - mid) * x0 + (2 * j + 1 - mid) * x1;
            }
            sA = wave_sum63(sA);
            sB = wave_sum63(sB);
            if (LINEAR) {
                lA = wave_sum63(lA);
                lB = wave_sum63(lB);
            }
            if ((t & 63) == 63) {
                red[t >> 6][0] = sA;
                red[t >> 6][1] = lA;
                red[t >> 6][2] = sB;
                red[t >> 6][3] = lB;
            }
        }
        __syncthreads();   // (also: the bin reads of the previous segment are done, cst[] is written)
        double mean, slope = 0.0;
        {
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
                totA = carry_sum;
                linA = carry_lin - 0.5 * a.nwin * carry_sum;
            }
            carry_sum = uniform(totB);
            if (LINEAR) carry_lin = uniform(linB);
            mean = (totA + totB) / a.nwin;
            if (LINEAR) {
                const double nn = (double)a.nwin;
                const double sxx = nn * (nn * nn - 1.0) / 12.0;
                slope = sxx > 0.0 ? (linA + linB) / sxx : 0.0;
            }
        }
        // ---- samples, trend off, window on, the pair's two outputs of the first pass
        for (int b = t; b < S0; b += NT) {
            double ar = 0.0, ai = 0.0, br = 0.0, bi = 0.0;
#pragma unroll 4
            for (int n = 0; n < R0; ++n) {
                const int j = b + n * S0;
                const unsigned at = 16u * (unsigned)j;
#if OSZ_SPLIT_ABL == 3
                const double x0 = 1.0 + at, x1 = 2.0;
                double g0, g1;
                if (at == 12345u) {
#else
                const double x0 = buf_load(rx, at, 0), x1 = buf_load(rx, at + 8, 0);
                double g0, g1;                                                           // 0 in the padding
                if (wpair) {
#endif
                    const buf_d2 gg = buf_load2(rw, at, 0);     // (an even window: a point lies inside it or beyond it)
                    g0 = gg[0];
                    g1 = gg[1];
                } else {
                    g0 = buf_load(rw, at, 0);
                    g1 = buf_load(rw, at + 8, 0);
                }
                double vr, vi;
                if (LINEAR) {
                    vr = (x0 - mean - slope * (2 * j - mid)) * g0;
                    vi = (x1 - mean - slope * (2 * j + 1 - mid)) * g1;
                } else {
                    vr = (x0 - mean) * g0;
                    vi = (x1 - mean) * g1;
                }
                const C2 ca = cst[0][n], cb = cst[1][n];
                ar = fma(vr, ca.re, fma(-vi, ca.im, ar));     // u_qa += v W_R0^(qa n)
                ai = fma(vr, ca.im, fma(vi, ca.re, ai));
                br = fma(vr, cb.re, fma(-vi, cb.im, br));     // u_qb += v W_R0^(qb n)
                bi = fma(vr, cb.im, fma(vi, cb.re, bi));
            }
            const C2 wa = *reinterpret_cast<const C2 *>(a.tw + 2 * (2 * b * qa));
            const C2 wb = *reinterpret_cast<const C2 *>(a.tw + 2 * (2 * b * qb));
            z[b] = cmul(C2{ar, ai}, wa);
            if (!lone) z[S0 + b] = cmul(C2{br, bi}, wb);
        }
        int tt = t;
        asm volatile("" : "+v"(tt));
        // ---- the passes of the two S0-point transforms, in place (specmix.h)
        int B = S0;
#if OSZ_SPLIT_ABL == 1
        for (int p = 1; p < 1; ++p) {
#else
        for (int p = 1; p < a.npass; ++p) {
#endif
            const int r = a.radix[p];
            const int S = B / r;
            const Split sp = lone ? Split{g.blkfast1[p], g.div1[p], g.inv1[p]} : Split{a.blkfast[p], a.div[p], a.inv[p]};
            C2 w1[kPreBf];
            pass_twiddles<NT>(w1, t, Ml / r, N / B, a.tw, sp);
            __syncthreads();
            if (r == 10) pass<10, NT>(z, t, Ml, S, w1, sp, N / B, a.tw);
            else if (r == 4) pass<4, NT>(z, t, Ml, S, w1, sp, N / B, a.tw);
            else if (r == 5) pass<5, NT>(z, t, Ml, S, w1, sp, N / B, a.tw);
            else if (r == 2) pass<2, NT>(z, t, Ml, S, w1, sp, N / B, a.tw);
            else if (r == 3) pass<3, NT>(z, t, Ml, S, w1, sp, N / B, a.tw);
            else pass<7, NT>(z, t, Ml, S, w1, sp, N / B, a.tw);
            B = S;
        }
        __syncthreads();
        // ---- bins k = qa + R0 i and M - k, out of the pair's two sub-transforms
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int m = 0; m < kAcc / 2; ++m) {
            const int i = tt + NT * m;
            asm volatile("" ::: "memory");   // one pair at a time
#if OSZ_SPLIT_ABL == 2
            if (i < 0) {
#else
            if (i < npairs) {
#endif
                int ia, ib, offa = 0, offb, k;
                if (pair) {
                    ia = i, ib = S0 - 1 - i, offb = S0, k = qa + R0 * i;
                } else if (i < np0) {
                    ia = i, ib = i == 0 ? 0 : S0 - i, offb = 0, k = R0 * i;
                } else {
                    ia = i - np0, ib = S0 - 1 - ia, offa = offb = S0, k = R0 / 2 + R0 * ia;
                }
                const C2 za = z[offa + a.pos[ia]], zb = z[offb + a.pos[ib]];
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
            const int i = t + NT * m;
            if (i < npairs) {
                const int k = pair ? qa + R0 * i : i < np0 ? R0 * i : R0 / 2 + R0 * (i - np0);
                o[k] = acc[2 * m];
                if (M - k != k) o[M - k] = acc[2 * m + 1];
            }
        }
    }
}

}  // namespace mix
}  // namespace osz
