// chain_zpn_body.h (compiled as chain_zpn_{2,4,6,8}.hip, one per mode count) -- FIR -> sosfiltfilt of a long stream, ONE real block per transform.
//
// Same scheme as chain_zp.hip (oaconvolve, core/numerical.py:158-298, into sosfiltfilt,
// :338-411, as one multiplication per bin by H_fir |H_iir|^2 plus mode bursts for what the
// cascade's two-sided ringing does in a finite window), on a different transform: fft::nega
// (fft4096.h) takes a window of 8192 samples of ONE real block through the 4096-point complex
// transform at the odd frequencies.  The FIR's tail (wlen - 1 samples) and the guard row the
// fit reads are paid once per 8192-sample window instead of once per 4096: at 1024 taps a
// transform carries 27 rows of 256 samples where the pair trick carries 2 x 11, and the fit
// and the bursts are one block's, not two.
//
// The window (rows of 256 samples; a thread holds sample t of every row: rows 0-15 in re[],
// rows 16-31 in im[]):
//     rows 0 .. NB-1        the block's outputs (rows 0 .. D-1 also take the previous block's
//                           tail rows, D = 32 - NB, carried in registers)
//     rows NB .. 30         FIR tail + ringing: the next block's rows 0 .. D-2
//     row 31                ringing only: the fit's samples (first and last nh), then the next
//                           block's row D-1
// The wrap is NEGACYCLIC: what leaves the window on one side returns on the other with its
// sign changed.  With mu (causal, at window sample 8192) and nu (anticausal, at sample -1)
// from the joint fit (spec::build_zpn gives nu its true sign) a block adds
//     +mu forwards from its row 0          (the wrapped right tail leaves the window)
//     +nu backwards from its row 31        (the wrapped left tail leaves the window)
// and receives the true tails of its neighbours:
//     +mu of the PREVIOUS block forwards from row D   (kapP, one block old)
//     +nu of the NEXT block backwards from row NB-1   (the last R rows of a block wait in
//                                                      registers for the next block's fit: the
//                                                      output stream runs L = 256 R samples late)
// Three burst evaluations per row index serve the four places (the left tail leaves row 31-r
// and arrives in the held rows with the same values).  Runs, the opening block (carry, held),
// NaN reach, hist: as in chain_zp.hip.  The closing block (1 .. S samples) is the same body
// with predicated loads and stores: every sum stays inside a thread's own column (block and
// bursts start at whole rows), so there is no accumulation through LDS; what lies behind the
// chunk's end goes to `carry`.
//
// A block's rows come in by LDS-DMA (zp_request_rows, chain_zp.h) behind the previous block's
// transform.
#include <type_traits>

#include "chain_zp.h"
#include "nega_window.h"

namespace osz {

// The fit and the burst amplitudes of a block in one stage.  Sixteen consecutive lanes (one DPP
// row) share one amplitude -- the right tail's mu of a slow mode (NS of them) or the left tail's
// nu of any mode (NM): each lane takes a sixteenth of the 2 nh fit samples for the real AND the
// imaginary row of M, four DPP steps add the parts up in the row's last lane, and that lane
// writes kappa[r] = amplitude * lambda^(256 r), r < R (row after row: one product by lambda^256 each;
// round 5 -- until then a table of R rows of powers sat in LDS for this one lane):
//   kmu[r][q], q < NS    this block's mu: read by this block's bursts AND, one block later, as the
//                        previous block's (the caller alternates between two such arrays)
//   knu[r][q], q < NM    this block's nu (rows behind the first are read for the slow modes only)
// M's rows: Re mu [NS], Im mu [NS], Re nu [NM], Im nu [NM].  l256[q] = lambda_q^256.
template <int NM, int NS, int PER, int RM, bool ZP>
__device__ __forceinline__ void zpn_fit_kappa_n(int tt, int R, const double *fitbuf, const double *mtab,
                                                const double *l256, double *kmu, double *knu) {
    constexpr int ns = 16 * PER;
    constexpr int NA = NS + (ZP ? NM : 0);       // amplitudes: the forward chain has no left tail
    if ((tt & ~63) >= 16 * NA) return;           // whole waves without an amplitude skip the stage
    const int qd = tt >> 4, p16 = tt & 15;
    const bool valid = qd < NA;
    const bool is_mu = qd < NS;
    const int q = valid ? (is_mu ? qd : qd - NS) : 0;
    const int rre = is_mu ? q : 2 * NS + q, rim = is_mu ? NS + q : 2 * NS + NM + q;
    const double *yb = fitbuf + PER * p16;
    const double *mr = mtab + ns * rre + PER * p16;
    const double *mi = mtab + ns * rim + PER * p16;
    double y[PER], a[PER], b[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        y[k] = yb[k];
        a[k] = mr[k];
        b[k] = mi[k];
    }
    double sr = 0.0, si = 0.0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        sr = fma(a[k], y[k], sr);
        si = fma(b[k], y[k], si);
    }
    if (!valid) sr = si = 0.0;
    sr += dpp_row_shr0<1>(sr);
    si += dpp_row_shr0<1>(si);
    sr += dpp_row_shr0<2>(sr);
    si += dpp_row_shr0<2>(si);
    sr += dpp_row_shr0<4>(sr);
    si += dpp_row_shr0<4>(si);
    sr += dpp_row_shr0<8>(sr);
    si += dpp_row_shr0<8>(si);
    if (p16 == 15 && valid) {
        const double lr = l256[q * 2 + 0], li = l256[q * 2 + 1];
        double kr = sr, ki = si;
#pragma unroll
        for (int r = 0; r < RM; ++r) {
            if (r < R) {
                if (is_mu) {
                    kmu[(r * NS + q) * 2 + 0] = kr;
                    kmu[(r * NS + q) * 2 + 1] = ki;
                } else {
                    knu[(r * NM + q) * 2 + 0] = kr;
                    knu[(r * NM + q) * 2 + 1] = ki;
                }
                const double nr = kr * lr - ki * li;
                ki = kr * li + ki * lr;
                kr = nr;
            }
        }
    }
}

// nh is 16, 24 or 32 (spec::build_zpn)
template <int NM, int NS, int RM, bool ZP>
__device__ __forceinline__ void zpn_fit_kappa(int tt, int nh, int R, const double *fitbuf, const double *mtab,
                                              const double *l256, double *kmu, double *knu) {
    if (nh == 24) zpn_fit_kappa_n<NM, NS, 3, RM, ZP>(tt, R, fitbuf, mtab, l256, kmu, knu);
    else if (nh == 32) zpn_fit_kappa_n<NM, NS, 4, RM, ZP>(tt, R, fitbuf, mtab, l256, kmu, knu);
    else zpn_fit_kappa_n<NM, NS, 2, RM, ZP>(tt, R, fitbuf, mtab, l256, kmu, knu);
}

// lambda_q^e, q = q0 .. q0 + NG - 1, e = 0..255, from the three-level table [20][NM][2]
template <int NM, int NG>
__device__ __forceinline__ void zpn_powers(const double *ptab, int q0, int e, double *pr, double *pi) {
    const double *p1 = ptab + ((e >> 5) * NM + q0) * 2, *p2 = ptab + ((8 + ((e >> 2) & 7)) * NM + q0) * 2,
                 *p3 = ptab + ((16 + (e & 3)) * NM + q0) * 2;
#pragma unroll
    for (int q = 0; q < NG; ++q) {
        const double ar = p1[2 * q] * p2[2 * q] - p1[2 * q + 1] * p2[2 * q + 1];
        const double ai = p1[2 * q] * p2[2 * q + 1] + p1[2 * q + 1] * p2[2 * q];
        pr[q] = ar * p3[2 * q] - ai * p3[2 * q + 1];
        pi[q] = ar * p3[2 * q + 1] + ai * p3[2 * q];
    }
}

// two bursts of one row at once: Re sum_q k0_q P_q and Re sum_q k1_q P_q
template <int NM>
__device__ __forceinline__ void zpn_dot2(const double *k0, const double *k1, const double *pr, const double *pi,
                                         double &d0, double &d1) {
    double ar = 0.0, ai = 0.0, br = 0.0, bi = 0.0;
#pragma unroll
    for (int q = 0; q < NM; ++q) {
        ar = fma(k0[2 * q], pr[q], ar);
        ai = fma(k0[2 * q + 1], pi[q], ai);
        br = fma(k1[2 * q], pr[q], br);
        bi = fma(k1[2 * q + 1], pi[q], bi);
    }
    d0 = ar - ai;
    d1 = br - bi;
}

// The forward bursts (RF rows, the NS slow modes): this block's mu leaves row r, the previous
// block's arrives in row D + r (both rows live in re[]: D + r <= 8 + 4).
template <int D, int NS, int RF>
__device__ __forceinline__ void zpn_fwd_bursts(double *re, const double *kmu, const double *kpm, const double *Pr,
                                               const double *Pi, double &first) {
#pragma unroll
    for (int r = 0; r < RF; ++r) {
        double ca, cp;
        zpn_dot2<NS>(kmu + (r * NS) * 2, kpm + (r * NS) * 2, Pr, Pi, ca, cp);
        re[r] += ca;
        re[D + r] += cp;
        if (r == 0) first = ca;
    }
}

// The backward bursts (RB rows): this block's nu leaves window row 31 - r (im[15 - r]) and
// arrives, with the same values, in the rows the previous block holds back (c7).  The slow modes
// in every row; the fast ones (all at once: one more round of table reads, not one per pair)
// in the first.
template <int NM, int NS, int RB>
__device__ __forceinline__ void zpn_bwd_bursts(double *im, double *c7, const double *knu, const double *ptab, int e,
                                               const double *Pr, const double *Pi) {
    double nb[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) nb[r] = zp_dot<NS>(knu + (r * NM) * 2, Pr, Pi);
    if (NM > NS) {
        constexpr int NF = NM > NS ? NM - NS : 1;
        double fr[NF], fi[NF];
        zpn_powers<NM, NF>(ptab, NS, e, fr, fi);
        nb[0] += zp_dot<NF>(knu + NS * 2, fr, fi);
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        im[15 - r] += nb[r];
        c7[r] = nb[r];
    }
}

// RM: burst rows the instance holds registers for (5, or 8 for the long left tails of blocks of
// 24 ... 26 rows).  ZP = false: the FORWARD chain (FIR -> sosfilt, chain_spec.hip's
// osz_chain_forward): no left tail -- no backward bursts, no rows held back, no output lag, the
// fit reads the right tail's amplitudes only (tables of spec::build_specn), the runs are cut
// evenly (the NaN seal behind the launch finds their ends that way).
// Diagnostic builds (benchmarks/zpn_variant.hip, never the library): OSZ_ABL_NOEPI / NOSTORE / NOH /
// NODMA / NOLDS drop the fit and bursts / the row stores / the spectrum loads / the row requests /
// the cube's LDS traffic (results then meaningless: OSZ_ABL_ANY also switches the NaN bookkeeping
// off so that garbage does not take the slow paths); OSZ_CLK records the shader clock and the
// 100 MHz clock over one workgroup's life (profiles/r05_zpn_ablation.txt).
#ifdef OSZ_CLK
__device__ unsigned long long g_zpn_clk[4];
#endif
template <int NB, int NM, int NS, int RM = kSpecRMax, bool ZP = true>
__global__ __launch_bounds__(256, 2) void chain_zpn_kernel(ZpArgs g) {
    constexpr int D = 32 - NB, S = 256 * NB, NHI = NB - 16, NP = (NB + 1) / 2;
    static_assert(NB >= 20 && NB <= 30, "rows per block");
    extern __shared__ fft::cube::C2 cube_lds[];
    const int Rt = g.R;                                // rows of the amplitude tables
    const int R = ZP ? g.R : 0;                        // rows of the left tail, and of the lag
    const int Rf = g.Rf, nh = g.nh, ns = 2 * nh;
    constexpr int NMROWS = 2 * NS + (ZP ? 2 * NM : 0);  // rows of the fit matrix
    double *xl = reinterpret_cast<double *>(cube_lds) + 2 * fft::cube::SLOTS;   // behind the cube
    double *fitbuf = xl;                               // [2 nh]
    double *kapP = fitbuf + ns;                        // [2 parity][Rt][NS][2]: this block's mu and the previous block's (slow modes)
    double *knu = kapP + 2 * Rt * NS * 2;              // [Rt][NM][2]: this block's nu (ZP)
    double *lrow = knu + (ZP ? Rt * NM * 2 : 0);       // [NM][2]: lambda^256
    double *ptab = lrow + NM * 2;                      // [20][NM][2]
    double *mtab = ptab + 20 * NM * 2;                 // [2 NS (+ 2 NM)][2 nh]
    fft::cube::C2 *tw2l = reinterpret_cast<fft::cube::C2 *>(mtab + NMROWS * ns);   // [4 q][16 n0]
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
    // scalar registers -- a chunk has far fewer than 2^31 blocks)
    const int p0 = __builtin_amdgcn_readfirstlane(
        (int)(ZP ? zp_run_start(run, g.W, g.nruns, g.wclose) : ((int64_t)run * g.W) / g.nruns));
    const int p1 = __builtin_amdgcn_readfirstlane(
        (int)(ZP ? zp_run_start(run + 1, g.W, g.nruns, g.wclose) : ((int64_t)(run + 1) * g.W) / g.nruns));
    const int first = run == 0 ? 0 : p0 - 1;
    const int lastf = p1 - 1;
    const bool closes = run == g.nruns - 1;

#ifdef OSZ_CLK
    unsigned long long clk0 = 0, rt0 = 0;
    if (blockIdx.x == 0 && blockIdx.y == 7)
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk0), "=s"(rt0)::"memory");
#endif
    NegaWindow<NHI> P{a, cube_lds, tw2l};
    {
        fft::cube::TwPow w2;                               // (its sixteen rows go to LDS below)
        fft::nega::tw_load(t, a.tb, P.tw1, w2);
        if (t < 16) {
#pragma unroll
            for (int q = 0; q < 4; ++q) tw2l[16 * q + t] = fft::cube::C2{w2.r[q], w2.i[q]};
        }
    }
    double cr[D];                    // the previous block's rows NB .. 31: this block's rows 0 .. D-1
#pragma unroll
    for (int j = 0; j < D; ++j) cr[j] = 0.0;
#ifndef OSZ_ABL_NODMA
    if (first <= lastf) zp_request_rows<NP>(xr + (int64_t)first * S, NB, t, cube_lds);
#endif
    {
        // lrow | ptab | mtab are one table on the device too (g.Lrow)
        const int ntab = NM * 2 + 20 * NM * 2 + NMROWS * ns;
#pragma unroll 8
        for (int i = t; i < ntab; i += 256) lrow[i] = g.Lrow[i];
    }
    for (int i = t; i < 2 * Rt * NS * 2; i += 256) kapP[i] = 0.0;
    double held[RM];                 // rows NB-1-r of the previous block, one burst short
#pragma unroll
    for (int r = 0; r < RM; ++r) held[r] = 0.0;
    // `bad` is uniform (a scalar), and sticky: the stream went bad in an earlier chunk, or a block
    // of this run held non-finite samples -- behind the transform they are everywhere
#ifdef OSZ_ABL_ANY
    bool bad = false;
#else
    bool bad = ZP ? g.nanpos[c] != 0x7fffffffffffffffLL : false;
#endif
    int64_t bad_at = 0;
    int par = 0;
    int younger = -1;                // vector-memory operations behind the pending requests
    __syncthreads();

    // a sample of the chunk (position i, value v) goes to the output, L samples late, or,
    // the chunk's last L samples, to `held`
#define OSZ_ZP_PUT(i_, v_)                      \
    do {                                        \
        const int64_t q_ = (i_) + L;            \
        if (q_ < n0) y0r[q_] = (v_);            \
        else if (q_ < n) yr[q_] = (v_);         \
        else ho[q_ - n] = (v_);                 \
    } while (0)

    // One block.  `closing` (a compile-time tag: the two bodies are allocated apart, the ragged
    // one does not weigh on the hot loop) is the chunk's last block of g.la samples.
    auto block = [&](const int p, auto closing_tag) __attribute__((always_inline)) {
        constexpr bool closing = decltype(closing_tag)::value;
        const int64_t o = (int64_t)p * S;
        double re[16], im[16];
        take_turns();
#ifdef OSZ_NEGA_STAMPS
#define OSZ_BSTAMP(slot_) OSZ_NSTAMP(P.sacc, P.slast, slot_)
#else
#define OSZ_BSTAMP(slot_) do { } while (0)
#endif
        OSZ_BSTAMP(16);   // the previous block's stores issued (+ loop overhead)
        if (!closing) {
            // the block's samples were requested behind the previous block's transform: younger
            // than they are only that block's stores (requests and stores retire in order on one
            // counter), `younger` of them when every one went through the row stores
            if (younger == NB) asm volatile("s_waitcnt vmcnt(%0) ; osz:dma" ::"n"(NB) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) ; osz:dma" ::: "memory");
            int tq = t;
            asm volatile("" : "+v"(tq));
            const double *xs = reinterpret_cast<const double *>(cube_lds) + 128 * (tq >> 6) + (tq & 63);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                re[j] = xs[512 * (j >> 1) + 64 * (j & 1)];
                im[j] = j < NHI ? xs[512 * ((j + 16) >> 1) + 64 * ((j + 16) & 1)] : 0.0;
            }
        } else {
            const int la = g.la;
            const double *xc = xr + o + t;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                re[j] = 256 * j + t < la ? xc[256 * j] : 0.0;
                im[j] = (j < NHI && 4096 + 256 * j + t < la) ? xc[4096 + 256 * j] : 0.0;
            }
        }
        OSZ_BSTAMP(0);    // the wait for this block's samples + their read from LDS
        // (the next block's samples are requested inside, once inverse pass 1 has read this wave's
        // pieces of the cube: they land while its arithmetic, fit, bursts and stores run)
        P.template transform<NP>(t, re, im, (!closing && p < lastf) ? xr + o + S : nullptr, NB);
        int nst = 0;             // row stores of this block, -1: some went another way
        int tt = t;
        asm volatile("" : "+v"(tt));
        const unsigned lane8 = 8u * (unsigned)tt;
#ifndef OSZ_ABL_NOEPI
        if (tt < nh) fitbuf[tt] = im[15];
        else if (tt >= 256 - nh) fitbuf[tt - 256 + ns] = im[15];
#endif
#pragma unroll
        for (int j = 0; j < D; ++j) re[j] += cr[j];
        OSZ_BSTAMP(11);   // fit samples to LDS + overlap add
#ifndef OSZ_ABL_NOEPI
        __syncthreads();
        OSZ_BSTAMP(12);   // barrier 5
        // (this block's mu where the next block will look for the previous one's)
        double *kmu = kapP + (par ^ 1) * (Rt * NS * 2);
        zpn_fit_kappa<NM, NS, RM, ZP>(tt, nh, Rt, fitbuf, mtab, lrow, kmu, knu);
        __syncthreads();
#endif
        OSZ_BSTAMP(13);   // fit + barrier 6
        double c7[RM];
#pragma unroll
        for (int r = 0; r < RM; ++r) c7[r] = 0.0;
#ifndef OSZ_ABL_NOEPI
        {
            double Pr[NS], Pi[NS];
            zpn_powers<NM, NS>(ptab, 0, tt, Pr, Pi);
            const double *kpm = kapP + par * (Rt * NS * 2);
            double ca = 0.0;
            switch (Rf) {
                case 1: zpn_fwd_bursts<D, NS, 1>(re, kmu, kpm, Pr, Pi, ca); break;
                case 2: zpn_fwd_bursts<D, NS, 2>(re, kmu, kpm, Pr, Pi, ca); break;
                case 3: zpn_fwd_bursts<D, NS, 3>(re, kmu, kpm, Pr, Pi, ca); break;
                case 4: zpn_fwd_bursts<D, NS, 4>(re, kmu, kpm, Pr, Pi, ca); break;
                default: zpn_fwd_bursts<D, NS, 5>(re, kmu, kpm, Pr, Pi, ca); break;   // (Rf <= 5: the tables)
            }
            // (no fence between the two halves: with two slow modes both sets of powers fit beside
            // the data, and the backward half's table reads go out behind the forward half's)
            double Qr[NS], Qi[NS];
            if (ZP) zpn_powers<NM, NS>(ptab, 0, 255 - tt, Qr, Qi);
            if (ZP) switch (R) {
                case 1: zpn_bwd_bursts<NM, NS, 1>(im, c7, knu, ptab, 255 - tt, Qr, Qi); break;
                case 2: zpn_bwd_bursts<NM, NS, 2>(im, c7, knu, ptab, 255 - tt, Qr, Qi); break;
                case 3: zpn_bwd_bursts<NM, NS, 3>(im, c7, knu, ptab, 255 - tt, Qr, Qi); break;
                case 4: zpn_bwd_bursts<NM, NS, 4>(im, c7, knu, ptab, 255 - tt, Qr, Qi); break;
                case 5: zpn_bwd_bursts<NM, NS, 5>(im, c7, knu, ptab, 255 - tt, Qr, Qi); break;
                case 6: zpn_bwd_bursts<NM, NS, RM >= 6 ? 6 : 5>(im, c7, knu, ptab, 255 - tt, Qr, Qi); break;
                case 7: zpn_bwd_bursts<NM, NS, RM >= 7 ? 7 : 5>(im, c7, knu, ptab, 255 - tt, Qr, Qi); break;
                case 8: zpn_bwd_bursts<NM, NS, RM >= 8 ? 8 : 5>(im, c7, knu, ptab, 255 - tt, Qr, Qi); break;
                case 9: zpn_bwd_bursts<NM, NS, RM >= 9 ? 9 : 5>(im, c7, knu, ptab, 255 - tt, Qr, Qi); break;
                case 10: zpn_bwd_bursts<NM, NS, RM >= 10 ? 10 : 5>(im, c7, knu, ptab, 255 - tt, Qr, Qi); break;
                case 11: zpn_bwd_bursts<NM, NS, RM >= 11 ? 11 : 5>(im, c7, knu, ptab, 255 - tt, Qr, Qi); break;
                default: zpn_bwd_bursts<NM, NS, RM >= 12 ? 12 : 5>(im, c7, knu, ptab, 255 - tt, Qr, Qi); break;
            }
            // non-finite samples are everywhere behind the transform: every amplitude of the fit
            // and with it every lane's burst values (looked at once, behind both halves)
#ifndef OSZ_ABL_ANY
            if (!bad && __builtin_amdgcn_readfirstlane((int)sos_not_finite(ca))) {
                bad = true;
                bad_at = o;
                // (the exact position where it can be had: chain_zp.h)
                if (ZP && zp_exact_nanpos(xr + o, closing ? g.la : S, tt, reinterpret_cast<long long *>(g.nanpos + c), g.pos + o))
                    bad_at = -1;
            }
#endif
        }
#endif
        OSZ_BSTAMP(14);   // bursts
        const double qn = spec_qnan();
        const bool own = p >= p0;                      // (a run's first block may be its neighbour's)
        const bool edge = p >= g.W - 1;                // its samples may be among the chunk's last L
        if (run == 0 && p == 0) {
            // the chunk opens: what the stream so far still owes these samples, and the
            // previous chunk's last L samples, complete with this block's +nu
            nst = -1;
            const double *ci = g.carry_in + (int64_t)c * kSpecLdc + tt;
#ifndef OSZ_ABL_ANY
            if (!bad &&
                __builtin_amdgcn_readfirstlane((int)sos_not_finite(g.carry_in[(int64_t)c * kSpecLdc + 4095 + 256 * R]))) {
                bad = true;
                bad_at = 0;
            }
#endif
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                re[j] += ci[256 * j];
                if (j < NHI) im[j] += ci[4096 + 256 * j];
            }
            if (ZP) {
                const double *hi = g.held_in + (int64_t)c * L + tt;
#pragma unroll
                for (int r = 0; r < RM; ++r)
                    if (r < R) {
                        const int64_t q = 256 * (R - 1 - r) + tt;
                        (q < n0 ? y0r : yr)[q] = bad ? qn : hi[256 * (R - 1 - r)] + c7[r];
                    }
            }
        } else if (ZP && p > first) {
            // the previous block's last R rows, complete now (also those of the block a run
            // starts early with: the run before this one leaves them to us)
            if (!bad && !closing && o - S + L >= n0) {
                const __amdgpu_buffer_rsrc_t ry = buf_rsrc(yr + (o - S + L));
#pragma unroll
                for (int r = 0; r < RM; ++r)
#ifdef OSZ_ABL_NOSTORE
                    if (r < R && held[r] + c7[r] == 1.2345e300) buf_store(held[r] + c7[r], ry, lane8, 2048u * (NB - 1 - r));
#else
                    if (r < R) buf_store(held[r] + c7[r], ry, lane8, 2048u * (NB - 1 - r));
#endif
                if (nst >= 0) nst += R;
            } else {
                nst = -1;
                const int64_t ob = o - S + tt;
#pragma unroll
                for (int r = 0; r < RM; ++r)
                    if (r < R) OSZ_ZP_PUT(ob + 256 * (NB - 1 - r), bad ? qn : held[r] + c7[r]);
            }
        }
        if (closing) {
            // the chunk's last block: la samples of output, everything behind them to the carry
            // (window rows up to 31, then this block's own +mu behind the window)
            const int la = g.la;
            double *co = g.carry_out + (int64_t)c * kSpecLdc;
            for (int i = tt; i < kSpecLdc; i += 256) co[i] = bad ? qn : 0.0;
            __syncthreads();           // (the zeros and the values below come from different threads)
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const int pp = 256 * j + tt;
                const double v = j < 16 ? re[j & 15] : im[j & 15];
                if (pp < la) OSZ_ZP_PUT(o + pp, bad ? qn : v);
                else if (!bad && pp - la < kSpecLdc) co[pp - la] = v;
            }
            if (!bad) {
                double Pr[NS], Pi[NS];
                zpn_powers<NM, NS>(ptab, 0, tt, Pr, Pi);
                for (int r = 0; r < Rf; ++r) {
                    const int k = 8192 + 256 * r + tt - la;
                    if (k < kSpecLdc) co[k] = zp_dot<NS>(kmu + (r * NS) * 2, Pr, Pi);
                }
            }
            if (g.hist) {
                double *hr = g.hist + (int64_t)c * g.hist_len;
                const double *src = xr + n - g.hist_len;
                for (int i = tt; i < g.hist_len; i += 256) hr[i] = src[i];
            }
        } else {
            if (own) {
                if (!bad && !edge && o + L >= n0) {
                    // the common case: whole rows into the current output
                    const __amdgpu_buffer_rsrc_t ry = buf_rsrc(yr + (o + L));
#ifdef OSZ_ABL_NOSTORE     // (the values stay live through a store that never happens)
                    double acc_ = 0.0;
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        acc_ += re[j];
                        if (j < NHI - R) acc_ += im[j];
                    }
                    if (acc_ == 1.2345e300) buf_store(acc_, ry, lane8, 0);
#else
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        if (RM <= NHI || j < NB - R) buf_store(re[j], ry, lane8, 2048u * j);   // (R <= RM <= NHI: every lower row)
                        if (j < NHI - R) buf_store(im[j], ry, lane8, 2048u * (j + 16));
                    }
#endif
                    if (nst >= 0) nst += NB - R;
                } else {
                    nst = -1;
                    int64_t off = o + tt;
                    asm volatile("" : "+v"(off));
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        if (RM <= NHI || j < NB - R) OSZ_ZP_PUT(off + 256 * j, bad ? qn : re[j]);
                        if (j < NHI - R) OSZ_ZP_PUT(off + 256 * (j + 16), bad ? qn : im[j]);
                    }
                }
            }
            // (row NB - 1 - r: in the window's upper half as long as r < NHI, below it for the long
            // left tails of blocks of 20 ... 23 rows)
#pragma unroll
            for (int r = 0; r < RM; ++r)
                if (r < R) held[r] = r < NHI ? im[(NHI - 1 - r) & 15] : re[(NB - 1 - r) & 15];
#pragma unroll
            for (int j = 0; j < D; ++j) cr[j] = im[NHI + j];
        }
        par ^= 1;
        younger = nst;
    };
#ifdef OSZ_NEGA_STAMPS
    for (int q = 0; q < 24; ++q) P.sacc[q] = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(P.slast)::"memory");
#endif
    for (int p = first; p <= lastf; ++p) block(p, std::false_type{});
#ifdef OSZ_NEGA_STAMPS
    if (g_nega_stamps && (t & 63) == 0) {
        unsigned long long *so = g_nega_stamps + (((int64_t)c * g.nruns + run) * 4 + (t >> 6)) * 24;
        for (int q = 0; q < 24; ++q) so[q] = P.sacc[q];
        so[23] = (unsigned long long)(lastf - first + 1);
    }
#endif
    if (closes) block(lastf + 1, std::true_type{});
#ifdef OSZ_CLK
    if (blockIdx.x == 0 && blockIdx.y == 7) {
        unsigned long long clk1, rt1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk1), "=s"(rt1)::"memory");
        if (threadIdx.x == 0) {
            g_zpn_clk[0] = clk1 - clk0;
            g_zpn_clk[1] = rt1 - rt0;
        }
    }
#endif
#undef OSZ_ZP_PUT
    // where the forward stream of this channel first went bad (chain_zp.hip: later launches start
    // bad, osz_chain_zp_seal settles the chunks the reference loses)
    if (ZP && bad && bad_at >= 0 && t == 0) atomicMin(reinterpret_cast<long long *>(g.nanpos + c), g.pos + bad_at);
}

// (one translation unit per mode count: chain_zpn_{2,4,6,8}.hip define OSZ_ZPN_NM and include
// this file; the dispatcher lives with NM = 2.  OSZ_ZPN_NO_DISPATCH: the kernel template alone, for
// the diagnostic builds of benchmarks/ that instantiate one instance)
#ifndef OSZ_ZPN_NO_DISPATCH
template <int NM, int NS>
static zp_kern_t zpn_kernel_nb(int nb, int r) {
    static const zp_kern_t k[7] = {chain_zpn_kernel<24, NM, NS>, chain_zpn_kernel<25, NM, NS>,
                                   chain_zpn_kernel<26, NM, NS>, chain_zpn_kernel<27, NM, NS>,
                                   chain_zpn_kernel<28, NM, NS>, chain_zpn_kernel<29, NM, NS>,
                                   chain_zpn_kernel<30, NM, NS>};
    // more than five burst rows: blocks of 24 ... 26 rows (R <= 32 - NB); more than eight: 20 ... 23
    static const zp_kern_t k8[3] = {chain_zpn_kernel<24, NM, NS, 8>, chain_zpn_kernel<25, NM, NS, 8>,
                                    chain_zpn_kernel<26, NM, NS, 8>};
    static const zp_kern_t k12[4] = {chain_zpn_kernel<20, NM, NS, 12>, chain_zpn_kernel<21, NM, NS, 12>,
                                     chain_zpn_kernel<22, NM, NS, 12>, chain_zpn_kernel<23, NM, NS, 12>};
    if (nb < 24) return r <= 12 ? k12[nb - 20] : nullptr;
    if (r > kSpecRMax) return nb <= 26 && r <= 8 ? k8[nb - 24] : nullptr;
    return k[nb - 24];
}

#define OSZ_ZPN_CAT2(a, b) a##b
#define OSZ_ZPN_CAT(a, b) OSZ_ZPN_CAT2(a, b)
// zpn_kernel_nm2 / 4 / 6 / 8 (nb, ns, r)
zp_kern_t OSZ_ZPN_CAT(zpn_kernel_nm, OSZ_ZPN_NM)(int nb, int ns, int r) {
    constexpr int NM = OSZ_ZPN_NM;
    if (nb < 20 || nb > 30) return nullptr;
    if (ns == 2) return zpn_kernel_nb<NM, 2>(nb, r);
    if (NM >= 4 && ns == 4) return zpn_kernel_nb<NM, (NM >= 4 ? 4 : 2)>(nb, r);
    if (NM >= 6 && ns == 6) return zpn_kernel_nb<NM, (NM >= 6 ? 6 : 2)>(nb, r);
    return nullptr;
}

// the forward chain's instances (ZP = false, five burst rows): zpn_fwd_kernel_nm2 / 4 / 6 / 8 (nb, ns)
template <int NM, int NS>
static zp_kern_t zpn_fwd_kernel_nb(int nb) {
    static const zp_kern_t k[7] = {
        chain_zpn_kernel<24, NM, NS, kSpecRMax, false>, chain_zpn_kernel<25, NM, NS, kSpecRMax, false>,
        chain_zpn_kernel<26, NM, NS, kSpecRMax, false>, chain_zpn_kernel<27, NM, NS, kSpecRMax, false>,
        chain_zpn_kernel<28, NM, NS, kSpecRMax, false>, chain_zpn_kernel<29, NM, NS, kSpecRMax, false>,
        chain_zpn_kernel<30, NM, NS, kSpecRMax, false>};
    return k[nb - 24];
}
zp_kern_t OSZ_ZPN_CAT(zpn_fwd_kernel_nm, OSZ_ZPN_NM)(int nb, int ns) {
    constexpr int NM = OSZ_ZPN_NM;
    if (nb < 24 || nb > 30) return nullptr;
    if (ns == 2) return zpn_fwd_kernel_nb<NM, 2>(nb);
    if (NM >= 4 && ns == 4) return zpn_fwd_kernel_nb<NM, (NM >= 4 ? 4 : 2)>(nb);
    if (NM >= 6 && ns == 6) return zpn_fwd_kernel_nb<NM, (NM >= 6 ? 6 : 2)>(nb);
    return nullptr;
}

#if OSZ_ZPN_NM == 2
zp_kern_t zpn_fwd_kernel_nm4(int nb, int ns);
zp_kern_t zpn_fwd_kernel_nm6(int nb, int ns);
zp_kern_t zpn_fwd_kernel_nm8(int nb, int ns);
zp_kern_t zpn_fwd_kernel_for(int nb, int nm, int ns) {
    return nm == 2 ? zpn_fwd_kernel_nm2(nb, ns) : nm == 4 ? zpn_fwd_kernel_nm4(nb, ns) : nm == 6 ? zpn_fwd_kernel_nm6(nb, ns)
           : nm == 8 ? zpn_fwd_kernel_nm8(nb, ns) : nullptr;
}
zp_kern_t zpn_kernel_nm4(int nb, int ns, int r);
zp_kern_t zpn_kernel_nm6(int nb, int ns, int r);
zp_kern_t zpn_kernel_nm8(int nb, int ns, int r);
zp_kern_t zpn_kernel_for(int nb, int nm, int ns, int r) {
    return nm == 2 ? zpn_kernel_nm2(nb, ns, r) : nm == 4 ? zpn_kernel_nm4(nb, ns, r) : nm == 6 ? zpn_kernel_nm6(nb, ns, r)
           : nm == 8 ? zpn_kernel_nm8(nb, ns, r) : nullptr;
}
#endif
#endif  // OSZ_ZPN_NO_DISPATCH

}  // namespace osz
