// chain_spec.hip -- K1+K2 fused IN THE SPECTRUM: the overlap-add FIR and the
// forward pass of a biquad cascade as ONE multiplication per bin.
//
// The reference chains oaconvolve (core/numerical.py:158-298) into sosfilt /
// the forward half of sosfiltfilt (:301-335, :374-386).  Both are linear and
// time invariant, so a zero-padded block x_b of S samples answers with
//     y_lin = x_b * (h * g),     g = impulse response of the cascade (infinite),
// and the 4096-point transform the FIR kernel already runs gives, with the
// spectrum  Hc[k] = H_fir[k] * prod_s B_s(w_k) / A_s(w_k)  in place of H_fir,
// the CIRCULAR answer  y_circ[n] = sum_m y_lin[n + 4096 m].  What wraps around
// is the cascade's ringing after the block's FIR output has ended (window
// sample n0 = S + wlen - 1): a solution of the homogeneous recurrence, i.e. a
// combination of the cascade's modes  Re(c_q lambda_q^n)  (lambda_q: one pole per
// conjugate pair, or a real pole).  Row 15 of the window (samples 3840..4095)
// is past n0 by construction (S = 256 NR with S + wlen - 1 <= 3840), holds
// nothing but that ringing, and its first 64 samples determine the mode
// amplitudes by a fixed least-squares matrix (host, long double).  With
//     mu_q = amplitude of mode q extrapolated to window sample 4096
// the linear answer is recovered exactly (to rounding):
//   * inside the window   y_lin[n] = y_circ[n] - Re sum_q mu_q lambda_q^n,
//   * behind the window   y_lin[4096 + n] =      + Re sum_q mu_q lambda_q^n,
// the second being the first delayed by one window: per block one burst of
// -mu at its first sample and one of +mu 4096 samples later, both evaluated
// only over the R rows of 256 samples in which they exceed 1e-18 of the
// output scale (R = 3 for the Butterworth band-pass of the headline
// configuration: the bursts are ~1e-8 of the signal to begin with).  The
// windows themselves overlap-add exactly as the FIR's do (registers, rows
// j >= NR of a block meet rows j - NR of the next).
//
// What this replaces: the time-parallel recurrence of sos_tile.h inside
// chain_kernel (chain.hip) -- 42 of its 149 vector instructions per sample,
// two LDS transpositions and nine of its fourteen barriers per pair.  Here a
// pair costs the FIR's transform plus ~12 FMAs per sample on 4 R of its 2 NR
// rows and three short barriers.  chain_kernel stays for cascades this scheme
// does not take (repeated poles, ringing longer than the guard rows, filters
// longer than 1793 taps).
//
// Carried state.  Between chunks the kernel carries ONE sequence per channel:
// `carry`[i] = what output sample (chunk end + i) would be if the input
// stopped -- window tails and pending bursts, already evaluated.  The next
// chunk adds it to its first rows.  The handles' own states (FIR overlap
// tail, DF2T section states) are not maintained by this kernel; they are
// rebuilt on demand (spec_settle) by replaying the last wlen - 1 + warm_len
// input samples, which the link keeps, through the plain kernels, and turned
// into a carry (spec_import) by filtering the pending FIR tail.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <functional>
#include <vector>

#include "common.h"
#include "fft4096.h"
#include "fir_pair.h"
#include "handles.h"
#include "sos_tile.h"
#include "chain_spec.h"
#include "chain_zp.h"

namespace osz {

struct SpecArgs {
    FirArgs f;                 // x, ldx, y, ldy, wlen, step, H (composite), tb
    int64_t n;                 // samples of this chunk
    int64_t W;                 // whole pairs on the fast path
    int nruns;
    int la, lb;                // lengths of the closing pair's two blocks
    int R;                     // burst rows
    const double *M;           // [2 NM][64]: mu = M y (re rows, then im rows)
    const double *P;           // [32][NM][2]: lambda^(16 i), i < 16, then lambda^i, i < 16
    const double *Lrow;        // [R][NM][2]: lambda^(256 r)
    const double *carry_in;    // (nch, kSpecLdc)
    double *carry_out;
    double *hist;              // (nch, hist_len): the chunk's last hist_len input samples, or null
    int hist_len;
};

// A workgroup walks a run of whole pairs of blocks of one channel (fast path);
// run 0 opens the chunk (adds the carried sequence to its first pair), later
// runs start one pair early from nothing and discard that pair's outputs (a
// pair is all the past a pair depends on: D < NR tail rows and the previous
// block's mu); the last run closes the chunk: one more pair, of any length,
// through a generic path that also writes the carry.
template <int NR, int NM, int HP>
__global__ __launch_bounds__(256, 2) void chain_spec_kernel(SpecArgs g) {
    constexpr int D = 16 - NR, S = 256 * NR;
    extern __shared__ fft::cube::C2 cube_lds[];
    double *xl = reinterpret_cast<double *>(cube_lds) + 2 * fft::cube::SLOTS;   // behind the cube
    double *fitbuf = xl;                              // [2 blk][64]
    double *mu = fitbuf + 2 * kSpecFit;               // [2 parity][2 blk][NM][2]
    double *kap = mu + 2 * 2 * NM * 2;                // [5 src][kSpecRMax][NM][2]
    double *lrow = kap + 5 * kSpecRMax * NM * 2;      // [kSpecRMax][NM][2]
    double *ptab = lrow + kSpecRMax * NM * 2;         // [32][NM][2]: lambda^(16 i), i < 16, then lambda^i
    double *mtab = ptab + 32 * NM * 2;                // [2 NM][64]
    const FirArgs &a = g.f;
    const int t = threadIdx.x, run = blockIdx.x, c = blockIdx.y;
    const int R = g.R;
    const double *xr = a.x + (int64_t)c * a.ldx;
    double *yr = a.y + (int64_t)c * a.ldy;
    const int64_t p0 = ((int64_t)run * g.W) / g.nruns;
    const int64_t p1 = ((int64_t)(run + 1) * g.W) / g.nruns;
    const int64_t ps = run == 0 ? 0 : p0 - 1;

    FirPair<NR, HP, 0, true> P{a, t, a.wlen - 1, xr, yr, 0, cube_lds};
    fft::cube2::tw_load(t, a.tb, P.tw1, P.tw2);
    if (HP < 0) {
        // the composite spectrum resident in registers (else: requested per pair, FirPair::transform)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = 256 * r + t;          // the spectrum is stored in cube2's bin order
            P.Hr[HP < 0 ? r : 0] = a.H[2 * k];
            P.Hi[HP < 0 ? r : 0] = a.H[2 * k + 1];
        }
    }
#pragma unroll
    for (int j = 0; j < D; ++j) P.cr[j] = 0.0;
    // 65536 / (R NM), rounded up: t / (R NM) for t < 256 by multiplication
    const int kinv = (65536 + R * NM - 1) / (R * NM);
    for (int i = t; i < R * NM * 2; i += 256) lrow[i] = g.Lrow[i];
    for (int i = t; i < 32 * NM * 2; i += 256) ptab[i] = g.P[i];
    for (int i = t; i < 2 * NM * kSpecFit; i += 256) mtab[i] = g.M[i];
    if (t < 2 * 2 * NM * 2) mu[t] = 0.0;
    // `bad` is uniform (a scalar) and sticky: non-finite samples are everywhere behind the
    // transform, every amplitude of the fit and with it every lane's burst values see them
    bool bad = false;
    int par = 0;
    const unsigned lane8 = 8u * (unsigned)t;   // a lane's byte offset inside a row of 256 samples
    __syncthreads();

    for (int64_t p = ps; p < p1; ++p) {
        const int64_t o = p * (2 * S);
        double re[16], im[16];
        {
            // rows of the pair: base and row offsets in scalar registers (buf_rsrc, common.h)
            const __amdgpu_buffer_rsrc_t rx = buf_rsrc(xr + o);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                re[j] = j < NR ? buf_load(rx, lane8, 2048u * j) : 0.0;
                im[j] = j < NR ? buf_load(rx, lane8, 2048u * (j + NR)) : 0.0;
            }
        }
        P.transform(re, im);
        // thread -> role indices, recomputed per pair from an opaque copy of t (kept across
        // the transform they cost registers the transform does not have)
        int tt = t;
        asm volatile("" : "+v"(tt));
        // fit: 8 consecutive lanes share one row of M (one block, one component of one
        // mode), each takes 8 of the 64 samples
        const int fg = tt >> 3, p8 = tt & 7;
        const bool fvalid = fg < 4 * NM;
        const int fblk = fvalid ? fg / (2 * NM) : 0, frow = fvalid ? fg % (2 * NM) : 0;
        const int fmu = (fblk * NM + frow % NM) * 2 + frow / NM;      // + parity * 2 * NM * 2
        // kappa stage: thread -> (source, burst row, mode)
        const int ksrc = (tt * kinv) >> 16, kr_ = (tt / NM) % R, kq = tt % NM;
        if (tt < 64) {
            fitbuf[tt] = re[15];
            fitbuf[kSpecFit + tt] = im[15];
        }
        // overlap add in registers (FirPair::fast_pair)
#pragma unroll
        for (int j = 0; j < D; ++j) {
            re[j] += P.cr[j];
            im[j] += re[j + NR];
            P.cr[j] = im[j + NR];
        }
        __syncthreads();
        {   // fit
            const double *yb = fitbuf + kSpecFit * fblk + 8 * p8;
            const double *mc = mtab + kSpecFit * frow + 8 * p8;
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) s = fma(mc[k], yb[k], s);
            if (!fvalid) s = 0.0;
            s += dpp_row_shr0<1>(s);
            s += dpp_row_shr0<2>(s);
            s += dpp_row_shr0<4>(s);
            if (p8 == 7 && fvalid) mu[par * (2 * NM * 2) + fmu] = s;
        }
        __syncthreads();
        // kappa[src][r][q] = (+-) mu_src lambda^(256 r):
        //   src 0: block a, its own wrap (-mu_a, rows 0..R)    1: block a, the previous block b (+, rows D..)
        //   src 2: block b, its own wrap (-mu_b)               3: block b, block a one window on (+mu_a)
        if (ksrc < 4) {
            const int mpar = ksrc == 1 ? par ^ 1 : par;
            const int mblk = (ksrc == 1 || ksrc == 2) ? 1 : 0;
            const double sg = (ksrc & 1) ? 1.0 : -1.0;
            const double mr = mu[((mpar * 2 + mblk) * NM + kq) * 2 + 0];
            const double mi = mu[((mpar * 2 + mblk) * NM + kq) * 2 + 1];
            const double lr = lrow[(kr_ * NM + kq) * 2 + 0], li = lrow[(kr_ * NM + kq) * 2 + 1];
            kap[((ksrc * kSpecRMax + kr_) * NM + kq) * 2 + 0] = sg * (mr * lr - mi * li);
            kap[((ksrc * kSpecRMax + kr_) * NM + kq) * 2 + 1] = sg * (mr * li + mi * lr);
        }
        __syncthreads();
        // lambda_q^t = lambda_q^(16 (t >> 4)) lambda_q^(t & 15), formed when needed: held
        // across the transform they would spill
        double Pre[NM], Pim[NM];
        {
            const double *ph = ptab + ((tt >> 4) * NM) * 2, *pl = ptab + ((16 + (tt & 15)) * NM) * 2;
#pragma unroll
            for (int q = 0; q < NM; ++q) {
                Pre[q] = ph[2 * q] * pl[2 * q] - ph[2 * q + 1] * pl[2 * q + 1];
                Pim[q] = ph[2 * q] * pl[2 * q + 1] + ph[2 * q + 1] * pl[2 * q];
            }
        }
#pragma unroll
        for (int r = 0; r < kSpecRMax; ++r) {
            if (r < R) {
                double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
#pragma unroll
                for (int q = 0; q < NM; ++q) {
                    const double *k0 = kap + ((0 * kSpecRMax + r) * NM + q) * 2;
                    const double *k1 = kap + ((1 * kSpecRMax + r) * NM + q) * 2;
                    const double *k2 = kap + ((2 * kSpecRMax + r) * NM + q) * 2;
                    const double *k3 = kap + ((3 * kSpecRMax + r) * NM + q) * 2;
                    c0 = fma(k0[0], Pre[q], fma(-k0[1], Pim[q], c0));
                    c1 = fma(k1[0], Pre[q], fma(-k1[1], Pim[q], c1));
                    c2 = fma(k2[0], Pre[q], fma(-k2[1], Pim[q], c2));
                    c3 = fma(k3[0], Pre[q], fma(-k3[1], Pim[q], c3));
                }
                re[r] += c0;
                re[(D + r) & 15] += c1;
                im[r] += c2;
                im[(D + r) & 15] += c3;
                if (r == 0 && !bad) bad = __builtin_amdgcn_readfirstlane((int)(sos_not_finite(c0) || sos_not_finite(c2))) != 0;
            }
        }
        if (run == 0 && p == 0) {
            // the chunk opens: what the stream so far still owes these samples
            const double *ci = g.carry_in + (int64_t)c * kSpecLdc + tt;
            // (a poisoned stream: NaN all the way to where a carry has long died)
            if (!bad)
                bad = __builtin_amdgcn_readfirstlane((int)sos_not_finite(g.carry_in[(int64_t)c * kSpecLdc + 4095 + 256 * R])) != 0;
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                re[j] += ci[256 * j];
                im[j] += ci[S + 256 * j];
            }
        }
        if (p >= p0) {
            const __amdgpu_buffer_rsrc_t ry = buf_rsrc(yr + o);
            if (!bad) {
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    buf_store(re[j], ry, lane8, 2048u * j);
                    buf_store(im[j], ry, lane8, 2048u * (j + NR));
                }
            } else {
                const double qn = spec_qnan();
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    buf_store(qn, ry, lane8, 2048u * j);
                    buf_store(qn, ry, lane8, 2048u * (j + NR));
                }
            }
        }
        par ^= 1;
    }

    if (run == g.nruns - 1) {
        // ---- the closing pair: blocks of la and lb samples (lb may be 0), everything
        // accumulated in LDS over the idle cube as acc[i], i = samples from the pair's start
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
        // thread -> role indices, recomputed per pair from an opaque copy of t (kept across
        // the transform they cost registers the transform does not have)
        int tt = t;
        asm volatile("" : "+v"(tt));
        // fit: 8 consecutive lanes share one row of M (one block, one component of one
        // mode), each takes 8 of the 64 samples
        const int fg = tt >> 3, p8 = tt & 7;
        const bool fvalid = fg < 4 * NM;
        const int fblk = fvalid ? fg / (2 * NM) : 0, frow = fvalid ? fg % (2 * NM) : 0;
        const int fmu = (fblk * NM + frow % NM) * 2 + frow / NM;      // + parity * 2 * NM * 2
        // kappa stage: thread -> (source, burst row, mode)
        const int ksrc = (tt * kinv) >> 16, kr_ = (tt / NM) % R, kq = tt % NM;
        if (tt < 64) {
            fitbuf[tt] = re[15];
            fitbuf[kSpecFit + tt] = im[15];
        }
        __syncthreads();     // also: every thread is done reading the cube
        {
            const double *yb = fitbuf + kSpecFit * fblk + 8 * p8;
            const double *mc = mtab + kSpecFit * frow + 8 * p8;
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) s = fma(mc[k], yb[k], s);
            if (!fvalid) s = 0.0;
            s += dpp_row_shr0<1>(s);
            s += dpp_row_shr0<2>(s);
            s += dpp_row_shr0<4>(s);
            if (p8 == 7 && fvalid) mu[par * (2 * NM * 2) + fmu] = s;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            acc[256 * j + t] = re[j] + (j < D ? P.cr[j < D ? j : 0] : 0.0);
            acc[4096 + 256 * j + t] = 0.0;
        }
        __syncthreads();
        // five sources: the previous block b one window on (offset 256 D), -mu_a (0),
        // -mu_b (la), +mu_a (4096), +mu_b (la + 4096)
        if (ksrc < 5) {
            const int mpar = ksrc == 0 ? par ^ 1 : par;
            const int mblk = (ksrc == 0 || ksrc == 2 || ksrc == 4) ? 1 : 0;
            const double sg = (ksrc == 1 || ksrc == 2) ? -1.0 : 1.0;
            const double mr = mu[((mpar * 2 + mblk) * NM + kq) * 2 + 0];
            const double mi = mu[((mpar * 2 + mblk) * NM + kq) * 2 + 1];
            const double lr = lrow[(kr_ * NM + kq) * 2 + 0], li = lrow[(kr_ * NM + kq) * 2 + 1];
            kap[((ksrc * kSpecRMax + kr_) * NM + kq) * 2 + 0] = sg * (mr * lr - mi * li);
            kap[((ksrc * kSpecRMax + kr_) * NM + kq) * 2 + 1] = sg * (mr * li + mi * lr);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[la + 256 * j + t] += im[j];
        {
            // non-finite amplitudes poison the rest of the stream
            bool nf = false;
            for (int i = 0; i < 2 * NM * 2; ++i) nf = nf || sos_not_finite(mu[par * (2 * NM * 2) + i]);
            if (!bad) bad = __builtin_amdgcn_readfirstlane((int)nf) != 0;
        }
        __syncthreads();
        // lambda_q^t = lambda_q^(16 (t >> 4)) lambda_q^(t & 15), formed when needed: held
        // across the transform they would spill
        double Pre[NM], Pim[NM];
        {
            const double *ph = ptab + ((tt >> 4) * NM) * 2, *pl = ptab + ((16 + (tt & 15)) * NM) * 2;
#pragma unroll
            for (int q = 0; q < NM; ++q) {
                Pre[q] = ph[2 * q] * pl[2 * q] - ph[2 * q + 1] * pl[2 * q + 1];
                Pim[q] = ph[2 * q] * pl[2 * q + 1] + ph[2 * q + 1] * pl[2 * q];
            }
        }
        for (int s = 0; s < 5; ++s) {
            const int off = s == 0 ? 256 * D : s == 1 ? 0 : s == 2 ? la : s == 3 ? 4096 : la + 4096;
            for (int r = 0; r < R; ++r) {
                double cs = 0.0;
#pragma unroll
                for (int q = 0; q < NM; ++q) {
                    const double *kk = kap + ((s * kSpecRMax + r) * NM + q) * 2;
                    cs = fma(kk[0], Pre[q], fma(-kk[1], Pim[q], cs));
                }
                const int i = off + 256 * r + t;
                if (i < 8192) acc[i] += cs;
            }
            __syncthreads();
        }
        const double qn = spec_qnan();
        const int ltot = la + lb;
        for (int i = t; i < ltot; i += 256) yr[o + i] = bad ? qn : acc[i];
        double *co = g.carry_out + (int64_t)c * kSpecLdc;
        for (int i = t; i < kSpecLdc; i += 256) {
            const int src = ltot + i;
            co[i] = bad ? qn : (src < 8192 ? acc[src] : 0.0);
        }
        if (g.hist) {
            // the input samples a later spec_settle replays
            double *hr = g.hist + (int64_t)c * g.hist_len;
            const double *src = xr + g.n - g.hist_len;
            for (int i = t; i < g.hist_len; i += 256) hr[i] = src[i];
        }
    }
}

// A NaN never leaves the cascade (sos_tile.h, "NaN reach"); the replay of spec_settle
// knows nothing of one that lies before the samples it replays: the section states of
// a channel whose carry is poisoned become NaN.
__global__ void spec_poison_kernel(const double *carry, int probe, double *state, int nsec, int nch) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nch) return;
    if (!sos_not_finite(carry[(int64_t)c * kSpecLdc + probe])) return;
    const double qn = spec_qnan();
    for (int q = 0; q < nsec; ++q) {
        state[((int64_t)q * nch + c) * 2 + 0] = qn;
        state[((int64_t)q * nch + c) * 2 + 1] = qn;
    }
}

// ---------------------------------------------------------------- host side
// The link between a FIR handle and a SOS handle that run this kernel together.
struct ChainSpec {
    osz_fir_s *fir = nullptr;
    osz_sos_s *sos = nullptr;
    bool eligible = false;
    int NR = 0, NM = 0, R = 0;
    double *dH = nullptr, *dM = nullptr, *dP = nullptr, *dL = nullptr;
    double *dcarry[2] = {nullptr, nullptr};
    int cur = 0;
    bool carry_valid = false;      // dcarry[cur] describes the stream
    bool true_valid = true;        // the handles' own states describe the stream
    // what spec_settle replays
    int hist_cap = 0;              // wlen - 1 + warm_len
    double *dhist[2] = {nullptr, nullptr};
    int hcur = 0;
    int64_t hist_n = 0, since_import = 0;
    double *dsnap_fir = nullptr, *dsnap_sos = nullptr;
    double *dscratch = nullptr;    // (nch, hist_cap)
    // one real block per transform (chain_zpn_body.h, ZP = false; tables of spec::build_specn): NR is
    // then its rows per block, R its burst rows; dT = L | P | M as the kernel's LDS holds them
    bool nega = false;
    int NS = 0, nh = 0;
    double *dT = nullptr;
};

static void spec_free(ChainSpec *s) {
    (void)hipFree(s->dH);
    (void)hipFree(s->dM);
    (void)hipFree(s->dP);
    (void)hipFree(s->dL);
    for (int q = 0; q < 2; ++q) {
        (void)hipFree(s->dcarry[q]);
        (void)hipFree(s->dhist[q]);
    }
    (void)hipFree(s->dsnap_fir);
    (void)hipFree(s->dsnap_sos);
    (void)hipFree(s->dscratch);
    (void)hipFree(s->dT);
    delete s;
}

// a handle goes away (or is paired anew): the other side forgets the link
void spec_unlink(ChainSpec *s) {
    if (!s) return;
    if (s->fir) s->fir->spec = nullptr;
    if (s->sos) s->sos->spec = nullptr;
    spec_free(s);
}

// tables of the scheme for (taps, cascade): spec_tables.h; s->eligible says whether it applies
static int spec_build(ChainSpec *s) {
    osz_fir_s *fir = s->fir;
    osz_sos_s *sos = s->sos;
    s->eligible = false;
    if (fir->parts.size() != 1 || fir->nch != sos->nch) return OSZ_OK;
    const int wlen = fir->ntaps, nsec = sos->nsec;
    auto up = [](double **d, const std::vector<double> &v) -> int {
        OSZ_HIP(hipMalloc(d, v.size() * sizeof(double)));
        OSZ_HIP(hipMemcpy(*d, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice));
        return OSZ_OK;
    };
    int rcu;
    // where the right tail's bursts are cut: the library's default unless the caller set one
    // (osz_chain_zp_tolerance names both links of a cascade)
    const spec::ld_t tol = sos->zp_tol > 0.0 ? (spec::ld_t)sos->zp_tol : spec::kTailTol;
    // one real block per transform where its tables exist (OSZ_SPEC_NEGA=0: the pair kernel below)
    static const bool nega_on = [] {
        const char *e = getenv("OSZ_SPEC_NEGA");
        return !(e && e[0] == '0');
    }();
    if (nega_on) {
        const bool forgets = sos->warm_len <= (1 << 20);
        const spec::TablesZp Tn = spec::kept_tables(spec::kKeptSpecn, fir->htaps, sos->coef, nsec, (double)tol, forgets, [&] {
            return spec::build_specn(fir->htaps.data(), wlen, sos->coef, nsec, forgets, 15360 - 1024, tol);
        });
        if (Tn.eligible) {
            const size_t nl = (size_t)Tn.NM * 2;               // lambda^256: row 1 of the table
            std::vector<double> cat(Tn.L.begin() + nl, Tn.L.begin() + 2 * nl);
            cat.insert(cat.end(), Tn.P.begin(), Tn.P.end());
            cat.insert(cat.end(), Tn.M.begin(), Tn.M.end());
            if ((rcu = up(&s->dH, spec_permuted_spectrum(Tn.H))) || (rcu = up(&s->dT, cat))) return rcu;
            const size_t cbn = sizeof(double) * (size_t)fir->nch * kSpecLdc;
            for (int q = 0; q < 2; ++q) {
                OSZ_HIP(hipMalloc(&s->dcarry[q], cbn));
                OSZ_HIP(hipMemset(s->dcarry[q], 0, cbn));
            }
            s->hist_cap = (wlen - 1) + (int)sos->warm_len;
            for (int q = 0; q < 2; ++q)
                OSZ_HIP(hipMalloc(&s->dhist[q], sizeof(double) * (size_t)fir->nch * s->hist_cap));
            OSZ_HIP(hipMalloc(&s->dsnap_fir, sizeof(double) * (size_t)fir->nch * (wlen - 1)));
            OSZ_HIP(hipMalloc(&s->dsnap_sos, sizeof(double) * (size_t)nsec * fir->nch * 2));
            s->NR = Tn.NR;
            s->NM = Tn.NM;
            s->NS = Tn.NS;
            s->R = Tn.R;
            s->nh = Tn.nh;
            s->nega = true;
            s->eligible = true;
            return OSZ_OK;
        }
    }
    const spec::Tables T = spec::build(fir->htaps.data(), wlen, sos->coef, nsec, sos->warm_len <= (1 << 20), tol);
    if (!T.eligible) return OSZ_OK;
    if ((rcu = up(&s->dH, spec_permuted_spectrum(T.H))) || (rcu = up(&s->dM, T.M)) || (rcu = up(&s->dP, T.P)) || (rcu = up(&s->dL, T.L)))
        return rcu;
    const size_t cb = sizeof(double) * (size_t)fir->nch * kSpecLdc;
    for (int q = 0; q < 2; ++q) {
        OSZ_HIP(hipMalloc(&s->dcarry[q], cb));
        OSZ_HIP(hipMemset(s->dcarry[q], 0, cb));
    }
    s->hist_cap = (wlen - 1) + (int)sos->warm_len;
    for (int q = 0; q < 2; ++q)
        OSZ_HIP(hipMalloc(&s->dhist[q], sizeof(double) * (size_t)fir->nch * s->hist_cap));
    OSZ_HIP(hipMalloc(&s->dsnap_fir, sizeof(double) * (size_t)fir->nch * (wlen - 1)));
    OSZ_HIP(hipMalloc(&s->dsnap_sos, sizeof(double) * (size_t)nsec * fir->nch * 2));
    s->NR = T.NR;
    s->NM = T.NM;
    s->R = T.R;
    s->eligible = true;
    return OSZ_OK;
}

// the handles' own states -> the carry: the pending FIR outputs go through the
// cascade from its current state, and on until the ringing has died
static int spec_import(ChainSpec *s, hipStream_t st) {
    osz_fir_s *fir = s->fir;
    osz_sos_s *sos = s->sos;
    FirPart &pt = fir->parts[0];
    const int wm1 = pt.ntaps - 1, nch = fir->nch;
    double *cb = s->dcarry[s->cur];
    OSZ_HIP(hipMemsetAsync(cb, 0, sizeof(double) * (size_t)nch * kSpecLdc, st));
    OSZ_HIP(hipMemcpy2DAsync(cb, sizeof(double) * kSpecLdc, pt.dstate[pt.cur], sizeof(double) * wm1,
                             sizeof(double) * wm1, nch, hipMemcpyDeviceToDevice, st));
    OSZ_HIP(hipMemcpyAsync(s->dsnap_fir, pt.dstate[pt.cur], sizeof(double) * (size_t)nch * wm1,
                           hipMemcpyDeviceToDevice, st));
    OSZ_HIP(hipMemcpyAsync(s->dsnap_sos, sos->dstate, sizeof(double) * (size_t)sos->nsec * nch * 2,
                           hipMemcpyDeviceToDevice, st));
    const int cl = 4096 + 256 * s->R;
    int rc = sos_forward_raw(sos, cb, kSpecLdc, cb, kSpecLdc, cl, st);
    if (rc) return rc;
    s->hist_n = 0;
    s->since_import = 0;
    s->carry_valid = true;
    s->true_valid = false;
    return OSZ_OK;
}

// the carry -> the handles' own states (FIR overlap tail, DF2T section states): the
// plain kernels over the input samples kept since the import, from the states
// saved there -- or, once more than hist_cap samples have gone by, over the last
// hist_cap of them from nothing (wlen - 1 samples fill the FIR tail, warm_len more
// and the cascade has forgotten its start: the bound of the time segments of sos.hip)
int spec_settle(ChainSpec *s, hipStream_t st) {
    if (!s || s->true_valid) return OSZ_OK;
    osz_fir_s *fir = s->fir;
    osz_sos_s *sos = s->sos;
    FirPart &pt = fir->parts[0];
    const int wm1 = pt.ntaps - 1, nch = fir->nch;
    const size_t fb = sizeof(double) * (size_t)nch * wm1, sb = sizeof(double) * (size_t)sos->nsec * nch * 2;
    if (s->since_import == s->hist_n) {
        OSZ_HIP(hipMemcpyAsync(pt.dstate[pt.cur], s->dsnap_fir, fb, hipMemcpyDeviceToDevice, st));
        OSZ_HIP(hipMemcpyAsync(sos->dstate, s->dsnap_sos, sb, hipMemcpyDeviceToDevice, st));
    } else {
        OSZ_HIP(hipMemsetAsync(pt.dstate[pt.cur], 0, fb, st));
        OSZ_HIP(hipMemsetAsync(sos->dstate, 0, sb, st));
    }
    if (s->hist_n > 0) {
        if (!s->dscratch)
            OSZ_HIP(hipMalloc(&s->dscratch, sizeof(double) * (size_t)nch * s->hist_cap));
        int rc = fir_push_raw(fir, s->dhist[s->hcur], s->hist_cap, s->hist_n, s->dscratch, s->hist_cap, 0, st);
        if (rc) return rc;
        rc = sos_forward_raw(sos, s->dscratch, s->hist_cap, s->dscratch, s->hist_cap, s->hist_n, st);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(spec_poison_kernel, dim3((nch + 255) / 256), dim3(256), 0, st, s->dcarry[s->cur],
                       4095 + 256 * s->R, sos->dstate, sos->nsec, nch);
    OSZ_HIP(hipGetLastError());
    s->true_valid = true;
    return OSZ_OK;
}

// the handles' own states are about to change under the link
int spec_touch(ChainSpec *s, hipStream_t st) {
    if (!s) return OSZ_OK;
    int rc = spec_settle(s, st);
    if (rc) return rc;
    s->carry_valid = false;
    return OSZ_OK;
}

using spec_kern_t = void (*)(SpecArgs);
template <int NM, int HP>
static spec_kern_t spec_kernel_for(int nr) {
    static const spec_kern_t k[8] = {
        chain_spec_kernel<8, NM, HP>,  chain_spec_kernel<9, NM, HP>,  chain_spec_kernel<10, NM, HP>,
        chain_spec_kernel<11, NM, HP>, chain_spec_kernel<12, NM, HP>, chain_spec_kernel<13, NM, HP>,
        chain_spec_kernel<14, NM, HP>, chain_spec_kernel<15, NM, HP>};
    return k[nr - 8];
}

// the same on one real block per transform: chain_zpn_kernel<NB, NM, NS, 5, false>
static int specn_forward(ChainSpec *s, const double *x, int64_t ldx, int64_t n, double *f, int64_t ldf,
                         hipStream_t st, const std::function<int()> &between, bool *taken) {
    osz_fir_s *fir = s->fir;
    const int NB = s->NR, S = 256 * NB;
    if (n < 2 * (int64_t)S) return OSZ_OK;                // a whole block and a closing one at least
    if (!s->carry_valid) {
        int rc = spec_import(s, st);
        if (rc) return rc;
    }
    {
        int rc = between();
        if (rc) return rc;
    }
    const int64_t W = (n - 1) / S;                        // whole blocks; the closing one has 1 .. S samples
    int64_t nruns = 512 / fir->nch;
    if (nruns > W) nruns = W;
    if (nruns < 1) nruns = 1;
    ZpArgs g{};
    g.f.x = x;
    g.f.y = f;
    g.f.ldx = ldx;
    g.f.ldy = ldf;
    g.f.n = n;
    g.f.skip = 0;
    g.f.wlen = fir->ntaps;
    g.f.step = S;
    g.f.H = s->dH;
    g.f.tb = fir->tb;
    g.n = n;
    g.W = W;
    g.nruns = (int)nruns;
    g.la = (int)(n - W * S);
    g.R = s->R;
    g.Rf = s->R;
    g.nh = s->nh;
    g.Lrow = s->dT;
    g.carry_in = s->dcarry[s->cur];
    g.carry_out = s->dcarry[s->cur ^ 1];
    if (n >= s->hist_cap) {
        g.hist = s->dhist[s->hcur];
        g.hist_len = s->hist_cap;
        s->hist_n = s->hist_cap;
    } else {
        const int64_t keep = std::min<int64_t>(s->hist_n, s->hist_cap - n);
        double *dst = s->dhist[s->hcur ^ 1];
        if (keep > 0)
            OSZ_HIP(hipMemcpy2DAsync(dst, sizeof(double) * s->hist_cap,
                                     s->dhist[s->hcur] + (s->hist_n - keep), sizeof(double) * s->hist_cap,
                                     sizeof(double) * keep, fir->nch, hipMemcpyDeviceToDevice, st));
        OSZ_HIP(hipMemcpy2DAsync(dst + keep, sizeof(double) * s->hist_cap, x, sizeof(double) * ldx,
                                 sizeof(double) * n, fir->nch, hipMemcpyDeviceToDevice, st));
        s->hcur ^= 1;
        s->hist_n = keep + n;
    }
    s->since_import += n;
    zp_kern_t kern = zpn_fwd_kernel_for(NB, s->NM, s->NS);
    if (!kern) return fail(OSZ_ERR_STATE, "forward chain kernel: no instance for %d rows, %d modes (%d slow)", NB, s->NM, s->NS);
    const int ns = 2 * s->nh;
    const size_t lds = sizeof(fft::cube::C2) * fft::cube::SLOTS +
                       sizeof(double) * (ns + 2 * s->R * s->NS * 2 + s->NM * 2 + 20 * s->NM * 2 + 2 * s->NS * ns) + 1024;
    OSZ_DYN_LDS(kern, lds);
    {
        KernelTimer kt("chain_fwd", st);
        hipLaunchKernelGGL(kern, dim3((unsigned)nruns, fir->nch), dim3(256), lds, st, g);
    }
    OSZ_HIP(hipGetLastError());
    {
        // NaN reach of the forward pass across runs (sos_tile.h): runs of (r W) / nruns blocks of S samples
        int rcs = sos_seal_launch(g.f.y, g.f.ldy, n, (int)nruns, W, nruns, (int64_t)S, nullptr, 0, fir->nch,
                                  g.carry_out, kSpecLdc, kSpecLdc, st);
        if (rcs) return rcs;
    }
    s->cur ^= 1;
    s->true_valid = false;
    *taken = true;
    return OSZ_OK;
}


// the link of this pair of handles (tables built on first use); a handle paired with
// another partner before gets its own state back first
static int spec_link(osz_fir_s *fir, osz_sos_s *sos, hipStream_t st, ChainSpec **out) {
    ChainSpec *s = sos->spec;
    if (s && s->fir != fir) {
        // the cascade is paired with another FIR from here on
        int rc = spec_settle(s, st);
        if (rc) return rc;
        spec_unlink(s);
        s = nullptr;
    }
    if (!s && fir->spec) {
        int rc = spec_settle(fir->spec, st);
        if (rc) return rc;
        spec_unlink(fir->spec);
    }
    if (!s) {
        s = new ChainSpec();
        s->fir = fir;
        s->sos = sos;
        fir->spec = sos->spec = s;
        int rc = spec_build(s);
        if (rc) {
            spec_unlink(s);
            return rc;
        }
    }
    *out = s;
    return OSZ_OK;
}

// which kernel osz_chain_forward runs whole blocks of this pair on (osz_chain_forward_route)
int spec_route(osz_fir_s *fir, osz_sos_s *sos, hipStream_t st, int *route) {
    ChainSpec *s = nullptr;
    int rc = spec_link(fir, sos, st, &s);
    if (rc) return rc;
    *route = !s->eligible ? 0 : s->nega ? 2 : 1;
    return OSZ_OK;
}

// FIR + forward cascade of one chunk by the spectral kernel, if it applies to this
// pair of handles and this chunk (*taken says); `between` runs before the launch
// (osz_chain_step starts the backward pass on its side stream there)
int spec_try_forward(osz_fir_s *fir, osz_sos_s *sos, const double *x, int64_t ldx, int64_t n, double *f,
                     int64_t ldf, hipStream_t st, const std::function<int()> &between, bool *taken) {
    *taken = false;
    ChainSpec *s = nullptr;
    {
        int rc = spec_link(fir, sos, st, &s);
        if (rc) return rc;
    }
    if (!s->eligible) return OSZ_OK;
    if (s->nega) return specn_forward(s, x, ldx, n, f, ldf, st, between, taken);
    const int NR = s->NR, S = 256 * NR;
    const int64_t pair = 2 * (int64_t)S;
    if (n < 2 * pair) return OSZ_OK;                      // an opening and a closing pair at least
    if (!s->carry_valid) {
        int rc = spec_import(s, st);
        if (rc) return rc;
    }
    {
        int rc = between();
        if (rc) return rc;
    }
    const int64_t npw = n / pair, rem = n - npw * pair;
    const int64_t W = rem == 0 ? npw - 1 : npw;
    const int64_t nlast = n - W * pair;                   // the closing pair: 1 .. 2 S samples
    // one round of resident workgroups (two per CU); a run has at least one pair of its own
    int64_t nruns = 512 / fir->nch;
    if (nruns > W) nruns = W;
    if (nruns < 1) nruns = 1;
    SpecArgs g{};
    g.f.x = x;
    g.f.y = f;
    g.f.ldx = ldx;
    g.f.ldy = ldf;
    g.f.n = n;
    g.f.skip = 0;
    g.f.wlen = fir->ntaps;
    g.f.step = S;
    g.f.H = s->dH;
    g.f.tb = fir->tb;
    g.n = n;
    g.W = W;
    g.nruns = (int)nruns;
    g.la = (int)std::min<int64_t>(nlast, S);
    g.lb = (int)(nlast - g.la);
    g.R = s->R;
    g.M = s->dM;
    g.P = s->dP;
    g.Lrow = s->dL;
    g.carry_in = s->dcarry[s->cur];
    g.carry_out = s->dcarry[s->cur ^ 1];
    // the input a later spec_settle replays: the kernel's last run copies the chunk's
    // last hist_cap samples; shorter chunks are appended here
    if (n >= s->hist_cap) {
        g.hist = s->dhist[s->hcur];
        g.hist_len = s->hist_cap;
        s->hist_n = s->hist_cap;
    } else {
        const int64_t keep = std::min<int64_t>(s->hist_n, s->hist_cap - n);
        double *dst = s->dhist[s->hcur ^ 1];
        if (keep > 0)
            OSZ_HIP(hipMemcpy2DAsync(dst, sizeof(double) * s->hist_cap,
                                     s->dhist[s->hcur] + (s->hist_n - keep), sizeof(double) * s->hist_cap,
                                     sizeof(double) * keep, fir->nch, hipMemcpyDeviceToDevice, st));
        OSZ_HIP(hipMemcpy2DAsync(dst + keep, sizeof(double) * s->hist_cap, x, sizeof(double) * ldx,
                                 sizeof(double) * n, fir->nch, hipMemcpyDeviceToDevice, st));
        s->hcur ^= 1;
        s->hist_n = keep + n;
        g.hist = nullptr;
        g.hist_len = 0;
    }
    s->since_import += n;
    // the composite spectrum is requested per pair (FirPair HPRE = 16): resident in registers
    // (HPRE = -1) the kernel spills 36 of its 64 words and the step takes 2.54 instead of 2.29 ms
    spec_kern_t kern = s->NM == 2 ? spec_kernel_for<2, 16>(NR) : s->NM == 4 ? spec_kernel_for<4, 16>(NR)
                                                                            : spec_kernel_for<6, 16>(NR);
    const size_t lds = sizeof(fft::cube::C2) * fft::cube::SLOTS +
                       sizeof(double) * (2 * kSpecFit + 2 * 2 * s->NM * 2 + 5 * kSpecRMax * s->NM * 2 +
                                         kSpecRMax * s->NM * 2 + 32 * s->NM * 2 + 2 * s->NM * kSpecFit);
    OSZ_DYN_LDS(kern, lds);
    {
        KernelTimer kt("chain_fwd", st);
        hipLaunchKernelGGL(kern, dim3((unsigned)nruns, fir->nch), dim3(256), lds, st, g);
    }
    OSZ_HIP(hipGetLastError());
    {
        // NaN reach of the forward pass across runs (sos_tile.h): from the first run whose last
        // output is not finite the rest of the chunk and the carry are NaN
        int rcs = sos_seal_launch(g.f.y, g.f.ldy, n, (int)nruns, W, nruns, 2 * (int64_t)S, nullptr, 0, fir->nch,
                                  g.carry_out, kSpecLdc, kSpecLdc, st);
        if (rcs) return rcs;
    }
    s->cur ^= 1;
    s->true_valid = false;
    *taken = true;
    return OSZ_OK;
}

}  // namespace osz
