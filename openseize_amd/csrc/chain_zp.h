// chain_zp.h -- what the two zero-phase chain kernels share: chain_zp.hip (two real blocks per
// 4096-point transform, the pair trick) and chain_zpn.hip (ONE real block of up to 30 rows per
// transform at the odd frequencies, fft::nega) -- launch arguments, the run partition, mode
// powers and burst sums, the request of a block's rows by LDS-DMA.
#pragma once

#include "chain_spec.h"
#include "fft4096.h"
#include "fir_pair.h"
#include "handles.h"
#include "nega_window.h"

namespace osz {

struct ZpArgs {
    FirArgs f;                 // x, ldx, wlen, step, H (zero-phase composite), tb; y / ldy: outputs n0 .. n
    double *y0;                // outputs 0 .. n0 (the tail of the caller's previous output chunk), or null
    int64_t ldy0, n0;
    int64_t n;                 // samples of this chunk
    int64_t W;                 // whole pairs on the fast path
    int nruns;
    int la, lb;                // lengths of the closing pair's two blocks
    int R, Rf, nh;             // burst rows backwards (and table stride) / forwards; fit samples at either end of row 15
    const double *M;           // [4 NM][2 nh]
    const double *P;           // [20][NM][2]
    const double *Lrow;        // [R][NM][2]
    const double *carry_in;    // (nch, kSpecLdc)
    double *carry_out;
    const double *held_in;     // (nch, 256 R): the previous chunk's last samples, one burst short
    double *held_out;
    double *hist;              // (nch, hist_len): the chunk's last input samples, or null
    int hist_len;
    long long *nanpos;         // (nch): stream position at which the forward stream went bad
    long long pos;             // stream position of this chunk's first sample
    int wclose;                // what closing the chunk costs its last run, in pairs
};

// First pair of run r (r = nruns: one past the last).  The runs of a channel do not cost the
// same: every run but the first starts one pair early, and the last one closes the chunk (a
// generic path, `wclose` pairs' worth); with balanced costs the last run gets fewer pairs of its
// own and the launch does not wait for it (256 channels, two runs each: 94.5 : 94.5 instead
// of 93 : 96).  Short runs keep the even split.
__host__ __device__ __forceinline__ int64_t zp_run_start(int64_t r, int64_t W, int nruns, int wclose) {
    if (r <= 0) return 0;
    if (r >= nruns) return W;
    if (W < 8 * (int64_t)nruns) return (r * W) / nruns;
    const int64_t V = W + (nruns - 1) + wclose;        // pairs, pre-roll pairs, the closing pair
    const int64_t s = (r * V) / nruns - (r - 1);
    return s < W - (nruns - r) ? s : W - (nruns - r);  // every later run keeps a pair of its own
}

// lambda^e for e = 0..255 from the three-level table [20][NM][2]
template <int NM>
__device__ __forceinline__ void zp_powers(const double *ptab, int e, double *pr, double *pi) {
    const double *p1 = ptab + ((e >> 5) * NM) * 2, *p2 = ptab + ((8 + ((e >> 2) & 7)) * NM) * 2,
                 *p3 = ptab + ((16 + (e & 3)) * NM) * 2;
#pragma unroll
    for (int q = 0; q < NM; ++q) {
        const double ar = p1[2 * q] * p2[2 * q] - p1[2 * q + 1] * p2[2 * q + 1];
        const double ai = p1[2 * q] * p2[2 * q + 1] + p1[2 * q + 1] * p2[2 * q];
        pr[q] = ar * p3[2 * q] - ai * p3[2 * q + 1];
        pi[q] = ar * p3[2 * q + 1] + ai * p3[2 * q];
    }
}

// Re sum_q kappa_q P_q
template <int NM>
__device__ __forceinline__ double zp_dot(const double *kk, const double *pr, const double *pi) {
    double c = 0.0;
#pragma unroll
    for (int q = 0; q < NM; ++q) c = fma(kk[2 * q], pr[q], fma(-kk[2 * q + 1], pi[q], c));
    return c;
}

// Where a channel's input FIRST holds a non-finite sample, exactly.  A block whose transform has
// gone bad says "somewhere in my samples"; the NaN reach of sosfiltfilt is decided per CHUNK
// (osz_chain_zp_seal), and the blocks of a step over several chunks (round 5: few channels) straddle
// chunk boundaries, so the block's start is not good enough -- it may lie a chunk early.  Called
// by a whole workgroup, once, when its run goes bad (a rare path): every thread looks through its
// share of the block's `count` input samples from `xb` on and lowers nanpos to `base` + the index
// of a non-finite one.  Returns whether any thread found one (if none: the bad values came in from
// before the block, and the caller falls back to the block's start).
__device__ __forceinline__ bool zp_exact_nanpos(const double *xb, int count, int t, long long *nanpos, long long base) {
    int hit = -1;
    for (int i = t; i < count && hit < 0; i += 256)
        if (!(fabs(xb[i]) <= 1.79769313486231570815e308)) hit = i;
    if (hit >= 0) atomicMin(nanpos, base + hit);
    return __syncthreads_or(hit >= 0) != 0;
}

// chain_zpn_*.hip: the kernel for NB rows per block (20 .. 30), NM modes (2, 4, 6, 8) of which the
// first NS (2, 4, 6) are slow
using zp_kern_t = void (*)(ZpArgs);
zp_kern_t zpn_kernel_for(int nb, int nm, int ns, int r);
// ... and its forward-chain instances (FIR -> sosfilt, no left tail; chain_spec.hip launches them)
zp_kern_t zpn_fwd_kernel_for(int nb, int nm, int ns);

}  // namespace osz
