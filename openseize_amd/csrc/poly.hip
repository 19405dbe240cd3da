// poly.hip -- K4: streaming polyphase rational resampler on gfx950.
//
// Replaces scipy.signal.resample_poly(padded, L, M, window=h) as the reference
// calls it per chunk (src/openseize/core/numerical.py:610, :631) together
// with the prior/next-chunk overhang padding (:590-632).  The chunk machinery
// reproduces the global definition
//     out[j] = sum_k L*h[k] * xup[j*M + half - k],   half = (ntaps-1)/2,
// xup = x zero-stuffed by L, zeros outside, ceil(N*L/M) outputs
// (resampling/resampling.py:91), which is what this kernel evaluates directly:
// only the taps k = (j*M + half) mod L, +L, +2L, ... hit non-zero samples.
// HBM sees each input once (8 B) and each output once (8*L/M B).  The handle
// carries the last `hist` input samples across pushes.  Two kernels:
// poly_block_kernel (LDS window, register-blocked, all practical L/M) and
// poly_kernel (one thread per output through L1/L2; only for M so large that
// the window of a 64-thread tile does not fit in LDS).
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "common.h"

namespace osz {

struct PolyArgs {
    const double *x;      // current chunk (nch, n)
    const double *hist;   // (nch, H): samples [nin - H, nin)
    double *y;
    const double *hL;     // L * h[k]
    int64_t ldx, ldy;
    int64_t nin;          // samples consumed before this push
    int64_t navail;       // nin + n
    int64_t j0, j1;       // outputs [j0, j1) are produced
    int m, L, M, H, half;
};

__global__ __launch_bounds__(256) void poly_kernel(PolyArgs a) {
    const int c = blockIdx.y;
    const double *xr = a.x + (int64_t)c * a.ldx;
    const double *hr = a.hist + (int64_t)c * a.H;
    double *yr = a.y + (int64_t)c * a.ldy;
    for (int64_t j = a.j0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < a.j1;
         j += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = j * a.M + a.half;
        int k = (int)(t % a.L);
        int64_t i = (t - k) / a.L;  // input index for tap k; decreases by 1 per L taps
        double acc = 0.0;
        for (; k < a.m && i >= 0; k += a.L, --i) {
            if (i >= a.navail) continue;
            const double v = i >= a.nin ? xr[i - a.nin] : (i >= a.nin - a.H ? hr[i - (a.nin - a.H)] : 0.0);
            acc = fma(a.hL[k], v, acc);
        }
        yr[j - a.j0] = acc;
    }
}

// Output j only meets the taps k = phi + L k', phi = (j M + half) mod L, and
// phi depends on j mod L alone, so the outputs of one residue class r = j mod L
// form a plain decimating FIR with its own sub-filter hsub[k'] = hL[phi + L k']:
//     out[jf + L q] = sum_k' hsub[k'] * x[itop + M q - k'],
//     itop = (jf M + half - phi) / L.
//
// Register-blocked polyphase kernel.  Within one residue class
//     out[o] = sum_k' hsub[k'] x[itop + M o - k'],
// and with u = msub - 1 - k' = M a + e the sum splits into M ordinary
// (non-decimating) FIRs over the phase streams x_e[i] = win[i M + e]:
//     out[o] = sum_e sum_a G[e][a] x_e[o + a],   G[e][a] = hsub[msub - 1 - (M a + e)].
// A thread owns R = 4 CONSECUTIVE outputs, so a block of 8 taps of one phase
// needs only 4 + 7 stream values for its 32 multiply-adds (2.9 per LDS read
// instead of 1), and the 8 coefficients are one wave-uniform scalar load.
// The window is staged deinterleaved by phase and padded one double per four
// (position i -> i + i/4), which makes the lane stride 5 doubles: conflict
// free, and every read of the blocked loop sits at a compile-time offset from
// one per-block base address.  Results leave through LDS so that consecutive
// lanes store consecutive outputs.
constexpr int kPolyR = 4;                 // consecutive outputs per thread
constexpr int kPolyBlk = 8;               // taps per coefficient block
constexpr int kPolyBatch = 24;            // staging loads in flight per thread

__device__ __forceinline__ int poly_pad(int i) { return i + (i >> 2); }

struct PolyBlockArgs {
    PolyArgs p;
    const double *G;   // [L][M][apad] blocked sub-filters, zero padded
    int apad;          // taps per phase stream, multiple of kPolyBlk
    int se;            // LDS doubles per phase stream
    int stepw, dqs;    // staging: threads of the workgroup rounded down to a multiple of M, / M
};

// ONE: L == 1 (decimation), a single class per tile.  NT threads produce
// NT * kPolyR outputs per class: large M shrinks the tile so that the window
// (M phase streams) still fits three workgroups' worth of LDS.
// EG = 2 (decimators whose window leaves room for a 128- or 64-thread tile only:
// M >= 8 or so): twice the threads on the same tile, the two halves of the
// workgroup take the lower and the upper half of the M phase streams and their
// partial sums meet in LDS -- the waves per CU that the large window took away.
template <int V>
__device__ __forceinline__ void poly_taps(double *acc, const double *g, const double *xv) {
#pragma unroll
    for (int q = 0; q < V; ++q)
#pragma unroll
        for (int s = 0; s < kPolyR; ++s) acc[s] = fma(g[q], xv[s + q], acc[s]);
}

template <bool ONE, int NT, int EG = 1>
__global__ __launch_bounds__(NT * EG, 3) void poly_block_kernel(PolyBlockArgs b) {
    constexpr int NJ = NT * kPolyR;
    constexpr int NTH = NT * EG;                    // threads of the workgroup
    static_assert(EG == 1 || ONE, "phase groups: decimators only");
    extern __shared__ double win[];
    const PolyArgs &a = b.p;
    const int c = blockIdx.y;
    const int tw = threadIdx.x;                     // staging / store index
    const int t = EG == 1 ? tw : tw % NT;           // output block
    const int eg = EG == 1 ? 0 : tw / NT;           // phase group (wave uniform: NT % 64 == 0)
    const double *xr = a.x + (int64_t)c * a.ldx;
    const double *hr = a.hist + (int64_t)c * a.H;
    double *yr = a.y + (int64_t)c * a.ldy;
    const int L = ONE ? 1 : a.L;
    // this workgroup produces the L * NJ consecutive outputs [J0, J0 + L NJ):
    // all L residue classes, one after the other, gathered in LDS so that the
    // stores are contiguous (a class on its own would write every L-th double)
    const int64_t J0 = a.j0 + (int64_t)blockIdx.x * NJ * L;
    if (J0 >= a.j1) return;
    double *outbuf = ONE ? win : win + a.M * b.se;      // L = 1: reuses the window
    const int nstream = NJ + b.apad;               // entries per phase stream
    const int wtot = nstream * a.M;
    const int dq = NTH / a.M, dr = NTH - dq * a.M;
    const int i00 = tw / a.M, e0 = tw - i00 * a.M;      // where a thread's first element goes
    for (int cls = 0; cls < L; ++cls) {
        const int64_t jf = J0 + cls;                    // first output of this class in the tile
        const int r = ONE ? 0 : (int)(jf % L);                  // its residue j mod L
        const int phi = ONE ? 0 : (int)(((int64_t)r * a.M + a.half) % L);
        const int msub = phi < a.m ? (a.m - phi + L - 1) / L : 0;
        const int64_t itop = (jf * a.M + a.half - phi) / L;
        const int64_t i0 = itop - (msub - 1);           // window: inputs [i0, i0 + wtot)
        // staging walks w = t, t + NT, ...: (i, e) = (w div M, w mod M) advance by
        // (NT div M, NT mod M) with a carry, no division in the loop
        int i = i00, e = e0;
        if (i0 >= a.nin && i0 + wtot <= a.navail && a.M <= NTH) {
            // The whole window lies in the current chunk: coalesced loads, kPolyBatch in flight
            // per thread, issued back to back, then placed.  A thread's successive elements
            // are a multiple of M apart (`stepw`, NTH rounded down), so its phase e never
            // changes and its stream position advances by a constant: placing an element costs
            // an add and the padding instead of a division's worth of carries; the NTH - stepw
            // elements two steps both cover are written twice with the same value.  The loads
            // go through a buffer descriptor (common.h) that ends with the window: the base and
            // the step's offset travel in scalar registers, what lies behind the window reads 0.
            const int stepw = b.stepw, dqs = b.dqs;
            const __amdgpu_buffer_rsrc_t rs =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(xr + (i0 - a.nin)), 0, wtot * 8, 0x00020000);
            const unsigned lane8 = 8u * (unsigned)tw;
            double *we = win + e0 * b.se;
            int iu = i00;
            for (int w0 = 0; w0 < wtot; w0 += kPolyBatch * stepw) {
                double v[kPolyBatch];
#pragma unroll
                for (int u = 0; u < kPolyBatch; ++u)      // (uniform: steps that start behind the window issue nothing)
                    v[u] = w0 + u * stepw < wtot ? buf_load(rs, lane8, 8u * (unsigned)(w0 + u * stepw)) : 0.0;
                if (w0 + kPolyBatch * stepw + NTH - stepw <= wtot) {
#pragma unroll
                    for (int u = 0; u < kPolyBatch; ++u) {
                        we[poly_pad(iu)] = v[u];
                        iu += dqs;
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < kPolyBatch; ++u) {
                        if (iu < nstream) we[poly_pad(iu)] = v[u];
                        iu += dqs;
                    }
                }
            }
        } else {
            for (int w = tw; w < wtot; w += NTH) {
                const int64_t g = i0 + w;
                double v = 0.0;
                if (g >= 0 && g < a.navail)
                    v = g >= a.nin ? xr[g - a.nin] : (g >= a.nin - a.H ? hr[g - (a.nin - a.H)] : 0.0);
                win[e * b.se + poly_pad(i)] = v;
                i += dq;
                e += dr;
                if (e >= a.M) { e -= a.M; ++i; }
            }
        }
        __syncthreads();
        double acc[kPolyR];
#pragma unroll
        for (int s = 0; s < kPolyR; ++s) acc[s] = 0.0;
        const double *Gr = b.G + (int64_t)r * a.M * b.apad;
        const int cq = msub > 0 ? (msub - 1) / a.M : 0, crem = msub > 0 ? (msub - 1) - cq * a.M : -1;   // (one division per class)
        const int ph0 = EG == 1 ? 0 : (eg * a.M) / EG, ph1 = EG == 1 ? a.M : ((eg + 1) * a.M) / EG;
        for (int ph = ph0; ph < ph1; ++ph) {
            const double *ge = Gr + (int64_t)ph * b.apad;
            const double *xe = win + ph * b.se + 5 * t;     // poly_pad(4 t) = 5 t
            // taps of this phase stream: the others of its apad are the table's padding -- skipped, not
            // multiplied (0 x NaN is NaN: a non-finite sample reaches the outputs whose TAPS touch it,
            // zero-valued taps of the caller's window included, and no other)
#ifdef OSZ_POLY_MULPAD     // (A/B builds: the table's padding multiplied as in rounds 1-4)
            const int cnt = b.apad;
#else
            const int cnt = ph <= crem ? cq + 1 : cq;      // = (msub - 1 - ph) / M + 1, or 0 behind the taps
#endif
            const int nfull = cnt & ~(kPolyBlk - 1);
            for (int a0 = 0; a0 < nfull; a0 += kPolyBlk) {
                double g[kPolyBlk], xv[kPolyBlk + kPolyR - 1];
#pragma unroll
                for (int q = 0; q < kPolyBlk; ++q) g[q] = ge[a0 + q];      // wave-uniform
                const double *xb = xe + (a0 + (a0 >> 2));                  // a0 multiple of 8
#pragma unroll
                for (int d = 0; d < kPolyBlk + kPolyR - 1; ++d) xv[d] = xb[d + (d >> 2)];
#pragma unroll
                for (int q = 0; q < kPolyBlk; ++q)
#pragma unroll
                    for (int s = 0; s < kPolyR; ++s) acc[s] = fma(g[q], xv[s + q], acc[s]);
            }
            if (cnt > nfull) {
                // the stream's last block: straight-line code per count of taps left, nothing of the
                // table's padding multiplied
                double g[kPolyBlk], xv[kPolyBlk + kPolyR - 1];
#pragma unroll
                for (int q = 0; q < kPolyBlk; ++q) g[q] = ge[nfull + q];
                const double *xb = xe + (nfull + (nfull >> 2));
#pragma unroll
                for (int d = 0; d < kPolyBlk + kPolyR - 1; ++d) xv[d] = xb[d + (d >> 2)];
                switch (cnt - nfull) {
                    case 1: poly_taps<1>(acc, g, xv); break;
                    case 2: poly_taps<2>(acc, g, xv); break;
                    case 3: poly_taps<3>(acc, g, xv); break;
                    case 4: poly_taps<4>(acc, g, xv); break;
                    case 5: poly_taps<5>(acc, g, xv); break;
                    case 6: poly_taps<6>(acc, g, xv); break;
                    default: poly_taps<7>(acc, g, xv); break;
                }
            }
        }
        __syncthreads();                                // everyone is done with the window
        if (EG == 1) {
#pragma unroll
            for (int s = 0; s < kPolyR; ++s) outbuf[cls + L * (kPolyR * t + s)] = acc[s];
        } else {
            // partial sums of phase group eg: outbuf[eg][output]
#pragma unroll
            for (int s = 0; s < kPolyR; ++s) outbuf[eg * NJ + kPolyR * t + s] = acc[s];
        }
    }
    __syncthreads();
    const int64_t left = a.j1 - J0;
    const int ntile = (int)(left < (int64_t)NJ * L ? left : (int64_t)NJ * L);
    for (int o = tw; o < ntile; o += NTH) {
        double v = outbuf[o];
#pragma unroll
        for (int g = 1; g < EG; ++g) v += outbuf[g * NJ + o];
        yr[(J0 - a.j0) + o] = v;
    }
}

// newhist = last H samples of (hist ++ x[0:n])
__global__ void poly_hist_kernel(const double *x, int64_t ldx, int64_t n, const double *hist,
                                 double *newhist, int H) {
    const int c = blockIdx.y;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= H) return;
    const int64_t rel = n - H + q;  // index relative to the start of x
    double v;
    if (rel >= 0)
        v = x[(int64_t)c * ldx + rel];
    else
        v = (rel + H >= 0) ? hist[(int64_t)c * H + rel + H] : 0.0;
    newhist[(int64_t)c * H + q] = v;
}

}  // namespace osz

using namespace osz;

struct osz_poly_s {
    int device;         // HIP device the handle's buffers live on
    int m, L, M, nch, H, half;
    double *dhL;
    double *dG;         // blocked sub-filters for poly_block_kernel, or null
    int apad, se, nt;   // nt: threads per workgroup (256, 128 or 64), 0 = kernel not usable
    double *dhist[2];
    int cur;
    int64_t nin, nout;
};

static int64_t ceil_div(int64_t a, int64_t b) { return a >= 0 ? (a + b - 1) / b : -((-a) / b); }

static int64_t poly_end(const osz_poly_s *h, int64_t navail, int final_) {
    // outputs j with all needed inputs available: j*M + half < navail*L
    int64_t e = final_ ? ceil_div(navail * h->L, h->M) : ceil_div(navail * h->L - h->half, h->M);
    if (e < h->nout) e = h->nout;
    return e;
}

extern "C" {

int osz_poly_create(osz_poly_t *h, const double *taps, int ntaps, int L, int M, int nch) {
    return osz_poly_create_centred(h, taps, ntaps, (ntaps - 1) / 2, L, M, nch);
}

int osz_poly_create_centred(osz_poly_t *h, const double *taps, int ntaps, int centre, int L, int M, int nch) {
    OSZ_REQUIRE(h && taps, "osz_poly_create: null argument");
    OSZ_REQUIRE(ntaps >= 1 && L >= 1 && M >= 1 && nch >= 1 && nch <= 65535,
                "osz_poly_create: bad sizes (ntaps=%d L=%d M=%d nch=%d; nch <= 65535)", ntaps, L, M, nch);
    OSZ_REQUIRE(centre >= 0 && centre < ntaps, "osz_poly_create_centred: centre=%d not a tap of %d", centre, ntaps);
    osz_poly_s *p = new osz_poly_s();
    p->device = 0;
    (void)hipGetDevice(&p->device);
    p->m = ntaps;
    p->L = L;
    p->M = M;
    p->nch = nch;
    p->half = centre;
    p->H = (ntaps - 1 + L - 1) / L + 1;
    p->cur = 0;
    p->nin = p->nout = 0;
    std::vector<double> hL(ntaps);
    for (int k = 0; k < ntaps; ++k) hL[k] = (double)L * taps[k];
    const size_t hb = sizeof(double) * (size_t)nch * p->H;
    OSZ_HIP(hipMalloc(&p->dhL, sizeof(double) * ntaps));
    OSZ_HIP(hipMalloc(&p->dhist[0], hb));
    OSZ_HIP(hipMalloc(&p->dhist[1], hb));
    OSZ_HIP(hipMemcpy(p->dhL, hL.data(), sizeof(double) * ntaps, hipMemcpyHostToDevice));
    OSZ_HIP(hipMemset(p->dhist[0], 0, hb));
    OSZ_HIP(hipMemset(p->dhist[1], 0, hb));
    // blocked sub-filters G[r][e][a] = hsub_r[msub_r - 1 - (M a + e)]
    p->dG = nullptr;
    {
        const int msub_max = (ntaps + L - 1) / L;
        int apad = (msub_max + M - 1) / M;                       // taps per phase stream
        apad = (apad + kPolyBlk - 1) / kPolyBlk * kPolyBlk;
        p->apad = apad;
        p->nt = 0;
        p->se = 0;
        for (int nt = 256; nt >= 64 && !p->nt; nt >>= 1) {
            const int nstream = nt * kPolyR + apad;
            const int se = (nstream + (nstream >> 2) + 2) | 1;   // odd: spreads the staging writes
            const size_t bytes = ((size_t)M * se + (L == 1 ? 0 : (size_t)nt * kPolyR * L)) * sizeof(double);
            // (the smallest tile may take most of a CU's 160 KB: one workgroup per CU then, still
            // fifty times the rate of the kernel that reads its window through the caches)
            if (bytes <= (nt == 64 ? 150 : 53) * 1024) {
                p->nt = nt;
                p->se = se;
            }
        }
        if (p->nt) {
            // The stream pitch decides how the staging writes fall on the banks: a ds_write_b64
            // is served in groups of 16 consecutive lanes, conflict free when their 16 double
            // addresses e * se + pad(i) differ mod 16 (MI355X_MICROARCH.md, LDS).  Lanes walk
            // (i, e) = (w div M, w mod M), so the best pitch depends on M: take, among the 16
            // pitches from the needed one up, the one with the fewest extra LDS cycles over the
            // first steps of a tile (it was "any odd pitch": 30 % of the LDS cycles were conflicts).
            // (threads per workgroup as osz_poly_push launches it: its phase-group rule)
            const size_t blds0 = ((size_t)M * p->se + (L == 1 ? 0 : (size_t)p->nt * kPolyR * L)) * sizeof(double);
            const int eg_on0 = blds0 > 53 * 1024 ? 4 : 2;
            const int nth = p->nt * ((L == 1 && p->nt <= 128) ? (eg_on0 >= 4 && M >= 4 ? 4 : M >= 2 ? 2 : 1) : 1);
            const int stepw = nth - nth % M, dqs = M <= nth ? stepw / M : 0;
            auto extra_cycles = [&](int se) {
                long cost = 0;
                for (int u = 0; u < 8; ++u)
                    for (int g0 = 0; g0 < nth; g0 += 16) {
                        int cnt[16] = {0}, worst = 0;
                        for (int tw = g0; tw < g0 + 16 && tw < nth; ++tw) {
                            const int i = tw / M + u * dqs, e = tw % M;
                            const int bank = (int)(((long)e * se + i + (i >> 2)) & 15);
                            worst = std::max(worst, ++cnt[bank]);
                        }
                        cost += worst - 1;
                    }
                return cost;
            };
            if (M <= nth) {
                int best = p->se;
                long best_cost = extra_cycles(best);
                for (int cand = p->se + 1; cand < p->se + 16 && best_cost > 0; ++cand) {
                    const size_t bytes = ((size_t)M * cand + (L == 1 ? 0 : (size_t)p->nt * kPolyR * L)) * sizeof(double);
                    if (bytes > (size_t)(p->nt == 64 ? 150 : 53) * 1024) break;
                    const long c = extra_cycles(cand);
                    if (c < best_cost) {
                        best_cost = c;
                        best = cand;
                    }
                }
                p->se = best;
            }
            std::vector<double> G((size_t)L * M * apad, 0.0);
            for (int r = 0; r < L; ++r) {
                const int phi = (int)(((int64_t)r * M + p->half) % L);
                const int msub = phi < ntaps ? (ntaps - phi + L - 1) / L : 0;
                for (int kk = 0; kk < msub; ++kk) {
                    const int u = msub - 1 - kk, aa = u / M, e = u % M;
                    G[((size_t)r * M + e) * apad + aa] = hL[phi + (size_t)L * kk];
                }
            }
            OSZ_HIP(hipMalloc(&p->dG, G.size() * sizeof(double)));
            OSZ_HIP(hipMemcpy(p->dG, G.data(), G.size() * sizeof(double), hipMemcpyHostToDevice));
        }
    }
    *h = p;
    return OSZ_OK;
}

int osz_poly_destroy(osz_poly_t h) {
    if (!h) return OSZ_OK;
    (void)hipFree(h->dhL);
    (void)hipFree(h->dG);
    (void)hipFree(h->dhist[0]);
    (void)hipFree(h->dhist[1]);
    delete h;
    return OSZ_OK;
}

int osz_poly_reset(osz_poly_t h, void *stream) {
    OSZ_REQUIRE(h, "osz_poly_reset: null handle");
    h->nin = h->nout = 0;
    OSZ_HIP(hipMemsetAsync(h->dhist[h->cur], 0, sizeof(double) * (size_t)h->nch * h->H,
                           as_stream(stream)));
    return OSZ_OK;
}

// ---- checkpoint / resume: [samples consumed, samples produced, history (nch x H)]
int64_t osz_poly_state_size(osz_poly_t h) { return h ? 2 + (int64_t)h->nch * h->H : -1; }

int osz_poly_get_state(osz_poly_t h, double *state, void *stream) {
    OSZ_REQUIRE(h && state, "osz_poly_get_state: null argument");
    OSZ_SAME_DEVICE(h, "osz_poly_get_state");
    hipStream_t st = as_stream(stream);
    state[0] = (double)h->nin;
    state[1] = (double)h->nout;
    OSZ_HIP(hipMemcpyAsync(state + 2, h->dhist[h->cur], sizeof(double) * (size_t)h->nch * h->H,
                           hipMemcpyDeviceToHost, st));
    OSZ_HIP(hipStreamSynchronize(st));
    return OSZ_OK;
}

int osz_poly_set_state(osz_poly_t h, const double *state, void *stream) {
    OSZ_REQUIRE(h && state, "osz_poly_set_state: null argument");
    OSZ_SAME_DEVICE(h, "osz_poly_set_state");
    OSZ_REQUIRE(state[0] >= 0 && state[1] >= 0, "osz_poly_set_state: negative counters");
    hipStream_t st = as_stream(stream);
    OSZ_HIP(hipMemcpyAsync(h->dhist[h->cur], state + 2, sizeof(double) * (size_t)h->nch * h->H,
                           hipMemcpyHostToDevice, st));
    OSZ_HIP(hipStreamSynchronize(st));
    h->nin = (int64_t)state[0];
    h->nout = (int64_t)state[1];
    return OSZ_OK;
}

int64_t osz_poly_out_count(osz_poly_t h, int64_t n, int final_) {
    if (!h || n < 0) return -1;
    return poly_end(h, h->nin + n, final_) - h->nout;
}

int osz_poly_push(osz_poly_t h, const double *x, int64_t ldx, int64_t n, int final_, double *y,
                  int64_t ldy, int64_t *n_out, void *stream) {
    OSZ_REQUIRE(h, "osz_poly_push: null handle");
    OSZ_REQUIRE(n >= 0 && (n == 0 || (x && ldx >= n)), "osz_poly_push: bad input");
    OSZ_SAME_DEVICE(h, "osz_poly_push");
    hipStream_t st = as_stream(stream);
    const int64_t navail = h->nin + n;
    const int64_t j1 = poly_end(h, navail, final_);
    const int64_t cnt = j1 - h->nout;
    OSZ_REQUIRE(cnt == 0 || (y && ldy >= cnt), "osz_poly_push: output too small for %lld samples",
                (long long)cnt);
    if (cnt > 0) {
        PolyArgs a{};
        a.x = x ? x : h->dhist[h->cur];
        a.hist = h->dhist[h->cur];
        a.y = y;
        a.hL = h->dhL;
        a.ldx = ldx;
        a.ldy = ldy;
        a.nin = h->nin;
        a.navail = navail;
        a.j0 = h->nout;
        a.j1 = j1;
        a.m = h->m;
        a.L = h->L;
        a.M = h->M;
        a.H = h->H;
        a.half = h->half;
        if (h->dG) {
            PolyBlockArgs b{a, h->dG, h->apad, h->se, 0, 0};
            const size_t blds = sizeof(double) * ((size_t)h->M * h->se +
                                                  (h->L == 1 ? 0 : (size_t)h->nt * kPolyR * h->L));
            using kern_t = void (*)(PolyBlockArgs);
            static const kern_t kerns[2][3] = {
                {poly_block_kernel<false, 256>, poly_block_kernel<false, 128>, poly_block_kernel<false, 64>},
                {poly_block_kernel<true, 256>, poly_block_kernel<true, 128>, poly_block_kernel<true, 64>}};
            // decimators on a 128- or 64-thread tile: two (or four) phase groups
            static const kern_t kerns2[2][2] = {
                {poly_block_kernel<true, 128, 2>, poly_block_kernel<true, 64, 2>},
                {poly_block_kernel<true, 128, 4>, poly_block_kernel<true, 64, 4>}};
            // two phase groups per workgroup (four measured no faster); four when the window
            // leaves room for one or two workgroups per CU only
            const int eg_on = blds > 53 * 1024 ? 4 : 2;
            const int eg = (h->L == 1 && h->nt <= 128) ? (eg_on >= 4 && h->M >= 4 ? 4 : eg_on >= 2 && h->M >= 2 ? 2 : 1) : 1;
            const bool split = eg > 1;
            const kern_t kern = split ? kerns2[eg == 4 ? 1 : 0][h->nt == 128 ? 0 : 1]
                                      : kerns[h->L == 1 ? 1 : 0][h->nt == 256 ? 0 : h->nt == 128 ? 1 : 2];
            b.stepw = eg * h->nt - (eg * h->nt) % h->M;
            b.dqs = b.stepw / h->M;
            OSZ_DYN_LDS(kern, blds > 64 * 1024 ? blds : 64 * 1024);
            const int64_t per = (int64_t)h->nt * kPolyR * h->L;
            const int64_t bx = (cnt + per - 1) / per;
            KernelTimer kt("poly_block", st);
            hipLaunchKernelGGL(kern,
                               dim3((unsigned)bx, h->nch), dim3(eg * h->nt), blds, st, b);
        } else {   // very large M: the window of even a 64-thread tile exceeds LDS
            int64_t bx = (cnt + 255) / 256;
            if (bx > 4096) bx = 4096;
            KernelTimer kt("poly", st);
            hipLaunchKernelGGL(poly_kernel, dim3((unsigned)bx, h->nch), dim3(256), 0, st, a);
        }
        OSZ_HIP(hipGetLastError());
    }
    if (n > 0) {
        hipLaunchKernelGGL(poly_hist_kernel, dim3((h->H + 255) / 256, h->nch), dim3(256), 0, st, x,
                           ldx, n, h->dhist[h->cur], h->dhist[h->cur ^ 1], h->H);
        OSZ_HIP(hipGetLastError());
        h->cur ^= 1;
    }
    h->nin = navail;
    h->nout = j1;
    if (n_out) *n_out = cnt;
    return OSZ_OK;
}

}  // extern "C"
