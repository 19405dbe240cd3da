// chain.hip -- K1+K2 fused: overlap-add FIR feeding the forward pass of a
// biquad cascade without the FIR output ever touching HBM.
//
// The reference chains the two as separate generators: oaconvolve
// (core/numerical.py:158-298) yields arrays that sosfilt / the forward half of
// sosfiltfilt (:301-335, :374-386) consumes.  Separately they cost 16 + 16 B
// of HBM traffic per channel-sample; fused, the 6144 (= 2 x 256 NR) output
// samples of a pair of FIR blocks go from the FIR's registers through the idle
// cube (LDS) into lane blocks of T = 2 NR consecutive samples -- exactly one
// tile of the time-parallel SOS recurrence (sos_tile.h) -- and only the
// cascade's output is written: 8 + 8 B per sample.
//
// Work split: a workgroup walks a run of consecutive pairs of one channel.
// Run 0 starts from the carried state of both iterators (FIR overlap tail,
// SOS section states).  A later run starts `pre_pairs` pairs early from zero
// states and discards those outputs: after (wlen - 1) samples the overlap
// tail is right, after warm_len more the cascade has forgotten its start
// (the same 1e-18 bound as the time segments of sos.hip).  The last run
// leaves both carried states.  Pairs are whole by construction; the host
// sends the ragged end of a chunk through the separate kernels.
#include <algorithm>
#include <cstdlib>

#include <functional>

#include "common.h"
#include "fft4096.h"
#include "fir_pair.h"
#include "handles.h"
#include "sos_tile.h"

namespace osz {

struct ChainArgs {
    FirArgs f;                   // x, ldx, wlen, step, H, tb; y / ldy = the cascade's output
    int nsec;
    const double *fir_state_in;  // (nch, wlen - 1) carried overlap tail
    double *fir_state_out;
    const double *sos_state_in;  // (nsec, nch, 2)
    double *sos_state_out;
    const double *lane_tab;      // [nsec][4][kSos2Tab] for T = 2 NR, or null (first scan variant)
    int64_t npairs;              // whole pairs of blocks in this call
    int nruns, pre_pairs;
};

template <int NR, bool V2>
__global__ __launch_bounds__(256, 2) void chain_kernel(ChainArgs g,
                                                       const SosSection *__restrict__ sec) {
    constexpr int T = 2 * NR;        // samples per lane of the SOS tile: one pair = one tile
    constexpr int PITCH = T + 1;     // lane-block pitch in LDS (odd: conflict-free b64)
    extern __shared__ fft::cube::C2 cube_lds[];
    double *tile = reinterpret_cast<double *>(cube_lds);   // 256 * PITCH doubles, over the idle cube
    double *agg = tile + 2 * fft::cube::SLOTS;             // [2][4][2] wave aggregates
    double *sst = agg + 2 * 4 * 2;                         // [2][kSosMaxSec][2] tile start states
    double *ltab = sst + 2 * kSosMaxSec * 2;               // [nsec][4][kSos2Tab] lane tables (if any)
    const FirArgs &a = g.f;
    const int t = threadIdx.x, run = blockIdx.x, c = blockIdx.y;
    const int w = t >> 6, l = t & 63;
    const int wm1 = a.wlen - 1;
    const int64_t p0 = ((int64_t)run * g.npairs) / g.nruns;
    const int64_t p1 = ((int64_t)(run + 1) * g.npairs) / g.nruns;
    const int64_t ps = run == 0 ? 0 : p0 - g.pre_pairs;
    const double *xr = a.x + (int64_t)c * a.ldx;
    double *yr = a.y + (int64_t)c * a.ldy;

    FirPair<NR, -1> P{a, t, wm1, xr, yr, 2 * p1, cube_lds};
    fft::cube::tw_load(t, a.tb, P.tw1, P.tw2);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k = t + 256 * fft::dr(r);
        P.Hr[r] = a.H[2 * k];
        P.Hi[r] = a.H[2 * k + 1];
    }
#pragma unroll
    for (int j = 0; j < 16 - NR; ++j) {
        const int p = 256 * j + t;
        P.cr[j] = (run == 0 && p < wm1) ? g.fir_state_in[(int64_t)c * wm1 + p] : 0.0;
    }
    if (V2)
        for (int i = t; i < g.nsec * 4 * kSos2Tab; i += 256) ltab[i] = g.lane_tab[i];
    if (t < g.nsec) {
        sst[(0 * kSosMaxSec + t) * 2 + 0] = run == 0 ? g.sos_state_in[((int64_t)t * gridDim.y + c) * 2 + 0] : 0.0;
        sst[(0 * kSosMaxSec + t) * 2 + 1] = run == 0 ? g.sos_state_in[((int64_t)t * gridDim.y + c) * 2 + 1] : 0.0;
    }
    __syncthreads();
    int parity = 0, aggbuf = 0;

    for (int64_t p = ps; p < p1; ++p) {
        const int64_t start = 2 * p * a.step;           // first sample of block a
        double re[16], im[16];
        {
            const double *pa = xr + start + t;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                re[j] = j < NR ? pa[256 * j] : 0.0;
                im[j] = j < NR ? pa[256 * (j + NR)] : 0.0;
            }
        }
        P.transform(re, im);
        // overlap add in registers (see FirPair::fast_pair)
#pragma unroll
        for (int j = 0; j < 16 - NR; ++j) {
            const int q = 256 * j + t;
            if (q < wm1) {
                re[j] += P.cr[j];
                im[j] += re[j + NR];
                P.cr[j] = im[j + NR];
            }
        }
        __syncthreads();   // every thread is done reading the cube (inverse pass 1)
        // rows (sample 256 j + t) -> lane blocks (samples [T u, T u + T) of the pair);
        // positions recomputed per pair from an opaque copy of t (hoisted out of
        // the loop they would pin 2 NR registers)
        int tt = t;
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int sa = 256 * j + tt, sb = 256 * (NR + j) + tt;
            tile[sa + sa / T] = re[j];
            tile[sb + sb / T] = im[j];
        }
        __syncthreads();
        double v[T];
        {
            const double *blk = tile + PITCH * t;
#pragma unroll
            for (int k = 0; k < T; ++k) v[k] = blk[k];
        }
        // one tile of the forward cascade; its first barrier also ends the block reads
        if (V2)
            sos_tile_full2<T, 4>(v, sec, ltab, g.nsec, sst, agg, parity, aggbuf, w, l);
        else
            sos_tile_full<T, 4>(v, sec, g.nsec, sst, agg, parity, aggbuf, w, l);
        if (p >= p0) {
            double *blk = tile + PITCH * t;
#pragma unroll
            for (int k = 0; k < T; ++k) blk[k] = v[k];
            __syncthreads();
            double *q = yr + start + t;
            asm volatile("" : "+v"(tt));
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int sa = 256 * j + tt, sb = 256 * (NR + j) + tt;
                q[256 * j] = tile[sa + sa / T];
                q[256 * (NR + j)] = tile[sb + sb / T];
            }
        }
        __syncthreads();   // before the next pair's pass 1 writes the cube
    }
    if (run == g.nruns - 1) {
#pragma unroll
        for (int j = 0; j < 16 - NR; ++j) {
            const int p = 256 * j + t;
            if (p < wm1) g.fir_state_out[(int64_t)c * wm1 + p] = P.cr[j];
        }
        if (t < g.nsec) {
            g.sos_state_out[((int64_t)t * gridDim.y + c) * 2 + 0] = sst[(parity * kSosMaxSec + t) * 2 + 0];
            g.sos_state_out[((int64_t)t * gridDim.y + c) * 2 + 1] = sst[(parity * kSosMaxSec + t) * 2 + 1];
        }
    }
}

}  // namespace osz

using namespace osz;

extern "C" {

// `between` runs after the head launches and before the fused kernel is queued
// (osz_chain_step starts the backward pass on its side stream there)
static int chain_forward_impl(osz_fir_t fir, osz_sos_t sos, const double *x, int64_t ldx, int64_t n,
                              double *f, int64_t ldf, void *stream,
                              const std::function<int()> &between) {
    OSZ_REQUIRE(fir && sos && x && f, "osz_chain_forward: null argument");
    OSZ_REQUIRE(fir->nch == sos->nch, "osz_chain_forward: %d FIR channels, %d SOS channels",
                fir->nch, sos->nch);
    OSZ_REQUIRE(n >= 0 && ldx >= n && ldf >= n, "osz_chain_forward: n=%lld ldx=%lld ldf=%lld",
                (long long)n, (long long)ldx, (long long)ldf);
    if (n == 0) return between();
    OSZ_SAME_DEVICE(fir, "osz_chain_forward");
    OSZ_SAME_DEVICE(sos, "osz_chain_forward");
    hipStream_t st = as_stream(stream);
    {
        // cascades whose ringing dies within the guard rows of the transform: FIR and
        // cascade as one multiplication per bin (chain_spec.hip), any chunk length
        bool taken = false;
        int rc = spec_try_forward(fir, sos, x, ldx, n, f, ldf, st, between, &taken);
        if (rc || taken) return rc;
        // not this time: the kernels below work on the handles' own states
        rc = spec_touch(sos->spec, st);
        if (rc) return rc;
    }
    // The samples that are not whole block pairs go FIRST, through the plain kernels
    // (FIR into the output rows, then the cascade in place): as the head of the chunk
    // they are queued before the fused kernel and, in osz_chain_step, before the
    // backward pass on the other stream.  (Three small launches behind two kernels
    // that fill the chip wait for a CU with 64 KB of LDS to spare: 0.46 ms instead of
    // 0.02 ms at 256 x 2^20.  With OSZ_CHAIN_DEFER the previous step's backward pass
    // may still hold that LDS, so the order alone does not buy the time back.)
    auto plain = [&](int64_t lo, int64_t hi) -> int {
        // (osz_sos_forward runs one workgroup per channel when y aliases x: every
        // tile is read before it is written)
        int rc = osz_fir_push(fir, x + lo, ldx, hi - lo, f + lo, ldf, 0, stream);
        if (rc) return rc;
        return osz_sos_forward(sos, f + lo, ldf, f + lo, ldf, hi - lo, stream);
    };
    int64_t whole = 0, head = 0;   // [head, head + whole): the fused kernel's samples
    if (fir->parts.size() == 1 && fir->ntaps >= 2) {
        FirPart &pt = fir->parts[0];
        const int nr = pt.step / 256;
        const int wm1 = pt.ntaps - 1;
        const int64_t pair = 2 * (int64_t)pt.step;
        const int64_t npairs = n / pair;
        const int64_t pre = (sos->warm_len + wm1 + pair - 1) / pair;
        int64_t nruns = 512 / fir->nch;
        if (nruns < 1) nruns = 1;
        if (pre > 0 && nruns > npairs / (4 * pre)) nruns = npairs / (4 * pre);
        if (nruns < 1) nruns = 1;
        if (npairs >= 4 && nr >= 8 && nr <= 15) {
            const SosSection *dsec = nullptr;
            int rc = sos_tables_for(sos, 2 * nr, &dsec);
            if (rc) return rc;
            const double *ltab = nullptr;
            rc = sos_lane_table_for(sos, 2 * nr, &ltab);     // null for more than 8 sections
            if (rc) return rc;
            // an even head keeps the 16-byte alignment of the rows
            head = n - npairs * pair;
            if (head & 1) head = 0;
            if (head > 0) {
                int rc2 = plain(0, head);
                if (rc2) return rc2;
            }
            {
                int rc2 = between();
                if (rc2) return rc2;
            }
            ChainArgs g{};
            g.f.x = x + head;
            g.f.y = f + head;
            g.f.ldx = ldx;
            g.f.ldy = ldf;
            g.f.n = npairs * pair;
            g.f.skip = 0;
            g.f.wlen = pt.ntaps;
            g.f.step = pt.step;
            g.f.H = pt.dH;
            g.f.tb = fir->tb;
            g.nsec = sos->nsec;
            g.fir_state_in = pt.dstate[pt.cur];
            g.fir_state_out = pt.dstate[pt.cur ^ 1];
            g.sos_state_in = sos->dstate;
            g.sos_state_out = sos->dstate_alt;
            g.lane_tab = ltab;
            g.npairs = npairs;
            g.nruns = (int)nruns;
            g.pre_pairs = (int)pre;
            using kern_t = void (*)(ChainArgs, const SosSection *);
            static const kern_t kerns1[8] = {chain_kernel<8, false>,  chain_kernel<9, false>,
                                             chain_kernel<10, false>, chain_kernel<11, false>,
                                             chain_kernel<12, false>, chain_kernel<13, false>,
                                             chain_kernel<14, false>, chain_kernel<15, false>};
            static const kern_t kerns2[8] = {chain_kernel<8, true>,  chain_kernel<9, true>,
                                             chain_kernel<10, true>, chain_kernel<11, true>,
                                             chain_kernel<12, true>, chain_kernel<13, true>,
                                             chain_kernel<14, true>, chain_kernel<15, true>};
            const kern_t *kerns = ltab ? kerns2 : kerns1;
            const size_t lds = sizeof(fft::cube::C2) * fft::cube::SLOTS +
                               sizeof(double) * (2 * 4 * 2 + 2 * kSosMaxSec * 2) +
                               (ltab ? sizeof(double) * (size_t)sos->nsec * 4 * kSos2Tab : 0);
            OSZ_DYN_LDS(kerns[nr - 8], lds);
            {
                KernelTimer kt("chain_fwd", st);
                hipLaunchKernelGGL(kerns[nr - 8], dim3((unsigned)nruns, fir->nch), dim3(256), lds,
                                   st, g, dsec);
            }
            OSZ_HIP(hipGetLastError());
            {
                // NaN reach (sos_tile.h): a run that started from zero states knows nothing of
                // a NaN the cascade met before its pre-roll
                int rcs = sos_seal_launch(f + head, ldf, npairs * pair, (int)nruns, npairs, nruns, pair,
                                          g.sos_state_out, sos->nsec, fir->nch, nullptr, 0, 0, st);
                if (rcs) return rcs;
            }
            pt.cur ^= 1;
            std::swap(sos->dstate, sos->dstate_alt);
            whole = npairs * pair;
        }
    }
    if (whole == 0) {
        int rc = between();
        if (rc) return rc;
    }
    if (head + whole < n) {
        // the ragged end left by an odd head (or everything, for shapes the fused
        // kernel does not take)
        int rc = plain(head + whole, n);
        if (rc) return rc;
    }
    return OSZ_OK;
}

int osz_chain_forward(osz_fir_t fir, osz_sos_t sos, const double *x, int64_t ldx, int64_t n,
                      double *f, int64_t ldf, void *stream) {
    return chain_forward_impl(fir, sos, x, ldx, n, f, ldf, stream, [] { return OSZ_OK; });
}

int osz_chain_forward_route(osz_fir_t fir, osz_sos_t sos, void *stream) {
    if (!fir || !sos || fir->nch != sos->nch) {
        osz::fail(OSZ_ERR_INVALID, "osz_chain_forward_route: null handle or channel counts differ");
        return -1;
    }
    int dev = -1;
    if (osz::current_device(&dev)) return -1;
    if (dev != fir->device || dev != sos->device) {
        osz::fail(OSZ_ERR_STATE, "osz_chain_forward_route: handles belong to another device");
        return -1;
    }
    int route = 0;
    if (spec_route(fir, sos, as_stream(stream), &route)) return -1;
    return route;
}

// Do two (nch, n) views, rows ld apart, share an element?  Views of one parent
// buffer (equal pitch) are compared column-wise -- the chunks of a ring buffer
// interleave in memory without touching; anything else by its address range.
static bool views_clash(const double *p, int64_t ldp, int64_t np, const double *q, int64_t ldq,
                        int64_t nq, int nch) {
    if (!p || !q || np <= 0 || nq <= 0) return false;
    const double *p1 = p + ((int64_t)(nch - 1) * ldp + np), *q1 = q + ((int64_t)(nch - 1) * ldq + nq);
    if (!(p < q1 && q < p1)) return false;
    if (ldp != ldq || np > ldp || nq > ldq) return true;
    int64_t e = (q - p) % ldp;                  // q's first column relative to p's, in [0, ld)
    if (e < 0) e += ldp;
    return e < np || e + nq > ldp;
}

// One steady-state step of FIR -> sosfiltfilt.  The fused forward kernel is bound by
// arithmetic and latency (two waves per SIMD, 2.2 TB/s), the backward pass of an earlier
// chunk by memory: side by side they fill each other's gaps (2.65-2.70 ms against 2.83 ms
// for either order on one stream, 256 x 2^20).  The backward pass runs on a stream of the
// SOS handle's own, behind everything queued on `stream` so far (its inputs fa, fb come
// from earlier steps).
//   flags = 0: `stream` is ordered behind the backward pass again before the call
//     returns -- one stream-ordered operation for the caller, f and y both ready for
//     whatever it queues next.
//   flags = OSZ_CHAIN_DEFER: the backward pass is left running; y (and the right to
//     overwrite fa / fb) belongs to the caller only after the NEXT osz_chain_step or
//     osz_chain_wait on this handle has been queued.  That next step waits for it only
//     if it has to: when its forward output f overlaps the chunks the pass reads --
//     with a ring of four forward buffers it never does, and the backward pass of step
//     k may finish under the forward kernel of step k + 1.
int osz_chain_step(osz_fir_t fir, osz_sos_t sos, const double *x, int64_t ldx, int64_t n, double *f,
                   int64_t ldf, const double *fa, int64_t ldfa, int64_t na, const double *fb,
                   int64_t ldfb, int64_t nb, double *y, int64_t ldy, int flags, void *stream) {
    OSZ_REQUIRE(fir && sos && x && f && fa && y, "osz_chain_step: null argument");
    OSZ_REQUIRE(na >= 1 && ldfa >= na && ldy >= na, "osz_chain_step: bad chunk a");
    OSZ_REQUIRE(!fb || (nb >= 1 && ldfb >= nb), "osz_chain_step: bad chunk b");
    OSZ_REQUIRE(n >= 0 && ldf >= n, "osz_chain_step: n=%lld ldf=%lld", (long long)n, (long long)ldf);
    OSZ_REQUIRE((flags & ~OSZ_CHAIN_DEFER) == 0, "osz_chain_step: unknown flags %d", flags);
    OSZ_SAME_DEVICE(fir, "osz_chain_step");
    OSZ_SAME_DEVICE(sos, "osz_chain_step");
    hipStream_t st = as_stream(stream);
    if (!sos->side) {
        const size_t sb = sizeof(double) * (size_t)sos->nsec * sos->nch * 2;
        OSZ_HIP(hipStreamCreateWithFlags(&sos->side, hipStreamNonBlocking));
        OSZ_HIP(hipEventCreateWithFlags(&sos->side_go, hipEventDisableTiming));
        OSZ_HIP(hipEventCreateWithFlags(&sos->side_done[0], hipEventDisableTiming));
        OSZ_HIP(hipEventCreateWithFlags(&sos->side_done[1], hipEventDisableTiming));
        OSZ_HIP(hipMalloc(&sos->dtmp_side, sb));
        OSZ_HIP(hipMalloc(&sos->dcarry_side, sb));
    }
    const bool pending = sos->side_busy;          // the deferred pass of the previous step
    const int prev = sos->side_cur, cur = prev ^ 1;
    bool waited = false;
    if (pending) {
        // before the forward kernel only if its output would land in a chunk that pass reads
        bool clash = false;
        for (int q = 0; q < 2; ++q)
            clash = clash || views_clash(sos->side_in[q], sos->side_ld[q], sos->side_n[q], f, ldf, n,
                                         sos->nch);
        if (clash) {
            OSZ_HIP(hipStreamWaitEvent(st, sos->side_done[prev], 0));
            waited = true;
        }
        sos->side_busy = false;
    }
    int rc;
    {
        KernelTimer whole("chain_step", st);
        // head of the chunk (plain kernels) -> backward pass on the side stream, behind
        // everything queued on `stream` so far -> fused kernel on `stream`
        rc = chain_forward_impl(fir, sos, x, ldx, n, f, ldf, stream, [&]() -> int {
            OSZ_HIP(hipEventRecord(sos->side_go, st));
            OSZ_HIP(hipStreamWaitEvent(sos->side, sos->side_go, 0));
            return sosfiltfilt_chunk_on(sos, fa, ldfa, na, fb, ldfb, nb, y, ldy, sos->dtmp_side,
                                        sos->dcarry_side, sos->side);
        });
        OSZ_HIP(hipEventRecord(sos->side_done[cur], sos->side));
        sos->side_cur = cur;
        // the previous step's y is the caller's from here on (behind this step's forward kernel)
        if (pending && !waited) OSZ_HIP(hipStreamWaitEvent(st, sos->side_done[prev], 0));
        if (rc == OSZ_OK && (flags & OSZ_CHAIN_DEFER)) {
            sos->side_busy = true;
            sos->side_in[0] = fa;
            sos->side_ld[0] = ldfa;
            sos->side_n[0] = na;
            sos->side_in[1] = fb;
            sos->side_ld[1] = ldfb;
            sos->side_n[1] = fb ? nb : 0;
        } else {
            // also on failure: the side stream must not run ahead of the caller's
            OSZ_HIP(hipStreamWaitEvent(st, sos->side_done[cur], 0));
        }
    }
    return rc;
}

// `stream` is ordered behind the deferred backward pass of the last osz_chain_step.
int osz_chain_wait(osz_sos_t sos, void *stream) {
    OSZ_REQUIRE(sos, "osz_chain_wait: null handle");
    if (sos->side_busy) {
        OSZ_HIP(hipStreamWaitEvent(as_stream(stream), sos->side_done[sos->side_cur], 0));
        sos->side_busy = false;
    }
    return OSZ_OK;
}

}  // extern "C"
