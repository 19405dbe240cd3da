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

int osz_chain_forward(osz_fir_t fir, osz_sos_t sos, const double *x, int64_t ldx, int64_t n,
                      double *f, int64_t ldf, void *stream) {
    OSZ_REQUIRE(fir && sos && x && f, "osz_chain_forward: null argument");
    OSZ_REQUIRE(fir->nch == sos->nch, "osz_chain_forward: %d FIR channels, %d SOS channels",
                fir->nch, sos->nch);
    OSZ_REQUIRE(n >= 0 && ldx >= n && ldf >= n, "osz_chain_forward: n=%lld ldx=%lld ldf=%lld",
                (long long)n, (long long)ldx, (long long)ldf);
    if (n == 0) return OSZ_OK;
    OSZ_SAME_DEVICE(fir, "osz_chain_forward");
    OSZ_SAME_DEVICE(sos, "osz_chain_forward");
    hipStream_t st = as_stream(stream);
    int64_t whole = 0;   // samples that go through the fused kernel
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
            if (!(getenv("OSZ_CHAIN_V2") && atoi(getenv("OSZ_CHAIN_V2")) == 0)) {
                rc = sos_lane_table_for(sos, 2 * nr, &ltab);
                if (rc) return rc;
            }
            ChainArgs g{};
            g.f.x = x;
            g.f.y = f;
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
            pt.cur ^= 1;
            std::swap(sos->dstate, sos->dstate_alt);
            whole = npairs * pair;
        }
    }
    if (whole < n) {
        // the ragged end (or everything, for shapes the fused kernel does not take):
        // FIR into the output rows, then the cascade in place (osz_sos_forward runs
        // one workgroup per channel when y aliases x: every tile is read before it
        // is written)
        int rc = osz_fir_push(fir, x + whole, ldx, n - whole, f + whole, ldf, 0, stream);
        if (rc) return rc;
        rc = osz_sos_forward(sos, f + whole, ldf, f + whole, ldf, n - whole, stream);
        if (rc) return rc;
    }
    return OSZ_OK;
}

}  // extern "C"
