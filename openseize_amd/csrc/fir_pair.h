// fir_pair.h -- device code of the overlap-add FIR shared by fir.hip (the
// stand-alone kernel) and chain.hip (FIR fused with the forward SOS pass): the
// launch arguments, the balanced run partition and FirPair, the per-thread
// state and per-pair work of one workgroup.
#pragma once

#include "common.h"
#include "fft4096.h"

namespace osz {

// In-kernel phase stamps for the diagnostic build only
// (benchmarks/fir_stamps.hip defines OSZ_FIR_STAMPS); the library build has none.
#ifdef OSZ_FIR_STAMPS
__device__ unsigned long long *g_fir_stamps = nullptr;   // [waves][12] cycle sums
__device__ unsigned long long g_fir_clock[2];            // {s_memtime, s_memrealtime} ticks of one run
#define OSZ_FSTAMP(slot)                                                             \
    do {                                                                             \
        unsigned long long now_;                                                     \
        __builtin_amdgcn_sched_barrier(0);                                           \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                           \
        stamp_acc[slot] += now_ - stamp_last;                                        \
        stamp_last = now_;                                                           \
    } while (0)
#else
#define OSZ_FSTAMP(slot) do { } while (0)
#endif

constexpr int kFirMaxTaps = 2049;  // NFFT - ntaps + 1 >= ntaps - 1

struct FirArgs {
    const double *x;
    double *y;
    int64_t ldx, ldy, n, skip;
    int wlen, step, R, nruns, accum;   // accum: y += (partitioned filters), else y =
    int64_t nblocks;
    const double *H;  // [4096][2], already divided by 4096
    fft::Tables tb;
    double *tails;    // [nch][nruns][wlen-1]
};

// Runs are balanced: run r owns the block PAIRS [r*npairs/nruns, (r+1)*npairs/nruns),
// i.e. it starts at block 2*floor(r*npairs/nruns).  (A remainder lumped into the
// last run makes that workgroup up to twice as long as the others and the
// whole launch waits for it.)
__host__ __device__ __forceinline__ int64_t fir_run_start(int64_t r, int64_t nblocks, int nruns) {
    const int64_t npairs = (nblocks + 1) / 2;
    const int64_t b = 2 * ((r * npairs) / nruns);
    return b < nblocks ? b : nblocks;
}

// NR = rows of 256 samples per block (block length step = 256 NR, chosen by
// the host as the largest multiple of 256 with step + ntaps - 1 <= 4096).
// A workgroup walks its run pair by pair on the cube layout of fft4096.h
// (interleaved complex, in-place exchanges, 64 KB of LDS): four barriers per
// pair, 16-byte LDS accesses, twiddle powers resident in registers, and an
// overlap that never leaves the registers: with whole blocks of 256 NR
// samples, sample p = 256 j + t of a's tail (register row j + NR) meets b's
// head at register row j of the SAME thread, and b's tail meets the next
// pair's a the same way, so the carried tail is cr[j], 16 - NR doubles per
// thread.  Which register rows carry samples is a compile-time fact there, so
// loads, stores and the first butterfly stage (whose rows >= NR are literal
// zeros) need no predication.  Ragged pairs (the first one of a push with a
// left cut, the last one) pass their
// tails through LDS, over the idle cube; they run in their own loops before
// and after the stretch of whole pairs so that their predicated code does not
// weigh on the register allocation of the hot loop.
// Measured and rejected on this kernel (profiles/README.md): requesting the
// filter spectrum per pair (before the second barrier or at its use) instead
// of keeping it resident WITHOUT using the freed registers (+3.5 %), requesting
// the NEXT pair's samples ahead with the spectrum resident (after the third
// barrier: +6 % time; before inverse pass 1: 162 spilled registers) -- the
// variant that pays is PF below: spectrum per pair, next samples into its
// registers, -3.4 % --, twiddles loaded from the tables per pass (+13 %),
// running the FIR of chunk k+1 beside the IIR step of chunk k on a second
// stream (+5 %, benchmarks/overlap_probe.py).
// CUBE2: the transform on fft::cube2 -- its second exchange stays inside a 16-lane row, so two
// of the four barriers per pair are wave-level fences -- with the spectrum stored in the
// order that layout leaves the bins in ([r][t]: fft::cube2::bin).
template <int NR, int HPRE, int PF = 0, bool CUBE2 = false>
struct FirPair {
    static constexpr bool PF2 = PF == 2;   // PF = 1: samples ahead only; 2: spectrum ahead as well
    using C2 = fft::cube::C2;
    static constexpr int NT_ = 16 - NR;   // register rows of the tail (wm1 <= 256 NT_)

    const FirArgs &a;
    const int t, wm1;
    const double *xr;
    double *yr;
    const int64_t blk1;
    C2 *L;
    fft::cube::TwPow tw1, tw2;
    double cr[NT_];
    double Hr[HPRE < 0 ? 16 : 1], Hi[HPRE < 0 ? 16 : 1];   // HPRE < 0: spectrum resident
    // PF: the NEXT pair's samples, requested right after this pair's spectrum
    // multiply -- into the registers the filter spectrum has just left (PF goes
    // with HPRE = 16: the spectrum is requested per pair, after pass 2, when the
    // data registers are dead).  They fly while the three inverse passes run,
    // and because they are OLDER than this pair's stores, waiting for them at
    // the next pair does not wait for those stores (one vmcnt on gfx950).
    double nx[PF ? 2 * NR : 1];
    int64_t next_blk = 0;
    // PF, whole pairs: the filter spectrum of the NEXT pair, requested after
    // inverse pass 1 and BEFORE this pair's stores -- requested behind them (as
    // the per-pair spectrum of HPRE does) it cannot be seen to land before they
    // have drained: one in-order counter (benchmarks/fir_stamps: 2800 ticks of
    // a 20 000-tick pair).  Same values every pair; the point is WHEN they
    // occupy registers: not during the inverse passes, where nx lives.
    typedef double d2_t __attribute__((ext_vector_type(2)));
    d2_t Hn[PF2 ? 16 : 1];

    __device__ __forceinline__ void request_spectrum() {
        int tt = t;
        asm volatile("" : "+v"(tt));   // per pair: hoisted, 16 addresses would spill
#pragma unroll
        for (int r = 0; r < (PF2 ? 16 : 0); ++r) {
            const double *ph = a.H + 2 * (tt + 256 * fft::dr(r));
            asm volatile("global_load_dwordx4 %0, %1, off ; osz:hn" : "=v"(Hn[PF2 ? r : 0]) : "v"(ph) : "memory");
        }
    }

    // The requests are inline assembly, outside the compiler's vmcnt bookkeeping:
    // at the loop top it would otherwise wait vmcnt(0..3), i.e. also for the
    // previous pair's stores issued a moment ago.  Loads and stores retire in
    // order on one counter, so nx has landed once at most 2 NR younger
    // operations (this pair's stores; with accum also its 2 NR loads of y) are
    // outstanding: wait_next() after the stores.  Untracked OLDER operations
    // only make the compiler's own waits more conservative, never too short.
    // (Spreading the requests of a pair over the three inverse passes instead of
    // one burst behind the multiply -- a burst holds the wave at the issue port
    // for ~120 cycles per load, benchmarks/fir_stamps -- measured 2 % SLOWER.)
    __device__ __forceinline__ void request_next(int64_t blk) {
        int64_t off = blk * a.step + t;
        asm volatile("" : "+v"(off));   // per pair: hoisted, 2 NR addresses would spill
        const double *pa = xr + off;
#pragma unroll
        for (int j = 0; j < (PF ? 2 * NR : 0); j += 2) {
            const double *pj = pa + 256 * j;
            asm volatile("global_load_dwordx2 %0, %1, off ; osz:nx" : "=v"(nx[PF ? j : 0]) : "v"(pj) : "memory");
            if (j + 1 < 2 * NR)
                asm volatile("global_load_dwordx2 %0, %1, off offset:2048 ; osz:nx"
                             : "=v"(nx[PF ? j + 1 : 0]) : "v"(pj) : "memory");
        }
    }
    // after the pair's stores: younger than nx are those 2 NR stores and, in
    // variant 2, the 16 spectrum requests issued just before them
    __device__ __forceinline__ void wait_next() {
        if (!PF) return;
        asm volatile("s_waitcnt vmcnt(%0) ; osz:nx" ::"n"(2 * NR + (PF2 ? 16 : 0)) : "memory");
#pragma unroll
        for (int j = 0; j < (PF ? 2 * NR : 0); ++j) asm volatile("" : "+v"(nx[PF ? j : 0]));
    }
    // variant 2, after the last whole pair of the run: its request for a next
    // pair's spectrum has no taker
    __device__ __forceinline__ void drain_spectrum() {
        if (!PF2) return;
        asm volatile("s_waitcnt vmcnt(0) ; osz:hn osz:nx" ::: "memory");
#pragma unroll
        for (int r = 0; r < (PF2 ? 16 : 0); ++r) asm volatile("" : "+v"(Hn[PF2 ? r : 0]));
    }
    // variant 2, before the spectrum multiply: younger than the spectrum requests
    // are the previous pair's 2 NR stores (nothing else is issued in between)
    __device__ __forceinline__ void wait_spectrum() {
        if (!PF2) return;
        asm volatile("s_waitcnt vmcnt(%0) ; osz:hn" ::"n"(2 * NR) : "memory");
#pragma unroll
        for (int r = 0; r < (PF2 ? 16 : 0); ++r) asm volatile("" : "+v"(Hn[PF2 ? r : 0]));
    }
#ifdef OSZ_FIR_STAMPS
    unsigned long long stamp_acc[12], stamp_last;
#endif

    __device__ __forceinline__ bool whole(int64_t blk) const {
        return blk + 1 < blk1 && (blk + 2) * a.step <= a.n && blk * a.step >= a.skip;
    }

    // forward transform, filter, inverse transform of the pair in re/im
    // REQ (whole pairs of the PF kernel only): request the next pair's samples
    // behind the spectrum multiply; the caller must wait_next()
    template <bool REQ = false>
    __device__ __forceinline__ void transform(double *re, double *im) {
        // LDS slot numbers are recomputed per pair from an opaque copy of the
        // thread index: hoisted out of the loop they would pin 33 registers
        int t = this->t;
        asm volatile("" : "+v"(t));
        if (CUBE2) fft::cube2::f1(t, re, im, tw1, L);
        else fft::cube::f1(t, re, im, tw1, L);
        OSZ_FSTAMP(1);   // sample loads landed + pass 1 + stores
        __syncthreads();
        OSZ_FSTAMP(2);   // barrier 1
        if (CUBE2) fft::cube2::f2(t, re, im, tw2, L);
        else fft::cube::f2(t, re, im, tw2, L);
        // HPRE > 0: that many filter-spectrum bins are requested before the barrier
        double hr[HPRE > 0 ? HPRE : 1], hi[HPRE > 0 ? HPRE : 1];
        // (a bin's byte offset from a 32-bit, provably small lane index: the loads take the
        // spectrum's address from scalar registers and need no 64-bit vector arithmetic)
        // (buffer addressing, common.h: the spectrum's address and a bin row's offset in scalar
        // registers, one 32-bit lane offset)
        const unsigned lane16 = 16u * ((unsigned)t & 255u);
        const __amdgpu_buffer_rsrc_t rh = buf_rsrc(a.H);
#pragma unroll
        for (int r = 0; r < ((HPRE > 0 && !(PF2 && REQ)) ? HPRE : 0); ++r) {
            const buf_d2 h = buf_load2(rh, lane16, 4096u * (CUBE2 ? r : fft::dr(r)));
            hr[r] = h.x;
            hi[r] = h.y;
        }
        OSZ_FSTAMP(3);   // pass 2
        if (CUBE2) wave_lds_fence();
        else __syncthreads();
        OSZ_FSTAMP(4);   // barrier 2
        if (CUBE2) fft::cube2::f3(t, re, im, L);
        else fft::cube::f3(t, re, im, L);
        OSZ_FSTAMP(5);   // pass 3
        if (PF2 && REQ) wait_spectrum();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (PF2 && REQ) fft::cube::cmul(re[r], im[r], Hn[PF2 ? r : 0].x, Hn[PF2 ? r : 0].y);
            else if (HPRE < 0) fft::cube::cmul(re[r], im[r], Hr[HPRE < 0 ? r : 0], Hi[HPRE < 0 ? r : 0]);
            else if (r < HPRE) fft::cube::cmul(re[r], im[r], hr[r < HPRE ? r : 0], hi[r < HPRE ? r : 0]);
            else {
                const buf_d2 h = buf_load2(rh, lane16, 4096u * (CUBE2 ? r : fft::dr(r)));
                fft::cube::cmul(re[r], im[r], h.x, h.y);
            }
        }
        // unconditional requests: nx must be dead above this line
        if (PF && REQ) request_next(next_blk);
        OSZ_FSTAMP(6);   // filter spectrum: loads + multiply
        if (CUBE2) fft::cube2::i3(t, re, im, L);
        else fft::cube::i3(t, re, im, L);
        OSZ_FSTAMP(7);   // inverse pass 3
        if (CUBE2) wave_lds_fence();
        else __syncthreads();
        OSZ_FSTAMP(8);   // barrier 3
        if (CUBE2) fft::cube2::i2(t, re, im, tw2, L);
        else fft::cube::i2(t, re, im, tw2, L);
        OSZ_FSTAMP(9);   // inverse pass 2
        __syncthreads();
        OSZ_FSTAMP(10);  // barrier 4
        if (CUBE2) fft::cube2::i1(t, re, im, tw1, L);
        else fft::cube::i1(t, re, im, tw1, L);
    }

    // a pair of whole blocks: no predication anywhere.  FROM_NX: the samples were
    // requested by the previous pair (PF); the first whole pair of a run loads
    // its own, so that on EVERY path into the steady loop the requested samples
    // are followed by one pair's stores and the compiler's wait at the loop top
    // is vmcnt(stores), not vmcnt(0).
    template <bool FROM_NX = false>
    __device__ __forceinline__ void fast_pair(int64_t blk) {
        const int64_t start_a = blk * a.step;
        double re[16], im[16];
        const unsigned lane8 = 8u * (unsigned)t;
        const __amdgpu_buffer_rsrc_t rx = buf_rsrc(xr + start_a);
        // first whole pair of the run: its spectrum goes out ahead of its (compiler
        // tracked) sample loads -- older than them, landed when they have
        if (PF2 && !FROM_NX) request_spectrum();
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (PF && FROM_NX) {
                re[j] = j < NR ? nx[PF ? (j < NR ? j : 0) : 0] : 0.0;
                im[j] = j < NR ? nx[PF ? (j < NR ? NR + j : 0) : 0] : 0.0;
            } else {
                re[j] = j < NR ? buf_load(rx, lane8, 2048u * j) : 0.0;
                im[j] = j < NR ? buf_load(rx, lane8, 2048u * (j + NR)) : 0.0;
            }
        }
        if (PF2 && !FROM_NX) {
            // first whole pair: everything requested so far has landed before pass 1
            asm volatile("s_waitcnt vmcnt(0) ; osz:hn" ::: "memory");
#pragma unroll
            for (int r = 0; r < (PF2 ? 16 : 0); ++r) asm volatile("" : "+v"(Hn[PF2 ? r : 0]));
        }
        OSZ_FSTAMP(0);   // sample loads issued (and the previous pair's stores)
        transform<true>(re, im);
        // re[j] = a[256 j + t], im[j] = b[256 j + t]
#pragma unroll
        for (int j = 0; j < NT_; ++j) {
            const int p = 256 * j + t;
            if (p < wm1) {
                re[j] += cr[j];
                im[j] += re[j + NR];
                cr[j] = im[j + NR];
            }
        }
        if (PF2) request_spectrum();   // the next pair's, ahead of the stores
        double *qa = yr + (start_a - a.skip) + t;
        if (a.accum) {   // a piece of a partitioned filter adds into the work row
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                qa[256 * j] += re[j];
                qa[256 * (j + NR)] += im[j];
            }
        } else {
            const __amdgpu_buffer_rsrc_t ry = buf_rsrc(yr + (start_a - a.skip));
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                buf_store(re[j], ry, lane8, 2048u * j);
                buf_store(im[j], ry, lane8, 2048u * (j + NR));
            }
        }
        OSZ_FSTAMP(11);  // inverse pass 1 + overlap add + stores issued
        wait_next();     // (the stamped build books this wait under the next pair's slot 0)
    }

    // any pair: ragged lengths, left cut, accumulate
    __device__ __forceinline__ void any_pair(int64_t blk) {
        double *scratch = reinterpret_cast<double *>(L);   // a's tail,
        double *carry = scratch + 2048;                    // b's tail (<= 2048 doubles each)
        const int64_t start_a = blk * a.step;
        const int64_t rem_a = a.n - start_a;
        const int len_a = rem_a < a.step ? (int)rem_a : a.step;
        const int64_t start_b = start_a + len_a;
        int len_b = 0;
        if (blk + 1 < blk1) {
            const int64_t rem_b = a.n - start_b;
            len_b = rem_b < a.step ? (int)rem_b : a.step;
        }
        double re[16], im[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int p = 256 * j + t;
            re[j] = p < len_a ? xr[start_a + p] : 0.0;
            im[j] = p < len_b ? xr[start_b + p] : 0.0;
        }
        transform(re, im);
        __syncthreads();   // every thread is done reading the cube
        const int ja0 = len_a >> 8, ja1 = (len_a + wm1 + 255) >> 8;
        const int jb0 = len_b >> 8, jb1 = (len_b + wm1 + 255) >> 8;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int p = 256 * j + t;
            if (j < NT_) {
                if (p < wm1) re[j] += cr[j < NT_ ? j : 0];
            }
            if (j >= ja0 && j < ja1) {
                const int q = p - len_a;
                if (q >= 0 && q < wm1) scratch[q] = re[j];
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int p = 256 * j + t;
            if (j < NT_) {
                if (p < wm1) im[j] += scratch[p];
            }
            if (j >= jb0 && j < jb1) {
                const int q = p - len_b;
                if (q >= 0 && q < wm1) carry[q] = im[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int p = 256 * j + t;
            const int64_t oa = start_a + p - a.skip;
            if (p < len_a && oa >= 0) yr[oa] = a.accum ? yr[oa] + re[j] : re[j];
            const int64_t ob = start_b + p - a.skip;
            if (p < len_b && ob >= 0) yr[ob] = a.accum ? yr[ob] + im[j] : im[j];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NT_; ++j) {
            const int p = 256 * j + t;
            cr[j] = p < wm1 ? carry[p] : 0.0;
        }
        __syncthreads();   // before the next transform overwrites the cube
    }
};

}  // namespace osz
