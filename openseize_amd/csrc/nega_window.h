// nega_window.h -- one real block of up to 8192 samples through fft::nega (fft4096.h) inside a
// 256-thread workgroup: what the zero-phase chain kernel (chain_zpn_body.h) and the overlap-add
// FIR kernel on one block per transform (fir.hip: fir_nega_kernel) share -- the window's
// transform with the filter's spectrum applied, and the request of a block's rows by LDS-DMA
// (zp_request_rows, chain_zp.h).
#pragma once

#include "common.h"
#include "fft4096.h"
#include "fir_pair.h"

namespace osz {

// In-kernel phase stamps for the diagnostic build only (benchmarks/zpn_stamps.hip defines
// OSZ_NEGA_STAMPS); the library build has none.  Slots: see that file.
#ifdef OSZ_NEGA_STAMPS
__device__ unsigned long long *g_nega_stamps = nullptr;   // [waves][24] cycle sums
#define OSZ_NSTAMP(acc_, last_, slot_)                                                \
    do {                                                                              \
        unsigned long long now_;                                                      \
        __builtin_amdgcn_sched_barrier(0);                                            \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");  \
        __builtin_amdgcn_sched_barrier(0);                                            \
        (acc_)[slot_] += now_ - (last_);                                              \
        (last_) = now_;                                                               \
    } while (0)
#else
#define OSZ_NSTAMP(acc_, last_, slot_) do { } while (0)
#endif

// A block's (a pair's) rows of 256 samples, requested by LDS-DMA into the cube.  A wave
// owns, in every 4 KB plane of the cube, the 1 KB piece [1024 w, 1024 w + 1024) -- the slots it
// reads last in a transform (inverse pass 1, view A) and writes first in the next one (pass 1) --
// so between the two nobody else touches it: piece m takes the wave's 64 samples of rows 2 m and
// 2 m + 1 (one 16-byte request per lane: lanes 0-31 row 2 m, lanes 32-63 row 2 m + 1), and the
// wave reads its own requests back behind its own vmcnt wait -- no barrier, no registers held
// while the samples are on their way.
// NP pieces = 2 NP rows from src on; rows >= nrows (an odd count's last half piece) lie behind the
// descriptor's range: their lanes request nothing.
template <int NP>
__device__ __forceinline__ void zp_request_rows(const double *src, int nrows, int t, const void *cube) {
    int tq = t;
    asm volatile("" : "+v"(tq));     // per block, not hoisted
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(src), 0, 2048 * nrows, 0x00020000);
    const unsigned voff = 2048u * (((unsigned)tq >> 5) & 1u) + 512u * ((unsigned)tq >> 6) + 16u * ((unsigned)tq & 31u);
    const unsigned ldsb = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<uintptr_t>(cube) +
                                                         1024u * ((unsigned)tq >> 6));
#pragma unroll
    for (int m = 0; m < NP; ++m) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_mov_b32 m0, %1\n\t"
                     "s_nop 0\n\t"
#ifdef OSZ_NO_NT      // (A/B builds only)
                     "buffer_load_dwordx4 %2, %3, 0 offen lds\n\t"
#else
                     "buffer_load_dwordx4 %2, %3, 0 offen nt lds\n\t"     // (read once: non-temporal)
#endif
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"(ldsb + 4096u * m), "v"(voff + 4096u * m), "s"(rx)   // (the range check sees the lane offset)
                     : "memory");
    }
}



// the transform of a block's window in place: pack, forward, x spectrum, inverse, unpack
template <int NHI>
struct NegaWindow {
    using C2 = fft::cube::C2;
    const FirArgs &a;
    C2 *L;
    const C2 *tw2l;               // [4 q][16 n0]: W256^(n0 2^q) in LDS (sixteen distinct rows: not worth registers)
    fft::nega::TwPowN tw1;
#ifdef OSZ_NEGA_STAMPS
    unsigned long long sacc[24], slast;
#endif
#ifndef OSZ_NEGA_STAMPS
#define OSZ_WSTAMP(slot_) do { } while (0)
#else
#define OSZ_WSTAMP(slot_) OSZ_NSTAMP(sacc, slast, slot_)
#endif

    __device__ __forceinline__ void load_tw2(int t, fft::cube::TwPow &w) const {
        const C2 *p = tw2l + (t & 15);        // [q][n0]: a 16-lane row reads 16 consecutive slots
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const C2 v = p[16 * q];
            w.r[q] = v.re;
            w.i[q] = v.im;
        }
    }

    // next: where the NEXT block's rows begin (nrows of them), or null -- requested by LDS-DMA as
    // soon as inverse pass 1 has read this wave's pieces of the cube, ahead of its arithmetic:
    // they land while that, the caller's epilogue and its stores run
    template <int NP = 1>
    __device__ __forceinline__ void transform(int t_in, double *re, double *im, const double *next = nullptr,
                                              int nrows = 0) {
        // LDS slot numbers are recomputed per block from an opaque copy of the thread index:
        // hoisted out of the loop they would pin registers
        int t = t_in;
        asm volatile("" : "+v"(t));
        fft::nega::f1<NHI>(t, re, im, tw1, L);
        OSZ_WSTAMP(1);    // pack + pass 1 + stores
        __syncthreads();
        OSZ_WSTAMP(2);    // barrier 1
        fft::cube::TwPow tw2;
        load_tw2(t, tw2);
        fft::cube2::f2(t, re, im, tw2, L);
        // the spectrum of the pair's sixteen bins is requested before the fence (from L2: resident
        // it spills); base and bin row in scalar registers, one 32-bit lane offset (buf_rsrc)
        double hr[16], hi[16];
        const unsigned lane16 = 16u * ((unsigned)t & 255u);
        const __amdgpu_buffer_rsrc_t rh = buf_rsrc(a.H);
#ifdef OSZ_ABL_NOH      // (diagnostic builds: no spectrum loads)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            hr[r] = 1.0 / 4096 + 1e-9 * r;
            hi[r] = 1e-7 * (double)(t & 3);
        }
        (void)rh;
        (void)lane16;
#else
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const buf_d2 h = buf_load2(rh, lane16, 4096u * r);
            hr[r] = h.x;
            hi[r] = h.y;
        }
#endif
        OSZ_WSTAMP(3);    // pass 2 + spectrum requests
        wave_lds_fence();
        fft::cube2::f3(t, re, im, L);
        OSZ_WSTAMP(4);    // pass 3
#pragma unroll
        for (int r = 0; r < 16; ++r) fft::cube::cmul(re[r], im[r], hr[r], hi[r]);
        OSZ_WSTAMP(5);    // spectrum lands + multiply
        fft::cube2::i3(t, re, im, L);
        wave_lds_fence();
        OSZ_WSTAMP(6);    // inverse pass 3
        load_tw2(t, tw2);
        fft::cube2::i2(t, re, im, tw2, L);
        OSZ_WSTAMP(7);    // inverse pass 2
        __syncthreads();
        OSZ_WSTAMP(8);    // barrier 4
        fft::nega::i1_load(t, re, im, L);
#ifndef OSZ_ABL_NODMA   // (diagnostic builds: no row requests)
        if (next) {
            asm volatile("s_waitcnt lgkmcnt(0) ; osz:dma" ::: "memory");
            zp_request_rows<NP>(next, nrows, t_in, L);
        }
#endif
        OSZ_WSTAMP(9);    // inverse pass 1's loads + the next block's requests
        fft::nega::i1_finish(re, im, tw1);
        OSZ_WSTAMP(10);   // inverse pass 1 + unpack
    }
};


}  // namespace osz
