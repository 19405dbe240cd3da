// spec.hip -- K5/K6: streaming segmenter + detrend + window + real FFT
// (+ power, one-sided doubling and segment averaging) on gfx950.
//
// Replaces, per nfft-sample segment, the reference's call chain
//   _spectra_estimatives FIFO   src/openseize/core/numerical.py:799-849
//   scipy.signal.detrend        :691     get_window * x   :694-695
//   np.fft.rfft                 :699     X *= sqrt(norm)  :703-716
//   periodogram re^2+im^2, x2   :782-794
//   psd running mean            src/openseize/spectra/estimators.py:149-152
//
// General-nfft path (this file): a "prep" kernel gathers every complete
// segment from (carry ++ chunk), removes the trend, applies the window and
// writes one row per (segment, channel); rocFFT runs ONE batched r2c over
// all rows; a "post" kernel scales and either stores the complex DFT, stores
// the periodogram, or folds the periodograms of the batch into the running
// sum in a fixed order (deterministic, no atomics).  The FIFO of the
// reference becomes a (nch, < nfft) carry buffer on the device.
#include <rocfft/rocfft.h>

#include <cmath>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include "common.h"
#include "fft4096.h"
#include "fft8.h"
#include "specmix.h"
#include "specsplit.h"
#include "nega_window.h"

namespace osz {

int get_fft_tables(fft::Tables &out);  // fir.hip

#define OSZ_FFT(call)                                                               \
    do {                                                                            \
        rocfft_status s_ = (call);                                                  \
        if (s_ != rocfft_status_success)                                            \
            return osz::fail(OSZ_ERR_HIP, "%s: rocfft status %d (%s:%d)", #call, (int)s_, \
                             __FILE__, __LINE__);                                   \
    } while (0)

struct PrepArgs {
    const double *x;      // chunk (nch, n)
    const double *carry;  // (nch, ncap): first ncarry columns valid
    double *rows;         // (nseg*nch, nfft)
    const double *window;
    int64_t ldx;
    int64_t ncarry, ncap;
    int64_t seg0;         // first segment of this batch
    int nwin, nfft, stride, nch, detrend;  // nwin samples per segment, zero-padded to nfft
};

__device__ __forceinline__ bool spec_finite(double v) { return __builtin_isfinite(v); }
__device__ __forceinline__ double spec_nan() { return __longlong_as_double(0x7ff8000000000000LL); }

__device__ __forceinline__ double block_sum(double v, double *red) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
    return s;
}

// grid: (nseg_batch, nch); block 256
__global__ __launch_bounds__(256) void spec_prep_kernel(PrepArgs a) {
    __shared__ double red[4];
    const int seg = blockIdx.x;
    const int c = blockIdx.y;
    const int64_t v0 = (a.seg0 + seg) * (int64_t)a.stride;  // virtual start in carry ++ chunk
    const double *cr = a.carry + (int64_t)c * a.ncap;
    const double *xr = a.x + (int64_t)c * a.ldx;
    double *row = a.rows + ((int64_t)seg * a.nch + c) * a.nfft;
    auto sample = [&](int i) -> double {
        const int64_t v = v0 + i;
        return v < a.ncarry ? cr[v] : xr[v - a.ncarry];
    };
    // trend: mean, and for 'linear' the least-squares slope about the centre
    double s0 = 0.0, s1 = 0.0;
    const double mid = 0.5 * (a.nwin - 1);
    for (int i = threadIdx.x; i < a.nwin; i += 256) {
        const double v = sample(i);
        s0 += v;
        s1 += (i - mid) * v;
    }
    const double mean = block_sum(s0, red) / a.nwin;
    double slope = 0.0;
    if (a.detrend == OSZ_DETREND_LINEAR) {
        const double sxy = block_sum(s1, red);
        // sum (i - mid)^2 = n (n^2 - 1) / 12
        const double sxx = (double)a.nwin * ((double)a.nwin * a.nwin - 1.0) / 12.0;
        slope = sxx > 0 ? sxy / sxx : 0.0;
    }
    for (int i = threadIdx.x; i < a.nfft; i += 256)
        row[i] = i < a.nwin ? (sample(i) - mean - slope * (i - mid)) * a.window[i] : 0.0;
}

struct PostArgs {
    const double *spec;  // (nseg*nch, nfreq) complex interleaved
    double *out;         // SEGMENTS: (nseg, nch, nfreq) f64 or c128; MEAN: accumulator (nch, nfreq)
    int64_t nseg;
    int nfreq, nch, nfft, mode;
    double scale;
};

// grid: (ceil(nfreq/256), nch)
__global__ void spec_post_kernel(PostArgs a) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (k >= a.nfreq) return;
    // one-sided doubling: every bin but DC and (even nfft) Nyquist
    const bool dbl = (k != 0) && !((a.nfft % 2 == 0) && k == a.nfreq - 1);
    double acc = 0.0;
    for (int64_t s = 0; s < a.nseg; ++s) {
        const int64_t idx = ((int64_t)s * a.nch + c) * a.nfreq + k;
        const double re = a.spec[2 * idx], im = a.spec[2 * idx + 1];
        if (a.mode == OSZ_SPEC_DFT_SEGMENTS) {
            a.out[2 * idx] = re * a.scale;
            a.out[2 * idx + 1] = im * a.scale;
        } else {
            const double sr = re * a.scale, si = im * a.scale;
            double p = sr * sr + si * si;
            if (dbl) p *= 2.0;
            if (a.mode == OSZ_SPEC_PSD_SEGMENTS)
                a.out[idx] = p;
            else
                acc += p;
        }
    }
    if (a.mode == OSZ_SPEC_PSD_MEAN) a.out[(int64_t)c * a.nfreq + k] += acc;
}

// newcarry = virtual[consumed : total)
__global__ void spec_carry_kernel(const double *x, int64_t ldx, const double *carry,
                                  double *newcarry, int64_t ncap, int64_t ncarry,
                                  int64_t consumed, int64_t nnew) {
    const int c = blockIdx.y;
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nnew) return;
    const int64_t v = consumed + q;
    newcarry[(int64_t)c * ncap + q] =
        v < ncarry ? carry[(int64_t)c * ncap + v] : x[(int64_t)c * ldx + (v - ncarry)];
}


// ---------------------------------------------------------------------------
// Fused on-chip path for nwin == nfft == 4096 (cfg-4 / cfg-5): one workgroup
// walks a run of consecutive segment PAIRS of one channel.  The two real
// segments of a pair ride one 4096-point complex transform (fft4096.h);
// detrend and window are applied in registers on the way in, the two spectra
// are separated with one LDS exchange (A[k] = (Z[k] + conj Z[N-k]) / 2,
// B[k] = (Z[k] - conj Z[N-k]) / 2i), and scaling, |.|^2, one-sided doubling
// and the segment sum happen in registers.  No staging rows: HBM sees each
// sample once (the 50 % overlap re-read of the neighbouring pair hits L2) and,
// for the segment modes, each output bin once.
struct FusedArgs {
    const double *x;
    const double *carry;
    const double *window;   // 4096
    void *out;              // SEGMENTS modes: (nseg, nch, 2049) f64 / c128
    double *partial;        // PSD_MEAN: (nch, nruns, 2049) sums of this launch
    int64_t ldx, ncarry, ncap;
    int64_t nseg;           // complete segments in carry ++ chunk
    int stride, nch, detrend, R, nruns;
    double scale;
    fft::Tables tb;
};

constexpr int kNF = 2049;   // nfft / 2 + 1

// The fused pass runs on the cube layout of fft4096.h (interleaved complex,
// in-place exchanges, resident twiddle powers and window): four barriers per
// pair of segments.  With HALF (stride == 2048, the 50 % overlap of cfg-4/5)
// segment b's first half IS segment a's second half in the same registers and
// b's second half is the next pair's first half, so a thread loads 16 new
// samples per pair instead of 32: each sample goes through the vector-memory
// path once.  (Requesting those 16 one pair ahead INTO REGISTERS was measured
// 14 % slower: the kernel is not latency-starved at two workgroups per CU, and
// the longer-lived registers cost more.  In the mean mode they now come by
// LDS-DMA into the wave's own pieces of the cube, which holds no register:
// nothing with the constant trend, 0.91 -> 0.83 ms with the linear one, whose
// registers were the tightest.)  The
// two spectra are separated through the view-C slots a thread already owns
// (Z[k] parked at slot_c(t, k >> 8), its mirror Z[N-k] read from thread
// 256 - t's slots).
template <int MODE, bool LINEAR, bool HALF>
__global__ __launch_bounds__(256, 2) void spec_cube_kernel(FusedArgs a) {
    using fft::cube::C2;
    extern __shared__ C2 cube_lds[];
    C2 *L = cube_lds;
    __shared__ double red[4][4];
    // pass 2's twiddles W256^(n0 k1), k1 = 1 .. 15, as a table [k1 - 1][n0] behind the cube: fifteen
    // products per pair where the power scheme (W^b, then (W^4)^a) takes twenty-six, and no
    // resident powers of that base (16 registers)
    C2 *tw2t = cube_lds + fft::cube::SLOTS;
    const int t = threadIdx.x;
    const int run = blockIdx.x;
    const int c = blockIdx.y;
    if (t < 240) {
        const int k1 = 1 + (t >> 4), n0 = t & 15;
        tw2t[t] = C2{a.tb.t2[(n0 * 16 + k1) * 2], a.tb.t2[(n0 * 16 + k1) * 2 + 1]};
    }
    const double *cr = a.carry + (int64_t)c * a.ncap;
    const double *xr = a.x + (int64_t)c * a.ldx;
    const int64_t npairs = (a.nseg + 1) / 2;
    const int64_t p0 = ((int64_t)run * npairs) / a.nruns;
    const int64_t p1 = ((int64_t)(run + 1) * npairs) / a.nruns;
    const double mid = 0.5 * (fft::N - 1);
    const double s2 = a.scale * a.scale;

    fft::cube::TwPow tw1;
    {
        fft::cube::TwPow tw2;
        fft::cube::tw_load(t, a.tb, tw1, tw2);
    }
    double win[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) win[j] = a.window[256 * j + t];

    // PSD_MEAN: the sum over a pair's two segments needs no separation,
    //   |A[k]|^2 + |B[k]|^2 = (|Z[k]|^2 + |Z[N-k]|^2) / 2   (Z = A + i B, a and b real),
    // so a thread sums |Z|^2 of the 16 bins it owns over its whole run and the bins
    // meet their mirrors ONCE, after the run: no park / barrier / mirror read per pair.
    double acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.0;
    double keep[HALF ? 8 : 1];
    bool have_keep = false;
    double half_sum = 0.0;          // this thread's sum over the half the next pair begins with
    bool have_sum = false;
    // mean mode at 50 % overlap: the next pair's sixteen new rows are requested by LDS-DMA into
    // this wave's own pieces of the cube (nega_window.h: zp_request_rows) as soon as pass 3 has
    // read them -- view C and view A of fft::cube are the same 1 KB piece per plane and wave --
    // and picked up at the top of the next pair: no registers held, no load latency there
    bool pending = false;

    auto ld = [&](int64_t v) { return v < a.ncarry ? cr[v] : xr[v - a.ncarry]; };

    double re[16], im[16];
    for (int64_t p = p0; p < p1; ++p) {
        const int64_t sa = 2 * p, sb = 2 * p + 1;
        const bool has_b = sb < a.nseg;
        const int64_t va = sa * (int64_t)a.stride, vb = va + a.stride;
        // ---- load both segments (virtual stream = carry ++ chunk)
        if (HALF && pending) {
            asm volatile("s_waitcnt vmcnt(0) ; osz:dma rows" ::: "memory");
            const char *mine = reinterpret_cast<const char *>(L) + 1024 * (t >> 6) + 8 * (t & 63);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                re[j] = keep[HALF ? j : 0];
                re[8 + j] = *reinterpret_cast<const double *>(mine + 4096 * (j >> 1) + 512 * (j & 1));
                im[j] = re[8 + j];
                im[8 + j] = *reinterpret_cast<const double *>(mine + 4096 * (4 + (j >> 1)) + 512 * (j & 1));
                keep[HALF ? j : 0] = im[8 + j];
            }
            have_keep = true;
        } else if (HALF) {
            const bool chunk_only = va + (have_keep ? 2048 : 0) >= a.ncarry;
            const double *q = xr + (va - a.ncarry) + t;
            if (have_keep) {
#pragma unroll
                for (int j = 0; j < 8; ++j) re[j] = keep[HALF ? j : 0];
            } else if (chunk_only) {
#pragma unroll
                for (int j = 0; j < 8; ++j) re[j] = q[256 * j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) re[j] = ld(va + 256 * j + t);
            }
            if (chunk_only) {
#pragma unroll
                for (int j = 0; j < 8; ++j) re[8 + j] = q[2048 + 256 * j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) re[8 + j] = ld(va + 2048 + 256 * j + t);
            }
            if (has_b) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    im[j] = re[8 + j];
                    im[8 + j] = chunk_only ? q[4096 + 256 * j] : ld(va + 4096 + 256 * j + t);
                    keep[HALF ? j : 0] = im[8 + j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) im[j] = 0.0;
            }
            have_keep = has_b;
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int i = 256 * j + t;
                re[j] = ld(va + i);
                im[j] = has_b ? ld(vb + i) : 0.0;
            }
        }
        // ---- trend: block sums
        double sum_a = 0.0, sum_b = 0.0, lin_a = 0.0, lin_b = 0.0;
        if (HALF && !LINEAR) {
            // at 50 % overlap a segment is two halves, and every half belongs to two segments: a
            // thread sums each half ONCE (the first half of segment a was the second half of the
            // previous pair's segment b)
            double s0 = half_sum, s1 = 0.0, s2 = 0.0;
            if (!have_sum) {
                s0 = 0.0;
#pragma unroll
                for (int j = 0; j < 8; ++j) s0 += re[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s1 += re[8 + j];
                s2 += im[8 + j];
            }
            sum_a = s0 + s1;
            sum_b = has_b ? s1 + s2 : 0.0;
            half_sum = s2;
            have_sum = has_b;
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int i = 256 * j + t;
                sum_a += re[j];
                sum_b += im[j];
                if (LINEAR) {
                    lin_a += (i - mid) * re[j];
                    lin_b += (i - mid) * im[j];
                }
            }
        }
        sum_a = wave_sum63(sum_a);
        sum_b = wave_sum63(sum_b);
        if (LINEAR) {
            lin_a = wave_sum63(lin_a);
            lin_b = wave_sum63(lin_b);
        }
        if ((t & 63) == 63) {
            red[t >> 6][0] = sum_a;
            red[t >> 6][1] = sum_b;
            red[t >> 6][2] = lin_a;
            red[t >> 6][3] = lin_b;
        }
        __syncthreads();   // also: every mirror read of the previous pair is done
        const double mean_a = (red[0][0] + red[1][0] + red[2][0] + red[3][0]) / fft::N;
        const double mean_b = (red[0][1] + red[1][1] + red[2][1] + red[3][1]) / fft::N;
        double slope_a = 0.0, slope_b = 0.0;
        if (LINEAR) {
            const double sxx = (double)fft::N * ((double)fft::N * fft::N - 1.0) / 12.0;
            slope_a = (red[0][2] + red[1][2] + red[2][2] + red[3][2]) / sxx;
            slope_b = (red[0][3] + red[1][3] + red[2][3] + red[3][3]) / sxx;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int i = 256 * j + t;
            if (LINEAR) {
                re[j] = (re[j] - mean_a - slope_a * (i - mid)) * win[j];
                im[j] = (im[j] - mean_b - slope_b * (i - mid)) * win[j];
            } else {
                re[j] = (re[j] - mean_a) * win[j];
                im[j] = (im[j] - mean_b) * win[j];
            }
        }
        if (!has_b) {       // (a run's last, odd segment: a scalar branch, not sixteen selects per pair)
#pragma unroll
            for (int j = 0; j < 16; ++j) im[j] = 0.0;
        }
        // A non-finite sample costs the reference ITS segment (detrend, window, rfft of a segment:
        // core/numerical.py:691-716); here two segments ride one transform, so the one that is not
        // finite goes in as zeros -- the other's spectrum stays what it is -- and comes out as NaN.
        // (The average, MODE 0, is lost either way: nothing to keep apart.)
        const bool bad_a = MODE != OSZ_SPEC_PSD_MEAN && !spec_finite(mean_a + slope_a);
        const bool bad_b = MODE != OSZ_SPEC_PSD_MEAN && has_b && !spec_finite(mean_b + slope_b);
        if (MODE != OSZ_SPEC_PSD_MEAN && bad_a != bad_b) {
            if (bad_a) {
#pragma unroll
                for (int j = 0; j < 16; ++j) re[j] = 0.0;
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) im[j] = 0.0;
            }
        }
        // ---- forward transform of a + i b
        int tt = t;   // opaque copy: keeps the LDS slot numbers out of loop-invariant registers
        asm volatile("" : "+v"(tt));
        fft::cube::f1(tt, re, im, tw1, L);
        __syncthreads();
        {
            // pass 2 (fft::cube::f2) with its twiddles from the table
            const int base = fft::cube::base_b(tt);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const C2 v = L[base + 16 * j];
                re[j] = v.re;
                im[j] = v.im;
            }
            fft::fwd16(re, im);
            const C2 *twp = tw2t + (tt >> 4);
#pragma unroll
            for (int r = 1; r < 16; ++r) {
                const int k1 = fft::dr(r);          // (register 0 holds k1 = 0)
                const C2 w = twp[16 * (k1 - 1)];
                fft::cube::cmul(re[r], im[r], w.re, w.im);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) L[base + 16 * fft::dr(r)] = C2{re[r], im[r]};
        }
        __syncthreads();
        if (MODE == OSZ_SPEC_PSD_MEAN && HALF) {
            // pass 3 with the request between its loads and its butterflies
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const C2 v = L[fft::cube::slot_c(tt, j)];
                re[j] = v.re;
                im[j] = v.im;
            }
            pending = false;
            if (p + 1 < p1) {
                const int64_t sa2 = 2 * (p + 1), va2 = sa2 * (int64_t)a.stride;
                const double *src = xr + (va2 - a.ncarry) + 2048;
                // a whole pair, all of it in the chunk, rows on 16-byte addresses
                if (sa2 + 1 < a.nseg && va2 + 2048 >= a.ncarry && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
                    asm volatile("s_waitcnt lgkmcnt(0) ; osz:dma" ::: "memory");
                    zp_request_rows<8>(src, 16, t, L);
                    pending = true;
                }
            }
            fft::fwd16(re, im);
        } else {
            fft::cube::f3(tt, re, im, L);
        }
        if (MODE == OSZ_SPEC_PSD_MEAN) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = fma(re[r], re[r], fma(im[r], im[r], acc[r]));
            continue;   // (the next pair's first barrier also ends this pair's cube reads)
        }
        // ---- separate the two spectra: park Z[k] in this thread's own slots
#pragma unroll
        for (int r = 0; r < 16; ++r) L[fft::cube::slot_c(tt, fft::dr(r))] = C2{re[r], im[r]};
        __syncthreads();
        const int tp = (256 - tt) & 255;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = fft::dr(r);
            if (j > 8) continue;
            const int k = t + 256 * j;
            if (j == 8 && t != 0) continue;
            const int jp = t == 0 ? ((16 - j) & 15) : 15 - j;   // N - k = tp + 256 jp
            const C2 q = L[fft::cube::slot_c(tp, jp)];
            const double zr = re[r], zi = im[r], qr = q.re, qi = q.im;
            double ar = 0.5 * (zr + qr), ai = 0.5 * (zi - qi);
            double br = 0.5 * (zi + qi), bi = -0.5 * (zr - qr);
            if (bad_a) ar = ai = spec_nan();
            if (bad_b) br = bi = spec_nan();
            const bool dbl = (k != 0) && (k != fft::N / 2);
            if (MODE == OSZ_SPEC_DFT_SEGMENTS) {
                double *o = (double *)a.out;
                const int64_t ia = ((sa * a.nch + c) * (int64_t)kNF + k) * 2;
                o[ia] = ar * a.scale;
                o[ia + 1] = ai * a.scale;
                if (has_b) {
                    const int64_t ib = ((sb * a.nch + c) * (int64_t)kNF + k) * 2;
                    o[ib] = br * a.scale;
                    o[ib + 1] = bi * a.scale;
                }
            } else {
                const double f = dbl ? 2.0 * s2 : s2;
                const double pa = (ar * ar + ai * ai) * f;
                const double pb = (br * br + bi * bi) * f;
                if (MODE == OSZ_SPEC_PSD_SEGMENTS) {
                    double *o = (double *)a.out;
                    o[(sa * a.nch + c) * (int64_t)kNF + k] = pa;
                    if (has_b) o[(sb * a.nch + c) * (int64_t)kNF + k] = pb;
                }
            }
        }
    }
    if (MODE == OSZ_SPEC_PSD_MEAN) {
        // fold bin k with its mirror N - k; one-sided doubling and the scale here
        double *D = reinterpret_cast<double *>(L);
        __syncthreads();   // the last pair's cube reads are done
#pragma unroll
        for (int r = 0; r < 16; ++r) D[t + 256 * fft::dr(r)] = acc[r];
        __syncthreads();
        double *o = a.partial + ((int64_t)c * a.nruns + run) * kNF;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = t + 256 * j;
            const double f = k != 0 ? 2.0 * s2 : s2;
            o[k] = 0.5 * (D[k] + D[(fft::N - k) & (fft::N - 1)]) * f;
        }
        if (t == 0) o[2048] = D[2048] * s2;
    }
}

// dsum[c][k] += sum over runs of partial[c][run][k], fixed order (deterministic)
__global__ void spec_partial_reduce_kernel(const double *partial, double *dsum, int nruns) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (k >= kNF) return;
    double s = 0.0;
    for (int r = 0; r < nruns; ++r) s += partial[((int64_t)c * nruns + r) * kNF + k];
    dsum[(int64_t)c * kNF + k] += s;
}

// ---------------------------------------------------------------------------
// Generic on-chip path for nfft = 512 ... 8192 (powers of two), nwin <= nfft,
// any stride: the same scheme as spec_cube_kernel -- two real segments per
// complex transform, detrend and window in registers, spectra separated with
// one LDS exchange, power / scaling / segment sums in registers -- on the
// 8-point-per-thread transforms of fft8.h: nfft / 8 threads per workgroup, about
// half the registers per thread of the 4096-point cube kernel, so twice the
// waves share a CU.  nfft = int(fs / resolution) of the reference
// (spectra/estimators.py:144) is whatever the user's fs makes it; the powers of
// two in this range stay on chip, everything else goes through rocFFT.
struct Spec8Args {
    const double *x;        // one contiguous source: segment s starts at column s * stride
    const double *window;   // nwin
    void *out;              // SEGMENTS modes: (nseg, nch, nfreq) f64 / c128
    double *partial;        // PSD_MEAN: (nch, nruns, nfreq) sums of this launch
    const double *tab;      // W_8192^j, j < 1024
    int64_t ldx;
    int64_t nseg;
    int stride, nwin, nch, nruns;
    double scale;
};

// All forward stages of one transform, a workgroup barrier between them.  The
// thread index is made opaque again before every stage: LDS slot numbers
// computed ahead of their stage would sit in registers the kernel does not have.
template <int N, int S>
struct Fwd8 {
    static __device__ __forceinline__ void run(int tid, double *re, double *im,
                                               const fft8::Twid<N> &tw, fft8::C2 *lds) {
        double wr = tw.wr[S], wi = tw.wi[S];
        asm volatile("" : "+v"(tid), "+v"(wr), "+v"(wi));   // no hoisting of slot numbers / twiddle powers
        fft8::fwd_stage<N, S>(tid, re, im, wr, wi, lds);
        if constexpr (S + 1 < fft8::Plan<N>::NS) {
            __syncthreads();
            Fwd8<N, S + 1>::run(tid, re, im, tw, lds);
        }
    }
};

// HALF (stride == nfft / 2 == nwin / 2, the default 50 % overlap): segment b's
// first half IS segment a's second half and b's second half is the next pair's
// first half, in the same registers of the same thread -- 8 new samples per
// thread and pair instead of 16.
template <int N, int MODE, bool LINEAR, bool HALF>
__global__ __launch_bounds__(N / 8, 4) void spec8_kernel(Spec8Args a) {
    using fft8::C2;
    constexpr int NT = N / 8, L = fft8::ilog2(N), NF = N / 2 + 1, NWV = (NT + 63) / 64;
    extern __shared__ C2 lds8[];
    __shared__ double red[NWV][4];
    const int t = threadIdx.x;
    const int run = blockIdx.x;
    const int c = blockIdx.y;
    const double *xr = a.x + (int64_t)c * a.ldx;
    const int64_t npairs = (a.nseg + 1) / 2;
    const int64_t p0 = ((int64_t)run * npairs) / a.nruns;
    const int64_t p1 = ((int64_t)(run + 1) * npairs) / a.nruns;
    const double mid = 0.5 * (a.nwin - 1);
    const double s2 = a.scale * a.scale;

    fft8::Twid<N> tw;
    fft8::twid_load<N>(t, a.tab, tw);
    double win[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int i = NT * r + t;
        const double wv = a.window[i < a.nwin ? i : 0];   // clamped address, no branch
        win[r] = i < a.nwin ? wv : 0.0;                   // zero padding up to nfft
    }
    // PSD_MEAN: |A[k]|^2 + |B[k]|^2 = (|Z[k]|^2 + |Z[N-k]|^2) / 2 -- a thread sums |Z|^2 of
    // its own 8 bins over the run; bins meet their mirrors once, after the run
    // (see spec_cube_kernel): no natural-order exchange per pair
    double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const bool fullwin = a.nwin == N;
    double keep[HALF ? 4 : 1];
    bool have_keep = false;

    for (int64_t p = p0; p < p1; ++p) {
        const int64_t sa = 2 * p, sb = 2 * p + 1;
        const bool has_b = sb < a.nseg;
        const int64_t va = sa * (int64_t)a.stride;
        double re[8], im[8];
        const double *pa = xr + va + t;
        if (HALF) {
            // rows 0..3 = first half, rows 4..7 = second half of a segment
            if (have_keep) {
#pragma unroll
                for (int r = 0; r < 4; ++r) re[r] = keep[HALF ? r : 0];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) re[r] = pa[NT * r];
            }
#pragma unroll
            for (int r = 4; r < 8; ++r) re[r] = pa[NT * r];
            if (has_b) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    im[r] = re[r + 4];
                    im[r + 4] = pa[NT * (r + 8)];
                    keep[HALF ? r : 0] = im[r + 4];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r) im[r] = 0.0;
            }
            have_keep = has_b;
        } else if (fullwin) {
#pragma unroll
            for (int r = 0; r < 8; ++r) re[r] = pa[NT * r];
            if (has_b) {
                const double *pb = pa + a.stride;
#pragma unroll
                for (int r = 0; r < 8; ++r) im[r] = pb[NT * r];
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r) im[r] = 0.0;
            }
        } else {
            // short window, zero padded to nfft: clamped index, zeros by select
            const int64_t db = has_b ? a.stride : 0;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int i = NT * r + t;
                const bool in = i < a.nwin;
                const int ii = in ? NT * r : -t;         // pa[ii] stays inside the segment
                const double xa = pa[ii], xb = pa[db + ii];
                re[r] = in ? xa : 0.0;
                im[r] = (in && has_b) ? xb : 0.0;
            }
        }
        // ---- trend: block sums over the nwin samples (padding holds zeros)
        double sum_a = 0.0, sum_b = 0.0, lin_a = 0.0, lin_b = 0.0;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int i = NT * r + t;
            sum_a += re[r];
            sum_b += im[r];
            if (LINEAR) {
                lin_a += (i - mid) * re[r];
                lin_b += (i - mid) * im[r];
            }
        }
        sum_a = wave_sum63(sum_a);
        sum_b = wave_sum63(sum_b);
        if (LINEAR) {
            lin_a = wave_sum63(lin_a);
            lin_b = wave_sum63(lin_b);
        }
        if ((t & 63) == 63) {
            red[t >> 6][0] = sum_a;
            red[t >> 6][1] = sum_b;
            red[t >> 6][2] = lin_a;
            red[t >> 6][3] = lin_b;
        }
        __syncthreads();   // also: every bin read of the previous pair is done
        double tot[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < NWV; ++q) {
            tot[0] += red[q][0];
            tot[1] += red[q][1];
            if (LINEAR) {
                tot[2] += red[q][2];
                tot[3] += red[q][3];
            }
        }
        const double mean_a = tot[0] / a.nwin, mean_b = tot[1] / a.nwin;
        double slope_a = 0.0, slope_b = 0.0;
        if (LINEAR) {
            const double nn = (double)a.nwin;
            const double sxx = nn * (nn * nn - 1.0) / 12.0;
            slope_a = sxx > 0.0 ? tot[2] / sxx : 0.0;
            slope_b = sxx > 0.0 ? tot[3] / sxx : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int i = NT * r + t;
            if (LINEAR) {
                re[r] = (re[r] - mean_a - slope_a * (i - mid)) * win[r];
                im[r] = has_b ? (im[r] - mean_b - slope_b * (i - mid)) * win[r] : 0.0;
            } else {
                re[r] = (re[r] - mean_a) * win[r];
                im[r] = has_b ? (im[r] - mean_b) * win[r] : 0.0;
            }
        }
        // (a segment that is not finite goes in as zeros and comes out as NaN: spec_cube_kernel)
        const bool bad_a = MODE != OSZ_SPEC_PSD_MEAN && !spec_finite(mean_a + slope_a);
        const bool bad_b = MODE != OSZ_SPEC_PSD_MEAN && has_b && !spec_finite(mean_b + slope_b);
        if (MODE != OSZ_SPEC_PSD_MEAN && bad_a != bad_b) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (bad_a) re[r] = 0.0;
                else im[r] = 0.0;
            }
        }
        // ---- forward transform of a + i b (one barrier per exchange)
        int tt = t;   // opaque copy: keeps the LDS slot numbers out of loop-invariant registers
        asm volatile("" : "+v"(tt));
        Fwd8<N, 0>::run(tt, re, im, tw, lds8);
        if (MODE == OSZ_SPEC_PSD_MEAN) {
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[r] = fma(re[r], re[r], fma(im[r], im[r], acc[r]));
            continue;   // (the next pair's first barrier also ends this pair's LDS reads)
        }
        // ---- natural-order exchange: Z[k] of both spectra side by side
        __syncthreads();   // every read of the last stage is done
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int r = 0; r < 8; ++r)
            lds8[fft8::swz_nat(fft8::revdigits<L>(fft8::idx_of<0>(tt, r)))] = C2{re[r], im[r]};
        __syncthreads();
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) {
            const int k = tt + NT * jj;                // jj = 4: the Nyquist bin, thread 0 only
            if (jj == 4 && t != 0) continue;
            const C2 z = lds8[fft8::swz_nat(k)];
            const C2 q = lds8[fft8::swz_nat((N - k) & (N - 1))];
            double ar = 0.5 * (z.re + q.re), ai = 0.5 * (z.im - q.im);
            double br = 0.5 * (z.im + q.im), bi = -0.5 * (z.re - q.re);
            if (bad_a) ar = ai = spec_nan();
            if (bad_b) br = bi = spec_nan();
            const bool dbl = (k != 0) && (k != N / 2);
            if (MODE == OSZ_SPEC_DFT_SEGMENTS) {
                double *o = (double *)a.out;
                const int64_t ia = ((sa * a.nch + c) * (int64_t)NF + k) * 2;
                o[ia] = ar * a.scale;
                o[ia + 1] = ai * a.scale;
                if (has_b) {
                    const int64_t ib = ((sb * a.nch + c) * (int64_t)NF + k) * 2;
                    o[ib] = br * a.scale;
                    o[ib + 1] = bi * a.scale;
                }
            } else {
                const double f = dbl ? 2.0 * s2 : s2;
                const double pa = (ar * ar + ai * ai) * f;
                const double pb = (br * br + bi * bi) * f;
                if (MODE == OSZ_SPEC_PSD_SEGMENTS) {
                    double *o = (double *)a.out;
                    o[(sa * a.nch + c) * (int64_t)NF + k] = pa;
                    if (has_b) o[(sb * a.nch + c) * (int64_t)NF + k] = pb;
                }
            }
        }
    }
    if (MODE == OSZ_SPEC_PSD_MEAN) {
        // fold bin k with its mirror N - k; one-sided doubling and the scale here
        double *D = reinterpret_cast<double *>(lds8);
        __syncthreads();   // the last pair's LDS reads are done
#pragma unroll
        for (int r = 0; r < 8; ++r) D[fft8::revdigits<L>(fft8::idx_of<0>(t, r))] = acc[r];
        __syncthreads();
        double *o = a.partial + ((int64_t)c * a.nruns + run) * NF;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int k = t + NT * jj;
            const double f = k != 0 ? 2.0 * s2 : s2;
            o[k] = 0.5 * (D[k] + D[(N - k) & (N - 1)]) * f;
        }
        if (t == 0) o[N / 2] = D[N / 2] * s2;
    }
}


// ---------------------------------------------------------------------------
// Any other transform length up to 4096 -- odd nfft, prime factors above 7: what
// nfft = int(fs / resolution) of the reference (spectra/estimators.py:144) is for a sampling
// rate like 173.61 Hz -- by Bluestein's chirp transform on the transforms of fft8.h:
//     X[k] = w[k] sum_j (x[j] w[j]) conj(w)[k - j],   w[j] = exp(-i pi j^2 / n),
// i.e. one circular convolution of length m = 2^L >= 2 n - 1 with a fixed sequence: a forward
// transform of x w (zero padded), a multiplication with the transformed chirp (host table, in
// the order the forward transform leaves its bins in, divided by m), an inverse transform,
// and w[k] on the way out.  Detrend, window, scaling, |.|^2, one-sided doubling and the segment
// sums as in spec8_kernel.  HBM sees each sample once and each
// output once, where the rocFFT route stages rows at 5-10 times that.
struct BlueArgs {
    const double *x;        // one contiguous source: segment s starts at column s * stride
    const double *window;   // nwin
    void *out;              // SEGMENTS modes: (nseg, nch, nfreq) f64 / c128
    double *partial;        // PSD_MEAN: (nch, nruns, nfreq) sums of this launch
    const double *tab;      // W_8192^j, j < 1024
    const double *chirp;    // [m][2]: w[j], j < n; zeros behind
    const double *bperm;    // [8][m / 8][2]: FFT_m(conj w, wrapped) / m at the bin register r of thread t holds
    int64_t ldx;
    int64_t nseg;
    int stride, nwin, nch, nruns, n, nfreq;
    double scale;
};

template <int N, int S>
struct Inv8 {
    static __device__ __forceinline__ void run(int tid, double *re, double *im,
                                               const fft8::Twid<N> &tw, fft8::C2 *lds) {
        double wr = tw.wr[S], wi = tw.wi[S];
        asm volatile("" : "+v"(tid), "+v"(wr), "+v"(wi));
        fft8::inv_stage<N, S>(tid, re, im, wr, wi, lds);
        if constexpr (S > 0) {
            __syncthreads();
            Inv8<N, S - 1>::run(tid, re, im, tw, lds);
        }
    }
};

// TWO real segments per pair of transforms: z = a + i b goes through as one complex sequence,
// X = A + i B.  The mean mode needs no separation,
//   |A[k]|^2 + |B[k]|^2 = (|X[k]|^2 + |X[n-k]|^2) / 2   (a, b real),
// so a thread sums |X|^2 of the bins it holds (k = NT r + t < n: r < 4) over its run and the
// bins meet their mirrors once, after the run (as in spec8_kernel); the segment modes park
// X[k], k < n, in LDS and read k and n - k: A = (X[k] + conj X[n-k]) / 2, B = (X[k] - conj X[n-k]) / 2i.
template <int N, int MODE, bool LINEAR>
__global__ __launch_bounds__(N / 8, 2) void spec_blue_kernel(BlueArgs a) {
    using fft8::C2;
    constexpr int NT = N / 8, NWV = (NT + 63) / 64;
    extern __shared__ C2 lds8[];
    __shared__ double red[NWV][4];
    const int t = threadIdx.x;
    const int run = blockIdx.x;
    const int c = blockIdx.y;
    const double *xr = a.x + (int64_t)c * a.ldx;
    const int64_t npairs = (a.nseg + 1) / 2;
    const int64_t p0 = ((int64_t)run * npairs) / a.nruns;
    const int64_t p1 = ((int64_t)(run + 1) * npairs) / a.nruns;
    const double mid = 0.5 * (a.nwin - 1);
    const double s2 = a.scale * a.scale;
    const int n = a.n, NF = a.nfreq;

    fft8::Twid<N> tw;
    fft8::twid_load<N>(t, a.tab, tw);
    constexpr bool RES = N <= 4096;       // (see spec_blue_kernel)
    double win[8], cr[RES ? 8 : 1], ci[RES ? 8 : 1], br[RES ? 8 : 1], bi[RES ? 8 : 1];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int i = NT * r + t;
        const double wv = a.window[i < a.nwin ? i : 0];
        win[r] = i < a.nwin ? wv : 0.0;
        if (RES) {
            cr[RES ? r : 0] = a.chirp[2 * i];
            ci[RES ? r : 0] = a.chirp[2 * i + 1];
            br[RES ? r : 0] = a.bperm[2 * i];
            bi[RES ? r : 0] = a.bperm[2 * i + 1];
        }
    }
    double acc[4] = {0.0, 0.0, 0.0, 0.0};

    for (int64_t p = p0; p < p1; ++p) {
        const int64_t sa = 2 * p, sb = 2 * p + 1;
        const bool has_b = sb < a.nseg;
        const double *pa = xr + sa * (int64_t)a.stride + t;
        const int64_t db = has_b ? a.stride : 0;
        double re[8], im[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int i = NT * r + t;
            const bool in = i < a.nwin;
            const int ii = in ? NT * r : -t;              // stays inside the segment
            const double xa = pa[ii], xb = pa[db + ii];
            re[r] = in ? xa : 0.0;
            im[r] = (in && has_b) ? xb : 0.0;
        }
        double sum_a = 0.0, sum_b = 0.0, lin_a = 0.0, lin_b = 0.0;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int i = NT * r + t;
            sum_a += re[r];
            sum_b += im[r];
            if (LINEAR) {
                lin_a += (i - mid) * re[r];
                lin_b += (i - mid) * im[r];
            }
        }
        sum_a = wave_sum63(sum_a);
        sum_b = wave_sum63(sum_b);
        if (LINEAR) {
            lin_a = wave_sum63(lin_a);
            lin_b = wave_sum63(lin_b);
        }
        if ((t & 63) == 63) {
            red[t >> 6][0] = sum_a;
            red[t >> 6][1] = sum_b;
            red[t >> 6][2] = lin_a;
            red[t >> 6][3] = lin_b;
        }
        __syncthreads();   // also: every LDS read of the previous pair is done
        double tot[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < NWV; ++q) {
            tot[0] += red[q][0];
            tot[1] += red[q][1];
            if (LINEAR) {
                tot[2] += red[q][2];
                tot[3] += red[q][3];
            }
        }
        const double mean_a = tot[0] / a.nwin, mean_b = tot[1] / a.nwin;
        double slope_a = 0.0, slope_b = 0.0;
        if (LINEAR) {
            const double nn = (double)a.nwin;
            const double sxx = nn * (nn * nn - 1.0) / 12.0;
            slope_a = sxx > 0.0 ? tot[2] / sxx : 0.0;
            slope_b = sxx > 0.0 ? tot[3] / sxx : 0.0;
        }
        // (a segment that is not finite goes in as zeros and comes out as NaN: spec_cube_kernel)
        const bool bad_a = MODE != OSZ_SPEC_PSD_MEAN && !spec_finite(mean_a + slope_a);
        const bool bad_b = MODE != OSZ_SPEC_PSD_MEAN && has_b && !spec_finite(mean_b + slope_b);
        const bool drop_a = bad_a && !bad_b, drop_b = bad_b && !bad_a;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int i = NT * r + t;
            double va = (LINEAR ? re[r] - mean_a - slope_a * (i - mid) : re[r] - mean_a) * win[r];
            double vb = has_b ? (LINEAR ? im[r] - mean_b - slope_b * (i - mid) : im[r] - mean_b) * win[r] : 0.0;
            if (MODE != OSZ_SPEC_PSD_MEAN) {
                if (drop_a) va = 0.0;
                if (drop_b) vb = 0.0;
            }
            fft8::cmul(va, vb, RES ? cr[RES ? r : 0] : a.chirp[2 * i], RES ? ci[RES ? r : 0] : a.chirp[2 * i + 1]);
            re[r] = va;
            im[r] = vb;
        }
        int tt = t;
        asm volatile("" : "+v"(tt));
        Fwd8<N, 0>::run(tt, re, im, tw, lds8);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int i = NT * r + tt;
            fft8::cmul(re[r], im[r], RES ? br[RES ? r : 0] : a.bperm[2 * i], RES ? bi[RES ? r : 0] : a.bperm[2 * i + 1]);
        }
        __syncthreads();   // the forward transform's last LDS reads are done
        asm volatile("" : "+v"(tt));
        Inv8<N, fft8::Plan<N>::NS - 1>::run(tt, re, im, tw, lds8);
        if (MODE == OSZ_SPEC_PSD_MEAN) {
            // |X[k]|^2 = |conv[k]|^2 (the chirp on the way out has modulus one)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = fma(re[r], re[r], fma(im[r], im[r], acc[r]));
            continue;
        }
        // ---- segment modes: X[k] = w[k] conv[k], k < n, side by side in LDS
        __syncthreads();   // the inverse transform's last LDS reads are done
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = NT * r + tt;
            if (k < n) {
                double zr = re[r], zi = im[r];
                fft8::cmul(zr, zi, RES ? cr[RES ? r : 0] : a.chirp[2 * k], RES ? ci[RES ? r : 0] : a.chirp[2 * k + 1]);
                lds8[k] = C2{zr, zi};
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int k = NT * r + tt;
            if (k >= NF) continue;
            const C2 z = lds8[k], q = lds8[k == 0 ? 0 : n - k];
            double ar = 0.5 * (z.re + q.re), ai = 0.5 * (z.im - q.im);
            double br_ = 0.5 * (z.im + q.im), bi_ = -0.5 * (z.re - q.re);
            if (bad_a) ar = ai = spec_nan();
            if (bad_b) br_ = bi_ = spec_nan();
            const bool dbl = k != 0 && !(2 * k == n);
            if (MODE == OSZ_SPEC_DFT_SEGMENTS) {
                double *o = (double *)a.out;
                const int64_t ia = ((sa * a.nch + c) * (int64_t)NF + k) * 2;
                o[ia] = ar * a.scale;
                o[ia + 1] = ai * a.scale;
                if (has_b) {
                    const int64_t ib = ((sb * a.nch + c) * (int64_t)NF + k) * 2;
                    o[ib] = br_ * a.scale;
                    o[ib + 1] = bi_ * a.scale;
                }
            } else {
                const double f = dbl ? 2.0 * s2 : s2;
                double *o = (double *)a.out;
                o[(sa * a.nch + c) * (int64_t)NF + k] = (ar * ar + ai * ai) * f;
                if (has_b) o[(sb * a.nch + c) * (int64_t)NF + k] = (br_ * br_ + bi_ * bi_) * f;
            }
        }
    }
    if (MODE != OSZ_SPEC_PSD_MEAN) return;
    // fold bin k with its mirror n - k; one-sided doubling and the scale here
    double *D = reinterpret_cast<double *>(lds8);
    __syncthreads();       // the last pair's LDS reads are done
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = NT * r + t;
        if (k < n) D[k] = acc[r];
    }
    __syncthreads();
    double *o = a.partial + ((int64_t)c * a.nruns + run) * NF;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int k = NT * r + t;
        if (k < NF) {
            const bool dbl = k != 0 && !(2 * k == n);
            o[k] = 0.5 * (D[k] + D[k == 0 ? 0 : n - k]) * (dbl ? 2.0 * s2 : s2);
        }
    }
}

// dsum[c][k] += sum over runs of partial[c][run][k] for any nfreq
__global__ void spec_partial_reduce_n_kernel(const double *partial, double *dsum, int nruns, int nf) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (k >= nf) return;
    double s = 0.0;
    for (int r = 0; r < nruns; ++r) s += partial[((int64_t)c * nruns + r) * nf + k];
    dsum[(int64_t)c * nf + k] += s;
}

// W_8192^j, j < 1024: one table per device for every transform size of fft8.h
int get_fft8_table(const double **out) {
    static std::mutex mu;
    static std::map<int, double *> per_device;
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    OSZ_HIP(hipGetDevice(&dev));
    double *&tab = per_device[dev];
    if (!tab) {
        const long double PI = acosl(-1.0L);
        std::vector<double> h(2 * fft8::kTabLen);
        for (int j = 0; j < fft8::kTabLen; ++j) {
            const long double ang = -2.0L * PI * (long double)j / (long double)fft8::kTabN;
            h[2 * j] = (double)cosl(ang);
            h[2 * j + 1] = (double)sinl(ang);
        }
        OSZ_HIP(hipMalloc(&tab, h.size() * sizeof(double)));
        OSZ_HIP(hipMemcpy(tab, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    *out = tab;
    return OSZ_OK;
}

}  // namespace osz

using namespace osz;

struct osz_spec_s {
    int device;         // HIP device the handle's buffers live on
    int nwin, nfft, stride, nfreq, detrend, mode, nch;
    double scale;
    double *dwindow;
    double *dcarry[2];
    int cur;
    int64_t ncarry, ncap;
    double *dsum;       // (nch, nfreq) PSD_MEAN accumulator
    int64_t count;      // segments accumulated
    int64_t *dcount;    // device scratch for the count of osz_welch_reduce
    // batched FFT staging, grow-only
    double *drows, *dspec;
    int64_t rows_cap;   // rows the staging buffers hold
    void *dwork;
    size_t work_cap;
    std::map<int64_t, rocfft_plan> *plans;  // keyed by batch (rows)
    bool fused;          // nwin == nfft == 4096: on-chip path (spec_cube_kernel)
    bool fused8;         // nfft = 512 ... 8192, a power of two: on-chip path (spec8_kernel)
    const double *tab8;  // twiddle table of fft8.h
    bool mixed;          // even nfft = 2 * (product of 2, 3, 5) <= 20412: on-chip path (specmix_kernel)
    int mix_npass, mix_radix[mix::kMaxPass];
    bool split;          // even nfft whose half is beyond the LDS, nfft / 2 = R0 S0: specsplit_kernel
    int split_r0, split_s0;
    int mix_blkfast[mix::kMaxPass];   // lane map per pass (specmix_lane_maps)
    int mix_blkfast1[mix::kMaxPass];  // specsplit.h: of a self-paired workgroup's S0 points
    bool blue;           // any other nfft <= 4096: Bluestein on the fft8 transforms (spec_blue_kernel)
    int blue_m;          // its convolution length, a power of two >= 2 nfft - 1 (>= 512)
    double *dchirp;      // [blue_m][2]: w[j] = exp(-i pi j^2 / nfft), zeros behind nfft
    double *dbperm;      // [blue_m][2]: the transformed chirp in the forward transform's bin order, / blue_m
    double *dtwn;        // specmix: W_nfft^j, j < nfft
    int *dpos;           // specmix: slot of Z[k] after the in-place passes
    double *dhead;       // fft8 path: carry ++ head of the chunk, (nch, ncap + nwin)
    double *dpartial;    // fused PSD_MEAN: (nch, nruns_cap, 2049)
    int64_t partial_cap;
    fft::Tables tb;
};

static const int64_t kSpecMaxElems = (int64_t)1 << 27;  // staging doubles per batch (1 GiB)

static int spec_plan(osz_spec_s *h, int64_t rows, rocfft_plan *out) {
    auto it = h->plans->find(rows);
    if (it != h->plans->end()) {
        *out = it->second;
        return OSZ_OK;
    }
    static bool setup_done = false;
    if (!setup_done) {
        OSZ_FFT(rocfft_setup());
        setup_done = true;
    }
    rocfft_plan plan = nullptr;
    size_t len[1] = {(size_t)h->nfft};
    OSZ_FFT(rocfft_plan_create(&plan, rocfft_placement_notinplace,
                               rocfft_transform_type_real_forward, rocfft_precision_double, 1, len,
                               (size_t)rows, nullptr));
    (*h->plans)[rows] = plan;
    *out = plan;
    return OSZ_OK;
}


template <int N>
static int spec8_launch(osz_spec_s *h, const Spec8Args &a, hipStream_t st) {
    using kern_t = void (*)(Spec8Args);
    static const kern_t ks[3][2][2] = {
        {{spec8_kernel<N, 0, false, false>, spec8_kernel<N, 0, false, true>},
         {spec8_kernel<N, 0, true, false>, spec8_kernel<N, 0, true, true>}},
        {{spec8_kernel<N, 1, false, false>, spec8_kernel<N, 1, false, true>},
         {spec8_kernel<N, 1, true, false>, spec8_kernel<N, 1, true, true>}},
        {{spec8_kernel<N, 2, false, false>, spec8_kernel<N, 2, false, true>},
         {spec8_kernel<N, 2, true, false>, spec8_kernel<N, 2, true, true>}}};
    const bool half = h->nwin == N && 2 * h->stride == N;
    const kern_t k = ks[h->mode][h->detrend == OSZ_DETREND_LINEAR ? 1 : 0][half ? 1 : 0];
    const size_t lds = sizeof(fft8::C2) * N;
    OSZ_DYN_LDS(k, lds);
    KernelTimer kt("spec_fused", st);
    hipLaunchKernelGGL(k, dim3((unsigned)a.nruns, h->nch), dim3(N / 8), lds, st, a);
    return OSZ_OK;
}

// head[c][0 : ncarry + m] = carry[c][0 : ncarry] ++ x[c][0 : m]
__global__ void spec_head_kernel(const double *carry, int64_t ncap, int64_t ncarry, const double *x,
                                 int64_t ldx, int64_t m, double *head, int64_t ldh) {
    const int c = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncarry + m) return;
    head[(int64_t)c * ldh + i] = i < ncarry ? carry[(int64_t)c * ncap + i] : x[(int64_t)c * ldx + (i - ncarry)];
}

// radix plan of the M = nfft / 2 point transform of specmix.h: 10s, 4s, a 2, 3s, 5s, 7s
static bool specmix_plan(int nfft, int *npass, int *radix) {
    if (nfft < 4 || (nfft & 1)) return false;
    int m = nfft / 2, n = 0;
    if (m > mix::kMaxM) return false;
    auto take = [&](int r) {
        while (m % r == 0 && n < mix::kMaxPass) {
            radix[n++] = r;
            m /= r;
        }
    };
    take(10);
    take(4);
    take(2);
    take(3);
    take(5);
    take(7);
    *npass = n;
    return m == 1 && n >= 1;
}

// threads of the specmix workgroup: a radix-10 pass has M / 10 butterflies -- one
// per thread; every pair of bins (k, M - k) then has its (thread, trip) as well
static int specmix_threads(int M) {
    return M <= 640 ? 64 : M <= 1280 ? 128 : M <= 2560 ? 256 : M <= 5120 ? 512 : 1024;
}

// LDS-array cycles of one wave-wide ds_read_b128 of 16-byte slots (MI355X_MICROARCH.md,
// LDS: four fixed groups of 16 lanes, bank = (byte address / 4) mod 64, distinct
// addresses on one bank serialise): the slot modulo 16 names the four banks a lane
// touches.
static int specmix_read_cycles(const int *slot, const bool *active) {
    static const int group[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                     {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                     {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                     {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
    int cycles = 0;
    for (int g = 0; g < 4; ++g) {
        int count[16] = {0}, seen[16], nseen = 0, worst = 0;
        for (int q = 0; q < 16; ++q) {
            const int l = group[g][q];
            if (!active[l]) continue;
            bool dup = false;
            for (int u = 0; u < nseen; ++u) dup = dup || seen[u] == slot[l];
            if (dup) continue;                       // same address: broadcast
            seen[nseen++] = slot[l];
            const int c = ++count[slot[l] & 15];
            worst = c > worst ? c : worst;
        }
        cycles += worst;
    }
    return cycles;
}

// For every pass: do consecutive lanes walk one block (inner index fastest) or
// the blocks (block index fastest)?  Whichever reads with fewer bank conflicts.
// (M points in LDS, the passes from p0 on, the first of them over blocks of B0 points)
static void specmix_lane_maps(osz_spec_s *h, int M, int B0, int p0, int *blkfast) {
    const int NT = specmix_threads(M);
    int B = B0;
    for (int p = p0; p < h->mix_npass; ++p) {
        const int r = h->mix_radix[p], S = B / r, nb = M / r, nblk = M / B;
        long cost[2] = {0, 0};
        for (int mode = 0; mode < 2; ++mode)
            for (int b0 = 0; b0 < nb; b0 += 64) {
                // a wave covers 64 consecutive butterflies of one trip (NT is a multiple of 64)
                int base[64];
                bool act[64];
                for (int l = 0; l < 64; ++l) {
                    const int b = b0 + l;
                    act[l] = b < nb;
                    const int bb = act[l] ? b : 0;
                    const int blk = mode ? bb % nblk : bb / S, inner = mode ? bb / nblk : bb % S;
                    base[l] = blk * B + inner;
                }
                for (int q = 0; q < r; ++q) {
                    int slot[64];
                    for (int l = 0; l < 64; ++l) slot[l] = base[l] + q * S;
                    cost[mode] += specmix_read_cycles(slot, act);
                }
            }
        (void)NT;
        blkfast[p] = cost[1] < cost[0] ? 1 : 0;
        B = S;
    }
}

static int specmix_tables(osz_spec_s *h) {
    const int N = h->nfft;
    // the whole half in LDS, or (specsplit.h) two of its R0 sub-transforms: their passes, their output slots
    const int M = h->split ? h->split_s0 : N / 2, p0 = h->split ? 1 : 0;
    specmix_lane_maps(h, h->split ? 2 * M : M, M, p0, h->mix_blkfast);
    if (h->split) specmix_lane_maps(h, M, M, p0, h->mix_blkfast1);
    const long double PI = acosl(-1.0L);
    std::vector<double> tw(2 * (size_t)N);
    for (int j = 0; j < N; ++j) {
        const long double ang = -2.0L * PI * (long double)j / (long double)N;
        tw[2 * j] = (double)cosl(ang);
        tw[2 * j + 1] = (double)sinl(ang);
    }
    std::vector<int> pos(M);
    for (int k = 0; k < M; ++k) {
        int kk = k, S = M, p = 0;
        for (int q = p0; q < h->mix_npass; ++q) {
            const int r = h->mix_radix[q];
            S /= r;
            p += (kk % r) * S;
            kk /= r;
        }
        pos[k] = p;
    }
    OSZ_HIP(hipMalloc(&h->dtwn, tw.size() * sizeof(double)));
    OSZ_HIP(hipMalloc(&h->dpos, pos.size() * sizeof(int)));
    OSZ_HIP(hipMemcpy(h->dtwn, tw.data(), tw.size() * sizeof(double), hipMemcpyHostToDevice));
    OSZ_HIP(hipMemcpy(h->dpos, pos.data(), pos.size() * sizeof(int), hipMemcpyHostToDevice));
    return OSZ_OK;
}

template <int NT>
static int specmix_launch_nt(osz_spec_s *h, const mix::Args &a, hipStream_t st) {
    using kern_t = void (*)(mix::Args);
    static const kern_t ks[3][2] = {
        {mix::specmix_kernel<0, false, NT>, mix::specmix_kernel<0, true, NT>},
        {mix::specmix_kernel<1, false, NT>, mix::specmix_kernel<1, true, NT>},
        {mix::specmix_kernel<2, false, NT>, mix::specmix_kernel<2, true, NT>}};
    const kern_t k = ks[h->mode][h->detrend == OSZ_DETREND_LINEAR ? 1 : 0];
    const size_t lds = sizeof(mix::C2) * (size_t)a.M;
    OSZ_DYN_LDS(k, lds);
    KernelTimer kt("spec_fused", st);
    hipLaunchKernelGGL(k, dim3((unsigned)a.nruns, h->nch), dim3(NT), lds, st, a);
    return OSZ_OK;
}

// one launch of specmix_kernel over nseg segments of a contiguous source
static int specmix_run(osz_spec_s *h, const double *src, int64_t ld, void *out, int64_t nseg,
                       hipStream_t st) {
    int64_t R = (nseg * h->nch) / 2048;   // segments per run
    if (R > 64) R = 64;
    if (R < 1) R = 1;
    const int64_t nruns = (nseg + R - 1) / R;
    mix::Args a{};
    a.x = src;
    a.window = h->dwindow;
    a.out = out;
    a.tw = h->dtwn;
    a.pos = h->dpos;
    a.ldx = ld;
    a.nseg = nseg;
    a.stride = h->stride;
    a.nwin = h->nwin;
    a.nch = h->nch;
    a.nruns = (int)nruns;
    a.N = h->nfft;
    a.M = h->nfft / 2;
    a.npass = h->mix_npass;
    {
        static const bool half_on = [] {           // (OSZ_MIX_HALF=0: every segment sums both its halves, A/B runs)
            const char *e = getenv("OSZ_MIX_HALF");
            return !(e && e[0] == '0');
        }();
        a.halfcarry = half_on && 2 * h->stride == h->nwin && h->nwin == h->nfft && h->mix_radix[0] % 2 == 0;
    }
    for (int q = 0, B = a.M; q < h->mix_npass; ++q) {
        a.radix[q] = h->mix_radix[q];
        const int S = B / a.radix[q], nblk = a.M / B;
        a.blkfast[q] = h->mix_blkfast[q];
        a.div[q] = a.blkfast[q] ? nblk : S;
        a.inv[q] = a.div[q] > 1 ? (unsigned)(((1ull << 32) + a.div[q] - 1) / a.div[q]) : 0u;
        B = S;
    }
    a.scale = h->scale;
    if (h->mode == OSZ_SPEC_PSD_MEAN) {
        const int64_t need = (int64_t)h->nch * nruns * h->nfreq;
        if (need > h->partial_cap) {
            OSZ_HIP(hipStreamSynchronize(st));
            (void)hipFree(h->dpartial);
            h->dpartial = nullptr;
            if (hipMalloc(&h->dpartial, sizeof(double) * need) != hipSuccess)
                return fail(OSZ_ERR_NOMEM, "osz_spec_push: partial sums (%lld doubles)", (long long)need);
            h->partial_cap = need;
        }
        a.partial = h->dpartial;
    }
    int rc;
    switch (specmix_threads(a.M)) {
        case 64: rc = specmix_launch_nt<64>(h, a, st); break;
        case 128: rc = specmix_launch_nt<128>(h, a, st); break;
        case 256: rc = specmix_launch_nt<256>(h, a, st); break;
        case 512: rc = specmix_launch_nt<512>(h, a, st); break;
        default: rc = specmix_launch_nt<1024>(h, a, st); break;
    }
    if (rc) return rc;
    OSZ_HIP(hipGetLastError());
    if (h->mode == OSZ_SPEC_PSD_MEAN) {
        hipLaunchKernelGGL(spec_partial_reduce_n_kernel, dim3((h->nfreq + 255) / 256, h->nch), dim3(256),
                           0, st, h->dpartial, h->dsum, (int)nruns, h->nfreq);
        OSZ_HIP(hipGetLastError());
    }
    return OSZ_OK;
}


// ---- lengths whose half is beyond the LDS (specsplit.h) ------------------------------------
// nfft / 2 = R0 S0 with the smallest R0 whose PAIR of S0-point transforms fits the LDS, S0 a
// product of 2, 3, 5, 7 (the fewer workgroups read a segment the better; R0 itself is any
// divisor up to 32: its pass is evaluated directly).  radix[0] = R0, radix[1 ..] = the plan of S0.
static bool specsplit_plan(int nfft, int *r0, int *npass, int *radix) {
    if (nfft < 4 || (nfft & 1)) return false;
    const int M = nfft / 2;
    static const int from = [] {                // (OSZ_SPLIT_FROM: halves above this many points are split, timing runs)
        const char *e = getenv("OSZ_SPLIT_FROM");
        return e ? atoi(e) : mix::kMaxM;
    }();
    if (M <= from) return false;
    static const int local_max = [] {           // (OSZ_SPLIT_LOCAL: fewer local points, more workgroups per segment)
        const char *e = getenv("OSZ_SPLIT_LOCAL");
        const int v = e ? atoi(e) : 0;
        return v >= 1282 && v < mix::kMaxSplitLocal ? v : mix::kMaxSplitLocal;
    }();
    for (int R = 2; R <= mix::kMaxSplitR0; ++R) {
        if (M % R) continue;
        const int S0 = M / R;
        if (2 * S0 > local_max || 2 * S0 <= 1280) continue;      // (three workgroup sizes are built)
        int np = 0, rad[mix::kMaxPass];
        if (!specmix_plan(2 * S0, &np, rad) || np + 1 > mix::kMaxPass) continue;
        radix[0] = R;
        for (int q = 0; q < np; ++q) radix[q + 1] = rad[q];
        *npass = np + 1;
        *r0 = R;
        return true;
    }
    return false;
}

template <int NT>
static int specsplit_launch_nt(osz_spec_s *h, const mix::SplitArgs &g, int64_t nblocks, hipStream_t st) {
    using kern_t = void (*)(mix::SplitArgs);
    static const kern_t ks[3][2] = {
        {mix::specsplit_kernel<0, false, NT>, mix::specsplit_kernel<0, true, NT>},
        {mix::specsplit_kernel<1, false, NT>, mix::specsplit_kernel<1, true, NT>},
        {mix::specsplit_kernel<2, false, NT>, mix::specsplit_kernel<2, true, NT>}};
    const kern_t k = ks[h->mode][h->detrend == OSZ_DETREND_LINEAR ? 1 : 0];
    const size_t lds = sizeof(mix::C2) * 2 * (size_t)g.S0;
    OSZ_DYN_LDS(k, lds);
    KernelTimer kt("spec_fused", st);
    hipLaunchKernelGGL(k, dim3((unsigned)nblocks), dim3(NT), lds, st, g);
    return OSZ_OK;
}

// one launch of specsplit_kernel over nseg segments of a contiguous source
static int specsplit_run(osz_spec_s *h, const double *src, int64_t ld, void *out, int64_t nseg,
                         hipStream_t st) {
    const int R0 = h->split_r0, S0 = h->split_s0, NW = (R0 + 1) / 2, Ml = 2 * S0;     // the pairs and the self-paired
    // segments per run: long runs (a run's first segment sums both its halves) while a few
    // rounds of workgroups remain
    int64_t R = (nseg * h->nch * NW) / 2048;
    if (R > 64) R = 64;
    if (R < 1) R = 1;
    const int64_t nruns = (nseg + R - 1) / R;
    mix::SplitArgs g{};
    mix::Args &a = g.a;
    a.x = src;
    a.window = h->dwindow;
    a.out = out;
    a.tw = h->dtwn;
    a.pos = h->dpos;
    a.ldx = ld;
    a.nseg = nseg;
    a.stride = h->stride;
    a.nwin = h->nwin;
    a.nch = h->nch;
    a.nruns = (int)nruns;
    a.N = h->nfft;
    a.M = h->nfft / 2;
    a.npass = h->mix_npass;
    {
        static const bool half_on = [] {
            const char *e = getenv("OSZ_MIX_HALF");
            return !(e && e[0] == '0');
        }();
        a.halfcarry = half_on && 2 * h->stride == h->nwin && h->nwin == h->nfft && a.M % 2 == 0;
    }
    a.radix[0] = R0;
    for (int q = 1, B = S0; q < h->mix_npass; ++q) {
        a.radix[q] = h->mix_radix[q];
        const int S = B / a.radix[q], nblk = Ml / B;
        a.blkfast[q] = h->mix_blkfast[q];
        a.div[q] = a.blkfast[q] ? nblk : S;
        a.inv[q] = a.div[q] > 1 ? (unsigned)(((1ull << 32) + a.div[q] - 1) / a.div[q]) : 0u;
        g.blkfast1[q] = h->mix_blkfast1[q];
        g.div1[q] = g.blkfast1[q] ? S0 / B : S;
        g.inv1[q] = g.div1[q] > 1 ? (unsigned)(((1ull << 32) + g.div1[q] - 1) / g.div1[q]) : 0u;
        B = S;
    }
    a.scale = h->scale;
    g.R0 = R0;
    g.S0 = S0;
    g.NW = NW;
    g.nunits = (int)(nruns * h->nch);
    if (h->mode == OSZ_SPEC_PSD_MEAN) {
        const int64_t need = (int64_t)h->nch * nruns * h->nfreq;
        if (need > h->partial_cap) {
            OSZ_HIP(hipStreamSynchronize(st));
            (void)hipFree(h->dpartial);
            h->dpartial = nullptr;
            if (hipMalloc(&h->dpartial, sizeof(double) * need) != hipSuccess)
                return fail(OSZ_ERR_NOMEM, "osz_spec_push: partial sums (%lld doubles)", (long long)need);
            h->partial_cap = need;
        }
        a.partial = h->dpartial;
    }
    // workgroup ids: eight XCDs x (groups of units) x NW, specsplit.h
    const int64_t nblocks = ((int64_t)g.nunits + 7) / 8 * NW * 8;
    int rc;
    switch (specmix_threads(Ml)) {
        case 256: rc = specsplit_launch_nt<256>(h, g, nblocks, st); break;
        case 512: rc = specsplit_launch_nt<512>(h, g, nblocks, st); break;
        case 1024: rc = specsplit_launch_nt<1024>(h, g, nblocks, st); break;
        default: return fail(OSZ_ERR_STATE, "specsplit_run: %d local points", Ml);
    }
    if (rc) return rc;
    OSZ_HIP(hipGetLastError());
    if (h->mode == OSZ_SPEC_PSD_MEAN) {
        hipLaunchKernelGGL(spec_partial_reduce_n_kernel, dim3((h->nfreq + 255) / 256, h->nch), dim3(256),
                           0, st, h->dpartial, h->dsum, (int)nruns, h->nfreq);
        OSZ_HIP(hipGetLastError());
    }
    return OSZ_OK;
}


// host tables of the chirp transform (long double; an iterative radix-2 transform for the
// m-point spectrum of the wrapped conjugate chirp)
template <int N>
static void blue_perm(const std::vector<long double> &Br, const std::vector<long double> &Bi, std::vector<double> &out) {
    constexpr int NT = N / 8, L = fft8::ilog2(N);
    out.assign((size_t)2 * N, 0.0);
    for (int r = 0; r < 8; ++r)
        for (int t = 0; t < NT; ++t) {
            const int k = fft8::revdigits<L>(fft8::idx_of<0>(t, r));
            out[2 * ((size_t)NT * r + t)] = (double)(Br[k] / N);
            out[2 * ((size_t)NT * r + t) + 1] = (double)(Bi[k] / N);
        }
}

static int blue_tables(osz_spec_s *h) {
    const int n = h->nfft, m = h->blue_m;
    const long double PI = acosl(-1.0L);
    std::vector<double> chirp((size_t)2 * m, 0.0);
    std::vector<long double> br(m, 0.0L), bi(m, 0.0L);
    for (int j = 0; j < n; ++j) {
        const long long q = ((long long)j * j) % (2LL * n);          // j^2 mod 2 n: the angle stays small
        const long double ang = PI * (long double)q / (long double)n;
        const long double cw = cosl(ang), sw = sinl(ang);
        chirp[2 * (size_t)j] = (double)cw;                            // w = exp(-i ang)
        chirp[2 * (size_t)j + 1] = (double)(-sw);
        br[j] = cw;                                                   // conj w at +j and, wrapped, at -j
        bi[j] = sw;
        if (j > 0) {
            br[m - j] = cw;
            bi[m - j] = sw;
        }
    }
    // in-place decimation-in-time transform of (br, bi), forward sign
    for (int i = 1, j = 0; i < m; ++i) {
        int bit = m >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            std::swap(br[i], br[j]);
            std::swap(bi[i], bi[j]);
        }
    }
    for (int len = 2; len <= m; len <<= 1) {
        for (int k = 0; k < len / 2; ++k) {
            const long double ang = -2.0L * PI * (long double)k / (long double)len;
            const long double wr = cosl(ang), wi = sinl(ang);
            for (int i = k; i < m; i += len) {
                const int j = i + len / 2;
                const long double ur = br[j] * wr - bi[j] * wi, ui = br[j] * wi + bi[j] * wr;
                br[j] = br[i] - ur;
                bi[j] = bi[i] - ui;
                br[i] += ur;
                bi[i] += ui;
            }
        }
    }
    std::vector<double> bperm;
    switch (m) {
        case 512: blue_perm<512>(br, bi, bperm); break;
        case 1024: blue_perm<1024>(br, bi, bperm); break;
        case 2048: blue_perm<2048>(br, bi, bperm); break;
        case 4096: blue_perm<4096>(br, bi, bperm); break;
        default: blue_perm<8192>(br, bi, bperm); break;
    }
    OSZ_HIP(hipMalloc(&h->dchirp, chirp.size() * sizeof(double)));
    OSZ_HIP(hipMalloc(&h->dbperm, bperm.size() * sizeof(double)));
    OSZ_HIP(hipMemcpy(h->dchirp, chirp.data(), chirp.size() * sizeof(double), hipMemcpyHostToDevice));
    OSZ_HIP(hipMemcpy(h->dbperm, bperm.data(), bperm.size() * sizeof(double), hipMemcpyHostToDevice));
    return OSZ_OK;
}

template <int N>
static int blue_launch(osz_spec_s *h, const BlueArgs &a, hipStream_t st) {
    using kern_t = void (*)(BlueArgs);
    static const kern_t ks[3][2] = {{spec_blue_kernel<N, 0, false>, spec_blue_kernel<N, 0, true>},
                                    {spec_blue_kernel<N, 1, false>, spec_blue_kernel<N, 1, true>},
                                    {spec_blue_kernel<N, 2, false>, spec_blue_kernel<N, 2, true>}};
    const kern_t k = ks[h->mode][h->detrend == OSZ_DETREND_LINEAR ? 1 : 0];
    const size_t lds = sizeof(fft8::C2) * N;
    OSZ_DYN_LDS(k, lds);
    KernelTimer kt("spec_fused", st);
    hipLaunchKernelGGL(k, dim3((unsigned)a.nruns, h->nch), dim3(N / 8), lds, st, a);
    return OSZ_OK;
}

// one launch of spec_blue_kernel over nseg segments of a contiguous source
static int blue_run(osz_spec_s *h, const double *src, int64_t ld, void *out, int64_t nseg, hipStream_t st) {
    // pairs of segments per run: a few rounds of the chip, the set-up amortised
    const int64_t nitem = (nseg + 1) / 2;
    int64_t R = (nitem * h->nch) / 2048;
    if (R > 64) R = 64;
    if (R < 1) R = 1;
    const int64_t nruns = (nitem + R - 1) / R;
    BlueArgs a{};
    a.x = src;
    a.window = h->dwindow;
    a.out = out;
    a.tab = h->tab8;
    a.chirp = h->dchirp;
    a.bperm = h->dbperm;
    a.ldx = ld;
    a.nseg = nseg;
    a.stride = h->stride;
    a.nwin = h->nwin;
    a.nch = h->nch;
    a.nruns = (int)nruns;
    a.n = h->nfft;
    a.nfreq = h->nfreq;
    a.scale = h->scale;
    if (h->mode == OSZ_SPEC_PSD_MEAN) {
        const int64_t need = (int64_t)h->nch * nruns * h->nfreq;
        if (need > h->partial_cap) {
            OSZ_HIP(hipStreamSynchronize(st));
            (void)hipFree(h->dpartial);
            h->dpartial = nullptr;
            if (hipMalloc(&h->dpartial, sizeof(double) * need) != hipSuccess)
                return fail(OSZ_ERR_NOMEM, "osz_spec_push: partial sums (%lld doubles)", (long long)need);
            h->partial_cap = need;
        }
        a.partial = h->dpartial;
    }
    int rc;
    switch (h->blue_m) {
        case 512: rc = blue_launch<512>(h, a, st); break;
        case 1024: rc = blue_launch<1024>(h, a, st); break;
        case 2048: rc = blue_launch<2048>(h, a, st); break;
        case 4096: rc = blue_launch<4096>(h, a, st); break;
        default: rc = blue_launch<8192>(h, a, st); break;
    }
    if (rc) return rc;
    OSZ_HIP(hipGetLastError());
    if (h->mode == OSZ_SPEC_PSD_MEAN) {
        hipLaunchKernelGGL(spec_partial_reduce_n_kernel, dim3((h->nfreq + 255) / 256, h->nch), dim3(256),
                           0, st, h->dpartial, h->dsum, (int)nruns, h->nfreq);
        OSZ_HIP(hipGetLastError());
    }
    return OSZ_OK;
}

// one launch of spec8_kernel over nseg segments of a contiguous source
static int spec8_run(osz_spec_s *h, const double *src, int64_t ld, void *out, int64_t nseg,
                     hipStream_t st) {
    if (h->mixed) return specmix_run(h, src, ld, out, nseg, st);
    if (h->split) return specsplit_run(h, src, ld, out, nseg, st);
    if (h->blue) return blue_run(h, src, ld, out, nseg, st);
    const int64_t npairs = (nseg + 1) / 2;
    // runs: enough workgroups for a few rounds of the chip, long enough that the
    // per-run set-up (twiddles, window, partial sums) stays small
    int64_t R = (npairs * h->nch) / 2048;
    if (R > 64) R = 64;
    if (R < 1) R = 1;
    const int64_t nruns = (npairs + R - 1) / R;
    Spec8Args a{};
    a.x = src;
    a.window = h->dwindow;
    a.out = out;
    a.tab = h->tab8;
    a.ldx = ld;
    a.nseg = nseg;
    a.stride = h->stride;
    a.nwin = h->nwin;
    a.nch = h->nch;
    a.nruns = (int)nruns;
    a.scale = h->scale;
    if (h->mode == OSZ_SPEC_PSD_MEAN) {
        const int64_t need = (int64_t)h->nch * nruns * h->nfreq;
        if (need > h->partial_cap) {
            OSZ_HIP(hipStreamSynchronize(st));
            (void)hipFree(h->dpartial);
            h->dpartial = nullptr;
            if (hipMalloc(&h->dpartial, sizeof(double) * need) != hipSuccess)
                return fail(OSZ_ERR_NOMEM, "osz_spec_push: partial sums (%lld doubles)", (long long)need);
            h->partial_cap = need;
        }
        a.partial = h->dpartial;
    }
    int rc;
    switch (h->nfft) {
        case 512: rc = spec8_launch<512>(h, a, st); break;
        case 1024: rc = spec8_launch<1024>(h, a, st); break;
        case 2048: rc = spec8_launch<2048>(h, a, st); break;
        case 4096: rc = spec8_launch<4096>(h, a, st); break;
        case 8192: rc = spec8_launch<8192>(h, a, st); break;
        default: return fail(OSZ_ERR_INVALID, "spec8_push: nfft=%d", h->nfft);
    }
    if (rc) return rc;
    OSZ_HIP(hipGetLastError());
    if (h->mode == OSZ_SPEC_PSD_MEAN) {
        hipLaunchKernelGGL(spec_partial_reduce_n_kernel, dim3((h->nfreq + 255) / 256, h->nch), dim3(256),
                           0, st, h->dpartial, h->dsum, (int)nruns, h->nfreq);
        OSZ_HIP(hipGetLastError());
    }
    return OSZ_OK;
}

// On-chip paths for power-of-two nfft in [512, 8192] (spec8_kernel) and for the
// other even nfft made of factors 2, 3, 5 (specmix_kernel).  The kernels
// read ONE contiguous source.  The segments that begin in the carry of the
// previous push (fewer than nwin / stride + 1 of them) are served from a small
// head buffer = carry ++ the first nwin samples of the chunk; every other
// segment lies inside the chunk.
static int spec8_push(osz_spec_s *h, const double *x, int64_t ldx, int64_t n, void *out,
                      int64_t nseg, hipStream_t st) {
    int64_t nhead = h->ncarry > 0 ? (h->ncarry + h->stride - 1) / h->stride : 0;
    if (nhead > nseg) nhead = nseg;
    const size_t seg_out = (size_t)h->nch * h->nfreq * (h->mode == OSZ_SPEC_DFT_SEGMENTS ? 2 : 1);
    if (nhead > 0) {
        const int64_t ldh = h->ncap + h->nwin;
        if (!h->dhead) {
            if (hipMalloc(&h->dhead, sizeof(double) * (size_t)h->nch * ldh) != hipSuccess)
                return fail(OSZ_ERR_NOMEM, "osz_spec_push: head buffer");
        }
        int64_t m = (nhead - 1) * h->stride + h->nwin - h->ncarry;   // chunk samples the head segments need
        if (m > n) m = n;
        if (m < 0) m = 0;
        hipLaunchKernelGGL(spec_head_kernel, dim3((unsigned)((h->ncarry + m + 255) / 256), h->nch), dim3(256),
                           0, st, h->dcarry[h->cur], h->ncap, h->ncarry, x, ldx, m, h->dhead, ldh);
        OSZ_HIP(hipGetLastError());
        int rc = spec8_run(h, h->dhead, ldh, out, nhead, st);
        if (rc) return rc;
    }
    if (nseg > nhead) {
        // segment nhead starts at virtual sample nhead * stride >= ncarry
        const int64_t off = nhead * (int64_t)h->stride - h->ncarry;
        void *o = out ? (void *)((double *)out + (size_t)nhead * seg_out) : nullptr;
        int rc = spec8_run(h, x + off, ldx, o, nseg - nhead, st);
        if (rc) return rc;
    }
    return OSZ_OK;
}

extern "C" {

int osz_spec_create(osz_spec_t *h, int nwin, int nfft, int stride, const double *window,
                    double scale, int detrend, int mode, int nch) {
    OSZ_REQUIRE(h && window, "osz_spec_create: null argument");
    OSZ_REQUIRE(nwin >= 1 && nfft >= nwin && stride >= 1 && stride <= nwin && nch >= 1 &&
                    nch <= 65535,
                "osz_spec_create: nwin=%d nfft=%d stride=%d nch=%d", nwin, nfft, stride, nch);
    OSZ_REQUIRE(detrend == OSZ_DETREND_CONSTANT || detrend == OSZ_DETREND_LINEAR,
                "osz_spec_create: unknown detrend %d", detrend);
    OSZ_REQUIRE(mode >= OSZ_SPEC_PSD_MEAN && mode <= OSZ_SPEC_DFT_SEGMENTS,
                "osz_spec_create: unknown mode %d", mode);
    osz_spec_s *p = new osz_spec_s();
    p->device = 0;
    (void)hipGetDevice(&p->device);
    p->nwin = nwin;
    p->nfft = nfft;
    p->stride = stride;
    p->nfreq = nfft / 2 + 1;
    p->detrend = detrend;
    p->mode = mode;
    p->nch = nch;
    p->scale = scale;
    p->cur = 0;
    p->ncarry = 0;
    p->ncap = nwin;  // carry always holds < nwin samples
    p->count = 0;
    p->drows = p->dspec = nullptr;
    p->rows_cap = 0;
    p->dwork = nullptr;
    p->work_cap = 0;
    p->plans = new std::map<int64_t, rocfft_plan>();
    p->dpartial = nullptr;
    p->partial_cap = 0;
    {
        const char *e = getenv("OSZ_SPEC_FUSED");
        p->fused = (nwin == fft::N && nfft == fft::N) && !(e && atoi(e) == 0);
        // OSZ_SPEC_V8: 0 = never the fft8 path, 2 = also for nwin == nfft == 4096
        const char *e8 = getenv("OSZ_SPEC_V8");
        const int v8 = e8 ? atoi(e8) : 1;
        const bool pow2 = nfft >= 512 && nfft <= 8192 && (nfft & (nfft - 1)) == 0;
        p->fused8 = pow2 && v8 != 0 && (!p->fused || v8 == 2);
        if (p->fused8) p->fused = false;
        p->tab8 = nullptr;
        p->dhead = nullptr;
        if (p->fused) {
            int rc = get_fft_tables(p->tb);
            if (rc) { delete p->plans; delete p; return rc; }
        }
        if (p->fused8) {
            int rc = get_fft8_table(&p->tab8);
            if (rc) { delete p->plans; delete p; return rc; }
        }
        // OSZ_SPEC_MIX=0: never the mixed-radix on-chip path
        const char *em = getenv("OSZ_SPEC_MIX");
        p->dtwn = nullptr;
        p->dpos = nullptr;
        const char *ef = getenv("OSZ_SPLIT_FROM");
        p->mixed = !p->fused && !p->fused8 && !(em && atoi(em) == 0) && !(ef && nfft / 2 > atoi(ef)) &&
                   specmix_plan(nfft, &p->mix_npass, p->mix_radix);
        // its half beyond the LDS: two of its R0 sub-transforms per workgroup (OSZ_SPEC_SPLIT=0: the staging route)
        const char *es = getenv("OSZ_SPEC_SPLIT");
        p->split = !p->fused && !p->fused8 && !p->mixed && !(em && atoi(em) == 0) && !(es && atoi(es) == 0) &&
                   specsplit_plan(nfft, &p->split_r0, &p->mix_npass, p->mix_radix);
        p->split_s0 = p->split ? nfft / 2 / p->split_r0 : 0;
        if (p->mixed || p->split) {
            int rc = specmix_tables(p);
            if (rc) { (void)hipFree(p->dtwn); (void)hipFree(p->dpos); delete p->plans; delete p; return rc; }
        }
        // every other length up to 4096: the chirp transform (OSZ_SPEC_MIX=0 keeps it off too)
        p->dchirp = p->dbperm = nullptr;
        p->blue = !p->fused && !p->fused8 && !p->mixed && !(em && atoi(em) == 0) && nfft >= 2 && nfft <= 4096;
        p->blue_m = 0;
        if (p->blue) {
            int m = 512;
            while (m < 2 * nfft - 1) m <<= 1;
            p->blue_m = m;
            int rc = get_fft8_table(&p->tab8);
            if (!rc) rc = blue_tables(p);
            if (rc) { (void)hipFree(p->dchirp); (void)hipFree(p->dbperm); delete p->plans; delete p; return rc; }
        }
    }
    const size_t cb = sizeof(double) * (size_t)nch * p->ncap;
    const size_t ab = sizeof(double) * (size_t)nch * p->nfreq;
    OSZ_HIP(hipMalloc(&p->dwindow, sizeof(double) * nwin));
    OSZ_HIP(hipMalloc(&p->dcarry[0], cb));
    OSZ_HIP(hipMalloc(&p->dcarry[1], cb));
    OSZ_HIP(hipMalloc(&p->dsum, ab));
    OSZ_HIP(hipMalloc(&p->dcount, sizeof(int64_t)));
    OSZ_HIP(hipMemcpy(p->dwindow, window, sizeof(double) * nwin, hipMemcpyHostToDevice));
    OSZ_HIP(hipMemset(p->dsum, 0, ab));
    *h = p;
    return OSZ_OK;
}

int osz_spec_destroy(osz_spec_t h) {
    if (!h) return OSZ_OK;
    for (auto &kv : *h->plans) rocfft_plan_destroy(kv.second);
    delete h->plans;
    (void)hipFree(h->dwindow);
    (void)hipFree(h->dcarry[0]);
    (void)hipFree(h->dcarry[1]);
    (void)hipFree(h->dsum);
    (void)hipFree(h->dcount);
    (void)hipFree(h->drows);
    (void)hipFree(h->dspec);
    (void)hipFree(h->dwork);
    (void)hipFree(h->dpartial);
    (void)hipFree(h->dhead);
    (void)hipFree(h->dtwn);
    (void)hipFree(h->dpos);
    (void)hipFree(h->dchirp);
    (void)hipFree(h->dbperm);
    delete h;
    return OSZ_OK;
}

int osz_spec_reset(osz_spec_t h, void *stream) {
    OSZ_REQUIRE(h, "osz_spec_reset: null handle");
    h->ncarry = 0;
    h->count = 0;
    OSZ_HIP(hipMemsetAsync(h->dsum, 0, sizeof(double) * (size_t)h->nch * h->nfreq,
                           as_stream(stream)));
    return OSZ_OK;
}

int64_t osz_spec_seg_count(osz_spec_t h, int64_t n) {
    if (!h || n < 0) return -1;
    const int64_t total = h->ncarry + n;
    return total >= h->nwin ? (total - h->nwin) / h->stride + 1 : 0;
}

int osz_spec_push(osz_spec_t h, const double *x, int64_t ldx, int64_t n, void *out, int64_t *nseg_out,
                  void *stream) {
    OSZ_REQUIRE(h, "osz_spec_push: null handle");
    OSZ_REQUIRE(n >= 0 && (n == 0 || (x && ldx >= n)), "osz_spec_push: bad input");
    OSZ_SAME_DEVICE(h, "osz_spec_push");
    hipStream_t st = as_stream(stream);
    const int64_t total = h->ncarry + n;
    const int64_t nseg = osz_spec_seg_count(h, n);
    OSZ_REQUIRE(nseg == 0 || h->mode == OSZ_SPEC_PSD_MEAN || out, "osz_spec_push: null output");
    if (nseg > 0 && (h->fused8 || h->mixed || h->split || h->blue)) {
        int rc = spec8_push(h, x, ldx, n, out, nseg, st);
        if (rc) return rc;
        h->count += nseg;
    } else if (nseg > 0 && h->fused) {
        const int64_t npairs = (nseg + 1) / 2;
        // run length: long runs (a workgroup reloads its twiddles and window and
        // publishes partial sums once per run) while ~512 workgroups remain
        int64_t R = (npairs * h->nch) / 512;
        if (R > 32) R = 32;
        if (R < 1) R = 1;
        const int64_t nruns = (npairs + R - 1) / R;
        FusedArgs fa{};
        fa.x = x ? x : h->dcarry[h->cur];
        fa.carry = h->dcarry[h->cur];
        fa.window = h->dwindow;
        fa.out = out;
        fa.ldx = ldx;
        fa.ncarry = h->ncarry;
        fa.ncap = h->ncap;
        fa.nseg = nseg;
        fa.stride = h->stride;
        fa.nch = h->nch;
        fa.detrend = h->detrend;
        fa.R = (int)R;
        fa.nruns = (int)nruns;
        fa.scale = h->scale;
        fa.tb = h->tb;
        if (h->mode == OSZ_SPEC_PSD_MEAN) {
            const int64_t need = (int64_t)h->nch * nruns * kNF;
            if (need > h->partial_cap) {
                OSZ_HIP(hipStreamSynchronize(st));
                (void)hipFree(h->dpartial);
                h->dpartial = nullptr;
                if (hipMalloc(&h->dpartial, sizeof(double) * need) != hipSuccess)
                    return fail(OSZ_ERR_NOMEM, "osz_spec_push: partial sums (%lld doubles)", (long long)need);
                h->partial_cap = need;
            }
            fa.partial = h->dpartial;
        }
        using kern_t = void (*)(FusedArgs);
        static const kern_t ck[3][2][2] = {
            {{spec_cube_kernel<0, false, false>, spec_cube_kernel<0, false, true>},
             {spec_cube_kernel<0, true, false>, spec_cube_kernel<0, true, true>}},
            {{spec_cube_kernel<1, false, false>, spec_cube_kernel<1, false, true>},
             {spec_cube_kernel<1, true, false>, spec_cube_kernel<1, true, true>}},
            {{spec_cube_kernel<2, false, false>, spec_cube_kernel<2, false, true>},
             {spec_cube_kernel<2, true, false>, spec_cube_kernel<2, true, true>}}};
        const size_t clds = sizeof(fft::cube::C2) * (fft::cube::SLOTS + 240);     // the cube + pass 2's twiddle table
        const kern_t ckern = ck[h->mode][h->detrend == OSZ_DETREND_LINEAR ? 1 : 0]
                               [h->stride == 2048 ? 1 : 0];
        OSZ_DYN_LDS(ckern, clds);
        {
            KernelTimer kt("spec_fused", st);
            const dim3 grid((unsigned)nruns, h->nch), block(256);
            hipLaunchKernelGGL(ckern, grid, block, clds, st, fa);
        }
        OSZ_HIP(hipGetLastError());
        if (h->mode == OSZ_SPEC_PSD_MEAN) {
            hipLaunchKernelGGL(spec_partial_reduce_kernel, dim3((kNF + 255) / 256, h->nch), dim3(256),
                               0, st, h->dpartial, h->dsum, (int)nruns);
            OSZ_HIP(hipGetLastError());
        }
        h->count += nseg;
    } else if (nseg > 0) {
        int64_t segs_per_batch = kSpecMaxElems / ((int64_t)h->nch * h->nfft);
        if (segs_per_batch < 1) segs_per_batch = 1;
        if (segs_per_batch > nseg) segs_per_batch = nseg;
        if (segs_per_batch > 65535) segs_per_batch = 65535;
        const int64_t rows_needed = segs_per_batch * h->nch;
        if (rows_needed > h->rows_cap) {
            OSZ_HIP(hipStreamSynchronize(st));
            (void)hipFree(h->drows);
            (void)hipFree(h->dspec);
            h->drows = h->dspec = nullptr;
            hipError_t e1 = hipMalloc(&h->drows, sizeof(double) * (size_t)rows_needed * h->nfft);
            hipError_t e2 = hipMalloc(&h->dspec, sizeof(double) * 2 * (size_t)rows_needed * h->nfreq);
            if (e1 != hipSuccess || e2 != hipSuccess)
                return fail(OSZ_ERR_NOMEM, "osz_spec_push: staging for %lld rows", (long long)rows_needed);
            h->rows_cap = rows_needed;
        }
        rocfft_execution_info info = nullptr;
        OSZ_FFT(rocfft_execution_info_create(&info));
        OSZ_FFT(rocfft_execution_info_set_stream(info, st));
        for (int64_t s0 = 0; s0 < nseg; s0 += segs_per_batch) {
            const int64_t ns = (nseg - s0 < segs_per_batch) ? nseg - s0 : segs_per_batch;
            const int64_t rows = ns * h->nch;
            PrepArgs pa{};
            pa.x = x ? x : h->dcarry[h->cur];
            pa.carry = h->dcarry[h->cur];
            pa.rows = h->drows;
            pa.window = h->dwindow;
            pa.ldx = ldx;
            pa.ncarry = h->ncarry;
            pa.ncap = h->ncap;
            pa.seg0 = s0;
            pa.nwin = h->nwin;
            pa.nfft = h->nfft;
            pa.stride = h->stride;
            pa.nch = h->nch;
            pa.detrend = h->detrend;
            {
                KernelTimer kt("spec_prep", st);
                hipLaunchKernelGGL(spec_prep_kernel, dim3((unsigned)ns, h->nch), dim3(256), 0, st,
                                   pa);
            }
            OSZ_HIP(hipGetLastError());
            rocfft_plan plan;
            int rc = spec_plan(h, rows, &plan);
            if (rc) return rc;
            size_t wb = 0;
            OSZ_FFT(rocfft_plan_get_work_buffer_size(plan, &wb));
            if (wb > h->work_cap) {
                OSZ_HIP(hipStreamSynchronize(st));
                (void)hipFree(h->dwork);
                h->dwork = nullptr;
                if (hipMalloc(&h->dwork, wb) != hipSuccess)
                    return fail(OSZ_ERR_NOMEM, "osz_spec_push: rocFFT work buffer %zu B", wb);
                h->work_cap = wb;
            }
            if (wb) OSZ_FFT(rocfft_execution_info_set_work_buffer(info, h->dwork, wb));
            void *in[1] = {h->drows};
            void *outb[1] = {h->dspec};
            {
                KernelTimer kt("spec_rocfft", st);
                OSZ_FFT(rocfft_execute(plan, in, outb, info));
            }
            PostArgs po{};
            po.spec = h->dspec;
            po.nseg = ns;
            po.nfreq = h->nfreq;
            po.nch = h->nch;
            po.nfft = h->nfft;
            po.mode = h->mode;
            po.scale = h->scale;
            if (h->mode == OSZ_SPEC_PSD_MEAN)
                po.out = h->dsum;
            else if (h->mode == OSZ_SPEC_PSD_SEGMENTS)
                po.out = (double *)out + (size_t)s0 * h->nch * h->nfreq;
            else
                po.out = (double *)out + 2 * (size_t)s0 * h->nch * h->nfreq;
            {
                KernelTimer kt("spec_post", st);
                hipLaunchKernelGGL(spec_post_kernel, dim3((h->nfreq + 255) / 256, h->nch),
                                   dim3(256), 0, st, po);
            }
            OSZ_HIP(hipGetLastError());
        }
        OSZ_FFT(rocfft_execution_info_destroy(info));
        h->count += nseg;
    }
    // new carry: everything from the start of the next segment on
    const int64_t consumed = nseg * h->stride;
    const int64_t nnew = total - consumed;
    if (n > 0 || consumed > 0) {
        if (nnew > 0) {
            hipLaunchKernelGGL(spec_carry_kernel, dim3((unsigned)((nnew + 255) / 256), h->nch),
                               dim3(256), 0, st, x ? x : h->dcarry[h->cur], ldx, h->dcarry[h->cur],
                               h->dcarry[h->cur ^ 1], h->ncap, h->ncarry, consumed, nnew);
            OSZ_HIP(hipGetLastError());
        }
        h->cur ^= 1;
        h->ncarry = nnew;
    }
    if (nseg_out) *nseg_out = nseg;
    return OSZ_OK;
}

int osz_spec_sum(osz_spec_t h, double **dsum, int64_t *count) {
    OSZ_REQUIRE(h && dsum && count, "osz_spec_sum: null argument");
    OSZ_REQUIRE(h->mode == OSZ_SPEC_PSD_MEAN, "osz_spec_sum: handle is not in PSD_MEAN mode");
    *dsum = h->dsum;
    *count = h->count;
    return OSZ_OK;
}

int osz_spec_export_sum(osz_spec_t h, double *dst, int64_t *count, void *stream) {
    OSZ_REQUIRE(h && dst && count, "osz_spec_export_sum: null argument");
    OSZ_REQUIRE(h->mode == OSZ_SPEC_PSD_MEAN, "osz_spec_export_sum: handle is not in PSD_MEAN mode");
    OSZ_HIP(hipMemcpyAsync(dst, h->dsum, sizeof(double) * (size_t)h->nch * h->nfreq,
                           hipMemcpyDeviceToDevice, as_stream(stream)));
    *count = h->count;
    return OSZ_OK;
}

// dst = sum / count on the device (the divide of estimators.py:149-152's mean)
__global__ void spec_mean_kernel(const double *sum, double *dst, size_t ne, double count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ne) dst[i] = sum[i] / count;
}

int osz_spec_mean_device(osz_spec_t h, double *dmean, int64_t *count, void *stream) {
    OSZ_REQUIRE(h && dmean && count, "osz_spec_mean_device: null argument");
    OSZ_REQUIRE(h->mode == OSZ_SPEC_PSD_MEAN, "osz_spec_mean_device: handle is not in PSD_MEAN mode");
    OSZ_SAME_DEVICE(h, "osz_spec_mean_device");
    const size_t ne = (size_t)h->nch * h->nfreq;
    hipLaunchKernelGGL(spec_mean_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0,
                       as_stream(stream), h->dsum, dmean, ne, h->count > 0 ? (double)h->count : 1.0);
    OSZ_HIP(hipGetLastError());
    *count = h->count;
    return OSZ_OK;
}

int osz_welch_reduce(osz_spec_t h, void *comm, void *stream) {
    OSZ_REQUIRE(h && comm, "osz_welch_reduce: null argument");
    OSZ_REQUIRE(h->mode == OSZ_SPEC_PSD_MEAN, "osz_welch_reduce: handle is not in PSD_MEAN mode");
    OSZ_SAME_DEVICE(h, "osz_welch_reduce");
    hipStream_t st = as_stream(stream);
    OSZ_HIP(hipMemcpyAsync(h->dcount, &h->count, sizeof(int64_t), hipMemcpyHostToDevice, st));
    int rc = rccl_allreduce_sum(h->dsum, (size_t)h->nch * h->nfreq, true, comm, st);
    if (rc) return rc;
    rc = rccl_allreduce_sum(h->dcount, 1, false, comm, st);
    if (rc) return rc;
    int64_t total = 0;
    OSZ_HIP(hipMemcpyAsync(&total, h->dcount, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    OSZ_HIP(hipStreamSynchronize(st));
    h->count = total;
    return OSZ_OK;
}

// ---- checkpoint / resume: [ncarry, count, carry (nch x ncap), sum (nch x nfreq)]
int64_t osz_spec_state_size(osz_spec_t h) {
    return h ? 2 + (int64_t)h->nch * h->ncap + (int64_t)h->nch * h->nfreq : -1;
}

int osz_spec_get_state(osz_spec_t h, double *state, void *stream) {
    OSZ_REQUIRE(h && state, "osz_spec_get_state: null argument");
    OSZ_SAME_DEVICE(h, "osz_spec_get_state");
    hipStream_t st = as_stream(stream);
    const size_t nc = (size_t)h->nch * h->ncap, ns = (size_t)h->nch * h->nfreq;
    state[0] = (double)h->ncarry;
    state[1] = (double)h->count;
    OSZ_HIP(hipMemcpyAsync(state + 2, h->dcarry[h->cur], sizeof(double) * nc, hipMemcpyDeviceToHost, st));
    OSZ_HIP(hipMemcpyAsync(state + 2 + nc, h->dsum, sizeof(double) * ns, hipMemcpyDeviceToHost, st));
    OSZ_HIP(hipStreamSynchronize(st));
    return OSZ_OK;
}

int osz_spec_set_state(osz_spec_t h, const double *state, void *stream) {
    OSZ_REQUIRE(h && state, "osz_spec_set_state: null argument");
    OSZ_SAME_DEVICE(h, "osz_spec_set_state");
    const int64_t ncarry = (int64_t)state[0], count = (int64_t)state[1];
    OSZ_REQUIRE(ncarry >= 0 && ncarry < h->nwin && count >= 0,
                "osz_spec_set_state: ncarry=%lld count=%lld", (long long)ncarry, (long long)count);
    hipStream_t st = as_stream(stream);
    const size_t nc = (size_t)h->nch * h->ncap, ns = (size_t)h->nch * h->nfreq;
    OSZ_HIP(hipMemcpyAsync(h->dcarry[h->cur], state + 2, sizeof(double) * nc, hipMemcpyHostToDevice, st));
    OSZ_HIP(hipMemcpyAsync(h->dsum, state + 2 + nc, sizeof(double) * ns, hipMemcpyHostToDevice, st));
    OSZ_HIP(hipStreamSynchronize(st));
    h->ncarry = ncarry;
    h->count = count;
    return OSZ_OK;
}

int osz_spec_mean(osz_spec_t h, double *mean, int64_t *count, void *stream) {
    OSZ_REQUIRE(h && mean && count, "osz_spec_mean: null argument");
    OSZ_REQUIRE(h->mode == OSZ_SPEC_PSD_MEAN, "osz_spec_mean: handle is not in PSD_MEAN mode");
    hipStream_t st = as_stream(stream);
    const size_t ne = (size_t)h->nch * h->nfreq;
    OSZ_HIP(hipMemcpyAsync(mean, h->dsum, sizeof(double) * ne, hipMemcpyDeviceToHost, st));
    OSZ_HIP(hipStreamSynchronize(st));
    if (h->count > 0)
        for (size_t i = 0; i < ne; ++i) mean[i] /= (double)h->count;
    *count = h->count;
    return OSZ_OK;
}

}  // extern "C"
