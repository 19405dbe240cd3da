// chain_spec.h -- what the two spectral chain kernels share (chain_spec.hip: FIR ->
// forward cascade; chain_zp.hip: FIR -> forward -> backward cascade).
#pragma once

#include <vector>

#include "common.h"
#include "fft4096.h"
#include "sos_tile.h"
#include "spec_tables.h"

namespace osz {

constexpr int kSpecFit = spec::kFit;         // samples of row 15 the fit reads (one wave)
constexpr int kSpecRMax = spec::kRMax;       // burst rows supported
constexpr int kSpecLdc = 2 * 15 * 256;       // row pitch of the carry buffers (>= 2 S, >= 4096 + 256 R)

__device__ __forceinline__ double spec_qnan() { return __longlong_as_double(0x7ff8000000000000LL); }

// NaN reach of the forward pass across runs (sos_tile.h, "NaN reach"): the
// workgroup of a channel that finishes last looks at the final output sample of
// every earlier run; from the first one that is not finite the rest of the chunk
// and the carry are NaN.
template <class EndOf>
__device__ __forceinline__ void spec_seal(int *__restrict__ segcnt, double *__restrict__ y, int64_t n,
                                          int nseg, EndOf end_of, double *__restrict__ carry_row, int c) {
    __syncthreads();
    int last = 0;
    if (threadIdx.x == 0) {
        __threadfence();
        last = atomicAdd(segcnt + c, 1) == nseg - 1;
    }
    last = __syncthreads_or(last);
    if (!last) return;
    if (threadIdx.x == 0) atomicExch(segcnt + c, 0);     // ready for the next launch
    __threadfence();
    int bad = nseg;
    for (int s = nseg - 2; s >= 0; --s) {
        const unsigned long long bits = __hip_atomic_load(
            reinterpret_cast<const unsigned long long *>(y + end_of(s) - 1), __ATOMIC_RELAXED,
            __HIP_MEMORY_SCOPE_AGENT);
        if (sos_not_finite(__longlong_as_double((long long)bits))) bad = s;
    }
    if (bad == nseg) return;
    const int64_t from = end_of(bad);
    sos_fill_nan(y + from, n - from);
    sos_fill_nan(carry_row, kSpecLdc);
}

// the composite spectrum in the order fft::cube2 leaves the bins in: [r][t] = H[bin(t, r)]
inline std::vector<double> spec_permuted_spectrum(const std::vector<double> &H) {
    std::vector<double> P(H.size());
    for (int r = 0; r < 16; ++r)
        for (int t = 0; t < 256; ++t) {
            const int k = fft::cube2::bin(t, r);
            P[2 * (256 * r + t)] = H[2 * k];
            P[2 * (256 * r + t) + 1] = H[2 * k + 1];
        }
    return P;
}

}  // namespace osz
