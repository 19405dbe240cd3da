// chain_spec.h -- what the two spectral chain kernels share (chain_spec.hip: FIR ->
// forward cascade; chain_zp.hip: FIR -> forward -> backward cascade).
#pragma once

#include <vector>

#include "common.h"
#include "fft4096.h"
#include "sos_tile.h"
#include "spec_kept.h"
#include "spec_tables.h"

namespace osz {

constexpr int kSpecFit = spec::kFit;         // samples of row 15 the fit reads (one wave)
constexpr int kSpecRMax = spec::kRMax;       // burst rows supported
constexpr int kSpecLdc = 2 * 15 * 256;       // row pitch of the carry buffers (>= 2 S, >= 4096 + 256 R)

__device__ __forceinline__ double spec_qnan() { return __longlong_as_double(0x7ff8000000000000LL); }

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
