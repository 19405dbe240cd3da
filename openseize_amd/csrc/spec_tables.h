// spec_tables.h -- the host-side tables of the spectral FIR -> cascade kernel
// (chain_spec.hip), in long double, free of HIP so that tests/host/spec_host_check.cpp
// can build them with g++ and tests/test_spec_host.py can hold them against NumPy
// and run the block algorithm with them on the CPU.
//
// For taps h (wlen of them) and a cascade of biquads (reference layout
// b0 b1 b2 1 a1 a2, core/numerical.py:301-335):
//   NR    rows of 256 samples per block: the largest with 256 NR + wlen - 1 <= 3840, so
//         that row 15 of the 4096-sample window holds nothing but the cascade's ringing
//   modes lambda_q: one pole per conjugate pair, every real pole (a double pole, a pole
//         on or outside the unit circle, more than 6 modes: not eligible)
//   H     [4096][2]  FFT(h) * prod_s B_s / A_s at the 4096 bins, divided by 4096
//   M     [2 NM][64] mu = M y: y = the first 64 samples of row 15, mu_q = the amplitude of
//         mode q extrapolated to window sample 4096 (re rows, then im rows); least squares
//         on the basis Re(lambda^l), -Im(lambda^l) by Householder QR
//   P     [32][NM][2] lambda^(16 i), i < 16, then lambda^i, i < 16
//   L     [5][NM][2]  lambda^(256 r)
//   R     rows of 256 samples over which a burst exceeds the tail tolerance (kTailTol, 1e-15 of
//         the composite impulse response's norm unless the caller relaxes it), from the tail
//         energy of that response (as the two-sided tables below)
#pragma once

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <vector>

namespace osz {
namespace spec {

typedef long double ld_t;

constexpr int kFit = 64;      // samples of row 15 the fit reads (one wave)
constexpr int kRMax = 5;      // burst rows supported (the pair kernels)
constexpr int kRMaxN = 12;    // ... by the one-block kernel (chain_zpn_body.h: its instances hold 5, 8 or 12 rows)
// Where a burst is cut off, relative to the norm of the composite impulse response.  What is cut
// is an error proportional to the INPUT's magnitude (a block's ringing is driven by everything in
// it, offsets included, and only cancels between neighbouring blocks as far as both were kept):
// about 0.3 tol max|x| in the output.  1e-15 puts that at the level of the rounding the
// reference's own float64 recurrences leave on such an input (eps max|x| and up), so no stream --
// whatever offset, step or rail it carries, wherever in the stream -- sees more from the cut than
// from float64 itself.  (Rounds 3-4 cut at 1e-12 and guarded large offsets by a look at the
// first samples of the first chunk; an offset appearing later got the loose cut.  A caller that
// knows its data may still relax it: osz_chain_zp_tolerance.)
constexpr ld_t kTailTol = 1e-15L;
constexpr int kN = 4096;

struct Tables {
    bool eligible = false;
    int NR = 0, NM = 0, nm = 0, R = 0;
    double fit_ratio = 0.0;                  // min |R_ii| / max |R_ii| of the fit's QR
    std::vector<double> H, M, P, L;
};

// FFT of the zero-padded taps, 4096 points, decimation in time.  quarter: the response at the
// quarter-shifted bins 2 pi (k + 1/4) / 4096 instead (the taps twisted by e^{-i 2 pi m / 16384}
// first): what fft::nega's odd-frequency transform multiplies by.
inline void fir_spectrum(const double *taps, int ntaps, std::vector<ld_t> &fr, std::vector<ld_t> &fi,
                         bool quarter = false) {
    const ld_t PI = acosl(-1.0L);
    std::vector<ld_t> wc(kN / 2), ws(kN / 2);
    for (int j = 0; j < kN / 2; ++j) {
        const ld_t ang = -2.0L * PI * (ld_t)j / (ld_t)kN;
        wc[j] = cosl(ang);
        ws[j] = sinl(ang);
    }
    fr.assign(kN, 0.0L);
    fi.assign(kN, 0.0L);
    for (int m = 0; m < ntaps && m < kN; ++m) {           // bit-reversed load
        unsigned r = 0;
        for (int b = 0; b < 12; ++b) r |= ((unsigned)(m >> b) & 1u) << (11 - b);
        if (quarter) {
            const ld_t ang = -2.0L * PI * (ld_t)m / (4.0L * kN);
            fr[r] = (ld_t)taps[m] * cosl(ang);
            fi[r] = (ld_t)taps[m] * sinl(ang);
        } else {
            fr[r] = (ld_t)taps[m];
        }
    }
    for (int len = 2; len <= kN; len <<= 1) {
        const int half = len >> 1, tstep = kN / len;
        for (int base = 0; base < kN; base += len)
            for (int j = 0; j < half; ++j) {
                const ld_t c = wc[j * tstep], sn = ws[j * tstep];
                const int u = base + j, v = u + half;
                const ld_t tr = fr[v] * c - fi[v] * sn, ti = fr[v] * sn + fi[v] * c;
                fr[v] = fr[u] - tr;
                fi[v] = fi[u] - ti;
                fr[u] += tr;
                fi[u] += ti;
            }
    }
}

// P = pinv(B) by Householder QR; B is m x n (column major), m >= n, P is n x m (row
// major); returns min |R_ii| / max |R_ii| (0: rank deficient)
inline ld_t pinv_qr(std::vector<ld_t> B, int m, int n, std::vector<ld_t> &Pinv) {
    std::vector<ld_t> Qt((size_t)m * m, 0.0L);             // Q^T, accumulated
    for (int i = 0; i < m; ++i) Qt[(size_t)i * m + i] = 1.0L;
    auto Bm = [&](int i, int j) -> ld_t & { return B[(size_t)j * m + i]; };
    for (int k = 0; k < n; ++k) {
        ld_t nrm = 0.0L;
        for (int i = k; i < m; ++i) nrm += Bm(i, k) * Bm(i, k);
        nrm = sqrtl(nrm);
        if (nrm == 0.0L) return 0.0L;
        const ld_t alpha = Bm(k, k) > 0 ? -nrm : nrm;
        std::vector<ld_t> v(m, 0.0L);
        for (int i = k; i < m; ++i) v[i] = Bm(i, k);
        v[k] -= alpha;
        ld_t vv = 0.0L;
        for (int i = k; i < m; ++i) vv += v[i] * v[i];
        if (vv == 0.0L) continue;
        for (int j = k; j < n; ++j) {
            ld_t d = 0.0L;
            for (int i = k; i < m; ++i) d += v[i] * Bm(i, j);
            d = 2.0L * d / vv;
            for (int i = k; i < m; ++i) Bm(i, j) -= d * v[i];
        }
        for (int j = 0; j < m; ++j) {
            ld_t d = 0.0L;
            for (int i = k; i < m; ++i) d += v[i] * Qt[(size_t)i * m + j];
            d = 2.0L * d / vv;
            for (int i = k; i < m; ++i) Qt[(size_t)i * m + j] -= d * v[i];
        }
    }
    ld_t rmin = fabsl(Bm(0, 0)), rmax = rmin;
    for (int k = 1; k < n; ++k) {
        rmin = std::min(rmin, fabsl(Bm(k, k)));
        rmax = std::max(rmax, fabsl(Bm(k, k)));
    }
    if (rmin == 0.0L) return 0.0L;
    Pinv.assign((size_t)n * m, 0.0L);                      // R^-1 (Q^T)[0:n, :]
    for (int j = 0; j < m; ++j)
        for (int i = n - 1; i >= 0; --i) {
            ld_t acc = Qt[(size_t)i * m + j];
            for (int k = i + 1; k < n; ++k) acc -= Bm(i, k) * Pinv[(size_t)k * m + j];
            Pinv[(size_t)i * m + j] = acc / Bm(i, i);
        }
    return rmin / rmax;
}

struct Mode {
    ld_t re, im;
    bool real;
};

inline void mode_pow(const Mode &m, int e, ld_t &pr, ld_t &pi) {
    ld_t br = m.re, bi = m.im;
    pr = 1.0L;
    pi = 0.0L;
    for (; e > 0; e >>= 1) {
        if (e & 1) {
            const ld_t x = pr * br - pi * bi, y = pr * bi + pi * br;
            pr = x;
            pi = y;
        }
        const ld_t x = br * br - bi * bi, y = 2.0L * br * bi;
        br = x;
        bi = y;
    }
}

// sos: nsec rows of (b0 b1 b2 a0 a1 a2), a0 == 1; forgets: the cascade's memory is
// bounded (osz_sos_s::warm_len is a finite number of samples)
inline Tables build(const double *taps, int wlen, const double *sos, int nsec, bool forgets,
                    ld_t tail_tol = kTailTol) {
    Tables T;
    if (wlen < 2 || !forgets) return T;
    int NR = (3841 - wlen) / 256;
    if (NR > 15) NR = 15;
    if (NR < 8) return T;
    const int S = 256 * NR, D = 16 - NR;
    std::vector<Mode> modes;
    for (int q = 0; q < nsec; ++q) {
        const ld_t a1 = sos[6 * q + 4], a2 = sos[6 * q + 5];
        if (a1 == 0.0L && a2 == 0.0L) continue;
        if (a2 == 0.0L) {
            modes.push_back({-a1, 0.0L, true});
            continue;
        }
        const ld_t disc = a1 * a1 - 4.0L * a2;
        if (disc < 0.0L) {
            modes.push_back({-a1 / 2, sqrtl(-disc) / 2, false});
        } else if (disc > 0.0L) {
            modes.push_back({(-a1 + sqrtl(disc)) / 2, 0.0L, true});
            modes.push_back({(-a1 - sqrtl(disc)) / 2, 0.0L, true});
        } else {
            return T;                                      // a double pole: n lambda^n is not in the basis
        }
    }
    const int nm = (int)modes.size();
    if (nm < 1 || nm > 6) return T;       // the tables of 8 modes do not fit beside two cubes per CU
    for (auto &m : modes)
        if (!(m.re * m.re + m.im * m.im < 1.0L)) return T;
    const int NM = (nm + 1) & ~1;                          // instantiated: 2, 4, 6
    // least squares on the first 64 samples of row 15:
    //   hom[l] = sum_q Re(gamma_q lambda_q^l) = sum_q a_q Re(lambda^l) - b_q Im(lambda^l)
    std::vector<int> col_mode, col_im;
    for (int q = 0; q < nm; ++q) {
        col_mode.push_back(q);
        col_im.push_back(0);
        if (!modes[q].real) {
            col_mode.push_back(q);
            col_im.push_back(1);
        }
    }
    const int nd = (int)col_mode.size();
    std::vector<ld_t> B((size_t)kFit * nd);
    for (int j = 0; j < nd; ++j)
        for (int l = 0; l < kFit; ++l) {
            ld_t pr, pi;
            mode_pow(modes[col_mode[j]], l, pr, pi);
            B[(size_t)j * kFit + l] = col_im[j] ? -pi : pr;
        }
    std::vector<ld_t> Pinv;
    const ld_t ratio = pinv_qr(B, kFit, nd, Pinv);
    T.fit_ratio = (double)ratio;
    if (!(ratio > 1e-9L)) return T;                        // modes too close to tell apart
    // mu = gamma lambda^256
    T.M.assign((size_t)2 * NM * kFit, 0.0);
    for (int q = 0; q < nm; ++q) {
        int ja = -1, jb = -1;
        for (int j = 0; j < nd; ++j)
            if (col_mode[j] == q) (col_im[j] ? jb : ja) = j;
        ld_t cr, ci;
        mode_pow(modes[q], 256, cr, ci);
        for (int l = 0; l < kFit; ++l) {
            const ld_t av = Pinv[(size_t)ja * kFit + l];
            const ld_t bv = jb >= 0 ? Pinv[(size_t)jb * kFit + l] : 0.0L;
            T.M[(size_t)q * kFit + l] = (double)(cr * av - ci * bv);
            T.M[(size_t)(NM + q) * kFit + l] = (double)(ci * av + cr * bv);
        }
    }
    // composite impulse response (the taps, then the cascade): how far a burst reaches
    const int glen = kN + 256 * 17;
    std::vector<ld_t> g(glen, 0.0L);
    for (int i = 0; i < wlen && i < glen; ++i) g[i] = taps[i];
    for (int q = 0; q < nsec; ++q) {
        const ld_t b0 = sos[6 * q], b1 = sos[6 * q + 1], b2 = sos[6 * q + 2];
        const ld_t a1 = sos[6 * q + 4], a2 = sos[6 * q + 5];
        ld_t z0 = 0.0L, z1 = 0.0L;
        for (int i = 0; i < glen; ++i) {
            const ld_t xin = g[i], y = b0 * xin + z0;
            z0 = b1 * xin - a1 * y + z1;
            z1 = b2 * xin - a2 * y;
            g[i] = y;
        }
    }
    std::vector<ld_t> tail2(glen + 1, 0.0L);
    for (int i = glen - 1; i >= 0; --i) tail2[i] = tail2[i + 1] + g[i] * g[i];
    int R = 1;
    while (R <= 16 && kN + 256 * R - S < glen && sqrtl(tail2[kN + 256 * R - S]) > tail_tol * sqrtl(tail2[0]))
        ++R;
    // the +mu burst of a block must end inside the next block (D + R <= NR) and the closing
    // pair's accumulator holds 8192 samples (NR + R <= 16)
    if (R > kRMax || R > D || R > 2 * NR - 16) return T;
    // composite spectrum, divided by 4096 (the inverse transform is unnormalised)
    std::vector<ld_t> fr, fi;
    fir_spectrum(taps, wlen, fr, fi);
    const ld_t PI = acosl(-1.0L);
    T.H.assign(2 * kN, 0.0);
    for (int k = 0; k < kN; ++k) {
        const ld_t ang = -2.0L * PI * (ld_t)k / (ld_t)kN;
        const ld_t zr = cosl(ang), zi = sinl(ang);          // z^-1 at bin k
        const ld_t z2r = zr * zr - zi * zi, z2i = 2.0L * zr * zi;
        ld_t hr = fr[k] / kN, hi = fi[k] / kN;
        for (int q = 0; q < nsec; ++q) {
            const ld_t b0 = sos[6 * q], b1 = sos[6 * q + 1], b2 = sos[6 * q + 2];
            const ld_t a1 = sos[6 * q + 4], a2 = sos[6 * q + 5];
            const ld_t nr = b0 + b1 * zr + b2 * z2r, ni = b1 * zi + b2 * z2i;
            const ld_t dr = 1.0L + a1 * zr + a2 * z2r, di = a1 * zi + a2 * z2i;
            const ld_t den = dr * dr + di * di;
            const ld_t qr = (nr * dr + ni * di) / den, qi = (ni * dr - nr * di) / den;
            const ld_t x = hr * qr - hi * qi, y = hr * qi + hi * qr;
            hr = x;
            hi = y;
        }
        T.H[2 * k] = (double)hr;
        T.H[2 * k + 1] = (double)hi;
    }
    T.P.assign((size_t)32 * NM * 2, 0.0);
    T.L.assign((size_t)kRMax * NM * 2, 0.0);
    for (int q = 0; q < nm; ++q) {
        for (int i = 0; i < 32; ++i) {
            ld_t pr, pi;
            mode_pow(modes[q], i < 16 ? 16 * i : i - 16, pr, pi);
            T.P[((size_t)i * NM + q) * 2 + 0] = (double)pr;
            T.P[((size_t)i * NM + q) * 2 + 1] = (double)pi;
        }
        for (int r = 0; r < kRMax; ++r) {
            ld_t pr, pi;
            mode_pow(modes[q], 256 * r, pr, pi);
            T.L[((size_t)r * NM + q) * 2 + 0] = (double)pr;
            T.L[((size_t)r * NM + q) * 2 + 1] = (double)pi;
        }
    }
    T.NR = NR;
    T.NM = NM;
    T.nm = nm;
    T.R = R;
    T.eligible = true;
    return T;
}

// ---- the two-sided scheme (chain_zp.hip): FIR -> forward cascade -> backward cascade,
// i.e. the zero-phase filter |H_iir|^2 H_fir, as one multiplication per bin.  A block now
// rings on both sides: after its FIR output has ended (causal modes, as above) and BEFORE
// its first sample (the backward pass: the same modes, decaying towards the past).  In the
// circular window the left tail sits at the END of row 15, running down from sample 4095.
// One joint least-squares fit on the first and the last `nh` samples of row 15 gives both
// sets of amplitudes: mu_q (causal, extrapolated to window sample 4096) and nu_q
// (anticausal, at window sample 4095).
//   H     [4096][2]  FFT(h) |prod_s B_s / A_s|^2 / 4096
//   M     [4 NM][2 nh]  rows: Re mu, Im mu, Re nu, Im nu (NM each); columns: samples
//         l = 0 .. nh-1, then 256-nh .. 255 of row 15
//   P     [20][NM][2]  lambda^(32 i), i < 8; lambda^(4 i), i < 8; lambda^i, i < 4
//   L     [5][NM][2]   lambda^(256 r)
//   R, Rf burst rows backwards / forwards: over how many rows of 256 samples the left / right
//         tail of the two-sided composite impulse response exceeds kTailTol of its norm; the
//         right tail has the window's guard rows in front of it and is the shorter one
//         (Rf <= R).  What is cut off is the error the kernel adds to the transform's own
//         rounding (1e-14 of the output scale): for the headline filters the tails are
//         7.6e-13 / 7.3e-13 at R, Rf = 2, 1 and 3.8e-16 / 3.6e-16 at 3, 2 -- measured against
//         SciPy 5e-13 and 7e-15.  The default (kTailTol above) is the tighter pair: the cut
//         scales with the input's magnitude, not with the output's.
// NR is the largest block height whose guard rows hold the bursts (R <= D, D + Rf <= NR).

struct TablesZp {
    bool eligible = false;
    int NR = 0, NM = 0, nm = 0, R = 0, Rf = 0, nh = 0;
    int NS = 0;   // build_zpn: the slow modes (the first NS of NM, slowest first) -- the only ones alive a row away
    double fit_ratio = 0.0;
    std::vector<double> H, M, P, L;
};

// lds_budget: bytes the kernel may use behind the cube; the fit reads as many samples as its
// matrix then leaves room for.  Two workgroups share a CU's 160 KB only up to 80 896 bytes
// each (measured: at 81 152 the second one no longer fits and the kernel takes 2.4 instead of
// 1.8 ms), i.e. 15 360 behind the 64 KB cube.
inline TablesZp build_zp(const double *taps, int wlen, const double *sos, int nsec, bool forgets,
                         int lds_budget = 15360, ld_t tail_tol = kTailTol) {
    TablesZp T;
    if (wlen < 2 || !forgets) return T;
    std::vector<Mode> modes;
    for (int q = 0; q < nsec; ++q) {
        const ld_t a1 = sos[6 * q + 4], a2 = sos[6 * q + 5];
        if (a1 == 0.0L && a2 == 0.0L) continue;
        if (a2 == 0.0L) {
            modes.push_back({-a1, 0.0L, true});
            continue;
        }
        const ld_t disc = a1 * a1 - 4.0L * a2;
        if (disc < 0.0L) {
            modes.push_back({-a1 / 2, sqrtl(-disc) / 2, false});
        } else if (disc > 0.0L) {
            modes.push_back({(-a1 + sqrtl(disc)) / 2, 0.0L, true});
            modes.push_back({(-a1 - sqrtl(disc)) / 2, 0.0L, true});
        } else {
            return T;
        }
    }
    const int nm = (int)modes.size();
    if (nm < 1 || nm > 6) return T;
    for (auto &m : modes)
        if (!(m.re * m.re + m.im * m.im < 1.0L)) return T;
    const int NM = (nm + 1) & ~1;
    // two-sided composite impulse response g2[m], m = -Lg .. Lg-1, stored at index Lg + m
    const int Lg = kN + 256 * 17;
    std::vector<ld_t> g2(2 * Lg, 0.0L);
    for (int i = 0; i < wlen && i < Lg; ++i) g2[Lg + i] = taps[i];
    for (int pass = 0; pass < 2; ++pass) {
        for (int q = 0; q < nsec; ++q) {
            const ld_t b0 = sos[6 * q], b1 = sos[6 * q + 1], b2 = sos[6 * q + 2];
            const ld_t a1 = sos[6 * q + 4], a2 = sos[6 * q + 5];
            ld_t z0 = 0.0L, z1 = 0.0L;
            for (int i = 0; i < 2 * Lg; ++i) {
                const int idx = pass == 0 ? i : 2 * Lg - 1 - i;
                const ld_t xin = g2[idx], y = b0 * xin + z0;
                z0 = b1 * xin - a1 * y + z1;
                z1 = b2 * xin - a2 * y;
                g2[idx] = y;
            }
        }
    }
    std::vector<ld_t> right2(2 * Lg + 1, 0.0L), left2(2 * Lg + 1, 0.0L);   // tail energies
    for (int i = 2 * Lg - 1; i >= 0; --i) right2[i] = right2[i + 1] + g2[i] * g2[i];
    for (int i = 0; i < 2 * Lg; ++i) left2[i + 1] = left2[i] + g2[i] * g2[i];
    const ld_t tot = sqrtl(right2[0]);
    int NRmax = (3841 - wlen) / 256;
    if (NRmax > 15) NRmax = 15;
    int NR = 0, R = 0, Rf = 0;
    for (int cand = NRmax; cand >= 8 && !NR; --cand) {
        const int S = 256 * cand, D = 16 - cand;
        int rb = 0, rf = 0;
        for (int r = 1; r <= kRMax && !(rb && rf); ++r) {
            const int ir = Lg + kN + 256 * r - S + 1, il = Lg - 256 * r;
            if (ir >= 2 * Lg || il < 0) break;
            if (!rf && sqrtl(right2[ir]) <= tail_tol * tot) rf = r;
            if (!rb && sqrtl(left2[il]) <= tail_tol * tot) rb = r;
        }
        if (rb && rf && rf <= rb && rb <= D && D + rf <= cand) {
            NR = cand;
            R = rb;
            Rf = rf;
        }
    }
    if (!NR) return T;
    // LDS behind the cube: fit samples [2][2 nh], kappa [8][R][NM][2], L [R][NM][2],
    // P [20][NM][2], M [4 NM][2 nh]
    int nh = 0;
    for (int cand = 32; cand >= 16 && !nh; cand -= 8) {     // 32, 24, 16: the kernel unrolls these
        const int bytes = 8 * (2 * 2 * cand + 8 * R * NM * 2 + R * NM * 2 + 20 * NM * 2 +
                               4 * NM * 2 * cand);
        if (bytes <= lds_budget) nh = cand;
    }
    if (!nh) return T;
    const int ns = 2 * nh;
    std::vector<int> col_mode, col_kind;      // kind: 0 Re causal, 1 Im causal, 2 Re anticausal, 3 Im anticausal
    for (int side = 0; side < 2; ++side)
        for (int q = 0; q < nm; ++q) {
            col_mode.push_back(q);
            col_kind.push_back(2 * side);
            if (!modes[q].real) {
                col_mode.push_back(q);
                col_kind.push_back(2 * side + 1);
            }
        }
    const int nd = (int)col_mode.size();
    std::vector<ld_t> B((size_t)ns * nd);
    for (int j = 0; j < nd; ++j)
        for (int i = 0; i < ns; ++i) {
            const int l = i < nh ? i : 256 - ns + i;
            ld_t pr, pi;
            mode_pow(modes[col_mode[j]], col_kind[j] < 2 ? l : 255 - l, pr, pi);
            B[(size_t)j * ns + i] = (col_kind[j] & 1) ? -pi : pr;
        }
    std::vector<ld_t> Pinv;
    const ld_t ratio = pinv_qr(B, ns, nd, Pinv);
    T.fit_ratio = (double)ratio;
    // modes the samples cannot tell apart, or too few samples for them (16 + 16 for twelve
    // two-sided modes lose three digits: 7e-12 against SciPy where 24 + 24 give 1e-14; the
    // ratio of R's extreme diagonal entries is 3e-5 there, 4e-3 here)
    if (!(ratio > 1e-3L)) return T;
    T.M.assign((size_t)4 * NM * ns, 0.0);
    for (int side = 0; side < 2; ++side)
        for (int q = 0; q < nm; ++q) {
            int ja = -1, jb = -1;
            for (int j = 0; j < nd; ++j)
                if (col_mode[j] == q && col_kind[j] / 2 == side) ((col_kind[j] & 1) ? jb : ja) = j;
            ld_t cr = 1.0L, ci = 0.0L;
            if (side == 0) mode_pow(modes[q], 256, cr, ci);       // mu = gamma lambda^256; nu as fitted
            for (int i = 0; i < ns; ++i) {
                const ld_t av = Pinv[(size_t)ja * ns + i];
                const ld_t bv = jb >= 0 ? Pinv[(size_t)jb * ns + i] : 0.0L;
                T.M[(size_t)((2 * side) * NM + q) * ns + i] = (double)(cr * av - ci * bv);
                T.M[(size_t)((2 * side + 1) * NM + q) * ns + i] = (double)(ci * av + cr * bv);
            }
        }
    // zero-phase composite spectrum / 4096
    std::vector<ld_t> fr, fi;
    fir_spectrum(taps, wlen, fr, fi);
    const ld_t PI = acosl(-1.0L);
    T.H.assign(2 * kN, 0.0);
    for (int k = 0; k < kN; ++k) {
        const ld_t ang = -2.0L * PI * (ld_t)k / (ld_t)kN;
        const ld_t zr = cosl(ang), zi = sinl(ang);
        const ld_t z2r = zr * zr - zi * zi, z2i = 2.0L * zr * zi;
        ld_t gain = 1.0L;
        for (int q = 0; q < nsec; ++q) {
            const ld_t b0 = sos[6 * q], b1 = sos[6 * q + 1], b2 = sos[6 * q + 2];
            const ld_t a1 = sos[6 * q + 4], a2 = sos[6 * q + 5];
            const ld_t nr = b0 + b1 * zr + b2 * z2r, ni = b1 * zi + b2 * z2i;
            const ld_t dr = 1.0L + a1 * zr + a2 * z2r, di = a1 * zi + a2 * z2i;
            gain *= (nr * nr + ni * ni) / (dr * dr + di * di);
        }
        T.H[2 * k] = (double)(fr[k] / kN * gain);
        T.H[2 * k + 1] = (double)(fi[k] / kN * gain);
    }
    T.P.assign((size_t)20 * NM * 2, 0.0);
    T.L.assign((size_t)kRMax * NM * 2, 0.0);
    for (int q = 0; q < nm; ++q) {
        for (int i = 0; i < 20; ++i) {
            ld_t pr, pi;
            mode_pow(modes[q], i < 8 ? 32 * i : i < 16 ? 4 * (i - 8) : i - 16, pr, pi);
            T.P[((size_t)i * NM + q) * 2 + 0] = (double)pr;
            T.P[((size_t)i * NM + q) * 2 + 1] = (double)pi;
        }
        for (int r = 0; r < kRMax; ++r) {
            ld_t pr, pi;
            mode_pow(modes[q], 256 * r, pr, pi);
            T.L[((size_t)r * NM + q) * 2 + 0] = (double)pr;
            T.L[((size_t)r * NM + q) * 2 + 1] = (double)pi;
        }
    }
    T.NR = NR;
    T.NM = NM;
    T.nm = nm;
    T.R = R;
    T.Rf = Rf;
    T.nh = nh;
    T.eligible = true;
    return T;
}


// ---- the two-sided scheme on ONE real block per transform (chain_zpn.hip) ---------------
// fft::nega (fft4096.h) takes a window of 8192 samples through the 4096-point transform at the
// odd frequencies: the filter's tail (wlen - 1 samples) and the guard row are paid once per
// 8192 instead of once per 4096 samples -- a block is NB <= 27 rows of 256 at 1024 taps where
// the pair trick has 2 x 11 -- and the fit and the bursts are one block's, not two.  What
// changes in the tables:
//   NB    rows per block: the largest with 256 NB + wlen - 1 <= 7936 (row 31 holds nothing but
//         ringing) whose guard rows hold the bursts (R <= D = 32 - NB, D + Rf <= NB); 24 .. 30
//   H     [4096][2]  the zero-phase composite response at 2 pi (j + 1/4) / 4096, / 4096
//   M     fitted on row 31; the wrap is NEGACYCLIC, so what sits at the window's end is MINUS
//         the left tail: the nu rows carry the sign, the kernel sees true amplitudes and ADDS
//         its in-window corrections
//   NS    the modes are sorted by radius, slowest first, and NS of them are "slow": still above
//         the tolerance a guard row (257 samples) behind where they start.  Every burst but the
//         left tail's first row starts at least that far out, so the right tail's amplitudes
//         mu are kept for the slow modes only (M: [2 NS] rows mu, then [2 NM] rows nu) and the
//         bursts run over NS modes -- all NM only in the left tail's first row.  (The FIT still
//         has every mode in its basis: the fast ones are alive in row 31's first samples.)
//         NM is 2, 4, 6 or 8, NS 2, 4 or 6.
//   R     up to kRMaxN = 12 rows (a left tail of 3072 samples) where the guard rows hold them:
//         R <= D = 32 - NB, i.e. blocks of 24 rows for R = 8, of 20 for R = 12 -- the cascade alone
//         (the identity as the FIR: nothing but the guard row in front of the left tail) needs
//         them first, the more so at the default cut of 1e-15.
inline TablesZp build_zpn(const double *taps, int wlen, const double *sos, int nsec, bool forgets,
                          int lds_budget = 15360, ld_t tail_tol = kTailTol, ld_t fit_floor = 3e-7L,
                          int rmax = kRMaxN) {
    TablesZp T;
    constexpr int kM = 8192;
    if (wlen < 2 || !forgets) return T;
    std::vector<Mode> modes;
    for (int q = 0; q < nsec; ++q) {
        const ld_t a1 = sos[6 * q + 4], a2 = sos[6 * q + 5];
        if (a1 == 0.0L && a2 == 0.0L) continue;
        if (a2 == 0.0L) {
            modes.push_back({-a1, 0.0L, true});
            continue;
        }
        const ld_t disc = a1 * a1 - 4.0L * a2;
        if (disc < 0.0L) {
            modes.push_back({-a1 / 2, sqrtl(-disc) / 2, false});
        } else if (disc > 0.0L) {
            modes.push_back({(-a1 + sqrtl(disc)) / 2, 0.0L, true});
            modes.push_back({(-a1 - sqrtl(disc)) / 2, 0.0L, true});
        } else {
            return T;
        }
    }
    const int nm = (int)modes.size();
    if (nm < 1 || nm > 8) return T;
    for (auto &m : modes)
        if (!(m.re * m.re + m.im * m.im < 1.0L)) return T;
    std::stable_sort(modes.begin(), modes.end(), [](const Mode &a, const Mode &b) {
        return a.re * a.re + a.im * a.im > b.re * b.re + b.im * b.im;
    });
    const int NM = (nm + 1) & ~1;
    const int Lg = kM + 256 * 17;
    std::vector<ld_t> g2(2 * Lg, 0.0L);
    for (int i = 0; i < wlen && i < Lg; ++i) g2[Lg + i] = taps[i];
    for (int pass = 0; pass < 2; ++pass) {
        for (int q = 0; q < nsec; ++q) {
            const ld_t b0 = sos[6 * q], b1 = sos[6 * q + 1], b2 = sos[6 * q + 2];
            const ld_t a1 = sos[6 * q + 4], a2 = sos[6 * q + 5];
            ld_t z0 = 0.0L, z1 = 0.0L;
            for (int i = 0; i < 2 * Lg; ++i) {
                const int idx = pass == 0 ? i : 2 * Lg - 1 - i;
                const ld_t xin = g2[idx], y = b0 * xin + z0;
                z0 = b1 * xin - a1 * y + z1;
                z1 = b2 * xin - a2 * y;
                g2[idx] = y;
            }
        }
    }
    std::vector<ld_t> right2(2 * Lg + 1, 0.0L), left2(2 * Lg + 1, 0.0L);
    for (int i = 2 * Lg - 1; i >= 0; --i) right2[i] = right2[i + 1] + g2[i] * g2[i];
    for (int i = 0; i < 2 * Lg; ++i) left2[i + 1] = left2[i] + g2[i] * g2[i];
    const ld_t tot = sqrtl(right2[0]);
    int NBmax = (7937 - wlen) / 256;
    if (NBmax > 30) NBmax = 30;
    int NB = 0, R = 0, Rf = 0;
    for (int cand = NBmax; cand >= 20 && !NB; --cand) {
        const int S = 256 * cand, D = 32 - cand;
        int rb = 0, rf = 0;
        for (int r = 1; r <= rmax && !(rb && rf); ++r) {
            const int ir = Lg + kM + 256 * r - S + 1, il = Lg - 256 * r;
            if (ir >= 2 * Lg || il < 0) break;
            if (!rf && sqrtl(right2[ir]) <= tail_tol * tot) rf = r;
            if (!rb && sqrtl(left2[il]) <= tail_tol * tot) rb = r;
        }
        // (the right tail: five rows at most, and its second landing place, rows D .. D + rf - 1,
        // inside the window's lower half, which the kernel's forward bursts address)
        if (rb && rf && rf <= rb && rf <= kRMax && rb <= D && D + rf <= cand && D + rf <= 16) {
            NB = cand;
            R = rb;
            Rf = rf;
        }
    }
    if (!NB) return T;
    // Slow modes: the two tails of the composite response are exact mode sums, g2[wlen - 1 + n] =
    // Re sum c_q lambda^n and g2[-1 - n] = Re sum d_q lambda^n (n >= 0); with the residues from
    // a least-squares fit on 64 samples of each, mode q's share of a tail beyond distance 257 is
    // at most |c_q| |lambda_q|^257 / sqrt(1 - |lambda_q|^2).
    int ns_needed = 0;
    {
        std::vector<int> cm, ck;
        for (int q = 0; q < nm; ++q) {
            cm.push_back(q);
            ck.push_back(0);
            if (!modes[q].real) {
                cm.push_back(q);
                ck.push_back(1);
            }
        }
        const int ndh = (int)cm.size(), nfit = 64;
        std::vector<ld_t> Bh((size_t)nfit * ndh);
        for (int j = 0; j < ndh; ++j)
            for (int i = 0; i < nfit; ++i) {
                ld_t pr, pi;
                mode_pow(modes[cm[j]], i, pr, pi);
                Bh[(size_t)j * nfit + i] = ck[j] ? -pi : pr;
            }
        std::vector<ld_t> Ph;
        if (pinv_qr(Bh, nfit, ndh, Ph) == 0.0L) return T;
        for (int q = 0; q < nm; ++q) {
            ld_t worst = 0.0L;
            for (int side = 0; side < 2; ++side) {
                ld_t cr = 0.0L, ci = 0.0L;
                for (int j = 0; j < ndh; ++j)
                    if (cm[j] == q)
                        for (int i = 0; i < nfit; ++i) {
                            const ld_t gv = side == 0 ? g2[Lg + wlen - 1 + i] : g2[Lg - 1 - i];
                            (ck[j] ? ci : cr) += Ph[(size_t)j * nfit + i] * gv;
                        }
                worst = std::max(worst, sqrtl(cr * cr + ci * ci));
            }
            const ld_t rad2 = modes[q].re * modes[q].re + modes[q].im * modes[q].im;
            const ld_t share = worst * powl(sqrtl(rad2), 257.0L) / sqrtl(1.0L - rad2);
            if (share > tail_tol * tot / (ld_t)(4 * nm)) ns_needed = q + 1;     // (sorted: slow modes first)
        }
    }
    int NS = (std::max(ns_needed, 1) + 1) & ~1;
    if (NS > NM) NS = NM;
    if (NS > 6) return T;                 // (eight slow modes: no instance)
    // LDS behind the cube: fit samples [2 nh], kappa: this block's and the previous block's mu
    // [2 parity][R][NS][2] and this block's nu [R][NM][2], lambda^256 [NM][2], P [20][NM][2],
    // M [2 NS + 2 NM][2 nh]
    int nh = 0;
    for (int cand = 32; cand >= 16 && !nh; cand -= 8) {
        const int bytes = 8 * (2 * cand + R * (2 * NS + NM) * 2 + NM * 2 + 20 * NM * 2 +
                               (2 * NS + 2 * NM) * 2 * cand);
        if (bytes <= lds_budget) nh = cand;
    }
    if (!nh) return T;
    const int ns = 2 * nh;
    std::vector<int> col_mode, col_kind;
    for (int side = 0; side < 2; ++side)
        for (int q = 0; q < nm; ++q) {
            col_mode.push_back(q);
            col_kind.push_back(2 * side);
            if (!modes[q].real) {
                col_mode.push_back(q);
                col_kind.push_back(2 * side + 1);
            }
        }
    const int nd = (int)col_mode.size();
    std::vector<ld_t> B((size_t)ns * nd);
    for (int j = 0; j < nd; ++j)
        for (int i = 0; i < ns; ++i) {
            const int l = i < nh ? i : 256 - ns + i;
            ld_t pr, pi;
            mode_pow(modes[col_mode[j]], col_kind[j] < 2 ? l : 255 - l, pr, pi);
            B[(size_t)j * ns + i] = (col_kind[j] & 1) ? -pi : pr;
        }
    std::vector<ld_t> Pinv;
    const ld_t ratio = pinv_qr(B, ns, nd, Pinv);
    T.fit_ratio = (double)ratio;
    // (the fit's conditioning: the transform's rounding, 1e-16 of the signal, reaches the amplitudes
    // multiplied by about 1 / ratio -- tests/test_spec_host.py measures up to 1e-16 / ratio of the
    // output scale (tests/test_gpu_zp.py); 3e-7 keeps that at 3e-10.  The pair kernel's floor of 1e-3 refused six-section
    // band-passes below 0.03 of Nyquist and every eight-section one.)
    if (!(ratio > fit_floor)) return T;
    // rows: Re mu [NS], Im mu [NS], Re nu [NM], Im nu [NM]
    T.M.assign((size_t)(2 * NS + 2 * NM) * ns, 0.0);
    for (int side = 0; side < 2; ++side)
        for (int q = 0; q < nm; ++q) {
            if (side == 0 && q >= NS) continue;                   // a fast mode's mu is nothing a row away
            int ja = -1, jb = -1;
            for (int j = 0; j < nd; ++j)
                if (col_mode[j] == q && col_kind[j] / 2 == side) ((col_kind[j] & 1) ? jb : ja) = j;
            ld_t cr = -1.0L, ci = 0.0L;                           // nu = -(what the window's end shows)
            if (side == 0) mode_pow(modes[q], 256, cr, ci);       // mu = gamma lambda^256
            const size_t rre = side == 0 ? (size_t)q : (size_t)(2 * NS + q);
            const size_t rim = side == 0 ? (size_t)(NS + q) : (size_t)(2 * NS + NM + q);
            for (int i = 0; i < ns; ++i) {
                const ld_t av = Pinv[(size_t)ja * ns + i];
                const ld_t bv = jb >= 0 ? Pinv[(size_t)jb * ns + i] : 0.0L;
                T.M[rre * ns + i] = (double)(cr * av - ci * bv);
                T.M[rim * ns + i] = (double)(ci * av + cr * bv);
            }
        }
    // the composite response at the quarter-shifted bins, / 4096 (the taps' part by the twisted
    // transform: a Horner sum per bin costs 4096 x wlen long-double products, 20 ms per stream)
    const ld_t PI = acosl(-1.0L);
    std::vector<ld_t> frq, fiq;
    fir_spectrum(taps, wlen, frq, fiq, true);
    T.H.assign(2 * kN, 0.0);
    for (int k = 0; k < kN; ++k) {
        const ld_t ang = -2.0L * PI * ((ld_t)k + 0.25L) / (ld_t)kN;
        const ld_t zr = cosl(ang), zi = sinl(ang);          // z^-1
        const ld_t z2r = zr * zr - zi * zi, z2i = 2.0L * zr * zi;
        const ld_t fr = frq[k], fi = fiq[k];
        ld_t gain = 1.0L;
        for (int q = 0; q < nsec; ++q) {
            const ld_t b0 = sos[6 * q], b1 = sos[6 * q + 1], b2 = sos[6 * q + 2];
            const ld_t a1 = sos[6 * q + 4], a2 = sos[6 * q + 5];
            const ld_t nr = b0 + b1 * zr + b2 * z2r, ni = b1 * zi + b2 * z2i;
            const ld_t dr = 1.0L + a1 * zr + a2 * z2r, di = a1 * zi + a2 * z2i;
            gain *= (nr * nr + ni * ni) / (dr * dr + di * di);
        }
        T.H[2 * k] = (double)(fr / kN * gain);
        T.H[2 * k + 1] = (double)(fi / kN * gain);
    }
    T.P.assign((size_t)20 * NM * 2, 0.0);
    T.L.assign((size_t)rmax * NM * 2, 0.0);
    for (int q = 0; q < nm; ++q) {
        for (int i = 0; i < 20; ++i) {
            ld_t pr, pi;
            mode_pow(modes[q], i < 8 ? 32 * i : i < 16 ? 4 * (i - 8) : i - 16, pr, pi);
            T.P[((size_t)i * NM + q) * 2 + 0] = (double)pr;
            T.P[((size_t)i * NM + q) * 2 + 1] = (double)pi;
        }
        for (int r = 0; r < rmax; ++r) {
            ld_t pr, pi;
            mode_pow(modes[q], 256 * r, pr, pi);
            T.L[((size_t)r * NM + q) * 2 + 0] = (double)pr;
            T.L[((size_t)r * NM + q) * 2 + 1] = (double)pi;
        }
    }
    T.NR = NB;
    T.NM = NM;
    T.nm = nm;
    T.R = R;
    T.Rf = Rf;
    T.nh = nh;
    T.NS = NS;
    T.eligible = true;
    return T;
}


// ---- the forward chain (FIR -> sosfilt) on one real block per transform ------------------
// The causal half of build_zpn: the composite response is the taps through the cascade, one
// sided; only the right tail rings, wraps (negacyclic: with its sign changed) and is fitted --
// on row 31, which holds nothing else -- and continued into the next block.  No left tail: no
// output lag, no rows held back.
//   NR    rows per block (24 .. 30): the largest with 256 NR + wlen - 1 <= 7936 and D + Rf <= NR
//   R=Rf  burst rows of the right tail (<= 5)
//   H     [4096][2]  H_fir H_iir at 2 pi (j + 1/4) / 4096, / 4096
//   M     [2 NS][2 nh]  mu of the slow modes (Re rows, Im rows) from the first and last nh samples
//         of row 31, every mode in the fit's basis
//   P, L  as above
inline TablesZp build_specn(const double *taps, int wlen, const double *sos, int nsec, bool forgets,
                            int lds_budget = 15360 - 1024, ld_t tail_tol = kTailTol, ld_t fit_floor = 3e-7L) {
    TablesZp T;
    constexpr int kM = 8192;
    if (wlen < 2 || !forgets) return T;
    std::vector<Mode> modes;
    for (int q = 0; q < nsec; ++q) {
        const ld_t a1 = sos[6 * q + 4], a2 = sos[6 * q + 5];
        if (a1 == 0.0L && a2 == 0.0L) continue;
        if (a2 == 0.0L) {
            modes.push_back({-a1, 0.0L, true});
            continue;
        }
        const ld_t disc = a1 * a1 - 4.0L * a2;
        if (disc < 0.0L) {
            modes.push_back({-a1 / 2, sqrtl(-disc) / 2, false});
        } else if (disc > 0.0L) {
            modes.push_back({(-a1 + sqrtl(disc)) / 2, 0.0L, true});
            modes.push_back({(-a1 - sqrtl(disc)) / 2, 0.0L, true});
        } else {
            return T;
        }
    }
    const int nm = (int)modes.size();
    if (nm < 1 || nm > 8) return T;
    for (auto &m : modes)
        if (!(m.re * m.re + m.im * m.im < 1.0L)) return T;
    std::stable_sort(modes.begin(), modes.end(), [](const Mode &a, const Mode &b) {
        return a.re * a.re + a.im * a.im > b.re * b.re + b.im * b.im;
    });
    const int NM = (nm + 1) & ~1;
    const int glen = kM + 256 * 17;
    std::vector<ld_t> g(glen, 0.0L);
    for (int i = 0; i < wlen && i < glen; ++i) g[i] = taps[i];
    for (int q = 0; q < nsec; ++q) {
        const ld_t b0 = sos[6 * q], b1 = sos[6 * q + 1], b2 = sos[6 * q + 2];
        const ld_t a1 = sos[6 * q + 4], a2 = sos[6 * q + 5];
        ld_t z0 = 0.0L, z1 = 0.0L;
        for (int i = 0; i < glen; ++i) {
            const ld_t xin = g[i], y = b0 * xin + z0;
            z0 = b1 * xin - a1 * y + z1;
            z1 = b2 * xin - a2 * y;
            g[i] = y;
        }
    }
    std::vector<ld_t> tail2(glen + 1, 0.0L);
    for (int i = glen - 1; i >= 0; --i) tail2[i] = tail2[i + 1] + g[i] * g[i];
    const ld_t tot = sqrtl(tail2[0]);
    int NBmax = (7937 - wlen) / 256;
    if (NBmax > 30) NBmax = 30;
    int NB = 0, Rf = 0;
    for (int cand = NBmax; cand >= 24 && !NB; --cand) {
        const int S = 256 * cand, D = 32 - cand;
        int rf = 0;
        for (int r = 1; r <= kRMax && !rf; ++r) {
            const int ir = kM + 256 * r - S + 1;
            if (ir >= glen) break;
            if (sqrtl(tail2[ir]) <= tail_tol * tot) rf = r;
        }
        if (rf && D + rf <= cand) {
            NB = cand;
            Rf = rf;
        }
    }
    if (!NB) return T;
    // slow modes: residues of the right tail g[wlen - 1 + n] = Re sum c_q lambda^n
    int ns_needed = 0;
    {
        std::vector<int> cm, ck;
        for (int q = 0; q < nm; ++q) {
            cm.push_back(q);
            ck.push_back(0);
            if (!modes[q].real) {
                cm.push_back(q);
                ck.push_back(1);
            }
        }
        const int ndh = (int)cm.size(), nfit = 64;
        std::vector<ld_t> Bh((size_t)nfit * ndh);
        for (int j = 0; j < ndh; ++j)
            for (int i = 0; i < nfit; ++i) {
                ld_t pr, pi;
                mode_pow(modes[cm[j]], i, pr, pi);
                Bh[(size_t)j * nfit + i] = ck[j] ? -pi : pr;
            }
        std::vector<ld_t> Ph;
        if (pinv_qr(Bh, nfit, ndh, Ph) == 0.0L) return T;
        for (int q = 0; q < nm; ++q) {
            ld_t cr = 0.0L, ci = 0.0L;
            for (int j = 0; j < ndh; ++j)
                if (cm[j] == q)
                    for (int i = 0; i < nfit; ++i) (ck[j] ? ci : cr) += Ph[(size_t)j * nfit + i] * g[wlen - 1 + i];
            const ld_t rad2 = modes[q].re * modes[q].re + modes[q].im * modes[q].im;
            const ld_t share = sqrtl(cr * cr + ci * ci) * powl(sqrtl(rad2), 257.0L) / sqrtl(1.0L - rad2);
            if (share > tail_tol * tot / (ld_t)(4 * nm)) ns_needed = q + 1;
        }
    }
    int NS = (std::max(ns_needed, 1) + 1) & ~1;
    if (NS > NM) NS = NM;
    if (NS > 6) return T;
    // LDS behind the cube: fit samples [2 nh], this block's and the previous block's mu
    // [2 parity][Rf][NS][2], lambda^256 [NM][2], P [20][NM][2], M [2 NS][2 nh]
    int nh = 0;
    for (int cand = 32; cand >= 16 && !nh; cand -= 8) {
        const int bytes = 8 * (2 * cand + Rf * 2 * NS * 2 + NM * 2 + 20 * NM * 2 + 2 * NS * 2 * cand);
        if (bytes <= lds_budget) nh = cand;
    }
    if (!nh) return T;
    const int ns = 2 * nh;
    std::vector<int> col_mode, col_im;
    for (int q = 0; q < nm; ++q) {
        col_mode.push_back(q);
        col_im.push_back(0);
        if (!modes[q].real) {
            col_mode.push_back(q);
            col_im.push_back(1);
        }
    }
    const int nd = (int)col_mode.size();
    std::vector<ld_t> B((size_t)ns * nd);
    for (int j = 0; j < nd; ++j)
        for (int i = 0; i < ns; ++i) {
            const int l = i < nh ? i : 256 - ns + i;
            ld_t pr, pi;
            mode_pow(modes[col_mode[j]], l, pr, pi);
            B[(size_t)j * ns + i] = col_im[j] ? -pi : pr;
        }
    std::vector<ld_t> Pinv;
    const ld_t ratio = pinv_qr(B, ns, nd, Pinv);
    T.fit_ratio = (double)ratio;
    if (!(ratio > fit_floor)) return T;
    T.M.assign((size_t)2 * NS * ns, 0.0);
    for (int q = 0; q < nm && q < NS; ++q) {
        int ja = -1, jb = -1;
        for (int j = 0; j < nd; ++j)
            if (col_mode[j] == q) (col_im[j] ? jb : ja) = j;
        ld_t cr, ci;
        mode_pow(modes[q], 256, cr, ci);                      // mu = gamma lambda^256
        for (int i = 0; i < ns; ++i) {
            const ld_t av = Pinv[(size_t)ja * ns + i];
            const ld_t bv = jb >= 0 ? Pinv[(size_t)jb * ns + i] : 0.0L;
            T.M[(size_t)q * ns + i] = (double)(cr * av - ci * bv);
            T.M[(size_t)(NS + q) * ns + i] = (double)(ci * av + cr * bv);
        }
    }
    const ld_t PI = acosl(-1.0L);
    std::vector<ld_t> frq, fiq;
    fir_spectrum(taps, wlen, frq, fiq, true);
    T.H.assign(2 * kN, 0.0);
    for (int k = 0; k < kN; ++k) {
        const ld_t ang = -2.0L * PI * ((ld_t)k + 0.25L) / (ld_t)kN;
        const ld_t zr = cosl(ang), zi = sinl(ang);
        const ld_t z2r = zr * zr - zi * zi, z2i = 2.0L * zr * zi;
        ld_t hr = frq[k] / kN, hi = fiq[k] / kN;
        for (int q = 0; q < nsec; ++q) {
            const ld_t b0 = sos[6 * q], b1 = sos[6 * q + 1], b2 = sos[6 * q + 2];
            const ld_t a1 = sos[6 * q + 4], a2 = sos[6 * q + 5];
            const ld_t nr = b0 + b1 * zr + b2 * z2r, ni = b1 * zi + b2 * z2i;
            const ld_t dr = 1.0L + a1 * zr + a2 * z2r, di = a1 * zi + a2 * z2i;
            const ld_t den = dr * dr + di * di;
            const ld_t qr = (nr * dr + ni * di) / den, qi = (ni * dr - nr * di) / den;
            const ld_t x = hr * qr - hi * qi, y = hr * qi + hi * qr;
            hr = x;
            hi = y;
        }
        T.H[2 * k] = (double)hr;
        T.H[2 * k + 1] = (double)hi;
    }
    T.P.assign((size_t)20 * NM * 2, 0.0);
    T.L.assign((size_t)kRMax * NM * 2, 0.0);
    for (int q = 0; q < nm; ++q) {
        for (int i = 0; i < 20; ++i) {
            ld_t pr, pi;
            mode_pow(modes[q], i < 8 ? 32 * i : i < 16 ? 4 * (i - 8) : i - 16, pr, pi);
            T.P[((size_t)i * NM + q) * 2 + 0] = (double)pr;
            T.P[((size_t)i * NM + q) * 2 + 1] = (double)pi;
        }
        for (int r = 0; r < kRMax; ++r) {
            ld_t pr, pi;
            mode_pow(modes[q], 256 * r, pr, pi);
            T.L[((size_t)r * NM + q) * 2 + 0] = (double)pr;
            T.L[((size_t)r * NM + q) * 2 + 1] = (double)pi;
        }
    }
    T.NR = NB;
    T.NM = NM;
    T.nm = nm;
    T.R = Rf;
    T.Rf = Rf;
    T.nh = nh;
    T.NS = NS;
    T.eligible = true;
    return T;
}

}  // namespace spec
}  // namespace osz
