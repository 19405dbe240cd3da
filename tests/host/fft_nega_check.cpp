// CPU replay of fft::nega (openseize_amd/csrc/fft4096.h): one real block of 8192 samples on the
// 4096-point transform at the odd frequencies.  Checks: forward against a long-double DFT at
// w = 2 pi (2 j + 1/2) / 8192 for the bins cube2::bin() names; inverse(forward) = 4096 x; a
// filter applied bin by bin gives the NEGACYCLIC convolution (what leaves the window comes back
// with its sign changed).  Built and run by tests/test_fft_host.py (g++).
#include <cmath>
#include <complex>
#include <cstdio>
#include <random>
#include <vector>
#include "../../openseize_amd/csrc/fft4096.h"
using namespace osz::fft;
constexpr int NHI = 11;   // rows 16 .. 26 hold samples, 27 .. 31 are zero (a block of 27 rows)
int main() {
    const long double PI = acosl(-1.0L);
    std::vector<double> t0(256 * 2), t1(16 * 256 * 2), t2(16 * 16 * 2);
    for (int t = 0; t < 256; ++t) {
        long double a = -PI * (long double)t / 8192.0L;
        t0[2 * t] = (double)cosl(a); t0[2 * t + 1] = (double)sinl(a); }
    for (int k0 = 0; k0 < 16; ++k0) for (int t = 0; t < 256; ++t) {
        long double a = -2.0L * PI * (long double)(t * k0) / 4096.0L;
        t1[(k0 * 256 + t) * 2] = (double)cosl(a); t1[(k0 * 256 + t) * 2 + 1] = (double)sinl(a); }
    for (int n0 = 0; n0 < 16; ++n0) for (int k1 = 0; k1 < 16; ++k1) {
        long double a = -2.0L * PI * (long double)(n0 * k1) / 256.0L;
        t2[(n0 * 16 + k1) * 2] = (double)cosl(a); t2[(n0 * 16 + k1) * 2 + 1] = (double)sinl(a); }
    Tables tb{t1.data(), t2.data(), t0.data()};
    std::mt19937_64 rng(11); std::normal_distribution<double> nd;
    const int S = 256 * (16 + NHI);
    std::vector<double> x(8192, 0.0);
    for (int n = 0; n < S; ++n) x[n] = nd(rng);
    std::vector<double> re(256 * 16), im(256 * 16);
    std::vector<cube::C2> L(4096);
    std::vector<nega::TwPowN> w1(256); std::vector<cube::TwPow> w2(256);
    for (int t = 0; t < 256; ++t) { nega::tw_load(t, tb, w1[t], w2[t]);
        for (int j = 0; j < 16; ++j) { re[t*16+j] = x[256*j+t]; im[t*16+j] = x[4096+256*j+t]; } }
    for (int t = 0; t < 256; ++t) nega::f1<NHI>(t, &re[t*16], &im[t*16], w1[t], L.data());
    for (int t = 0; t < 256; ++t) cube2::f2(t, &re[t*16], &im[t*16], w2[t], L.data());
    for (int t = 0; t < 256; ++t) cube2::f3(t, &re[t*16], &im[t*16], L.data());
    double maxerr = 0, maxmag = 0;
    for (int t = 0; t < 256; t += 7) for (int r = 0; r < 16; r += 3) {
        const int j = cube2::bin(t, r);
        long double sr = 0, si = 0;
        for (int n = 0; n < 8192; ++n) {
            const long double a = -2.0L * PI * ((long double)(2 * j) + 0.5L) * (long double)n / 8192.0L;
            sr += x[n] * cosl(a); si += x[n] * sinl(a); }
        maxerr = fmax(maxerr, fmax(fabs(re[t*16+r] - (double)sr), fabs(im[t*16+r] - (double)si)));
        maxmag = fmax(maxmag, fmax(fabs((double)sr), fabs((double)si))); }
    printf("forward err %.3e (mag %.3e)\n", maxerr, maxmag);
    // a short filter h applied at the odd frequencies: bin j times H(2 pi (j + 1/4) / 4096)
    const int wl = 300;
    std::vector<double> h(wl); for (auto &v : h) v = nd(rng) / wl;
    for (int t = 0; t < 256; ++t) for (int r = 0; r < 16; ++r) {
        const int j = cube2::bin(t, r);
        long double hr = 0, hi = 0;
        for (int n = 0; n < wl; ++n) { const long double a = -2.0L * PI * ((long double)j + 0.25L) * (long double)n / 4096.0L;
            hr += h[n] * cosl(a); hi += h[n] * sinl(a); }
        cube::cmul(re[t*16+r], im[t*16+r], (double)hr, (double)hi); }
    for (int t = 0; t < 256; ++t) cube2::i3(t, &re[t*16], &im[t*16], L.data());
    for (int t = 0; t < 256; ++t) cube2::i2(t, &re[t*16], &im[t*16], w2[t], L.data());
    for (int t = 0; t < 256; ++t) nega::i1(t, &re[t*16], &im[t*16], w1[t], L.data());
    // reference: negacyclic convolution (here nothing wraps: S + wl - 1 <= 8192) plus one that does
    double cerr = 0;
    for (int n = 0; n < 8192; ++n) {
        long double acc = 0;
        for (int k = 0; k < wl; ++k) { const int m = n - k; if (m >= 0) acc += (long double)h[k] * x[m]; else acc -= (long double)h[k] * x[m + 8192]; }
        const double got = (n < 4096 ? re[(n & 255) * 16 + (n >> 8)] : im[(n & 255) * 16 + ((n - 4096) >> 8)]) / 4096.0;
        cerr = fmax(cerr, fabs(got - (double)acc)); }
    printf("negacyclic convolution err %.3e\n", cerr);
    const int bad = maxerr > 1e-10 * maxmag || cerr > 1e-12;
    printf(bad ? "FAILED\n" : "OK\n");
    return bad;
}
