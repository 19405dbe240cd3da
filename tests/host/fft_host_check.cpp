// CPU replay of the 256-thread 4096-point FFT (openseize_amd/csrc/fft4096.h):
// runs every phase for t = 0..255 with a plain array standing in for LDS and
// compares with a direct DFT.  Built and run by tests/test_fft_host.py (g++).
#include <cmath>
#include <complex>
#include <cstdio>
#include <random>
#include <vector>

#include "../../openseize_amd/csrc/fft4096.h"

using namespace osz::fft;
using cd = std::complex<double>;

int main() {
    const long double PI = acosl(-1.0L);
    std::vector<double> t1(16 * 256 * 2), t2(16 * 16 * 2);
    for (int k0 = 0; k0 < 16; ++k0)
        for (int t = 0; t < 256; ++t) {
            long double a = -2.0L * PI * (long double)(t * k0) / 4096.0L;
            t1[(k0 * 256 + t) * 2] = (double)cosl(a);
            t1[(k0 * 256 + t) * 2 + 1] = (double)sinl(a);
        }
    for (int n0 = 0; n0 < 16; ++n0)
        for (int k1 = 0; k1 < 16; ++k1) {
            long double a = -2.0L * PI * (long double)(n0 * k1) / 256.0L;
            t2[(n0 * 16 + k1) * 2] = (double)cosl(a);
            t2[(n0 * 16 + k1) * 2 + 1] = (double)sinl(a);
        }
    Tables tb{t1.data(), t2.data()};

    std::mt19937_64 rng(7);
    std::normal_distribution<double> nd;
    std::vector<cd> x(N);
    for (auto &v : x) v = cd(nd(rng), nd(rng));

    // direct DFT on a subset of bins (long double accumulation)
    std::vector<double> re(NT * 16), im(NT * 16), pr(PLANE), pi(PLANE);
    for (int t = 0; t < NT; ++t)
        for (int j = 0; j < 16; ++j) {
            re[t * 16 + j] = x[256 * j + t].real();
            im[t * 16 + j] = x[256 * j + t].imag();
        }
    for (int t = 0; t < NT; ++t) f1(t, &re[t * 16], &im[t * 16], tb, pr.data(), pi.data());
    for (int t = 0; t < NT; ++t) {
        f2_load(t, &re[t * 16], &im[t * 16], pr.data(), pi.data());
        f2_compute(t, &re[t * 16], &im[t * 16], tb);
    }
    for (int t = 0; t < NT; ++t) f2_store(t, &re[t * 16], &im[t * 16], pr.data(), pi.data());
    for (int t = 0; t < NT; ++t) f3(t, &re[t * 16], &im[t * 16], pr.data(), pi.data());

    double maxerr = 0, maxmag = 0;
    for (int k = 0; k < N; k += 7) {
        long double sr = 0, si = 0;
        for (int n = 0; n < N; ++n) {
            long double a = -2.0L * PI * (long double)((long long)n * k % N) / (long double)N;
            long double c = cosl(a), s = sinl(a);
            sr += x[n].real() * c - x[n].imag() * s;
            si += x[n].real() * s + x[n].imag() * c;
        }
        const int t = k % 256, j = k / 256;
        int r = 0;
        for (; r < 16; ++r)
            if (dr(r) == j) break;
        double er = fabs(re[t * 16 + r] - (double)sr), ei = fabs(im[t * 16 + r] - (double)si);
        maxerr = fmax(maxerr, fmax(er, ei));
        maxmag = fmax(maxmag, fmax(fabs((double)sr), fabs((double)si)));
    }
    printf("forward max abs err %.3e (max |X| %.3e)\n", maxerr, maxmag);
    int bad = maxerr > 1e-10 * maxmag;

    // inverse of the forward result must return 4096 * x
    for (int t = 0; t < NT; ++t) i3(t, &re[t * 16], &im[t * 16], pr.data(), pi.data());
    for (int t = 0; t < NT; ++t) {
        i2_load(t, &re[t * 16], &im[t * 16], tb, pr.data(), pi.data());
    }
    for (int t = 0; t < NT; ++t) i2_compute_store(t, &re[t * 16], &im[t * 16], pr.data(), pi.data());
    for (int t = 0; t < NT; ++t) i1(t, &re[t * 16], &im[t * 16], tb, pr.data(), pi.data());
    double ierr = 0;
    for (int t = 0; t < NT; ++t)
        for (int j = 0; j < 16; ++j) {
            ierr = fmax(ierr, fabs(re[t * 16 + j] / N - x[256 * j + t].real()));
            ierr = fmax(ierr, fabs(im[t * 16 + j] / N - x[256 * j + t].imag()));
        }
    printf("round trip max abs err %.3e\n", ierr);
    bad |= ierr > 1e-12;
    // resident-twiddle phase versions must give the same round trip
    std::vector<TwBase> tw(NT);
    for (int t = 0; t < NT; ++t) {
        tw_load_base(t, tb, tw[t]);
        for (int j = 0; j < 16; ++j) {
            re[t * 16 + j] = x[256 * j + t].real();
            im[t * 16 + j] = x[256 * j + t].imag();
        }
    }
    for (int t = 0; t < NT; ++t) f1_w(t, &re[t * 16], &im[t * 16], tw[t], pr.data(), pi.data());
    for (int t = 0; t < NT; ++t) {
        f2_load(t, &re[t * 16], &im[t * 16], pr.data(), pi.data());
        f2_compute_w(&re[t * 16], &im[t * 16], tw[t]);
    }
    for (int t = 0; t < NT; ++t) f2_store(t, &re[t * 16], &im[t * 16], pr.data(), pi.data());
    for (int t = 0; t < NT; ++t) f3(t, &re[t * 16], &im[t * 16], pr.data(), pi.data());
    for (int t = 0; t < NT; ++t) i3(t, &re[t * 16], &im[t * 16], pr.data(), pi.data());
    for (int t = 0; t < NT; ++t) i2_load_w(t, &re[t * 16], &im[t * 16], tw[t], pr.data(), pi.data());
    for (int t = 0; t < NT; ++t) i2_compute_store(t, &re[t * 16], &im[t * 16], pr.data(), pi.data());
    for (int t = 0; t < NT; ++t) i1_w(t, &re[t * 16], &im[t * 16], tw[t], pr.data(), pi.data());
    double werr = 0;
    for (int t = 0; t < NT; ++t)
        for (int j = 0; j < 16; ++j) {
            werr = fmax(werr, fabs(re[t * 16 + j] / N - x[256 * j + t].real()));
            werr = fmax(werr, fabs(im[t * 16 + j] / N - x[256 * j + t].imag()));
        }
    printf("resident-twiddle round trip max abs err %.3e\n", werr);
    bad |= werr > 1e-12;
    // cube layout (interleaved complex, in-place exchanges): same transform
    {
        std::vector<cube::C2> L(cube::SLOTS);
        std::vector<double> fr(NT * 16), fi(NT * 16);
        for (int t = 0; t < NT; ++t)
            for (int j = 0; j < 16; ++j) {
                re[t * 16 + j] = x[256 * j + t].real();
                im[t * 16 + j] = x[256 * j + t].imag();
            }
        // reference spectrum: the plane-layout forward transform checked above
        for (int t = 0; t < NT; ++t) f1(t, &re[t * 16], &im[t * 16], tb, pr.data(), pi.data());
        for (int t = 0; t < NT; ++t) {
            f2_load(t, &re[t * 16], &im[t * 16], pr.data(), pi.data());
            f2_compute(t, &re[t * 16], &im[t * 16], tb);
        }
        for (int t = 0; t < NT; ++t) f2_store(t, &re[t * 16], &im[t * 16], pr.data(), pi.data());
        for (int t = 0; t < NT; ++t) f3(t, &re[t * 16], &im[t * 16], pr.data(), pi.data());
        fr = re;
        fi = im;
        for (int t = 0; t < NT; ++t)
            for (int j = 0; j < 16; ++j) {
                re[t * 16 + j] = x[256 * j + t].real();
                im[t * 16 + j] = x[256 * j + t].imag();
            }
        // every slot must be owned exactly once in each view
        std::vector<int> seen(cube::SLOTS);
        int own_bad = 0;
        for (int view = 0; view < 3; ++view) {
            std::fill(seen.begin(), seen.end(), 0);
            for (int t = 0; t < NT; ++t)
                for (int j = 0; j < 16; ++j) {
                    const int s = view == 0 ? cube::slot_a(t, j)
                                : view == 1 ? cube::base_b(t) + 16 * j : cube::slot_c(t, j);
                    if (s < 0 || s >= cube::SLOTS) { own_bad = 1; continue; }
                    seen[s]++;
                }
            for (int s = 0; s < cube::SLOTS; ++s) own_bad |= seen[s] != 1;
        }
        printf("cube views are bijections: %s\n", own_bad ? "no" : "yes");
        bad |= own_bad;
        for (int t = 0; t < NT; ++t) cube::f1(t, &re[t * 16], &im[t * 16], tb, L.data());
        for (int t = 0; t < NT; ++t) cube::f2(t, &re[t * 16], &im[t * 16], tb, L.data());
        for (int t = 0; t < NT; ++t) cube::f3(t, &re[t * 16], &im[t * 16], L.data());
        double cerr = 0;
        for (size_t i = 0; i < re.size(); ++i)
            cerr = fmax(cerr, fmax(fabs(re[i] - fr[i]), fabs(im[i] - fi[i])));
        printf("cube forward vs plane forward max abs diff %.3e\n", cerr);
        bad |= cerr > 1e-12 * maxmag;
        for (int t = 0; t < NT; ++t) cube::i3(t, &re[t * 16], &im[t * 16], L.data());
        for (int t = 0; t < NT; ++t) cube::i2(t, &re[t * 16], &im[t * 16], tb, L.data());
        for (int t = 0; t < NT; ++t) cube::i1(t, &re[t * 16], &im[t * 16], tb, L.data());
        double rerr = 0;
        for (int t = 0; t < NT; ++t)
            for (int j = 0; j < 16; ++j) {
                rerr = fmax(rerr, fabs(re[t * 16 + j] / N - x[256 * j + t].real()));
                rerr = fmax(rerr, fabs(im[t * 16 + j] / N - x[256 * j + t].imag()));
            }
        printf("cube round trip max abs err %.3e\n", rerr);
        bad |= rerr > 1e-12;
    }
    printf(bad ? "FAIL\n" : "OK\n");
    return bad;
}
