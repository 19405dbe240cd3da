// fft8_host_check.cpp -- replays the N / 8 threads of fft8.h on the CPU (the
// stage functions are __host__ __device__): forward result against a naive DFT,
// inverse(forward(x)) = N x, the revdigits map, and brute-force LDS bank
// conflict freedom of every stage access for the lane groups of
// MI355X_MICROARCH.md (ds_read_b128: 4 groups of 16 lanes; ds_write_b128: 8
// contiguous lanes, 32 banks).
//   g++ -O2 -std=c++17 -I openseize_amd/csrc tests/host/fft8_host_check.cpp -o fft8_check
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fft8.h"

using namespace osz::fft8;
typedef std::complex<double> cd;

static std::vector<double> make_table() {
    std::vector<double> tab(2 * kTabLen);
    const long double PI = acosl(-1.0L);
    for (int j = 0; j < kTabLen; ++j) {
        const long double ang = -2.0L * PI * j / (long double)kTabN;
        tab[2 * j] = (double)cosl(ang);
        tab[2 * j + 1] = (double)sinl(ang);
    }
    return tab;
}

template <int N, int S>
struct Run {
    static void fwd(std::vector<double> &re, std::vector<double> &im, const std::vector<Twid<N>> &tw,
                    std::vector<C2> &lds) {
        for (int t = 0; t < N / 8; ++t) fwd_stage<N, S>(t, &re[8 * t], &im[8 * t], tw[t].wr[S], tw[t].wi[S], lds.data());
        if constexpr (S + 1 < Plan<N>::NS) Run<N, S + 1>::fwd(re, im, tw, lds);
    }
    static void inv(std::vector<double> &re, std::vector<double> &im, const std::vector<Twid<N>> &tw,
                    std::vector<C2> &lds) {
        for (int t = 0; t < N / 8; ++t) inv_stage<N, S>(t, &re[8 * t], &im[8 * t], tw[t].wr[S], tw[t].wi[S], lds.data());
        if constexpr (S > 0) Run<N, S - 1>::inv(re, im, tw, lds);
    }
};

static const int RD[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                              {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                              {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                              {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};

template <int N, int S>
static int conflicts_from(int worst) {
    constexpr int P = Plan<N>::pos(S);
    constexpr int NT = N / 8;
    for (int r = 0; r < 8; ++r)
        for (int wave = 0; wave < (NT + 63) / 64; ++wave) {
            for (int g = 0; g < 4; ++g) {   // reads: 16 columns of 16 B over 64 banks
                int cnt[16] = {0};
                for (int q = 0; q < 16; ++q) {
                    const int lane = RD[g][q];
                    if (wave * 64 + lane >= NT) continue;
                    cnt[swz(idx_of<P>(wave * 64 + lane, r)) & 15]++;
                }
                for (int c = 0; c < 16; ++c) worst = cnt[c] > worst ? cnt[c] : worst;
            }
            for (int g = 0; g < 8; ++g) {   // writes: 8 columns over 32 banks
                int cnt[8] = {0};
                for (int q = 0; q < 8; ++q) {
                    const int lane = 8 * g + q;
                    if (wave * 64 + lane >= NT) continue;
                    cnt[swz(idx_of<P>(wave * 64 + lane, r)) & 7]++;
                }
                for (int c = 0; c < 8; ++c) worst = cnt[c] > worst ? cnt[c] : worst;
            }
        }
    if constexpr (S + 1 < Plan<N>::NS) return conflicts_from<N, S + 1>(worst);
    return worst;
}

template <int N>
static int check(const std::vector<double> &tab) {
    constexpr int NT = N / 8, L = Plan<N>::L;
    std::vector<cd> x(N);
    srand(N);
    for (auto &v : x) v = cd(rand() / (double)RAND_MAX - 0.5, rand() / (double)RAND_MAX - 0.5);
    std::vector<cd> ref(N);
    const long double PI = acosl(-1.0L);
    for (int k = 0; k < N; ++k) {
        std::complex<long double> s = 0;
        for (int n = 0; n < N; ++n) {
            const long double ang = -2.0L * PI * (long double)(((long long)k * n) % N) / N;
            s += std::complex<long double>(x[n].real(), x[n].imag()) *
                 std::complex<long double>(cosl(ang), sinl(ang));
        }
        ref[k] = cd((double)s.real(), (double)s.imag());
    }
    std::vector<double> re(N), im(N);
    std::vector<Twid<N>> tw(NT);
    for (int t = 0; t < NT; ++t) {
        twid_load<N>(t, tab.data(), tw[t]);
        for (int r = 0; r < 8; ++r) {
            re[8 * t + r] = x[NT * r + t].real();
            im[8 * t + r] = x[NT * r + t].imag();
        }
    }
    std::vector<C2> lds(N);
    Run<N, 0>::fwd(re, im, tw, lds);
    double err = 0, scale = 0;
    for (int t = 0; t < NT; ++t)
        for (int r = 0; r < 8; ++r) {
            const int k = revdigits<L>(idx_of<0>(t, r));
            err = fmax(err, std::abs(cd(re[8 * t + r], im[8 * t + r]) - ref[k]));
            scale = fmax(scale, std::abs(ref[k]));
        }
    Run<N, Plan<N>::NS - 1>::inv(re, im, tw, lds);
    double err2 = 0;
    for (int t = 0; t < NT; ++t)
        for (int r = 0; r < 8; ++r)
            err2 = fmax(err2, std::abs(cd(re[8 * t + r], im[8 * t + r]) / (double)N - x[NT * r + t]));
    // swz and swz_nat are bijections of [0, N)
    std::vector<int> seen(N, 0), seen2(N, 0);
    int bij = 1;
    for (int i = 0; i < N; ++i) {
        const int a = swz(i), b = swz_nat(i);
        if (a < 0 || a >= N || seen[a]++ || b < 0 || b >= N || seen2[b]++) bij = 0;
    }
    const int ways = conflicts_from<N, 0>(1);
    printf("N=%d fwd_err=%.2e inv_err=%.2e bijective=%d worst_bank_ways=%d\n", N, err / scale, err2, bij,
           ways);
    return (err / scale < 1e-14 && err2 < 1e-14 && bij && ways == 1) ? 0 : 1;
}

int main() {
    const std::vector<double> tab = make_table();
    int bad = 0;
    bad += check<512>(tab);
    bad += check<1024>(tab);
    bad += check<2048>(tab);
    bad += check<4096>(tab);
    bad += check<8192>(tab);
    return bad;
}
