// ab_chain.cpp -- the headline step (1024-tap FIR -> 6-section band-pass sosfiltfilt, 256 ch x 2^20,
// one osz_chain_zp_step per chunk) timed on several builds of the library in one process, the builds
// taking turns: A/B/C... of kernel variants on ONE box within seconds of each other.
//   g++ -O2 -std=c++17 benchmarks/ab_chain.cpp -o benchmarks/bin/ab_chain -ldl -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -L/opt/rocm/lib -lamdhip64
//   benchmarks/bin/ab_chain [-c channels] [-r rounds] libA.so libB.so ...
// Per build: median and best ms per step over the rounds (each round: 12 untimed + 30 timed steps),
// and a checksum of the last output chunk (variants that must not change results print the same).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef void *h_t;
struct Lib {
    const char *path;
    void *dl;
    int (*fir_create)(h_t *, const double *, int, int);
    int (*sos_create)(h_t *, const double *, int, int);
    int64_t (*zp_lag)(h_t, h_t);
    int (*zp_open)(h_t, h_t, int64_t, void *);
    int (*zp_step)(h_t, h_t, const double *, int64_t, int64_t, double *, int64_t, int64_t, double *, int64_t, void *);
    int (*sos_set_state)(h_t, const double *);
    const char *(*last_error)();
    int (*dbg_clk)(unsigned long long *) = nullptr;   // diagnostic builds: shader / 100 MHz clock ticks of one workgroup's life
    double ghz = 0.0;
    h_t fir = nullptr, sos = nullptr;
    int64_t lag = 0;
    std::vector<float> ms;
    double sum = 0.0;
};

template <class F>
static void sym(void *dl, const char *name, F &f) {
    f = reinterpret_cast<F>(dlsym(dl, name));
    if (!f) {
        fprintf(stderr, "missing %s\n", name);
        exit(2);
    }
}

int main(int argc, char **argv) {
    int nch = 256, rounds = 5, a = 1;
    for (; a + 1 < argc && argv[a][0] == '-'; a += 2) {
        if (!strcmp(argv[a], "-c")) nch = atoi(argv[a + 1]);
        else if (!strcmp(argv[a], "-r")) rounds = atoi(argv[a + 1]);
    }
    const int ntaps = 1024;
    const int64_t n = 1 << 20;
    std::vector<double> h(ntaps);
    for (int i = 0; i < ntaps; ++i) {
        const double u = 0.2 * M_PI * (i - ntaps / 2 + 0.5);
        h[i] = 0.2 * sin(u) / u * (0.54 - 0.46 * cos(2 * M_PI * i / (ntaps - 1)));
    }
    // butter(6, [0.05, 0.3], 'bandpass', output='sos')
    const double sos[36] = {
        0.0010516467963076106, 0.0021032935926152212, 0.0010516467963076106, 1.0, -0.9934971416327785, 0.2812393218014878,
        1.0, 2.0, 1.0, 1.0, -0.9221078391223956, 0.40562659992935945,
        1.0, 2.0, 1.0, 1.0, -1.6277782762853907, 0.6696971094852364,
        1.0, -2.0, 1.0, 1.0, -1.0322360808114461, 0.733692832051055,
        1.0, -2.0, 1.0, 1.0, -1.7937843270963478, 0.821773851594998,
        1.0, -2.0, 1.0, 1.0, -1.916801367713827, 0.9412643725997867};
    std::vector<Lib> libs;
    for (; a < argc; ++a) {
        Lib L{};
        L.path = argv[a];
        L.dl = dlopen(argv[a], RTLD_NOW | RTLD_LOCAL);
        if (!L.dl) {
            fprintf(stderr, "%s\n", dlerror());
            return 2;
        }
        sym(L.dl, "osz_fir_create", L.fir_create);
        sym(L.dl, "osz_sos_create", L.sos_create);
        sym(L.dl, "osz_chain_zp_lag", L.zp_lag);
        sym(L.dl, "osz_chain_zp_open", L.zp_open);
        sym(L.dl, "osz_chain_zp_step", L.zp_step);
        sym(L.dl, "osz_last_error", L.last_error);
        L.dbg_clk = reinterpret_cast<int (*)(unsigned long long *)>(dlsym(L.dl, "osz_dbg_clk"));
        libs.push_back(L);
    }
    double *x[3], *y[2];
    std::vector<double> hx((size_t)nch * n);
    for (int k = 0; k < 3; ++k) {
        unsigned long long s = 88172645463325252ULL + k;
        for (size_t i = 0; i < hx.size(); ++i) {      // xorshift noise in [-1, 1)
            s ^= s << 13;
            s ^= s >> 7;
            s ^= s << 17;
            hx[i] = (double)(int64_t)(s >> 11) / 4503599627370496.0 - 1.0;
        }
        hipMalloc(&x[k], sizeof(double) * nch * n);
        hipMemcpy(x[k], hx.data(), hx.size() * 8, hipMemcpyHostToDevice);
    }
    for (int k = 0; k < 2; ++k) hipMalloc(&y[k], sizeof(double) * nch * n);
    for (auto &L : libs) {
        if (L.fir_create(&L.fir, h.data(), ntaps, nch) || L.sos_create(&L.sos, sos, 6, nch)) {
            fprintf(stderr, "%s: %s\n", L.path, L.last_error());
            return 1;
        }
        L.lag = L.zp_lag(L.fir, L.sos);
        if (L.lag < 0) {
            fprintf(stderr, "%s: not eligible\n", L.path);
            return 1;
        }
        if (L.zp_open(L.fir, L.sos, 0, nullptr)) {
            fprintf(stderr, "%s: %s\n", L.path, L.last_error());
            return 1;
        }
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    int k = 0;
    for (int r = 0; r < rounds; ++r)
        for (auto &L : libs) {
            auto step = [&]() {
                if (L.zp_step(L.fir, L.sos, x[k % 3], n, n, y[(k + 1) & 1] + (n - L.lag), n, L.lag, y[k & 1], n, nullptr)) {
                    fprintf(stderr, "%s: %s\n", L.path, L.last_error());
                    exit(1);
                }
                ++k;
            };
            for (int i = 0; i < 12; ++i) step();
            hipEventRecord(e0);
            for (int i = 0; i < 30; ++i) step();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            L.ms.push_back(ms / 30);
            if (L.dbg_clk) {
                unsigned long long c[4] = {0, 0, 0, 0};
                if (L.dbg_clk(c) == 0 && c[1]) L.ghz = (double)c[0] / (double)c[1] * 0.1;
            }
            if (r == rounds - 1) {
                std::vector<double> out((size_t)n);
                hipMemcpy(out.data(), y[(k - 1) & 1] + 5 * n, n * 8, hipMemcpyDeviceToHost);   // channel 5 of the last chunk
                double s = 0.0;
                for (int64_t i = 0; i < n - L.lag; ++i) s += out[i] * (1.0 + (i % 7));
                L.sum = s;
            }
        }
    for (auto &L : libs) {
        std::vector<float> m = L.ms;
        std::sort(m.begin(), m.end());
        printf("%-40s lag %4lld  median %.4f  best %.4f ms  (", L.path, (long long)L.lag, m[m.size() / 2], m[0]);
        for (float v : L.ms) printf(" %.4f", v);
        printf(" )  sum %.12e", L.sum);
        if (L.ghz > 0.0) printf("  shader clock %.3f GHz", L.ghz);
        printf("\n");
    }
    return 0;
}
