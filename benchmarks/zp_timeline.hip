// zp_timeline.hip -- diagnostic build of the zero-phase chain kernel with absolute time marks per
// workgroup (s_memtime, 10 ns ticks): where does the FIXED cost of a launch go (start ramp, tables,
// the closing pair, exit, the gap to the next launch)?  Not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude benchmarks/zp_timeline.hip -o benchmarks/bin/zp_timeline -L/opt/rocm/lib -lrocfft -ldl
//   benchmarks/bin/zp_timeline [channels]
#define OSZ_ZP_MARKS 1
#include "../openseize_amd/csrc/lib.hip"
#include "../openseize_amd/csrc/fir.hip"
#include "../openseize_amd/csrc/sos.hip"
#include "../openseize_amd/csrc/chain.hip"
#include "../openseize_amd/csrc/chain_spec.hip"
#include "../openseize_amd/csrc/chain_zp.hip"

#include <algorithm>
#include <vector>

int main(int argc, char **argv) {
    const int nch = argc > 1 ? atoi(argv[1]) : 32, ntaps = 1024;
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
    osz_fir_t f;
    osz_sos_t s;
    if (osz_fir_create(&f, h.data(), ntaps, nch) || osz_sos_create(&s, sos, 6, nch)) { printf("%s\n", osz_last_error()); return 1; }
    if (osz_chain_zp_lag(f, s) < 0) { printf("not eligible\n"); return 1; }
    double *x, *y;
    hipMalloc(&x, sizeof(double) * nch * n);
    hipMalloc(&y, sizeof(double) * nch * n);
    std::vector<double> hx((size_t)nch * n);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (double)((i * 2654435761u) % 1000) / 500.0 - 1.0;
    hipMemcpy(x, hx.data(), hx.size() * 8, hipMemcpyHostToDevice);
    const int nruns = std::max(1, 512 / nch);
    const size_t nwg = (size_t)nch * nruns, nm = nwg * 8;
    unsigned long long *mk[2];
    for (int q = 0; q < 2; ++q) { hipMalloc(&mk[q], nm * 8); hipMemset(mk[q], 0, nm * 8); }
    if (osz_chain_zp_open(f, s, 0, nullptr)) { printf("%s\n", osz_last_error()); return 1; }
    for (int k = 0; k < 4; ++k)
        if (osz_chain_zp_step(f, s, x, n, n, nullptr, 0, 0, y, n, nullptr)) { printf("%s\n", osz_last_error()); return 1; }
    hipDeviceSynchronize();
    // two launches back to back, marks of each in its own table (the symbol is set between
    // them by a stream-ordered copy)
    unsigned long long **sym;
    hipGetSymbolAddress((void **)&sym, HIP_SYMBOL(osz::g_zp_marks));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipMemcpyAsync(sym, &mk[0], sizeof(void *), hipMemcpyHostToDevice, nullptr);
    hipEventRecord(e0);
    osz_chain_zp_step(f, s, x, n, n, nullptr, 0, 0, y, n, nullptr);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> m(nm);
    hipMemcpy(m.data(), mk[0], nm * 8, hipMemcpyDeviceToHost);
    double fclk = 0.0;   // GHz, mean over the workgroups
    {
        // shader clock: s_memtime ticks per s_memrealtime tick (100 MHz), per workgroup
        double lo = 1e30, hi = 0, sm = 0; unsigned long long r0 = ~0ull, r1 = 0;
        for (size_t w = 0; w < nwg; ++w) {
            const double f = (double)(m[w * 8 + 4] - m[w * 8]) / (double)(m[w * 8 + 7] - m[w * 8 + 6]) * 0.1;
            lo = std::min(lo, f); hi = std::max(hi, f); sm += f;
            r0 = std::min(r0, m[w * 8 + 6]); r1 = std::max(r1, m[w * 8 + 7]);
        }
        fclk = sm / nwg;
        printf("  shader clock while the kernel ran: mean %.3f GHz (min %.3f, max %.3f); first entry -> last exit by the 100 MHz clock %.1f us\n",
               sm / nwg, lo, hi, (double)(r1 - r0) * 0.01);
    }
    auto us = [&](double ticks) { return ticks / (fclk * 1e3); };   // shader-clock ticks -> us at the measured clock
    printf("zero-phase chain, %d ch x 2^20, %d runs per channel: event time %.1f us\n", nch, nruns, ms * 1e3);
    auto stat = [&](const char *name, auto fn, int only_run) {
        double sm = 0, mn = 1e30, mx = 0; size_t cnt = 0;
        for (int c = 0; c < nch; ++c)
            for (int r = 0; r < nruns; ++r) {
                if (only_run == -2 && (r == 0 || r == nruns - 1)) continue;
                if (only_run >= 0 && r != only_run) continue;
                const double v = fn(&m[((size_t)c * nruns + r) * 8]);
                sm += v; mn = std::min(mn, v); mx = std::max(mx, v); ++cnt;
            }
        if (cnt) printf("  %-46s mean %8.1f  min %8.1f  max %8.1f us  (%zu workgroups)\n", name, us(sm / cnt), us(mn), us(mx), cnt);
    };
    stat("tables + twiddles (entry -> ready)", [&](const unsigned long long *p) { return (double)(p[1] - p[0]); }, -1);
    stat("whole pairs, first run (opens the chunk)", [&](const unsigned long long *p) { return (double)(p[2] - p[1]); }, 0);
    stat("whole pairs, middle runs", [&](const unsigned long long *p) { return (double)(p[2] - p[1]); }, -2);
    stat("whole pairs, last run", [&](const unsigned long long *p) { return (double)(p[2] - p[1]); }, nruns - 1);
    stat("closing pair + carry export (last run)", [&](const unsigned long long *p) { return (double)(p[3] - p[2]); }, nruns - 1);
    stat("seal check + exit", [&](const unsigned long long *p) { return (double)(p[4] - p[3]); }, -1);
    stat("entry -> exit, first run", [&](const unsigned long long *p) { return (double)(p[4] - p[0]); }, 0);
    stat("entry -> exit, middle runs", [&](const unsigned long long *p) { return (double)(p[4] - p[0]); }, -2);
    stat("entry -> exit, last run", [&](const unsigned long long *p) { return (double)(p[4] - p[0]); }, nruns - 1);
    // where the workgroups ran: duration by XCD and by CU (the two workgroups of a CU)
    {
        double xs[16] = {0}; int xn[16] = {0}; double xmin[16], xmax[16];
        for (int q = 0; q < 16; ++q) { xmin[q] = 1e30; xmax[q] = 0; }
        std::vector<std::pair<unsigned long long, double>> cu;
        for (size_t w = 0; w < nwg; ++w) {
            const unsigned long long id = m[w * 8 + 5];
            const unsigned hw = (unsigned)id, xcc = (unsigned)(id >> 32) & 15;
            const double d = (double)(m[w * 8 + 4] - m[w * 8]);
            xs[xcc] += d; ++xn[xcc]; xmin[xcc] = std::min(xmin[xcc], d); xmax[xcc] = std::max(xmax[xcc], d);
            const unsigned cuid = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            cu.push_back({((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cuid, d});
        }
        for (int q = 0; q < 16; ++q)
            if (xn[q]) printf("  XCD %d: %3d workgroups, entry -> exit mean %8.1f  min %8.1f  max %8.1f\n", q, xn[q], us(xs[q] / xn[q]), us(xmin[q]), us(xmax[q]));
        std::sort(cu.begin(), cu.end());
        size_t ncu = 0, i = 0; int hist[8] = {0};
        double dsum = 0;
        while (i < cu.size()) {
            size_t j = i; double lo = 1e30, hi = 0;
            while (j < cu.size() && cu[j].first == cu[i].first) { lo = std::min(lo, cu[j].second); hi = std::max(hi, cu[j].second); ++j; }
            ++ncu; ++hist[std::min<size_t>(j - i, 7)];
            if (j - i >= 2) dsum += (hi - lo) / hi;
            i = j;
        }
        printf("  %zu CUs in use; workgroups per CU histogram:", ncu);
        for (int q = 1; q < 8; ++q) if (hist[q]) printf("  %d x %d", hist[q], q);
        printf("; mean (slowest - fastest) / slowest inside a CU %.3f\n", dsum / std::max<size_t>(1, ncu));
    }
    return 0;
}
