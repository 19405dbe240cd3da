// fir_stamps.hip -- diagnostic build of the FIR kernel with in-kernel phase
// stamps (s_memtime): where do a wave's cycles go per pair of blocks?  Not part
// of the library; the stamped build's run time is not representative, its
// SHARES are.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude benchmarks/fir_stamps.hip -o /tmp/fir_stamps
#define OSZ_FIR_STAMPS 1
#include "../openseize_amd/csrc/lib.hip"
#include "../openseize_amd/csrc/fir.hip"

#include <vector>

int main() {
    const int nch = 256, ntaps = 1024;
    const int64_t n = 1 << 20;
    std::vector<double> h(ntaps);
    for (int i = 0; i < ntaps; ++i) h[i] = 0.2 * (i == ntaps / 2 ? 1.0 : sin(0.2 * M_PI * (i - ntaps / 2 + 0.5)) / (0.2 * M_PI * (i - ntaps / 2 + 0.5))) / 5;
    osz_fir_t f;
    if (osz_fir_create(&f, h.data(), ntaps, nch)) { printf("%s\n", osz_last_error()); return 1; }
    double *x, *y;
    hipMalloc(&x, sizeof(double) * nch * n);
    hipMalloc(&y, sizeof(double) * nch * n);
    std::vector<double> hx((size_t)nch * n);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (double)((i * 2654435761u) % 1000) / 500.0 - 1.0;
    hipMemcpy(x, hx.data(), hx.size() * 8, hipMemcpyHostToDevice);
    unsigned long long *st;
    const size_t nst = (size_t)nch * 64 * 4 * 12;   // up to 64 runs per channel
    hipMalloc(&st, nst * 8);
    hipMemset(st, 0, nst * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(osz::g_fir_stamps), &st, sizeof(st));
    osz_fir_push(f, x, n, n, y, n, 0, nullptr);
    hipDeviceSynchronize();
    hipMemset(st, 0, nst * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    osz_fir_push(f, x, n, n, y, n, 0, nullptr);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> hs(nst);
    hipMemcpy(hs.data(), st, nst * 8, hipMemcpyDeviceToHost);
    const char *names[12] = {"load issue (+prev stores)", "loads land + pass 1", "barrier 1", "pass 2 (+H request)",
                             "barrier 2", "pass 3", "H loads + multiply", "inverse pass 3", "barrier 3",
                             "inverse pass 2", "barrier 4", "inverse pass 1 + OA + stores"};
    double tot[12] = {0}, all = 0;
    size_t waves = 0;
    for (size_t wv = 0; wv < nst / 12; ++wv) {
        double s = 0;
        for (int i = 0; i < 12; ++i) s += hs[wv * 12 + i];
        if (s == 0) continue;
        ++waves;
        for (int i = 0; i < 12; ++i) { tot[i] += hs[wv * 12 + i]; all += hs[wv * 12 + i]; }
    }
    unsigned long long clk[2] = {0, 0};
    hipMemcpyFromSymbol(clk, HIP_SYMBOL(osz::g_fir_clock), sizeof(clk));
    printf("one run: %llu s_memtime ticks over %llu s_memrealtime ticks (100 MHz) => s_memtime at %.0f MHz\n",
           clk[0], clk[1], clk[1] ? 100.0 * clk[0] / clk[1] : 0.0);
    const double pairs_per_wave = (double)((n / (3072 * 2)) * nch) * 4 / waves;
    printf("FIR 256 ch x 2^20, 1024 taps (stamped build): %.3f ms; %zu waves, %.1f pairs per wave\n", ms, waves, pairs_per_wave);
    printf("mean s_memtime ticks per pair per wave: %.0f\n", all / waves / pairs_per_wave);
    for (int i = 0; i < 12; ++i) printf("  %-30s %5.1f %%  (%.0f ticks per pair)\n", names[i], 100.0 * tot[i] / all, tot[i] / waves / pairs_per_wave);
    return 0;
}
