// zp_stamps.hip -- diagnostic build of the zero-phase chain kernel with in-kernel phase stamps
// (s_memtime): where do a wave's cycles go per pair of blocks?  Not part of the library; the
// stamped build's run time is not representative, its SHARES are.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude benchmarks/zp_stamps.hip -o /tmp/zp_stamps -L/opt/rocm/lib -lrocfft -ldl
#define OSZ_FIR_STAMPS 1
#include "../openseize_amd/csrc/lib.hip"
#include "../openseize_amd/csrc/fir.hip"
#include "../openseize_amd/csrc/sos.hip"
#include "../openseize_amd/csrc/chain.hip"
#include "../openseize_amd/csrc/chain_spec.hip"
#include "../openseize_amd/csrc/chain_zp.hip"

#include <vector>

int main() {
    const int nch = 256, ntaps = 1024;
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
    unsigned long long *st;
    const size_t nst = (size_t)nch * 64 * 4 * 16;
    hipMalloc(&st, nst * 8);
    hipMemset(st, 0, nst * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(osz::g_fir_stamps), &st, sizeof(st));
    if (osz_chain_zp_open(f, s, 0, nullptr)) { printf("%s\n", osz_last_error()); return 1; }
    for (int k = 0; k < 3; ++k)
        if (osz_chain_zp_step(f, s, x, n, n, nullptr, 0, 0, y, n, nullptr)) { printf("%s\n", osz_last_error()); return 1; }
    hipDeviceSynchronize();
    hipMemset(st, 0, nst * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    osz_chain_zp_step(f, s, x, n, n, nullptr, 0, 0, y, n, nullptr);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> hs(nst);
    hipMemcpy(hs.data(), st, nst * 8, hipMemcpyDeviceToHost);
    const char *names[16] = {"stores of the pair before + load issue", "loads land + pass 1", "barrier 1", "pass 2 (+ spectrum request)",
                             "fence 2", "pass 3", "spectrum lands + multiply", "inverse pass 3", "fence 3",
                             "inverse pass 2", "barrier 4", "inverse pass 1", "fit samples to LDS + overlap add",
                             "barrier 5", "fit + amplitudes + barrier 6", "bursts"};
    double tot[16] = {0}, all = 0;
    size_t waves = 0;
    for (size_t wv = 0; wv < nst / 16; ++wv) {
        double sm = 0;
        for (int i = 0; i < 16; ++i) sm += hs[wv * 16 + i];
        if (sm == 0) continue;
        ++waves;
        for (int i = 0; i < 16; ++i) { tot[i] += hs[wv * 16 + i]; all += hs[wv * 16 + i]; }
    }
    const double pairs_per_wave = 95.0;
    printf("zero-phase chain, 256 ch x 2^20 (stamped build): %.3f ms; %zu waves\n", ms, waves);
    printf("mean s_memtime ticks per pair per wave: %.0f\n", all / waves / pairs_per_wave);
    for (int i = 0; i < 16; ++i) printf("  %-42s %5.1f %%  (%.0f ticks per pair)\n", names[i], 100.0 * tot[i] / all, tot[i] / waves / pairs_per_wave);
    return 0;
}
