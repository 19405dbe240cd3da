// zpn_stamps.hip -- diagnostic build of the one-block zero-phase chain kernel (chain_zpn_body.h)
// with in-kernel phase stamps (s_memtime): where do a wave's cycles go per block?  Not part of
// the library; the stamped build's run time is not representative, its SHARES are.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude benchmarks/zpn_stamps.hip -o benchmarks/bin/zpn_stamps -L/opt/rocm/lib -lrocfft -ldl -pthread
#define OSZ_NEGA_STAMPS 1
#include "../openseize_amd/csrc/lib.hip"
#include "../openseize_amd/csrc/fir.hip"
#include "../openseize_amd/csrc/sos.hip"
#include "../openseize_amd/csrc/chain.hip"
#include "../openseize_amd/csrc/chain_spec.hip"
#include "../openseize_amd/csrc/chain_zp.hip"
#define OSZ_ZPN_NM 6
#define OSZ_ZPN_NO_DISPATCH          // the headline's instance alone (seconds instead of minutes to compile)
#include "../openseize_amd/csrc/chain_zpn_body.h"
namespace osz {
zp_kern_t zpn_kernel_for(int nb, int nm, int ns, int r) {
    return (nm == 6 && nb == 27 && ns == 2 && r <= 5) ? chain_zpn_kernel<27, 6, 2> : nullptr;
}
zp_kern_t zpn_fwd_kernel_for(int, int, int) { return nullptr; }
}

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
    const size_t nst = (size_t)nch * 64 * 4 * 24;
    hipMalloc(&st, nst * 8);
    hipMemset(st, 0, nst * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(osz::g_nega_stamps), &st, sizeof(st));
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
    const char *names[17] = {"wait for the block's samples + read from LDS", "pack + pass 1 + stores", "barrier 1",
                             "pass 2 + spectrum requests", "pass 3", "spectrum lands + multiply", "inverse pass 3",
                             "inverse pass 2", "barrier 4", "inverse pass 1 loads + next block's requests",
                             "inverse pass 1 + unpack", "fit samples to LDS + overlap add", "barrier 5",
                             "fit + barrier 6", "bursts", "", "stores of the block (+ loop)"};
    double tot[24] = {0}, all = 0, blocks = 0;
    size_t waves = 0;
    for (size_t wv = 0; wv < nst / 24; ++wv) {
        double sm = 0;
        for (int i = 0; i < 23; ++i) sm += hs[wv * 24 + i];
        if (sm == 0) continue;
        ++waves;
        blocks += hs[wv * 24 + 23];
        for (int i = 0; i < 23; ++i) { tot[i] += hs[wv * 24 + i]; all += hs[wv * 24 + i]; }
    }
    printf("one-block zero-phase chain, 256 ch x 2^20 (stamped build): %.3f ms; %zu waves, %.1f blocks per wave\n", ms, waves, blocks / waves);
    printf("mean s_memtime ticks per block per wave: %.0f\n", all / blocks);
    for (int i = 0; i < 17; ++i)
        if (names[i][0]) printf("  %-48s %5.1f %%  (%.0f ticks per block)\n", names[i], 100.0 * tot[i] / all, tot[i] / blocks);
    return 0;
}
