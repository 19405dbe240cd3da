// sos_stamps.hip -- diagnostic build of the SOS kernel with in-kernel phase
// stamps (s_memtime): where do a wave's cycles go per tile?  Not part of the
// library; the stamped build's run time is not representative, its SHARES are.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 benchmarks/sos_stamps.hip -o /tmp/sos_stamps -L/opt/rocm/lib -lrocfft
#define OSZ_SOS_STAMPS 1
#include "../openseize_amd/csrc/lib.hip"
#include "../openseize_amd/csrc/sos.hip"

#include <vector>

int main() {
    const int nch = 256, nsec = 6;
    const int64_t n = 1 << 20;
    // butter(6, [0.05, 0.3], 'bandpass') from scipy, rounded: timing only
    const double sos[36] = {
        0.00136, 0.00272, 0.00136, 1, -1.0486, 0.4370, 1, 2, 1, 1, -1.1717, 0.5772,
        1, 0, -1, 1, -1.4042, 0.6301, 1, -2, 1, 1, -1.7380, 0.7878,
        1, -2, 1, 1, -1.8214, 0.8484, 1, -2, 1, 1, -1.9169, 0.9406};
    osz_sos_t h;
    if (osz_sos_create(&h, sos, nsec, nch)) { printf("%s\n", osz_last_error()); return 1; }
    double *x, *y;
    hipMalloc(&x, sizeof(double) * nch * n);
    hipMalloc(&y, sizeof(double) * nch * n);
    std::vector<double> hx((size_t)nch * n);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (double)((i * 2654435761u) % 1000) / 500.0 - 1.0;
    hipMemcpy(x, hx.data(), hx.size() * 8, hipMemcpyHostToDevice);
    unsigned long long *st;
    const size_t nst = (size_t)nch * 4 * 8;
    hipMalloc(&st, nst * 8);
    hipMemset(st, 0, nst * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(osz::g_sos_stamps), &st, sizeof(st));
    // (256 channels: the launch is one workgroup per channel and segment as the library cuts it;
    // the OSZ_SOS_WGS knob that forced one workgroup per channel left with the other knobs)
    osz_sos_forward(h, x, n, y, n, n, nullptr);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    osz_sos_forward(h, x, n, y, n, n, nullptr);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> hs(nst);
    hipMemcpy(hs.data(), st, nst * 8, hipMemcpyDeviceToHost);
    const char *names[8] = {"stage_in", "zero_state", "scan", "barrier", "replay+start", "fixup", "stage_out", "-"};
    double tot[8] = {0}, all = 0;
    for (size_t wv = 0; wv < (size_t)nch * 4; ++wv)
        for (int i = 0; i < 8; ++i) { tot[i] += hs[wv * 8 + i]; all += hs[wv * 8 + i]; }
    printf("forward 256 ch x 2^20 (stamped build): %.3f ms; mean cycles per wave %.0f\n", ms, all / (nch * 4));
    for (int i = 0; i < 7; ++i) printf("  %-13s %5.1f %%  (%.0f cycles per tile per wave)\n", names[i], 100.0 * tot[i] / all, tot[i] / (nch * 4) / 128.0);
    return 0;
}
