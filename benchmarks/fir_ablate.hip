// fir_ablate.hip -- ablation study of the FIRST overlap-add FIR kernel (plane LDS
// layout, benchmarks/fft4096_planes.h); its findings led to the cube layout (not part of
// the library).  Variants of the pair loop with one cost centre removed each;
// outputs are wrong by construction, only the timings matter.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 benchmarks/fir_ablate.hip -o /tmp/fir_ablate && /tmp/fir_ablate
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#include "fft4096_planes.h"

using namespace osz;

struct Args {
    const double *x;
    double *y;
    int64_t ldx, ldy, n;
    int wlen, step, R, nruns;
    int64_t nblocks;
    const double *H;
    fft::Tables tb;
};

#define KEEP(v) asm volatile("" ::"v"(v))

enum { FULL = 0, NO_GLOBAL = 1, NO_H = 2, NO_TW = 4, NO_OA = 8, NO_LDS = 16, NO_BARRIER = 32, COPY = 64 };

__device__ unsigned long long g_cycles[4];   // [0] sum of per-wave cycles, [1] waves

template <int V>
__global__ __launch_bounds__(256) void fir_kernel(Args a) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    extern __shared__ double lds[];
    double *pr = lds, *pi = lds + fft::PLANE, *carry = lds + 2 * fft::PLANE;
    const int t = threadIdx.x, run = blockIdx.x, c = blockIdx.y, wm1 = a.wlen - 1;
    const double *xr = a.x + (int64_t)c * a.ldx;
    double *yr = a.y + (int64_t)c * a.ldy;
    const int64_t npairs = (a.nblocks + 1) / 2;
    const int64_t blk0 = 2 * (((int64_t)run * npairs) / a.nruns);
    int64_t blk1 = 2 * (((int64_t)(run + 1) * npairs) / a.nruns);
    if (blk1 > a.nblocks) blk1 = a.nblocks;
    for (int i = t; i < wm1; i += 256) carry[i] = 0.0;
    __syncthreads();
    fft::Tables tb = a.tb;
    double re[16], im[16];
    for (int64_t blk = blk0; blk < blk1; blk += 2) {
        const int64_t start_a = blk * a.step;
        const int len_a = a.step, len_b = (blk + 1 < blk1) ? a.step : 0;
        const int64_t start_b = start_a + len_a;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int p = 256 * j + t;
            if (V & NO_GLOBAL) {
                re[j] = p * 1e-3;
                im[j] = p * 2e-3;
            } else {
                re[j] = p < len_a ? xr[start_a + p] : 0.0;
                im[j] = p < len_b ? xr[start_b + p] : 0.0;
            }
        }
        if (!(V & COPY)) {
#define BAR() do { if (!(V & NO_BARRIER) && !(V & NO_LDS)) __syncthreads(); } while (0)
            if (V & NO_LDS) {
                fft::fwd16(re, im); fft::fwd16(re, im); fft::fwd16(re, im);
            } else {
                if (V & NO_TW) {
                    fft::fwd16(re, im);
#pragma unroll
                    for (int r = 0; r < 16; ++r) { pr[fft::dr(r) * fft::S1 + t] = re[r]; pi[fft::dr(r) * fft::S1 + t] = im[r]; }
                } else {
                    fft::f1<true>(t, re, im, tb, pr, pi);
                }
                BAR();
                fft::f2_load(t, re, im, pr, pi);
                if (V & NO_TW) fft::fwd16(re, im); else fft::f2_compute(t, re, im, tb);
                BAR();
                fft::f2_store(t, re, im, pr, pi);
                BAR();
                fft::f3(t, re, im, pr, pi);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = t + 256 * fft::dr(r);
                double hr = 0.5, hi = 0.25;
                if (!(V & NO_H)) { hr = a.H[2 * k]; hi = a.H[2 * k + 1]; }
                const double u = re[r], v = im[r];
                re[r] = u * hr - v * hi;
                im[r] = u * hi + v * hr;
            }
            if (V & NO_LDS) {
                fft::inv16(re, im); fft::inv16(re, im); fft::inv16(re, im);
            } else {
                fft::i3(t, re, im, pr, pi);
                BAR();
                if (V & NO_TW) {
                    const int k0 = t & 15, n0 = t >> 4;
#pragma unroll
                    for (int r = 0; r < 16; ++r) { re[r] = pr[n0 * fft::S2 + fft::dr(r) * 16 + k0]; im[r] = pi[n0 * fft::S2 + fft::dr(r) * 16 + k0]; }
                } else {
                    fft::i2_load(t, re, im, tb, pr, pi);
                }
                BAR();
                fft::i2_compute_store(t, re, im, pr, pi);
                BAR();
                if (V & NO_TW) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { re[r] = pr[fft::dr(r) * fft::S1 + t]; im[r] = pi[fft::dr(r) * fft::S1 + t]; }
                    fft::inv16(re, im);
                } else {
                    fft::i1<true>(t, re, im, tb, pr, pi);
                }
                BAR();
            }
            if (!(V & NO_OA) && !(V & NO_LDS)) {
                double *xb = pr;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int p = 256 * j + t;
                    if (p < wm1) re[j] += carry[p];
                    const int q = p - len_a;
                    if (q >= 0 && q < wm1) xb[q] = re[j];
                }
                BAR();
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int p = 256 * j + t;
                    if (p < wm1) im[j] += xb[p];
                    const int q = p - len_b;
                    if (q >= 0 && q < wm1) carry[q] = im[j];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int p = 256 * j + t;
            if (V & NO_GLOBAL) {
                KEEP(re[j]);
                KEEP(im[j]);
            } else {
                if (p < len_a) yr[start_a + p] = re[j];
                if (p < len_b) yr[start_b + p] = im[j];
            }
        }
        if (!(V & NO_LDS)) BAR();
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&g_cycles[0], __builtin_amdgcn_s_memtime() - c0);
        atomicAdd(&g_cycles[1], 1ull);
    }
}

static size_t g_extra_lds = 0;   // pad the allocation to force 1 workgroup per CU

template <int V>
float run(Args a, int nch, const char *name) {
    const size_t lds = sizeof(double) * (2 * fft::PLANE + 1024) + g_extra_lds;
    hipFuncSetAttribute(reinterpret_cast<const void *>(fir_kernel<V>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(fir_kernel<V>, dim3(a.nruns, nch), dim3(256), lds, 0, a);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(fir_kernel<V>, dim3(a.nruns, nch), dim3(256), lds, 0, a);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    unsigned long long hc[4] = {0, 0, 0, 0}, zero[4] = {0, 0, 0, 0};
    hipMemcpyFromSymbol(hc, HIP_SYMBOL(g_cycles), sizeof(hc));
    hipMemcpyToSymbol(HIP_SYMBOL(g_cycles), zero, sizeof(zero));
    const double cyc_per_wave = hc[1] ? (double)hc[0] / hc[1] : 0.0;
    printf("%-12s %.3f ms  mean wave lifetime %.0f cycles (%.2f us at 2.1 GHz)  %s\n", name, ms,
           cyc_per_wave, cyc_per_wave / 2100.0, hipGetErrorString(hipGetLastError()));
    return ms;
}

int main() {
    const int nch = 256, wlen = 1024;
    const int64_t n = 1 << 20;
    Args a{};
    a.wlen = wlen;
    a.step = fft::N - wlen + 1;
    a.nblocks = n / a.step;  // whole blocks only
    a.n = a.nblocks * a.step;
    a.R = 32;
    a.nruns = (int)((a.nblocks + a.R - 1) / a.R);
    a.ldx = a.ldy = n;
    double *x, *y, *H, *t1, *t2;
    hipMalloc(&x, sizeof(double) * nch * n);
    hipMalloc(&y, sizeof(double) * nch * n);
    hipMalloc(&H, sizeof(double) * 2 * fft::N);
    hipMalloc(&t1, sizeof(double) * 16 * 256 * 2);
    hipMalloc(&t2, sizeof(double) * 16 * 16 * 2);
    std::vector<double> hx((size_t)nch * n, 0.5), hh(2 * fft::N, 0.01), h1(16 * 256 * 2), h2(16 * 16 * 2);
    for (int k0 = 0; k0 < 16; ++k0)
        for (int t = 0; t < 256; ++t) {
            const double ang = -2.0 * M_PI * (t * k0) / 4096.0;
            h1[(k0 * 256 + t) * 2] = cos(ang);
            h1[(k0 * 256 + t) * 2 + 1] = sin(ang);
        }
    for (int n0 = 0; n0 < 16; ++n0)
        for (int k1 = 0; k1 < 16; ++k1) {
            const double ang = -2.0 * M_PI * (n0 * k1) / 256.0;
            h2[(n0 * 16 + k1) * 2] = cos(ang);
            h2[(n0 * 16 + k1) * 2 + 1] = sin(ang);
        }
    hipMemcpy(x, hx.data(), hx.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(H, hh.data(), hh.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(t1, h1.data(), h1.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(t2, h2.data(), h2.size() * 8, hipMemcpyHostToDevice);
    a.x = x;
    a.y = y;
    a.H = H;
    a.tb.t1 = t1;
    a.tb.t2 = t2;
    printf("blocks %lld runs %d\n", (long long)a.nblocks, a.nruns);
    run<FULL>(a, nch, "full");
    run<NO_GLOBAL>(a, nch, "no_global");
    run<NO_H>(a, nch, "no_H");
    run<NO_TW>(a, nch, "no_twiddle");
    run<NO_OA>(a, nch, "no_oa");
    run<NO_BARRIER>(a, nch, "no_barrier");
    run<NO_LDS>(a, nch, "no_lds");
    run<COPY>(a, nch, "copy_only");
    printf("-- compute only (no global loads/stores), two workgroups per CU\n");
    run<NO_GLOBAL | NO_H>(a, nch, "c-no_H");
    run<NO_GLOBAL | NO_TW>(a, nch, "c-no_tw");
    run<NO_GLOBAL | NO_H | NO_TW>(a, nch, "c-no_H_tw");
    run<NO_GLOBAL | NO_OA>(a, nch, "c-no_oa");
    run<NO_GLOBAL | NO_BARRIER>(a, nch, "c-no_bar");
    run<NO_GLOBAL | NO_LDS>(a, nch, "c-no_lds");
    run<NO_GLOBAL | NO_LDS | NO_H | NO_TW>(a, nch, "c-flops");
    g_extra_lds = 20 * 1024;   // 98 KB per workgroup: only one fits per CU
    printf("-- one workgroup per CU\n");
    run<FULL>(a, nch, "full");
    run<NO_GLOBAL>(a, nch, "no_global");
    run<NO_GLOBAL | NO_LDS | NO_H | NO_TW>(a, nch, "c-flops");
    run<COPY>(a, nch, "copy_only");
    return 0;
}
