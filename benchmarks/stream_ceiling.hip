// stream_ceiling.hip -- what a read-once / write-once stream reaches on this
// box, as a function of access width and of how the loads are issued: the
// practical ceiling the FIR / SOS kernels are measured against.
//
//   /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 benchmarks/stream_ceiling.hip \
//       -o gpurun_out/stream_ceiling && gpurun_out/stream_ceiling
//
// Kernels (2 GiB in, 2 GiB out, float64 rows like a 256 x 2^20 chunk):
//   copy16 / copy8      grid-stride copy, 16 or 8 bytes per lane, full occupancy
//   burst<W, ROWS>      the access pattern of fir_oa: a 256-thread workgroup
//                       with 64 KB of LDS (two per CU) loads ROWS rows of 256
//                       lanes x W bytes, waits for all of them, stores them
//   burst_pf<W, ROWS>   same, with the next burst requested before the stores
// Prints one JSON line per kernel: TB/s of (bytes read + bytes written).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));            \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

template <typename V>
__global__ __launch_bounds__(256) void copy_kernel(const V *__restrict__ x, V *__restrict__ y,
                                                   size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        V a = x[i], b = x[i + stride], c = x[i + 2 * stride], d = x[i + 3 * stride];
        y[i] = a;
        y[i + stride] = b;
        y[i + 2 * stride] = c;
        y[i + 3 * stride] = d;
    }
    for (; i < n; i += stride) y[i] = x[i];
}

// one workgroup walks a run of bursts of one row of the (C, n) matrix
template <typename V, int ROWS, bool PF>
__global__ __launch_bounds__(256, 2) void burst_kernel(const V *__restrict__ x, V *__restrict__ y,
                                                       size_t row_elems, int runs_per_row) {
    extern __shared__ char lds_pad[];   // 64 KB: two workgroups per CU, like fir_oa
    const int t = threadIdx.x;
    const size_t per_run = row_elems / runs_per_row;          // elements of V
    const size_t base = (size_t)blockIdx.y * row_elems + (size_t)blockIdx.x * per_run;
    const size_t nburst = per_run / (256 * ROWS);
    const V *p = x + base + t;
    V *q = y + base + t;
    V v[ROWS], w[ROWS];
    if (PF) {
#pragma unroll
        for (int j = 0; j < ROWS; ++j) w[j] = p[256 * j];
    }
    for (size_t b = 0; b < nburst; ++b) {
        if (PF) {
#pragma unroll
            for (int j = 0; j < ROWS; ++j) v[j] = w[j];
            if (b + 1 < nburst) {
#pragma unroll
                for (int j = 0; j < ROWS; ++j) w[j] = p[(b + 1) * 256 * ROWS + 256 * j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < ROWS; ++j) v[j] = p[b * 256 * ROWS + 256 * j];
        }
        if (lds_pad[0] == 77 && t == 1000) v[0] = v[1];   // keep the LDS allocation alive
#pragma unroll
        for (int j = 0; j < ROWS; ++j) q[b * 256 * ROWS + 256 * j] = v[j];
    }
}

template <typename F>
static double time_ms(F launch, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    const int C = 256;
    const size_t n = (size_t)1 << 20;            // doubles per row
    const size_t bytes = (size_t)C * n * 8;      // 2 GiB
    double *x, *y;
    CK(hipMalloc(&x, bytes));
    CK(hipMalloc(&y, bytes));
    CK(hipMemset(x, 1, bytes));
    CK(hipMemset(y, 0, bytes));
    const int reps = 10;
    auto report = [&](const char *name, double ms) {
        printf("{\"kernel\": \"%s\", \"ms\": %.4f, \"TBps\": %.3f}\n", name, ms,
               2.0 * bytes / (ms * 1e-3) / 1e12);
        fflush(stdout);
    };
    for (int grid : {2048, 4096, 8192}) {
        char nm[64];
        snprintf(nm, sizeof nm, "copy16_grid%d", grid);
        report(nm, time_ms([&] {
                   hipLaunchKernelGGL(copy_kernel<double2>, dim3(grid), dim3(256), 0, 0,
                                      (const double2 *)x, (double2 *)y, bytes / 16);
               }, reps));
        snprintf(nm, sizeof nm, "copy8_grid%d", grid);
        report(nm, time_ms([&] {
                   hipLaunchKernelGGL(copy_kernel<double>, dim3(grid), dim3(256), 0, 0,
                                      (const double *)x, y, bytes / 8);
               }, reps));
    }
    report("hipMemcpyDtoD", time_ms([&] { CK(hipMemcpyAsync(y, x, bytes, hipMemcpyDeviceToDevice, 0)); },
                                   reps));
    const size_t lds = 64 * 1024;
#define BURST(V, ROWS, PF, NAME)                                                              \
    {                                                                                         \
        auto k = burst_kernel<V, ROWS, PF>;                                                   \
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k),                             \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));        \
        const size_t row_elems = n * 8 / sizeof(V);                                           \
        for (int runs : {4, 16}) {                                                            \
            char nm[64];                                                                      \
            snprintf(nm, sizeof nm, NAME "_runs%d", runs);                                    \
            report(nm, time_ms([&] {                                                          \
                       hipLaunchKernelGGL(k, dim3(runs, C), dim3(256), lds, 0, (const V *)x,  \
                                          (V *)y, row_elems, runs);                           \
                   }, reps));                                                                 \
        }                                                                                     \
    }
    // 2^20 doubles per row = 4 runs x 32 bursts (or 16 runs x 8 bursts) of 8192 doubles
    BURST(double, 32, false, "burst8B_32rows")
    BURST(double2, 16, false, "burst16B_16rows")
    BURST(double, 32, true, "burst8B_32rows_prefetch")
    BURST(double2, 16, true, "burst16B_16rows_prefetch")
    BURST(double, 16, false, "burst8B_16rows")
    BURST(double2, 8, false, "burst16B_8rows")
    CK(hipFree(x));
    CK(hipFree(y));
    return 0;
}
