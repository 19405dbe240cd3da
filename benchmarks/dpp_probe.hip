// dpp_probe.hip -- what row_bcast:15 / row_bcast:31 / wave_shr:1 deliver on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out) {
    int v = threadIdx.x + 100;
    out[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x142, 0xA, 0xF, false);        // row_bcast15, rows 1,3
    out[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x143, 0xC, 0xF, false);   // row_bcast31, rows 2,3
    out[128 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xF, 0xF, false);  // wave_shr1
    out[192 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x111, 0xF, 0xF, false);  // row_shr1
}
int main() {
    int *d, h[256];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[4] = {"row_bcast15 mask 0xA", "row_bcast31 mask 0xC", "wave_shr1", "row_shr1"};
    for (int q = 0; q < 4; ++q) {
        printf("%s:", names[q]);
        for (int i = 0; i < 64; ++i) printf(" %d", h[64 * q + i]);
        printf("\n");
    }
    return 0;
}
