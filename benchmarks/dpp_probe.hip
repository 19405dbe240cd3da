// dpp_probe.hip -- what the DPP controls used by the SOS scan deliver on gfx950
// (row_shr, row_bcast:15, row_bcast:31, wave_shr:1; with an `old` value and a row
// mask, and with bound_ctrl and full masks).
//   hipcc -O3 --offload-arch=gfx950 benchmarks/dpp_probe.hip -o benchmarks/bin/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out) {
    int v = threadIdx.x + 100;
    int q = 0;
    out[64 * q++ + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x142, 0xA, 0xF, false);  // row_bcast15, rows 1,3
    out[64 * q++ + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x143, 0xC, 0xF, false);  // row_bcast31, rows 2,3
    out[64 * q++ + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xF, 0xF, false);  // wave_shr1
    out[64 * q++ + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x111, 0xF, 0xF, false);  // row_shr1
    out[64 * q++ + threadIdx.x] = __builtin_amdgcn_mov_dpp(v, 0x142, 0xF, 0xF, true);          // row_bcast15 bound_ctrl
    out[64 * q++ + threadIdx.x] = __builtin_amdgcn_mov_dpp(v, 0x143, 0xF, 0xF, true);          // row_bcast31 bound_ctrl
    out[64 * q++ + threadIdx.x] = __builtin_amdgcn_mov_dpp(v, 0x138, 0xF, 0xF, true);          // wave_shr1 bound_ctrl
    out[64 * q++ + threadIdx.x] = __builtin_amdgcn_mov_dpp(v, 0x112, 0xF, 0xF, true);          // row_shr2 bound_ctrl
}
int main() {
    int *d, h[512];
    if (hipMalloc(&d, sizeof h) != hipSuccess) return 1;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    const char *names[8] = {"row_bcast15 mask 0xA old -1", "row_bcast31 mask 0xC old -1", "wave_shr1 old -1", "row_shr1 old -1",
                            "row_bcast15 bound_ctrl", "row_bcast31 bound_ctrl", "wave_shr1 bound_ctrl", "row_shr2 bound_ctrl"};
    for (int q = 0; q < 8; ++q) {
        printf("%s:", names[q]);
        for (int i = 0; i < 64; ++i) printf(" %d", h[64 * q + i]);
        printf("\n");
    }
    return 0;
}
