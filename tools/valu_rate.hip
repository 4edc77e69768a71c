// Micro-benchmark: integer VALU issue rate per SIMD on gfx950 as a function of waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o gpurun_out/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t* out, int iters) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 ^ 17, a7 = a0 ^ 19;
    const uint32_t m = 0x7F7F7F7Fu ^ blockIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {  // 8 x 8 = 64 independent-ish VALU ops per iteration
            a0 = (a0 & m) + 0x01010101u; a1 = (a1 ^ m) + 3u; a2 = (a2 | 5u) ^ m; a3 = (a3 + m) & 0x3FFFFFFFu;
            a4 = (a4 & m) + 0x01010101u; a5 = (a5 ^ m) + 3u; a6 = (a6 | 5u) ^ m; a7 = (a7 + m) & 0x3FFFFFFFu;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
int main() {
    uint32_t* d; hipMalloc(&d, 256 * 1024 * 8 * 4);
    const int iters = 4096;
    for (int wpb : {64, 128, 256, 512, 1024}) {      // threads per block; 1 block per CU
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(256), dim3(wpb), 0, 0, d, 16);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(wpb), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double waves_per_simd = wpb / 64.0 / 4.0;
        double instr_per_wave = (double)iters * 128;   // 16 ops x 8
        double cyc = ms * 1e-3 * 2.4e9;
        printf("threads/CU %4d waves/SIMD %.2f  %.3f ms  cycles per VALU instr per wave %.2f  per SIMD %.2f\n", wpb,
               waves_per_simd, ms, cyc / instr_per_wave, cyc / (instr_per_wave * (waves_per_simd < 1 ? 1 : waves_per_simd)));
    }
    return 0;
}
