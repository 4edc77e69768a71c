// Micro-benchmark: do two workgroups with a large LDS allocation share a CU on gfx950?
// 512 workgroups on 256 CUs, each spinning a fixed number of VALU operations: ~1x the single
// workgroup time if two are co-resident, ~2x if not.
// Build: hipcc -O3 --offload-arch=gfx950 tools/occupancy_probe.hip -o tools/occupancy_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void spin(uint32_t* out, int iters) {
    extern __shared__ uint32_t lds[];
    uint32_t a = threadIdx.x + 1, b = blockIdx.x + 3;
    lds[threadIdx.x] = a;
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
        a = a * 1664525u + b;
        b = (b ^ a) + 0x9E3779B9u;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ lds[(threadIdx.x + 1) % blockDim.x];
}

int main() {
    uint32_t* d;
    hipMalloc(&d, 2048 * 1024 * 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(spin), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int threads : {640, 512, 1024}) {
        for (int ldskb : {16, 32, 48, 64, 66, 68, 72, 80}) {
            float t[2];
            for (int pass = 0; pass < 2; ++pass) {
                const int grid = pass == 0 ? 256 : 512;
                hipEvent_t e0, e1;
                hipEventCreate(&e0); hipEventCreate(&e1);
                hipLaunchKernelGGL(spin, dim3(grid), dim3(threads), ldskb * 1024, 0, d, 1000);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                hipLaunchKernelGGL(spin, dim3(grid), dim3(threads), ldskb * 1024, 0, d, 200000);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&t[pass], e0, e1);
            }
            printf("threads %4d  LDS %3d KB per WG: 256 WGs %.3f ms, 512 WGs %.3f ms  ratio %.2f (%s)\n", threads, ldskb,
                   t[0], t[1], t[1] / t[0], t[1] / t[0] < 1.5 ? "two per CU" : "one per CU");
        }
    }
    return 0;
}
