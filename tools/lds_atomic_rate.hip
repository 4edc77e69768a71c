// Micro-benchmark: LDS ds_add_u32 rate per CU on gfx950 for the address patterns the counting
// kernel produces (16 waves per CU, 64 KiB histogram).
// Build: hipcc -O3 --offload-arch=gfx950 tools/lds_atomic_rate.hip -o tools/lds_atomic_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// MODE 0 random bins, all lanes       1 conflict-free (bank = lane & 31)
//      2 random bins, ~47 % of lanes   3 random bins with returning atomic
//      4 random rows, bank = lane & 31 5 all lanes one bin
//      6 random bins, 25 % of lanes    7 random bins within 256-bin window (hot region)
template <int MODE>
__global__ __launch_bounds__(1024) void k(uint32_t* out, int iters) {
    __shared__ uint32_t h[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) h[i] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    uint32_t acc = 0;
    bool live = true;       // modes 8..10: a fixed subset of lanes runs the whole loop (no per-iteration branch)
    if (MODE == 8) live = (((threadIdx.x * 2654435761u) >> 13) & 127) < 60;
    if (MODE == 9) live = lane < 32;
    if (MODE == 10) live = (lane & 3) == 0;
    if (MODE == 11) live = (lane & 1) == 0;
    if (live)
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s = s * 1664525u + 1013904223u;
            const uint32_t r = s >> 10;
            uint32_t a;
            bool on = true;
            if (MODE == 1) a = ((i * 8 + j) * 64 + lane) & 16383;
            else if (MODE == 4) a = (r & 16383 & ~31u) | (lane & 31);
            else if (MODE == 5) a = (i * 8 + j) & 16383;
            else if (MODE == 7) a = r & 255;
            else if (MODE == 12) a = r & 15;
            else if (MODE == 13) a = ((r & 15) << 2) | (lane >> 4);
            else if (MODE == 14) a = ((r & 15) << 1) | (lane >> 5);
            else a = r & 16383;
            if (MODE == 2 || MODE >= 12) on = ((s >> 3) & 127) < 60;
            if (MODE == 6) on = ((s >> 3) & 3) == 0;
            if (on) {
                if (MODE == 3 || MODE >= 12) acc += atomicAdd(&h[a], 1u);
                else atomicAdd(&h[a], 1u);
            }
        }
    }
    __syncthreads();
    uint32_t t = acc;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) t += h[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

template <int MODE>
void run(uint32_t* d, const char* what, double lanes_frac) {
    const int iters = 2048;
    for (int wpb : {256, 1024}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(wpb), 0, 0, d, 8);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(wpb), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double winstr = (double)iters * 8 * (wpb / 64);          // wave instructions per CU
        const double cyc = ms * 1e-3 * 2.4e9;
        printf("%-34s waves/CU %2d  %.3f ms  %.2f cycles per wave ds_add  %.2f G lane-atomics/s/CU\n", what, wpb / 64, ms,
               cyc / winstr, winstr * 64 * lanes_frac / (ms * 1e-3) / 1e9);
    }
}

int main() {
    uint32_t* d; hipMalloc(&d, 256 * 1024 * 4);
    run<0>(d, "random bins, all lanes", 1.0);
    run<1>(d, "conflict-free", 1.0);
    run<4>(d, "random rows, bank=lane&31", 1.0);
    run<2>(d, "random bins, 47% lanes", 0.47);
    run<6>(d, "random bins, 25% lanes", 0.25);
    run<3>(d, "random bins, returning", 1.0);
    run<5>(d, "one bin for all lanes", 1.0);
    run<7>(d, "random within 256 bins", 1.0);
    run<12>(d, "16 counters, returning, 47% lanes", 0.47);
    run<14>(d, "32 counters (2 lane groups), rtn, 47%", 0.47);
    run<13>(d, "64 counters (4 lane groups), rtn, 47%", 0.47);
    run<8>(d, "random bins, fixed 47% of lanes", 0.47);
    run<9>(d, "random bins, lanes 0..31", 0.5);
    run<11>(d, "random bins, even lanes", 0.5);
    run<10>(d, "random bins, every 4th lane", 0.25);
    return 0;
}
