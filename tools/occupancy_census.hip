// Census: how many workgroups of a given shape (threads, LDS bytes, VGPRs) does one gfx950 CU hold
// at the same time?  Every workgroup records which CU it ran on (HW_ID + XCC_ID) and the interval it
// was resident (s_memrealtime, 100 MHz), and only SLEEPS in between (s_sleep polling the real-time
// counter), so the answer cannot be confused with ALU throughput the way a spinning probe can.
// Replaces tools/occupancy_probe.hip.  Output: per configuration the largest number of workgroups
// whose intervals overlap on one CU, and waves per SIMD that follows.
// Build: hipcc -O3 --offload-arch=gfx950 tools/occupancy_census.hip -o tools/occupancy_census.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>

struct Rec {
    uint32_t hwid, xcc;
    uint64_t t0, t1;
};

template <int VGPRS>
__global__ __launch_bounds__(1024) void census(Rec* out, uint32_t ticks) {
    extern __shared__ uint32_t lds[];
    // force the kernel's VGPR allocation: touching v[VGPRS-1] makes next_free_vgpr = VGPRS
    if (VGPRS == 64) asm volatile("v_mov_b32 v63, 0" ::: "v63");
    if (VGPRS == 80) asm volatile("v_mov_b32 v79, 0" ::: "v79");
    if (VGPRS == 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
    if (VGPRS == 112) asm volatile("v_mov_b32 v111, 0" ::: "v111");
    if (VGPRS == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    uint64_t t = t0;
    while (t - t0 < ticks) {
        __builtin_amdgcn_s_sleep(64);
        t = __builtin_amdgcn_s_memrealtime();
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x] = {hwid, xcc, t0, t + lds[1] - 1};
    }
}

template <int VGPRS>
void run(Rec* d, int threads, int ldskb) {
    const int grid = 1024;  // four per CU offered; the CU takes what fits
    hipFuncSetAttribute(reinterpret_cast<const void*>(census<VGPRS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(census<VGPRS>, dim3(grid), dim3(threads), ldskb * 1024, 0, d, 20000u);  // 200 us each
    if (hipDeviceSynchronize() != hipSuccess) {
        printf("threads %4d LDS %3d KB VGPRs %3d: launch failed\n", threads, ldskb, VGPRS);
        return;
    }
    std::vector<Rec> r(grid);
    hipMemcpy(r.data(), d, grid * sizeof(Rec), hipMemcpyDeviceToHost);
    std::map<uint64_t, std::vector<std::pair<uint64_t, int>>> ev;  // CU key -> (time, +1/-1)
    for (const Rec& x : r) {
        // HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID: [3:0]
        const uint64_t key = (static_cast<uint64_t>(x.xcc & 0xF) << 16) | ((x.hwid >> 8) & 0xFF);
        ev[key].push_back({x.t0, +1});
        ev[key].push_back({x.t1, -1});
    }
    int best = 0;
    for (auto& kv : ev) {
        std::sort(kv.second.begin(), kv.second.end());
        int cur = 0;
        for (auto& e : kv.second) {
            cur += e.second;
            best = std::max(best, cur);
        }
    }
    printf("threads %4d LDS %3d KB VGPRs %3d: %3zu CUs seen, max %d workgroups resident on one CU = %d waves/CU = %.1f waves/SIMD\n",
           threads, ldskb, VGPRS, ev.size(), best, best * threads / 64, best * threads / 64 / 4.0);
}

int main() {
    Rec* d;
    hipMalloc(&d, 4096 * sizeof(Rec));
    run<64>(d, 1024, 1);     // does a CU hold 32 wavefronts?  (two 1024-thread workgroups)
    run<64>(d, 1024, 67);    // ... with K1's LDS
    run<64>(d, 512, 1);
    run<64>(d, 256, 1);
    run<96>(d, 640, 66);     // the shape round 1 concluded "never co-resident"
    run<96>(d, 640, 1);
    run<80>(d, 768, 67);
    run<80>(d, 768, 1);
    run<112>(d, 1024, 67);   // K1 as shipped: one per CU expected (VGPRs)
    run<112>(d, 512, 67);
    run<128>(d, 512, 1);
    run<64>(d, 1024, 78);    // the dense kernel's 78,912 B
    run<64>(d, 1024, 79);
    run<64>(d, 1024, 80);    // exactly half of the CU's 160 KB
    run<64>(d, 896, 80);
    return 0;
}
