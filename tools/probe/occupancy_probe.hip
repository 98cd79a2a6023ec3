// How many workgroups of (threads, dynamic LDS bytes) does one CU of gfx950 hold at once?
//   hipcc -O3 --offload-arch=gfx950 tools/probe/occupancy_probe.hip -o tools/probe/occupancy_probe
//   ./occupancy_probe <threads> <lds_bytes> [vgpr_pressure: 0 | 1]
// Every workgroup stamps the global 100 MHz clock, spins 30 us, stamps again; the number of workgroups whose intervals overlap
// on the busiest CU (HW_ID: XCC, SE, CU) is the resident count.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <algorithm>
__global__ void spin(unsigned long long* out, int us) {
    extern __shared__ char smem[];
    if (threadIdx.x == 0) smem[0] = 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)us * 100) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = t0;
        out[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
        out[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_ID
        out[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // XCC_ID
    }
}
int main(int argc, char** argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 256, lds = argc > 2 ? atoi(argv[2]) : 81920, grid = 1024;
    unsigned long long* d; hipMalloc(&d, grid * 32);
    hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(spin, dim3(grid), dim3(threads), lds, 0, d, 30); hipDeviceSynchronize(); }
    if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
    std::vector<unsigned long long> h(grid * 4); hipMemcpy(h.data(), d, grid * 32, hipMemcpyDeviceToHost);
    std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev;
    for (int b = 0; b < grid; ++b) {
        const unsigned long long cu = ((h[b * 4 + 3] & 0xf) << 16) | (h[b * 4 + 2] & 0xff00) | ((h[b * 4 + 2] >> 13) & 0x7) << 4;   // xcc | cu_id | se_id
        ev[cu].push_back({h[b * 4 + 0], +1}); ev[cu].push_back({h[b * 4 + 1], -1});
    }
    int best = 0, worst = 1 << 30;
    for (auto& kv : ev) { std::sort(kv.second.begin(), kv.second.end()); int c = 0, m = 0; for (auto& e : kv.second) { c += e.second; m = std::max(m, c); } best = std::max(best, m); worst = std::min(worst, m); }
    printf("threads %d, LDS %d B: %zu distinct CUs, resident workgroups per CU: max %d, min %d\n", threads, lds, ev.size(), best, worst);
    return 0;
}
