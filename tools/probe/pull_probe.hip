// How fast does ONE CU pull weight fragments, as a function of the access pattern and of how many CUs pull at once?
// (Round 4: sizing of a fused FC1 -> GELU -> FC2 launch for the text rows, where a workgroup would own a hidden slice and
// pull 98-196 KB of weights by itself instead of 12-24 KB per single-wave workgroup today.)
//
//   hipcc -O3 --offload-arch=gfx950 tools/probe/pull_probe.hip -o tools/probe/pull_probe && tools/probe/pull_probe
//
// Every wave requests NF 16-byte-per-lane fragments back to back (all in flight), then folds them (so nothing is dead).
//   pattern 0 "rows":   the MFMA operand pattern of skinny.hip on a row-major [N][K] matrix: lane (r = lane & 15, q = lane >> 4)
//                       reads 16 B at row r, column 8 q + 32 k  -> 16 segments of 64 B per wave instruction, row pitch K * 2 B
//   pattern 1 "packed": fragment-major storage: a wave instruction reads 1 KiB contiguous, a wave's fragments are contiguous
// Reported per configuration: launch-to-launch time (HIP events over many back-to-back launches) and the in-kernel span
// from the first request to the last fragment's arrival (s_memrealtime, median over waves), warm (the same 14 MB every
// launch: Infinity Cache) and cold (40 buffers in rotation: HBM).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int NF, int PATTERN>
__global__ __launch_bounds__(256) void pull_kernel(const unsigned char* __restrict__ w, size_t wg_bytes, int K2, unsigned* out,
                                                   unsigned long long* stamps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const unsigned char* base = w + (size_t)blockIdx.x * wg_bytes + (size_t)wave * (wg_bytes / nw);
    u32x4 f[NF];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int k = 0; k < NF; ++k) {
        const unsigned char* p;
        if (PATTERN == 0) {
            // a 16-row tile holds K2 bytes per row; fragments walk along K (64 B per row per k-step), next tile after K2 / 64 steps
            const int steps = K2 / 64, tile = k / steps, ks = k - tile * steps;
            p = base + ((size_t)tile * 16 + (lane & 15)) * K2 + ks * 64 + (lane >> 4) * 16;
        } else {
            p = base + (size_t)k * 1024 + lane * 16;
        }
        f[k] = *(const u32x4*)p;
    }
    u32x4 acc = f[0];
#pragma unroll
    for (int k = 1; k < NF; ++k) acc ^= f[k];
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[threadIdx.x] = 1;     // never true on random data: keeps the loads
    if (lane == 0) stamps[(size_t)blockIdx.x * nw + wave] = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int NF, int PATTERN>
int run(const char* name, int wgs, int waves, int K2, unsigned char* bufs, size_t buf_bytes, int nbuf, unsigned* out, unsigned long long* stamps) {
    const size_t wg_bytes = (size_t)waves * NF * 1024;
    if (wg_bytes * wgs > buf_bytes) { printf("%s: buffer too small\n", name); return 0; }
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int cold = 0; cold < 2; ++cold) {
        const int iters = 200;
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((pull_kernel<NF, PATTERN>), dim3(wgs), dim3(waves * 64), 0, 0, bufs + (size_t)(cold ? i % nbuf : 0) * buf_bytes, wg_bytes, K2, out, stamps);
        CK(hipEventRecord(a, 0));
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((pull_kernel<NF, PATTERN>), dim3(wgs), dim3(waves * 64), 0, 0, bufs + (size_t)(cold ? i % nbuf : 0) * buf_bytes, wg_bytes, K2, out, stamps);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        std::vector<unsigned long long> st((size_t)wgs * waves);
        CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
        std::sort(st.begin(), st.end());
        const double med = st[st.size() / 2] / 100.0, mx = st.back() / 100.0;      // 100 MHz ticks -> us
        printf("%-7s wgs %3d x %d waves x %2d KiB (%3zu KiB/wg) %s: %6.2f us/launch, in-kernel pull median %5.2f max %5.2f us -> %5.1f GB/s per CU (median), %4.2f TB/s chip\n",
               name, wgs, waves, NF, wg_bytes >> 10, cold ? "cold" : "warm", ms * 1e3 / iters, med, mx, wg_bytes / med * 1e-3,
               (double)wg_bytes * wgs / (ms * 1e-3 / iters) * 1e-12);
    }
    return 0;
}

int main() {
    const size_t buf_bytes = (size_t)20 << 20;
    const int nbuf = 40;
    unsigned char* bufs;
    unsigned* out;
    unsigned long long* stamps;
    CK(hipMalloc(&bufs, buf_bytes * nbuf));
    CK(hipMalloc(&out, 4096));
    CK(hipMalloc(&stamps, 8 * 4096));
    {   // random-ish fill (not all zero: DVFS / compression effects)
        std::vector<unsigned> h(buf_bytes / 4);
        unsigned x = 12345;
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x; }
        for (int i = 0; i < nbuf; ++i) CK(hipMemcpy(bufs + (size_t)i * buf_bytes, h.data(), buf_bytes, hipMemcpyHostToDevice));
    }
    // today's launches: single-wave workgroups of 24 KiB (FC1: 192) and 12 KiB (FC2 split-K: 384)
    run<24, 0>("rows", 192, 1, 1536, bufs, buf_bytes, nbuf, out, stamps);
    run<12, 0>("rows", 384, 1, 6144, bufs, buf_bytes, nbuf, out, stamps);
    run<24, 1>("packed", 192, 1, 0, bufs, buf_bytes, nbuf, out, stamps);
    // fused slices: 4 waves x 48 KiB = 192 KiB per workgroup (48 workgroups), 2 waves x 48 KiB (96), 4 x 24 KiB (96)
    run<48, 0>("rows", 48, 4, 1536, bufs, buf_bytes, nbuf, out, stamps);
    run<48, 1>("packed", 48, 4, 0, bufs, buf_bytes, nbuf, out, stamps);
    run<48, 0>("rows", 96, 2, 1536, bufs, buf_bytes, nbuf, out, stamps);
    run<48, 1>("packed", 96, 2, 0, bufs, buf_bytes, nbuf, out, stamps);
    run<24, 0>("rows", 96, 4, 1536, bufs, buf_bytes, nbuf, out, stamps);
    run<24, 1>("packed", 96, 4, 0, bufs, buf_bytes, nbuf, out, stamps);
    run<24, 1>("packed", 48, 8, 0, bufs, buf_bytes, nbuf, out, stamps);
    run<12, 1>("packed", 48, 16, 0, bufs, buf_bytes, nbuf, out, stamps);
    run<48, 1>("packed", 192, 1, 0, bufs, buf_bytes, nbuf, out, stamps);
    run<48, 0>("rows", 192, 1, 1536, bufs, buf_bytes, nbuf, out, stamps);
    return 0;
}
