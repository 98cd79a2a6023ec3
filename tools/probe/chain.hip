// Prices the two ways to sequence the token loop's dependent phases on this box (diagnostic, not product):
//   (1) a chain of dependent kernel launches on one stream (eager and hipGraph replay);
//   (2) one resident grid with a hand-rolled grid barrier between phases (flat counter, XCD-hierarchical
//       counters; with and without the agent-scope release/acquire fences).
// Build: hipcc -O3 --offload-arch=gfx950 tools/probe/chain.hip -o tools/probe/chain ; run: tools/probe/chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 12345) *p = 1; }

// touches `bytes_per_block` of a buffer: a "real" short streaming kernel
__global__ void stream_kernel(const uint4* in, float* out, int n16_per_block) {
    const uint4* p = in + (size_t)blockIdx.x * n16_per_block;
    unsigned acc = 0;
    for (int i = threadIdx.x; i < n16_per_block; i += blockDim.x) { uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[blockIdx.x] = 1.f;
}

typedef __attribute__((address_space(1))) unsigned gu32;
#define RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// flat monotonic counter barrier; FENCE: release before arrive + acquire after
template <bool FENCE>
__device__ __forceinline__ bool bar_flat(unsigned* cnt, unsigned target, unsigned* tmo) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        if (FENCE) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __hip_atomic_fetch_add((gu32*)cnt, 1u, RLX);
        unsigned spins = 0;
        while (__hip_atomic_load((gu32*)cnt, RLX) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { *tmo = 1; ok = false; break; }
        }
        if (FENCE) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    }
    __syncthreads();
    return ok;
}

// XCD-hierarchical: blocks with equal (blockIdx % 8) arrive on counter g; the last arriver of a group arrives on the
// top counter; the last arriver at the top bumps the generation word everyone polls.
template <bool FENCE>
__device__ __forceinline__ bool bar_xcd(unsigned* st, unsigned phase, unsigned nblk, unsigned* tmo) {
    // st layout (each word on its own 128-B line): [0..7] group counters, [8] top counter, [9] generation
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        if (FENCE) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        const unsigned g = blockIdx.x & 7;
        const unsigned gsize = (nblk >> 3) + ((g < (nblk & 7)) ? 1 : 0);
        const unsigned ngroups = nblk < 8 ? nblk : 8;
        const unsigned old = __hip_atomic_fetch_add((gu32*)(st + g * 32), 1u, RLX);
        if (old + 1 == (phase + 1) * gsize) {
            const unsigned o2 = __hip_atomic_fetch_add((gu32*)(st + 8 * 32), 1u, RLX);
            if (o2 + 1 == (phase + 1) * ngroups) __hip_atomic_store((gu32*)(st + 9 * 32), phase + 1, RLX);
        }
        unsigned spins = 0;
        while (__hip_atomic_load((gu32*)(st + 9 * 32), RLX) < phase + 1) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { *tmo = 1; ok = false; break; }
        }
        if (FENCE) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    }
    __syncthreads();
    return ok;
}

template <int KIND, bool FENCE>
__global__ void barrier_kernel(unsigned* st, int phases, unsigned* tmo, const uint4* in, float* out, int n16_per_block) {
    for (int ph = 0; ph < phases; ++ph) {
        if (n16_per_block) {
            const uint4* p = in + ((size_t)blockIdx.x + (size_t)(ph & 7) * gridDim.x) * n16_per_block;
            unsigned acc = 0;
            for (int i = threadIdx.x; i < n16_per_block; i += blockDim.x) { uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
            if (acc == 0x12345678u) out[blockIdx.x] = 1.f;
        }
        bool ok;
        if (KIND == 0) ok = bar_flat<FENCE>(st, (unsigned)(ph + 1) * gridDim.x, tmo);
        else ok = bar_xcd<FENCE>(st, (unsigned)ph, gridDim.x, tmo);
        if (!ok) return;
    }
}

static double time_ms(hipStream_t s, hipEvent_t a, hipEvent_t b) {
    CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    return ms;
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    int* dp; CK(hipMalloc(&dp, 4));
    const size_t buf_bytes = (size_t)512 << 20;
    uint4* in; CK(hipMalloc(&in, buf_bytes)); CK(hipMemset(in, 1, buf_bytes));
    float* out; CK(hipMalloc(&out, 1 << 20));
    unsigned* st; CK(hipMalloc(&st, 4096)); unsigned* tmo; CK(hipMalloc(&tmo, 4));

    const int N = 2000;
    struct Shape { int g, b; } shapes[] = {{1, 64}, {192, 64}, {256, 256}, {192, 1024}};
    for (auto sh : shapes) {
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(empty_kernel, dim3(sh.g), dim3(sh.b), 0, s, dp);
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(a, s));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_kernel, dim3(sh.g), dim3(sh.b), 0, s, dp);
        CK(hipEventRecord(b, s));
        printf("eager chain  empty<<<%d,%d>>>: %.2f us per launch\n", sh.g, sh.b, time_ms(s, a, b) * 1e3 / N);
    }
    // hipGraph of 200 launches, replayed
    for (auto sh : shapes) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty_kernel, dim3(sh.g), dim3(sh.b), 0, s, dp);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(a, s));
        for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(b, s));
        printf("graph replay empty<<<%d,%d>>>: %.2f us per launch\n", sh.g, sh.b, time_ms(s, a, b) * 1e3 / 2000);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    // streaming kernels: 192 or 256 blocks x 256 threads, each reading KB per block
    for (int kb : {16, 64, 256}) {
        for (int grid : {192, 256}) {
            const int n16 = kb * 1024 / 16;
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(stream_kernel, dim3(grid), dim3(256), 0, s, in + (size_t)(i & 7) * grid * n16, out, n16);
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(a, s));
            for (int i = 0; i < 400; ++i) hipLaunchKernelGGL(stream_kernel, dim3(grid), dim3(256), 0, s, in + (size_t)(i & 7) * grid * n16, out, n16);
            CK(hipEventRecord(b, s));
            const double us = time_ms(s, a, b) * 1e3 / 400;
            printf("eager chain  stream %3d KB x %d blocks: %.2f us per launch (%.2f TB/s)\n", kb, grid, us, (double)kb * 1024 * grid / us * 1e-6);
        }
    }
    // grid barriers
    const int P = 2000;
    auto run_bar = [&](const char* name, auto kern, int grid, int threads, int n16) {
        CK(hipMemsetAsync(st, 0, 4096, s)); CK(hipMemsetAsync(tmo, 0, 4, s));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, s, st, 20, tmo, (const uint4*)in, out, n16);
        CK(hipStreamSynchronize(s));
        CK(hipMemsetAsync(st, 0, 4096, s));
        CK(hipEventRecord(a, s));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, s, st, P, tmo, (const uint4*)in, out, n16);
        CK(hipEventRecord(b, s));
        const double us = time_ms(s, a, b) * 1e3 / P;
        unsigned t = 0; CK(hipMemcpy(&t, tmo, 4, hipMemcpyDeviceToHost));
        printf("%-34s grid %4d x %4d, %3d KB/blk/phase: %.2f us per phase%s\n", name, grid, threads, n16 * 16 / 1024, us, t ? "  TIMEOUT" : "");
    };
    for (int grid : {64, 128, 192, 256}) {
        for (int threads : {256, 1024}) {
            run_bar("barrier flat  +fences", barrier_kernel<0, true>, grid, threads, 0);
            run_bar("barrier flat  no fences", barrier_kernel<0, false>, grid, threads, 0);
            run_bar("barrier xcd   +fences", barrier_kernel<1, true>, grid, threads, 0);
            run_bar("barrier xcd   no fences", barrier_kernel<1, false>, grid, threads, 0);
        }
    }
    for (int kb : {16, 64, 256}) {
        run_bar("barrier xcd   no fences + stream", barrier_kernel<1, false>, 256, 256, kb * 1024 / 16);
        run_bar("barrier xcd   +fences + stream", barrier_kernel<1, true>, 256, 256, kb * 1024 / 16);
        run_bar("barrier flat  no fences + stream", barrier_kernel<0, false>, 192, 1024, kb * 1024 / 16);
    }
    printf("done\n");
    return 0;
}
