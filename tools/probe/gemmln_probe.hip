// Diagnostic build of the 256x256 GEMM (csrc/gemm256.hip) with s_memrealtime stamps at the phase boundaries of the
// residual + LayerNorm epilogue: where does a fused launch spend its tail?  Not part of the product library.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -DLN_STAMPS \
//            -I real-time-video-captioning_amd/csrc -I include tools/probe/gemmln_probe.hip -o tools/probe/gemmln_probe
#include "gemm256.hip"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <typename T> T* dalloc(size_t n, int fill) {
    T* p; CK(hipMalloc(&p, n * sizeof(T)));
    std::vector<T> h(n);
    unsigned s = 12345u + fill;
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        if (sizeof(T) == 2) h[i] = (T)(0x3c00 + ((s >> 16) & 0x1ff) + ((s >> 31) << 15));
        else { float f = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; h[i] = *(T*)&f; }
    }
    CK(hipMemcpy(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice));
    return p;
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 19200, N = 768;
    for (int K : {768, 3072}) {
        GemmArgs a{};
        a.A = dalloc<bf16_t>((size_t)M * K, 1); a.lda = K; a.W = dalloc<bf16_t>((size_t)N * K, 2); a.bias = dalloc<float>(N, 3);
        a.M = M; a.N = N; a.K = K;
        float* x = dalloc<float>((size_t)M * N, 4);
        a.out = x; a.ldo = N; a.resid = x; a.ldr = N;
        a.ln_g = dalloc<float>(N, 5); a.ln_b = dalloc<float>(N, 6); a.ln_eps = 1e-5f;
        a.ln_out = dalloc<bf16_t>((size_t)M * N, 7); a.ld_ln = N;
        CK(hipMalloc(&a.ln_stats, (size_t)M * 16 * sizeof(float2)));
        CK(hipMalloc(&a.ln_cnt, (size_t)(M / 256 + 1) * 8)); CK(hipMemset(a.ln_cnt, 0, (size_t)(M / 256 + 1) * 8));
        const int grid = (M / 256) * (N / 256);
        unsigned long long* st; CK(hipMalloc(&st, (size_t)grid * 16 * 8));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_ln_stamps), &st, sizeof(st)));
        for (int epi : {(int)EPI_BIAS_RESID_F32, (int)EPI_RESID_LN_PRE, (int)EPI_RESID_LN_POST}) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int i = 0; i < 3; ++i) CK(launch_gemm256(a, epi, 0));
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 10; ++i) CK(launch_gemm256(a, epi, 0));
            CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemset(st, 0, (size_t)grid * 16 * 8));
            CK(launch_gemm256(a, epi, 0)); CK(hipDeviceSynchronize());
            std::vector<unsigned long long> h((size_t)grid * 16);
            CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull, t7 = 0;
            for (int b = 0; b < grid; ++b) { t0 = std::min(t0, h[b * 16]); t7 = std::max(t7, h[b * 16 + 7]); }
            printf("K=%d epi=%d: %.1f us per launch (events); stamped span %.1f us; per workgroup [median (min..max)] in us since the first start:\n",
                   K, epi, ms * 100.f, (t7 - t0) * 0.01);
            const char* names[8] = {"start", "K loop done", "x + segment stats", "published + barrier", "arrived, x stores issued", "siblings arrived", "rows merged", "done"};
            for (int i = 0; i < 8; ++i) {
                if (epi == EPI_BIAS_RESID_F32 && i >= 2 && i <= 6) continue;
                std::vector<double> v;
                for (int b = 0; b < grid; ++b) v.push_back((h[b * 16 + i] - t0) * 0.01);
                std::sort(v.begin(), v.end());
                printf("   %-26s %7.2f (%7.2f .. %7.2f)\n", names[i], v[v.size() / 2], v.front(), v.back());
            }
        }
    }
    return 0;
}
