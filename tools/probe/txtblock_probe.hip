// Diagnostic build of the text-row attention sub-layer (csrc/txtblock.hip) with s_memrealtime stamps at its
// phase boundaries: prints where a launch spends its time.  Not part of the product library.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -DTXT_STAMPS \
//            -I real-time-video-captioning_amd/csrc tools/probe/txtblock_probe.hip -o tools/probe/txtblock_probe
#include "txtblock.hip"
#ifdef PROBE_PACKED_KV
#include "rowops.hip"
#endif
#include <cmath>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

int device_cus() { return 256; }     // gitcap.hip defines it in the product library
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// read-only sweep of a big buffer: evicts weights / K/V from L2 and the Infinity Cache WITHOUT leaving dirty lines behind
__global__ void sweep_kernel(const uint4* p, size_t n16, unsigned* out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345u) *out = acc;
}

template <typename T> T* dalloc(size_t n, int fill) {
    T* p; CK(hipMalloc(&p, n * sizeof(T)));
    std::vector<T> h(n);
    unsigned s = 12345u + fill;
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        if (sizeof(T) == 2) h[i] = (T)(((fill == 8 || fill == 9) && getenv("PROBE_BIG") ? 0x3f00 : 0x3c00) + ((s >> 16) & 0x1ff) + ((s >> 31) << 15));   // bf16 of magnitude ~0.01 .. 0.03 (PROBE_BIG: q, k, v of 0.5 .. 2: a peaked softmax)
        else if (fill < 0) h[i] = (T)0;
        else { float f = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; h[i] = *(T*)&f; }
    }
    CK(hipMemcpy(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice));
    return p;
}

int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 16, S = argc > 2 ? atoi(argv[2]) : 1182;
    const int T = getenv("PROBE_T") ? atoi(getenv("PROBE_T")) : 1, t0 = getenv("PROBE_T0") ? atoi(getenv("PROBE_T0")) : 10;
    const int D = 768, H = 12, Tmax = 32, M = rows * T;
    TxtBlockArgs a{};
    a.eps = 1e-5f;
    a.kv_img = dalloc<bf16_t>((size_t)rows * S * 3 * D, 8);
    a.kv_txt = dalloc<bf16_t>((size_t)rows * Tmax * 3 * D, 9);
#ifdef PROBE_PACKED_KV     // the experimental form of tools/experiments/txtblock_mfma8.hip.txt (fragment-major copy of the image K/V)
    bf16_t* kvp = dalloc<bf16_t>((size_t)rows * H * ((S + 31) / 32) * 4096, -1);
    a.kvp_img = kvp;
    CK(launch_pack_kv(a.kv_img, kvp, rows, S, H, D, 0));
    CK(hipDeviceSynchronize());
#endif
    a.rows = rows; a.beams = 1; a.t0 = t0; a.T = T; a.Tmax = Tmax; a.S_img = S; a.H = H; a.D = D;
    a.aow = dalloc<bf16_t>((size_t)D * D, 10); a.aob = dalloc<float>(D, 11); a.g1 = dalloc<float>(D, 12); a.b1 = dalloc<float>(D, 13);
    a.xin = dalloc<float>((size_t)M * D, 3);
    a.part = dalloc<float>((size_t)M * H * D, -1); a.cnt = dalloc<unsigned>(M + 1000064, -1);
    a.xs = dalloc<float>((size_t)M * D, -1); a.xsb = dalloc<bf16_t>((size_t)M * D, 14);
    const int nblk = 8 * (M + (M + 1) / 2);
    unsigned long long* st; CK(hipMalloc(&st, (size_t)nblk * 16 * 8)); CK(hipMemset(st, 0, (size_t)nblk * 16 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_txt_stamps), &st, sizeof(st)));
    // a second buffer streamed between launches so that weights / K/V are not cache resident (as in the real loop)
    const size_t sweep_mb = argc > 4 ? (size_t)atoi(argv[4]) : 512;          // 512: evicts L2, Infinity Cache and TLBs; 48: L2 only
    char* flush; CK(hipMalloc(&flush, (size_t)512 << 20));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9, sum = 0;
    const int iters = 10;
    for (int it = 0; it < iters; ++it) {
        CK(hipMemsetAsync(st, 0, (size_t)nblk * 16 * 8, s));
        if (!(argc > 3 && atoi(argv[3]))) hipLaunchKernelGGL(sweep_kernel, dim3(2048), dim3(256), 0, s, (const uint4*)flush, (sweep_mb << 20) / 16, (unsigned*)a.cnt + 1000000);
        CK(hipEventRecord(e0, s));
        CK(launch_txt_block(a, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms); sum += ms;
    }
    printf("rows %d S_img %d: launch (event bracket) best %.1f us, mean %.1f us\n", rows, S, best * 1e3, sum / iters * 1e3);
    std::vector<unsigned long long> h((size_t)nblk * 16);
    CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int b = 0; b < nblk; ++b) if (h[b * 16]) { tmin = std::min(tmin, h[b * 16]); for (int i = 0; i < 7; ++i) tmax = std::max(tmax, h[b * 16 + i]); }
    printf("first block start -> last stamp: %.2f us (s_memrealtime, 100 MHz)\n", (tmax - tmin) / 100.0);
    const int seg[10][2] = {{0, 3}, {0, 11}, {11, 12}, {12, 13}, {13, 14}, {14, 15}, {15, 3}, {3, 4}, {4, 5}, {5, 6}};
    const char* names[10] = {"attention", " wave 0: first group reduced", " wave 0: other groups reduced", " wave 0: lane merge",
                             " barrier (slowest wave)", " merge of 16 states", " barrier + ctx", "out-proj + store drain", "ticket", "reducer (last unit only)"};
    std::vector<double> start;
    for (int b = 0; b < nblk; ++b) if (h[b * 16]) start.push_back((h[b * 16] - tmin) / 100.0);
    std::sort(start.begin(), start.end());
    printf("block start offsets: median %.2f us, max %.2f us (%zu active blocks)\n", start[start.size() / 2], start.back(), start.size());
    for (int i = 0; i < 10; ++i) {
        std::vector<double> d;
        for (int b = 0; b < nblk; ++b) if (h[b * 16 + seg[i][0]] && h[b * 16 + seg[i][1]]) d.push_back(((double)h[b * 16 + seg[i][1]] - (double)h[b * 16 + seg[i][0]]) / 100.0);
        if (d.empty()) continue;
        std::sort(d.begin(), d.end());
        printf("  %-26s min %6.2f  median %6.2f  max %6.2f us  (%zu blocks)\n", names[i], d.front(), d[d.size() / 2], d.back(), d.size());
    }
    if (argc > 5 && atoi(argv[5])) {   // CPU check of row 0: attention -> bf16 context -> output dense -> + bias + residual -> LayerNorm
        auto bf = [](bf16_t v) { unsigned u = (unsigned)(unsigned short)v << 16; return __builtin_bit_cast(float, u); };
        auto tobf = [&](float f) { unsigned u = __builtin_bit_cast(unsigned, f); u += 0x7fffu + ((u >> 16) & 1u); return bf((bf16_t)(u >> 16)); };
        std::vector<bf16_t> kvi((size_t)S * 3 * D), kvt((size_t)Tmax * 3 * D), w((size_t)D * D);
        std::vector<float> aob(D), g1(D), b1(D), xin(D), xs(D);
        CK(hipMemcpy(kvi.data(), a.kv_img, kvi.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(kvt.data(), a.kv_txt, kvt.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(w.data(), a.aow, w.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(aob.data(), a.aob, D * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(g1.data(), a.g1, D * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b1.data(), a.b1, D * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(xs.data(), a.xs, D * 4, hipMemcpyDeviceToHost));
        std::vector<float> xin0(D); CK(hipMemcpy(xin0.data(), a.xin, D * 4, hipMemcpyDeviceToHost));
        std::vector<double> x(D);
        std::vector<float> ctx(D);
        const int Lk = S + t0 + 1;
        for (int h = 0; h < H; ++h) {
            std::vector<double> sc(Lk); double mx = -1e30;
            for (int k = 0; k < Lk; ++k) {
                const bf16_t* kp = k < S ? &kvi[(size_t)k * 3 * D + D + h * 64] : &kvt[(size_t)(k - S) * 3 * D + D + h * 64];
                const bf16_t* qp = &kvt[(size_t)t0 * 3 * D + h * 64];
                double s = 0; for (int d = 0; d < 64; ++d) s += (double)bf(qp[d]) * bf(kp[d]);
                sc[k] = s * 0.125; mx = std::max(mx, sc[k]);
            }
            double l = 0; std::vector<double> o(64, 0.0);
            for (int k = 0; k < Lk; ++k) {
                const double p = std::exp(sc[k] - mx); l += p;
                const bf16_t* vp = k < S ? &kvi[(size_t)k * 3 * D + 2 * D + h * 64] : &kvt[(size_t)(k - S) * 3 * D + 2 * D + h * 64];
                for (int d = 0; d < 64; ++d) o[d] += p * bf(vp[d]);
            }
            for (int d = 0; d < 64; ++d) ctx[h * 64 + d] = tobf((float)(o[d] / l));
        }
        double mean = 0, var = 0;
        for (int n = 0; n < D; ++n) { double s = 0; for (int k = 0; k < D; ++k) s += (double)ctx[k] * bf(w[(size_t)n * D + k]); x[n] = s + aob[n] + xin0[n]; mean += x[n]; }
        mean /= D; for (int n = 0; n < D; ++n) var += (x[n] - mean) * (x[n] - mean); var /= D;
        double err = 0; for (int n = 0; n < D; ++n) err = std::max(err, std::fabs((x[n] - mean) / std::sqrt(var + a.eps) * g1[n] + b1[n] - xs[n]));
        printf("row 0 against a double-precision CPU reference: max |dx1| = %.5f (x1 of order 1)\n", err);
    }
    return 0;
}
