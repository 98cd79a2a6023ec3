// What does the text-row attention of one batch cost the image-pass GEMMs of another?  (diagnostic, not product)
// Stream A: a train of 256x256 GEMM launches (QKV shape of the ViT, M = 18944).  Stream B: a train of text-attention
// launches, (a) the 16-wave workgroup form (txt_block_kernel: needs a CU to itself), (b) the single-wave split form
// (attn_split_kernel: <= 48 VGPRs, 6 KiB LDS, fits beside a GEMM workgroup).  Prints the GEMM's mean time alone and
// next to each, and each attention form's mean time alone and next to the GEMM.
// Result (MI355X): GEMM alone 69 us; + one attention launch per GEMM 76.8 (block form) / 78.6 us (split form); dense
// attention train 88.6 / 98.5 us: the cost is the HBM bandwidth of the streamed K/V, not CU occupancy.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -I real-time-video-captioning_amd/csrc \
//            tools/probe/coattn_probe.hip -o tools/probe/coattn_probe
#include "gemm256.hip"
#include "txtblock.hip"
#include "../experiments/attn_split.hip"
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
        else if (fill < 0) h[i] = (T)0;
        else { float f = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; h[i] = *(T*)&f; }
    }
    CK(hipMemcpy(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice));
    return p;
}

int main() {
    const int rows = 16, S = 1182, D = 768, H = 12, Tmax = 32, t0 = 10, M = rows;
    // GEMM operands
    GemmArgs g{};
    const int GM = 18944, GN = 2304, GK = 768;
    g.A = dalloc<bf16_t>((size_t)GM * GK, 1); g.lda = GK; g.W = dalloc<bf16_t>((size_t)GN * GK, 2); g.bias = dalloc<float>(GN, 3);
    g.M = GM; g.N = GN; g.K = GK; g.out = dalloc<bf16_t>((size_t)GM * GN, 4); g.ldo = GN;
    // attention operands
    bf16_t* kv_img = dalloc<bf16_t>((size_t)rows * S * 3 * D, 8);
    bf16_t* kv_txt = dalloc<bf16_t>((size_t)rows * Tmax * 3 * D, 9);
    TxtBlockArgs a{};
    a.eps = 1e-5f; a.kv_img = kv_img; a.kv_txt = kv_txt;
    a.rows = rows; a.beams = 1; a.t0 = t0; a.T = 1; a.Tmax = Tmax; a.S_img = S; a.H = H; a.D = D;
    a.aow = dalloc<bf16_t>((size_t)D * D, 10); a.aob = dalloc<float>(D, 11); a.g1 = dalloc<float>(D, 12); a.b1 = dalloc<float>(D, 13);
    a.xin = dalloc<float>((size_t)M * D, 3);
    a.part = dalloc<float>((size_t)M * H * D, -1); a.cnt = dalloc<unsigned>(M, -1);
    a.xs = dalloc<float>((size_t)M * D, -1); a.xsb = dalloc<bf16_t>((size_t)M * D, 14);
    TxtSplitArgs sp{};
    sp.kv_img = kv_img; sp.kv_txt = kv_txt; sp.rows = rows; sp.beams = 1; sp.t0 = t0; sp.T = 1; sp.Tmax = Tmax; sp.S_img = S; sp.H = H; sp.D = D;
    const int nc = attn_split_chunks(S + Tmax);
    sp.part_o = dalloc<float>((size_t)M * H * nc * 64, -1); sp.part_ml = dalloc<float>((size_t)M * H * nc * 2, -1);
    sp.cnt = dalloc<unsigned>((size_t)M * H, -1); sp.ctx = dalloc<bf16_t>((size_t)M * D, 15);

    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t a0, a1, b0, b1; CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
    const int NG = 60;
    auto run = [&](const char* name, int kind, int nattn) {       // kind 0: none, 1: 16-wave block form, 2: split form
        CK(hipDeviceSynchronize());
        if (kind) {
            CK(hipEventRecord(b0, sb));
            for (int i = 0; i < nattn; ++i) CK(kind == 1 ? launch_txt_block(a, sb) : launch_attn_split(sp, sb));
            CK(hipEventRecord(b1, sb));
        }
        CK(hipEventRecord(a0, sa));
        for (int i = 0; i < NG; ++i) CK(launch_gemm256(g, EPI_BIAS_BF16, sa));
        CK(hipEventRecord(a1, sa));
        CK(hipDeviceSynchronize());
        float ga = 0, at = 0;
        CK(hipEventElapsedTime(&ga, a0, a1));
        if (kind) CK(hipEventElapsedTime(&at, b0, b1));
        printf("%-44s GEMM %.1f us each (train %.2f ms)   attention %.2f us each (train %.2f ms)\n", name, ga * 1e3 / NG, ga,
               kind ? at * 1e3 / nattn : 0.f, at);
    };
    auto alone = [&](const char* name, int kind, int nattn) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(b0, sb));
        for (int i = 0; i < nattn; ++i) CK(kind == 1 ? launch_txt_block(a, sb) : launch_attn_split(sp, sb));
        CK(hipEventRecord(b1, sb));
        CK(hipDeviceSynchronize());
        float at = 0; CK(hipEventElapsedTime(&at, b0, b1));
        printf("%-44s alone: %.2f us each\n", name, at * 1e3 / nattn);
    };
    for (int rep = 0; rep < 2; ++rep) {
        run("GEMM alone", 0, 0);
        alone("16-wave block form", 1, 200);
        alone("single-wave split form", 2, 200);
        // attention trains sized to last about as long as the GEMM train
        run("GEMM + 16-wave block form", 1, 220);
        run("GEMM + single-wave split form", 2, 220);
        run("GEMM + 16-wave block form (sparse: 1 per GEMM)", 1, 60);
        run("GEMM + split form (sparse: 1 per GEMM)", 2, 60);
    }
    return 0;
}
