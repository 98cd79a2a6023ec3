// Persistent form of the 256x256x64 ping-pong GEMM (see gemm256.hip for the schedule of one K-tile).
//
// One workgroup per CU walks its output tiles v = b, b+G, b+2G, ... (G = grid size <= 256) and the
// LDS-DMA prefetch runs CONTINUOUSLY over the flattened (tile, K-tile) sequence: the first one and a
// half K-tiles of the next output tile are requested during the last K-tiles of the current one, so
// only the very first tile of a workgroup pays the DMA latency, and the epilogue's global traffic
// runs under that prefetch.  Measured on gemm256.hip: ~10 us of fixed cost per 15-us (K=768) tile.
//
// The epilogue cannot reuse the operand stages (they already hold the next tile): it transposes
// through the 32 KiB above them (4 KiB per wave), 16 rows x 256 B per pass, with the 16-byte chunk
// index XORed by the row instead of padding: accumulator-layout writes (16 lanes = 16 rows of one
// chunk) and row-wise reads (16 lanes = 16 chunks of one row) are both bank-conflict free.
// Global stores/loads are full 256-byte row segments, 16 B per lane.
#include "kernels.h"

namespace {

constexpr int STAGE = 65536, HALF = 16384;
constexpr int EP_BASE = 2 * STAGE, EP_WAVE = 4096;
constexpr int LDS_TOTAL = 2 * STAGE + 8 * EP_WAVE;     // 163840 B = all of a CU's LDS

#define BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#define WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

struct Cursor {            // position in the flattened (tile, K-tile) sequence of this workgroup
    int it, t;             // it = index into this workgroup's tile list, t = K-tile inside the tile
    size_t offW, offA;     // element offsets of the tile's first W row / A row
    int m0, n0;
};

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm256p_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2;
    const int wn = wid >> 2, wm = wid & 3;
    const int ntn = a.N >> 8;
    const int ntiles = (a.M >> 8) * ntn;
    const int G = gridDim.x, b = blockIdx.x;
    const int nmy = (ntiles - b + G - 1) / G;                  // tiles of this workgroup (>= 1)
    const int nt = a.K >> 6;
    const int ng = nmy * nt;

    auto seek = [&](Cursor& c, int it) {
        c.it = it; c.t = 0;
        const int v = b + it * G;
        const int lid = xcd_remap(v < ntiles ? v : b, ntiles);  // tiles sharing operand panels share an XCD
        const int tm = lid / ntn, tn = lid - tm * ntn;
        c.m0 = tm << 8; c.n0 = tn << 8;
        c.offW = (size_t)c.n0 * a.K; c.offA = (size_t)c.m0 * a.lda;
    };
    auto advance = [&](Cursor& c) {
        if (++c.t == nt) seek(c, c.it + 1);
    };

    // ---- per-lane parts of the LDS-DMA source addresses (tile independent)
    size_t laneW[2], laneA[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wid * 2 + i) * 8 + (lane >> 3);
        const int chunk = swz_chunk(row, lane & 7);
        laneW[i] = (size_t)row * a.K + chunk * 8;
        laneA[i] = (size_t)row * a.lda + chunk * 8;
    }
    const size_t hiW = (size_t)128 * a.K, hiA = (size_t)128 * a.lda;
    const int dma_off = wid * 2048;
    auto dma_half = [&](char* stage, int which, const Cursor& c) {  // 0 W-lo, 1 W-hi, 2 A-lo, 3 A-hi
        const int k0 = c.t << 6;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bf16_t* src = which < 2 ? a.W + c.offW + laneW[i] + (which & 1) * hiW + k0
                                          : a.A + c.offA + laneA[i] + (which & 1) * hiA + k0;
            __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(stage + which * HALF + dma_off + i * 1024), 16, 0, 0);
        }
    };

    // ---- fragment read offsets (bytes inside a stage)
    const int frow = lane & 15, fq = lane >> 4;
    const int g = (frow >> 1) & 7;
    const int offW = wn * HALF + frow * 128;
    const int offA = 2 * HALF + (wm >> 1) * HALF + ((wm & 1) * 64 + frow) * 128;
    const int c0 = ((0 + fq) ^ g) << 4, c1 = ((4 + fq) ^ g) << 4;

    f32x4 acc[2][4][2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[x][i][y][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    Cursor cur, n1, n2;
    seek(cur, 0);
    n1 = cur; advance(n1);
    n2 = n1; advance(n2);
#pragma unroll
    for (int w = 0; w < 4; ++w) dma_half(smem, w, cur);
    if (ng > 1) {
        dma_half(smem + STAGE, 2, n1);
        dma_half(smem + STAGE, 3, n1);
        dma_half(smem + STAGE, 0, n1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        WAIT_VM0();
    }
    BARRIER();
    if (grp == 1) BARRIER();

    bf16x8 wf[4][2], af[2][2][2];
    for (int gk = 0; gk < ng; ++gk) {
        const char* sb = smem + (gk & 1) * STAGE;
        char* cb = smem + (gk & 1) * STAGE;
        char* nb = smem + ((gk + 1) & 1) * STAGE;
        const bool has1 = (gk + 1) < ng, has2 = (gk + 2) < ng;

        // ---------------- L0 ----------------
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            wf[i][0] = *(const bf16x8*)(sb + offW + i * 2048 + c0);
            wf[i][1] = *(const bf16x8*)(sb + offW + i * 2048 + c1);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            af[0][j][0] = *(const bf16x8*)(sb + offA + j * 2048 + c0);
            af[0][j][1] = *(const bf16x8*)(sb + offA + j * 2048 + c1);
        }
        WAIT_LGKM0();
        SCHED_FENCE();
        BARRIER();
        // ---------------- C0: (N0, M0) ----------------
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[0][i][0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][ks], af[0][j][ks], acc[0][i][0][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        SCHED_FENCE();
        BARRIER();
        // ---------------- L1: M1 frags; DMA W-hi(gk+1) ----------------
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            af[1][j][0] = *(const bf16x8*)(sb + offA + 32 * 128 + j * 2048 + c0);
            af[1][j][1] = *(const bf16x8*)(sb + offA + 32 * 128 + j * 2048 + c1);
        }
        if (has1) dma_half(nb, 1, n1);
        WAIT_LGKM0();
        SCHED_FENCE();
        BARRIER();
        // ---------------- C1: (N0, M1) ----------------
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[0][i][1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][ks], af[1][j][ks], acc[0][i][1][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        SCHED_FENCE();
        BARRIER();
        // ---------------- L2: N1 frags; DMA A-lo(gk+2) ----------------
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            wf[i][0] = *(const bf16x8*)(sb + offW + 64 * 128 + i * 2048 + c0);
            wf[i][1] = *(const bf16x8*)(sb + offW + 64 * 128 + i * 2048 + c1);
        }
        if (has2) dma_half(cb, 2, n2);
        WAIT_LGKM0();
        SCHED_FENCE();
        BARRIER();
        // ---------------- C2: (N1, M1) ----------------
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[1][i][1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][ks], af[1][j][ks], acc[1][i][1][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        SCHED_FENCE();
        BARRIER();
        // ---------------- L3: DMA A-hi, W-lo (gk+2); retire gk+1 ----------------
        if (has2) {
            dma_half(cb, 3, n2);
            dma_half(cb, 0, n2);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            WAIT_VM0();
        }
        SCHED_FENCE();
        BARRIER();
        // ---------------- C3: (N1, M0) ----------------
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[1][i][0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][ks], af[0][j][ks], acc[1][i][0][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        SCHED_FENCE();
        BARRIER();

        // ---------------- tile finished: epilogue (no barriers; private LDS region) ----------------
        if (cur.t == nt - 1) {
            constexpr bool OUT_BF16 = (EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_QGELU_BF16 || EPI == EPI_BIAS_GELU_BF16);
            char* ep = smem + EP_BASE + wid * EP_WAVE;
            const int rrow = lane >> 4, rchk = lane & 15;                  // row-wise role of this lane
            const int nbase = cur.n0 + wn * 128, mbase = cur.m0 + wm * 64;
            if (OUT_BF16) {
                f32x4 bias4[2][4];
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        bias4[x][i] = a.bias ? *(const f32x4*)(a.bias + nbase + x * 64 + i * 16 + fq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int y = 0; y < 2; ++y)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
#pragma unroll
                        for (int x = 0; x < 2; ++x)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                f32x4 v = acc[x][i][y][j] + bias4[x][i];
                                if (EPI == EPI_BIAS_QGELU_BF16) {
#pragma unroll
                                    for (int r = 0; r < 4; ++r) v[r] = quick_gelu(v[r]);
                                } else if (EPI == EPI_BIAS_GELU_BF16) {
#pragma unroll
                                    for (int r = 0; r < 4; ++r) v[r] = erf_gelu(v[r]);
                                }
                                uint2 o;
                                o.x = pack_bf2(v[0], v[1]);
                                o.y = pack_bf2(v[2], v[3]);
                                const int chunk = x * 8 + i * 2 + (fq >> 1);           // 16-B chunk of the 256-B row
                                *(uint2*)(ep + frow * 256 + ((chunk ^ frow) << 4) + (fq & 1) * 8) = o;
                            }
                        WAIT_LGKM0();
#pragma unroll
                        for (int it = 0; it < 4; ++it) {
                            const int row = it * 4 + rrow;
                            const uint4 raw = *(const uint4*)(ep + row * 256 + ((rchk ^ row) << 4));
                            const int m = mbase + y * 32 + j * 16 + row;
                            *(uint4*)((bf16_t*)a.out + (size_t)m * a.ldo + nbase + rchk * 8) = raw;
                        }
                        WAIT_LGKM0();
                    }
            } else {
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    f32x4 bias4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        bias4[i] = (EPI != EPI_PATCH_F32 && a.bias) ? *(const f32x4*)(a.bias + nbase + x * 64 + i * 16 + fq * 4)
                                                                    : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int y = 0; y < 2; ++y)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int chunk = i * 4 + fq;
                                *(f32x4*)(ep + frow * 256 + ((chunk ^ frow) << 4)) = acc[x][i][y][j] + bias4[i];
                            }
                            WAIT_LGKM0();
#pragma unroll
                            for (int it = 0; it < 4; ++it) {
                                const int row = it * 4 + rrow;
                                f32x4 v = *(const f32x4*)(ep + row * 256 + ((rchk ^ row) << 4));
                                const int m = mbase + y * 32 + j * 16 + row;
                                const int n = nbase + x * 64 + rchk * 4;
                                if (EPI == EPI_BIAS_RESID_F32) {
                                    v += *(const f32x4*)(a.resid + (size_t)m * a.ldr + n);
                                    *(f32x4*)((float*)a.out + (size_t)m * a.ldo + n) = v;
                                } else if (EPI == EPI_BIAS_F32) {
                                    *(f32x4*)((float*)a.out + (size_t)m * a.ldo + n) = v;
                                } else if (m < a.valid_rows) {      // EPI_PATCH_F32
                                    const int frame = m / a.patches_per_frame;
                                    const int patch = m - frame * a.patches_per_frame;
                                    v += *(const f32x4*)(a.pos + (size_t)(1 + patch) * a.N + n);
                                    const size_t orow = (size_t)frame * a.tokens_per_frame + 1 + patch;
                                    *(f32x4*)((float*)a.out + orow * a.ldo + n) = v;
                                }
                            }
                            WAIT_LGKM0();
                        }
                }
            }
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int y = 0; y < 2; ++y)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[x][i][y][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        advance(cur);
        advance(n1);
        advance(n2);
    }
    if (grp == 0) BARRIER();
}

template <int EPI>
hipError_t launch_t(const GemmArgs& a, hipStream_t s) {
    static bool attr_done[64] = {false};            // per device: the attribute belongs to the device's code object
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    bool& attr_set = attr_done[dev_ & 63];
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm256p_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
    }
    const int ntiles = (a.M >> 8) * (a.N >> 8);
    const int grid = ntiles < ncu ? ntiles : ncu;
    hipLaunchKernelGGL(gemm256p_kernel<EPI>, dim3(grid), dim3(512), LDS_TOTAL, s, a);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_gemm256p(const GemmArgs& a, int epi, hipStream_t s) {
    if (!gemm256_ok(a)) return hipErrorInvalidValue;
    switch (epi) {
        case EPI_BIAS_BF16: return launch_t<EPI_BIAS_BF16>(a, s);
        case EPI_BIAS_QGELU_BF16: return launch_t<EPI_BIAS_QGELU_BF16>(a, s);
        case EPI_BIAS_GELU_BF16: return launch_t<EPI_BIAS_GELU_BF16>(a, s);
        case EPI_BIAS_RESID_F32: return launch_t<EPI_BIAS_RESID_F32>(a, s);
        case EPI_BIAS_F32: return launch_t<EPI_BIAS_F32>(a, s);
        case EPI_PATCH_F32: return launch_t<EPI_PATCH_F32>(a, s);
    }
    return hipErrorInvalidValue;
}
