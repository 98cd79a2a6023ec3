// 256x256x128-tile fp8 (OCP e4m3) MFMA GEMM for gfx950: the opt-in "fp8 compute" form of the two FFN GEMMs of the image
// pass (BASELINE configs[4]; gitcap_set_compute).
//
//   C[m][n] = (sum_k A8[m][k] * W8[n][k]) * ascale * wscale[n]  (+ fused epilogue)
//
// A8 = activations quantised to e4m3 with ONE static power-of-two scale per producer (LayerNorm outputs, GELU outputs:
// written by the producing epilogues), W8 = the e4m3 weight codes exactly as e4m3 storage keeps them (row-major [N][K]
// bytes + a power-of-two scale per row): no staging, no expansion.
//
// The kernel IS gemm256.hip with one thing changed: a K-tile is 128 bytes = 128 k instead of 64 k.  Same LDS image (128 rows
// x 128 B per half-tile, same XOR swizzle), same LDS-DMA pieces, same ping-pong schedule and barriers, same number of
// ds_read_b128 per K-tile; a lane's operand for v_mfma_f32_16x16x128_f8f6f4 is 32 contiguous bytes of its row (k = 32 fq
// .. 32 fq + 31) = the two 16-byte chunks 2 fq, 2 fq + 1, and the two 16x16x32 bf16 MFMAs per (i, j) and K-tile become ONE
// 16x16x128 fp8 MFMA of twice the cycles: the same MFMA time per K-tile for twice the k, i.e. half the K loop.  The
// accumulator layout is that of every 16x16 MFMA, so the epilogues of gemm_epilogue.h apply unchanged, after the
// accumulators are multiplied by ascale * wscale[n] (SCALED form).
#include "gemm_epilogue.h"
#include "host_logic.h"

namespace {

constexpr int STAGE = 65536, HALF = 16384;
constexpr int LDS_TOTAL = 8 * EPI_REGION;        // 139264 B >= 2 * STAGE
typedef __attribute__((ext_vector_type(8))) int v8i_t;

#define BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#define WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

// two 16-byte LDS reads -> the 32-byte (8-VGPR) operand of the K = 128 MFMA
__device__ __forceinline__ v8i_t cat8(const bf16x8& lo, const bf16x8& hi) {
    typedef __attribute__((ext_vector_type(4))) int v4i_t;
    const v4i_t a = __builtin_bit_cast(v4i_t, lo), b = __builtin_bit_cast(v4i_t, hi);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm256f8_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2;                      // ping-pong group
    const int wn = wid >> 2, wm = wid & 3;
    const int ntn = a.N >> 8;
    int tm, tn;
    if ((EPI == EPI_RESID_LN_PRE || EPI == EPI_RESID_LN_POST) && a.ln_rowblock_map) {
        // The tiles of a row block wait for each other (statistics exchange): they must never straddle two XCDs' dispatch
        // sequences.  Workgroup b runs on XCD b % 8 as that XCD's (b / 8)-th workgroup, so XCD x is handed WHOLE row blocks
        // (a balanced contiguous range) whose ntn tiles are consecutive in its sequence: a waiting tile only ever waits
        // for a sibling that is resident on the same XCD or next in line for it.  The grid is padded to
        // 8 * ntn * ceil(row blocks / 8); the surplus workgroups (last in every sequence) leave at once.
        if (!ln_tile_of_block(blockIdx.x, a.M >> 8, ntn, &tm, &tn)) return;     // host_logic.h (tested on the CPU)
    } else {
        const int lid = xcd_remap(blockIdx.x, gridDim.x);
        tm = lid / ntn; tn = lid - tm * ntn;
    }
    const int m0 = tm << 8, n0 = tn << 8;

    // ---- LDS-DMA source addresses: wave w moves pieces 2w, 2w+1 (8 rows each) of every half-tile
    const unsigned char* srcW[2];
    const unsigned char* srcA[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wid * 2 + i) * 8 + (lane >> 3);            // row inside a 128-row half-tile
        const int chunk = swz_chunk(row, lane & 7);
        srcW[i] = (const unsigned char*)a.W + (size_t)(n0 + row) * a.K + chunk * 16;
        srcA[i] = (const unsigned char*)a.A + (size_t)(m0 + row) * a.lda + chunk * 16;
    }
    const size_t hiW = (size_t)128 * a.K, hiA = (size_t)128 * a.lda;
    const int dma_off = wid * 2048;                                  // this wave's pieces inside a half-tile

    // ---- fragment read offsets (bytes inside a stage)
    const int frow = lane & 15, fq = lane >> 4;
    const int g = (frow >> 1) & 7;
    const int offW = wn * HALF + frow * 128;                                        // + (Nh*64 + i*16)*128
    const int offA = 2 * HALF + (wm >> 1) * HALF + ((wm & 1) * 64 + frow) * 128;     // + (Mh*32 + j*16)*128
    const int c0 = ((2 * fq) ^ g) << 4, c1 = ((2 * fq + 1) ^ g) << 4;            // the two 16-byte halves of the lane's 32 k

    f32x4 acc[2][4][2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[x][i][y][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // one half-tile (2 DMA instructions per wave): which = 0 W-lo, 1 W-hi, 2 A-lo, 3 A-hi
    auto dma_half = [&](char* stage, int which, int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned char* src = (which < 2 ? srcW[i] + (which & 1) * hiW : srcA[i] + (which & 1) * hiA) + k0;
            __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(stage + which * HALF + dma_off + i * 1024), 16, 0, 0);
        }
    };

    // Prefetch runs ~1.5 K-tiles ahead with the 8 DMA instructions of a tile spread over four LOAD
    // phases (2 each), each issued as soon as BOTH groups have finished reading the half-tile it
    // overwrites:   L2(t): A-lo(t+2)   L3(t): A-hi(t+2), W-lo(t+2)   L1(t+1): W-hi(t+2)
    // (A halves are last read in C0(t), W halves in C1(t); group 1 trails group 0 by one slot.)
    // The only wait is a COUNTED one in L3(t): vmcnt(6) leaves the six youngest DMAs (all of them
    // for tile t+2) in flight and retires everything of tile t+1, which is first read one barrier
    // later, in C3(t) (W-lo rows of tile t+1) and L0(t+1).
    const int nt = a.K >> 7;
    LN_STAMP(0);
#pragma unroll
    for (int w = 0; w < 4; ++w) dma_half(smem, w, 0);
    if (nt > 1) {
        dma_half(smem + STAGE, 2, 128);
        dma_half(smem + STAGE, 3, 128);
        dma_half(smem + STAGE, 0, 128);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        WAIT_VM0();
    }
    BARRIER();
    if (grp == 1) BARRIER();                       // group 1 runs one slot behind group 0

    // Fragment reads are issued one COMPUTE phase ahead of their use (LDS reads between MFMAs are nearly
    // free), so the LOAD phases only issue LDS-DMA and drain lgkmcnt:
    //   C0 (N0,M0): + read M1        C1 (N0,M1): + read N1 -> wf2      C2 (N1,M1)
    //   C3 (N1,M0): + read N0 of tile t+1 -> wf (valid: C3 follows the vmcnt wait + barrier of L3)
    //   L0: read M0 of this tile (its registers are still in use during the previous C3)
    v8i_t wf[4], wf2[4], af[2][2];          // 32-byte operands: the two 16-byte reads land in adjacent registers
    {
        const char* sb0 = smem;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            wf[i] = cat8(*(const bf16x8*)(sb0 + offW + i * 2048 + c0), *(const bf16x8*)(sb0 + offW + i * 2048 + c1));
        }
    }
    for (int t = 0; t < nt; ++t) {
        const char* sb = smem + (t & 1) * STAGE;
        char* cb = smem + (t & 1) * STAGE;          // stage of tile t == stage of tile t+2
        char* nb = smem + ((t + 1) & 1) * STAGE;
        const bool has1 = (t + 1) < nt, has2 = (t + 2) < nt;
        const int k1 = (t + 1) << 7, k2 = (t + 2) << 7;

        // ---------------- L0: act rows M0 ----------------------------------------------------------
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            af[0][j] = cat8(*(const bf16x8*)(sb + offA + j * 2048 + c0), *(const bf16x8*)(sb + offA + j * 2048 + c1));
        }
        WAIT_LGKM0();
        SCHED_FENCE();
        BARRIER();
        // ---------------- C0: (N0, M0); prefetch M1 ------------------------------------------------
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            af[1][j] = cat8(*(const bf16x8*)(sb + offA + 32 * 128 + j * 2048 + c0), *(const bf16x8*)(sb + offA + 32 * 128 + j * 2048 + c1));
        }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[0][i][0][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[i], af[0][j], acc[0][i][0][j], 0, 0, 0, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
        }
        __builtin_amdgcn_s_setprio(0);
        SCHED_FENCE();
        BARRIER();
        // ---------------- L1: DMA W-hi of tile t+1 --------------------------------------------------
        if (has1) dma_half(nb, 1, k1);
        WAIT_LGKM0();
        SCHED_FENCE();
        BARRIER();
        // ---------------- C1: (N0, M1); prefetch N1 -------------------------------------------------
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            wf2[i] = cat8(*(const bf16x8*)(sb + offW + 64 * 128 + i * 2048 + c0), *(const bf16x8*)(sb + offW + 64 * 128 + i * 2048 + c1));
        }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[0][i][1][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[i], af[1][j], acc[0][i][1][j], 0, 0, 0, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
        }
        __builtin_amdgcn_s_setprio(0);
        SCHED_FENCE();
        BARRIER();
        // ---------------- L2: DMA A-lo of tile t+2 ---------------------------------------------------
        if (has2) dma_half(cb, 2, k2);
        WAIT_LGKM0();
        SCHED_FENCE();
        BARRIER();
        // ---------------- C2: (N1, M1) ---------------------------------------------------------------
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[1][i][1][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf2[i], af[1][j], acc[1][i][1][j], 0, 0, 0, 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        SCHED_FENCE();
        BARRIER();
        // ---------------- L3: DMA A-hi, W-lo of tile t+2; retire tile t+1 ---------------------------
        if (has2) {
            dma_half(cb, 3, k2);
            dma_half(cb, 0, k2);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            WAIT_VM0();
        }
        SCHED_FENCE();
        BARRIER();
        // ---------------- C3: (N1, M0); prefetch N0 of tile t+1 --------------------------------------
        __builtin_amdgcn_s_setprio(1);
        // (on the last tile this reads the other stage's stale image: in bounds, never used)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            wf[i] = cat8(*(const bf16x8*)(nb + offW + i * 2048 + c0), *(const bf16x8*)(nb + offW + i * 2048 + c1));
        }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[1][i][0][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf2[i], af[0][j], acc[1][i][0][j], 0, 0, 0, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
        }
        __builtin_amdgcn_s_setprio(0);
        SCHED_FENCE();
        BARRIER();
    }
    if (grp == 0) BARRIER();                       // matches group 1's extra leading barrier
    LN_STAMP(1);

    // ---- epilogue through LDS (gemm_epilogue.h; the operand stages are dead after the last barrier) ----
    if (EPI == EPI_RESID_LN_PRE || EPI == EPI_RESID_LN_POST)
        gemm_epilogue_tile_ln<EPI == EPI_RESID_LN_POST, true, EPI == EPI_RESID_LN_PRE>(a, acc, smem, m0, n0, tm, tn, wid, wm, wn, lane);   // (pre-LN: the LN8 form, for its register allocation -- gemm256.hip: launch_gemm256)
    else
        gemm_epilogue_wave<EPI, true>(a, acc, smem + wid * EPI_REGION, m0 + wm * 64, n0 + wn * 128, lane);
#ifdef LN_STAMPS
    if (EPI != EPI_RESID_LN_PRE && EPI != EPI_RESID_LN_POST) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); LN_STAMP(7); }
#endif
}

template <int EPI>
hipError_t launch_t(const GemmArgs& a0, hipStream_t s) {
    static bool attr_done[64] = {false};            // per device: the attribute belongs to the device's code object
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    bool& attr_set = attr_done[dev_ & 63];
    constexpr int LDS = (EPI == EPI_RESID_LN_PRE || EPI == EPI_RESID_LN_POST) ? LN_LDS_TOTAL : LDS_TOTAL;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm256f8_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    constexpr bool LN = (EPI == EPI_RESID_LN_PRE || EPI == EPI_RESID_LN_POST);
    GemmArgs a = a0;
    int grid = (a.M >> 8) * (a.N >> 8);
    if (LN) {                                       // whole row blocks per XCD wherever the grid runs in rounds (host_logic.h)
        a.ln_rowblock_map = ln_use_rowblock_map(a.M >> 8, a.N >> 8, device_cus()) ? 1 : 0;
        if (a.ln_rowblock_map) grid = ln_grid_size(a.M >> 8, a.N >> 8);
    }
    hipLaunchKernelGGL(gemm256f8_kernel<EPI>, dim3(grid), dim3(512), LDS, s, a);
    return hipGetLastError();
}

}  // namespace

// A = e4m3 bytes [M][lda], W = e4m3 bytes [N][K], K a multiple of 128; a.wscale [N] and a.ascale scale the accumulators
bool gemm256f8_ok(const GemmArgs& a) {
    return a.M > 0 && (a.M & 255) == 0 && (a.N & 255) == 0 && (a.K & 127) == 0 && a.wscale && a.ascale > 0.f && (a.lda & 15) == 0;
}

hipError_t launch_gemm256f8(const GemmArgs& a, int epi, hipStream_t s) {
    if (!gemm256f8_ok(a)) return hipErrorInvalidValue;
    switch (epi) {
        case EPI_BIAS_BF16: return launch_t<EPI_BIAS_BF16>(a, s);
        case EPI_BIAS_F32: return launch_t<EPI_BIAS_F32>(a, s);
        case EPI_BIAS_QGELU_F8: return a.out8_inv > 0.f ? launch_t<EPI_BIAS_QGELU_F8>(a, s) : hipErrorInvalidValue;
        case EPI_BIAS_GELU_F8: return a.out8_inv > 0.f ? launch_t<EPI_BIAS_GELU_F8>(a, s) : hipErrorInvalidValue;
        case EPI_RESID_LN_PRE: return gemm256_ln_ok(a) && a.resid ? launch_t<EPI_RESID_LN_PRE>(a, s) : hipErrorInvalidValue;
        case EPI_RESID_LN_POST: return gemm256_ln_ok(a) && a.out ? launch_t<EPI_RESID_LN_POST>(a, s) : hipErrorInvalidValue;
    }
    return hipErrorInvalidValue;
}
