// 256x256x64-tile bf16 MFMA GEMM for gfx950 (the dominant kernel of the caption path).
//
//   C[m][n] = sum_k A[m][k] * W[n][k]  (+ fused epilogue), A and W both K-contiguous.
//
// Why 256^2: a 128x128x64 tile needs 32 KiB of operands per 2.1 MFLOP, i.e. 64 B/clk/CU of
// global->LDS fill at the MFMA peak, which is the whole L1/LDS-DMA rate of a CU; at 256^2 the same
// ratio is 32 B/clk.  One workgroup (512 threads = 8 waves, 2 per SIMD) per CU, 128 KiB of LDS.
//
// Wave w = (wn = w >> 2, wm = w & 3) owns the 128(n) x 64(m) block of the tile; weight rows are the
// MFMA A operand and activation rows the B operand (v_mfma_f32_16x16x32_bf16), so an accumulator
// quad is 4 consecutive n of one m and the epilogue moves 8/16-byte vectors.
//
// Schedule ("ping-pong"): waves 0-3 (group 0) and waves 4-7 (group 1) share the SIMDs pairwise
// and run the SAME program one barrier apart, so while one wave of a SIMD is in a LOAD phase
// (LDS-DMA issue for the next K-tiles, lgkmcnt drain) its partner is in a COMPUTE phase
// (16 MFMAs = one 64x32 quadrant over K=64, with the ds_read_b128 fragment reads of the NEXT
// quadrant interleaved between the MFMAs by sched_group_barrier):
//
//     slot      8t+0   8t+1   8t+2   8t+3   8t+4   8t+5   8t+6   8t+7
//     group 0   L0(t)  C0(t)  L1(t)  C1(t)  L2(t)  C2(t)  L3(t)  C3(t)
//     group 1   C3(t-1) L0(t) C0(t)  L1(t)  C1(t)  L2(t)  C2(t)  L3(t)
//
//   L0: read act frags M0 (4)                     C0 (N0,M0): + read act frags M1 (4)
//   L1: LDS-DMA W-hi(t+1)                         C1 (N0,M1): + read W frags N1 (8) -> second register set
//   L2: LDS-DMA A-lo(t+2)                         C2 (N1,M1)
//   L3: LDS-DMA A-hi(t+2), W-lo(t+2); vmcnt(6)    C3 (N1,M0): + read W frags N0 of tile t+1 (8)
//   quadrant order (N0,M0) (N0,M1) (N1,M1) (N1,M0): M0/M1 fragments stay in registers.
// Every slot ends with one s_barrier executed by all 8 waves.  The LDS-DMA prefetch runs ~1.5
// K-tiles ahead through only two stages: a half-tile of tile t is overwritten by tile t+2 as soon
// as both groups are past its last read (A halves: C0(t); W halves: C1(t)).  RAW: the counted
// vmcnt(6) in L3(t) retires every DMA of tile t+1 (the six youngest belong to tile t+2) and tile
// t+1 is first read one barrier later; every LOAD phase drains lgkmcnt before its barrier (WAR).
//
// LDS image per stage (64 KiB): [W-lo | W-hi | A-lo | A-hi], each 128 rows x 128 B, 16-B chunks
// XOR-swizzled by (row>>1)&7 on the DMA SOURCE address and again on the read (conflict-free
// ds_read_b128 for the 16x16x32 operand map, see common.h).
#include "gemm_epilogue.h"
#include "host_logic.h"

namespace {

constexpr int STAGE = 65536, HALF = 16384;
constexpr int LDS_TOTAL = 8 * EPI_REGION;        // 139264 B >= 2 * STAGE

#define BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#define WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

template <int EPI, bool LN8 = false>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(GemmArgs a) {
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
    const bf16_t* srcW[2];
    const bf16_t* srcA[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wid * 2 + i) * 8 + (lane >> 3);            // row inside a 128-row half-tile
        const int chunk = swz_chunk(row, lane & 7);
        srcW[i] = a.W + (size_t)(n0 + row) * a.K + chunk * 8;
        srcA[i] = a.A + (size_t)(m0 + row) * a.lda + chunk * 8;
    }
    const size_t hiW = (size_t)128 * a.K, hiA = (size_t)128 * a.lda;
    const int dma_off = wid * 2048;                                  // this wave's pieces inside a half-tile

    // ---- fragment read offsets (bytes inside a stage)
    const int frow = lane & 15, fq = lane >> 4;
    const int g = (frow >> 1) & 7;
    const int offW = wn * HALF + frow * 128;                                        // + (Nh*64 + i*16)*128
    const int offA = 2 * HALF + (wm >> 1) * HALF + ((wm & 1) * 64 + frow) * 128;     // + (Mh*32 + j*16)*128
    const int c0 = ((0 + fq) ^ g) << 4, c1 = ((4 + fq) ^ g) << 4;                  // k-step 0 / 1 chunk offsets

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
            const bf16_t* src = (which < 2 ? srcW[i] + (which & 1) * hiW : srcA[i] + (which & 1) * hiA) + k0;
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
    const int nt = a.K >> 6;
    LN_STAMP(0);
#pragma unroll
    for (int w = 0; w < 4; ++w) dma_half(smem, w, 0);
    if (nt > 1) {
        dma_half(smem + STAGE, 2, 64);
        dma_half(smem + STAGE, 3, 64);
        dma_half(smem + STAGE, 0, 64);
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
    bf16x8 wf[4][2], wf2[4][2], af[2][2][2];
    {
        const char* sb0 = smem;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            wf[i][0] = *(const bf16x8*)(sb0 + offW + i * 2048 + c0);
            wf[i][1] = *(const bf16x8*)(sb0 + offW + i * 2048 + c1);
        }
    }
    for (int t = 0; t < nt; ++t) {
        const char* sb = smem + (t & 1) * STAGE;
        char* cb = smem + (t & 1) * STAGE;          // stage of tile t == stage of tile t+2
        char* nb = smem + ((t + 1) & 1) * STAGE;
        const bool has1 = (t + 1) < nt, has2 = (t + 2) < nt;
        const int k1 = (t + 1) << 6, k2 = (t + 2) << 6;

        // ---------------- L0: act rows M0 ----------------------------------------------------------
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            af[0][j][0] = *(const bf16x8*)(sb + offA + j * 2048 + c0);
            af[0][j][1] = *(const bf16x8*)(sb + offA + j * 2048 + c1);
        }
        WAIT_LGKM0();
        SCHED_FENCE();
        BARRIER();
        // ---------------- C0: (N0, M0); prefetch M1 ------------------------------------------------
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            af[1][j][0] = *(const bf16x8*)(sb + offA + 32 * 128 + j * 2048 + c0);
            af[1][j][1] = *(const bf16x8*)(sb + offA + 32 * 128 + j * 2048 + c1);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[0][i][0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][ks], af[0][j][ks], acc[0][i][0][j], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // MFMA
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
            wf2[i][0] = *(const bf16x8*)(sb + offW + 64 * 128 + i * 2048 + c0);
            wf2[i][1] = *(const bf16x8*)(sb + offW + 64 * 128 + i * 2048 + c1);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[0][i][1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][ks], af[1][j][ks], acc[0][i][1][j], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // MFMA
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
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[1][i][1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf2[i][ks], af[1][j][ks], acc[1][i][1][j], 0, 0, 0);
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
            wf[i][0] = *(const bf16x8*)(nb + offW + i * 2048 + c0);
            wf[i][1] = *(const bf16x8*)(nb + offW + i * 2048 + c1);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[1][i][0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf2[i][ks], af[0][j][ks], acc[1][i][0][j], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // MFMA
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
        gemm_epilogue_tile_ln<EPI == EPI_RESID_LN_POST, false, LN8>(a, acc, smem, m0, n0, tm, tn, wid, wm, wn, lane);
    else
        gemm_epilogue_wave<EPI>(a, acc, smem + wid * EPI_REGION, m0 + wm * 64, n0 + wn * 128, lane);
#ifdef LN_STAMPS
    if (EPI != EPI_RESID_LN_PRE && EPI != EPI_RESID_LN_POST) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); LN_STAMP(7); }
#endif
}

template <int EPI, bool LN8 = false>
hipError_t launch_t(const GemmArgs& a0, hipStream_t s) {
    static bool attr_done[64] = {false};            // per device: the attribute belongs to the device's code object
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    bool& attr_set = attr_done[dev_ & 63];
    constexpr int LDS = (EPI == EPI_RESID_LN_PRE || EPI == EPI_RESID_LN_POST) ? LN_LDS_TOTAL : LDS_TOTAL;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm256_kernel<EPI, LN8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
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
    hipLaunchKernelGGL((gemm256_kernel<EPI, LN8>), dim3(grid), dim3(512), LDS, s, a);
    return hipGetLastError();
}

}  // namespace

bool gemm256_ok(const GemmArgs& a) { return a.M > 0 && (a.M & 255) == 0 && (a.N & 255) == 0 && (a.K & 63) == 0; }

hipError_t launch_gemm256(const GemmArgs& a, int epi, hipStream_t s) {
    if (!gemm256_ok(a)) return hipErrorInvalidValue;
    switch (epi) {
        case EPI_BIAS_BF16: return launch_t<EPI_BIAS_BF16>(a, s);
        case EPI_BIAS_QGELU_BF16: return launch_t<EPI_BIAS_QGELU_BF16>(a, s);
        case EPI_BIAS_GELU_BF16: return launch_t<EPI_BIAS_GELU_BF16>(a, s);
        case EPI_BIAS_RESID_F32: return launch_t<EPI_BIAS_RESID_F32>(a, s);
        case EPI_BIAS_F32: return launch_t<EPI_BIAS_F32>(a, s);
        case EPI_PATCH_F32: return launch_t<EPI_PATCH_F32>(a, s);
        // (LN8: the instantiation that can also write the e4m3 copy of the LayerNorm output (a.ln_out8, fp8 compute).  The pre-LN
        // form ALWAYS takes it: its epilogue runs at the 256-VGPR limit, and without the run-time `if (a.ln_out8)` block in the
        // normalise loop the register allocator moves its 72 bytes of spills into the residual-add / store loops of phase 1, each
        // reload behind a vmcnt(0): 52.5 instead of 45.5 us per launch at N = K = 768, 96 instead of 84.6 at 1024
        // (tools/gemm_ln_ab.py; tests/test_isa_lint.py keeps the spills out of those loops).)
        case EPI_RESID_LN_PRE: return !(gemm256_ln_ok(a) && a.resid) ? hipErrorInvalidValue : launch_t<EPI_RESID_LN_PRE, true>(a, s);
        case EPI_RESID_LN_POST: return !(gemm256_ln_ok(a) && a.out) ? hipErrorInvalidValue : a.ln_out8 ? launch_t<EPI_RESID_LN_POST, true>(a, s) : launch_t<EPI_RESID_LN_POST>(a, s);
    }
    return hipErrorInvalidValue;
}

bool gemm256_ln_ok(const GemmArgs& a) {
    return gemm256_ok(a) && (a.N == 768 || a.N == 1024) && a.ln_g && a.ln_b && a.ln_out && a.ln_stats && a.ln_cnt;
}
