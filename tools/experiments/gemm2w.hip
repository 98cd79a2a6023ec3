// 256(n) x 128(m) x 64-tile bf16 MFMA GEMM for gfx950, TWO independent workgroups per CU (plain epilogues).
//
// STATUS (round 5): measured experiment, NOT in libgitcap.so (tools/build_diag.py builds it behind gitcap_dbg_gemm(tile = 258);
// tools/gemm2w_bench.py).  Bitwise equal to gemm256 on every plain epilogue.  Per launch at the bench shape it wins where the
// epilogue is long -- FC1 + QuickGELU -8 %, FC1 + erf GELU -9 % -- is even on q|k|v and loses where the K loop dominates (K = 3072:
// +8 ... +14 %; GIT-large K = 1024: +6 %): its K loop is ~10 % slower (1.5 x the operand bytes through the CU's global->LDS path,
// 12 LDS-DMA pieces per wave and K-tile).  End to end with FC1 + GELU routed to it: serial +0.3 %, pipelined -1.7 % (1898 vs
// 1931 captions/s, three interleaved rounds): two 80 KiB workgroups fill every CU, and the token loops of the batches in flight
// lose the CUs a 256 x 256 launch leaves them.  profiles/r05_gemm2w_two_workgroups_per_cu.txt has the numbers and the ablations.
//
//   C[m][n] = sum_k A[m][k] * W[n][k]  (+ fused epilogue), A and W both K-contiguous.
//
// Why (profiles/r05_gemm_timeline.txt): with one 256 x 256 workgroup per CU (gemm256.hip) every CU of a round is in the same
// phase.  The K loop runs at 82 % of what the clock allows (16.1 us for K = 768), but around it a tile spends 0.5 us waiting
// for dispatch, 1.6 us for its first operands and 4.7 - 6.4 us in an epilogue that is a chip-wide write burst at fabric speed
// (33.5 MB per round at 7 TB/s) while every MFMA pipe idles: 30 - 35 % of a round.  Two workgroups that share a CU drift apart
// and one's prologue / epilogue runs under the other's K loop.
//
// A workgroup is ONE of gemm256's two ping-pong groups: 256 threads = 4 waves (one per SIMD), wave w = (wn = w >> 1,
// wm = w & 1) owns the same 128(n) x 64(m) block, reads the same fragments in the same order and issues the same sequence of
// v_mfma_f32_16x16x32_bf16 over ascending k: results are bitwise those of gemm256 / gemm / gemm_mt.  What differs is the LDS
// budget: 80 KiB per workgroup (2 x 80 = the CU's 160 KiB), against 96 KiB for two stages of [W 256 rows | A 128 rows] x 128 B.
// The W panel keeps two stages (4 slots of 16 KiB); the A panel has ONE slot: every A fragment of a K-tile is in registers
// after the first compute phase (as in gemm256), so the slot is refilled with the next K-tile right behind it.  Rows are
// 128 B (full cache lines per LDS-DMA row; the round-1 experiment with 64-byte rows, tools/experiments/gemm2b.hip, saturated
// the CU's global->LDS path).
//
// One K-tile t of a wave (3 barriers; the partner of every stall is the OTHER workgroup's wave on the same SIMD):
//
//     L0          read A frags M0(t)                                   [A(t) landed: B0 of tile t-1]
//     C0 (N0,M0)  16 MFMA  + read A frags M1(t)
//     -- B1 --    A slot free                  -> LDS-DMA A(t+1)       (4 instructions per wave)
//     C1 (N0,M1)  16 MFMA  + read W frags N1(t) -> second register set
//     -- B2 --    W stage t&1 free             -> LDS-DMA W(t+2)       (8 instructions per wave)
//     C2 (N1,M1)  16 MFMA
//     vmcnt(8)  -- B3 --                          W(t+1) and A(t+1) landed (W(t+2) stays in flight)
//     C3 (N1,M0)  16 MFMA  + read W frags N0(t+1)
//
// LDS image of a slot: 128 rows x 128 B, 16-B chunks XOR-swizzled by (row>>1)&7 on the DMA SOURCE address and again on the
// read (common.h), exactly gemm256's half-tile.
#include "gemm_epilogue.h"

namespace {

// num_records of a raw buffer resource over `elems` bf16 elements (bytes, clamped to the int builtin argument)
__device__ __forceinline__ int dma_range(size_t elems) {
    const size_t b = elems * 2;
    return b > 0x7fffffffull ? 0x7fffffff : (int)b;
}

constexpr int SLOT = 16384;                      // 128 rows x 128 B
constexpr int W_STAGE = 2 * SLOT;                // W-lo | W-hi of one K-tile
constexpr int A_OFF = 2 * W_STAGE;               // the A slot
constexpr int LDS2W = A_OFF + SLOT;              // 81920: two workgroups fill the CU's 160 KiB
static_assert(LDS2W >= 4 * EPI_REGION, "epilogue staging must fit the operand slots");

#define BARRIER2W() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#define SCHED_FENCE2W() __builtin_amdgcn_sched_barrier(0)

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm2w_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wid >> 1, wm = wid & 1;
    const int ntn = a.N >> 8;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = lid / ntn, tn = lid - tm * ntn;
    const int m0 = tm << 7, n0 = tn << 8;

    // ---- LDS-DMA sources: wave w moves pieces 4w .. 4w+3 (8 rows x 128 B each) of every 128-row slot, as buffer loads to LDS
    // (buffer_load_dwordx4 v_off, s[rsrc], s_off offen lds): a wave-uniform byte offset in an SGPR (tile origin + piece + half +
    // K-tile) plus ONE 32-bit lane offset (row-in-piece, swizzled chunk) -- 4 instead of 8 address bytes per lane, no 64-bit
    // pointer in VGPRs.  The swizzle (row >> 1) & 7 of row = (4 wid + i) 8 + (lane >> 3) is (4 i + (lane >> 4)) & 7: two lane
    // offsets, for even and odd i.
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)a.W, 0, dma_range((size_t)a.N * a.K), 0x00020000);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, dma_range((size_t)a.M * a.lda), 0x00020000);
    unsigned lofW[2], lofA[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const unsigned chunk = (unsigned)(lane & 7) ^ (unsigned)((4 * e + (lane >> 4)) & 7);
        lofW[e] = ((unsigned)(lane >> 3) * (unsigned)a.K + chunk * 8u) * 2u;
        lofA[e] = ((unsigned)(lane >> 3) * (unsigned)a.lda + chunk * 8u) * 2u;
    }
    const unsigned uW = (unsigned)(n0 + wid * 32) * (unsigned)a.K * 2u, uA = (unsigned)(m0 + wid * 32) * (unsigned)a.lda * 2u;   // wave-uniform
    const unsigned pieceW = 8u * (unsigned)a.K * 2u, pieceA = 8u * (unsigned)a.lda * 2u, hiW = 128u * (unsigned)a.K * 2u;
    const int dma_off = wid * 4096;
    auto dma_w = [&](char* stage, int k0) {          // W-lo and W-hi of one K-tile: 8 instructions
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, LDS_PTR(stage + h * SLOT + dma_off + i * 1024), 16, lofW[i & 1],
                                                         uW + h * hiW + i * pieceW + (unsigned)k0 * 2u, 0, 0);
    };
    auto dma_a = [&](int k0) {                       // the A slot: 4 instructions
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LDS_PTR(smem + A_OFF + dma_off + i * 1024), 16, lofA[i & 1],
                                                     uA + i * pieceA + (unsigned)k0 * 2u, 0, 0);
    };

    // ---- fragment read offsets
    const int frow = lane & 15, fq = lane >> 4;
    const int g = (frow >> 1) & 7;
    const int offW = wn * SLOT + frow * 128;                         // + (Nh*64 + i*16)*128 inside a W stage
    const int offA = A_OFF + (wm * 64 + frow) * 128;                 // + (Mh*32 + j*16)*128
    const int c0 = ((0 + fq) ^ g) << 4, c1 = ((4 + fq) ^ g) << 4;   // k-step 0 / 1 chunk offsets

    f32x4 acc[2][4][2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[x][i][y][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nt = a.K >> 6;
    dma_w(smem, 0);
    dma_a(0);
    if (nt > 1) {
        dma_w(smem + W_STAGE, 64);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    BARRIER2W();

    bf16x8 wf[4][2], wf2[4][2], af[2][2][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        wf[i][0] = *(const bf16x8*)(smem + offW + i * 2048 + c0);
        wf[i][1] = *(const bf16x8*)(smem + offW + i * 2048 + c1);
    }
    // (has1 / has2 are compile-time in each instantiation: the LDS-DMA pieces sit in the same basic block as the MFMAs they are
    // interleaved with; the last two K-tiles are peeled)
    auto ktile = [&](int t, auto has1_c, auto has2_c) {
        constexpr bool has1 = decltype(has1_c)::value, has2 = decltype(has2_c)::value;
        const char* sb = smem + (t & 1) * W_STAGE;          // W stage of tile t (== stage of tile t+2)
        char* cb = smem + (t & 1) * W_STAGE;
        const char* nb = smem + ((t + 1) & 1) * W_STAGE;

        // ---------------- L0: act rows M0 ----------------------------------------------------------
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            af[0][j][0] = *(const bf16x8*)(smem + offA + j * 2048 + c0);
            af[0][j][1] = *(const bf16x8*)(smem + offA + j * 2048 + c1);
        }
        // ---------------- C0: (N0, M0); prefetch M1 ------------------------------------------------
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            af[1][j][0] = *(const bf16x8*)(smem + offA + 32 * 128 + j * 2048 + c0);
            af[1][j][1] = *(const bf16x8*)(smem + offA + 32 * 128 + j * 2048 + c1);
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
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // every A fragment of the tile is in registers
        SCHED_FENCE2W();
        BARRIER2W();                                              // B1: the A slot is free
        // ---------------- C1: (N0, M1); prefetch N1; LDS-DMA A(t+1) between the MFMAs ----------------
        if (has1) dma_a((t + 1) << 6);
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
            if (q & 1) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);   // VMEM (an LDS-DMA piece)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // every W fragment of the tile is in registers
        SCHED_FENCE2W();
        BARRIER2W();                                              // B2: W stage t&1 is free
        // ---------------- C2: (N1, M1); LDS-DMA W(t+2) between the MFMAs -----------------------------
        if (has2) dma_w(cb, (t + 2) << 6);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[1][i][1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf2[i][ks], af[1][j][ks], acc[1][i][1][j], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // MFMA
            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);   // VMEM (an LDS-DMA piece)
        }
        SCHED_FENCE2W();
        // W(t+1) and A(t+1) landed; W(t+2) (8 pieces, issued behind them) stays in flight
        if (has2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (has1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SCHED_FENCE2W();
        BARRIER2W();                                              // B3
        // ---------------- C3: (N1, M0); prefetch N0 of tile t+1 --------------------------------------
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
        SCHED_FENCE2W();
        // (no barrier here: the next tile reads the A slot, published at B3, and nothing is overwritten before its B1)
        if (!has1) {                                              // last K-tile: the operand slots become the epilogue's staging
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            SCHED_FENCE2W();
            BARRIER2W();
        }
    };
    for (int t = 0; t + 2 < nt; ++t) ktile(t, std::true_type{}, std::true_type{});
    if (nt > 1) ktile(nt - 2, std::true_type{}, std::false_type{});
    ktile(nt - 1, std::false_type{}, std::false_type{});

    // ---- epilogue through LDS (gemm_epilogue.h; the operand slots are dead after the last barrier) ----
    gemm_epilogue_wave<EPI>(a, acc, smem + wid * EPI_REGION, m0 + wm * 64, n0 + wn * 128, lane);
}

template <int EPI>
hipError_t launch_t(const GemmArgs& a, hipStream_t s) {
    static bool attr_done[64] = {false};            // per device: the attribute belongs to the device's code object
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    bool& attr_set = attr_done[dev_ & 63];
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm2w_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2W);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int grid = (a.M >> 7) * (a.N >> 8);
    hipLaunchKernelGGL(gemm2w_kernel<EPI>, dim3(grid), dim3(256), LDS2W, s, a);
    return hipGetLastError();
}

}  // namespace

bool gemm2w_ok(const GemmArgs& a) { return a.M > 0 && (a.M & 127) == 0 && (a.N & 255) == 0 && (a.K & 63) == 0; }

hipError_t launch_gemm2w(const GemmArgs& a, int epi, hipStream_t s) {
    if (!gemm2w_ok(a)) return hipErrorInvalidValue;
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
