// 256(n) x 128(m) x 32-tile bf16 MFMA GEMM for gfx950, TWO independent workgroups per CU.
//
//   C[m][n] = sum_k A[m][k] * W[n][k]  (+ fused epilogue), A and W both K-contiguous.
//
// Why: with one 256x256 workgroup per CU (gemm256.hip) every CU of a round is in the same phase, so
// the HBM-bound epilogue bursts (all outputs of a round at once) and the first-tile latency never
// overlap the MFMA-bound K loops (tools/gemm_timeline.py: loop 15.6 us of a 22-25 us round at K=768).
// Here a workgroup is one HALF of that schedule: 256 threads = 4 waves (one per SIMD), 72 KiB of LDS,
// so two workgroups share a CU and drift apart: one's prologue/epilogue runs under the other's K loop.
// The wave block is the same 128(n) x 64(m) (wave w: wn = w >> 1, wm = w & 1), the arithmetic per
// output element is the same sequence of v_mfma_f32_16x16x32_bf16 over ascending k, so results are
// bitwise identical to gemm256 / gemm.
//
// STATUS: measured experiment, reachable only through gitcap_dbg_gemm(tile = 258).  It is bitwise equal to
// gemm256 but 3-25 % slower: the overlap is real (with the in-loop LDS-DMA removed it beats gemm256 by
// 18-20 % on the multi-round shapes) but two 256x128 tiles per CU move 1.5x the operand bytes of one
// 256x256 tile through the CU's global->LDS path, in 64-byte row segments, and that path becomes the
// limit (ablations: dropping the vmcnt wait or the barrier changes nothing, dropping the DMA -25 %).
//
// K step 32 (one MFMA K), three LDS stages of 24 KiB: [W 256 rows x 64 B | A 128 rows x 64 B].
// Per step and wave: 6 LDS-DMA instructions (16 rows x 64 B each), 12 ds_read_b128, 32 MFMAs.
// The fragments of step t+1 are read (into a second register set) between the MFMAs of step t:
//
//   iteration t:  s_waitcnt lgkmcnt(0) + vmcnt(6 | 0)   frags(t) in registers, own DMAs of step t+1 landed
//                 s_barrier                             => step t+1 visible, stage t%3 no longer read
//                 LDS-DMA step t+3 -> stage t%3
//                 32 MFMAs of step t  ||  12 reads of step t+1     (sched_group_barrier 8 : 3)
//
// 64-byte rows: 16-B chunk c of row r is stored at chunk position c ^ s(r), s = {0,3,2,1}[(r>>2)&3]
// (applied on the DMA source address, the LDS image is lane-linear, and again on the read), which makes
// every 16-lane group of a ds_read_b128 touch 16 distinct 16-B bank groups of the 256-B LDS row.
#include "gemm_epilogue.h"

namespace {

constexpr int TN = 256, TM = 128;
constexpr int W_BYTES = TN * 64, A_BYTES = TM * 64;
constexpr int STAGE2 = W_BYTES + A_BYTES;            // 24576
constexpr int NSTAGE = 3;
constexpr int LDS2 = NSTAGE * STAGE2;                // 73728 >= 4 * EPI_REGION (69632)
static_assert(LDS2 >= 4 * EPI_REGION, "epilogue staging must fit the operand stages");

#define BARRIER2() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

__device__ __forceinline__ int swz4(int row_q) { return (4 - row_q) & 3; }   // row_q = (row >> 2) & 3

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm2b_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wid >> 1, wm = wid & 1;
    const int ntn = a.N >> 8;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = lid / ntn, tn = lid - tm * ntn;
    const int m0 = tm * TM, n0 = tn * TN;

    // ---- LDS-DMA sources: a piece is 16 rows x 64 B; wave w moves W pieces 4w..4w+3 and A pieces 2w, 2w+1
    const int prow = lane >> 2;                                      // row inside a piece
    const int schunk = (lane & 3) ^ swz4((prow >> 2) & 3);           // source chunk of this lane's LDS slot
    const bf16_t* srcW = a.W + (size_t)(n0 + wid * 64 + prow) * a.K + schunk * 8;
    const bf16_t* srcA = a.A + (size_t)(m0 + wid * 32 + prow) * a.lda + schunk * 8;
    const size_t pW = (size_t)16 * a.K, pA = (size_t)16 * a.lda;     // one piece further down
    auto dma_step = [&](char* stage, int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(srcW + i * pW + k0), LDS_PTR(stage + (wid * 4 + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(srcA + i * pA + k0), LDS_PTR(stage + W_BYTES + (wid * 2 + i) * 1024), 16, 0, 0);
    };

    // ---- fragment read offsets inside a stage: W frag f (8 of them) at +f*1024, A frag g (4) at +g*1024
    const int frow = lane & 15, fq = lane >> 4;
    const int coff = (fq ^ swz4((frow >> 2) & 3)) << 4;
    const int offW = (wn * 128 + frow) * 64 + coff;
    const int offA = W_BYTES + (wm * 64 + frow) * 64 + coff;

    f32x4 acc[2][4][2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[x][i][y][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 wfa[8], afa[4], wfb[8], afb[4];           // two fragment sets (even / odd steps)
    auto read_frags = [&](const char* stage, bf16x8 (&wf)[8], bf16x8 (&af)[4]) {
#pragma unroll
        for (int f = 0; f < 8; ++f) wf[f] = *(const bf16x8*)(stage + offW + f * 1024);
#pragma unroll
        for (int g = 0; g < 4; ++g) af[g] = *(const bf16x8*)(stage + offA + g * 1024);
    };
    auto mfma_step = [&](const bf16x8 (&wf)[8], const bf16x8 (&af)[4]) {
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int y = 0; y < 2; ++y)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[x][i][y][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[x * 4 + i], af[y * 2 + j], acc[x][i][y][j], 0, 0, 0);
    };
    auto interleave = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);   // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);   // DS read
        }
    };

    const int nt = a.K >> 5;                         // even (K % 64 == 0), >= 2
    // stage rotation: s_cur holds step t, s_nxt step t+1, s_fre = stage of step t-1 (refilled with t+2)
    char* s_cur = smem;
    char* s_nxt = smem + STAGE2;
    char* s_fre = smem + 2 * STAGE2;

    dma_step(s_cur, 0);
    dma_step(s_nxt, 32);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    BARRIER2();
    if (nt > 2) dma_step(s_fre, 64);
    read_frags(s_cur, wfa, afa);

    // one iteration: finish step t (fragments in `wf/af`), fetch the fragments of step t+1 into `wg/ag`
    auto iteration = [&](int t, const bf16x8 (&wf)[8], const bf16x8 (&af)[4], bf16x8 (&wg)[8], bf16x8 (&ag)[4]) {
        // own DMAs of step t+1 landed (step t+2, if any, stays in flight); fragments of step t arrived
        if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        BARRIER2();
        // rotate: the stage of step t is free now (every wave holds its fragments of step t)
        char* s_old = s_cur;
        s_cur = s_nxt; s_nxt = s_fre; s_fre = s_old;            // s_cur: step t+1, s_nxt: step t+2, s_fre: free
        if (t + 3 < nt) dma_step(s_fre, (t + 3) << 5);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(s_cur, wg, ag);                               // (last iteration: stale, in bounds, unused)
        mfma_step(wf, af);
        interleave();
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int t = 0; t < nt; t += 2) {
        iteration(t, wfa, afa, wfb, afb);
        iteration(t + 1, wfb, afb, wfa, afa);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    BARRIER2();                                       // every wave is done with the operand stages

    gemm_epilogue_wave<EPI>(a, acc, smem + wid * EPI_REGION, m0 + wm * 64, n0 + wn * 128, lane);
}

template <int EPI>
hipError_t launch_t(const GemmArgs& a, hipStream_t s) {
    static bool attr_done[64] = {false};            // per device: the attribute belongs to the device's code object
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    bool& attr_set = attr_done[dev_ & 63];
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm2b_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int grid = (a.M / TM) * (a.N / TN);
    hipLaunchKernelGGL(gemm2b_kernel<EPI>, dim3(grid), dim3(256), LDS2, s, a);
    return hipGetLastError();
}

}  // namespace

bool gemm2b_ok(const GemmArgs& a) { return a.M > 0 && (a.M % TM) == 0 && (a.N % TN) == 0 && (a.K & 63) == 0; }

hipError_t launch_gemm2b(const GemmArgs& a, int epi, hipStream_t s) {
    if (!gemm2b_ok(a)) return hipErrorInvalidValue;
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
