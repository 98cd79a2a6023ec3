// bf16 MFMA GEMM for gfx950:  C[m][n] = sum_k A[m][k] * W[n][k]  with fused epilogues.
//
// Both operands are K-contiguous ("B^T" form), which is exactly what the MFMA A/B lane maps
// want: lane l of v_mfma_f32_16x16x32_bf16 holds 8 consecutive k of row (l & 15).
//
// Tiles 128(m) x 128(n) x 64(k) and 64 x 64 x 64, 256 threads = 4 waves as 2(m) x 2(n), each wave a quarter of the
// tile.  The WEIGHT rows are the MFMA "A" operand and the ACTIVATION rows the "B" operand, so an accumulator register
// quad holds 4 consecutive n of one m: the epilogue reads bias / residual and writes the output as 8- or 16-byte
// vectors along the contiguous dimension.
//
// Staging: global_load_lds_dwordx4 (LDS-DMA, 1 KiB per wave-instruction) into a ring of NST stages.  The LDS image is
// lane-linear, so the bank-conflict swizzle is applied to the per-lane SOURCE address and again on the ds_read_b128
// (rule 21 of the CDNA guide).  One barrier per k-tile: tiles t+1 .. t+NST-1 are in flight while tile t is multiplied.
#include "kernels.h"

namespace {

constexpr int BK = 64;

// BM x BN x 64 tile, 4 waves as 2(m) x 2(n), each (BM/2) x (BN/2) = FM x FN MFMA tiles; NST-stage LDS ring with the
// LDS-DMA prefetch NST-1 K-tiles ahead and ONE counted vmcnt wait + one barrier per K-tile.
//   128 x 128, 2 stages (64 KiB, two workgroups per CU): mid-size launches.
//    64 x  64, 3 stages (48 KiB, three per CU): the few-hundred-row launches of the single-clip (webcam) case, where
//   128 x 128 tiles would leave most CUs idle and each K-tile would wait out a whole DMA latency.
// Every variant accumulates an output over ascending k with the same MFMA, so all give the same bits.
template <int EPI, int BM, int BN, int NST>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmArgs a) {
    constexpr int FM = BM / 32, FN = BN / 32;               // MFMA tiles per wave along m / n
    constexpr int PA = BM / 32, PW = BN / 32;               // 8-row DMA pieces per wave per stage (A / W)
    constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + W_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntn = a.N / BN;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = lid / ntn, tn = lid - tm * ntn;
    const int m0 = tm * BM, n0 = tn * BN;
    const int wm = wid >> 1, wn = wid & 1;

    // ---- staging addresses: wave w moves pieces p = PA*w .. (8 rows x 128 B each) ----------------
    const int srow = lane >> 3;                     // row inside a piece
    const int schunk = lane & 7;                    // LDS chunk position inside the row
    const bf16_t* gA[PA];
    const bf16_t* gW[PW];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int row = (wid * PA + i) * 8 + srow;
        gA[i] = a.A + (size_t)(m0 + row) * a.lda + swz_chunk(row, schunk) * 8;
    }
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int row = (wid * PW + i) * 8 + srow;
        gW[i] = a.W + (size_t)(n0 + row) * a.K + swz_chunk(row, schunk) * 8;
    }
    auto stage = [&](int buf, int k0) {
        char* base = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < PA; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(gA[i] + k0), LDS_PTR(base + (wid * PA + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < PW; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(gW[i] + k0), LDS_PTR(base + A_BYTES + (wid * PW + i) * 1024), 16, 0, 0);
    };

    // ---- fragment read offsets ---------------------------------------------------------------
    const int frow = lane & 15, fq = lane >> 4;     // operand row within a 16-tile, k-quarter
    int offW[FN], offA[FM];                         // byte offsets of the row starts in a stage
#pragma unroll
    for (int i = 0; i < FN; ++i) offW[i] = A_BYTES + (wn * (BN / 2) + i * 16 + frow) * 128;
#pragma unroll
    for (int j = 0; j < FM; ++j) offA[j] = (wm * (BM / 2) + j * 16 + frow) * 128;
    const int g = (frow >> 1) & 7;                  // swizzle key of this lane's rows (16-aligned bases)

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // "tile t has landed" = at most (PA + PW) * (tiles issued beyond t) DMA instructions of this wave outstanding; the
    // barrier then publishes every wave's pieces of tile t and proves every wave is done with tile t-1, whose stage the
    // DMA issued right after it overwrites.
    const int nt = a.K / BK;
#pragma unroll
    for (int i = 0; i < NST - 1; ++i)
        if (i < nt) stage(i, i * BK);
    for (int t = 0; t < nt; ++t) {
        const int ahead = min(NST - 2, nt - 1 - t);
        if (NST > 2 && ahead >= 1) {
            if (PA + PW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // a bare s_barrier: __syncthreads() is a fence too, and for the fence the compiler waits with vmcnt(0) -- for the tiles that
        // were just requested, which made every k-tile cost a full LDS-DMA round trip whatever NST (tools/isa_waits.py).  The counted
        // wait above is what orders this wave's DMA; the barrier publishes every wave's pieces and retires the previous tile's reads.
        // (lgkmcnt(0): the previous tile's fragment reads have been consumed by its MFMAs, so the wait is free; it makes the
        // write-after-read order of the LDS-DMA issued below independent of where the compiler places its own waits)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + NST - 1 < nt) stage((t + NST - 1) % NST, (t + NST - 1) * BK);
        const char* sb = smem + (t % NST) * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int coff = ((ks * 4 + fq) ^ g) * 16;
            bf16x8 wf[FN], af[FM];
#pragma unroll
            for (int i = 0; i < FN; ++i) wf[i] = *(const bf16x8*)(sb + offW[i] + coff);
#pragma unroll
            for (int j = 0; j < FM; ++j) af[j] = *(const bf16x8*)(sb + offA[j] + coff);
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: lane holds n = nb + 4*fq + {0..3}, m = mb + frow for each (i, j) ------------
    // Everything the epilogue reads (bias vectors, the residual or position values) is requested before its first store: with
    // LDS-DMA earlier in the kernel the compiler waits for a load with vmcnt(0), which on gfx9 also waits for every store issued
    // before it -- "load, add, store" per fragment was one full round trip per fragment.
    f32x4 bias_v[FN], add_v[(EPI == EPI_BIAS_RESID_F32 || EPI == EPI_PATCH_F32) ? FN : 1][(EPI == EPI_BIAS_RESID_F32 || EPI == EPI_PATCH_F32) ? FM : 1];
#pragma unroll
    for (int i = 0; i < FN; ++i) {
        const int n = n0 + wn * (BN / 2) + i * 16 + fq * 4;
        bias_v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (EPI != EPI_PATCH_F32 && a.bias) bias_v[i] = *(const f32x4*)(a.bias + n);
        if (EPI == EPI_BIAS_RESID_F32 || EPI == EPI_PATCH_F32) {
#pragma unroll
            for (int j = 0; j < FM; ++j) {
                const int m = m0 + wm * (BM / 2) + j * 16 + frow;
                if (EPI == EPI_BIAS_RESID_F32) add_v[i][j] = *(const f32x4*)(a.resid + (size_t)m * a.ldr + n);
                else add_v[i][j] = *(const f32x4*)(a.pos + (size_t)(1 + min(m, a.valid_rows - 1) % a.patches_per_frame) * a.N + n);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < FN; ++i) {
        const int n = n0 + wn * (BN / 2) + i * 16 + fq * 4;
        const f32x4 bias4 = bias_v[i];
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            const int m = m0 + wm * (BM / 2) + j * 16 + frow;
            f32x4 v = acc[i][j] + bias4;
            if (EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_QGELU_BF16 || EPI == EPI_BIAS_GELU_BF16) {
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
                *(uint2*)((bf16_t*)a.out + (size_t)m * a.ldo + n) = o;
            } else if (EPI == EPI_BIAS_RESID_F32) {
                const f32x4 r4 = add_v[EPI == EPI_BIAS_RESID_F32 ? i : 0][EPI == EPI_BIAS_RESID_F32 ? j : 0];
                *(f32x4*)((float*)a.out + (size_t)m * a.ldo + n) = v + r4;
            } else if (EPI == EPI_BIAS_F32) {
                *(f32x4*)((float*)a.out + (size_t)m * a.ldo + n) = v;
            } else {  // EPI_PATCH_F32: m = frame*P + patch  ->  row frame*N + 1 + patch, + pos[1+patch]
                if (m < a.valid_rows) {
                    const int frame = m / a.patches_per_frame;
                    const int patch = m - frame * a.patches_per_frame;
                    const f32x4 p4 = add_v[EPI == EPI_PATCH_F32 ? i : 0][EPI == EPI_PATCH_F32 ? j : 0];
                    const size_t orow = (size_t)frame * a.tokens_per_frame + 1 + patch;
                    *(f32x4*)((float*)a.out + orow * a.ldo + n) = acc[i][j] + p4;
                }
            }
        }
    }
}

template <int EPI, int BM, int BN, int NST>
hipError_t launch_t(const GemmArgs& a, hipStream_t s) {
    constexpr int LDS_BYTES = NST * (BM + BN) * BK * 2;
    static bool attr_done[64] = {false};            // per device: the attribute belongs to the device's code object
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    bool& attr_set = attr_done[dev_ & 63];
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_kernel<EPI, BM, BN, NST>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int grid = (a.M / BM) * (a.N / BN);
    hipLaunchKernelGGL((gemm_bf16_kernel<EPI, BM, BN, NST>), dim3(grid), dim3(256), LDS_BYTES, s, a);
    return hipGetLastError();
}

template <int BM, int BN, int NST>
hipError_t launch_tile(const GemmArgs& a, int epi, hipStream_t s) {
    if (a.M % BM || a.N % BN || a.K % BK || a.M <= 0) return hipErrorInvalidValue;
    switch (epi) {
        case EPI_BIAS_BF16: return launch_t<EPI_BIAS_BF16, BM, BN, NST>(a, s);
        case EPI_BIAS_QGELU_BF16: return launch_t<EPI_BIAS_QGELU_BF16, BM, BN, NST>(a, s);
        case EPI_BIAS_GELU_BF16: return launch_t<EPI_BIAS_GELU_BF16, BM, BN, NST>(a, s);
        case EPI_BIAS_RESID_F32: return launch_t<EPI_BIAS_RESID_F32, BM, BN, NST>(a, s);
        case EPI_BIAS_F32: return launch_t<EPI_BIAS_F32, BM, BN, NST>(a, s);
        case EPI_PATCH_F32: return launch_t<EPI_PATCH_F32, BM, BN, NST>(a, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace

hipError_t launch_gemm(const GemmArgs& a, int epi, hipStream_t s) { return launch_tile<128, 128, 2>(a, epi, s); }
hipError_t launch_gemm64(const GemmArgs& a, int epi, hipStream_t s) { return launch_tile<64, 64, 3>(a, epi, s); }
