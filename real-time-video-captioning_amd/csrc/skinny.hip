// Skinny GEMM for the text path (M = rows*T, a handful of 16-row tiles): out[m][n] = X[m][:] . W[n][:].
//
// This is HBM/latency bound weight streaming, not MFMA bound: each weight element is read once
// from HBM straight into VGPRs (no LDS round trip: the weight tile is not shared between waves,
// cdna_hip_programming.md "GEMV / M <= 16 decode weights").  The MFMA (16x16x32 bf16) is used only
// because 16 activation rows fit its N dimension exactly, so the dot products cost no VALU.
//
// One block = one 16-row weight tile; its 4 waves split K in quarters and reduce through LDS.
#include "kernels.h"

namespace {

constexpr int MT_MAX = 4;   // m-tiles (16 rows each) handled per pass

template <int EPI>
__global__ __launch_bounds__(256) void skinny_kernel(SkinnyArgs a) {
    __shared__ __attribute__((aligned(16))) float red[4][MT_MAX][64][4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int kq = a.K >> 2;
    const int kbeg = wid * kq;
    const bf16_t* wp = a.W + (size_t)(n0 + frow) * a.K + kbeg + fq * 8;
    const int mtiles = (a.M + 15) >> 4;

    for (int mt0 = 0; mt0 < mtiles; mt0 += MT_MAX) {
        const bf16_t* xp[MT_MAX];
#pragma unroll
        for (int t = 0; t < MT_MAX; ++t) {
            int m = (mt0 + t) * 16 + frow;
            m = m < a.M ? m : a.M - 1;                         // clamp: padded rows are discarded
            xp[t] = a.X + (size_t)m * a.ldx + kbeg + fq * 8;
        }
        f32x4 acc[MT_MAX];
#pragma unroll
        for (int t = 0; t < MT_MAX; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int nact = min(MT_MAX, mtiles - mt0);            // wave-uniform

        for (int k = 0; k < kq; k += 64) {
            // K % 128 == 0  ->  kq % 32 == 0; handle two 32-steps per iteration when available
            const bf16x8 w0 = *(const bf16x8*)(wp + k);
            const bool two = (k + 32) < kq;
            const bf16x8 w1 = two ? *(const bf16x8*)(wp + k + 32) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < MT_MAX; ++t) {
                if (t < nact) {
                    const bf16x8 x0 = *(const bf16x8*)(xp[t] + k);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, x0, acc[t], 0, 0, 0);
                    if (two) {
                        const bf16x8 x1 = *(const bf16x8*)(xp[t] + k + 32);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, x1, acc[t], 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < MT_MAX; ++t) *(f32x4*)red[wid][t][lane] = acc[t];
        __syncthreads();
        // wave w finishes m-tile mt0 + w
        if (wid < nact) {
            f32x4 v = *(const f32x4*)red[0][wid][lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) v += *(const f32x4*)red[w][wid][lane];
            const int m = (mt0 + wid) * 16 + frow;
            const int n = n0 + fq * 4;
            if (m < a.M) {
                const int orow = (m / a.T) * a.row_stride + a.row_off + (m % a.T);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (n + r < a.N) {
                        float y = v[r] + (a.bias ? a.bias[n + r] : 0.f);
                        if (EPI == SK_BIAS_GELU_BF16) y = erf_gelu(y);
                        if (EPI == SK_BIAS_RESID_F32) y += a.resid[(size_t)m * a.ldr + n + r];
                        if (EPI == SK_BIAS_BF16 || EPI == SK_BIAS_GELU_BF16)
                            ((bf16_t*)a.out)[(size_t)orow * a.ldo + n + r] = f2bf(y);
                        else
                            ((float*)a.out)[(size_t)orow * a.ldo + n + r] = y;
                    }
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace

hipError_t launch_skinny(const SkinnyArgs& a, int epi, hipStream_t s) {
    if (a.K % 128 || a.M <= 0 || a.T <= 0) return hipErrorInvalidValue;
    const int grid = (a.N + 15) / 16;
    switch (epi) {
        case SK_BIAS_BF16: hipLaunchKernelGGL(skinny_kernel<SK_BIAS_BF16>, dim3(grid), dim3(256), 0, s, a); break;
        case SK_BIAS_GELU_BF16: hipLaunchKernelGGL(skinny_kernel<SK_BIAS_GELU_BF16>, dim3(grid), dim3(256), 0, s, a); break;
        case SK_BIAS_RESID_F32: hipLaunchKernelGGL(skinny_kernel<SK_BIAS_RESID_F32>, dim3(grid), dim3(256), 0, s, a); break;
        case SK_BIAS_F32: hipLaunchKernelGGL(skinny_kernel<SK_BIAS_F32>, dim3(grid), dim3(256), 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
