// Canonical LayerNorm arithmetic for the image rows: ONE definition of the statistics that two producers share
// bit for bit -- the stand-alone row kernel (rowops.hip layernorm_kernel) and the 256x256 GEMM's residual epilogue
// that normalises its own rows (gemm_epilogue.h EPI_*_LN) -- so a row's result does not depend on which of them
// ran (tile choice follows the batch size; results must not).
//
// A row of D = 64 * NSEG columns is cut into 64-column SEGMENTS.  A segment lives in one 16-lane DPP row, lane q
// holding the 4 consecutive columns 4q .. 4q+3:
//     group    g[q]  = (x0 + x1) + (x2 + x3)
//     segment  sum   = tree over the 16 lanes: row_half_mirror, row_mirror, quad xor 1, quad xor 2
//                      (all involutions of the lane index -> every lane ends with the same bits)
//     mean_s = sum / 64;   M2_s = the same tree over the groups of (x - mean_s)^2
// Segments are merged left to right with Chan's update (count, mean, M2); rstd = rsqrt(M2 / D + eps);
// y = fma((x - mean) * rstd, gamma, beta).  Every multiply-add below is spelled out (fmaf or separate operations,
// contraction off), so both producers compile to the same operations.
#pragma once
#include "common.h"

template <int CTRL>
__device__ __forceinline__ float ln_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

// sum over the 16 lanes of a DPP row; the same bits in all 16 lanes
__device__ __forceinline__ float ln_seg16_sum(float g) {
#pragma clang fp contract(off)
    const float y = g + ln_dpp<0x141>(g);        // row_half_mirror: q <-> 7 - q inside each 8
    const float z = y + ln_dpp<0x140>(y);        // row_mirror:      q <-> 15 - q
    const float u = z + ln_dpp<0xB1>(z);         // quad_perm [1,0,3,2]
    return u + ln_dpp<0x4E>(u);                  // quad_perm [2,3,0,1]
}

// (mean, M2) of the 64-column segment this lane's 16-lane row holds
__device__ __forceinline__ float2 ln_seg_stats(const f32x4 v) {
#pragma clang fp contract(off)
    const float mean = ln_seg16_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.f / 64.f);
    const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
    const float m2 = ln_seg16_sum(__builtin_fmaf(d1, d1, d0 * d0) + __builtin_fmaf(d3, d3, d2 * d2));
    return float2{mean, m2};
}

// left-to-right merge of NSEG segments of 64 values each -> row mean and 1/sqrt(var + eps)
template <int NSEG, typename F>
__device__ __forceinline__ void ln_merge(F seg, const float eps, float& mean, float& rstd) {
#pragma clang fp contract(off)
    float2 s0 = seg(0);
    float mu = s0.x, m2 = s0.y;
#pragma unroll
    for (int s = 1; s < NSEG; ++s) {
        const float2 t = seg(s);
        const float na = 64.f * (float)s, n = na + 64.f;
        const float d = t.x - mu;
        mu = __builtin_fmaf(d, 64.f / n, mu);
        m2 = __builtin_fmaf(d * d, na * 64.f / n, m2 + t.y);
    }
    mean = mu;
    rstd = rsqrtf(m2 * (1.f / (64.f * (float)NSEG)) + eps);
}

__device__ __forceinline__ f32x4 ln_apply(const f32x4 v, const float mean, const float rstd, const f32x4 g, const f32x4 b) {
#pragma clang fp contract(off)
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = __builtin_fmaf((v[e] - mean) * rstd, g[e], b[e]);
    return y;
}
