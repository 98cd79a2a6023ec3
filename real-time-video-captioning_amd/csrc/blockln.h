// Row reduce + LayerNorm by a whole workgroup, shared by the text attention sub-layer's reducer tail (txtblock.hip) and by the
// FFN launch that runs that reducer in its prologue when there are only one or two rows (ffn_txt.hip): ONE definition, so both give
// the same bits.
#pragma once
#include "common.h"

#ifndef TXT_STAMP
#define TXT_STAMP(i) do {} while (0)
#endif

// The row is held as 16 "virtual waves" of 64 columns (column c = 64 v + lane).  With 16 physical waves a thread holds one
// column (NC = 1); with 8 it holds columns tid and tid + 512, i.e. virtual waves wid and wid + 8 (NC = 2); with 4 (the FFN
// launch) columns tid + 256 c, virtual waves wid + 4 c (NC = 4).  Each virtual
// wave's 64 values are summed by the same butterfly, the 16 sums are added in virtual-wave order: the same bits whatever
// the workgroup size.  `red` is a 16-float LDS array no one else is using.
template <int NC>
__device__ __forceinline__ float block_sum(const float (&v)[NC], float* red, int tid) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const float s = wave_sum(v[c]);
        if ((tid & 63) == 0) red[(tid >> 6) + c * (16 / NC)] = s;
    }
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += red[w];
    return s;
}

// y = LayerNorm over the D values of the row (thread: columns tid + c * 1024 / NC while < D); g, b = the thread's gamma /
// beta (loaded by the caller together with its other loads, so that they are not a round trip of their own behind the two
// block sums)
template <int NC>
__device__ __forceinline__ void block_layernorm(float (&v)[NC], const bool (&act)[NC], int D, float eps, const float (&g)[NC],
                                                const float (&b)[NC], float (*red)[16], int tid) {
#pragma clang fp contract(off)
    TXT_STAMP(8);
    float t[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) t[c] = act[c] ? v[c] : 0.f;
    const float mean = block_sum<NC>(t, red[0], tid) / (float)D;
    TXT_STAMP(9);
    float d[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { d[c] = act[c] ? v[c] - mean : 0.f; t[c] = d[c] * d[c]; }
    const float rstd = rsqrtf(block_sum<NC>(t, red[1], tid) / (float)D + eps);
#pragma unroll
    for (int c = 0; c < NC; ++c) v[c] = act[c] ? __builtin_fmaf(d[c] * rstd, g[c], b[c]) : 0.f;
}

