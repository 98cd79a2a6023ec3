// Row LayerNorm of the text rows, shared by the stand-alone row kernels (rowops.hip) and by the q|k|v projection that
// computes its own input rows when there are only one or two of them (skinny.hip, "row prologue"): ONE wave per row, lane
// holds NV float4 at columns 256 i + 4 lane, two-pass (mean, then centred variance) in registers, fp32 throughout.
// The same inline code wherever a row is normalised, and every multiply-add spelled out (contraction off), so the result does
// not depend on which kernel ran it nor on what the compiler would fuse in that kernel's context.
#pragma once
#include "common.h"

// this lane's NV float4 of a D-vector (columns 256 i + 4 lane).  Unconditional (a column past D reads column 0 and is not used):
// a load under a predicate is waited for where the paths meet, i.e. at once -- callers request what a row needs as early as
// they can, so that it is in flight behind the reductions instead of being a round trip of its own after them
template <int NV>
__device__ __forceinline__ void row_load_vec(f32x4 (&o)[NV], const float* __restrict__ p, const int D, const int lane) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 256 + lane * 4;
        o[i] = *(const f32x4*)(p + (c < D ? c : 0));
    }
}

// v (this lane's values of the row, zero beyond D) and their lane-sum s -> v = LayerNorm(row) * gamma + beta (g, b: row_load_vec)
template <int NV>
__device__ __forceinline__ void row_layernorm_v(f32x4 (&v)[NV], const float s, const int lane, const int D, const float eps,
                                                const f32x4 (&gv)[NV], const f32x4 (&bv)[NV]) {
#pragma clang fp contract(off)
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 256 + lane * 4;
        if (c < D) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q = __builtin_fmaf(d, d, q); }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 256 + lane * 4;
        if (c < D) {
            const f32x4 g = gv[i], b = bv[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i][e] = __builtin_fmaf((v[i][e] - mean) * rstd, g[e], b[e]);
        }
    }
}
template <int NV>
__device__ __forceinline__ void row_layernorm(f32x4 (&v)[NV], const float s, const int lane, const int D, const float eps,
                                              const float* __restrict__ gamma, const float* __restrict__ beta) {
    f32x4 gv[NV], bv[NV];                           // requested before the two wave reductions, used after them
    row_load_vec<NV>(gv, gamma, D, lane);
    row_load_vec<NV>(bv, beta, D, lane);
    row_layernorm_v<NV>(v, s, lane, D, eps, gv, bv);
}

// t = the fixed-order sum of slabs k0 .. k0+7 of row m (missing slabs count as zero): all 8 x NV vectors are requested
// before the first add (no serial latency chain)
template <int NV>
__device__ __forceinline__ void row_slab_tree(f32x4 (&t)[NV], const float* __restrict__ slabs, const int nslab, const int k0,
                                              const int M, const int D, const int m, const int lane) {
    f32x4 p[8][NV];
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = i * 256 + lane * 4;
            p[k][i] = (k0 + k < nslab && c < D) ? *(const f32x4*)(slabs + ((size_t)(k0 + k) * M + m) * D + c)
                                                : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
    for (int i = 0; i < NV; ++i)
        t[i] = ((p[0][i] + p[1][i]) + (p[2][i] + p[3][i])) + ((p[4][i] + p[5][i]) + (p[6][i] + p[7][i]));
}

// v += bias + resid[m]; returns the lane-sum (bv, rv: row_load_vec of bias and of resid + m D)
template <int NV>
__device__ __forceinline__ float row_add_bias_resid_v(f32x4 (&v)[NV], const f32x4 (&bv)[NV], const f32x4 (&rv)[NV], const int D, const int lane) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 256 + lane * 4;
        if (c < D) {
            v[i] += bv[i] + rv[i];
            s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        }
    }
    return s;
}
template <int NV>
__device__ __forceinline__ float row_add_bias_resid(f32x4 (&v)[NV], const float* __restrict__ bias, const float* __restrict__ resid,
                                                    const int D, const int m, const int lane) {
    f32x4 bv[NV], rv[NV];
    row_load_vec<NV>(bv, bias, D, lane);
    row_load_vec<NV>(rv, resid + (size_t)m * D, D, lane);
    return row_add_bias_resid_v<NV>(v, bv, rv, D, lane);
}

// v = sum_k slab[k][m] + bias + resid[m]; returns the lane-sum.  THE summation order of the text rows' split-K reduce:
// groups of 8 slabs by the tree above, the groups added in ascending order -- whoever computes it (one wave here; one wave
// per group in ln_reduce_kernel) gets the same bits, independent of launch geometry and timing.
template <int NV>
__device__ __forceinline__ float row_load_reduce(f32x4 (&v)[NV], const float* __restrict__ slabs, const int nslab,
                                                 const float* __restrict__ bias, const float* __restrict__ resid,
                                                 const int M, const int D, const int m, const int lane) {
    f32x4 bv[NV], rv[NV];                           // bias and the residual row: requested before the slabs, used after them
    row_load_vec<NV>(bv, bias, D, lane);
    row_load_vec<NV>(rv, resid + (size_t)m * D, D, lane);
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    int k0 = 0;
    // two groups per round trip while there are that many (48 slabs: 3 round trips instead of 6); the adds keep their order
    for (; k0 + 8 < nslab; k0 += 16) {
        f32x4 ta[NV], tb[NV];
        row_slab_tree<NV>(ta, slabs, nslab, k0, M, D, m, lane);
        row_slab_tree<NV>(tb, slabs, nslab, k0 + 8, M, D, m, lane);
#pragma unroll
        for (int i = 0; i < NV; ++i) { v[i] += ta[i]; v[i] += tb[i]; }
    }
    for (; k0 < nslab; k0 += 8) {
        f32x4 t[NV];
        row_slab_tree<NV>(t, slabs, nslab, k0, M, D, m, lane);
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] += t[i];
    }
    return row_add_bias_resid_v<NV>(v, bv, rv, D, lane);
}

// v = word[tok] + pos[position]; returns the lane-sum
template <int NV>
__device__ __forceinline__ float row_load_embed_tok(f32x4 (&v)[NV], int64_t tok, const int position, const float* __restrict__ word,
                                                    const float* __restrict__ pos, const int D, const int vocab, const int lane) {
    tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);          // never index outside the table
    const float* wr = word + (size_t)tok * D;
    const float* pr = pos + (size_t)position * D;
    f32x4 wv[NV], pv[NV];                           // both rows requested before the first add
    row_load_vec<NV>(wv, wr, D, lane);
    row_load_vec<NV>(pv, pr, D, lane);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 256 + lane * 4;
        if (c < D) {
            v[i] = wv[i] + pv[i];
            s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        } else v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    return s;
}

// v = word[ids[r*ld_ids + j]] + pos[t0 + j], m = r*T + j; returns the lane-sum
template <int NV>
__device__ __forceinline__ float row_load_embed(f32x4 (&v)[NV], const int64_t* __restrict__ ids, const int ld_ids,
                                                const int T, const int t0, const float* __restrict__ word,
                                                const float* __restrict__ pos, const int D, const int vocab, const int m,
                                                const int lane) {
    const int r = m / T, j = m - r * T;
    return row_load_embed_tok<NV>(v, ids[(size_t)r * ld_ids + j], t0 + j, word, pos, D, vocab, lane);
}

// fp32 and/or bf16 copies of the row (either pointer may be null)
template <int NV>
__device__ __forceinline__ void row_store(const f32x4 (&y)[NV], const int lane, const int D, float* xf_row, bf16_t* xb_row) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 256 + lane * 4;
        if (c < D) {
            if (xf_row) *(f32x4*)(xf_row + c) = y[i];
            if (xb_row) {
                uint2 o;
                o.x = pack_bf2(y[i][0], y[i][1]);
                o.y = pack_bf2(y[i][2], y[i][3]);
                *(uint2*)(xb_row + c) = o;
            }
        }
    }
}
