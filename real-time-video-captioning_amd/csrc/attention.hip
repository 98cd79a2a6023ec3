// Attention kernels for gfx950 (head_dim = 64 everywhere in GIT: ViT-B/16, ViT-L/14, decoder).
//
// attn_full_kernel: unmasked self-attention over groups of S rows (ViT frames, S = N; decoder
//   image prefix, S = F*N).  Flash-style, one workgroup = 128 queries of one (group, head), 4 waves x
//   32 queries.  Per 64-key tile:
//     S^T = K . Q^T   (v_mfma_f32_32x32x16_bf16, K rows as A operand, Q as B operand) so that a lane
//                     owns ONE query column: softmax max/sum are in-lane + one cross-half shuffle,
//                     and the running rescale of O is a per-lane scalar.
//     O^T += V^T . P^T  P^T is taken straight from the S^T accumulator registers (an accumulator
//                     tile is a valid B operand for a product that sums over its row index,
//                     cdna_hip_programming.md par. 3); V^T fragments come from ds_read_b64_tr_b16.
//   K tile image: [64 keys][128 B], 16-B chunks XOR-swizzled (swz_chunk) -> conflict-free b128 reads.
//   V tile image: [2 d-halves][64 keys][64 B] -> the 4 rows x 64 B of one transposed read cover the
//                 256-B bank row exactly once.
//   Both are filled by LDS-DMA (global_load_lds_dwordx4), double buffered, one barrier per tile.
//
// (The attention of the TEXT rows -- decode steps and teacher-forced prefixes -- is in txtblock.hip.)
#include "kernels.h"
#include <type_traits>
#include <cstdlib>

namespace {

constexpr float kScaleLog2e = 0.125f * 1.4426950408889634f;  // 1/sqrt(64) * log2(e)

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// phase stamps: only in the diagnostic build of tools/probe/attn_probe.hip (the product kernel has none)
#ifdef ATTN_STAMPS
__device__ unsigned long long* g_attn_stamps;
#define ATTN_STAMP(i) do { if ((threadIdx.x & 63) == 0 && g_attn_stamps) g_attn_stamps[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ATTN_STAMP(i) do {} while (0)
#endif

template <int NST>
__global__ __launch_bounds__(256) void attn_full_kernel(const bf16_t* __restrict__ qkv,
                                                        bf16_t* __restrict__ ctx, int S, int H, int G, int nqb) {
    __shared__ __attribute__((aligned(16))) char lds[NST * 16384];   // per stage: K 8 KiB | V 8 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 1-D grid, XCD-aware: workgroups are dealt round-robin over the 8 XCDs (id % 8), each with a
    // private L2.  All q-blocks of one (group, head) read the same K/V, so they are given ids with the
    // same id % 8 (measured before this remap: 249 MB fetched beyond L2 per launch vs 87 MB of qkv).
    // Placement only affects speed.
    int qb, pair;
    {
        const int id = blockIdx.x, npairs = H * G;
        if ((npairs & 7) == 0) {
            const int slot = id >> 3;
            pair = (slot / nqb) * 8 + (id & 7);
            qb = slot - (slot / nqb) * nqb;
        } else {
            pair = id / nqb;
            qb = id - pair * nqb;
        }
    }
    const int head = pair % H, grp = pair / H;
    const int W = H * 64, ld = 3 * W;
    const size_t row0 = (size_t)grp * S;
    const bf16_t* qbase = qkv + row0 * ld + head * 64;
    const bf16_t* kbase = qbase + W;
    const bf16_t* vbase = qbase + 2 * W;

    const int l31 = lane & 31, h2 = lane >> 5;
    const int q0 = qb * 128 + wid * 32;
    const int q = q0 + l31;
    const bool wave_active = q0 < S;                 // wave-uniform

    // Q fragments (B operand: B[k = d][col = q]); rows clamped, junk queries are never stored
    bf16x8 qf[4];
    {
        const bf16_t* qp = qbase + (size_t)min(q, S - 1) * ld + h2 * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
    }

    // ---- staging: wave w moves K pieces 2w,2w+1 (8 keys each) and V pieces 2w,2w+1 (16 keys of one d-half)
    int k_key[2], k_src[2], v_key[2], v_src[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = wid * 2 + i;
        k_key[i] = p * 8 + (lane >> 3);
        k_src[i] = swz_chunk(k_key[i], lane & 7) * 8;
        v_key[i] = (p & 3) * 16 + (lane >> 2);
        v_src[i] = (p >> 2) * 32 + (lane & 3) * 8;
    }
    auto stage = [&](int buf, int key0) {
        char* base = lds + buf * 16384 + wid * 2048;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int kr = min(key0 + k_key[i], S - 1);
            const int vr = min(key0 + v_key[i], S - 1);
            __builtin_amdgcn_global_load_lds(GLB_PTR(kbase + (size_t)kr * ld + k_src[i]), LDS_PTR(base + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLB_PTR(vbase + (size_t)vr * ld + v_src[i]), LDS_PTR(base + 8192 + i * 1024), 16, 0, 0);
        }
    };

    // ---- per-lane read offsets
    // K operand rows: key (32*kt + l31), chunk (2*ks + h2) swizzled by g(row); 32-aligned bases keep g = (l31>>1)&7
    const int kg = (l31 >> 1) & 7;
    int koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = l31 * 128 + (((2 * ks + h2) ^ kg) << 4);
    // V transposed reads: 16-lane group (dgrp = (lane>>4)&1, h2); lane i=lane&15 supplies row i>>2, cols 4*(i&3)
    const int v_off = 8192 + (4 * h2 + ((lane & 15) >> 2)) * 64 + (((lane >> 4) & 1) * 16 + 4 * (lane & 3)) * 2;

    f32x16 o_acc[2];
    o_acc[0] = f32x16{0.f}; o_acc[1] = f32x16{0.f};
#pragma unroll
    for (int r = 0; r < 16; ++r) { o_acc[0][r] = 0.f; o_acc[1][r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    // NST-stage LDS ring, LDS-DMA prefetch NST-1 tiles ahead.  A tile is 4 DMA instructions per
    // wave, so "tile t has landed" is the counted wait vmcnt(4 * tiles issued beyond t); the raw
    // s_barrier then (a) publishes every wave's pieces of tile t and (b) proves every wave is done
    // with tile t-1, whose stage the next DMA overwrites.
    const int ntiles = (S + 63) >> 6;
    ATTN_STAMP(0);
#pragma unroll
    for (int i = 0; i < NST - 1; ++i)
        if (i < ntiles) stage(i, i * 64);
    // entering tile t: its DMA has landed, every wave is past tile t-1, the next prefetch is issued
    auto enter = [&](const int t) {
        const int ahead = min(NST - 2, ntiles - 1 - t);
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + NST - 1 < ntiles) stage((t + NST - 1) % NST, (t + NST - 1) * 64);
        if (t < 6) ATTN_STAMP(1 + 2 * t);          // barrier passed: tile t is in LDS
    };
    // One 64-key tile for this wave's 32 queries; NKT = number of 32-key halves computed.  The ragged last tile
    // (S = 197: 5 of 64 keys, S = 1182: 30 of 64) with no valid key in its second half runs the NKT = 1 form: no
    // K.Q^T MFMAs, exponentials or V^T.P^T steps for that half.  The skipped terms are exact zeros, so the result is
    // bit for bit the masked full tile's.  The choice is wave-uniform.
    auto tile = [&](auto nkt_c, const int t) {
        constexpr int NKT = decltype(nkt_c)::value;
        const char* sb = lds + (t % NST) * 16384;
        const int key0 = t * 64;
        // ---- S^T = K . Q^T : all 8 K fragments are requested before the first MFMA -----------------
        bf16x8 kf[NKT][4];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kf[kt][ks] = *(const bf16x8*)(sb + kt * 4096 + koff[ks]);
        f32x16 s_acc[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_acc[kt][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                s_acc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kt][ks], qf[ks], s_acc[kt], 0, 0, 0);
        }
        // ---- V^T fragments of the first 32 keys: requested now, consumed after the softmax; the second 32 keys'
        //      fragments are requested when the first half's MFMAs have been issued (16 fewer live registers) -----
        bf16x4 vlo[NKT][2][2], vhi[NKT][2][2];
        auto load_v = [&](int kt) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const char* vp = sb + v_off + dt * 4096 + (32 * kt + 16 * s2) * 64;
                    vlo[kt][s2][dt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(vp));
                    vhi[kt][s2][dt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(vp + 8 * 64));
                }
        };
        load_v(0);
        // ---- mask the ragged last tile (wave-uniform branch) ---------------------------------
        if (key0 + 64 > S) {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * h2;
                    if (key >= S) s_acc[kt][r] = -INFINITY;
                }
        }
        // ---- online softmax (lane = query; its 32 keys + the other half-wave's 32) ------------
        float mt = s_acc[0][0];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s_acc[kt][r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = fast_exp2((m_run - m_new) * kScaleLog2e);
        const float mb = -(m_new * kScaleLog2e);
        float psum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = fast_exp2(__builtin_fmaf(s_acc[kt][r], kScaleLog2e, mb));     // spelled out: -ffp-contract=off
                s_acc[kt][r] = p;
                psum += p;
            }
        l_run = __builtin_fmaf(l_run, alpha, psum);
        // O only needs rescaling when some query's running max moved (rare after the first tiles);
        // the branch is wave-uniform and exact (alpha == 1 for every lane otherwise)
        if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { o_acc[0][r] *= alpha; o_acc[1][r] *= alpha; }
        }
        m_run = m_new;

        // ---- O^T += V^T . P^T -----------------------------------------------------------------
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            if (kt == 0 && NKT == 2) load_v(1);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                // 8 probabilities -> one bf16x8 B fragment (4 x v_cvt_pk_bf16_f32)
                typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
                u32x4 pk;
#pragma unroll
                for (int j = 0; j < 4; ++j) pk[j] = pack_bf2(s_acc[kt][8 * s + 2 * j], s_acc[kt][8 * s + 2 * j + 1]);
                const bf16x8 pf = __builtin_bit_cast(bf16x8, pk);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const bf16x8 vf = __builtin_shufflevector(vlo[kt][s][dt], vhi[kt][s][dt], 0, 1, 2, 3, 4, 5, 6, 7);
                    o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o_acc[dt], 0, 0, 0);
                }
            }
        }
    };
    // full-form tiles; a last tile whose second half has no valid key runs the half form, outside the loop
    const bool half_last = S - (ntiles - 1) * 64 <= 32;
    const int nfull = half_last ? ntiles - 1 : ntiles;
    for (int t = 0; t < nfull; ++t) {
        enter(t);
        if (wave_active) tile(std::integral_constant<int, 2>{}, t);
        if (t < 6) ATTN_STAMP(2 + 2 * t);          // tile t consumed (issue side)
    }
    if (half_last) {
        enter(ntiles - 1);
        if (wave_active) tile(std::integral_constant<int, 1>{}, ntiles - 1);
    }
    ATTN_STAMP(13);

    if (wave_active) {
        const float l_tot = l_run + __shfl_xor(l_run, 32);
        const float inv = 1.0f / l_tot;
        if (q < S) {
            bf16_t* op = ctx + (row0 + q) * (size_t)W + head * 64;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int d = dt * 32 + 8 * rq + 4 * h2;
                    uint2 v;
                    v.x = pack_bf2(o_acc[dt][4 * rq + 0] * inv, o_acc[dt][4 * rq + 1] * inv);
                    v.y = pack_bf2(o_acc[dt][4 * rq + 2] * inv, o_acc[dt][4 * rq + 3] * inv);
                    *(uint2*)(op + d) = v;
                }
        }
    }
    ATTN_STAMP(14);
}

}  // namespace

hipError_t launch_attn_full(const bf16_t* qkv, bf16_t* ctx, int G, int S, int H, hipStream_t s) {
    if (G <= 0 || S <= 0 || H <= 0) return hipErrorInvalidValue;
    const int nqb = (S + 127) / 128;
    dim3 grid(nqb * H * G);
    static const int nst = getenv("GITCAP_ATTN_NST") ? atoi(getenv("GITCAP_ATTN_NST")) : 2;
    if (nst == 2) hipLaunchKernelGGL(attn_full_kernel<2>, grid, dim3(256), 0, s, qkv, ctx, S, H, G, nqb);
    else if (nst == 4) hipLaunchKernelGGL(attn_full_kernel<4>, grid, dim3(256), 0, s, qkv, ctx, S, H, G, nqb);
    else hipLaunchKernelGGL(attn_full_kernel<3>, grid, dim3(256), 0, s, qkv, ctx, S, H, G, nqb);
    return hipGetLastError();
}
