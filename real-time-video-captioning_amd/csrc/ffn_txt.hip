// Feed-forward sub-layer of the GIT decoder for TEXT rows (a decode step's handful of rows, or rows x T teacher-forced
// positions), fused over hidden slices:
//
//     h = GELU(x W1^T + b1)  (bf16)          slab[s] = h[:, 64 s .. 64 s + 63] . W2[:, 64 s .. 64 s + 63]^T  (fp32)
//
// One 4-wave workgroup owns 64 hidden units.  Wave w computes 16 of them for the 16 rows of an m-tile (full K = D, the
// 16x16x32 MFMA chain of skinny_full_kernel), rounds to bf16 exactly where the FC1 launch did and leaves them in LDS;
// after one barrier every wave multiplies the 16 x 64 tile of h into its quarter of the D output columns (two k-steps per
// 16-column tile) and writes its fp32 partial.  The dec_ffn / 64 slabs are summed in a fixed order by ln_reduce_kernel
// (rowops.hip) or by the row prologue of the next layer's q|k|v launch (rowln.h), with bias + residual + LayerNorm.
//
// What it replaces: the FC1 + GELU launch (192 single-wave workgroups, h through HBM) and the FC2 split-K launch (384):
// one dependent stage (~5 us) of every decoder layer of every token step.  What makes it possible: the fragment-major
// weight copies (launch_pack_frags) -- a workgroup here pulls 192 KiB of weights by itself, 4 us in the row-major operand
// pattern (47 GB/s per CU), 1.2-2.8 us from the packed copy (tools/probe/pull_probe.hip).  All fragments of a wave
// (48 KiB) are requested before its first MFMA.
//
// No inter-workgroup dependency, no atomics; a row's arithmetic does not depend on M (batch invariant).  Bitwise equal to
// launch_skinny(SK_BIAS_GELU_BF16) followed by launch_skinny_splitk(ksplit = F / 64): the same MFMA sequence per
// element (tests/test_kernels_gpu.py; GITCAP_NO_FFN_FUSE / gitcap_dbg_config(7, 0) selects that pair of launches).
#include "kernels.h"
#include <type_traits>

namespace {

template <int K32, bool FP8>
__global__ __launch_bounds__(256) void ffn_txt_kernel(FfnTxtArgs a) {
    constexpr int D = K32 * 32;
    constexpr int NT = (D / 16) / 4;                        // FC2 output tiles (16 columns) per wave
    static_assert((D / 16) % 4 == 0, "D must be a multiple of 64");
    __shared__ __attribute__((aligned(16))) bf16_t hs[2][16][72];     // h of one m-tile (two buffers), row pitch 144 B
    // The 16 activation rows of an m-tile in LDS (two buffers; global -> registers -> LDS, see gload_x): 16-byte piece c of row m sits in slot 4 K32 m + (c ^ (m & 7)).
    // (The FC1 chain used to fetch its activation fragments from global memory: 8 requested up front, the other 16 one at a time INSIDE
    // the chain -- there are no registers to hold them next to 192 of weight fragments -- i.e. 16 dependent round trips per launch:
    // "L W1 L W1 ..." in tools/isa_waits.py, 6 of the launch's 9 us.  From LDS a fragment is a 64-cycle read that waits for nothing else.)
    __shared__ __attribute__((aligned(16))) char xs[2][16 * K32 * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int s = blockIdx.x;
    const int F32 = a.F >> 5;

    const int mtiles = (a.M + 15) >> 4;
    // m-tile mt -> registers -> LDS buffer mt & 1: 4 K32 16-byte pieces per row, 64 K32 in all, 256 threads; rows past M are clamped.
    // (Round 4 first staged by LDS-DMA.  With DMA in the kernel the compiler stops counting: it waits with vmcnt(0) in front of every LDS
    // access that follows a DMA request -- and on gfx9 vmcnt counts stores -- so every further m-tile cost a DMA round trip plus the drain
    // of the previous tile's 12 slab stores: 32 rows 11.0 us against 7.2 for 16; tools/isa_waits.py: [ W0 | D x 6 W0 | S x 12.  Plain
    // loads are counted: the rows of tile mt + 1 are requested behind the FC1 reads of tile mt and written to LDS behind its stores,
    // waiting for "all but the NT stores".)
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    constexpr int NP = (64 * K32 + 255) / 256;
    u32x4 st[NP];
    auto gload_x = [&](int mt) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int slot0 = i * 256 + (int)threadIdx.x, slot = (64 * K32 % 256 == 0 || slot0 < 64 * K32) ? slot0 : 0;
            const int row = slot / (4 * K32), cs = slot - row * (4 * K32);
            int m = mt * 16 + row;
            m = m < a.M ? m : a.M - 1;
            st[i] = *(const u32x4*)(a.X + (size_t)m * a.ldx + (cs ^ (row & 7)) * 8);
        }
    };
    gload_x(0);                                             // ahead of the weights: the LDS copy below waits for these alone

    // ---- every weight fragment of the wave, requested back to back: 16 hidden rows of W1 (K32 k-steps), then the two
    //      k-steps (hidden 64 s .. + 63) of its NT column tiles of W2
    bf16x8 w1[K32], w2[NT][2];
    {
        const size_t f1 = (((size_t)s * 4 + wave) * K32) * 64 + lane;
        const size_t f2 = (((size_t)wave * NT) * F32 + (size_t)s * 2) * 64 + lane;
        if (FP8) {
            const uint2 *p1 = (const uint2*)a.W1pk + f1, *p2 = (const uint2*)a.W2pk + f2;
            uint2 r1[K32], r2[NT][2];
#pragma unroll
            for (int k = 0; k < K32; ++k) r1[k] = p1[(size_t)k * 64];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                r2[t][0] = p2[(size_t)t * F32 * 64];
                r2[t][1] = p2[(size_t)t * F32 * 64 + 64];
            }
            const float sc1 = a.w1scale[(s * 4 + wave) * 16 + frow];
#pragma unroll
            for (int k = 0; k < K32; ++k) w1[k] = fp8x8_to_bf16x8(r1[k], sc1);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float sc2 = a.w2scale[(wave * NT + t) * 16 + frow];
                w2[t][0] = fp8x8_to_bf16x8(r2[t][0], sc2);
                w2[t][1] = fp8x8_to_bf16x8(r2[t][1], sc2);
            }
        } else {
            const bf16x8 *p1 = (const bf16x8*)a.W1pk + f1, *p2 = (const bf16x8*)a.W2pk + f2;
#pragma unroll
            for (int k = 0; k < K32; ++k) w1[k] = p1[(size_t)k * 64];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                w2[t][0] = p2[(size_t)t * F32 * 64];
                w2[t][1] = p2[(size_t)t * F32 * 64 + 64];
            }
        }
    }
    const int nh = s * 64 + wave * 16 + fq * 4;             // this lane's 4 hidden units
    float bias[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bias[r] = a.b1[nh + r];
    float* slab = a.slabs + (size_t)s * a.M * D;

    auto lstore_x = [&](int mt) {
        char* dst = xs[mt & 1];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int slot = i * 256 + (int)threadIdx.x;
            if (64 * K32 % 256 == 0 || slot < 64 * K32) *(u32x4*)(dst + slot * 16) = st[i];
        }
    };
    __builtin_amdgcn_sched_barrier(0);                      // not in front of the weight requests (the copy waits for the rows)
    lstore_x(0);
    // PREFETCH (launches of several m-tiles): the next tile's rows are requested and copied UNCONDITIONALLY (the last tile re-reads
    // itself): a load under a branch is waited for where the branch ends, i.e. at once
    auto do_tile = [&](const int mt, auto prefetch_c) {
        constexpr bool PREFETCH = decltype(prefetch_c)::value;
        int m = mt * 16 + frow;
        m = m < a.M ? m : a.M - 1;                           // clamp: a padded row repeats row M - 1
        // the tile's rows are in LDS (this wave's pieces: lgkmcnt; everyone's: the barrier); everyone is past the previous tile.
        // Bare s_barrier: __syncthreads() is a fence too, and for the fence the compiler waits with vmcnt(0), i.e. for the slab stores
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // ---- FC1 + GELU: h[m][nh .. nh + 3] -> LDS (bf16) ----
        const char* xb = xs[mt & 1] + frow * (64 * K32);
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < K32; ++k) {
            const bf16x8 xf = *(const bf16x8*)(xb + (((k * 4 + fq) ^ (frow & 7)) << 4));
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[k], xf, acc, 0, 0, 0);
        }
        // the next tile's rows: in flight under GELU, FC2 and the stores
        if (PREFETCH) gload_x(mt + 1 < mtiles ? mt + 1 : mt);
        uint2 hv;
        hv.x = pack_bf2(erf_gelu(acc[0] + bias[0]), erf_gelu(acc[1] + bias[1]));
        hv.y = pack_bf2(erf_gelu(acc[2] + bias[2]), erf_gelu(acc[3] + bias[3]));
        bf16_t (*hb)[72] = hs[mt & 1];
        *(uint2*)(&hb[frow][wave * 16 + fq * 4]) = hv;
        // one barrier per m-tile: buffer (mt + 1) & 1 is rewritten only after every wave has passed this barrier, i.e. after
        // it finished reading that buffer in iteration mt - 1
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // this wave's h values are in LDS
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // ---- FC2 share of the slice: out[m][n] for the wave's NT tiles ----
        const bf16x8 h0 = *(const bf16x8*)(&hb[frow][fq * 8]), h1 = *(const bf16x8*)(&hb[frow][32 + fq * 8]);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
            o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[t][0], h0, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[t][1], h1, o, 0, 0, 0);
            // unconditional: a lane of a padded row read row M - 1's activations (the clamp) and holds row M - 1's values, bit for bit --
            // it stores them to row M - 1 again.  (Under `if (mvalid)` every store sits in its own exec branch, the compiler cannot count
            // them, and the wait for the next tile's rows behind them becomes a drain of the stores.)
            *(f32x4*)(slab + (size_t)m * D + (wave * NT + t) * 16 + fq * 4) = o;
        }
        // xs[(mt + 1) & 1] was last read by the FC1 of tile mt - 1: every wave is past two barriers since
        if (PREFETCH) lstore_x(mt + 1);
    };
    // Pairs: the second tile of a pair is straight-line code behind the first, where the compiler's wait counts are exact.  (Across
    // a loop's back edge its scoreboard merges "the weights may still be arriving" with "12 slab stores are in flight" and the counted
    // waits of the FC1 chain come out as vmcnt(1): from the second iteration on that is a drain of the previous tile's stores.)
    if (mtiles == 1) {
        do_tile(0, std::false_type{});
    } else {
        for (int mt = 0; mt < mtiles; mt += 2) {
            do_tile(mt, std::true_type{});
            if (mt + 1 < mtiles) do_tile(mt + 1, std::true_type{});
        }
    }
}

}  // namespace

bool ffn_txt_ok(int D, int F) { return (D == 128 || D == 768) && F > 0 && F % 64 == 0; }

hipError_t launch_ffn_txt(const FfnTxtArgs& a, hipStream_t s) {
    if (!ffn_txt_ok(a.D, a.F) || a.M <= 0 || !a.X || !a.W1pk || !a.W2pk || !a.b1 || !a.slabs || (a.ldx & 7) ||
        ((a.w1scale == nullptr) != (a.w2scale == nullptr)))
        return hipErrorInvalidValue;
    const dim3 grid(a.F / 64);
    const bool f8 = a.w1scale != nullptr;
    if (a.D == 768) {
        if (f8) hipLaunchKernelGGL((ffn_txt_kernel<24, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((ffn_txt_kernel<24, false>), grid, dim3(256), 0, s, a);
    } else {
        if (f8) hipLaunchKernelGGL((ffn_txt_kernel<4, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((ffn_txt_kernel<4, false>), grid, dim3(256), 0, s, a);
    }
    return hipGetLastError();
}
