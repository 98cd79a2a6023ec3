// Skinny GEMMs for the text path (M = rows*T is a handful of 16-row tiles):
//     out[m][n] = X[m][:] . W[n][:]
// HBM/latency bound weight streaming, not MFMA bound.  Every weight element is read exactly once,
// straight from HBM into VGPRs (no LDS round trip: a weight tile is not shared between waves;
// cdna_hip_programming.md "GEMV / M <= 16 decode weights").  The 16x16x32 bf16 MFMA is used because
// 16 activation rows fit its N dimension exactly, so the dot products cost no VALU.
//
// One wave (= one 64-thread block, to spread over as many CUs as possible) owns one 16-row weight
// tile; all of its weight fragments are requested back-to-back before the first MFMA so that a
// whole matrix is in flight at once (a 768x768 matrix is 1.2 MB = 48 waves x 24 KiB).
//
//   skinny_full  : whole K per wave, fused epilogue (bias [+GELU] -> bf16, or fp32 logits + a
//                  per-tile arg-max partial for the vocabulary head).
//   skinny_head  : the vocabulary head (SK_BIAS_F32 over many tiles): four tiles per workgroup share the activation m-tile
//                  through LDS instead of fetching it once per tile (round 4; same MFMA chain, same bits).
//   skinny_rows3 : the row-prologue form over the fused FFN's 48 slabs: three waves share the slab reduce (round 4).
//   skinny_splitk: K split over blocks (more waves in flight for the K=3072 matrix and for the
//                  N=768 ones that would otherwise use 48 waves); each wave writes its fp32
//                  partial tile to a slab; ln_reduce_kernel (rowops.hip) sums the slabs in a fixed
//                  order, adds bias + residual and applies LayerNorm.  No atomics: results are
//                  bitwise reproducible and independent of the batch size.
//
// Row prologue (LNR, one or two text rows: the single-clip case): the token loop is a chain of dependent launches of
// ~5 us each, and with one row the launch that only normalises it costs as much as one that streams a matrix.  With
// LNR every workgroup of the q|k|v launch computes the row(s) itself (rowln.h: the code of the stand-alone kernels, so
// the same bits) right after requesting its weight fragments -- the two latencies overlap -- and keeps them in LDS;
// 24 KB of slabs per row, read by 144 workgroups through L2, is nothing.  (For more rows the redundant reads and the
// serial row loop cost more than the launch.)
#include "kernels.h"
#include "rowln.h"

namespace {

// weight fragments of one 16-row tile, K32 k-steps.  FP8: e4m3 bytes (8 per lane and k-step, half the stream),
// expanded in registers to bf16 times the row's power-of-two scale (exact: the bf16-stored weight bit for bit).
// Wpk != nullptr: the fragment-major copy (launch_pack_frags): fragment k of this wave = 1 KiB (512 B) contiguous per
// wave instruction, `kofs` / 32 k-steps into tile `row` / 16 -- 3-4 x the per-CU pull rate of the row-major pattern.
template <int K32, bool FP8>
__device__ __forceinline__ void load_wfrags(const void* W, const void* Wpk, const float* wscale, size_t row, int K, int kofs, bf16x8 (&wf)[K32]) {
    if (Wpk) {
        const int lane = threadIdx.x & 63;
        const size_t f0 = ((row >> 4) * (size_t)(K >> 5) + (size_t)(kofs >> 5)) * 64 + lane;      // fragment index of k-step 0
        if (FP8) {
            const uint2* wp = (const uint2*)Wpk + f0;
            const float sc = wscale[row];
            uint2 raw[K32];
#pragma unroll
            for (int k = 0; k < K32; ++k) raw[k] = wp[(size_t)k * 64];
#pragma unroll
            for (int k = 0; k < K32; ++k) wf[k] = fp8x8_to_bf16x8(raw[k], sc);
        } else {
            const bf16x8* wp = (const bf16x8*)Wpk + f0;
#pragma unroll
            for (int k = 0; k < K32; ++k) wf[k] = wp[(size_t)k * 64];
        }
        return;
    }
    if (FP8) {
        const unsigned char* wp = (const unsigned char*)W + row * K + kofs;
        const float sc = wscale[row];
        uint2 raw[K32];
#pragma unroll
        for (int k = 0; k < K32; ++k) raw[k] = *(const uint2*)(wp + k * 32);
#pragma unroll
        for (int k = 0; k < K32; ++k) wf[k] = fp8x8_to_bf16x8(raw[k], sc);
    } else {
        const bf16_t* wp = (const bf16_t*)W + row * K + kofs;
#pragma unroll
        for (int k = 0; k < K32; ++k) wf[k] = *(const bf16x8*)(wp + k * 32);
    }
}

// fp32 logits of one 16 x 16 tile (lane: out[m][n .. n+3]) and the tile's arg-max partial per row (ascending n: first max wins)
__device__ __forceinline__ void logits_epilogue(const SkinnyArgs& a, const float (&y)[4], const int m, const bool mvalid, const int n,
                                                const int fq, const int tile, const int ntiles) {
    if (mvalid && a.out) {
        float* op = (float*)a.out + (size_t)m * a.ldo + n;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (n + r < a.N) op[r] = y[r];
    }
    if (a.amax_val) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (n + r < a.N && (y[r] > best)) { best = y[r]; bi = n + r; }
#pragma unroll
        for (int off = 16; off < 64; off <<= 1) {
            const float v2 = __shfl_xor(best, off);
            const int i2 = __shfl_xor(bi, off);
            if (v2 > best || (v2 == best && i2 < bi)) { best = v2; bi = i2; }
        }
        if (fq == 0 && mvalid) {
            a.amax_val[(size_t)m * ntiles + tile] = best;
            a.amax_idx[(size_t)m * ntiles + tile] = bi;
        }
    }
}

template <int K32, int EPI, bool FP8, bool LNR>
__global__ __launch_bounds__(64) void skinny_full_kernel(SkinnyArgs a) {
    const int lane = threadIdx.x;
    const int frow = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * 16;
    bf16x8 wf[K32];
    load_wfrags<K32, FP8>(a.W, a.Wpk, a.wscale, (size_t)(n0 + frow), a.K, fq * 8, wf);
    __shared__ __attribute__((aligned(16))) bf16_t xrow[LNR ? 2 : 1][LNR ? K32 * 32 : 8];
    if (LNR) {
        constexpr int NV = (K32 * 32 + 255) / 256;
        const SkinnyArgs::RowPrologue& p = a.ln;
        for (int m = 0; m < a.M; ++m) {
            f32x4 v[NV], gv[NV], bev[NV];
            row_load_vec<NV>(gv, p.g, a.K, lane);                // gamma / beta: requested ahead of the row's own loads
            row_load_vec<NV>(bev, p.b, a.K, lane);
            const float s = p.kind == 1 ? row_load_reduce<NV>(v, p.slabs, p.nslab, p.bias, p.resid, a.M, a.K, m, lane)
                                        : row_load_embed<NV>(v, p.ids, p.ld_ids, p.T, p.t0, p.word, p.pos, a.K, p.vocab, m, lane);
            row_layernorm_v<NV>(v, s, lane, a.K, p.eps, gv, bev);
            row_store<NV>(v, lane, a.K, blockIdx.x == 0 ? p.xf + (size_t)m * a.K : nullptr, (bf16_t*)nullptr);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = i * 256 + lane * 4;
                if (c < K32 * 32) {
                    uint2 o;
                    o.x = pack_bf2(v[i][0], v[i][1]);
                    o.y = pack_bf2(v[i][2], v[i][3]);
                    *(uint2*)(&xrow[m][c]) = o;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }

    const int n = n0 + fq * 4;
    float bias[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bias[r] = (a.bias && n + r < a.N) ? a.bias[n + r] : 0.f;

    const int mtiles = (a.M + 15) >> 4;
    // the activation fragments of m-tile mt + 1 are requested before the MFMAs of m-tile mt (32 or more rows: a second
    // m-tile used to add a whole load -> MFMA -> store round trip to the launch)
    bf16x8 xnext[LNR ? 1 : K32];
    auto load_x = [&](int mt) {
        if (LNR) return;
        int m = mt * 16 + frow;
        m = m < a.M ? m : a.M - 1;
        const bf16_t* xp = a.X + (size_t)m * a.ldx + fq * 8;
#pragma unroll
        for (int k = 0; k < K32; ++k) xnext[LNR ? 0 : k] = *(const bf16x8*)(xp + k * 32);
    };
    load_x(0);
    for (int mt = 0; mt < mtiles; ++mt) {
        int m = mt * 16 + frow;
        const bool mvalid = m < a.M;
        m = mvalid ? m : a.M - 1;                               // clamp: padded rows are discarded
        bf16x8 xcur[LNR ? 1 : K32];
        if (!LNR) {
#pragma unroll
            for (int k = 0; k < K32; ++k) xcur[k] = xnext[k];
            if (mt + 1 < mtiles) load_x(mt + 1);
        }
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < K32; ++k) {
            const bf16x8 xf = LNR ? *(const bf16x8*)(&xrow[m][fq * 8 + k * 32]) : xcur[LNR ? 0 : k];
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[k], xf, acc, 0, 0, 0);
        }
        // lane holds out[m][n .. n+3]
        float y[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] = acc[r] + bias[r];
        if (EPI == SK_BIAS_GELU_BF16) {
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = erf_gelu(y[r]);
        } else if (EPI == SK_BIAS_RELU_BF16) {
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = fmaxf(y[r], 0.f);
        }
        if (EPI == SK_BIAS_BF16 || EPI == SK_BIAS_GELU_BF16 || EPI == SK_BIAS_RELU_BF16) {
            if (mvalid) {
                const int orow = (m / a.T) * a.row_stride + a.row_off + (m % a.T);
                bf16_t* op = (bf16_t*)a.out + (size_t)orow * a.ldo + n;
                if (n + 3 < a.N) {
                    uint2 v;
                    v.x = pack_bf2(y[0], y[1]);
                    v.y = pack_bf2(y[2], y[3]);
                    *(uint2*)op = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n + r < a.N) op[r] = f2bf(y[r]);
                }
            }
        } else {  // SK_BIAS_F32: logits (+ arg-max partial of this 16-column tile)
            logits_epilogue(a, y, m, mvalid, n, fq, blockIdx.x, gridDim.x);
        }
    }
}

// ---- vocabulary head with shared activation rows ------------------------------------------------------------------------
// skinny_full with SK_BIAS_F32 runs one single-wave workgroup per 16-column tile and every one of them fetches the same
// 16 x K activation tile -- at K = 768 as many bytes as the weight tile it owns, 1908 times over for the 30522-word
// vocabulary (through L2, but through the CU's one vector-memory port all the same).  Here four waves = four neighbouring
// tiles form a workgroup that fetches the activation m-tile ONCE, into LDS (row pitch + 16 B: the 16 lanes of a
// ds_read_b128 group fall into 16 different 16-byte slots), requested ahead of the weight fragments so that the LDS copy
// waits for it alone (vmcnt counts in order) while the weights stay in flight; m-tile mt + 1 is in flight under m-tile mt
// (two LDS images).  Every output element is the same MFMA chain over the same operands as in skinny_full: same bits.
template <int K32, bool FP8>
__global__ __launch_bounds__(256) void skinny_head_kernel(SkinnyArgs a, const int ntiles) {
    constexpr int PITCH = K32 * 64 + 16, CPR = K32 * 4, NC = (16 * CPR + 255) / 256;     // 16 x CPR 16-byte pieces over 256 threads
    __shared__ __attribute__((aligned(16))) char xs[2][16 * PITCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    const int tile_raw = blockIdx.x * 4 + wave;
    const bool active = tile_raw < ntiles;                       // the last workgroup may hold fewer than four tiles: such a
    const int tile = active ? tile_raw : ntiles - 1;             // wave repeats the last tile and stores nothing
    const int n0 = tile * 16;
    const int mtiles = (a.M + 15) >> 4;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    u32x4 st[NC];
    auto gload = [&](int mt) {
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c0 = (int)threadIdx.x + 256 * i, c = c0 < 16 * CPR ? c0 : 0;      // (K = 576: 4.5 pieces per thread)
            const int row = c / CPR, col = c - row * CPR;
            int m = mt * 16 + row;
            m = m < a.M ? m : a.M - 1;                           // clamp: padded rows are discarded
            st[i] = *(const u32x4*)(a.X + (size_t)m * a.ldx + col * 8);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = (int)threadIdx.x + 256 * i, row = c / CPR, col = c - row * CPR;
            if (c < 16 * CPR) *(u32x4*)(&xs[buf][row * PITCH + col * 16]) = st[i];
        }
    };
    gload(0);
    bf16x8 wf[K32];
    load_wfrags<K32, FP8>(a.W, a.Wpk, a.wscale, (size_t)(n0 + frow), a.K, fq * 8, wf);
    const int n = n0 + fq * 4;
    float bias[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bias[r] = (a.bias && n + r < a.N) ? a.bias[n + r] : 0.f;
    lstore(0);
    __syncthreads();
    for (int mt = 0; mt < mtiles; ++mt) {
        if (mt + 1 < mtiles) gload(mt + 1);
        const char* xb = &xs[mt & 1][frow * PITCH + fq * 16];
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < K32; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[k], *(const bf16x8*)(xb + k * 64), acc, 0, 0, 0);
        const int m = mt * 16 + frow;
        const bool mvalid = m < a.M && active;
        float y[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] = acc[r] + bias[r];
        logits_epilogue(a, y, m < a.M ? m : a.M - 1, mvalid, n, fq, tile, ntiles);
        if (mt + 1 < mtiles) lstore((mt + 1) & 1);              // the image m-tile mt - 1 was read from: every wave is past the
        __syncthreads();                                         // barrier that closed that iteration
    }
}

// ---- one/two-row prologue over many slabs: three waves share the reduce ---------------------------------------------------
// The fused FFN launch (ffn_txt.hip) leaves dec_ffn / 64 = 48 slabs; the single wave of skinny_full<LNR> walks them in three
// dependent round trips (two 8-slab groups each) before it can normalise its row -- in every one of the 144 workgroups, 150
// times per single-clip caption.  Here a workgroup is three waves = three neighbouring 16-column tiles: wave w sums groups
// 2w and 2w + 1 of the row (one round trip, side by side), waves 1 and 2 hand their group sums over through LDS, wave 0 adds
// them in the canonical order (rowln.h: 0 + t0 + t1 + ... ascending; the same adds as row_load_reduce and ln_reduce_kernel),
// adds bias + residual, normalises and leaves the bf16 row in LDS; then every wave runs its own tile's MFMA chain.  Same bits.
template <int K32, int EPI>
__global__ __launch_bounds__(192) void skinny_rows3_kernel(SkinnyArgs a) {
    constexpr int NV = (K32 * 32 + 255) / 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    const int ntiles = (a.N + 15) / 16;
    const int tile_raw = blockIdx.x * 3 + wave;
    const bool active = tile_raw < ntiles;                       // (a wave past the last tile repeats it and stores nothing)
    const int n0 = (active ? tile_raw : ntiles - 1) * 16;
    bf16x8 wf[K32];
    load_wfrags<K32, false>(a.W, a.Wpk, a.wscale, (size_t)(n0 + frow), a.K, fq * 8, wf);
    __shared__ __attribute__((aligned(16))) bf16_t xrow[2][K32 * 32];
    __shared__ __attribute__((aligned(16))) float part[2][2][NV * 256];        // group sums of waves 1, 2
    const SkinnyArgs::RowPrologue& p = a.ln;
    const int ngroups = (p.nslab + 7) >> 3;
    for (int m = 0; m < a.M; ++m) {
        // what wave 0 needs after the merge: requested by every wave ahead of the slabs (a branch around loads is waited for where
        // it ends), in flight behind the trees
        f32x4 bv[NV], rv[NV], gv[NV], bev[NV];
        row_load_vec<NV>(bv, p.bias, a.K, lane);
        row_load_vec<NV>(rv, p.resid + (size_t)m * a.K, a.K, lane);
        row_load_vec<NV>(gv, p.g, a.K, lane);
        row_load_vec<NV>(bev, p.b, a.K, lane);
        f32x4 ta[NV], tb[NV];
        row_slab_tree<NV>(ta, p.slabs, p.nslab, wave * 16, a.M, a.K, m, lane);
        row_slab_tree<NV>(tb, p.slabs, p.nslab, wave * 16 + 8, a.M, a.K, m, lane);
        if (wave > 0) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                *(f32x4*)(&part[wave - 1][0][i * 256 + lane * 4]) = ta[i];
                *(f32x4*)(&part[wave - 1][1][i * 256 + lane * 4]) = tb[i];
            }
        }
        __syncthreads();
        if (wave == 0) {
            f32x4 v[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) { v[i] = f32x4{0.f, 0.f, 0.f, 0.f}; v[i] += ta[i]; }
            if (ngroups > 1) {
#pragma unroll
                for (int i = 0; i < NV; ++i) v[i] += tb[i];
            }
            for (int g = 2; g < ngroups; ++g)
#pragma unroll
                for (int i = 0; i < NV; ++i) v[i] += *(const f32x4*)(&part[(g >> 1) - 1][g & 1][i * 256 + lane * 4]);
            const float s = row_add_bias_resid_v<NV>(v, bv, rv, a.K, lane);
            row_layernorm_v<NV>(v, s, lane, a.K, p.eps, gv, bev);
            row_store<NV>(v, lane, a.K, blockIdx.x == 0 ? p.xf + (size_t)m * a.K : nullptr, (bf16_t*)nullptr);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = i * 256 + lane * 4;
                if (c < K32 * 32) {
                    uint2 o;
                    o.x = pack_bf2(v[i][0], v[i][1]);
                    o.y = pack_bf2(v[i][2], v[i][3]);
                    *(uint2*)(&xrow[m][c]) = o;
                }
            }
        }
        __syncthreads();                                        // the row is in LDS; `part` may be overwritten
    }
    const int n = n0 + fq * 4;
    float bias[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bias[r] = (a.bias && n + r < a.N) ? a.bias[n + r] : 0.f;
    int m = frow;
    const bool mvalid = m < a.M && active;
    m = m < a.M ? m : a.M - 1;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < K32; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[k], *(const bf16x8*)(&xrow[m][fq * 8 + k * 32]), acc, 0, 0, 0);
    float y[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = acc[r] + bias[r];
    if (EPI == SK_BIAS_RELU_BF16) {
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] = fmaxf(y[r], 0.f);
    }
    if (mvalid) {
        const int orow = (m / a.T) * a.row_stride + a.row_off + (m % a.T);
        bf16_t* op = (bf16_t*)a.out + (size_t)orow * a.ldo + n;
        if (n + 3 < a.N) {
            uint2 v;
            v.x = pack_bf2(y[0], y[1]);
            v.y = pack_bf2(y[2], y[3]);
            *(uint2*)op = v;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n + r < a.N) op[r] = f2bf(y[r]);
        }
    }
}

// grid = (n_tiles, ksplit); slab[ks][m][n] fp32 partial sums (no bias)
template <int K32, bool FP8>
__global__ __launch_bounds__(64) void skinny_splitk_kernel(SkinnyArgs a) {
    const int lane = threadIdx.x;
    const int frow = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * 16, ks = blockIdx.y;
    const int kbeg = ks * K32 * 32;
    bf16x8 wf[K32];
    load_wfrags<K32, FP8>(a.W, a.Wpk, a.wscale, (size_t)(n0 + frow), a.K, kbeg + fq * 8, wf);
    const int n = n0 + fq * 4;
    float* slab = (float*)a.out + (size_t)ks * a.M * a.ldo;
    const int mtiles = (a.M + 15) >> 4;
    bf16x8 xnext[K32];                                          // m-tile mt + 1's fragments in flight under m-tile mt (skinny_full)
    auto load_x = [&](int mt) {
        int m = mt * 16 + frow;
        m = m < a.M ? m : a.M - 1;
        const bf16_t* xp = a.X + (size_t)m * a.ldx + kbeg + fq * 8;
#pragma unroll
        for (int k = 0; k < K32; ++k) xnext[k] = *(const bf16x8*)(xp + k * 32);
    };
    load_x(0);
    for (int mt = 0; mt < mtiles; ++mt) {
        int m = mt * 16 + frow;
        const bool mvalid = m < a.M;
        m = mvalid ? m : a.M - 1;
        bf16x8 xcur[K32];
#pragma unroll
        for (int k = 0; k < K32; ++k) xcur[k] = xnext[k];
        if (mt + 1 < mtiles) load_x(mt + 1);
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < K32; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[k], xcur[k], acc, 0, 0, 0);
        if (mvalid) *(f32x4*)(slab + (size_t)m * a.ldo + n) = acc;
    }
}

template <int K32, bool FP8>
hipError_t launch_full(const SkinnyArgs& a, int epi, hipStream_t s) {
    const int grid = (a.N + 15) / 16;
    if (a.ln.kind) return hipErrorInvalidValue;          // row prologue: launch_full_rows
    if constexpr (K32 * 64 * 32 + 512 <= 64 * 1024) {          // two LDS images of 16 rows (K <= 768)
        // the vocabulary head (many tiles over few rows): four tiles per workgroup share the activation rows through LDS
        if (epi == SK_BIAS_F32 && g_head_share && grid >= 4 && a.ldx % 8 == 0 && ((uintptr_t)a.X & 15) == 0) {
            hipLaunchKernelGGL((skinny_head_kernel<K32, FP8>), dim3((grid + 3) / 4), dim3(256), 0, s, a, grid);
            return hipGetLastError();
        }
    }
    switch (epi) {
        case SK_BIAS_BF16: hipLaunchKernelGGL((skinny_full_kernel<K32, SK_BIAS_BF16, FP8, false>), dim3(grid), dim3(64), 0, s, a); break;
        case SK_BIAS_GELU_BF16: hipLaunchKernelGGL((skinny_full_kernel<K32, SK_BIAS_GELU_BF16, FP8, false>), dim3(grid), dim3(64), 0, s, a); break;
        case SK_BIAS_RELU_BF16: hipLaunchKernelGGL((skinny_full_kernel<K32, SK_BIAS_RELU_BF16, FP8, false>), dim3(grid), dim3(64), 0, s, a); break;
        case SK_BIAS_F32: hipLaunchKernelGGL((skinny_full_kernel<K32, SK_BIAS_F32, FP8, false>), dim3(grid), dim3(64), 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// with the row prologue: a bf16-output projection of one or two text rows (GIT q|k|v; the student's q|k|v, cross q, FC1)
template <int K32>
hipError_t launch_full_rows(const SkinnyArgs& a, int epi, hipStream_t s) {
    const SkinnyArgs::RowPrologue& p = a.ln;
    if (!skinny_row_prologue_ok(a.M, a.K, a.wscale != nullptr) || !p.g || !p.b || !p.xf ||
        (p.kind == 1 && (!p.slabs || p.nslab <= 0 || !p.bias || !p.resid || p.resid == p.xf)) ||
        (p.kind == 2 && (!p.ids || p.T <= 0 || !p.word || !p.pos)) || (p.kind != 1 && p.kind != 2))
        return hipErrorInvalidValue;
    const dim3 grid((a.N + 15) / 16);
    // many slabs (the fused FFN's 48): three waves per workgroup share the reduce (skinny_rows3_kernel)
    if (g_rows3 && p.kind == 1 && p.nslab > 16 && p.nslab <= 48) {
        const dim3 grid3((grid.x + 2) / 3);
        if (epi == SK_BIAS_BF16) hipLaunchKernelGGL((skinny_rows3_kernel<K32, SK_BIAS_BF16>), grid3, dim3(192), 0, s, a);
        else if (epi == SK_BIAS_RELU_BF16) hipLaunchKernelGGL((skinny_rows3_kernel<K32, SK_BIAS_RELU_BF16>), grid3, dim3(192), 0, s, a);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
    if (epi == SK_BIAS_BF16) hipLaunchKernelGGL((skinny_full_kernel<K32, SK_BIAS_BF16, false, true>), grid, dim3(64), 0, s, a);
    else if (epi == SK_BIAS_RELU_BF16) hipLaunchKernelGGL((skinny_full_kernel<K32, SK_BIAS_RELU_BF16, false, true>), grid, dim3(64), 0, s, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace

bool skinny_full_ok(int K) {
    const int k32 = K / 32;
    return K % 32 == 0 && (k32 == 2 || k32 == 4 || k32 == 8 || k32 == 18 || k32 == 24 || k32 == 32);
}

bool skinny_row_prologue_ok(int M, int K, bool fp8) { return M >= 1 && M <= 2 && (K == 64 || K == 128 || K == 576 || K == 768) && !fp8; }

hipError_t launch_skinny(const SkinnyArgs& a, int epi, hipStream_t s) {
    if (a.K % 32 || a.M <= 0 || a.T <= 0) return hipErrorInvalidValue;
    if (a.ln.kind)
        return a.K == 64 ? launch_full_rows<2>(a, epi, s) : a.K == 128 ? launch_full_rows<4>(a, epi, s)
             : a.K == 576 ? launch_full_rows<18>(a, epi, s) : launch_full_rows<24>(a, epi, s);
    if (a.wscale) {                                     // e4m3 weights (GIT decoder widths only)
        switch (a.K / 32) {
            case 4: return launch_full<4, true>(a, epi, s);
            case 24: return launch_full<24, true>(a, epi, s);
        }
        return hipErrorInvalidValue;
    }
    switch (a.K / 32) {
        case 2: return launch_full<2, false>(a, epi, s);      // K = 64  (tiny student test config)
        case 4: return launch_full<4, false>(a, epi, s);      // K = 128 (tiny test config)
        case 8: return launch_full<8, false>(a, epi, s);      // K = 256
        case 18: return launch_full<18, false>(a, epi, s);    // K = 576 (student decoder)
        case 24: return launch_full<24, false>(a, epi, s);    // K = 768
        case 32: return launch_full<32, false>(a, epi, s);    // K = 1024
    }
    return hipErrorInvalidValue;
}

// K slabs: 8 (else 16) for the K >= 2048 matrices, else the largest of 4, 3, 2 that leaves a supported per-slab depth
// (768 -> 4 x 6, 3072 -> 8 x 12, 4096 -> 16 x 8, 1024 -> 4 x 8; student decoder: 576 -> 3 x 6)
int skinny_ksplit(int K) {
    auto ok = [&](int ks) { const int k32 = K / (ks * 32); return K % (ks * 32) == 0 && (k32 == 1 || k32 == 2 || k32 == 6 || k32 == 8 || k32 == 12); };
    if (K >= 2048) return ok(8) ? 8 : (ok(16) ? 16 : 0);
    for (int ks = 4; ks >= 1; --ks)
        if (ok(ks)) return ks;
    return 0;
}

hipError_t launch_skinny_splitk(const SkinnyArgs& a, hipStream_t s) {
    const int ks = a.ksplit > 0 ? a.ksplit : skinny_ksplit(a.K);
    if (!ks || a.M <= 0 || a.N % 16 || a.K % (ks * 32)) return hipErrorInvalidValue;
    dim3 grid(a.N / 16, ks);
    if (a.wscale) {
        switch (a.K / ks / 32) {
            case 1: hipLaunchKernelGGL((skinny_splitk_kernel<1, true>), grid, dim3(64), 0, s, a); break;
            case 2: hipLaunchKernelGGL((skinny_splitk_kernel<2, true>), grid, dim3(64), 0, s, a); break;
            case 12: hipLaunchKernelGGL((skinny_splitk_kernel<12, true>), grid, dim3(64), 0, s, a); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    switch (a.K / ks / 32) {
        case 1: hipLaunchKernelGGL((skinny_splitk_kernel<1, false>), grid, dim3(64), 0, s, a); break;
        case 2: hipLaunchKernelGGL((skinny_splitk_kernel<2, false>), grid, dim3(64), 0, s, a); break;
        case 6: hipLaunchKernelGGL((skinny_splitk_kernel<6, false>), grid, dim3(64), 0, s, a); break;
        case 8: hipLaunchKernelGGL((skinny_splitk_kernel<8, false>), grid, dim3(64), 0, s, a); break;
        case 12: hipLaunchKernelGGL((skinny_splitk_kernel<12, false>), grid, dim3(64), 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
