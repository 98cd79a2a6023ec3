// RETIRED from the product in round 6 (a second full GEMM code path for -1 ... -2 % on synchronous calls only).  Kept as the measured
// experiment it was: docs/LAB_NOTEBOOK.md round 3, profiles/r03_*.  Builds against csrc/ of commit eba5b05 (kernels.h declared its launchers there).
// 256(n) x (32*MT)(m) x 64 tile bf16 MFMA GEMM for gfx950, MT = 7 (224 rows) or 8 (256 rows).
//
//   C[m][n] = sum_k A[m][k] * W[n][k]  (+ fused epilogue), A and W both K-contiguous.
//
// Why a 224-row tile: one workgroup per CU and 256 CUs quantise a launch into whole rounds.  The bench shape (16 clips x
// 6 frames = 18 912 rows) is 74 blocks of 256 rows: 222 / 666 / 888 tiles for N = 768 / 2304 / 3072 = 0.87 / 2.6 / 3.47
// rounds, i.e. 13 % of every GEMM launch is idle CUs.  85 blocks of 224 rows give 255 / 765 / 1020 tiles = 1 / 3 / 4
// FULL rounds of 7/8 the work each.  The host picks the row count per launch (host_logic.h: pick_tile_rows; gitcap.hip:
// launch_gemm_auto -- for synchronous calls only: in the pipeline the CUs a 256-row launch leaves idle are what the token
// loops of the other batches run on, docs/LAB_NOTEBOOK.md par. 6); both give the same bits: every output is accumulated over
// ascending k by the same v_mfma_f32_16x16x32_bf16.  Measured per launch: -3 ... -5 % against gemm256.hip at the bench
// shape (the K loop is 40-60 % of a launch, and this wave layout runs it 1.5-3.5 % slower on equal tiles: the MT = 8
// instantiation exists for that comparison and for the tests, the product launches MT = 7 only).
//
// Wave w = (wm = w >> 2, wn = w & 3) owns the 64(n) x 16*MT(m) block: weight rows are the MFMA A operand, activation rows
// the B operand, so an accumulator quad is 4 consecutive n of one m.  The two wave groups (wm = 0 / 1) share the SIMDs
// pairwise and run the SAME program one barrier apart ("ping-pong", as gemm256.hip): while one wave of a SIMD issues
// LDS-DMA and drains its LDS reads, its partner multiplies.  A K-tile is four compute phases, split by k-step and m:
//
//   C0: ks 0, m-tiles 0-3   (16 MFMA)   + reads: A(m 4.., ks 0), W(ks 1)
//   C1: ks 0, m-tiles 4..   (4*(MT-4))  + reads: A(m 0-3, ks 1)
//   C2: ks 1, m-tiles 0-3   (16)        + reads: A(m 4.., ks 1)
//   C3: ks 1, m-tiles 4..   (4*(MT-4))  + reads: W(ks 0) of tile t+1
//   L0: reads A(m 0-3, ks 0) of this tile; LDS-DMA A-g0(t+1)          L1: LDS-DMA A-g1(t+1)
//   L2: LDS-DMA W-lo(t+2); vmcnt(6) retires W(t+1)                    L3: LDS-DMA W-hi(t+2); vmcnt(4) retires A(t+1)
//
// so a wave holds 8 W fragments + 8 A fragments (64 VGPRs) next to 16*MT accumulator registers.
//
// LDS image per stage (64 KiB): [W-lo | W-hi | A-g0 | A-g1], each 128 rows x 128 B; A-g0 = rows m0 .. m0+127 (group 0
// uses the first 16*MT), A-g1 = rows m0+16*MT .. +127: for MT = 7 each region carries 16 rows nobody reads (the DMA
// pieces stay whole: every wave moves 2 pieces of every region) and the A buffer must be readable 16 rows past M.
// 16-B chunks XOR-swizzled by (row>>1)&7 on the DMA source address and again on the read (common.h).
//
// Hazards, with group 1 one slot behind group 0 (slot = one phase; L0(t) of group 0 is slot 8t):
//   WAR  W(t) is last read in C0(t) (slots 8t+1 / 8t+2), drained by the lgkmcnt(0) of L1(t) (8t+2 / 8t+3): W(t+2) is
//        issued from L2(t) (8t+4 / 8t+5).  A-g(t-1) is last read by its own group in C2(t-1) (8t-3 / 8t-2) and drained in
//        L3(t-1) (8t-2 / 8t-1): A(t+1) is issued from L0(t) (8t / 8t+1) and L1(t).
//   RAW  every wave moves pieces of every region, so a region is readable one barrier after BOTH groups' covering wait:
//        W(t+1): vmcnt(6) in L2(t) (8t+4 / 8t+5) -> first read C3(t) (8t+7 / 8t+8).  A(t+1): vmcnt(4) in L3(t)
//        (8t+6 / 8t+7) -> first read L0(t+1) (8t+8 / 8t+9).
#include "gemm_epilogue.h"
#include <type_traits>
#include "host_logic.h"

namespace {

constexpr int STAGE = 65536, HALF = 16384;
constexpr int LDS_PLAIN = 8 * EPI_REGION;                       // 139264 B >= 2 * STAGE
constexpr int MT_STATS_OFF = 8 * EPI_REGION;                    // float2 [256 rows][4 segments]
constexpr int MT_ROWS_OFF = MT_STATS_OFF + 256 * 4 * 8;         // float2 [8 waves][128 rows]
constexpr int LDS_LN = MT_ROWS_OFF + 8 * 128 * 8;               // 155648 B

#define BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#define WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_mt;

// ---- plain epilogues: the wave's 64(n) x 16*NM(m) sub-block through its LDS region to full row segments -----------
// acc[i][j]: n = nw + i*16 + 4*(lane>>4) + r, m = mw + j*16 + (lane&15)
template <int EPI, int NM>
__device__ __forceinline__ void epilogue_block(const GemmArgs& a, const f32x4 (&acc)[4][8], const int j0, char* ep,
                                               const int mw, const int nw, const int lane, const f32x4 (&bias_v)[4]) {
    const int frow = lane & 15, fq = lane >> 4;
    constexpr bool OUT_BF16 = (EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_QGELU_BF16 || EPI == EPI_BIAS_GELU_BF16);
    constexpr int ESZ = OUT_BF16 ? 2 : 4;
    constexpr int RS = 64 * ESZ + 16;                      // padded row stride (bytes)
    constexpr int LPR = 64 * ESZ / 16;                     // lanes per row on the row-wise side (8 or 16)
    constexpr int RPI = 64 / LPR;                          // rows per wave-instruction (8 or 4)
    const int rr = lane / LPR, rc = lane % LPR;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int nl = i * 16 + fq * 4;
        const f32x4 bias4 = bias_v[i];
#pragma unroll
        for (int j = 0; j < NM; ++j) {
            const int ml = j * 16 + frow;
            f32x4 v = acc[i][j0 + j] + bias4;
            if (EPI == EPI_BIAS_QGELU_BF16) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = quick_gelu(v[r]);
            } else if (EPI == EPI_BIAS_GELU_BF16) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = erf_gelu(v[r]);
            }
            if (OUT_BF16) {
                uint2 o;
                o.x = pack_bf2(v[0], v[1]);
                o.y = pack_bf2(v[2], v[3]);
                *(uint2*)(ep + ml * RS + nl * 2) = o;
            } else {
                *(f32x4*)(ep + ml * RS + nl * 4) = v;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // patch embedding: the position rows of the sub-block, requested before its first store (a load behind a store is
    // waited for with vmcnt(0), one round trip per row; rows past valid_rows are clamped: loaded, not used)
    f32x4 pos_v[EPI == EPI_PATCH_F32 ? 16 * NM / RPI : 1];
    if (EPI == EPI_PATCH_F32) {
#pragma unroll
        for (int it = 0; it < 16 * NM / RPI; ++it) {
            const int mc = min(mw + j0 * 16 + it * RPI + rr, a.valid_rows - 1);
            pos_v[EPI == EPI_PATCH_F32 ? it : 0] = *(const f32x4*)(a.pos + (size_t)(1 + mc % a.patches_per_frame) * a.N + nw + rc * (16 / ESZ));
        }
    }
#pragma unroll
    for (int it = 0; it < 16 * NM / RPI; ++it) {
        const int ml = it * RPI + rr;
        const int m = mw + j0 * 16 + ml;
        const int n = nw + rc * (16 / ESZ);
        const uint4 raw = *(const uint4*)(ep + ml * RS + rc * 16);
        if (OUT_BF16) {
            *(uint4*)((bf16_t*)a.out + (size_t)m * a.ldo + n) = raw;
        } else {
            f32x4 v = __builtin_bit_cast(f32x4, raw);
            if (EPI == EPI_BIAS_RESID_F32) {
                v += *(const f32x4*)(a.resid + (size_t)m * a.ldr + n);
                *(f32x4*)((float*)a.out + (size_t)m * a.ldo + n) = v;
            } else if (EPI == EPI_BIAS_F32) {
                *(f32x4*)((float*)a.out + (size_t)m * a.ldo + n) = v;
            } else {  // EPI_PATCH_F32: m = frame*P + patch -> row frame*N + 1 + patch, + pos[1+patch]
                if (m < a.valid_rows) {
                    const int frame = m / a.patches_per_frame;
                    const int patch = m - frame * a.patches_per_frame;
                    v += pos_v[EPI == EPI_PATCH_F32 ? it : 0];
                    const size_t orow = (size_t)frame * a.tokens_per_frame + 1 + patch;
                    *(f32x4*)((float*)a.out + orow * a.ldo + n) = v;
                }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // reads done before the next sub-block overwrites the region
}

// ---- residual + LayerNorm epilogue (the arithmetic and the exchange protocol of gemm_epilogue.h, this kernel's layout) --
// A wave holds ONE 64-column LayerNorm segment (wn) of its 16*MT rows.  x = acc + bias [+ resid] replaces the accumulators
// in the row-wise layout (16 lanes = the segment of one row), the segment statistics go to LDS, the tile's 4 segments per
// row to global memory (16-byte write-through stores), one arrival per tile on the row block's self-resetting barrier, then
// every wave fetches the 128 B of statistics of each of its rows (sc1 loads), merges the N/64 segments in the canonical
// order and normalises its registers.
template <bool POST, int MT>
__device__ __forceinline__ void epilogue_tile_ln(const GemmArgs& a, const f32x4 (&acc)[4][8], char* smem, const int m0,
                                                 const int n0, const int tm, const int tn, const int wid, const int wm,
                                                 const int wn, const int lane) {
    constexpr int BM = 32 * MT, WR = 16 * MT;              // tile rows, rows per wave
    const int frow = lane & 15, fq = lane >> 4;
    constexpr int RS = 64 * 4 + 16;
    char* ep = smem + wid * EPI_REGION;
    float2* st_lds = (float2*)(smem + MT_STATS_OFF);
    float2* row_lds = (float2*)(smem + MT_ROWS_OFF) + wid * 128;
    const int rr = lane >> 4, rc = lane & 15;              // row-wise role: 16 lanes per row, 4 rows per instruction
    const int mw = m0 + wm * WR, nw = n0 + wn * 64;
    unsigned* bar = a.ln_cnt + 2 * tm;
    unsigned my_gen = 0;
    if (threadIdx.x == 0) my_gen = __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    f32x4 xr[4 * MT];
    f32x4 bias4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bias4[i] = a.bias ? *(const f32x4*)(a.bias + nw + i * 16 + fq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 2; ++c) {                          // m-tiles 0-3, then 4 .. MT-1
        constexpr int NM0 = 4;
        const int j0 = c * NM0, nm = c == 0 ? NM0 : MT - NM0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < nm) *(f32x4*)(ep + (j * 16 + frow) * RS + (i * 16 + fq * 4) * 4) = acc[i][j0 + j] + bias4[i];
        // the residual values of the sub-block together, into the registers x will live in (gemm_epilogue.h: written as "load,
        // add, store" per row every load waits with vmcnt(0) for itself and for the previous row's store)
        if (a.resid) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned loff = ((unsigned)rr * (unsigned)a.ldr + rc * 4) * 4u;
#pragma unroll
            for (int it = 0; it < 16; ++it)
                if (it < 4 * nm) xr[j0 * 4 + it] = *(const f32x4*)((const char*)(a.resid + (size_t)(mw + j0 * 16 + it * 4) * a.ldr + nw) + (size_t)loff);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            if (it < 4 * nm) {
                const int ml = it * 4 + rr;
                const int m = mw + j0 * 16 + ml, n = nw + rc * 4;
                f32x4 v = *(const f32x4*)(ep + ml * RS + rc * 16);
                if (a.resid) v += xr[j0 * 4 + it];
                if (!POST && a.out) *(f32x4*)((float*)a.out + (size_t)m * a.ldo + n) = v;      // x itself: the residual stream
                xr[j0 * 4 + it] = v;
                const float2 st = ln_seg_stats(v);
                if (rc == 0) st_lds[(wm * WR + j0 * 16 + ml) * 4 + wn] = st;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void*)a.ln_stats, 0, a.ln_stats_rows * 128, 0x00020000);
    {   // publish the tile's BM x 4 segment statistics: thread -> (row, two segments) = one 16-byte store
        const int t = threadIdx.x, row = t >> 1, sg = (t & 1) * 2;
        if (row < BM) {
            const u32x4_mt v = *(const u32x4_mt*)(st_lds + row * 4 + sg);
            __builtin_amdgcn_raw_buffer_store_b128(v, srs, ((m0 + row) * 16 + tn * 4 + sg) * 8, 0, 16);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // written through before the arrival is announced
    __syncthreads();
    bool last = false;
    if (threadIdx.x == 0) {
        last = __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(a.N >> 8) - 1u;
        if (last) {
            __hip_atomic_store(bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(bar + 1, my_gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (threadIdx.x == 0 && !last) {
        unsigned spins = 0;
        const unsigned limit = a.ln_spin_limit ? a.ln_spin_limit : LN_SPIN_DEFAULT;
        while (__hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == my_gen) {
            __builtin_amdgcn_s_sleep(12);
            // once ANY tile of ANY launch has given up (the host-visible word is up) nobody waits out the full bound again: the
            // launches queued behind the first failure finish at once instead of 30 s each
            if ((spins & 1023u) == 1023u && a.ln_fail && __hip_atomic_load(a.ln_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) break;
            if (++spins > limit) {                          // ~30 s: fail soft (gemm_epilogue.h; host_logic.h: ExchangeHealth)
                if (a.ln_fail) __hip_atomic_store(a.ln_fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __syncthreads();
    {   // the wave's WR rows x 128 B of statistics: 2*MT coalesced 16-byte loads per lane into the wave's staging region.
        // Row pitch 144 B (conflict-free 16-byte stores and reads) while WR rows fit the region (MT = 7: 16128 B); with 128
        // rows the pitch is 136 B (17408 B = the whole region; 8-byte aligned rows, so 8-byte stores).
        constexpr int PITCH = MT == 8 ? 136 : 144;
        static_assert(16 * MT * PITCH <= EPI_REGION, "statistics staging must fit the wave's region");
#pragma unroll
        for (int i = 0; i < 2 * MT; ++i) {
            const int row = i * 8 + (lane >> 3), ch = lane & 7;
            const u32x4_mt v = __builtin_amdgcn_raw_buffer_load_b128(srs, ((mw + row) * 16 + ch * 2) * 8, 0, 16);
            if (MT == 8) {
                *(uint2*)(ep + row * PITCH + ch * 16) = make_uint2(v[0], v[1]);
                *(uint2*)(ep + row * PITCH + ch * 16 + 8) = make_uint2(v[2], v[3]);
            } else {
                *(u32x4_mt*)(ep + row * PITCH + ch * 16) = v;
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int row = p * 64 + lane;
            if (row < WR) {
                const float2* src = (const float2*)(ep + row * PITCH);
                float mean, rstd;
                if (a.N == 768) ln_merge<12>([&](int q) { return src[q]; }, a.ln_eps, mean, rstd);
                else ln_merge<16>([&](int q) { return src[q]; }, a.ln_eps, mean, rstd);
                row_lds[row] = float2{mean, rstd};
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const int n = nw + rc * 4;
    const f32x4 g4 = *(const f32x4*)(a.ln_g + n), b4 = *(const f32x4*)(a.ln_b + n);
    // (two instantiations for the optional per-row addend: a load inside the loop, even one that is never executed, leaves a
    // vmcnt(0) behind every row's stores)
    auto phase2 = [&](auto has_add_c) {
    constexpr bool HAS_ADD = decltype(has_add_c)::value;
#pragma unroll
    for (int it = 0; it < 4 * MT; ++it) {
        const int ml = it * 4 + rr;
        const float2 mr = row_lds[ml];
        f32x4 y = ln_apply(xr[it], mr.x, mr.y, g4, b4);
        const size_t m = (size_t)(mw + ml);
        if (HAS_ADD) y += *(const f32x4*)(a.ln_add + (size_t)(((mw + ml) / a.ln_add_div) % a.ln_add_mod) * a.N + n);
        if (!POST && a.ln_out_f32 && mw + ml < a.valid_rows) *(f32x4*)(a.ln_out_f32 + m * a.ld_ln_f32 + n) = y;
        if (POST) *(f32x4*)((float*)a.out + m * a.ldo + n) = y;
        uint2 o;
        o.x = pack_bf2(y[0], y[1]);
        o.y = pack_bf2(y[2], y[3]);
        *(uint2*)(a.ln_out + m * a.ld_ln + n) = o;
    }
    };
    if (!POST && a.ln_add) phase2(std::true_type{}); else phase2(std::false_type{});
}

template <int EPI, int MT>
__global__ __launch_bounds__(512, 2) void gemm_mt_kernel(GemmArgs a) {
    static_assert(MT == 7 || MT == 8, "m-tiles per wave");
    constexpr bool LN = (EPI == EPI_RESID_LN_PRE || EPI == EPI_RESID_LN_POST);
    constexpr int BM = 32 * MT, WR = 16 * MT, NB = MT - 4;      // tile rows, rows per wave, m-tiles of the second m group
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2;                      // ping-pong group = m half of the tile
    const int wm = wid >> 2, wn = wid & 3;
    const int ntn = a.N >> 8;
    int tm, tn;
    if (LN && a.ln_rowblock_map) {
        if (!ln_tile_of_block(blockIdx.x, a.M / BM, ntn, &tm, &tn)) return;     // host_logic.h: whole row blocks per XCD
    } else {
        const int lid = xcd_remap(blockIdx.x, gridDim.x);
        tm = lid / ntn; tn = lid - tm * ntn;
    }
    const int m0 = tm * BM, n0 = tn << 8;

    // ---- LDS-DMA source addresses: wave w moves pieces 2w, 2w+1 (8 rows each) of every 128-row region
    const bf16_t* srcW[2];
    const bf16_t* srcA[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wid * 2 + i) * 8 + (lane >> 3);
        const int chunk = swz_chunk(row, lane & 7);
        srcW[i] = a.W + (size_t)(n0 + row) * a.K + chunk * 8;
        srcA[i] = a.A + (size_t)(m0 + row) * a.lda + chunk * 8;
    }
    const size_t hiW = (size_t)128 * a.K, hiA = (size_t)WR * a.lda;
    const int dma_off = wid * 2048;

    // ---- fragment read offsets (bytes inside a stage)
    const int frow = lane & 15, fq = lane >> 4;
    const int g = (frow >> 1) & 7;
    const int offW = (wn >> 1) * HALF + ((wn & 1) * 64 + frow) * 128;      // + i*2048
    const int offA = (2 + wm) * HALF + frow * 128;                         // + j*2048
    const int c0 = ((0 + fq) ^ g) << 4, c1 = ((4 + fq) ^ g) << 4;          // k-step 0 / 1 chunk offsets

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // which = 0 W-lo, 1 W-hi, 2 A-g0, 3 A-g1
    auto dma_region = [&](char* stage, int which, int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bf16_t* src = (which < 2 ? srcW[i] + (which & 1) * hiW : srcA[i] + (which & 1) * hiA) + k0;
            __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(stage + which * HALF + dma_off + i * 1024), 16, 0, 0);
        }
    };

    const int nt = a.K >> 6;
    // prologue: tile 0 (W, A) and W of tile 1; the counted wait leaves W(1) in flight
    dma_region(smem, 0, 0);
    dma_region(smem, 1, 0);
    dma_region(smem, 2, 0);
    dma_region(smem, 3, 0);
    if (nt > 1) {
        dma_region(smem + STAGE, 0, 64);
        dma_region(smem + STAGE, 1, 64);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        WAIT_VM0();
    }
    BARRIER();
    if (grp == 1) BARRIER();                       // group 1 runs one slot behind group 0

    bf16x8 wf0[4], wf1[4], aX[4], aY[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wf0[i] = *(const bf16x8*)(smem + offW + i * 2048 + c0);

    for (int t = 0; t < nt; ++t) {
        const char* sb = smem + (t & 1) * STAGE;
        char* cb = smem + (t & 1) * STAGE;          // stage of tile t == stage of tile t+2
        char* nb = smem + ((t + 1) & 1) * STAGE;
        const bool has1 = (t + 1) < nt, has2 = (t + 2) < nt;
        const int k1 = (t + 1) << 6, k2 = (t + 2) << 6;

        // ---------------- L0: A(m 0-3, ks 0) of this tile; DMA A-g0 of tile t+1 -----------------------------
#pragma unroll
        for (int j = 0; j < 4; ++j) aX[j] = *(const bf16x8*)(sb + offA + j * 2048 + c0);
        if (has1) dma_region(nb, 2, k1);
        WAIT_LGKM0();
        SCHED_FENCE();
        BARRIER();
        // ---------------- C0: ks 0, m 0-3; prefetch A(m 4.., ks 0), W(ks 1) ----------------------------------
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < NB; ++j) aY[j] = *(const bf16x8*)(sb + offA + (4 + j) * 2048 + c0);
#pragma unroll
        for (int i = 0; i < 4; ++i) wf1[i] = *(const bf16x8*)(sb + offW + i * 2048 + c1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[i], aX[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NB + 4; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
        }
        __builtin_amdgcn_s_setprio(0);
        SCHED_FENCE();
        BARRIER();
        // ---------------- L1: DMA A-g1 of tile t+1 -------------------------------------------------------------
        if (has1) dma_region(nb, 3, k1);
        WAIT_LGKM0();
        SCHED_FENCE();
        BARRIER();
        // ---------------- C1: ks 0, m 4..; prefetch A(m 0-3, ks 1) -----------------------------------------------
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 4; ++j) aX[j] = *(const bf16x8*)(sb + offA + j * 2048 + c1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
                acc[i][4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[i], aY[j], acc[i][4 + j], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, NB, 0);  // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
        }
        __builtin_amdgcn_s_setprio(0);
        SCHED_FENCE();
        BARRIER();
        // ---------------- L2: DMA W-lo of tile t+2; retire W of tile t+1 -----------------------------------------
        if (has2) {
            dma_region(cb, 0, k2);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else if (has1) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
        WAIT_LGKM0();
        SCHED_FENCE();
        BARRIER();
        // ---------------- C2: ks 1, m 0-3; prefetch A(m 4.., ks 1) -----------------------------------------------
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < NB; ++j) aY[j] = *(const bf16x8*)(sb + offA + (4 + j) * 2048 + c1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[i], aX[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
        }
        __builtin_amdgcn_s_setprio(0);
        SCHED_FENCE();
        BARRIER();
        // ---------------- L3: DMA W-hi of tile t+2; retire A of tile t+1 -----------------------------------------
        if (has2) {
            dma_region(cb, 1, k2);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            WAIT_VM0();
        }
        WAIT_LGKM0();
        SCHED_FENCE();
        BARRIER();
        // ---------------- C3: ks 1, m 4..; prefetch W(ks 0) of tile t+1 ------------------------------------------
        __builtin_amdgcn_s_setprio(1);
        // (on the last tile this reads the other stage's stale image: in bounds, never used)
#pragma unroll
        for (int i = 0; i < 4; ++i) wf0[i] = *(const bf16x8*)(nb + offW + i * 2048 + c0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
                acc[i][4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[i], aY[j], acc[i][4 + j], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, NB, 0);  // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
        }
        __builtin_amdgcn_s_setprio(0);
        SCHED_FENCE();
        BARRIER();
    }
    if (grp == 0) BARRIER();                       // matches group 1's extra leading barrier
    WAIT_LGKM0();                                  // the last C3's stale prefetch has returned: the stages are dead

    // ---- epilogue through LDS (the operand stages are dead after the last barrier) ----
    if (LN) {
        epilogue_tile_ln<EPI == EPI_RESID_LN_POST, MT>(a, acc, smem, m0, n0, tm, tn, wid, wm, wn, lane);
    } else {
        char* ep = smem + wid * EPI_REGION;
        // the bias vectors once, before the first store: a load between the two sub-blocks' stores is waited for with vmcnt(0)
        // (LDS-DMA earlier in the kernel: the compiler counts nothing), which on gfx9 also waits for every store before it
        f32x4 bias_v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bias_v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (EPI != EPI_PATCH_F32 && a.bias) bias_v[i] = *(const f32x4*)(a.bias + n0 + wn * 64 + i * 16 + (lane >> 4) * 4);
        }
        epilogue_block<EPI, 4>(a, acc, 0, ep, m0 + wm * WR, n0 + wn * 64, lane, bias_v);
        epilogue_block<EPI, NB>(a, acc, 4, ep, m0 + wm * WR, n0 + wn * 64, lane, bias_v);
    }
}

template <int EPI, int MT>
hipError_t launch_t(const GemmArgs& a0, hipStream_t s) {
    static bool attr_done[64] = {false};            // per device: the attribute belongs to the device's code object
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    bool& attr_set = attr_done[dev_ & 63];
    constexpr bool LN = (EPI == EPI_RESID_LN_PRE || EPI == EPI_RESID_LN_POST);
    constexpr int LDS = LN ? LDS_LN : LDS_PLAIN;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_mt_kernel<EPI, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    GemmArgs a = a0;
    const int nrb = a.M / (32 * MT), ntn = a.N >> 8;
    int grid = nrb * ntn;
    if (LN) {
        a.ln_rowblock_map = ln_use_rowblock_map(nrb, ntn, device_cus()) ? 1 : 0;
        if (a.ln_rowblock_map) grid = ln_grid_size(nrb, ntn);
    }
    hipLaunchKernelGGL((gemm_mt_kernel<EPI, MT>), dim3(grid), dim3(512), LDS, s, a);
    return hipGetLastError();
}

template <int MT>
hipError_t launch_mt(const GemmArgs& a, int epi, hipStream_t s) {
    switch (epi) {
        case EPI_BIAS_BF16: return launch_t<EPI_BIAS_BF16, MT>(a, s);
        case EPI_BIAS_QGELU_BF16: return launch_t<EPI_BIAS_QGELU_BF16, MT>(a, s);
        case EPI_BIAS_GELU_BF16: return launch_t<EPI_BIAS_GELU_BF16, MT>(a, s);
        case EPI_BIAS_RESID_F32: return launch_t<EPI_BIAS_RESID_F32, MT>(a, s);
        case EPI_BIAS_F32: return launch_t<EPI_BIAS_F32, MT>(a, s);
        case EPI_PATCH_F32: return launch_t<EPI_PATCH_F32, MT>(a, s);
        case EPI_RESID_LN_PRE: return gemm_mt_ln_ok(a, 32 * MT) && a.resid ? launch_t<EPI_RESID_LN_PRE, MT>(a, s) : hipErrorInvalidValue;
        case EPI_RESID_LN_POST: return gemm_mt_ln_ok(a, 32 * MT) && a.out ? launch_t<EPI_RESID_LN_POST, MT>(a, s) : hipErrorInvalidValue;
    }
    return hipErrorInvalidValue;
}

}  // namespace

bool gemm_mt_ok(const GemmArgs& a, int tile_rows) {
    return (tile_rows == 224 || tile_rows == 256) && a.M > 0 && a.M % tile_rows == 0 && (a.N & 255) == 0 && (a.K & 63) == 0;
}

bool gemm_mt_ln_ok(const GemmArgs& a, int tile_rows) {
    return gemm_mt_ok(a, tile_rows) && (a.N == 768 || a.N == 1024) && a.ln_g && a.ln_b && a.ln_out && a.ln_stats && a.ln_cnt &&
           a.ln_stats_rows >= a.M;
}

// tile_rows = 224: the A buffer must be readable up to row M + 15 (the DMA pieces of the second m group stay whole)
hipError_t launch_gemm_mt(const GemmArgs& a, int epi, int tile_rows, hipStream_t s) {
    if (!gemm_mt_ok(a, tile_rows)) return hipErrorInvalidValue;
    return tile_rows == 224 ? launch_mt<7>(a, epi, s) : launch_mt<8>(a, epi, s);
}
