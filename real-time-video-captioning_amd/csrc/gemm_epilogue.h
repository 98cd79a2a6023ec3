// LDS-staged epilogue of the 256x256 kernel (gemm256.hip; the experimental tile kernels under tools/experiments/ reuse it):
// a wave holds a 128(n) x 64(m) accumulator block acc[x][i][y][j]
// (n = x*64 + i*16 + 4*(lane>>4) + r, m = y*32 + j*16 + (lane&15)).
//
// The accumulator layout (lane = one m, 4 consecutive n) would scatter 32-byte pieces over 16 rows
// per store instruction.  Instead the wave transposes its two 64(m) x 64(n) half-blocks through a
// private LDS region (the operand stages are dead by then) and writes/reads global memory as full row
// segments: 16 B per lane, 128 B (bf16) or 256 B (fp32) per row.
#pragma once
#include <type_traits>
#include "kernels.h"
#include "ln_canon.h"

constexpr int EPI_REGION = 64 * (64 * 4 + 16);   // per-wave staging (fp32 worst case): 17408 B

// phase stamps: only in the diagnostic build of tools/probe/gemmln_probe.hip (the product kernels have none)
#ifdef LN_STAMPS
__device__ unsigned long long* g_ln_stamps;
#define LN_STAMP(i) do { if (threadIdx.x == 0 && g_ln_stamps) g_ln_stamps[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define LN_STAMP(i) do {} while (0)
#endif

// ep: this wave's EPI_REGION bytes of LDS; mw / nw: first m / n of the wave's block.  SCALED (fp8 kernel): the accumulators
// are first multiplied by ascale * wscale[n].
template <int EPI, bool SCALED = false>
__device__ __forceinline__ void gemm_epilogue_wave(const GemmArgs& a, const f32x4 (&acc)[2][4][2][2], char* ep,
                                                   const int mw, const int nw, const int lane) {
    const int frow = lane & 15, fq = lane >> 4;
    constexpr bool OUT_BF16 = (EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_QGELU_BF16 || EPI == EPI_BIAS_GELU_BF16);
    constexpr bool OUT_F8 = (EPI == EPI_BIAS_QGELU_F8 || EPI == EPI_BIAS_GELU_F8);
    constexpr int ESZ = OUT_F8 ? 1 : OUT_BF16 ? 2 : 4;
    constexpr int RS = 64 * ESZ + 16;                      // padded row stride (bytes)
    constexpr int LPR = 64 * ESZ / 16;                     // lanes per row on the row-wise side (8 or 16)
    constexpr int RPI = 64 / LPR;                          // rows per wave-instruction (8 or 4)
    const int rr = lane / LPR, rc = lane % LPR;            // row-wise role of this lane
    // The bias (and scale) vectors of BOTH half-blocks are requested before the first store: a load between the two halves' stores
    // is waited for with vmcnt(0) (LDS-DMA earlier in the kernel: the compiler counts nothing), and on gfx9 that also waits for
    // every store of the first half to be acknowledged.
    f32x4 bias8[2][4], sc8[2][SCALED ? 4 : 1];
    int nsat = 0;                                          // OUT_F8: codes of valid rows this lane clamped at +-448 (gitcap_fp8_saturations)
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n8 = nw + x * 64 + i * 16 + fq * 4;
            bias8[x][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (EPI != EPI_PATCH_F32 && a.bias) bias8[x][i] = *(const f32x4*)(a.bias + n8);
            if (SCALED) sc8[x][i] = *(const f32x4*)(a.wscale + n8) * a.ascale;       // powers of two: exact
        }
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const int nb = nw + x * 64;                     // first n of this half-block
        // 1) accumulator layout -> LDS [m_local][n_local]
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int nl = i * 16 + fq * 4;
            const f32x4 bias4 = bias8[x][i];
            const f32x4 sc4 = SCALED ? sc8[x][SCALED ? i : 0] : f32x4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int ml = y * 32 + j * 16 + frow;
                    f32x4 v = SCALED ? acc[x][i][y][j] * sc4 + bias4 : acc[x][i][y][j] + bias4;
                    if (EPI == EPI_BIAS_QGELU_BF16 || EPI == EPI_BIAS_QGELU_F8) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = quick_gelu(v[r]);
                    } else if (EPI == EPI_BIAS_GELU_BF16 || EPI == EPI_BIAS_GELU_F8) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = erf_gelu(v[r]);
                    }
                    if (OUT_F8) {
                        const f32x4 c = v * a.out8_inv;
                        if (mw + ml < a.valid_rows) nsat += count_fp8_clamped(c);
                        *(unsigned*)(ep + ml * RS + nl) = pack_fp8x4(c[0], c[1], c[2], c[3]);
                    } else if (OUT_BF16) {
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
        // 2) LDS rows -> global, 16 B per lane along n
        // patch embedding: the position rows of the sub-block, requested before its first store (a load behind a store is
        // waited for with vmcnt(0), one round trip per row; rows past valid_rows are clamped: loaded, not used)
        f32x4 pos_v[EPI == EPI_PATCH_F32 ? 64 / RPI : 1];
        if (EPI == EPI_PATCH_F32) {
#pragma unroll
            for (int it = 0; it < 64 / RPI; ++it) {
                const int mc = min(mw + it * RPI + rr, a.valid_rows - 1);
                pos_v[EPI == EPI_PATCH_F32 ? it : 0] = *(const f32x4*)(a.pos + (size_t)(1 + mc % a.patches_per_frame) * a.N + nb + rc * (16 / ESZ));
            }
        }
#pragma unroll
        for (int it = 0; it < 64 / RPI; ++it) {
            const int ml = it * RPI + rr;
            const int m = mw + ml;
            const int n = nb + rc * (16 / ESZ);
            const uint4 raw = *(const uint4*)(ep + ml * RS + rc * 16);
            if (OUT_F8) {
                *(uint4*)((unsigned char*)a.out + (size_t)m * a.ldo + n) = raw;
            } else if (OUT_BF16) {
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
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                      // reads done before the next half-block overwrites
    }
    if (OUT_F8) report_fp8_clamped(a.f8_sat, nsat, lane);
}

// ---- residual + LayerNorm epilogue of a whole 256x256 tile (EPI_RESID_LN_*; gemm256.hip) ------------------------
// LDS beyond the 8 staging regions: the tile's segment statistics and each wave's row (mean, rstd).
constexpr int LN_STATS_OFF = 8 * EPI_REGION;                 // float2 [256 rows][4 segments]
constexpr int LN_ROWS_OFF = LN_STATS_OFF + 256 * 4 * 8;      // float2 [8 waves][64 rows]
constexpr int LN_LDS_TOTAL = LN_ROWS_OFF + 8 * 64 * 8;       // 151552 B

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// The wave's 128(n) x 64(m) block goes through LDS to the row-wise layout as in gemm_epilogue_wave (lane = 4 consecutive
// n of row it*4 + (lane >> 4)), where 16 lanes hold one 64-column segment of a row: x = acc + bias [+ resid] is written
// (PRE) and stays in registers (they replace the accumulators), the segment statistics go to LDS, then -- the tile's 4
// segments per row together -- to global memory with 16-byte agent-scope stores; one arrival per tile on the row block's
// barrier; then every wave fetches the 128 B of statistics of each of its 64 rows (coalesced), merges the NSEG segments
// in the canonical order and normalises its registers.  The tiles of a row block are consecutive workgroups of ONE XCD's
// dispatch sequence (gemm256.hip hands every XCD whole row blocks), and a tile never waits for anything before it
// publishes, so the wait is bounded by the slowest sibling; an XCD dispatches its workgroups in index order, so a
// waiting tile can only wait for a tile that is resident on the same XCD or that is dispatched as soon as any tile of
// that XCD with all siblings resident retires: no deadlock as long as an XCD can hold N/256 workgroups of this kernel
// at once ("co-residency"); the spin is bounded and fails soft (a host-visible flag, no trap).  Ordering: every wave drains its write-through
// (sc1) statistics stores with s_waitcnt vmcnt(0) before the workgroup barrier that precedes the arrival, the arrival and
// the poll are agent-scope atomics, and the statistics are fetched with sc1 loads issued after the poll succeeded (the
// "write-through store; drain; flag" form of MI355X_MICROARCH.md; agent-scope fences measured 5.4 vs 1.9 us per exchange).
// LN8: the instantiation that also writes the e4m3 copy of the LayerNorm output (a.ln_out8, fp8 compute) and counts the codes it
// clamps; the default instantiation carries none of that code (registers: the epilogue runs at the 256-VGPR limit).
template <bool POST, bool SCALED = false, bool LN8 = false>
__device__ __forceinline__ void gemm_epilogue_tile_ln(const GemmArgs& a, const f32x4 (&acc)[2][4][2][2], char* smem,
                                                      const int m0, const int n0, const int tm, const int tn,
                                                      const int wid, const int wm, const int wn, const int lane) {
    const int frow = lane & 15, fq = lane >> 4;
    constexpr int RS = 64 * 4 + 16;
    char* ep = smem + wid * EPI_REGION;
    float2* st_lds = (float2*)(smem + LN_STATS_OFF);
    float2* row_lds = (float2*)(smem + LN_ROWS_OFF) + wid * 64;
    const int rr = lane >> 4, rc = lane & 15;              // row-wise role: 16 lanes per row, 4 rows per instruction
    const int mw = m0 + wm * 64, nw = n0 + wn * 128;
    // the row block's barrier {count, generation}; the generation is read now (it cannot move before this tile arrives)
    unsigned* bar = a.ln_cnt + 2 * tm;
    unsigned my_gen = 0;
    if (threadIdx.x == 0) my_gen = __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    f32x4 xr[2][16];
    // The residual values of a half-block are requested TOGETHER, into the registers x will live in, before the first of them is
    // used: written as "load, add, store" per row the compiler may not move a load above the previous row's store (out and resid
    // are the same buffer in every caller) and waits with vmcnt(0) -- which on gfx9 counts stores too -- for each of them in turn:
    // 32 round trips per wave, the "HBM-bound" 25 us of this epilogue (ISA: L W0 S L W0 S ...; profiles/r04_gemm_ln_epilogue_waits.txt).
    // Same operations per element, same bits.  (The branch around the loads makes the compiler wait for all of them where the
    // paths meet: exactly the wait wanted here.)
    const bool HAS_RESID = a.resid != nullptr;             // (wave-uniform)
    // (bias / scale vectors of both half-blocks before the first store, as in gemm_epilogue_wave)
    f32x4 bias8[2][4], sc8[2][SCALED ? 4 : 1];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n8 = nw + x * 64 + i * 16 + fq * 4;
            bias8[x][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (a.bias) bias8[x][i] = *(const f32x4*)(a.bias + n8);
            if (SCALED) sc8[x][i] = *(const f32x4*)(a.wscale + n8) * a.ascale;       // fp8 kernel: powers of two, exact
        }
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const int nb = nw + x * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int nl = i * 16 + fq * 4;
            const f32x4 bias4 = bias8[x][i];
            const f32x4 sc4 = SCALED ? sc8[x][SCALED ? i : 0] : f32x4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int ml = y * 32 + j * 16 + frow;
                    *(f32x4*)(ep + ml * RS + nl * 4) = SCALED ? acc[x][i][y][j] * sc4 + bias4 : acc[x][i][y][j] + bias4;
                }
        }
        if (HAS_RESID) {
            __builtin_amdgcn_sched_barrier(0);      // behind the LDS writes above: the accumulators of this half are dead by now
            // wave-uniform base per row group (SGPRs) + one 32-bit lane offset: 16 per-lane pointers would spill
            const unsigned loff = ((unsigned)rr * (unsigned)a.ldr + rc * 4) * 4u;
#pragma unroll
            for (int it = 0; it < 16; ++it)
                xr[x][it] = *(const f32x4*)((const char*)(a.resid + (size_t)(mw + it * 4) * a.ldr + nb) + (size_t)loff);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int ml = it * 4 + rr;
            const int m = mw + ml, n = nb + rc * 4;
            f32x4 v = *(const f32x4*)(ep + ml * RS + rc * 16);
            if (HAS_RESID) v += xr[x][it];
            if (!POST && a.out) *(f32x4*)((float*)a.out + (size_t)m * a.ldo + n) = v;      // x itself: the residual stream
            xr[x][it] = v;
            const float2 st = ln_seg_stats(v);
            if (rc == 0) st_lds[(wm * 64 + ml) * 4 + wn * 2 + x] = st;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    LN_STAMP(2);
    __syncthreads();
    // [M][16] float2 through a buffer resource: 16-byte agent-scope (sc1, aux = 16) stores and loads
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void*)a.ln_stats, 0, a.M * 128, 0x00020000);
    {   // publish the tile's 256 x 4 segment statistics: thread -> (row, two segments) = one 16-byte store
        const int t = threadIdx.x, row = t >> 1, sg = (t & 1) * 2;
        const u32x4 v = *(const u32x4*)(st_lds + row * 4 + sg);
        __builtin_amdgcn_raw_buffer_store_b128(v, srs, ((m0 + row) * 16 + tn * 4 + sg) * 8, 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // written through before the arrival is announced
    __syncthreads();
    LN_STAMP(3);
    // arrival on the row block's barrier {count, generation}: the last of the N/256 tiles resets the count and bumps the
    // generation (nothing else touches the pair until the next launch on the stream); the others wait for the bump
    bool last = false;
    if (threadIdx.x == 0) {
        last = __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(a.N >> 8) - 1u;
        if (last) {
            __hip_atomic_store(bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(bar + 1, my_gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    LN_STAMP(4);
    if (threadIdx.x == 0 && !last) {
        unsigned spins = 0;
        const unsigned limit = a.ln_spin_limit ? a.ln_spin_limit : LN_SPIN_DEFAULT;
        while (__hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == my_gen) {
            __builtin_amdgcn_s_sleep(12);                   // ~0.3 us between polls: 200 spinning tiles must not load the fabric
            // once ANY tile of ANY launch has given up (the host-visible word is up) nobody waits out the full bound again: the
            // launches queued behind the first failure finish at once instead of 30 s each
            if ((spins & 1023u) == 1023u && a.ln_fail && __hip_atomic_load(a.ln_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) break;
            if (++spins > limit) {                          // ~30 s: far beyond any preemption of a sibling.  No trap: tell the
                if (a.ln_fail) __hip_atomic_store(a.ln_fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // host and finish
                break;                                      // (the rows of this block are undefined; host_logic.h: ExchangeHealth)
            }
        }
    }
    __syncthreads();
    LN_STAMP(5);
    {   // the wave's 64 rows x 128 B of statistics: 8 coalesced 16-byte loads per lane into the wave's staging region,
        // then lane -> row lane: merge the row's segments (all tiles) in the canonical order
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = i * 8 + (lane >> 3), ch = lane & 7;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(srs, ((mw + row) * 16 + ch * 2) * 8, 0, 16);
            *(u32x4*)(ep + row * 144 + ch * 16) = v;               // 144-byte row pitch: conflict-free 16-byte reads below
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const float2* src = (const float2*)(ep + lane * 144);
        float mean, rstd;
        if (a.N == 768) ln_merge<12>([&](int q) { return src[q]; }, a.ln_eps, mean, rstd);
        else ln_merge<16>([&](int q) { return src[q]; }, a.ln_eps, mean, rstd);
        row_lds[lane] = float2{mean, rstd};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    LN_STAMP(6);
    // (the optional per-row addend -- the temporal embedding on the last ViT block -- as a second instantiation: a load inside the
    // loop, even one that is never executed, leaves a vmcnt(0) behind every row's stores)
    int nsat = 0;                                          // ln_out8: codes of valid rows this lane clamped at +-448
    auto phase2 = [&](auto has_add_c) {
    constexpr bool HAS_ADD = decltype(has_add_c)::value;
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const int n = nw + x * 64 + rc * 4;
        const f32x4 g4 = *(const f32x4*)(a.ln_g + n), b4 = *(const f32x4*)(a.ln_b + n);
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int ml = it * 4 + rr;
            const float2 mr = row_lds[ml];
            f32x4 y = ln_apply(xr[x][it], mr.x, mr.y, g4, b4);
            const size_t m = (size_t)(mw + ml);
            if (HAS_ADD) y += *(const f32x4*)(a.ln_add + (size_t)(((mw + ml) / a.ln_add_div) % a.ln_add_mod) * a.N + n);   // (one launch per step)
            if (!POST && a.ln_out_f32 && mw + ml < a.valid_rows) *(f32x4*)(a.ln_out_f32 + m * a.ld_ln_f32 + n) = y;
            if (POST) *(f32x4*)((float*)a.out + m * a.ldo + n) = y;
            uint2 o;
            o.x = pack_bf2(y[0], y[1]);
            o.y = pack_bf2(y[2], y[3]);
            *(uint2*)(a.ln_out + m * a.ld_ln + n) = o;
            // fp8 compute: the e4m3 copy the next (fp8) GEMM reads, straight from the fp32 value
            if (LN8 && a.ln_out8) {      // (the run-time test keeps the block out of the stores' schedule: 72 instead of 200 bytes of scratch)
                *(unsigned*)(a.ln_out8 + m * a.ld_ln8 + n) = pack_fp8x4(y[0] * a.ln_out8_inv, y[1] * a.ln_out8_inv, y[2] * a.ln_out8_inv, y[3] * a.ln_out8_inv);
                if (mw + ml < a.valid_rows) nsat += count_fp8_clamped(y * a.ln_out8_inv);
            }
        }
    }
    };
    if (!POST && a.ln_add) phase2(std::true_type{}); else phase2(std::false_type{});
    if (LN8) report_fp8_clamped(a.f8_sat, nsat, lane);
#ifdef LN_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LN_STAMP(7);
#endif
}
