// Attention sub-layer of the GIT decoder for TEXT rows (one decode step, or T teacher-forced positions), fused from the
// attention itself to the LayerNorm that closes the sub-layer.
//
// One 16-wave workgroup = one unit (text row m = (r, j), head):
//   1. attention over the image keys of the row's clip + the text keys 0..t of the row.  q, k, v of the text rows were
//      written to the text K/V cache by the q|k|v projection launch (skinny.hip).  K/V are streamed from HBM: the wave's
//      first 32-key group by LDS-DMA (no registers while in flight), the others straight to VGPRs, 8 lanes per key,
//      per-group online softmax; wave w takes the 32-key groups w, w+16, ... so the summation order depends on the key
//      count only (batch invariant, bitwise)
//   2. this head's share of the output dense: ctx_h[64] . Wo[:, 64h:64h+64]^T -> part[m][head][D] (fp32, write-through);
//      the 6 KiB of weight fragments a wave needs arrive by LDS-DMA under the attention
//   3. the LAST of the H units of a row to arrive (ticket counter; no unit ever waits for another) sums the H partials
//      in head order, adds bias + residual and applies LayerNorm -> x1 (fp32) and bf16(x1) for the FC1 launch.
// This replaces three launches of the first version (attention, split-K output dense, reduce + LayerNorm) by one.
// Hand-off in 3 follows cdna_hip_programming.md Guideline 16 / MI355X_MICROARCH.md "Valid forms": every payload store
// is an agent-scope (sc1) store, every storing wave drains vmcnt before the workgroup barrier, ONE lane adds the ticket,
// the reducer's loads are agent-scope (sc1) loads issued after the barrier its ticket lane joined.  No spin anywhere.
//
// Small batches (the webcam case: one clip = 12 units on 256 CUs, each streaming 300 KB of K/V through ONE CU: 16 us per
// launch, a third of the single-clip caption): KEY SPLIT.  The 16 waves of a unit are dealt to S = 2 .. 16 workgroups of
// WPB = 16 / S waves; every wave does exactly the work it does in the 16-wave workgroup (same key groups, same order), the
// per-wave partial states (m, l, o[64]) go to global memory (write-through), and the LAST of a unit's S workgroups to
// arrive (ticket, nobody waits) merges the 16 partials in wave order -- the arithmetic of the one-workgroup form, so
// the result is bit for bit the same whatever S is -- and carries on with the output dense, the row ticket and the
// LayerNorm.  S is chosen from the row count (units x S <= 256 workgroups), so batch invariance is preserved.
//
// Why the q|k|v projection is NOT in here (measured, tools/probe/txtblock_probe.hip): a CU pulls weight fragments at
// ~30 GB/s whatever serves them (HBM, Infinity Cache or L2); 295 KB of q|k|v weights per (row, head) unit cost 10 us in
// front of the attention, against 6 us for a launch of single-wave tiles that reads every weight byte once.
#include "kernels.h"
#include "host_logic.h"

namespace {

constexpr float kScaleLog2e = 0.125f * 1.4426950408889634f;  // 1/sqrt(64) * log2(e)
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }

// Every multiply-add of the softmax arithmetic is spelled out (contraction off): the kernel is instantiated for several
// workgroup sizes (key split), and the instantiations must round identically -- a contraction the compiler chose for one of
// them and not for another moved a context value across a bf16 rounding boundary once in ~100 launches.
struct Part { float m, l; float o[8]; };
__device__ __forceinline__ void merge(Part& a, float m2, float l2, const float* o2) {
#pragma clang fp contract(off)
    const float M = fmaxf(a.m, m2);
    const float s1 = (a.m == -INFINITY) ? 0.f : ex2(a.m - M);
    const float s2 = (m2 == -INFINITY) ? 0.f : ex2(m2 - M);
    a.l = __builtin_fmaf(a.l, s1, l2 * s2);
#pragma unroll
    for (int d = 0; d < 8; ++d) a.o[d] = __builtin_fmaf(a.o[d], s1, o2[d] * s2);
    a.m = M;
}

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// phase stamps: only in the diagnostic build of tools/probe/txtblock_probe.hip (the product kernel has none)
#ifdef TXT_STAMPS
__device__ unsigned long long* g_txt_stamps;
#define TXT_STAMP(i) do { if (threadIdx.x == 0 && g_txt_stamps) g_txt_stamps[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TXT_STAMP(i) do {} while (0)
#endif

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_tb;

template <int K32, bool FP8, int WPB>
__global__ __launch_bounds__(64 * WPB) void txt_block_kernel(TxtBlockArgs a) {
    constexpr int D = K32 * 32;
    constexpr int S = 16 / WPB;                    // workgroups per (row, head) unit
    constexpr bool SPLIT = WPB < 16;
    __shared__ __attribute__((aligned(16))) bf16_t ctxs[64];
    __shared__ float red[2][16];
    __shared__ __attribute__((aligned(16))) float wsm[16][8][12];      // per wave and 8-column group: m, l, o[8] (+ 2 pad)
    __shared__ int last_flag;
    // per wave 8 KiB: the wave's first 32-key group (K 4 KiB | V 4 KiB) by LDS-DMA, then reused for the wave's
    // output-dense weight fragments
    __shared__ __attribute__((aligned(16))) char kvpre[WPB * 8192];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wl = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave inside this workgroup
    const int M = a.rows * a.T, H = a.H;
    // unit -> (m, head).  H == 12: blocks that share an XCD (blockIdx % 8 equal) take whole heads (8 heads on 8 XCDs,
    // heads 8..11 as half-heads of Mh | M - Mh rows), so a head's output-dense slice stays in ONE L2 and the beams of a
    // clip (same image K/V) meet there too.  Placement only affects speed.
    int m, head, split = 0;
    if (SPLIT) {
        split = blockIdx.x % S;
        const int unit = blockIdx.x / S;
        m = unit / H; head = unit - m * H;
        if (m >= M) return;
    } else if (H == 12) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        if (slot < M) { m = slot; head = xcd; }
        else {
            const int s2 = slot - M;
            head = 8 + (xcd >> 1);
            m = (xcd & 1) ? a.Mh + s2 : s2;
            if (m >= ((xcd & 1) ? M : a.Mh)) return;
        }
    } else {
        m = blockIdx.x / H; head = blockIdx.x - m * H;
        if (m >= M) return;
    }
    const int wid = split * WPB + wl;              // this wave's place among the unit's 16: decides its key groups
    const int r = m / a.T, j = m - r * a.T;
    const int clip = r / a.beams;
    const int ld = 3 * D;
    const int tq = a.t0 + j;
    const int frow = lane & 15, fq = lane >> 4;
    TXT_STAMP(0);

    const int Lk = a.S_img + tq + 1;
    const int sub = lane & 7, kk = lane >> 3;
    const bf16_t* img = a.kv_img + (size_t)clip * a.S_img * ld + D + head * 64 + sub * 8;
    const bf16_t* txt = a.kv_txt + (size_t)r * a.Tmax * ld + D + head * 64 + sub * 8;
    char* mypre = kvpre + wl * 8192;
    // group 0 of this wave (keys 32 wid .. +31) -> LDS, lane-linear (lane = kk*8 + sub, one 1-KiB piece per 8 keys)
    auto dma_group0 = [&]() {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int key = wid * 32 + u * 8 + kk;
            key = key < Lk ? key : 0;
            const bf16_t* kp = key < a.S_img ? img + (size_t)key * ld : txt + (size_t)(key - a.S_img) * ld;
            __builtin_amdgcn_global_load_lds(GLB_PTR(kp), LDS_PTR(mypre + u * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLB_PTR(kp + D), LDS_PTR(mypre + 4096 + u * 1024), 16, 0, 0);
        }
    };
    if (wid * 32 < Lk) dma_group0();

    TXT_STAMP(2);
    // ---- 1: attention of (row, position tq, head) ------------------------------------------------------------------
    float qv[8];
    {
        const bf16x8 q8 = *(const bf16x8*)(a.kv_txt + ((size_t)r * a.Tmax + tq) * ld + head * 64 + sub * 8);
#pragma unroll
        for (int d = 0; d < 8; ++d) qv[d] = bf2f((bf16_t)q8[d]) * kScaleLog2e;
    }
    Part st;
    st.m = -INFINITY; st.l = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) st.o[d] = 0.f;

    // Each wave owns only 2-3 groups of 32 keys (wid, wid+16, ...): group 0 is already in LDS, the loads of group
    // i+1 are issued before group i is reduced (two register sets, static names).
    auto load_group = [&](int g0, bf16x8* kf, bf16x8* vf, bool* valid) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int key = g0 + u * 8 + kk;
            valid[u] = key < Lk;
            key = valid[u] ? key : 0;
            const bf16_t* kp = key < a.S_img ? img + (size_t)key * ld : txt + (size_t)(key - a.S_img) * ld;
            kf[u] = *(const bf16x8*)kp;
            vf[u] = *(const bf16x8*)(kp + D);
        }
    };
    auto reduce_group = [&](const bf16x8* kf, const bf16x8* vf, const bool* valid) {
#pragma clang fp contract(off)
        float sc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < 8; ++d) s = __builtin_fmaf(qv[d], bf2f((bf16_t)kf[u][d]), s);
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            s += __shfl_xor(s, 4);
            sc[u] = valid[u] ? s : -INFINITY;
        }
        const float mt = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
        const float m_new = fmaxf(st.m, mt);
        if (m_new != -INFINITY) {
            const float alpha = (st.m == -INFINITY) ? 0.f : ex2(st.m - m_new);
            st.l *= alpha;
#pragma unroll
            for (int d = 0; d < 8; ++d) st.o[d] *= alpha;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float p = ex2(sc[u] - m_new);             // exp2(-inf) = 0 for masked keys
                st.l += p;
                const float pb = bf2f(f2bf(p));                 // P enters the PV product as bf16 (same rule as the MFMA path)
#pragma unroll
                for (int d = 0; d < 8; ++d) st.o[d] = __builtin_fmaf(pb, bf2f((bf16_t)vf[u][d]), st.o[d]);
            }
            st.m = m_new;
        }
    };
    constexpr int NT = (D / 16 + 15) / 16;                                 // out-projection: 16-column tiles per wave
    auto dma_out_weights = [&]() {                                         // -> mypre, lane-linear
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int t = wid + 16 * i;
            if (t < D / 16) {
                if (FP8) {      // e4m3: 8 bytes per lane and k-half, as two 4-byte pieces [tile][k-half][piece][lane]
                    const unsigned char* wp = (const unsigned char*)a.aow + (size_t)(t * 16 + frow) * D + head * 64 + fq * 8;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        __builtin_amdgcn_global_load_lds(GLB_PTR(wp + (q >> 1) * 32 + (q & 1) * 4), LDS_PTR(mypre + (4 * i + q) * 256), 4, 0, 0);
                } else {        // bf16: [tile][k-half][lane] 16 bytes
                    const bf16_t* wp = (const bf16_t*)a.aow + (size_t)(t * 16 + frow) * D + head * 64 + fq * 8;
                    __builtin_amdgcn_global_load_lds(GLB_PTR(wp), LDS_PTR(mypre + (2 * i) * 1024), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds(GLB_PTR(wp + 32), LDS_PTR(mypre + (2 * i + 1) * 1024), 16, 0, 0);
                }
            }
        }
    };
    {
        bf16x8 kA[4], vA[4], kB[4], vB[4];
        bool okA[4], okB[4];
        int g = wid * 32;
        if (g < Lk) {
            // Loads are issued unconditionally (a group past the last key reads key 0 and is masked): no branch
            // ever merges a loaded register with an undefined one, so nothing waits for a load before its use.
            load_group(g + 512, kB, vB, okB);
            // vmcnt counts in issue order: all but the 8 youngest operations (group 1's loads) done = group 0 is in LDS
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                okA[u] = g + u * 8 + kk < Lk;
                kA[u] = *(const bf16x8*)(mypre + u * 1024 + lane * 16);
                vA[u] = *(const bf16x8*)(mypre + 4096 + u * 1024 + lane * 16);
            }
            reduce_group(kA, vA, okA);                                     // (waits for the LDS reads)
            // the wave's 8 KiB are free again: its out-projection weight fragments arrive under the rest of the attention
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (!SPLIT) dma_out_weights();
            g += 512;
            while (g < Lk) {                                               // kB holds group g
                load_group(g + 512, kA, vA, okA);
                reduce_group(kB, vB, okB);
                g += 512;
                if (g >= Lk) break;
                load_group(g + 512, kB, vB, okB);
                reduce_group(kA, vA, okA);
                g += 512;
            }
        } else if (!SPLIT) {
            dma_out_weights();
        }
    }
    // merge the 8 key-groups of the wave (lanes with equal sub)
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {
        const float m2 = __shfl_xor(st.m, off), l2 = __shfl_xor(st.l, off);
        float o2[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) o2[d] = __shfl_xor(st.o[d], off);
        merge(st, m2, l2, o2);
    }
    if (!SPLIT) {
        if (kk == 0) {
            wsm[wid][sub][0] = st.m; wsm[wid][sub][1] = st.l;
#pragma unroll
            for (int d = 0; d < 8; ++d) wsm[wid][sub][2 + d] = st.o[d];
        }
        __syncthreads();
    } else {
        // this wave's partial state -> the unit's scratch [16 waves][8][12 floats]: three 16-byte write-through stores per
        // lane; every storing wave drains, the workgroup meets, ONE lane takes the ticket (Guideline 16 / "Valid forms")
        const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void*)a.kpart, 0, a.kpart_bytes, 0x00020000);
        const int ubase = ((m * H + head) * 16) * 8 * 12 * 4;             // bytes
        if (kk == 0) {
            const int off = ubase + (wid * 8 + sub) * 48;
            const f32x4 v0 = {st.m, st.l, st.o[0], st.o[1]}, v1 = {st.o[2], st.o[3], st.o[4], st.o[5]}, v2 = {st.o[6], st.o[7], 0.f, 0.f};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_tb, v0), krs, off, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_tb, v1), krs, off + 16, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_tb, v2), krs, off + 32, 0, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(a.kcnt + m * H + head, 1u, RLX_AGENT);
            const int last = old == (unsigned)(S - 1);
            if (last) __hip_atomic_store(a.kcnt + m * H + head, 0u, RLX_AGENT);   // all S have arrived: ready for the next launch
            last_flag = last;
        }
        __syncthreads();
        if (!last_flag) return;
        // the last arriver gathers the 16 partial states (6 KiB, write-through stored by their owners: sc1 loads)
        for (int i = tid; i < 16 * 8 * 3; i += 64 * WPB) {
            const u32x4_tb v = __builtin_amdgcn_raw_buffer_load_b128(krs, ubase + i * 16, 0, 16);
            *(u32x4_tb*)((char*)&wsm[0][0][0] + i * 16) = v;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (tid < 8) {
#pragma clang fp contract(off)
        Part t;
        t.m = wsm[0][tid][0]; t.l = wsm[0][tid][1];
#pragma unroll
        for (int d = 0; d < 8; ++d) t.o[d] = wsm[0][tid][2 + d];
        for (int w = 1; w < 16; ++w) merge(t, wsm[w][tid][0], wsm[w][tid][1], &wsm[w][tid][2]);
        const float inv = 1.0f / t.l;
        uint4 v;
        v.x = pack_bf2(t.o[0] * inv, t.o[1] * inv); v.y = pack_bf2(t.o[2] * inv, t.o[3] * inv);
        v.z = pack_bf2(t.o[4] * inv, t.o[5] * inv); v.w = pack_bf2(t.o[6] * inv, t.o[7] * inv);
        *(uint4*)(ctxs + tid * 8) = v;                                     // context enters the out-projection as bf16
    }
    __syncthreads();
    TXT_STAMP(3);

    // ---- 2: this head's share of the output dense ---------------------------------------------------------------
    if (SPLIT) {
        // the last arriver did not know it would be: no prefetched weights.  Tiles wl, wl + WPB, ... of the D/16; the
        // fragments of up to CH tiles are requested together (one round trip), straight to registers
        constexpr int CH = 12;
        const bf16x8 c0 = *(const bf16x8*)(ctxs + fq * 8), c1 = *(const bf16x8*)(ctxs + 32 + fq * 8);
        float* pp = a.part + ((size_t)m * H + head) * D;
        for (int t0 = wl; t0 < D / 16; t0 += WPB * CH) {
            bf16x8 w0[CH], w1[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int t = t0 + i * WPB;
                if (t < D / 16) {
                    if (FP8) {
                        const unsigned char* wp = (const unsigned char*)a.aow + (size_t)(t * 16 + frow) * D + head * 64 + fq * 8;
                        const float sc = a.aoscale[t * 16 + frow];
                        w0[i] = fp8x8_to_bf16x8(*(const uint2*)wp, sc);
                        w1[i] = fp8x8_to_bf16x8(*(const uint2*)(wp + 32), sc);
                    } else {
                        const bf16_t* wp = (const bf16_t*)a.aow + (size_t)(t * 16 + frow) * D + head * 64 + fq * 8;
                        w0[i] = *(const bf16x8*)wp;
                        w1[i] = *(const bf16x8*)(wp + 32);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int t = t0 + i * WPB;
                if (t < D / 16) {
                    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[i], c0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[i], c1, acc, 0, 0, 0);
                    if (frow == 0) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) __hip_atomic_store(pp + t * 16 + fq * 4 + e, acc[e], RLX_AGENT);
                    }
                }
            }
        }
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the wave's own weight DMA (read by itself only)
        const bf16x8 c0 = *(const bf16x8*)(ctxs + fq * 8), c1 = *(const bf16x8*)(ctxs + 32 + fq * 8);
        float* pp = a.part + ((size_t)m * H + head) * D;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int t = wid + 16 * i;
            if (t < D / 16) {
                bf16x8 w0, w1;
                if (FP8) {      // e4m3 x the row's power-of-two scale = the bf16-stored weight, bit for bit
                    const unsigned* q = (const unsigned*)(mypre + (4 * i) * 256) + lane;
                    const float sc = a.aoscale[t * 16 + frow];
                    w0 = fp8x8_to_bf16x8(make_uint2(q[0], q[64]), sc);
                    w1 = fp8x8_to_bf16x8(make_uint2(q[128], q[192]), sc);
                } else {
                    w0 = *(const bf16x8*)(mypre + (2 * i) * 1024 + lane * 16);
                    w1 = *(const bf16x8*)(mypre + (2 * i + 1) * 1024 + lane * 16);
                }
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, c0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, c1, acc, 0, 0, 0);
                if (frow == 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) __hip_atomic_store(pp + t * 16 + fq * 4 + e, acc[e], RLX_AGENT);
                }
            }
        }
    }
    // ---- 3: ticket; the last unit of the row reduces ---------------------------------------------------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // every storing wave, before the barrier
    __syncthreads();
    TXT_STAMP(4);
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(a.cnt + m, 1u, RLX_AGENT);
        const int last = old == (unsigned)(H - 1);
        if (last) __hip_atomic_store(a.cnt + m, 0u, RLX_AGENT);            // all H have arrived: ready for the next launch
        last_flag = last;
    }
    __syncthreads();
    TXT_STAMP(5);
    if (!last_flag) return;
    {
        // The row's LayerNorm with 64 * WPB threads: wave wl takes the 64-column chunks wl, wl + WPB, ... (the 16-wave
        // workgroup: chunk = wave).  The sums are formed the same way for every WPB -- shuffle tree inside a chunk, then the
        // 16 chunk sums in order -- and every multiply-add is spelled out (no contraction), so the result does not depend
        // on how many workgroups shared the unit.
#pragma clang fp contract(off)
        constexpr int NCH = (D + 63) / 64, CPW = (NCH + WPB - 1) / WPB;
        float v[CPW], g[CPW], b[CPW];
        if (tid < 16) { red[0][tid] = 0.f; red[1][tid] = 0.f; }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < CPW; ++j) {
            const int ch = wl + j * WPB, col = ch * 64 + lane;
            v[j] = 0.f; g[j] = 0.f; b[j] = 0.f;
            if (ch < NCH && col < D) {
                g[j] = a.g1[col]; b[j] = a.b1[col];
                // all H partials are requested before the first add; summed in head order
                float s = 0.f;
                const float* pp = a.part + (size_t)m * H * D + col;
                for (int h0 = 0; h0 < H; h0 += 12) {
                    float p[12];
#pragma unroll
                    for (int h = 0; h < 12; ++h) p[h] = (h0 + h < H) ? __hip_atomic_load(pp + (size_t)(h0 + h) * D, RLX_AGENT) : 0.f;
#pragma unroll
                    for (int h = 0; h < 12; ++h) s += p[h];
                }
                v[j] = s + (a.aob[col] + a.xin[(size_t)m * D + col]);
            }
        }
        TXT_STAMP(8);
#pragma unroll
        for (int j = 0; j < CPW; ++j) {
            const int ch = wl + j * WPB;
            const float ws = wave_sum(v[j]);
            if (ch < NCH && lane == 0) red[0][ch] = ws;
        }
        __syncthreads();
        float sm = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) sm += red[0][w];
        const float mean = sm / (float)D;
        TXT_STAMP(9);
        float d[CPW];
#pragma unroll
        for (int j = 0; j < CPW; ++j) {
            const int ch = wl + j * WPB, col = ch * 64 + lane;
            d[j] = (ch < NCH && col < D) ? v[j] - mean : 0.f;
            const float dd = d[j] * d[j];
            const float ws = wave_sum(dd);
            if (ch < NCH && lane == 0) red[1][ch] = ws;
        }
        __syncthreads();
        float sq = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) sq += red[1][w];
        const float rstd = rsqrtf(sq / (float)D + a.eps);
#pragma unroll
        for (int j = 0; j < CPW; ++j) {
            const int ch = wl + j * WPB, col = ch * 64 + lane;
            if (ch < NCH && col < D) {
                const float y = __builtin_fmaf(d[j] * rstd, g[j], b[j]);
                a.xs[(size_t)m * D + col] = y; a.xsb[(size_t)m * D + col] = f2bf(y);
            }
        }
    }
    TXT_STAMP(6);
}

}  // namespace

bool txt_block_ok(int D) { return D == 128 || D == 768; }

// workgroups per unit for a launch of M rows x H heads: the largest power of two (<= 8: at least two waves stay together
// for the last arriver's share) that keeps units x S within the 256 CUs; 1 = the 16-wave workgroup.  Needs the scratch.
std::atomic<bool> g_key_split{!env_flag("GITCAP_NO_KEY_SPLIT")};      // gitcap_dbg_config(5, .)

int txt_block_split(int M, int H, bool have_scratch, int scratch_rows) {
    if (!g_key_split || !have_scratch || M > scratch_rows) return 1;
    int S = 1;
    while (S < 8 && M * H * S * 2 <= 256) S *= 2;
    return S;
}

template <int K32, bool FP8>
static hipError_t launch_split(const TxtBlockArgs& a, int S, int grid1, hipStream_t s) {
    const int M = a.rows * a.T;
    switch (S) {
        case 8: hipLaunchKernelGGL((txt_block_kernel<K32, FP8, 2>), dim3(M * a.H * 8), dim3(128), 0, s, a); break;
        case 4: hipLaunchKernelGGL((txt_block_kernel<K32, FP8, 4>), dim3(M * a.H * 4), dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL((txt_block_kernel<K32, FP8, 8>), dim3(M * a.H * 2), dim3(512), 0, s, a); break;
        default: hipLaunchKernelGGL((txt_block_kernel<K32, FP8, 16>), dim3(grid1), dim3(1024), 0, s, a); break;
    }
    return hipGetLastError();
}

hipError_t launch_txt_block(const TxtBlockArgs& a_in, hipStream_t s) {
    TxtBlockArgs a = a_in;
    const int M = a.rows * a.T;
    if (M <= 0 || a.H * 64 != a.D || a.beams <= 0 || !a.xin) return hipErrorInvalidValue;
    // first "half" of the rows for heads 8..11 (H == 12 mapping): whole clips (all beams of a clip stay together)
    const int unit = a.T == 1 ? a.beams : 1;
    a.Mh = ((M / unit + 1) / 2) * unit;
    const int grid = a.H == 12 ? 8 * (M + (a.Mh > M - a.Mh ? a.Mh : M - a.Mh)) : M * a.H;
    const int S = txt_block_split(M, a.H, a.kpart != nullptr && a.kcnt != nullptr, a.kpart_rows);
    const int key = a.D * 2 + (a.aoscale ? 1 : 0);
    switch (key) {
        case 256: return launch_split<4, false>(a, S, grid, s);
        case 257: return launch_split<4, true>(a, S, grid, s);
        case 1536: return launch_split<24, false>(a, S, grid, s);
        case 1537: return launch_split<24, true>(a, S, grid, s);
        default: return hipErrorInvalidValue;
    }
}
