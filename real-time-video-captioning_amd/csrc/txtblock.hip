// Attention sub-layer of the GIT decoder for TEXT rows (one decode step, or T teacher-forced positions), fused from the
// attention itself to the LayerNorm that closes the sub-layer.
//
// One 16-wave workgroup = one unit (text row m = (r, j), head):
//   1. attention over the image keys of the row's clip + the text keys 0..t of the row.  q, k, v of the text rows were
//      written to the text K/V cache by the q|k|v projection launch (skinny.hip).  K/V are streamed from HBM straight to
//      VGPRs (two register sets: the loads of the next 32-key group are in flight while one is reduced), 8 lanes per key,
//      per-group online softmax; wave w takes the 32-key groups w, w+16, ... so the summation order depends on the key
//      count only (batch invariant, bitwise)
//   2. this head's share of the output dense: ctx_h[64] . Wo[:, 64h:64h+64]^T -> part[m][head][D] (fp32, write-through);
//      the weight fragments a wave needs (6 KiB) are requested behind the attention and arrive across its merge
//   3. the LAST of the H units of a row to arrive (ticket counter; no unit ever waits for another) sums the H partials
//      in head order, adds bias + residual and applies LayerNorm -> x1 (fp32) and bf16(x1) for the FC1 launch.
// This replaces three launches of the first version (attention, split-K output dense, reduce + LayerNorm) by one.
// Hand-off in 3 follows cdna_hip_programming.md Guideline 16 / MI355X_MICROARCH.md "Valid forms": every payload store
// is an agent-scope (sc1) store, every storing wave drains vmcnt before the workgroup barrier, ONE lane adds the ticket,
// the reducer's loads are agent-scope (sc1) loads issued after the barrier its ticket lane joined.  No spin anywhere.
//
// Why the q|k|v projection is NOT in here (measured, tools/probe/txtblock_probe.hip): a CU pulls weight fragments at
// ~30 GB/s whatever serves them (HBM, Infinity Cache or L2); 295 KB of q|k|v weights per (row, head) unit cost 10 us in
// front of the attention, against 6 us for a launch of single-wave tiles that reads every weight byte once.
//
// No LDS-DMA in this kernel (it had some until the middle of round 4: the first key group and the output-dense fragments): behind
// a global_load_lds the compiler's wait-count pass takes the wave to have a FLAT operation pending and turns every wait for a
// plain load into s_waitcnt vmcnt(0); a run-time branch around a load and a conditionally issued load do the same where the paths
// meet.  The counted waits written in the source then never took effect -- the first reduce started when the last load in flight
// had landed -- and three A/Bs of the load schedule measured nothing.  Rules kept here: plain loads only, issued unconditionally
// (masked groups read key 0), cache policy as a template flag, branches around reduces but not around loads.  Stamps, the ISA
// evidence and the matrix-core form that was tried instead: profiles/r04_text_attention_phase_stamps.txt.
#include "kernels.h"
#include <atomic>
#include <cstdlib>

namespace {

constexpr float kScaleLog2e = 0.125f * 1.4426950408889634f;  // 1/sqrt(64) * log2(e)
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }

// Every multiply-add of the softmax arithmetic is spelled out (contraction off), so the rounding does not depend on what the
// compiler chooses to fuse in a given instantiation or compiler version.  (Round 3 built a key-split form of this kernel
// for small row counts -- the 16 waves of a unit dealt to 2-8 workgroups, per-wave partial states merged by the last
// arriver in wave order, bit for bit the same -- and found (a) that two instantiations of the SAME source had been
// contracted differently, one context value crossing a bf16 rounding boundary once in ~100 launches, hence this; and
// (b) no gain: one clip, 20 tokens 5.95 / 7.44 ms with 4 / 8 workgroups per unit against 5.87 ms, the last arriver's
// serial tail -- gather, unprefetched output-dense weights, tickets -- costs what the shorter K/V stream saves;
// profiles/r03_text_attention_key_split_latency.txt.  Removed.)
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

// The row is held as 16 "virtual waves" of 64 columns (column c = 64 v + lane).  With 16 physical waves a thread holds one
// column (NC = 1); with 8 it holds columns tid and tid + 512, i.e. virtual waves wid and wid + 8 (NC = 2).  Each virtual
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

// NW = physical waves per workgroup: 16 (one unit per CU: the K/V stream of a long image prefix wants every wave of the CU),
// or 8 (two units per CU) for launches of more units than CUs -- 32 single frames are 384 units of 50 KB
// of K/V each, two rounds of 16-wave workgroups.  The keys are dealt to 16 VIRTUAL waves either way (virtual wave v takes
// the 32-key groups v, v + 16, ...; a physical wave of the 8-wave form runs virtual waves wid and wid + 8 one after the
// other) and every merge is in virtual-wave order, so both forms give the same bits (speed switch 9 / tests).
// V8 (opt-in kv_cache = v_e4m3): the V rows of the IMAGE keys come as e4m3 codes + one power-of-two scale per (key, head)
// (rowops.hip: kv_quant_v_kernel; 3/4 of the bytes of the K/V stream), K and the row's own text K/V stay bf16.  The image keys are
// dealt to the 16 virtual waves in 32-key groups as before; the text keys 0..t of the row (a second format) are one more run of
// groups handled by the virtual wave that is next in the dealing, tw = ceil(S_img / 32) mod 16, after its image groups and into the
// same softmax state.  The summation order is a function of (S_img, t) only: batch invariant, cached == teacher-forced bitwise.
// p * (code * scale) is computed as (p * scale) * code: the same real product, rounded once by the same FMA.
template <int K32, bool FP8, int NW, bool NT_KV, bool V8 = false>
__global__ __launch_bounds__(NW * 64, 4) void txt_block_kernel(TxtBlockArgs a) {
    constexpr int D = K32 * 32;
    constexpr int NC = 16 / NW;                                 // columns (virtual waves) per thread in the row reducer
    static_assert(NW == 16 || (D / 16) % 8 == 0, "8-wave form: whole out-projection tiles per wave");
    __shared__ __attribute__((aligned(16))) bf16_t ctxs[64];
    __shared__ float red[2][16];
    __shared__ float wsm[16][8][10];
    __shared__ int last_flag;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = a.rows * a.T, H = a.H;
    // unit -> (m, head).  H == 12: blocks that share an XCD (blockIdx % 8 equal) take whole heads (8 heads on 8 XCDs,
    // heads 8..11 as half-heads of Mh | M - Mh rows), so a head's output-dense slice stays in ONE L2 and the beams of a
    // clip (same image K/V) meet there too.  Placement only affects speed.
    int m, head;
    if (H == 12) {
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
    const int r = m / a.T, j = m - r * a.T;
    const int clip = r / a.beams;
    const int ld = 3 * D;
    const int tq = a.t0 + j;
    const int frow = lane & 15, fq = lane >> 4;
    TXT_STAMP(0);

    const int Lk = a.S_img + tq + 1;
    const int sub = lane & 7, kk = lane >> 3;
    // (Round 4 timed a head-major image K/V cache -- one contiguous 256-B record per key and head, a contiguous stream per
    // unit -- by addressing this buffer that way: 301 vs 307 us per token step at 16 clips, 209 vs 211 at one
    // (profiles/r04_text_attention_head_major_kv_timing.txt).  Not worth a second GEMM epilogue and attention read path.)
    const bf16_t* img = a.kv_img + (size_t)clip * a.S_img * ld + D + head * 64 + sub * 8;
    const bf16_t* txt = a.kv_txt + (size_t)r * a.Tmax * ld + D + head * 64 + sub * 8;
    TXT_STAMP(2);
    // ---- 1: attention of (row, position tq, head) ------------------------------------------------------------------
    // q is requested first and converted only when the first two key groups have been requested as well (the conversion is the
    // first thing that waits for a load: ahead of the requests it would hold them back by a round trip)
    float qv[8];
    const bf16x8 q8 = *(const bf16x8*)(a.kv_txt + ((size_t)r * a.Tmax + tq) * ld + head * 64 + sub * 8);
    auto convert_q = [&]() {
        __builtin_amdgcn_sched_barrier(0);          // (the scheduler would start unpacking q between the requests, waiting there)
#pragma unroll
        for (int d = 0; d < 8; ++d) qv[d] = bf2f((bf16_t)q8[d]) * kScaleLog2e;
    };
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
            // NT_KV: read once per launch and too large to stay cached: do not displace what the GEMMs re-read.  (A template flag: a
            // run-time branch around a load makes the compiler merge "loaded" with "not loaded" registers behind it and wait there.)
            kf[u] = NT_KV ? __builtin_nontemporal_load((const bf16x8*)kp) : *(const bf16x8*)kp;
            vf[u] = NT_KV ? __builtin_nontemporal_load((const bf16x8*)(kp + D)) : *(const bf16x8*)(kp + D);
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
    // lanes with equal sub merge their 8 key rows; the virtual wave's state goes to LDS
    auto finish_virtual = [&](int vw) {
#pragma unroll
        for (int off = 8; off < 64; off <<= 1) {
            const float m2 = __shfl_xor(st.m, off), l2 = __shfl_xor(st.l, off);
            float o2[8];
#pragma unroll
            for (int d = 0; d < 8; ++d) o2[d] = __shfl_xor(st.o[d], off);
            merge(st, m2, l2, o2);
        }
        if (kk == 0) {
            wsm[vw][sub][0] = st.m; wsm[vw][sub][1] = st.l;
#pragma unroll
            for (int d = 0; d < 8; ++d) wsm[vw][sub][2 + d] = st.o[d];
        }
    };
    if constexpr (V8) {
        const int S = a.S_img;
        // head-major codes: this unit's V stream is one contiguous run of 64-byte records (8 keys = 512 B per wave instruction), its
        // scales one contiguous run of floats: the 32 scales of a key group are ONE load (lane i: key g0 + (i & 31)), handed to the
        // lanes of each key by ds_bpermute
        const unsigned char* v8p = a.v8_img + ((size_t)head * a.v8_pitch + (size_t)clip * S) * 64 + sub * 8;
        const float* vsp = a.vs_img + (size_t)head * a.v8_pitch + (size_t)clip * S;
        typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
        struct G8 { bf16x8 k[4]; u32x2 v[4]; float sc; bool ok[4]; };
        auto load8 = [&](int g0, G8& x) {                                  // image keys g0 .. g0 + 31 (past the last: key 0, masked)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int key = g0 + u * 8 + kk;
                x.ok[u] = key < S;
                key = x.ok[u] ? key : 0;
                const bf16_t* kp = img + (size_t)key * ld;
                const u32x2* vp = (const u32x2*)(v8p + (size_t)key * 64);
                x.k[u] = NT_KV ? __builtin_nontemporal_load((const bf16x8*)kp) : *(const bf16x8*)kp;
                x.v[u] = NT_KV ? __builtin_nontemporal_load(vp) : *vp;
            }
            const int sk = g0 + (lane & 31);
            x.sc = vsp[sk < S ? sk : 0];
        };
        auto reduce8 = [&](const G8& x) {
#pragma clang fp contract(off)
            float sc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float s = 0.f;
#pragma unroll
                for (int d = 0; d < 8; ++d) s = __builtin_fmaf(qv[d], bf2f((bf16_t)x.k[u][d]), s);
                s += __shfl_xor(s, 1);
                s += __shfl_xor(s, 2);
                s += __shfl_xor(s, 4);
                sc[u] = x.ok[u] ? s : -INFINITY;
            }
            float vsc[4];                                            // (all lanes take part: not under the branch below)
#pragma unroll
            for (int u = 0; u < 4; ++u) vsc[u] = __shfl(x.sc, u * 8 + kk);
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
                    const float pbs = bf2f(f2bf(p)) * vsc[u];       // P enters the PV product as bf16; x the key's power-of-two scale: exact
                    typedef __attribute__((ext_vector_type(2))) float f32x2;
                    const f32x2 c01 = __builtin_amdgcn_cvt_pk_f32_fp8((int)x.v[u][0], false), c23 = __builtin_amdgcn_cvt_pk_f32_fp8((int)x.v[u][0], true);
                    const f32x2 c45 = __builtin_amdgcn_cvt_pk_f32_fp8((int)x.v[u][1], false), c67 = __builtin_amdgcn_cvt_pk_f32_fp8((int)x.v[u][1], true);
                    const float c[8] = {c01[0], c01[1], c23[0], c23[1], c45[0], c45[1], c67[0], c67[1]};
#pragma unroll
                    for (int d = 0; d < 8; ++d) st.o[d] = __builtin_fmaf(pbs, c[d], st.o[d]);
                }
                st.m = m_new;
            }
        };
        const int tw = ((S + 31) >> 5) & 15;                               // the virtual wave that also takes the row's text keys
        bf16x8 kT[4], vT[4];
        bool okT[4];
        auto text_keys = [&]() {                                           // keys S .. Lk - 1, bf16, 32 at a time (loads and reduces in one branch)
            for (int tg = S; tg < Lk; tg += 32) {
                load_group(tg, kT, vT, okT);
                reduce_group(kT, vT, okT);
            }
        };
        G8 A, B;
        if (NW == 16) {
            int g = wid * 32;
            if (g < S) {
                load8(g, A);
                load8(g + 512, B);
                convert_q();
                reduce8(A);
                load8(g + 1024, A);
                __builtin_amdgcn_sched_barrier(0);
                if (g + 512 < S) reduce8(B);
                load8(g + 1536, B);
                __builtin_amdgcn_sched_barrier(0);
                if (g + 1024 < S) reduce8(A);
                g += 1536;
                while (g < S) {                                            // longer prefixes: B holds group g
                    load8(g + 512, A);
                    reduce8(B);
                    g += 512;
                    if (g >= S) break;
                    load8(g + 512, B);
                    reduce8(A);
                    g += 512;
                }
            } else {
                convert_q();
            }
            if (wid == tw) text_keys();
            finish_virtual(wid);
        } else {
            convert_q();
            for (int g = wid * 32; g < S; g += 512) {
                load8(g, A);
                reduce8(A);
            }
            if (wid == tw) text_keys();
            finish_virtual(wid);
            st.m = -INFINITY; st.l = 0.f;
#pragma unroll
            for (int d = 0; d < 8; ++d) st.o[d] = 0.f;
            for (int g = (wid + 8) * 32; g < S; g += 512) {
                load8(g, A);
                reduce8(A);
            }
            if (wid + 8 == tw) text_keys();
            finish_virtual(wid + 8);
        }
    } else
    {
        bf16x8 kA[4], vA[4], kB[4], vB[4];
        bool okA[4], okB[4];
        int g = wid * 32;
        if (g < Lk) {
            // No LDS-DMA in this kernel: behind a global_load_lds the compiler's wait-count pass turns every wait for a register load
            // into s_waitcnt vmcnt(0) ("pending FLAT"), so that the first reduce started when the LAST load in flight had landed
            // and no reduce ever ran under the next group's loads (in-kernel stamps, profiles/r04_text_attention_phase_stamps.txt).
            // Loads are issued unconditionally (a group past the last key reads key 0 and is masked): no branch ever merges a
            // loaded register with an undefined one, so the waits the compiler counts are "all but the 8 youngest".
            load_group(g, kA, vA, okA);
            if (NW == 16) {
                // the first three groups (all there are at 6 frames) in straight-line code: branches around reduces only
                load_group(g + 512, kB, vB, okB);
                convert_q();
                reduce_group(kA, vA, okA);
                TXT_STAMP(11);
                load_group(g + 1024, kA, vA, okA);
                __builtin_amdgcn_sched_barrier(0);  // every request of a group goes out before the next reduce waits for anything
                if (g + 512 < Lk) reduce_group(kB, vB, okB);
                load_group(g + 1536, kB, vB, okB);
                __builtin_amdgcn_sched_barrier(0);
                if (g + 1024 < Lk) reduce_group(kA, vA, okA);
                g += 1536;
                while (g < Lk) {                                           // longer prefixes: kB holds group g
                    load_group(g + 512, kA, vA, okA);
                    reduce_group(kB, vB, okB);
                    g += 512;
                    if (g >= Lk) break;
                    load_group(g + 512, kB, vB, okB);
                    reduce_group(kA, vA, okA);
                    g += 512;
                }
            } else {
                // 8-wave form: one register set (128 VGPRs, two workgroups per CU: the other seven waves of the SIMD's
                // four cover the round trip); the same groups in the same order
                convert_q();
                reduce_group(kA, vA, okA);
                for (g += 512; g < Lk; g += 512) {
                    load_group(g, kA, vA, okA);
                    reduce_group(kA, vA, okA);
                }
            }
        }
        TXT_STAMP(12);
        finish_virtual(wid);
        TXT_STAMP(13);
        if (NW == 8) {                                                     // the wave's second virtual wave: keys 32 (wid + 8) ...
            st.m = -INFINITY; st.l = 0.f;
#pragma unroll
            for (int d = 0; d < 8; ++d) st.o[d] = 0.f;
            for (g = (wid + 8) * 32; g < Lk; g += 512) {
                load_group(g, kA, vA, okA);
                reduce_group(kA, vA, okA);
            }
            finish_virtual(wid + 8);
        }
    }
    // the wave's out-projection fragments (its 16-column tiles x the head's two k-steps), requested here, in flight across the
    // two barriers and the merge between them: from the fragment-major copy, or from the row-major matrix (bf16, or e4m3 bytes x the
    // row's power-of-two scale = the bf16-stored weight, bit for bit)
    constexpr int NTW = (D / 16 + NW - 1) / NW;
    constexpr bool WHOLE = (D / 16) % NW == 0;                             // every wave has NTW tiles (else: a guard per tile)
    bf16x8 ow[NTW][2];
    {
        asm volatile("" ::: "memory");                                     // not above the attention: its registers are taken
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            const int t = WHOLE ? wid + NW * i : min(wid + NW * i, D / 16 - 1);      // (clamped: loaded, not used)
            if (FP8) {
                const unsigned char* wp = (const unsigned char*)a.aow + (size_t)(t * 16 + frow) * D + head * 64 + fq * 8;
                const float sc = a.aoscale[t * 16 + frow];
                ow[i][0] = fp8x8_to_bf16x8(*(const uint2*)wp, sc);
                ow[i][1] = fp8x8_to_bf16x8(*(const uint2*)(wp + 32), sc);
            } else if (a.aowpk) {
                const bf16x8* wp = (const bf16x8*)a.aowpk + ((size_t)t * K32 + head * 2) * 64 + lane;
                ow[i][0] = wp[0]; ow[i][1] = wp[64];
            } else {
                const bf16_t* wp = (const bf16_t*)a.aow + (size_t)(t * 16 + frow) * D + head * 64 + fq * 8;
                ow[i][0] = *(const bf16x8*)wp; ow[i][1] = *(const bf16x8*)(wp + 32);
            }
        }
    }
    __syncthreads();
    TXT_STAMP(14);
    if (tid < 64) {
        // The 16 virtual waves' states in order, one thread per context element (sub = tid >> 3, d = tid & 7: element tid of the
        // head's 64): every thread runs the chain of maxima and scale factors and its own component of the sums -- the arithmetic
        // of merge() for that component, bit for bit -- with all 48 values it needs requested before the first step.  (8 threads
        // used to walk the chain for 9 components each, reading the next state from LDS at every step: 1.6 us of a 20 us launch.)
#pragma clang fp contract(off)
        const int sb = tid >> 3, dd = tid & 7;
        float mw[16], lw[16], ov[16];
#pragma unroll
        for (int w = 0; w < 16; ++w) { mw[w] = wsm[w][sb][0]; lw[w] = wsm[w][sb][1]; ov[w] = wsm[w][sb][2 + dd]; }
        float mm = mw[0], ll = lw[0], oo = ov[0];
#pragma unroll
        for (int w = 1; w < 16; ++w) {
            const float Mx = fmaxf(mm, mw[w]);
            const float s1 = (mm == -INFINITY) ? 0.f : ex2(mm - Mx);
            const float s2 = (mw[w] == -INFINITY) ? 0.f : ex2(mw[w] - Mx);
            ll = __builtin_fmaf(ll, s1, lw[w] * s2);
            oo = __builtin_fmaf(oo, s1, ov[w] * s2);
            mm = Mx;
        }
        const float inv = 1.0f / ll;
        ctxs[tid] = f2bf(oo * inv);                                        // context enters the out-projection as bf16
    }
    TXT_STAMP(15);
    __syncthreads();
    TXT_STAMP(3);

    // ---- 2: this head's share of the output dense ---------------------------------------------------------------
    {
        const bf16x8 c0 = *(const bf16x8*)(ctxs + fq * 8), c1 = *(const bf16x8*)(ctxs + 32 + fq * 8);
        float* pp = a.part + ((size_t)m * H + head) * D;
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            const int t = wid + NW * i;
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ow[i][0], c0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ow[i][1], c1, acc, 0, 0, 0);
            if (frow == 0 && (WHOLE || t < D / 16)) {
#pragma unroll
                for (int e = 0; e < 4; ++e) __hip_atomic_store(pp + t * 16 + fq * 4 + e, acc[e], RLX_AGENT);
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
        float v[NC], g[NC], b[NC];
        bool act[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = tid + c * NW * 64;
            act[c] = col < D;
            v[c] = 0.f; g[c] = 0.f; b[c] = 0.f;
            if (act[c]) {
                g[c] = a.g1[col]; b[c] = a.b1[col];
                const float ab = a.aob[col], xi = a.xin[(size_t)m * D + col];     // requested with gamma / beta, ahead of the partials
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
                v[c] = s + (ab + xi);
            }
        }
        block_layernorm<NC>(v, act, D, a.eps, g, b, red, tid);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = tid + c * NW * 64;
            if (act[c]) { a.xs[(size_t)m * D + col] = v[c]; a.xsb[(size_t)m * D + col] = f2bf(v[c]); }
        }
    }
    TXT_STAMP(6);
}

}  // namespace

bool txt_block_ok(int D) { return D == 128 || D == 768; }

// 8-wave workgroups (two units per CU) for launches of more units than the device has CUs (speed only: same bits);
// gitcap_dbg_config(9, 0) / GITCAP_NO_TXT8: always 16 waves
std::atomic<bool> g_txt8{!(getenv("GITCAP_NO_TXT8") && atoi(getenv("GITCAP_NO_TXT8")))};

hipError_t launch_txt_block(const TxtBlockArgs& a_in, hipStream_t s) {
    TxtBlockArgs a = a_in;
    static const int nt = getenv("GITCAP_TXT_NT") ? atoi(getenv("GITCAP_TXT_NT")) : -1;    // A/B switch: 0 never, 1 always
    if (nt >= 0) a.nt_kv = nt;
    const int M = a.rows * a.T;
    if (M <= 0 || a.H * 64 != a.D || a.beams <= 0 || !a.xin || (a.v8_img != nullptr) != (a.vs_img != nullptr)) return hipErrorInvalidValue;
    // first "half" of the rows for heads 8..11 (H == 12 mapping): whole clips (all beams of a clip stay together)
    const int unit = a.T == 1 ? a.beams : 1;
    a.Mh = ((M / unit + 1) / 2) * unit;
    const int grid = a.H == 12 ? 8 * (M + (a.Mh > M - a.Mh ? a.Mh : M - a.Mh)) : M * a.H;
    const bool w8 = g_txt8 && M * a.H > device_cus();
    const int key = a.D * 2 + (a.aoscale ? 1 : 0);
#define TXT_LAUNCH_V(K32, F8, V8) do { \
        if (w8) { if (a.nt_kv) hipLaunchKernelGGL((txt_block_kernel<K32, F8, 8, true, V8>), dim3(grid), dim3(512), 0, s, a); \
                  else hipLaunchKernelGGL((txt_block_kernel<K32, F8, 8, false, V8>), dim3(grid), dim3(512), 0, s, a); } \
        else { if (a.nt_kv) hipLaunchKernelGGL((txt_block_kernel<K32, F8, 16, true, V8>), dim3(grid), dim3(1024), 0, s, a); \
               else hipLaunchKernelGGL((txt_block_kernel<K32, F8, 16, false, V8>), dim3(grid), dim3(1024), 0, s, a); } } while (0)
#define TXT_LAUNCH(K32, F8) do { if (a.v8_img) TXT_LAUNCH_V(K32, F8, true); else TXT_LAUNCH_V(K32, F8, false); } while (0)
    switch (key) {
        case 256: TXT_LAUNCH(4, false); break;
        case 257: TXT_LAUNCH(4, true); break;
        case 1536: TXT_LAUNCH(24, false); break;
        case 1537: TXT_LAUNCH(24, true); break;
        default: return hipErrorInvalidValue;
    }
#undef TXT_LAUNCH
#undef TXT_LAUNCH_V
    return hipGetLastError();
}
