// HBM-bound row kernels: LayerNorm (one wave per row, 16-byte vector loads), patch gather
// (im2col of NCHW fp32 frames, coalesced 16-B reads along W), text embedding, argmax.
#include "kernels.h"
#include <algorithm>
#include "ln_canon.h"
#include "rowln.h"

namespace {

// ---- LayerNorm -------------------------------------------------------------------------------
// One wave per row; lane holds NV float4 at columns 256*i + 4*lane (1 KiB contiguous per wave-instruction), fp32
// throughout.  CANON (D a multiple of 64): the segmented statistics of ln_canon.h -- 16 lanes x 4 columns are one
// 64-column segment, so (i, lane >> 4) names segment 4 i + (lane >> 4) -- bit for bit what the GEMM epilogue that
// normalises its own rows computes.  Otherwise: two-pass (mean, then centred variance) over the wave.
template <int NV, bool CANON>
__global__ __launch_bounds__(256) void layernorm_kernel(LnArgs a) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    const float* xr = a.x + (size_t)row * a.ldx;
    const bool is_cls = a.cls != nullptr && row % a.cls_period == 0;     // wave-uniform
    f32x4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 256 + lane * 4;
        if (c >= a.D) v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        else if (is_cls) v[i] = *(const f32x4*)(a.cls + c) + *(const f32x4*)(a.cls_pos + c);
        else v[i] = *(const f32x4*)(xr + c);
    }
    float mean, rstd;
    if (CANON) {
        float2 st[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) st[i] = ln_seg_stats(v[i]);        // segment 4 i + (lane >> 4), in all of its 16 lanes
        auto seg = [&](int sidx) {                                      // segment sidx -> every lane (compile-time index)
            const int src = (sidx & 3) * 16;
            return float2{__shfl(st[sidx >> 2].x, src), __shfl(st[sidx >> 2].y, src)};
        };
        if (a.D == 768) ln_merge<12>(seg, a.eps, mean, rstd);
        else if (a.D == 1024) ln_merge<16>(seg, a.eps, mean, rstd);
        else if (a.D == 128) ln_merge<2>(seg, a.eps, mean, rstd);
        else if (a.D == 64) ln_merge<1>(seg, a.eps, mean, rstd);
        else if (a.D == 256) ln_merge<4>(seg, a.eps, mean, rstd);
        else if (a.D == 512) ln_merge<8>(seg, a.eps, mean, rstd);
        else { mean = 0.f; rstd = 0.f; }                                // launcher never sends other widths here
    } else {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        mean = wave_sum(s) / (float)a.D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = i * 256 + lane * 4;
            if (c < a.D) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
            }
        }
        rstd = rsqrtf(wave_sum(q) / (float)a.D + a.eps);
    }
    const float* addv = a.add_vec ? a.add_vec + (size_t)((row / a.add_div) % a.add_mod) * a.D : nullptr;
    const bool two = CANON && a.gamma2 != nullptr;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 256 + lane * 4;
        if (c < a.D) {
            const f32x4 g = *(const f32x4*)(a.gamma + c);
            const f32x4 b = *(const f32x4*)(a.beta + c);
            f32x4 y;
            if (CANON) {
                y = ln_apply(v[i], mean, rstd, g, b);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
            }
            if (addv) y += *(const f32x4*)(addv + c);
            if (a.out_f32) *(f32x4*)(a.out_f32 + (size_t)row * a.ld_f32 + c) = y;
            if (a.out_bf16 && !two) {
                uint2 o;
                o.x = pack_bf2(y[0], y[1]);
                o.y = pack_bf2(y[2], y[3]);
                *(uint2*)(a.out_bf16 + (size_t)row * a.ld_bf16 + c) = o;
            }
            v[i] = y;
        }
    }
    if (CANON) {
        if (two) {       // second LayerNorm over the values just produced (what a second launch would read back)
            float2 st[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) st[i] = ln_seg_stats(v[i]);
            auto seg = [&](int sidx) {
                const int src = (sidx & 3) * 16;
                return float2{__shfl(st[sidx >> 2].x, src), __shfl(st[sidx >> 2].y, src)};
            };
            float mean2, rstd2;
            if (a.D == 768) ln_merge<12>(seg, a.eps2, mean2, rstd2);
            else if (a.D == 1024) ln_merge<16>(seg, a.eps2, mean2, rstd2);
            else if (a.D == 128) ln_merge<2>(seg, a.eps2, mean2, rstd2);
            else if (a.D == 64) ln_merge<1>(seg, a.eps2, mean2, rstd2);
            else if (a.D == 256) ln_merge<4>(seg, a.eps2, mean2, rstd2);
            else ln_merge<8>(seg, a.eps2, mean2, rstd2);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = i * 256 + lane * 4;
                if (c < a.D && a.out_bf16) {
                    const f32x4 y = ln_apply(v[i], mean2, rstd2, *(const f32x4*)(a.gamma2 + c), *(const f32x4*)(a.beta2 + c));
                    uint2 o;
                    o.x = pack_bf2(y[0], y[1]);
                    o.y = pack_bf2(y[2], y[3]);
                    *(uint2*)(a.out_bf16 + (size_t)row * a.ld_bf16 + c) = o;
                }
            }
        }
    }
}

// ---- im2col ----------------------------------------------------------------------------------
// thread = 4 consecutive px of one (patch row, c, py): one 16-B fp32 read, one 8-B bf16 write.
// p % 4 == 0 fast path; generic path handles p = 14 (ViT-L/14) with 2-element pieces.
template <int VEC>
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ frames, bf16_t* __restrict__ out,
                                                     int nf, int img, int p, int Kp) {
    const int G = img / p;
    const int kvec = Kp / VEC;                                   // pieces per patch row (incl. zero pad)
    const int64_t total = (int64_t)nf * G * G * kvec;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int kv = (int)(idx % kvec);
    const int64_t prow = idx / kvec;                             // frame*G*G + gy*G + gx
    const int k = kv * VEC;
    bf16_t* o = out + prow * Kp + k;
    if (k >= 3 * p * p) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = 0;
        return;
    }
    const int gx = (int)(prow % G), gy = (int)((prow / G) % G);
    const int64_t frame = prow / (G * G);
    const int c = k / (p * p), rem = k - c * p * p, py = rem / p, px = rem - py * p;
    const float* src = frames + ((frame * 3 + c) * img + (gy * p + py)) * (int64_t)img + gx * p + px;
    if (VEC == 4) {
        const f32x4 v = *(const f32x4*)src;
        uint2 w;
        w.x = pack_bf2(v[0], v[1]);
        w.y = pack_bf2(v[2], v[3]);
        *(uint2*)o = w;
    } else {
        *(unsigned*)o = pack_bf2(src[0], src[1]);
    }
}



__global__ void cast_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4 v = *(const f32x4*)(in + i * 4);
    uint2 w;
    w.x = pack_bf2(v[0], v[1]);
    w.y = pack_bf2(v[2], v[3]);
    *(uint2*)(out + i * 4) = w;
}

// per-layer hidden states -> caller layout [B][entries][S_img + T][D]; one 256-thread block per output row
__global__ __launch_bounds__(256) void gather_hidden_kernel(const float* __restrict__ img, const float* __restrict__ txt,
                                                            float* __restrict__ out, int n_entries, int S_img, int T, int D,
                                                            size_t img_stride, size_t txt_stride) {
    const int S = S_img + T;
    const size_t row = blockIdx.x;                       // ((b * n_entries) + e) * S + s
    const int s = (int)(row % S), e = (int)((row / S) % n_entries), b = (int)(row / ((size_t)S * n_entries));
    const float* src = s < S_img ? img + e * img_stride + ((size_t)b * S_img + s) * D
                                 : txt + e * txt_stride + ((size_t)b * T + (s - S_img)) * D;
    for (int c = threadIdx.x * 4; c < D; c += 1024) *(f32x4*)(out + row * D + c) = *(const f32x4*)(src + c);
}

// e4m3 weight panel -> bf16 staging panel for the big-tile GEMMs (16 values per thread: 16-B read, 32-B write)
__global__ __launch_bounds__(256) void dequant_fp8_kernel(const unsigned char* __restrict__ w8, const float* __restrict__ scale,
                                                          bf16_t* __restrict__ out, int K, int64_t n16) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n16) return;
    const int64_t e = i * 16;
    const float sc = scale[e / K];
    const uint4 q = *(const uint4*)(w8 + e);
    const bf16x8 lo = fp8x8_to_bf16x8(make_uint2(q.x, q.y), sc), hi = fp8x8_to_bf16x8(make_uint2(q.z, q.w), sc);
    const uint4 o4l = __builtin_bit_cast(uint4, lo), o4h = __builtin_bit_cast(uint4, hi);
    const unsigned o[8] = {o4l.x, o4l.y, o4l.z, o4l.w, o4h.x, o4h.y, o4h.z, o4h.w};
    *(uint4*)(out + e) = make_uint4(o[0], o[1], o[2], o[3]);
    *(uint4*)(out + e + 8) = make_uint4(o[4], o[5], o[6], o[7]);
}

__global__ __launch_bounds__(256) void dequant_fp8_batch_kernel(DequantBatch b) {
    const int which = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (which >= b.n || i >= b.n16[which]) return;
    const int64_t e = i * 16;
    const float sc = b.scale[which][e / b.K[which]];
    const uint4 q = *(const uint4*)(b.w8[which] + e);
    const bf16x8 lo = fp8x8_to_bf16x8(make_uint2(q.x, q.y), sc), hi = fp8x8_to_bf16x8(make_uint2(q.z, q.w), sc);
    *(uint4*)(b.out[which] + e) = __builtin_bit_cast(uint4, lo);
    *(uint4*)(b.out[which] + e + 8) = __builtin_bit_cast(uint4, hi);
}

// ---- text embedding + LayerNorm: one wave per (row, position) (rowln.h) --------------------------
template <int NV>
__global__ __launch_bounds__(256) void embed_text_kernel(const int64_t* __restrict__ ids, int ld_ids, int rows, int T,
                                                         int t0, const float* __restrict__ word,
                                                         const float* __restrict__ pos, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps, int D, int vocab,
                                                         float* __restrict__ xf, bf16_t* __restrict__ xb) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= rows * T) return;
    f32x4 v[NV];
    const float s = row_load_embed<NV>(v, ids, ld_ids, T, t0, word, pos, D, vocab, m, lane);
    row_layernorm<NV>(v, s, lane, D, eps, gamma, beta);
    row_store<NV>(v, lane, D, xf + (size_t)m * D, xb + (size_t)m * D);
}

// ---- split-K reduce + bias + residual + LayerNorm (text rows) (rowln.h) ---------------------------------------------
// One workgroup per row, one wave per group of 8 slabs: every wave sums its group (row_slab_tree), wave 0 adds the group
// sums in ascending order, then bias + residual, and normalises the row.  With up to 8 slabs that is one wave, the kernel of
// rounds 1-3; the fused FC1 -> GELU -> FC2 launch of round 4 (ffn_txt.hip) leaves dec_ffn / 64 = 48 slabs, 147 KB per
// row: six waves fetch them side by side.  Same bits as row_load_reduce in one wave (the row prologue of skinny.hip).
template <int NV>
__global__ __launch_bounds__(512) void ln_reduce_kernel(const float* __restrict__ slabs, int nslab,
                                                        const float* __restrict__ bias, const float* __restrict__ resid,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float eps, int M, int D, float* __restrict__ xf,
                                                        bf16_t* __restrict__ xb) {
    __shared__ __attribute__((aligned(16))) float part[7][NV * 256];           // group sums of waves 1..7
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6, m = blockIdx.x;
    const int ngroups = (nslab + 7) >> 3;
    // bias, residual row, gamma and beta of the row: requested before the slabs (every wave: a branch around the loads would be
    // waited for where it ends), in flight behind the slab tree and the merge instead of four round trips of wave 0 after them
    f32x4 bv[NV], rv[NV], gv[NV], bev[NV];
    row_load_vec<NV>(bv, bias, D, lane);
    row_load_vec<NV>(rv, resid + (size_t)m * D, D, lane);
    row_load_vec<NV>(gv, gamma, D, lane);
    row_load_vec<NV>(bev, beta, D, lane);
    f32x4 v[NV];
    row_slab_tree<NV>(v, slabs, nslab, g * 8, M, D, m, lane);
    if (ngroups > 1) {
        if (g > 0) {
#pragma unroll
            for (int i = 0; i < NV; ++i) *(f32x4*)(&part[g - 1][i * 256 + lane * 4]) = v[i];
        }
        __syncthreads();
        if (g > 0) return;
        for (int j = 1; j < ngroups; ++j)
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] += *(const f32x4*)(&part[j - 1][i * 256 + lane * 4]);
    }
    // (0 + t0) + t1 + ... of row_load_reduce: 0 + t0 == t0 exactly
    const float s = row_add_bias_resid_v<NV>(v, bv, rv, D, lane);
    row_layernorm_v<NV>(v, s, lane, D, eps, gv, bev);
    row_store<NV>(v, lane, D, xf + (size_t)m * D, xb + (size_t)m * D);
}

// ---- fragment-major copy of a GEMM weight for the weight-streaming text kernels ------------------------------------------
// src [Npad16][K] (ESZ-byte elements: bf16, or e4m3 codes) -> dst [tile = n / 16][k32 = k / 32][lane][8 elements] with
// lane = n % 16 + 16 * ((k % 32) / 8): the MFMA operand a lane of skinny.hip / txtblock.hip / ffn_txt.hip loads for k-step
// k32 of tile n / 16 is 16 (8) contiguous bytes, a wave instruction reads 1 KiB (512 B) contiguous and a tile's fragments
// are one contiguous run.  Measured (tools/probe/pull_probe.hip): a CU pulls 48 KiB per wave at 47 GB/s in the row-major
// pattern (16 segments of 64 B per instruction) and at 170 GB/s (71 from cold caches) from this layout.
template <typename T>
__global__ __launch_bounds__(256) void pack_frags_kernel(const T* __restrict__ src, T* __restrict__ dst, int K, int64_t total8) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;            // one 8-element fragment piece per thread
    if (i >= total8) return;
    const int lane = (int)(i & 63);
    const int64_t tk = i >> 6;
    const int K32 = K >> 5;
    const int64_t tile = tk / K32;
    const int k32 = (int)(tk - tile * K32);
    const T* sp = src + ((size_t)tile * 16 + (lane & 15)) * K + k32 * 32 + (lane >> 4) * 8;
    T* dp = dst + (size_t)i * 8;
    if (sizeof(T) == 2) *(uint4*)dp = *(const uint4*)sp;
    else *(uint2*)dp = *(const uint2*)sp;
}

// ---- final arg-max over the per-tile partials written by the vocabulary-head kernel ------------
template <int NV>
__global__ __launch_bounds__(256) void argmax_final_kernel(const float* __restrict__ val, const int* __restrict__ idx,
                                                           int ntiles, int row_stride, int row_off,
                                                           int64_t* __restrict__ out, int ld_out,
                                                           int32_t* __restrict__ sep_cnt, int step, int sep_id, NextEmbed emb) {
    __shared__ float sv[4];
    __shared__ int si[4];
    __shared__ int chosen;
    const int r = blockIdx.x, tid = threadIdx.x;
    const size_t base = (size_t)(r * row_stride + row_off) * ntiles;
    // gamma / beta of the next step's input row do not depend on the token chosen below: requested now (wave 0 uses them)
    f32x4 gv[NV > 0 ? NV : 1], bv[NV > 0 ? NV : 1];
    if (NV > 0) {
        row_load_vec<(NV > 0 ? NV : 1)>(gv, emb.gamma, emb.D, tid & 63);
        row_load_vec<(NV > 0 ? NV : 1)>(bv, emb.beta, emb.D, tid & 63);
    }
    float best = -INFINITY;
    int bi = 0x7fffffff;
    // eight partials per thread and round trip (30 522 words = 1908 tiles: one round trip instead of eight); an index past the end
    // re-reads the last tile, which cannot change the result (max value, smallest index among equals: order does not matter)
    for (int i0 = tid; i0 < ntiles; i0 += 256 * 8) {
        float v8[8];
        int j8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = min(i0 + 256 * u, ntiles - 1);
            v8[u] = val[base + i];
            j8[u] = idx[base + i];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (v8[u] > best || (v8[u] == best && j8[u] < bi)) { best = v8[u]; bi = j8[u]; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float v2 = __shfl_xor(best, o);
        const int i2 = __shfl_xor(bi, o);
        if (v2 > best || (v2 == best && i2 < bi)) { best = v2; bi = i2; }
    }
    if ((tid & 63) == 0) { sv[tid >> 6] = best; si[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
        if (bi == 0x7fffffff) bi = 0;
        out[(size_t)r * ld_out] = bi;
        if (sep_cnt && bi == sep_id) atomicAdd(&sep_cnt[step], 1);
        chosen = bi;
    }
    if (NV > 0) {       // the next step's input row: embedding of the token just chosen + LayerNorm (one wave)
        __syncthreads();
        if (tid < 64) {
            f32x4 v[NV > 0 ? NV : 1];
            const float s = row_load_embed_tok<(NV > 0 ? NV : 1)>(v, (int64_t)chosen, emb.position, emb.word, emb.pos, emb.D, emb.vocab, tid);
            row_layernorm_v<(NV > 0 ? NV : 1)>(v, s, tid, emb.D, emb.eps, gv, bv);
            row_store<(NV > 0 ? NV : 1)>(v, tid, emb.D, emb.xf + (size_t)r * emb.D, emb.xb + (size_t)r * emb.D);
        }
    }
}

// ---- beam candidates: top-K of (log_softmax(logits[b*beams+j]) + beam_score[b*beams+j]) over j, v ----
// Replaces log_softmax (:557) + add (:561) + view + topk (:563-565) of the reference search loop
// (src/models/model.py); flat index = j*V + v, sorted descending, ties by smaller flat index.
// Two launches (a single workgroup per clip scanned beams x V = 122 K logits three times: 371 us at the configs[4] shape,
// a quarter of its search loop):
//   chunks: one workgroup per (row, 2048-logit chunk): chunk max, sum of exp(x - max), and the chunk's top K by value
//           (inside a row the order by logit IS the order by score);
//   merge:  one wave per clip: log-sum-exp of every row from the chunk statistics (fixed order), score of every candidate,
//           top K of the beams x chunks x K candidates.  The global top K is a subset of the per-chunk top K's: exact.
constexpr int BT_CHUNK = 2048;

template <int KMAX>
__global__ __launch_bounds__(256) void beam_topk_chunks_kernel(const float* __restrict__ logits, int ld, int V, int K, int nch,
                                                               float2* __restrict__ stats, float* __restrict__ cval,
                                                               int* __restrict__ cidx) {
    __shared__ float red[4];
    __shared__ int redi[4];
    const int row = blockIdx.x / nch, c = blockIdx.x - row * nch;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float* src = logits + (size_t)row * ld;
    float x[8];
    float m = -INFINITY;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int i = c * BT_CHUNK + q * 256 + tid;
        x[q] = i < V ? src[i] : -INFINITY;
        m = fmaxf(m, x[q]);
    }
    m = wave_max(m);
    if (lane == 0) red[wid] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) sum += x[q] == -INFINITY ? 0.f : expf(x[q] - m);      // (a chunk of -inf only: m = -inf, sum = 0)
    sum = wave_sum(sum);
    if (lane == 0) red[wid] = sum;
    __syncthreads();
    if (tid == 0) stats[blockIdx.x] = float2{m, (red[0] + red[1]) + (red[2] + red[3])};
    __syncthreads();
    float tv[KMAX];
    int ti[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) { tv[k] = -INFINITY; ti[k] = 0x7fffffff; }
#pragma unroll
    for (int q = 0; q < 8; ++q) {                                        // ascending ids per thread: strict > keeps the earlier one
        float v = x[q];
        int id = c * BT_CHUNK + q * 256 + tid;
        if (v > tv[KMAX - 1]) {
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (v > tv[k]) { const float fv = tv[k]; const int fi = ti[k]; tv[k] = v; ti[k] = id; v = fv; id = fi; }
        }
    }
    for (int k = 0; k < K; ++k) {                                        // K rounds of block arg-max over the list heads
        float bv = tv[0];
        int bi = ti[0];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float v2 = __shfl_xor(bv, o);
            const int i2 = __shfl_xor(bi, o);
            if (v2 > bv || (v2 == bv && i2 < bi)) { bv = v2; bi = i2; }
        }
        if (lane == 0) { red[wid] = bv; redi[wid] = bi; }
        __syncthreads();
        bv = red[0]; bi = redi[0];
        for (int w = 1; w < 4; ++w)
            if (red[w] > bv || (red[w] == bv && redi[w] < bi)) { bv = red[w]; bi = redi[w]; }
        if (tid == 0) { cval[(size_t)blockIdx.x * K + k] = bv; cidx[(size_t)blockIdx.x * K + k] = bi; }
        if (ti[0] == bi && bi != 0x7fffffff) {                           // the owner pops its head
#pragma unroll
            for (int q = 0; q + 1 < KMAX; ++q) { tv[q] = tv[q + 1]; ti[q] = ti[q + 1]; }
            tv[KMAX - 1] = -INFINITY; ti[KMAX - 1] = 0x7fffffff;
        }
        __syncthreads();
    }
}

template <int KMAX>
__global__ __launch_bounds__(64) void beam_topk_merge_kernel(const float2* __restrict__ stats, const float* __restrict__ cval,
                                                             const int* __restrict__ cidx, const float* __restrict__ beam_scores,
                                                             int beams, int V, int K, int nch, float* __restrict__ out_scores,
                                                             int* __restrict__ out_idx) {
    __shared__ float add[16];
    const int b = blockIdx.x, lane = threadIdx.x;
    // (the chunk statistics of four beams per round trip, the candidates eight per round trip: requested together, then used --
    // one at a time every load was a round trip of its own, 30 of them for 4 beams x 30 chunks x 8 candidates)
    for (int j0 = 0; j0 < beams; j0 += 4) {
        float2 st4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) st4[u] = stats[(size_t)(b * beams + min(j0 + u, beams - 1)) * nch + min(lane, nch - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
        const int j = j0 + u;
        if (j >= beams) break;
        const float2 st = lane < nch ? st4[u] : float2{-INFINITY, 0.f};  // log-sum-exp of row j from its chunks (nch <= 64)
        const float M = wave_max(st.x);
        const float S = wave_sum(st.y == 0.f ? 0.f : st.y * expf(st.x - M));
        if (lane == 0) add[j] = beam_scores[b * beams + j] - (M + logf(S));
        }
    }
    __syncthreads();
    float tv[KMAX];
    int ti[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) { tv[k] = -INFINITY; ti[k] = 0x7fffffff; }
    const int per_row = nch * K, total = beams * per_row;
    for (int t0 = lane; t0 < total; t0 += 64 * 8) {
        int ci8[8];
        float cv8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const size_t at = (size_t)(b * beams) * per_row + min(t0 + 64 * u, total - 1);
            ci8[u] = cidx[at];
            cv8[u] = cval[at];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int t = t0 + 64 * u;
            if (t >= total || ci8[u] == 0x7fffffff) continue;            // past the end / an empty slot of a short last chunk
            const int j = t / per_row;
            float v = cv8[u] + add[j];
            int id = j * V + ci8[u];
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (v > tv[k] || (v == tv[k] && id < ti[k])) { const float fv = tv[k]; const int fi = ti[k]; tv[k] = v; ti[k] = id; v = fv; id = fi; }
        }
    }
    for (int k = 0; k < K; ++k) {                                        // K rounds of wave arg-max over the list heads
        float bv = tv[0];
        int bi = ti[0];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float v2 = __shfl_xor(bv, o);
            const int i2 = __shfl_xor(bi, o);
            if (v2 > bv || (v2 == bv && i2 < bi)) { bv = v2; bi = i2; }
        }
        if (lane == 0) { out_scores[b * K + k] = bv; out_idx[b * K + k] = bi; }
        if (ti[0] == bi && bi != 0x7fffffff) {
#pragma unroll
            for (int q = 0; q + 1 < KMAX; ++q) { tv[q] = tv[q + 1]; ti[q] = ti[q + 1]; }
            tv[KMAX - 1] = -INFINITY; ti[KMAX - 1] = 0x7fffffff;
        }
    }
}

// ---- device-side beam bookkeeping: one step of GeneratorWithBeamSearchV2.search (model.py:573-621) -----
// One block per batch element; thread 0 walks the <= 16 sorted candidates exactly like the reference's
// Python loop (finished hypotheses kept n_best = 1: score = sum_logprob / len^length_penalty, :503, :592-594;
// is_done test :576), then the block copies the surviving prefixes into the next id buffer (:620-621).
struct BeamState {
    int64_t* ids[2];          // [B*beams][max_len] prefixes, double buffered
    float* beam_scores;       // [B*beams]
    int64_t* words;           // [B*beams] token chosen for the next position (input of the next decoder step)
    int32_t* src_rows;        // [B*beams] row each new beam continues (for the KV reorder)
    int32_t* done;            // [B]
    int32_t* hyp_len;         // [B] 0 = no finished hypothesis yet
    float* hyp_score;         // [B]
    int64_t* hyp_ids;         // [B][max_len]
};
__global__ __launch_bounds__(64) void beam_step_kernel(BeamState st, const float* __restrict__ cand_scores,
                                                       const int* __restrict__ cand_idx, int beams, int K, int V,
                                                       int cur_len, int max_len, int eos, float length_penalty, int cur) {
    __shared__ int s_src[16];
    __shared__ long long s_word[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t* ids_old = st.ids[cur];
    int64_t* ids_new = st.ids[cur ^ 1];
    if (tid == 0) {
        const float* cs = cand_scores + (size_t)b * K;
        const int* ci = cand_idx + (size_t)b * K;
        bool done = st.done[b] != 0;
        if (!done && st.hyp_len[b] > 0)                                   // BeamHypotheses.is_done(best_sum_logprobs)
            done = st.hyp_score[b] >= cs[0] / powf((float)(max_len - 1), length_penalty);
        int kept = 0;
        if (!done) {
            for (int c = 0; c < K && kept < beams; ++c) {
                const int beam_id = ci[c] / V, word = ci[c] - beam_id * V;
                if (word == eos || cur_len + 1 == max_len) {              // finished hypothesis: ids[:cur_len]
                    const float score = cs[c] / powf((float)cur_len, length_penalty);
                    if (st.hyp_len[b] == 0 || score > st.hyp_score[b]) {
                        st.hyp_score[b] = score;
                        st.hyp_len[b] = cur_len;
                        const int64_t* srow = ids_old + (size_t)(b * beams + beam_id) * max_len;
                        for (int t = 0; t < cur_len; ++t) st.hyp_ids[(size_t)b * max_len + t] = srow[t];
                    }
                } else {
                    s_src[kept] = b * beams + beam_id;
                    s_word[kept] = word;
                    st.beam_scores[b * beams + kept] = cs[c];
                    ++kept;
                }
            }
        }
        if (kept < beams) {                                               // done, or the last step: pad (0, eos, row 0)
            for (int j = 0; j < beams; ++j) {
                s_src[j] = 0; s_word[j] = eos;
                st.beam_scores[b * beams + j] = 0.f;
            }
        }
        st.done[b] = done ? 1 : 0;
    }
    __syncthreads();
    for (int j = 0; j < beams; ++j) {
        const int r = b * beams + j, src = s_src[j];
        for (int t = tid; t < cur_len; t += 64) ids_new[(size_t)r * max_len + t] = ids_old[(size_t)src * max_len + t];
        if (tid == 0) {
            ids_new[(size_t)r * max_len + cur_len] = s_word[j];
            st.words[r] = s_word[j];
            st.src_rows[r] = src;
        }
    }
}

// decoded[b] = best hypothesis + EOS padding (model.py:653-678, num_keep_best = 1)
__global__ void beam_finish_kernel(BeamState st, int max_len, int eos, int64_t* __restrict__ decoded, float* __restrict__ logprobs) {
    const int b = blockIdx.x;
    const int n = st.hyp_len[b];
    for (int t = threadIdx.x; t < max_len; t += blockDim.x)
        decoded[(size_t)b * max_len + t] = t < n ? st.hyp_ids[(size_t)b * max_len + t] : eos;
    if (threadIdx.x == 0) logprobs[b] = n > 0 ? st.hyp_score[b] : -1e5f;
}

__global__ void beam_init_kernel(BeamState st, int B, int beams, int max_len, int cls) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < B * beams) {
        st.ids[0][(size_t)r * max_len] = cls;
        st.words[r] = cls;
        st.beam_scores[r] = (r % beams == 0) ? 0.f : -1e9f;                 // model.py:508-509
        st.src_rows[r] = r;
    }
    if (r < B) { st.done[r] = 0; st.hyp_len[r] = 0; st.hyp_score[r] = 0.f; }
}

// ---- argmax: one block per row; lowest index wins ties (torch.argmax on CPU) -------------------
__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ logits, int ld, int V,
                                                     int64_t* __restrict__ out, int ld_out,
                                                     int32_t* __restrict__ sep_cnt, int step, int sep_id) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int r = blockIdx.x, tid = threadIdx.x;
    const float* p = logits + (size_t)r * ld;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < V; i += 256) {
        const float v = p[i];
        if (v > best || (v == best && i < bi)) { best = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float v2 = __shfl_xor(best, o);
        const int i2 = __shfl_xor(bi, o);
        if (v2 > best || (v2 == best && i2 < bi)) { best = v2; bi = i2; }
    }
    if ((tid & 63) == 0) { sv[tid >> 6] = best; si[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
        if (bi == 0x7fffffff) bi = 0;                               // all-NaN row: deterministic answer
        out[(size_t)r * ld_out] = bi;
        if (sep_cnt && bi == sep_id) atomicAdd(&sep_cnt[step], 1);
    }
}

// steps_out = number of generated columns that are valid under the stop rule
__global__ void finish_steps_kernel(const int32_t* sep_cnt, int rows, int max_len, int stop, int32_t* steps_out) {
    int steps = max_len;
    if (stop == 1) {
        for (int t = 0; t < max_len; ++t)
            if (sep_cnt[t] == rows) { steps = t + 1; break; }
    }
    *steps_out = steps;
}

__global__ void fill_i64_kernel(int64_t* p, int ld, int rows, int64_t v) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rows) p[(size_t)r * ld] = v;
}

// dst[l][r][t][:] = src[l][src_rows[r]][t][:] for t < t_len, every layer l = blockIdx.y in one launch
// (text K/V rows, `width` bf16 per position, `layer_stride` elements between layers)
__global__ void gather_txt_rows_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                       const int32_t* __restrict__ src_rows, int t_len, int Tmax, int width, size_t layer_stride) {
    const int r = blockIdx.x, sr = src_rows[r];
    const int n8 = t_len * width / 8;
    const uint4* s = (const uint4*)(src + blockIdx.y * layer_stride + (size_t)sr * Tmax * width);
    uint4* d = (uint4*)(dst + blockIdx.y * layer_stride + (size_t)r * Tmax * width);
    for (int i = threadIdx.x; i < n8; i += blockDim.x) d[i] = s[i];
}

}  // namespace

hipError_t launch_layernorm(const LnArgs& a, hipStream_t s) {
    if (a.rows <= 0 || a.D % 4 || a.D > 1024) return hipErrorInvalidValue;
    const int grid = (a.rows + 3) / 4;
    const int nv = (a.D + 255) / 256;
    const bool canon = a.D == 64 || a.D == 128 || a.D == 256 || a.D == 512 || a.D == 768 || a.D == 1024;
    if (a.gamma2 && (!canon || !a.beta2 || !a.out_bf16)) return hipErrorInvalidValue;
    switch (nv * 2 + (canon ? 1 : 0)) {
        case 2: hipLaunchKernelGGL((layernorm_kernel<1, false>), dim3(grid), dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL((layernorm_kernel<1, true>), dim3(grid), dim3(256), 0, s, a); break;
        case 4: hipLaunchKernelGGL((layernorm_kernel<2, false>), dim3(grid), dim3(256), 0, s, a); break;
        case 5: hipLaunchKernelGGL((layernorm_kernel<2, true>), dim3(grid), dim3(256), 0, s, a); break;
        case 6: hipLaunchKernelGGL((layernorm_kernel<3, false>), dim3(grid), dim3(256), 0, s, a); break;
        case 7: hipLaunchKernelGGL((layernorm_kernel<3, true>), dim3(grid), dim3(256), 0, s, a); break;
        case 8: hipLaunchKernelGGL((layernorm_kernel<4, false>), dim3(grid), dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((layernorm_kernel<4, true>), dim3(grid), dim3(256), 0, s, a); break;
    }
    return hipGetLastError();
}

hipError_t launch_ln_reduce(const float* slabs, int nslab, const float* bias, const float* resid, const float* gamma,
                            const float* beta, float eps, int M, int D, float* xf, bf16_t* xb, hipStream_t s) {
    if (M <= 0 || D % 4 || D > 1024 || nslab < 1 || nslab > 64) return hipErrorInvalidValue;
    const int nv = (D + 255) / 256;
    const dim3 block(64 * ((nslab + 7) / 8));
    if (nv == 1) hipLaunchKernelGGL(ln_reduce_kernel<1>, dim3(M), block, 0, s, slabs, nslab, bias, resid, gamma, beta, eps, M, D, xf, xb);
    else if (nv == 2) hipLaunchKernelGGL(ln_reduce_kernel<2>, dim3(M), block, 0, s, slabs, nslab, bias, resid, gamma, beta, eps, M, D, xf, xb);
    else if (nv == 3) hipLaunchKernelGGL(ln_reduce_kernel<3>, dim3(M), block, 0, s, slabs, nslab, bias, resid, gamma, beta, eps, M, D, xf, xb);
    else hipLaunchKernelGGL(ln_reduce_kernel<4>, dim3(M), block, 0, s, slabs, nslab, bias, resid, gamma, beta, eps, M, D, xf, xb);
    return hipGetLastError();
}

hipError_t launch_pack_frags(const void* src, void* dst, int rows16, int K, int elem_bytes, hipStream_t s) {
    if (rows16 <= 0 || rows16 % 16 || K % 32 || (elem_bytes != 1 && elem_bytes != 2)) return hipErrorInvalidValue;
    const int64_t total8 = (int64_t)rows16 * K / 8;
    const dim3 grid((unsigned)((total8 + 255) / 256));
    if (elem_bytes == 2) hipLaunchKernelGGL(pack_frags_kernel<unsigned short>, grid, dim3(256), 0, s, (const unsigned short*)src, (unsigned short*)dst, K, total8);
    else hipLaunchKernelGGL(pack_frags_kernel<unsigned char>, grid, dim3(256), 0, s, (const unsigned char*)src, (unsigned char*)dst, K, total8);
    return hipGetLastError();
}

hipError_t launch_argmax_final(const float* amax_val, const int* amax_idx, int ntiles, int rows, int row_stride, int row_off,
                               int64_t* out, int ld_out, int32_t* sep_cnt, int step, int sep_id, hipStream_t s, const NextEmbed* emb) {
    const NextEmbed e = emb ? *emb : NextEmbed{};
    const int nv = emb ? (e.D + 255) / 256 : 0;
    if (emb && (nv < 1 || nv > 4 || (e.D & 3) || !e.word || !e.pos || !e.gamma || !e.beta || !e.xf || !e.xb)) return hipErrorInvalidValue;
#define AF_LAUNCH(NV) hipLaunchKernelGGL(argmax_final_kernel<NV>, dim3(rows), dim3(256), 0, s, amax_val, amax_idx, ntiles, row_stride, row_off, \
                                         out, ld_out, sep_cnt, step, sep_id, e)
    switch (nv) {
        case 0: AF_LAUNCH(0); break;
        case 1: AF_LAUNCH(1); break;
        case 2: AF_LAUNCH(2); break;
        case 3: AF_LAUNCH(3); break;
        default: AF_LAUNCH(4); break;
    }
#undef AF_LAUNCH
    return hipGetLastError();
}

size_t beam_topk_scratch_bytes(int B, int beams, int V, int K) {
    const size_t n = (size_t)B * beams * ((V + BT_CHUNK - 1) / BT_CHUNK);
    return n * 8 + n * K * 8;                          // chunk (max, sum) + K (value, index) candidates per chunk
}

hipError_t launch_beam_topk(const float* logits, int ld, const float* beam_scores, int B, int beams, int V, int K,
                            float* out_scores, int* out_idx, void* scratch, hipStream_t s) {
    const int nch = (V + BT_CHUNK - 1) / BT_CHUNK;
    if (B <= 0 || beams <= 0 || beams > 16 || K <= 0 || K > 16 || K > beams * V || nch > 64 || !scratch) return hipErrorInvalidValue;
    const size_t n = (size_t)B * beams * nch;
    float2* stats = (float2*)scratch;
    float* cval = (float*)(stats + n);
    int* cidx = (int*)(cval + n * K);
    if (K <= 8) {
        hipLaunchKernelGGL(beam_topk_chunks_kernel<8>, dim3((unsigned)n), dim3(256), 0, s, logits, ld, V, K, nch, stats, cval, cidx);
        hipLaunchKernelGGL(beam_topk_merge_kernel<8>, dim3(B), dim3(64), 0, s, stats, cval, cidx, beam_scores, beams, V, K, nch, out_scores, out_idx);
    } else {
        hipLaunchKernelGGL(beam_topk_chunks_kernel<16>, dim3((unsigned)n), dim3(256), 0, s, logits, ld, V, K, nch, stats, cval, cidx);
        hipLaunchKernelGGL(beam_topk_merge_kernel<16>, dim3(B), dim3(64), 0, s, stats, cval, cidx, beam_scores, beams, V, K, nch, out_scores, out_idx);
    }
    return hipGetLastError();
}

hipError_t launch_beam_init(const BeamBuffers& bb, int B, int beams, int max_len, int cls, hipStream_t s) {
    BeamState st{{bb.ids0, bb.ids1}, bb.beam_scores, bb.words, bb.src_rows, bb.done, bb.hyp_len, bb.hyp_score, bb.hyp_ids};
    hipLaunchKernelGGL(beam_init_kernel, dim3((B * beams + 63) / 64), dim3(64), 0, s, st, B, beams, max_len, cls);
    return hipGetLastError();
}

hipError_t launch_beam_step(const BeamBuffers& bb, const float* cand_scores, const int* cand_idx, int B, int beams, int K,
                            int V, int cur_len, int max_len, int eos, float length_penalty, int cur, hipStream_t s) {
    if (beams > 16 || K > 16) return hipErrorInvalidValue;
    BeamState st{{bb.ids0, bb.ids1}, bb.beam_scores, bb.words, bb.src_rows, bb.done, bb.hyp_len, bb.hyp_score, bb.hyp_ids};
    hipLaunchKernelGGL(beam_step_kernel, dim3(B), dim3(64), 0, s, st, cand_scores, cand_idx, beams, K, V, cur_len, max_len, eos,
                       length_penalty, cur);
    return hipGetLastError();
}

hipError_t launch_beam_finish(const BeamBuffers& bb, int B, int max_len, int eos, int64_t* decoded, float* logprobs, hipStream_t s) {
    BeamState st{{bb.ids0, bb.ids1}, bb.beam_scores, bb.words, bb.src_rows, bb.done, bb.hyp_len, bb.hyp_score, bb.hyp_ids};
    hipLaunchKernelGGL(beam_finish_kernel, dim3(B), dim3(64), 0, s, st, max_len, eos, decoded, logprobs);
    return hipGetLastError();
}

hipError_t launch_im2col(const float* frames, bf16_t* patches, int nf, int img, int p, int Kp, hipStream_t s) {
    const int G = img / p;
    if (p % 4 == 0 && img % 4 == 0) {
        const int64_t total = (int64_t)nf * G * G * (Kp / 4);
        hipLaunchKernelGGL(im2col_kernel<4>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, frames, patches, nf, img, p, Kp);
    } else if (p % 2 == 0) {
        const int64_t total = (int64_t)nf * G * G * (Kp / 2);
        hipLaunchKernelGGL(im2col_kernel<2>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, frames, patches, nf, img, p, Kp);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}



// ---- V of the image prefix as OCP e4m3 codes + one power-of-two scale per (token, head) (opt-in kv_cache = v_e4m3) -------
// kv: [rows][3D] bf16 (q | k | v of one decoder layer's image rows) -> v8 [H][pitch][64] codes, vs [H][pitch] scales, HEAD-MAJOR: the
// V stream of one (row, head) unit of txt_block is one contiguous run of 64-byte records (two keys per cache line).  Eight lanes per
// (row, head): 8 values each; scale = the smallest 2^e with amax <= 448 * 2^e (exact: amax / 2^(E-8) lies in [256, 512), compared
// with 448 after an exact scaling), codes round to nearest even; code * scale is a bf16 value.
namespace {
__global__ __launch_bounds__(256) void kv_quant_v_kernel(const bf16_t* __restrict__ kv, unsigned char* __restrict__ v8, float* __restrict__ vs,
                                                         int64_t groups, int D, int H, int64_t pitch) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t gi = t >> 3;
    const int sub = (int)(t & 7);
    const bool live = gi < groups;
    const int64_t g = live ? gi : groups - 1;
    const int64_t row = g / H;
    const int head = (int)(g - row * H);
    const bf16x8 x = *(const bf16x8*)(kv + row * 3 * D + 2 * D + head * 64 + sub * 8);
    float f[8], amax = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) { f[d] = bf2f((bf16_t)x[d]); amax = fmaxf(amax, fabsf(f[d])); }
    amax = fmaxf(amax, __shfl_xor(amax, 1));
    amax = fmaxf(amax, __shfl_xor(amax, 2));
    amax = fmaxf(amax, __shfl_xor(amax, 4));
    float scale = 1.0f;
    if (amax > 0.f) {
        const int E = (int)((__float_as_uint(amax) >> 23) & 0xff) - 127;          // floor(log2 amax) (bf16 values: never denormal in fp32)
        scale = __uint_as_float((unsigned)(E - 8 + 127) << 23);                    // amax / scale in [256, 512)
        if (amax > 448.0f * scale) scale *= 2.0f;
    }
    const float inv = 1.0f / scale;                                                // a power of two: exact
    uint2 q;
    q.x = pack_fp8x4(f[0] * inv, f[1] * inv, f[2] * inv, f[3] * inv);
    q.y = pack_fp8x4(f[4] * inv, f[5] * inv, f[6] * inv, f[7] * inv);
    if (live) {
        *(uint2*)(v8 + ((int64_t)head * pitch + row) * 64 + sub * 8) = q;
        if (sub == 0) vs[(int64_t)head * pitch + row] = scale;
    }
}
}  // namespace

hipError_t launch_kv_quant_v(const bf16_t* kv, unsigned char* v8, float* vs, int rows, int D, int H, int64_t pitch, hipStream_t s) {
    if (rows <= 0 || H * 64 != D || pitch < rows) return hipErrorInvalidValue;
    const int64_t groups = (int64_t)rows * H;
    hipLaunchKernelGGL(kv_quant_v_kernel, dim3((unsigned)((groups * 8 + 255) / 256)), dim3(256), 0, s, kv, v8, vs, groups, D, H, pitch);
    return hipGetLastError();
}

hipError_t launch_cast_bf16(const float* in, bf16_t* out, int64_t n, hipStream_t s) {
    if (n % 4) return hipErrorInvalidValue;
    const int64_t n4 = n / 4;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, in, out, n4);
    return hipGetLastError();
}

hipError_t launch_gather_hidden(const float* img, const float* txt, float* out, int n_entries, int B, int S_img, int T, int D,
                                size_t img_entry_stride, size_t txt_entry_stride, hipStream_t s) {
    if (B <= 0 || n_entries <= 0 || S_img <= 0 || T <= 0 || D % 4) return hipErrorInvalidValue;
    const size_t rows = (size_t)B * n_entries * (S_img + T);
    hipLaunchKernelGGL(gather_hidden_kernel, dim3((unsigned)rows), dim3(256), 0, s, img, txt, out, n_entries, S_img, T, D,
                       img_entry_stride, txt_entry_stride);
    return hipGetLastError();
}

hipError_t launch_dequant_fp8_batch(const DequantBatch& b, hipStream_t s) {
    if (b.n <= 0 || b.n > 4) return hipErrorInvalidValue;
    int64_t mx = 0;
    for (int i = 0; i < b.n; ++i) {
        if ((b.K[i] & 15) || !b.w8[i] || !b.scale[i] || !b.out[i]) return hipErrorInvalidValue;
        mx = std::max(mx, b.n16[i]);
    }
    hipLaunchKernelGGL(dequant_fp8_batch_kernel, dim3((unsigned)((mx + 255) / 256), (unsigned)b.n), dim3(256), 0, s, b);
    return hipGetLastError();
}

hipError_t launch_dequant_fp8(const unsigned char* w8, const float* scale, bf16_t* out, int rows, int K, hipStream_t s) {
    if (rows <= 0 || K <= 0 || K % 16) return hipErrorInvalidValue;
    const int64_t n16 = (int64_t)rows * K / 16;
    hipLaunchKernelGGL(dequant_fp8_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, s, w8, scale, out, K, n16);
    return hipGetLastError();
}

hipError_t launch_embed_text(const int64_t* ids, int ld_ids, int rows, int T, int t0, const float* word,
                             const float* pos, const float* gamma, const float* beta, float eps, int D, int vocab,
                             float* x_f32, bf16_t* x_bf16, hipStream_t s) {
    const int grid = (rows * T + 3) / 4;
    const int nv = (D + 255) / 256;
    if (nv == 1) hipLaunchKernelGGL(embed_text_kernel<1>, dim3(grid), dim3(256), 0, s, ids, ld_ids, rows, T, t0, word, pos, gamma, beta, eps, D, vocab, x_f32, x_bf16);
    else if (nv == 2) hipLaunchKernelGGL(embed_text_kernel<2>, dim3(grid), dim3(256), 0, s, ids, ld_ids, rows, T, t0, word, pos, gamma, beta, eps, D, vocab, x_f32, x_bf16);
    else if (nv == 3) hipLaunchKernelGGL(embed_text_kernel<3>, dim3(grid), dim3(256), 0, s, ids, ld_ids, rows, T, t0, word, pos, gamma, beta, eps, D, vocab, x_f32, x_bf16);
    else hipLaunchKernelGGL(embed_text_kernel<4>, dim3(grid), dim3(256), 0, s, ids, ld_ids, rows, T, t0, word, pos, gamma, beta, eps, D, vocab, x_f32, x_bf16);
    return hipGetLastError();
}

hipError_t launch_argmax(const float* logits, int ld, int rows, int V, int64_t* out, int ld_out,
                         int32_t* sep_flags, int step, int sep_id, hipStream_t s) {
    hipLaunchKernelGGL(argmax_kernel, dim3(rows), dim3(256), 0, s, logits, ld, V, out, ld_out, sep_flags, step, sep_id);
    return hipGetLastError();
}

hipError_t launch_finish_steps(const int32_t* sep_cnt, int rows, int max_len, int stop, int32_t* steps_out, hipStream_t s) {
    hipLaunchKernelGGL(finish_steps_kernel, dim3(1), dim3(1), 0, s, sep_cnt, rows, max_len, stop, steps_out);
    return hipGetLastError();
}

hipError_t launch_fill_i64(int64_t* p, int ld, int rows, int64_t v, hipStream_t s) {
    hipLaunchKernelGGL(fill_i64_kernel, dim3((rows + 63) / 64), dim3(64), 0, s, p, ld, rows, v);
    return hipGetLastError();
}

hipError_t launch_gather_txt_rows(const bf16_t* src, bf16_t* dst, const int32_t* src_rows, int rows,
                                  int t_len, int Tmax, int width, int layers, size_t layer_stride, hipStream_t s) {
    if (rows <= 0 || layers <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gather_txt_rows_kernel, dim3(rows, layers), dim3(256), 0, s, src, dst, src_rows, t_len, Tmax, width, layer_stride);
    return hipGetLastError();
}
