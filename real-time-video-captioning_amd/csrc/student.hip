// Student caption decoder (SURVEY.md par. 8 row f.2): the reference's StudentCandidateV1 decoder
// (src/models/model.py:50-187 -- nn.TransformerDecoder, post-LN, ReLU, causal + key-padding mask,
// cross-attention over one memory token per frame) with an exact KV cache, behind the C ABI declared in
// include/gitcap.h ("student decoder" section).  The TinyViT image encoder is outside this path.
//
// Everything here is the decode-loop regime of the GIT text path: M = rows x T is a handful of rows, so
// the dense layers are the weight-streaming skinny GEMMs (skinny.hip), the LayerNorms are fused with the
// split-K reduction (ln_reduce_kernel), and the two attentions (<= 64 keys) are one wave per (row, head).
//
// Exact KV cache: the self-attention K/V of position j depend only on tokens <= j and on the PAD flags
// of tokens <= j, so the reference's full recompute per step (model.py:173-177) and the cached loop agree.
#include "../../include/gitcap.h"
#include "kernels.h"
#include "host_util.h"

#include <cmath>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

namespace {

// ---- kernels -------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void student_embed_kernel(const int64_t* __restrict__ ids, int ld_ids, int T, int t0,
                                                           const float* __restrict__ embed, const float* __restrict__ pe,
                                                           int D, int vocab, float sqrt_d, float* __restrict__ xf,
                                                           bf16_t* __restrict__ xb) {
    const int m = blockIdx.x, r = m / T, pos = t0 + m % T;
    long long id = ids[(size_t)r * ld_ids + pos];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const float* e = embed + (size_t)id * D;
    const float* p = pe + (size_t)pos * D;
    for (int c = threadIdx.x * 4; c < D; c += 256) {
        const f32x4 a = *(const f32x4*)(e + c), b = *(const f32x4*)(p + c);
        f32x4 y;
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = (a[i] + b[i]) / sqrt_d;      // model.py:142-144: (embed + pe) / sqrt(D)
        *(f32x4*)(xf + (size_t)m * D + c) = y;
        uint2 o;
        o.x = pack_bf2(y[0], y[1]);
        o.y = pack_bf2(y[2], y[3]);
        *(uint2*)(xb + (size_t)m * D + c) = o;
    }
}

// one wave per (query, head); lane i owns key i for the scores and output column(s) lane, lane + 64 for P.V
__global__ __launch_bounds__(64) void attn_small_kernel(SmallAttnArgs a) {
    __shared__ float qs[128];
    __shared__ float ps[64];
    const int lane = threadIdx.x, m = blockIdx.x, h = blockIdx.y;
    const int r = m / a.T, j = m % a.T;
    const int nk = a.nkeys > 0 ? a.nkeys : a.t0 + j + 1;
    const bf16_t* q = a.q + (size_t)(r * a.q_row_stride + a.q_row_off + j) * a.ldq + h * a.hd;
    for (int d = lane; d < a.hd; d += 64) qs[d] = bf2f(q[d]);
    __syncthreads();
    float s = -INFINITY;
    if (lane < nk) {
        const bf16_t* k = a.k + (size_t)(r * a.keys_stride + lane) * a.ldkv + h * a.hd;
        float acc = 0.f;
        for (int c = 0; c < a.hd; c += 8) {
            const bf16x8 kk = *(const bf16x8*)(k + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += qs[c + e] * bf2f((bf16_t)kk[e]);
        }
        const bool masked = a.ids && a.ids[(size_t)r * a.ld_ids + lane] == a.pad_id;
        s = masked ? -INFINITY : acc * rsqrtf((float)a.hd);
    }
    const float mx = wave_max(s);
    const float p = lane < nk ? __expf(s - mx) : 0.f;        // every key masked: -inf - -inf = NaN, like torch
    const float den = wave_sum(p);
    ps[lane] = p / den;
    __syncthreads();
    const bf16_t* v = a.v + (size_t)r * a.keys_stride * a.ldkv + h * a.hd;
    for (int d = lane; d < a.hd; d += 64) {
        float acc = 0.f;
        for (int i = 0; i < nk; ++i) acc += ps[i] * bf2f(v[(size_t)i * a.ldkv + d]);    // fixed order: batch invariant
        a.ctx[(size_t)m * a.ldc + h * a.hd + d] = f2bf(acc);
    }
}

struct StuLayer {
    const bf16_t *sa_in_w, *sa_out_w, *ca_in_w, *ca_out_w, *l1w, *l2w;
    const float *sa_in_b, *sa_out_b, *ca_in_b, *ca_out_b, *l1b, *l2b, *n1w, *n1b, *n2w, *n2b, *n3w, *n3b;
};

}  // namespace

hipError_t launch_attn_small(const SmallAttnArgs& a, hipStream_t s) {
    if (a.M <= 0 || a.H <= 0 || a.hd % 8 || a.hd > 128 || a.nkeys > 64 || (a.nkeys == 0 && a.t0 + a.T > 64)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(attn_small_kernel, dim3(a.M, a.H), dim3(64), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_student_embed(const int64_t* ids, int ld_ids, int rows, int T, int t0, const float* embed,
                                const float* pe, int D, int vocab, float* xf, bf16_t* xb, hipStream_t s) {
    if (rows <= 0 || T <= 0 || D % 4) return hipErrorInvalidValue;
    // torch.sqrt(torch.tensor(D)) of model.py:144 in fp32
    hipLaunchKernelGGL(student_embed_kernel, dim3(rows * T), dim3(64), 0, s, ids, ld_ids, T, t0, embed, pe, D, vocab,
                       sqrtf((float)D), xf, xb);
    return hipGetLastError();
}

// ---- device-resident beam search of the student (model.py:189-318): bookkeeping kernels ----------------------------------
// dst[(b * k + i)][:] = src[b][:]  (the k beams of a clip share its memory rows)
__global__ __launch_bounds__(256) void repeat_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int k, int n4) {
    const int r = blockIdx.y;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256)
        ((f32x4*)dst)[(size_t)r * n4 + i] = ((const f32x4*)src)[(size_t)(r / k) * n4 + i];
}
// beam scores before the first step: only beam 0 of every clip is live (all k rows hold the same prefix [CLS])
__global__ void beam_scores_init_kernel(float* scores, int rows, int k) {
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r < rows) scores[r] = (r % k == 0) ? 0.f : -1e9f;
}
// one step of model.py:252-287: candidate j of clip b (sorted best first by beam_topk; flat index = beam * V + token) becomes
// row b * k + j: its prefix is the prefix of row b * k + beam, its new token the candidate's, its score the candidate's
__global__ __launch_bounds__(64) void student_beam_step_kernel(const float* __restrict__ cand_scores, const int* __restrict__ cand_idx,
                                                               const int64_t* __restrict__ ids_cur, int64_t* __restrict__ ids_next,
                                                               float* __restrict__ scores, int32_t* __restrict__ src_rows,
                                                               int k, int V, int t, int ld) {
    const int b = blockIdx.x, lane = threadIdx.x;
    for (int j = 0; j < k; ++j) {
        const int idx = cand_idx[b * k + j];
        const int beam = idx / V, tok = idx - beam * V;
        const int src = b * k + beam, dst = b * k + j;
        for (int i = lane; i <= t; i += 64) ids_next[(size_t)dst * ld + i] = ids_cur[(size_t)src * ld + i];
        if (lane == 0) {
            ids_next[(size_t)dst * ld + t + 1] = tok;
            scores[dst] = cand_scores[b * k + j];
            src_rows[dst] = src;
        }
    }
}
// the best beam of every clip (row b * k: the candidates arrive sorted) -> out [B][max_len]
__global__ void student_beam_finish_kernel(const int64_t* __restrict__ ids, int64_t* __restrict__ out, int k, int ld, int max_len) {
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < max_len; i += 64) out[(size_t)b * max_len + i] = ids[(size_t)b * k * ld + i];
}

struct gitcap_student {
    gitcap_student_config c;
    int device = 0;
    mutable std::string err;
    std::map<std::string, DevTensor> w;
    bool finalized = false, have_memory = false;
    int D = 0, H = 0, hd = 0, FF = 0, L = 0, V = 0, F = 0, R = 0, Tmax = 0, Mt = 0, cur_B = 0;
    std::vector<void*> allocs;
    int64_t ws_bytes = 0;
    // workspace: text rows
    float *xf = nullptr, *xf2 = nullptr, *slabs = nullptr, *amax_val = nullptr;
    int* amax_idx = nullptr;
    bf16_t *xb = nullptr, *qc = nullptr, *ctx = nullptr, *ffn = nullptr, *kvs = nullptr, *memb = nullptr, *memkv = nullptr;
    int32_t* sep_cnt = nullptr;
    // greedy loop captured as a hipGraph (one per (B, max_len, stop, row-prologue switch)); it works on the handle's own
    // ids / steps buffers, which are copied to the caller's after the replay
    int64_t* g_ids = nullptr;
    int32_t* g_steps = nullptr;
    struct GreedyGraph { int B, max_len, stop; bool rows_pro, head_share; hipGraphExec_t exec; };
    std::vector<GreedyGraph> graphs;
    hipStream_t cap_stream = nullptr;   // capture only (the legacy default stream cannot be captured); replays run on the caller's stream
    // device-resident beam search (gitcap_student_beam_search), allocated on first use
    struct BeamWs {
        float *memrep = nullptr, *scores = nullptr, *cand_scores = nullptr, *logits = nullptr;
        int* cand_idx = nullptr; int32_t* src_rows = nullptr;
        int64_t *ids0 = nullptr, *ids1 = nullptr;
        bf16_t* kvs2 = nullptr; char* topk_scratch = nullptr;
    } bw;
    // resolved weights
    const float *embed = nullptr, *pe = nullptr, *head_b = nullptr;
    const bf16_t* head_w = nullptr;
    std::vector<StuLayer> layers;
};

namespace {

std::string g_student_create_err;

int sfail(const gitcap_student* h, int code, const std::string& msg) {
    if (h) h->err = msg; else g_student_create_err = msg;
    return code;
}

#define S_HIP_OK(h, expr)                                                                             \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return sfail(h, GITCAP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));       \
    } while (0)
#define S_GUARD(h) DeviceGuard guard_((h)->device); if (!guard_.ok) return sfail(h, GITCAP_ERR_HIP, "cannot select the handle's device")

bool student_is_gemm_weight(const std::string& n) {
    auto ends = [&](const char* s) { size_t l = strlen(s); return n.size() >= l && n.compare(n.size() - l, l, s) == 0; };
    return n == "linear.weight" || ends("in_proj_weight") || ends("out_proj.weight") || ends("linear1.weight") || ends("linear2.weight");
}

// the reference's state_dict keys (gitcap/student_config.py: student_shapes)
void expected_shapes(const gitcap_student_config& c, std::vector<std::pair<std::string, std::vector<int64_t>>>& out) {
    const int64_t D = c.d_model, FF = c.d_ffn, V = c.vocab_size;
    auto add = [&](const std::string& n, std::vector<int64_t> s) { out.emplace_back(n, std::move(s)); };
    add("embed.weight", {V, D});
    add("pos_enc.pe", {1, c.max_pos, D});
    for (int i = 0; i < c.num_layers; ++i) {
        const std::string p = "decoder.layers." + std::to_string(i) + ".";
        for (const char* att : {"self_attn", "multihead_attn"}) {
            add(p + att + ".in_proj_weight", {3 * D, D}); add(p + att + ".in_proj_bias", {3 * D});
            add(p + att + ".out_proj.weight", {D, D}); add(p + att + ".out_proj.bias", {D});
        }
        add(p + "linear1.weight", {FF, D}); add(p + "linear1.bias", {FF});
        add(p + "linear2.weight", {D, FF}); add(p + "linear2.bias", {D});
        for (const char* n : {"norm1", "norm2", "norm3"}) { add(p + n + ".weight", {D}); add(p + n + ".bias", {D}); }
    }
    add("linear.weight", {V, D});
    add("linear.bias", {V});
}

template <typename T>
int s_alloc(gitcap_student* h, T** p, size_t count) {
    void* q = nullptr;
    const size_t bytes = count * sizeof(T);
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) return sfail(h, GITCAP_ERR_NOMEM, std::string("hipMalloc workspace: ") + hipGetErrorString(e));
    e = hipMemset(q, 0, bytes);
    if (e != hipSuccess) return sfail(h, GITCAP_ERR_HIP, std::string("hipMemset workspace: ") + hipGetErrorString(e));
    h->allocs.push_back(q);
    h->ws_bytes += (int64_t)bytes;
    *p = (T*)q;
    return 0;
}

int sk_full(gitcap_student* h, hipStream_t s, int epi, const bf16_t* X, int ldx, const bf16_t* W, const float* bias, int M,
            int N, int K, void* out, int ldo, int T = 1, int row_stride = 1, int row_off = 0) {
    SkinnyArgs a{};
    a.X = X; a.ldx = ldx; a.W = W; a.bias = bias; a.M = M; a.N = N; a.K = K; a.out = out; a.ldo = ldo;
    a.T = T; a.row_stride = row_stride; a.row_off = row_off;
    S_HIP_OK(h, launch_skinny(a, epi, s));
    return 0;
}

// model.py:128-154 for rows x T query positions t0..t0+T-1 (K/V of earlier positions come from the cache)
int text_forward(gitcap_student* h, const int64_t* ids, int ld_ids, int rows, int t0, int T, float* logits_out,
                 int64_t* argmax_out, int ld_argmax, int32_t* sep_cnt, int step, hipStream_t s) {
    const gitcap_student_config& c = h->c;
    if (!h->finalized) return sfail(h, GITCAP_ERR_STATE, "student: weights not finalized");
    if (!h->have_memory) return sfail(h, GITCAP_ERR_STATE, "student: decoder called before set_memory");
    if (!ids || rows <= 0 || T <= 0 || t0 < 0) return sfail(h, GITCAP_ERR_ARG, "student: bad arguments");
    if (rows != h->cur_B) return sfail(h, GITCAP_ERR_ARG, "student: rows != rows of the current memory");
    if (t0 + T > h->Tmax) return sfail(h, GITCAP_ERR_ARG, "student: t0+T exceeds max_text_len+1");
    if (t0 + T > c.max_pos) return sfail(h, GITCAP_ERR_ARG, "student: position exceeds the positional table");
    const int D = h->D, M = rows * T;
    int rc;
    S_HIP_OK(h, launch_student_embed(ids, ld_ids, rows, T, t0, h->embed, h->pe, D, h->V, h->xf, h->xb, s));
    const size_t kvs_layer = (size_t)h->R * h->Tmax * 3 * D, mem_layer = (size_t)h->R * h->F * 2 * D;
    // Each post-LN sub-layer is split-K partial slabs -> sum + bias + residual + LayerNorm.  With one or two rows (the webcam
    // case) that row kernel is not launched: the projection that consumes its output computes the rows itself (skinny.hip
    // "row prologue", same code -> same bits) and workgroup 0 writes the fp32 residual rows to the other of two buffers.
    const bool rows_pro = g_row_prologue && skinny_row_prologue_ok(M, D, false);
    float *xcur = h->xf, *xalt = h->xf2;
    struct { bool on; const float *bias, *g, *b; int nslab; } pend{false, nullptr, nullptr, nullptr, 0};
    // out = epi(LN-output . W^T + bias): the LN output is xb, or -- when a reduce + LayerNorm is pending -- computed in place
    auto proj = [&](int epi, const bf16_t* W, const float* bias, int N, void* out, int ldo, int Tq, int row_stride, int row_off) -> int {
        if (!pend.on) return sk_full(h, s, epi, h->xb, D, W, bias, M, N, D, out, ldo, Tq, row_stride, row_off);
        SkinnyArgs a{};
        a.X = h->xb; a.ldx = D; a.W = W; a.bias = bias; a.M = M; a.N = N; a.K = D; a.out = out; a.ldo = ldo;
        a.T = Tq; a.row_stride = row_stride; a.row_off = row_off;
        a.ln.kind = 1; a.ln.slabs = h->slabs; a.ln.nslab = pend.nslab; a.ln.bias = pend.bias; a.ln.resid = xcur;
        a.ln.g = pend.g; a.ln.b = pend.b; a.ln.eps = h->c.ln_eps; a.ln.xf = xalt;
        S_HIP_OK(h, launch_skinny(a, epi, s));
        std::swap(xcur, xalt);
        pend.on = false;
        return 0;
    };
    // x = LayerNorm(x + X.W^T + bias): split-K partial slabs, then the row kernel now or (defer) inside the next projection
    auto dense_ln = [&](const bf16_t* X, int K, const bf16_t* W, const float* bias, const float* g, const float* b, bool defer) -> int {
        SkinnyArgs a{};
        a.X = X; a.ldx = K; a.W = W; a.M = M; a.N = D; a.K = K; a.out = h->slabs; a.ldo = D; a.T = 1; a.row_stride = 1;
        S_HIP_OK(h, launch_skinny_splitk(a, s));
        if (rows_pro && defer) { pend = {true, bias, g, b, skinny_ksplit(K)}; return 0; }
        S_HIP_OK(h, launch_ln_reduce(h->slabs, skinny_ksplit(K), bias, xcur, g, b, h->c.ln_eps, M, D, xcur, h->xb, s));
        return 0;
    };
    for (int l = 0; l < h->L; ++l) {
        const StuLayer& Ly = h->layers[l];
        bf16_t* kv = h->kvs + (size_t)l * kvs_layer;
        // self-attention: q | k | v of the new positions go straight into the cache rows (r, t0 + j)
        if ((rc = proj(SK_BIAS_BF16, Ly.sa_in_w, Ly.sa_in_b, 3 * D, kv, 3 * D, T, h->Tmax, t0))) return rc;
        SmallAttnArgs sa{kv, 3 * D, T, h->Tmax, t0, kv + D, kv + 2 * D, 3 * D, h->Tmax, 0, t0,
                         ids, ld_ids, c.pad_token_id, h->ctx, D, M, h->H, h->hd};
        S_HIP_OK(h, launch_attn_small(sa, s));
        if ((rc = dense_ln(h->ctx, D, Ly.sa_out_w, Ly.sa_out_b, Ly.n1w, Ly.n1b, true))) return rc;
        // cross-attention over the frame tokens (K/V precomputed by set_memory)
        if ((rc = proj(SK_BIAS_BF16, Ly.ca_in_w, Ly.ca_in_b, D, h->qc, D, 1, 1, 0))) return rc;
        const bf16_t* mkv = h->memkv + (size_t)l * mem_layer;
        SmallAttnArgs ca{h->qc, D, T, T, 0, mkv, mkv + D, 2 * D, h->F, h->F, 0, nullptr, 0, 0, h->ctx, D, M, h->H, h->hd};
        S_HIP_OK(h, launch_attn_small(ca, s));
        if ((rc = dense_ln(h->ctx, D, Ly.ca_out_w, Ly.ca_out_b, Ly.n2w, Ly.n2b, true))) return rc;
        // feed-forward (the last layer's LayerNorm is a launch: the vocabulary head reads its bf16 output)
        if ((rc = proj(SK_BIAS_RELU_BF16, Ly.l1w, Ly.l1b, h->FF, h->ffn, h->FF, 1, 1, 0))) return rc;
        if ((rc = dense_ln(h->ffn, h->FF, Ly.l2w, Ly.l2b, Ly.n3w, Ly.n3b, l + 1 < h->L))) return rc;
    }
    if (!logits_out && !argmax_out) return 0;
    // vocabulary head: all positions when logits are requested, else the last position of every row
    const int V = h->V, ntiles = (V + 15) / 16;
    SkinnyArgs ha{};
    ha.W = h->head_w; ha.bias = h->head_b; ha.N = V; ha.K = D; ha.ldo = V; ha.T = 1; ha.row_stride = 1; ha.row_off = 0;
    int am_stride = 1, am_off = 0;
    if (logits_out) {
        ha.X = h->xb; ha.ldx = D; ha.M = M; ha.out = logits_out;
        am_stride = T; am_off = T - 1;
    } else {
        ha.X = h->xb + (size_t)(T - 1) * D; ha.ldx = T * D; ha.M = rows;
    }
    if (argmax_out) { ha.amax_val = h->amax_val; ha.amax_idx = h->amax_idx; }
    S_HIP_OK(h, launch_skinny(ha, SK_BIAS_F32, s));
    if (argmax_out)
        S_HIP_OK(h, launch_argmax_final(h->amax_val, h->amax_idx, ntiles, rows, am_stride, am_off, argmax_out, ld_argmax,
                                        sep_cnt, step, c.sep_token_id, s));
    return 0;
}

int set_memory(gitcap_student* h, const float* memory, int B, hipStream_t s) {
    if (!h->finalized) return sfail(h, GITCAP_ERR_STATE, "student: weights not finalized");
    if (!memory || B <= 0 || B > h->R) return sfail(h, GITCAP_ERR_ARG, "student: set_memory: bad arguments / B exceeds max_rows");
    const int D = h->D, Mm = B * h->F;
    S_HIP_OK(h, launch_cast_bf16(memory, h->memb, (int64_t)Mm * D, s));
    const size_t mem_layer = (size_t)h->R * h->F * 2 * D;
    for (int l = 0; l < h->L; ++l) {
        const StuLayer& Ly = h->layers[l];
        // k | v = memory . W[D:3D]^T + b[D:3D]   (rows D..3D of the cross-attention in_proj)
        int rc = sk_full(h, s, SK_BIAS_BF16, h->memb, D, Ly.ca_in_w + (size_t)D * D, Ly.ca_in_b + D, Mm, 2 * D, D,
                         h->memkv + (size_t)l * mem_layer, 2 * D);
        if (rc) return rc;
    }
    h->cur_B = B;
    h->have_memory = true;
    return 0;
}

}  // namespace

extern "C" {

const char* gitcap_student_last_error(const gitcap_student_t* h) { return h ? h->err.c_str() : g_student_create_err.c_str(); }

int gitcap_student_create(const gitcap_student_config* cfg, int device, gitcap_student_t** out) {
    if (!cfg || !out) return sfail(nullptr, GITCAP_ERR_ARG, "student_create: null argument");
    const gitcap_student_config& c = *cfg;
    if (c.d_model <= 0 || c.n_head <= 0 || c.d_model % c.n_head || c.d_model % 32 || c.d_ffn % 32 || c.d_model > 1024)
        return sfail(nullptr, GITCAP_ERR_ARG, "student_create: d_model/d_ffn must be multiples of 32, d_model <= 1024 and divisible by n_head");
    const int hd = c.d_model / c.n_head;
    if (hd % 8 || hd > 128) return sfail(nullptr, GITCAP_ERR_ARG, "student_create: head_dim must be a multiple of 8, <= 128");
    if (!skinny_full_ok(c.d_model) || !skinny_ksplit(c.d_model) || !skinny_ksplit(c.d_ffn))
        return sfail(nullptr, GITCAP_ERR_ARG, "student_create: no skinny-GEMM instantiation for this d_model / d_ffn (d_model in {64,128,256,576,768,1024})");
    if (c.num_layers <= 0 || c.vocab_size <= 0 || c.mem_tokens <= 0 || c.mem_tokens > 64 || c.max_rows <= 0)
        return sfail(nullptr, GITCAP_ERR_ARG, "student_create: bad layer / vocabulary / memory sizes (mem_tokens <= 64)");
    if (c.max_text_len <= 0 || c.max_text_len + 1 > 64 || c.max_text_len + 1 > c.max_pos)
        return sfail(nullptr, GITCAP_ERR_ARG, "student_create: max_text_len must be in 1..63 and fit the positional table");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return sfail(nullptr, GITCAP_ERR_HIP, "student_create: no such HIP device (libgitcap has no CPU fallback)");
    gitcap_student* h = new gitcap_student();
    h->c = c; h->device = device;
    h->D = c.d_model; h->H = c.n_head; h->hd = hd; h->FF = c.d_ffn; h->L = c.num_layers; h->V = c.vocab_size;
    h->F = c.mem_tokens; h->R = c.max_rows; h->Tmax = c.max_text_len + 1; h->Mt = h->R * h->Tmax;
    std::vector<std::pair<std::string, std::vector<int64_t>>> shapes;
    expected_shapes(c, shapes);
    for (auto& kv : shapes) {
        DevTensor t;
        t.shape = kv.second;
        t.bf16 = student_is_gemm_weight(kv.first);
        h->w[kv.first] = t;
    }
    *out = h;
    return 0;
}

void gitcap_student_destroy(gitcap_student_t* h) {
    if (!h) return;
    DeviceGuard g(h->device);
    for (auto& kv : h->w)
        if (kv.second.p) (void)hipFree(kv.second.p);
    for (auto& g : h->graphs) (void)hipGraphExecDestroy(g.exec);
    if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
    for (void* p : h->allocs) (void)hipFree(p);
    delete h;
}

int gitcap_student_load_tensor(gitcap_student_t* h, const char* name, const float* data, const int64_t* shape, int rank) {
    if (!h || !name || !data || !shape) return sfail(h, GITCAP_ERR_ARG, "student_load_tensor: null argument");
    auto it = h->w.find(name);
    if (it == h->w.end()) return sfail(h, GITCAP_ERR_ARG, std::string("student_load_tensor: unknown tensor '") + name + "'");
    S_GUARD(h);
    DevTensor& t = it->second;
    if ((int)t.shape.size() != rank) return sfail(h, GITCAP_ERR_ARG, std::string("student_load_tensor: rank mismatch for ") + name);
    int64_t rows = 1;
    for (int i = 0; i < rank; ++i) {
        if (t.shape[i] != shape[i]) return sfail(h, GITCAP_ERR_ARG, std::string("student_load_tensor: shape mismatch for ") + name);
        if (i + 1 < rank) rows *= shape[i];
    }
    const int64_t cols = shape[rank - 1];
    if (t.p) { (void)hipFree(t.p); t.p = nullptr; }
    if (t.bf16) {   // GEMM weights: bf16, rows padded to 16 (zero rows)
        const int64_t prow = pad_to((int)rows, 16);
        std::vector<uint16_t> hb((size_t)prow * cols, 0);
        for (int64_t i = 0; i < rows * cols; ++i) hb[(size_t)i] = host_f2bf(data[i]);
        S_HIP_OK(h, hipMalloc(&t.p, hb.size() * 2));
        S_HIP_OK(h, hipMemcpy(t.p, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
    } else {
        const size_t bytes = (size_t)rows * cols * 4;
        S_HIP_OK(h, hipMalloc(&t.p, bytes));
        S_HIP_OK(h, hipMemcpy(t.p, data, bytes, hipMemcpyHostToDevice));
    }
    t.loaded = true;
    h->finalized = false;
    return 0;
}

int gitcap_student_finalize(gitcap_student_t* h) {
    if (!h) return sfail(h, GITCAP_ERR_ARG, "student_finalize: null handle");
    S_GUARD(h);
    for (auto& kv : h->w)
        if (!kv.second.loaded) return sfail(h, GITCAP_ERR_STATE, "student_finalize: tensor '" + kv.first + "' was never loaded");
    auto Fp = [&](const std::string& n) { return (const float*)h->w[n].p; };
    auto Wt = [&](const std::string& n) { return (const bf16_t*)h->w[n].p; };
    for (auto& g : h->graphs) (void)hipGraphExecDestroy(g.exec);     // weight pointers are baked into the nodes
    h->graphs.clear();
    h->embed = Fp("embed.weight"); h->pe = Fp("pos_enc.pe"); h->head_w = Wt("linear.weight"); h->head_b = Fp("linear.bias");
    h->layers.resize(h->L);
    for (int i = 0; i < h->L; ++i) {
        const std::string p = "decoder.layers." + std::to_string(i) + ".";
        StuLayer& Ly = h->layers[i];
        Ly.sa_in_w = Wt(p + "self_attn.in_proj_weight"); Ly.sa_in_b = Fp(p + "self_attn.in_proj_bias");
        Ly.sa_out_w = Wt(p + "self_attn.out_proj.weight"); Ly.sa_out_b = Fp(p + "self_attn.out_proj.bias");
        Ly.ca_in_w = Wt(p + "multihead_attn.in_proj_weight"); Ly.ca_in_b = Fp(p + "multihead_attn.in_proj_bias");
        Ly.ca_out_w = Wt(p + "multihead_attn.out_proj.weight"); Ly.ca_out_b = Fp(p + "multihead_attn.out_proj.bias");
        Ly.l1w = Wt(p + "linear1.weight"); Ly.l1b = Fp(p + "linear1.bias");
        Ly.l2w = Wt(p + "linear2.weight"); Ly.l2b = Fp(p + "linear2.bias");
        Ly.n1w = Fp(p + "norm1.weight"); Ly.n1b = Fp(p + "norm1.bias");
        Ly.n2w = Fp(p + "norm2.weight"); Ly.n2b = Fp(p + "norm2.bias");
        Ly.n3w = Fp(p + "norm3.weight"); Ly.n3b = Fp(p + "norm3.bias");
    }
    if (h->allocs.empty()) {
        const size_t Mt = (size_t)h->Mt, D = (size_t)h->D;
        const int ks = std::max(skinny_ksplit(h->D), skinny_ksplit(h->FF));
        const size_t ntiles = ((size_t)h->V + 15) / 16;
        int rc;
        if ((rc = s_alloc(h, &h->xf2, 2 * D)) || (rc = s_alloc(h, &h->xf, Mt * D)) || (rc = s_alloc(h, &h->xb, Mt * D)) || (rc = s_alloc(h, &h->qc, Mt * D)) ||
            (rc = s_alloc(h, &h->ctx, Mt * D)) || (rc = s_alloc(h, &h->ffn, Mt * h->FF)) ||
            (rc = s_alloc(h, &h->slabs, (size_t)ks * Mt * D)) || (rc = s_alloc(h, &h->amax_val, Mt * ntiles)) ||
            (rc = s_alloc(h, &h->amax_idx, Mt * ntiles)) || (rc = s_alloc(h, &h->kvs, (size_t)h->L * Mt * 3 * D)) ||
            (rc = s_alloc(h, &h->memb, (size_t)h->R * h->F * D)) ||
            (rc = s_alloc(h, &h->memkv, (size_t)h->L * h->R * h->F * 2 * D)) || (rc = s_alloc(h, &h->sep_cnt, (size_t)h->Tmax + 1)) ||
            (rc = s_alloc(h, &h->g_ids, (size_t)h->R * h->Tmax)) || (rc = s_alloc(h, &h->g_steps, (size_t)1)))
            return rc;
    }
    h->finalized = true;
    return 0;
}

int gitcap_student_set_memory(gitcap_student_t* h, const float* memory, int B, void* stream) {
    if (!h) return sfail(h, GITCAP_ERR_ARG, "student_set_memory: null handle");
    S_GUARD(h);
    return set_memory(h, memory, B, (hipStream_t)stream);
}

int gitcap_student_forward_decoder(gitcap_student_t* h, const int64_t* ids, int ld_ids, int B, int T, float* logits, void* stream) {
    if (!h) return sfail(h, GITCAP_ERR_ARG, "student_forward_decoder: null handle");
    if (!logits) return sfail(h, GITCAP_ERR_ARG, "student_forward_decoder: null logits");
    S_GUARD(h);
    return text_forward(h, ids, ld_ids, B, 0, T, logits, nullptr, 0, nullptr, 0, (hipStream_t)stream);
}

int gitcap_student_greedy(gitcap_student_t* h, const float* memory, int B, int max_len, int stop, int64_t* ids_out,
                          int32_t* steps_out, void* stream) {
    if (!h) return sfail(h, GITCAP_ERR_ARG, "student_greedy: null handle");
    if (!ids_out || max_len <= 0) return sfail(h, GITCAP_ERR_ARG, "student_greedy: bad arguments");
    if (max_len + 1 > h->Tmax) return sfail(h, GITCAP_ERR_ARG, "student_greedy: max_len exceeds max_text_len");
    if (stop != GITCAP_STOP_NEVER && stop != GITCAP_STOP_ALL_SEP) return sfail(h, GITCAP_ERR_ARG, "student_greedy: unknown stop rule");
    S_GUARD(h);
    hipStream_t s = (hipStream_t)stream;
    int rc = set_memory(h, memory, B, s);           // reads the caller's buffer: outside the graph
    if (rc) return rc;
    const int ld = max_len + 1;
    // The token loop is launch-latency bound (26 kernels per token): it is captured once per (B, max_len, stop)
    // and replayed.  GITCAP_STUDENT_GRAPH=0 launches it kernel by kernel (same kernels, same results).
    static const bool use_graph = !(getenv("GITCAP_STUDENT_GRAPH") && atoi(getenv("GITCAP_STUDENT_GRAPH")) == 0);
    auto enqueue_loop = [&](hipStream_t q) -> int {
        S_HIP_OK(h, launch_fill_i64(h->g_ids, ld, B, h->c.cls_token_id, q));                 // model.py:171
        S_HIP_OK(h, hipMemsetAsync(h->sep_cnt, 0, ((size_t)h->Tmax + 1) * 4, q));
        for (int t = 0; t < max_len; ++t) {                                                  // model.py:173-182
            int r = text_forward(h, h->g_ids, ld, B, t, 1, nullptr, h->g_ids + t + 1, ld, h->sep_cnt, t, q);
            if (r) return r;
        }
        S_HIP_OK(h, launch_finish_steps(h->sep_cnt, B, max_len, stop, h->g_steps, q));       // model.py:184
        return 0;
    };
    if (use_graph) {
        hipGraphExec_t exec = nullptr;
        for (auto& g : h->graphs)
            if (g.B == B && g.max_len == max_len && g.stop == stop && g.rows_pro == g_row_prologue && g.head_share == g_head_share) exec = g.exec;
        if (!exec) {
            hipGraph_t graph = nullptr;
            if (!h->cap_stream) S_HIP_OK(h, hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking));
            S_HIP_OK(h, hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
            rc = enqueue_loop(h->cap_stream);
            hipError_t e = hipStreamEndCapture(h->cap_stream, &graph);
            if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
            S_HIP_OK(h, e);
            e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            S_HIP_OK(h, e);
            h->graphs.push_back({B, max_len, stop, g_row_prologue, g_head_share, exec});
        }
        S_HIP_OK(h, hipGraphLaunch(exec, s));
    } else if ((rc = enqueue_loop(s))) {
        return rc;
    }
    S_HIP_OK(h, hipMemcpyAsync(ids_out, h->g_ids, (size_t)B * ld * sizeof(int64_t), hipMemcpyDeviceToDevice, s));
    if (steps_out) S_HIP_OK(h, hipMemcpyAsync(steps_out, h->g_steps, sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    return 0;
}

// StudentCandidateV1.beam_search (model.py:189-318) on the device with the exact KV cache: k beams per clip as rows
// b * k + i, no end-of-sequence handling (as the reference), no host round trip.  The reference lets every beam propose its
// top k and keeps the k best of the k * k candidates; the k best of ALL beams x vocabulary candidates are the same set (a
// candidate among the global k best is among its own beam's k best), which is what beam_topk ranks (log_softmax + beam
// score, best first, ties to the smaller beam-major index).  Before the first step only beam 0 is live (score 0, the others
// -1e9), so the k rows start as the top k of the single prefix [CLS] (model.py:221-227).  After every step the self-attention
// K/V rows follow their beams (gather into the second cache buffer) and so do the id rows the PAD-key mask reads.
int gitcap_student_beam_search(gitcap_student_t* h, const float* memory, int B, int k, int max_len, int64_t* ids_out, void* stream) {
    if (!h) return sfail(h, GITCAP_ERR_ARG, "student_beam_search: null handle");
    if (!memory || !ids_out || B <= 0 || k <= 0 || max_len < 2) return sfail(h, GITCAP_ERR_ARG, "student_beam_search: bad arguments");
    if (k > 16) return sfail(h, GITCAP_ERR_ARG, "student_beam_search: at most 16 beams");
    if ((int64_t)B * k > h->R) return sfail(h, GITCAP_ERR_ARG, "student_beam_search: B * k exceeds max_rows");
    if (max_len > h->Tmax) return sfail(h, GITCAP_ERR_ARG, "student_beam_search: max_len exceeds max_text_len + 1");
    S_GUARD(h);
    hipStream_t s = (hipStream_t)stream;
    const int rows = B * k, D = h->D, V = h->V, ld = h->Tmax + 1;
    int rc = 0;
    gitcap_student::BeamWs& w = h->bw;
    if (!w.memrep) {
        const size_t R = h->R;
        rc = rc ? rc : s_alloc(h, &w.memrep, R * h->F * D);
        rc = rc ? rc : s_alloc(h, &w.scores, R);
        rc = rc ? rc : s_alloc(h, &w.cand_scores, R);
        rc = rc ? rc : s_alloc(h, &w.cand_idx, R);
        rc = rc ? rc : s_alloc(h, &w.src_rows, R);
        rc = rc ? rc : s_alloc(h, &w.ids0, R * (size_t)ld);
        rc = rc ? rc : s_alloc(h, &w.ids1, R * (size_t)ld);
        rc = rc ? rc : s_alloc(h, &w.logits, R * (size_t)V);
        rc = rc ? rc : s_alloc(h, &w.kvs2, (size_t)h->L * h->R * h->Tmax * 3 * D);
        rc = rc ? rc : s_alloc(h, &w.topk_scratch, beam_topk_scratch_bytes(h->R, 1, V, 16));     // rows x chunks, whatever the split into clips x beams
        if (rc) { w = gitcap_student::BeamWs{}; return rc; }
    }
    hipLaunchKernelGGL(repeat_rows_kernel, dim3(4, rows), dim3(256), 0, s, memory, w.memrep, k, h->F * D / 4);
    S_HIP_OK(h, hipGetLastError());
    if ((rc = set_memory(h, w.memrep, rows, s))) return rc;
    S_HIP_OK(h, launch_fill_i64(w.ids0, ld, rows, h->c.cls_token_id, s));
    hipLaunchKernelGGL(beam_scores_init_kernel, dim3((rows + 63) / 64), dim3(64), 0, s, w.scores, rows, k);
    S_HIP_OK(h, hipGetLastError());
    int64_t *cur = w.ids0, *nxt = w.ids1;
    bf16_t* const kvs_home = h->kvs;
    const size_t kv_layer = (size_t)h->R * h->Tmax * 3 * D;
    for (int t = 0; t + 1 < max_len && !rc; ++t) {                      // the token at position t is decoded, position t + 1 chosen
        if (t > 0) {                                                    // rows continue beam src_rows[r]: positions 0 .. t-1, all layers
            bf16_t* other = h->kvs == kvs_home ? w.kvs2 : kvs_home;
            const hipError_t eg = launch_gather_txt_rows(h->kvs, other, w.src_rows, rows, t, h->Tmax, 3 * D, h->L, kv_layer, s);
            if (eg != hipSuccess) { rc = sfail(h, GITCAP_ERR_HIP, std::string("student_beam_search: gather_txt_rows: ") + hipGetErrorString(eg)); break; }   // (h->kvs is restored below on every path)
            h->kvs = other;
        }
        rc = text_forward(h, cur, ld, rows, t, 1, w.logits, nullptr, 0, nullptr, 0, s);
        if (rc) break;
        hipError_t e = launch_beam_topk(w.logits, V, w.scores, B, k, V, k, w.cand_scores, w.cand_idx, w.topk_scratch, s);
        if (e != hipSuccess) { rc = sfail(h, GITCAP_ERR_HIP, std::string("student_beam_search: beam_topk: ") + hipGetErrorString(e)); break; }
        hipLaunchKernelGGL(student_beam_step_kernel, dim3(B), dim3(64), 0, s, w.cand_scores, w.cand_idx, cur, nxt, w.scores, w.src_rows, k, V, t, ld);
        if (hipGetLastError() != hipSuccess) { rc = sfail(h, GITCAP_ERR_HIP, "student_beam_search: beam step launch"); break; }
        std::swap(cur, nxt);
    }
    h->kvs = kvs_home;                                                  // (the captured greedy graphs hold this pointer)
    if (rc) return rc;
    hipLaunchKernelGGL(student_beam_finish_kernel, dim3(B), dim3(64), 0, s, cur, ids_out, k, ld, max_len);
    S_HIP_OK(h, hipGetLastError());
    return 0;
}

}  // extern "C"
