// Launcher declarations for the gfx950 kernels behind libgitcap's C ABI.
#pragma once
#include "common.h"
#include <atomic>
#include <cstddef>

// ---- big-tile bf16 MFMA GEMM:  C[m][n] = sum_k A[m][k] * W[n][k]  (+ epilogue) -------------
enum GemmEpi {
    EPI_BIAS_BF16 = 0,        // out bf16 = acc + bias
    EPI_BIAS_QGELU_BF16 = 1,  // out bf16 = quick_gelu(acc + bias)      (CLIP MLP)
    EPI_BIAS_GELU_BF16 = 2,   // out bf16 = erf_gelu(acc + bias)        (BERT intermediate)
    EPI_BIAS_RESID_F32 = 3,   // out f32  = acc + bias + resid          (out may alias resid)
    EPI_BIAS_F32 = 4,         // out f32  = acc + bias
    EPI_PATCH_F32 = 5,        // out f32 row (frame*N + 1 + patch) = acc + pos[1+patch]
    // 256x256 kernel only, N = 768 or 1024: the tiles of one row block exchange LayerNorm statistics (ln_canon.h)
    // and each normalises its own columns -> the LayerNorm launch behind the GEMM and its re-read disappear.
    EPI_RESID_LN_PRE = 6,     // x = acc + bias + resid: out f32 = x (nullable), ln_out bf16 = LayerNorm(x) [+ ln_add] (pre-LN ViT block; ln_post)
    EPI_RESID_LN_POST = 7,    // x = acc + bias [+ resid]: out f32 = LayerNorm(x), ln_out bf16 = the same (post-LN decoder)
    // fp8 kernel only (gemm_f8.hip): the GELU output as OCP e4m3 codes of y * out8_inv -- the activation operand of the next fp8 GEMM
    EPI_BIAS_QGELU_F8 = 8,    // out e4m3 = quick_gelu(acc + bias)
    EPI_BIAS_GELU_F8 = 9      // out e4m3 = erf_gelu(acc + bias)
};
struct GemmArgs {
    const bf16_t* A; int lda;     // activations [M][lda], M multiple of 128 (padded rows are junk)
    const bf16_t* W;              // weights [N][K] (torch Linear layout), N multiple of 128
    const float* bias;            // [N] or nullptr
    int M, N, K;                  // K multiple of 64
    void* out; int ldo;
    const float* resid; int ldr;
    const float* pos;             // EPI_PATCH: [tokens_per_frame][N]
    int tokens_per_frame, patches_per_frame, valid_rows;
    // EPI_RESID_LN_*: LayerNorm of the output rows
    const float *ln_g, *ln_b; float ln_eps;
    bf16_t* ln_out; int ld_ln;    // bf16 LayerNorm output
    const float* ln_add;          // nullable (PRE): LayerNorm output += ln_add[((row / ln_add_div) % ln_add_mod) * N + n]  (temporal embedding)
    int ln_add_div, ln_add_mod;
    float* ln_out_f32; int ld_ln_f32;   // nullable (PRE): fp32 copy of the LayerNorm output, rows < valid_rows only (caller's unpadded buffer)
    float2* ln_stats;             // [ln_stats_rows][16] per-segment (mean, M2) exchanged between the tiles of a row block
    unsigned* ln_cnt;             // [row blocks][2] per row block {arrivals, generation}: zero before the first launch, self-resetting
    int ln_stats_rows;            // rows of ln_stats (>= M)
    int ln_rowblock_map;          // set by the launcher: workgroup -> tile map hands every XCD whole row blocks (host_logic.h)
    // fp8 compute (gemm_f8.hip): A and W point at e4m3 codes; the accumulators are multiplied by ascale * wscale[n]
    const float* wscale; float ascale;
    float out8_inv;               // EPI_BIAS_*GELU_F8: 1 / (static scale of the e4m3 output)
    unsigned char* ln_out8; int ld_ln8; float ln_out8_inv;   // nullable (EPI_RESID_LN_*): e4m3 copy of the LayerNorm output * ln_out8_inv
    unsigned* ln_fail;            // nullable: host-visible word raised when a tile gave up waiting for its siblings (no trap)
    unsigned ln_spin_limit;       // polls (~0.3 us each) before giving up; 0 = the default (~30 s)
    unsigned long long* f8_sat;   // nullable: device counter of e4m3 activation codes (valid rows) the epilogue clamped at +-448
                                  // (LAST: the residual + LayerNorm epilogue runs at the 256-VGPR limit and where the compiler spills
                                  // depends on the kernarg layout -- tests/test_isa_lint.py: test_gemm_ln_epilogue_keeps_its_spills_out_of_the_row_loops)
};
static_assert(offsetof(GemmArgs, f8_sat) + sizeof(unsigned long long*) == sizeof(GemmArgs), "f8_sat must stay the LAST field of GemmArgs (see its comment)");
constexpr unsigned LN_SPIN_DEFAULT = 1u << 26;
int device_cus();                 // compute units of the current device (cached per device; 256 when the query fails)
hipError_t launch_gemm(const GemmArgs& a, int epi, hipStream_t s);      // 128x128 tile (any M%128, N%128)
hipError_t launch_gemm64(const GemmArgs& a, int epi, hipStream_t s);    // 64x64 tile, 3-stage ring (few-hundred-row launches)
bool gemm256_ok(const GemmArgs& a);
hipError_t launch_gemm256(const GemmArgs& a, int epi, hipStream_t s);   // 256x256 tile, 8-wave ping-pong
bool gemm256_ln_ok(const GemmArgs& a);                                   // shape the EPI_RESID_LN_* epilogues accept
bool gemm256f8_ok(const GemmArgs& a);                                    // fp8 operands: K % 128 == 0, wscale / ascale set
hipError_t launch_gemm256f8(const GemmArgs& a, int epi, hipStream_t s);  // the same tile kernel on e4m3 operands (gemm_f8.hip)

// ---- skinny GEMMs (text rows; M = a few 16-row tiles): weight streaming, one wave per tile ----
enum SkinnyEpi { SK_BIAS_BF16 = 0, SK_BIAS_GELU_BF16 = 1, SK_BIAS_RELU_BF16 = 2, SK_BIAS_F32 = 3 };
struct SkinnyArgs {
    const bf16_t* X; int ldx;     // bf16 activations, row m at X + m*ldx
    const void* W;                // [Npad16][K] bf16, or e4m3 bytes when wscale != nullptr
    const float* wscale;          // nullable: [Npad16] per-row power-of-two scale of e4m3 weights
    const float* bias;            // [N] (full kernel only)
    int M, N, K;                  // M valid rows, N valid cols (weight rows padded to 16)
    void* out;                    // full: row m -> out + orow(m)*ldo, orow(m) = (m / T)*row_stride + row_off + m % T
    int ldo, T, row_stride, row_off;   // splitk: out = fp32 slabs [ksplit][M][ldo]
    float* amax_val; int* amax_idx;    // SK_BIAS_F32 only (nullable): per-tile arg-max partials [M][ntiles]
    // Optional row prologue (kind != 0; M <= 2, K = the row width, bf16 weights): every workgroup computes the M input rows
    // itself -- the LayerNorm that would otherwise be the launch in front of this one -- while its weight fragments are
    // in flight; X is not read.  Workgroup 0 also writes the fp32 rows to xf (must not alias resid).
    struct RowPrologue {
        int kind;                          // 0 none, 1 split-K slabs + bias + residual -> LayerNorm, 2 text embedding -> LayerNorm
        const float* slabs; int nslab;     // kind 1
        const float *bias, *resid;
        const int64_t* ids;                // kind 2
        int ld_ids, T, t0, vocab;
        const float *word, *pos;
        const float *g, *b; float eps;     // LayerNorm
        float* xf;                         // [M][K]
    } ln;
    // optional fragment-major copy of W (launch_pack_frags: [tile][k32][lane][8]); when set the kernels read it instead of W
    const void* Wpk;
    int ksplit;                            // splitk: number of K slabs (0: skinny_ksplit(K))
};
extern std::atomic<bool> g_row_prologue;                                 // gitcap.hip: GITCAP_NO_ROW_PROLOGUE / gitcap_dbg_config(1, .)
// vocabulary head: four 16-column tiles per workgroup share the activation rows through LDS (skinny.hip: skinny_head_kernel);
// GITCAP_NO_HEAD_SHARE / gitcap_dbg_config(10, 0): one single-wave workgroup per tile.  Same bits either way.
extern std::atomic<bool> g_head_share;
// one/two-row prologue over more than 16 slabs: three-wave workgroups that share the reduce (skinny.hip: skinny_rows3_kernel);
// GITCAP_NO_ROWS3 / gitcap_dbg_config(11, 0): the single-wave form.  Same bits either way.
extern std::atomic<bool> g_rows3;
bool skinny_row_prologue_ok(int M, int K, bool fp8);         // shapes the row-prologue form is instantiated for
hipError_t launch_skinny(const SkinnyArgs& a, int epi, hipStream_t s);
bool skinny_full_ok(int K);                                  // K depths launch_skinny is instantiated for
int skinny_ksplit(int K);                                    // number of K slabs launch_skinny_splitk writes
hipError_t launch_skinny_splitk(const SkinnyArgs& a, hipStream_t s);
// x = LayerNorm(sum_s slab[s][m][:] + bias + resid[m][:]) -> xf (fp32) and xb (bf16); one wave per row
hipError_t launch_ln_reduce(const float* slabs, int nslab, const float* bias, const float* resid,
                            const float* gamma, const float* beta, float eps, int M, int D,
                            float* xf, bf16_t* xb, hipStream_t s);
// fragment-major copy of a GEMM weight [rows16][K] (bf16: elem_bytes 2, e4m3 codes: 1) for the weight-streaming text kernels
hipError_t launch_pack_frags(const void* src, void* dst, int rows16, int K, int elem_bytes, hipStream_t s);

// ---- fused FC1 -> GELU -> FC2 of the text rows over hidden slices (ffn_txt.hip) -----------------------------------------
// Workgroup s owns hidden units 64 s .. 64 s + 63: h = GELU(X W1_s^T + b1_s) rounded to bf16 (exactly the FC1 launch's
// output), then its split-K share of FC2, slab[s][m][:] = h W2[:, slice]^T (fp32).  The slabs are summed by
// launch_ln_reduce / the row prologue (nslab = F / 64).  Bitwise equal to launch_skinny (FC1 + GELU) followed by
// launch_skinny_splitk with ksplit = F / 64.
struct FfnTxtArgs {
    const bf16_t* X; int ldx;                   // [M][D] bf16
    const void *W1pk, *W2pk;                    // fragment-major FC1 [F][D] / FC2 [D][F]: bf16, or e4m3 codes when w1scale != nullptr
    const float *w1scale, *w2scale;             // nullable: per-row power-of-two scales of e4m3 weights ([F] / [D])
    const float* b1;                            // [F]
    int M, D, F;
    float* slabs;                               // [F / 64][M][D]
};
bool ffn_txt_ok(int D, int F);
hipError_t launch_ffn_txt(const FfnTxtArgs& a, hipStream_t s);
// out[r*ld_out] = index of the max over the per-tile partials of row r*row_stride + row_off.
// emb (nullable; greedy loop, one position per row): the kernel goes on to embed the token it just chose at text position
// emb->position -- word + position embedding -> LayerNorm -> xf / xb row r (rowln.h: the code of launch_embed_text) -- so
// the next token step starts at its q|k|v projection.
struct NextEmbed { const float *word, *pos, *gamma, *beta; float eps; int D, vocab, position; float* xf; bf16_t* xb; };
hipError_t launch_argmax_final(const float* amax_val, const int* amax_idx, int ntiles, int rows, int row_stride,
                               int row_off, int64_t* out, int ld_out, int32_t* sep_cnt, int step, int sep_id, hipStream_t s,
                               const NextEmbed* emb = nullptr);

// ---- attention ---------------------------------------------------------------------------
// Full (unmasked) self-attention over groups of S rows: qkv [G*S][3*W] bf16 (q | k | v, head h
// at columns h*64), ctx [G*S][W] bf16.  Used for the ViT frames (S = N) and for the image
// prefix of the decoder (S = F*N).
hipError_t launch_attn_full(const bf16_t* qkv, bf16_t* ctx, int G, int S, int H, hipStream_t s);

// Attention sub-layer for text rows (txtblock.hip): query (r, t0+j) attends the image keys of clip r/beams and the text
// keys 0..t0+j of row r -> this head's share of the output dense -> the last head of a row to arrive adds bias +
// residual and applies LayerNorm.  One workgroup per (text row, head).
struct TxtBlockArgs {
    const bf16_t* kv_img;                       // [B*S_img][3D] this layer
    const bf16_t* kv_txt;                       // [R][Tmax][3D] this layer (q | k | v of the text rows)
    int rows, beams, t0, T, Tmax, S_img, H, D;
    const void* aow;                            // output dense [D][D]: bf16, or e4m3 bytes when aoscale != nullptr
    const void* aowpk;                          // nullable (bf16 only): its fragment-major copy (launch_pack_frags)
    const float* aoscale;                       // nullable: [D] per-row power-of-two scale of e4m3 weights
    const float *aob, *g1, *b1;                 // its bias; LayerNorm of the sub-layer
    const float* xin;                           // [M][D] the sub-layer's input (residual)
    float eps;
    float* part;                                // [M][H][D] fp32 per-head partials of the output dense
    unsigned* cnt;                              // [M] arrival tickets (zero between launches)
    float* xs; bf16_t* xsb;                     // out: x1 [M][D] fp32 and bf16 (xs may alias xin)
    // opt-in kv_cache = v_e4m3: V of the image keys as e4m3 codes [H][v8_pitch][64] + one power-of-two scale per (key, head)
    // [H][v8_pitch], head-major (launch_kv_quant_v); K and the text rows' own K/V stay bf16.  nullptr: bf16 V from kv_img.
    const unsigned char* v8_img; const float* vs_img; int64_t v8_pitch;
    int Mh;                                     // set by the launcher
    int nt_kv;                                  // 1: the K/V rows are streamed with non-temporal loads (they do not fit the caches anyway)
};
bool txt_block_ok(int D);
extern std::atomic<bool> g_txt8;                                          // txtblock.hip: 8-wave workgroups where units > CUs (speed switch 9)
hipError_t launch_txt_block(const TxtBlockArgs& a, hipStream_t s);

// Small attention of the student decoder (student.hip): one wave per (query row, head), at most 64 keys,
// any head_dim that is a multiple of 8 (<= 128).  Query m = (r, j), r = m / T: q at
// q + (r*q_row_stride + q_row_off + j)*ldq + h*hd; key/value i of row r at k|v + (r*keys_stride + i)*ldkv + h*hd.
// nkeys > 0: every query sees keys 0..nkeys-1 (cross-attention); nkeys == 0: causal, keys 0..t0+j.
// ids != nullptr: key i of row r is masked when ids[r*ld_ids + i] == pad_id (all keys masked -> NaN, as torch).
struct SmallAttnArgs {
    const bf16_t* q; int ldq, T, q_row_stride, q_row_off;
    const bf16_t* k; const bf16_t* v; int ldkv, keys_stride, nkeys, t0;
    const int64_t* ids; int ld_ids, pad_id;
    bf16_t* ctx; int ldc;
    int M, H, hd;
};
hipError_t launch_attn_small(const SmallAttnArgs& a, hipStream_t s);
// x[m] = (embed[ids[r*ld_ids + t0 + j]] + pe[t0 + j]) / sqrt(D), m = r*T + j  -> xf (fp32) and xb (bf16)
hipError_t launch_student_embed(const int64_t* ids, int ld_ids, int rows, int T, int t0, const float* embed,
                                const float* pe, int D, int vocab, float* xf, bf16_t* xb, hipStream_t s);

// ---- row ops -----------------------------------------------------------------------------
struct LnArgs {
    const float* x; int ldx;      // [rows][ldx]
    const float* gamma; const float* beta; float eps;
    int rows, D;
    float* out_f32; int ld_f32;   // nullable
    bf16_t* out_bf16; int ld_bf16;// nullable
    const float* add_vec;         // nullable: out += add_vec[((row / add_div) % add_mod) * D + c]
    int add_div, add_mod;
    // optional SECOND LayerNorm of the first one's fp32 output (canonical widths only): out_f32 = LN(x), out_bf16 =
    // LN2(LN(x)) -- ln_pre and the first block's LN1 of the ViT in one pass over the rows.  Same bits as two launches.
    const float* gamma2; const float* beta2; float eps2;
    // optional: rows with row % cls_period == 0 are not read from x but are cls[c] + cls_pos[c] (the CLS token + its position
    // embedding of the ViT front end: x[frame * N] of model.py:378's encoder input) -- the row a cls_rows launch would write
    const float* cls; const float* cls_pos; int cls_period;
};
hipError_t launch_layernorm(const LnArgs& a, hipStream_t s);

// frames [nf][3][H][W] f32 -> patches bf16 [nf*G*G][Kp] (k = c*p*p + py*p + px, zero padded)
hipError_t launch_im2col(const float* frames, bf16_t* patches, int nf, int img, int p, int Kp, hipStream_t s);
// out[b][e][s][:] = s < S_img ? img[e][b*S_img + s][:] : txt[e][b*T + s - S_img][:]   (e < n_entries; fp32 rows of D)
hipError_t launch_gather_hidden(const float* img, const float* txt, float* out, int n_entries, int B, int S_img, int T, int D,
                                size_t img_entry_stride, size_t txt_entry_stride, hipStream_t s);
// V slice of kv [rows][3D] (bf16) -> e4m3 codes v8 [H][pitch][64] + power-of-two scales vs [H][pitch] per (row, head), head-major
hipError_t launch_kv_quant_v(const bf16_t* kv, unsigned char* v8, float* vs, int rows, int D, int H, int64_t pitch, hipStream_t s);
// f32 -> bf16 copy
hipError_t launch_cast_bf16(const float* in, bf16_t* out, int64_t n, hipStream_t s);
// e4m3 weight rows [rows][K] (+ per-row power-of-two scale) -> bf16 [rows][K] (exact); K % 16 == 0
hipError_t launch_dequant_fp8(const unsigned char* w8, const float* scale, bf16_t* out, int rows, int K, hipStream_t s);
// up to four such matrices in ONE launch (the weights of a transformer layer); n16[i] = rows_i * K_i / 16
struct DequantBatch { const unsigned char* w8[4]; const float* scale[4]; bf16_t* out[4]; int K[4]; int64_t n16[4]; int n; };
hipError_t launch_dequant_fp8_batch(const DequantBatch& b, hipStream_t s);
// text embedding + LN: row m=(r, j) -> token ids[r*ld_ids + j], position t0 + j
hipError_t launch_embed_text(const int64_t* ids, int ld_ids, int rows, int T, int t0,
                             const float* word, const float* pos, const float* gamma, const float* beta,
                             float eps, int D, int vocab, float* x_f32, bf16_t* x_bf16, hipStream_t s);
// argmax over [rows][V] (row stride ld) -> out[r*ld_out] (int64); lowest index wins ties.
// If sep_cnt != nullptr: sep_cnt[step] += 1 for every row whose argmax is sep_id (zero it per call).
hipError_t launch_argmax(const float* logits, int ld, int rows, int V, int64_t* out, int ld_out,
                         int32_t* sep_flags, int step, int sep_id, hipStream_t s);
hipError_t launch_finish_steps(const int32_t* sep_cnt, int rows, int max_len, int stop, int32_t* steps_out, hipStream_t s);
hipError_t launch_fill_i64(int64_t* p, int ld, int rows, int64_t v, hipStream_t s);
hipError_t launch_gather_txt_rows(const bf16_t* src, bf16_t* dst, const int32_t* src_rows, int rows,
                                  int t_len, int Tmax, int width, int layers, size_t layer_stride, hipStream_t s);
// top-K over (beam, vocab) of log_softmax(logits) + beam_scores, one block per batch element
size_t beam_topk_scratch_bytes(int B, int beams, int V, int K);   // device scratch launch_beam_topk needs (V <= 131072)
hipError_t launch_beam_topk(const float* logits, int ld, const float* beam_scores, int B, int beams, int V, int K,
                            float* out_scores, int* out_idx, void* scratch, hipStream_t s);
// uint8 HWC BGR frames [nf][H][W][3] -> CLIP-normalised fp32 NCHW [nf][3][crop][crop] (bicubic resize + centre crop)
hipError_t launch_preprocess(const unsigned char* in, float* out, int nf, int H, int W, int crop, hipStream_t s);
// the same transform fused with the patch gather: -> bf16 patch rows [nf*G*G][Kp] (layout of launch_im2col)
hipError_t launch_preprocess_patches(const unsigned char* in, bf16_t* patches, int nf, int H, int W, int crop, int p, int Kp, hipStream_t s);
// device-resident beam search state + one bookkeeping step per decoder step (rowops.hip)
struct BeamBuffers {
    int64_t *ids0, *ids1, *words, *hyp_ids;
    float *beam_scores, *hyp_score;
    int32_t *src_rows, *done, *hyp_len;
};
hipError_t launch_beam_init(const BeamBuffers& bb, int B, int beams, int max_len, int cls, hipStream_t s);
hipError_t launch_beam_step(const BeamBuffers& bb, const float* cand_scores, const int* cand_idx, int B, int beams, int K,
                            int V, int cur_len, int max_len, int eos, float length_penalty, int cur, hipStream_t s);
hipError_t launch_beam_finish(const BeamBuffers& bb, int B, int max_len, int eos, int64_t* decoded, float* logprobs, hipStream_t s);
