/*
 * gitcap.h - C ABI of libgitcap.so: MI355X (gfx950) native GIT-style video-caption inference.
 *
 * The reference (farazali7/real-time-video-captioning) has no FFI/plugin layer: its boundary is
 * duck-typed Python methods on the model object.  Each entry point below names the reference
 * call it stands behind (paths relative to the reference root), so a maintainer can bind it with
 * ctypes (gitcap/model.py is that binding).
 *
 * Conventions
 *   - every function returns 0 on success or a negative gitcap_status; the message is read with
 *     gitcap_last_error(h) (h may be NULL for errors raised by gitcap_create);
 *   - no C++ exception crosses this boundary;
 *   - the caller owns every input/output buffer (device memory unless stated otherwise);
 *     the handle owns weights, KV cache and workspace;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream) and is
 *     asynchronous; nothing on the data path synchronises the device -- only the set-up and diagnosis calls do
 *     (gitcap_load_tensor / _finalize_weights, gitcap_set_compute / _set_fp8_scale, gitcap_fp8_saturations, and gitcap_poll_errors
 *     when it has a failure to report);
 *   - a handle is bound to one device and is not thread-safe.
 */
#ifndef GITCAP_H
#define GITCAP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gitcap gitcap_t;

typedef enum {
    GITCAP_OK = 0,
    GITCAP_ERR_ARG = -1,      /* bad argument / shape outside what the handle was created for */
    GITCAP_ERR_STATE = -2,    /* call order violated (e.g. decode before prefill, weights missing) */
    GITCAP_ERR_HIP = -3,      /* a HIP runtime call failed */
    GITCAP_ERR_NOMEM = -4,
    GITCAP_ERR_EXCHANGE = -5  /* a fused GEMM + LayerNorm launch gave up waiting for its sibling tiles (see gitcap_poll_errors) */
} gitcap_status;

typedef enum { GITCAP_F32 = 0, GITCAP_BF16 = 1 } gitcap_dtype;

/* greedy stop rules */
typedef enum {
    GITCAP_STOP_NEVER = 0,    /* always run max_len steps (fixed work; bench) */
    GITCAP_STOP_ALL_SEP = 1   /* reference rule: stop when ALL rows emit SEP in the same step
                                 (src/models/model.py:184); rows keep generating after their own SEP */
} gitcap_stop;

/* Hyper-parameters: src/models/model.py:681-718 (get_git_model) and
 * data/teacher_configs/GIT_LARGE_MSRVTT/parameter.yaml:1-3.  Field order is mirrored by
 * gitcap.config.CGitCapConfig. */
typedef struct gitcap_config {
    int32_t image_size, patch_size;
    int32_t enc_width, enc_layers, enc_heads, enc_ffn;
    int32_t dec_width, dec_layers, dec_heads, dec_ffn;
    int32_t vocab_size, max_text_pos;
    int32_t num_frames;            /* num_image_with_embedding; 0 = no temporal embedding */
    int32_t cls_token_id, sep_token_id, pad_token_id;
    float enc_ln_eps, dec_ln_eps, proj_ln_eps;
    int32_t max_batch;             /* clips per call the workspace is sized for */
    int32_t max_frames;            /* frames per clip */
    int32_t max_text_len;          /* text positions per row (CLS + generated) */
    int32_t max_beams;             /* >= 1 */
} gitcap_config;

/* Replaces: get_git_model(tokenizer, param)            src/models/model.py:681-718
 *           GenerativeImageTextTeacher.__init__          src/models/model.py:726-745 */
int gitcap_create(const gitcap_config* cfg, int device, gitcap_t** out);
void gitcap_destroy(gitcap_t* h);
const char* gitcap_last_error(const gitcap_t* h);

/* Replaces: load_state_dict(self.model, ckpt)            src/models/model.py:736-738
 * `name` is a canonical tensor name (gitcap/weights.py); `data` is HOST memory, fp32, row-major.
 * The library converts GEMM weights to bf16 (round-to-nearest-even) and keeps tables, biases and
 * LayerNorm parameters in fp32.  gitcap_finalize_weights fails if any tensor is missing. */
int gitcap_load_tensor(gitcap_t* h, const char* name, const float* data,
                       const int64_t* shape, int rank);
int gitcap_finalize_weights(gitcap_t* h);

/* Storage of the GEMM weights in HBM (BASELINE.json configs[4]: "GIT-large fp8 weights"; the reference's teacher is
 * data/teacher_configs/GIT_LARGE_MSRVTT/parameter.yaml:1-3).  Call before the first gitcap_load_tensor.
 * GITCAP_W_FP8_E4M3: every GEMM weight is kept as OCP e4m3 bytes + one power-of-two fp32 scale per output row, half
 * the bytes of bf16.  The values passed to gitcap_load_tensor must already be e4m3 x 2^k (weight-only quantisation is
 * the caller's choice: gitcap.weights.quantize_weights_fp8); anything else is refused, nothing is rounded silently.
 * The weight-streaming text kernels read the e4m3 bytes and expand them in registers (times the row scale: exactly the
 * bf16 value); the big-tile GEMMs of the image pass read each panel through a bf16 staging buffer.  Arithmetic is unchanged (bf16 MFMA, fp32 accumulate): results
 * are bitwise those of bf16 storage of the same values.  Under bf16 compute this is a CAPACITY option (half the weight bytes in
 * HBM), not a speed option: the staging launches cost 0.25 ms per BASELINE configs[4] batch (14.1 vs 13.8 ms; pipelined 353 vs 358
 * captions/s) and the token loop's saving on the weight stream does not make that up at 4 clips x 4 beams; it pays together with
 * GITCAP_COMPUTE_FP8_FFN below (12.8 ms; pipelined 398 captions/s), whose GEMMs read the codes as stored. */
typedef enum { GITCAP_W_BF16 = 0, GITCAP_W_FP8_E4M3 = 1 } gitcap_weight_storage;
int gitcap_set_weight_storage(gitcap_t* h, int storage);
/* device bytes of all loaded tensors (weights, scales, tables, biases) */
/* Arithmetic of the image pass (north_star: "MFMA bf16/fp8 GEMMs"; BASELINE configs[4]).  GITCAP_COMPUTE_BF16 (default): every GEMM
 * on bf16 operands.  GITCAP_COMPUTE_FP8_FFN (opt-in; needs e4m3 weight storage and 768- / 1024-wide models): FC1 and FC2 of the
 * image rows run on v_mfma_f32_16x16x128_f8f6f4 with the e4m3 weight codes read as stored and activations quantised to e4m3 with a
 * static scale (default: codes of value * 16, saturating at +-28; gitcap_set_fp8_scale / gitcap_fp8_saturations below) by the
 * producing epilogues; everything else stays bf16.  Results differ from
 * bf16 compute by the activation rounding (measured |dlogit| <= 0.3 of a spread of 4 on GIT-large: docs/LAB_NOTEBOOK.md par. 3 / 6); the oracle's
 * counterpart is GitOracle(emulate_fp8_act="ffn"). */
enum { GITCAP_COMPUTE_BF16 = 0, GITCAP_COMPUTE_FP8_FFN = 1 };
int gitcap_set_compute(gitcap_t* h, int compute);
/* The static scale of the e4m3 activation codes of GITCAP_COMPUTE_FP8_FFN (no reference counterpart: the reference computes in
 * fp32, src/models/model.py:378, :412-418).  A code holds value / scale; codes reach +-448, so the default 1/16 covers +-28 and
 * e.g. 1/4 covers +-112 at four times the rounding step.  `scale` must be a power of two in [2^-16, 2^8].  The mode saturates,
 * but never silently: every code of a valid row that the producing epilogues clamp at +-448 is counted on the device, and
 * gitcap_fp8_saturations reads (and, with reset != 0, clears) the count since the last reset.  Both calls synchronise the
 * device.  A caller calibrates by running representative clips and raising the scale until the count stays 0; the oracle's
 * counterpart is GitOracle(emulate_fp8_act="ffn", fp8_scale=...). */
int gitcap_set_fp8_scale(gitcap_t* h, float scale);
int gitcap_fp8_saturations(gitcap_t* h, int64_t* count, int reset);
int gitcap_weight_bytes(const gitcap_t* h, int64_t* bytes);

/* Format of the image-prefix V rows the TOKEN LOOP reads (north_star: "a KV cache for the decode loop ... bf16/fp8"; the reference
 * has no cache at all, src/models/model.py:412-418 recomputes the prefix for every token).  GITCAP_KV_BF16 (default): the q|k|v
 * GEMM output of the decoder's image rows as it is.  GITCAP_KV_V_E4M3 (opt-in): a second copy of the V rows as OCP e4m3 codes with
 * one power-of-two scale per (token, head), written once per clip behind each decoder layer's q|k|v GEMM; the text rows' attention
 * reads K in bf16 and V from the codes (3/4 of the bytes of its K/V stream).  K stays bf16 in every mode: a peaked head's scores do
 * not survive a 6 % step on a key (profiles/r05_fp8_kv_cache_study.txt: up to 2.5 on logits of std 4).  The image rows' own
 * attention, the text rows' own K/V and all arithmetic are unchanged; results differ from the default by the rounding of V
 * (measured |dlogit| 0.10 plain / 0.5 stress weights); the oracle's counterpart is GitOracle(emulate_fp8_v=True).  The exact
 * KV-cache property (cached step == teacher-forced pass, bitwise) and batch invariance hold in either mode.  Synchronises the device;
 * +3/8 of the image K/V bytes of workspace (4 slots). */
enum { GITCAP_KV_BF16 = 0, GITCAP_KV_V_E4M3 = 1 };
int gitcap_set_kv_cache(gitcap_t* h, int mode);

/* Replaces: self.image_encoder(torch.stack(batch['image'])) + temporal add + cat(dim=1)
 *                                                         src/models/model.py:378-382
 *           the 'linearLn' visual projection              src/models/model.py:699
 *           and the image half of self.textual(...)       src/models/model.py:412-418
 * frames: device [B,F,3,H,W] fp32 NCHW (layout of src/utils/dataloader.py:60-82).
 * visual_out (nullable): device fp32 [B, F*N, enc_width] = ln_post + temporal embedding
 * (the `visual_features` the reference returns at model.py:424 / :460).
 * Because image tokens never attend to text (GIT block mask) their decoder K/V are text
 * independent: this call also runs the projected image tokens through the decoder layers and
 * leaves their K/V in the handle (the exact KV cache every later text call reads). */
int gitcap_encode(gitcap_t* h, const float* frames, int B, int F, float* visual_out, void* stream);

/* Same as the second half of gitcap_encode, starting from caller-supplied visual features
 * (device fp32 [B, S_img, enc_width]); lets forward_decoder(y, memory) honour `memory`. */
int gitcap_set_visual(gitcap_t* h, const float* visual, int B, int S_img, void* stream);

/* Replaces: self.textual(visual_features, caption_tokens)  src/models/model.py:412-418
 *           scores = step(input_ids)                        src/models/model.py:519
 * Runs text positions t0 .. t0+T-1 of `rows` rows through the decoder against the cached image
 * K/V of clip (row / beams) and the row's own cached text K/V (text->image full, text->text
 * causal), appending their K/V to the text cache.  Positions < t0 must have been run before.
 * ids: device int64, token of row r / position t0+j at ids[r*ld_ids + j].
 * logits_out (nullable): device fp32 [rows, T, vocab] when all_positions != 0 (teacher-forced
 * logits of forward_output_logits, model.py:747-760), else [rows, vocab] for position t0+T-1.
 * argmax_out (nullable): device int64, argmax of the last position's logits written to
 * argmax_out[r*ld_argmax]. */
int gitcap_text_forward(gitcap_t* h, const int64_t* ids, int ld_ids, int rows, int beams,
                        int t0, int T, float* logits_out, int all_positions,
                        int64_t* argmax_out, int ld_argmax, void* stream);

/* Replaces: the third return of GenerativeImageTextModel.forward_one_custom (hidden_states, stacked per layer)
 *                                                         src/models/model.py:419-424, :747-760
 * Opt-in (it costs a copy of every row after every layer, and the image rows of the LAST decoder layer, which are
 * otherwise never computed: only their K/V are needed).  While enabled, gitcap_encode / gitcap_set_visual and a
 * gitcap_text_forward with t0 = 0 keep the decoder stack's input and the output of each of its layers;
 * gitcap_hidden_states_read writes them as out[b][e][s][:], device fp32 [B][dec_layers + 1][S_img + T][dec_width]
 * (e = 0: projected image tokens ; text embeddings, e = l: output of layer l; s over [image ; text]).
 * Synchronous entry points only (not the pipelined submissions). */
int gitcap_hidden_states_enable(gitcap_t* h, int enable);
int gitcap_hidden_states_read(gitcap_t* h, int B, int S_img, int T, float* out, void* stream);

/* Replaces: StudentCandidateV1.greedy_decode(src, max_len) src/models/model.py:156-187
 *           as called by src/real_time_inference.py:58 and src/inference.py:51
 * Encodes, prefills with CLS and runs max_len greedy steps entirely on the device.
 * ids_out: device int64 [B, max_len+1] (column 0 = CLS).  steps_out: device int32[1] = number
 * of generated columns that are valid under `stop` (the host truncates to 1+steps). */
int gitcap_greedy(gitcap_t* h, const float* frames, int B, int F, int max_len, int stop,
                  int64_t* ids_out, int32_t* steps_out, void* stream);

/* Replaces: image_transform()                          src/utils/dataloader.py:18-32
 *                                                         src/real_time_inference.py:16-28
 * ToTensor -> Resize(crop, bicubic, tensor path) -> CenterCrop(crop) -> BGR->RGB -> Normalize(CLIP),
 * applied on the device to nf raw frames: frames_hwc_bgr device uint8 [nf][H][W][3] (OpenCV layout,
 * real_time_inference.py:39) -> out_nchw device fp32 [nf][3][crop][crop], the layout gitcap_encode reads.
 * Stateless (no handle). */
int gitcap_preprocess(const uint8_t* frames_hwc_bgr, int nf, int H, int W, float* out_nchw, int crop, void* stream);

/* The two calls above for RAW camera frames (what src/real_time_inference.py:39-57 has before its transform):
 * frames_hwc_bgr device uint8 [B][F][H][W][3].  The transform of gitcap_preprocess (crop = image_size) is fused with
 * the patch gather of the encoder (SURVEY.md par. 8f.1): the bf16 patch rows are written directly, no fp32 frame tensor
 * exists.  Results are bitwise those of gitcap_preprocess followed by gitcap_encode / gitcap_greedy. */
int gitcap_encode_raw(gitcap_t* h, const uint8_t* frames_hwc_bgr, int B, int F, int H, int W, float* visual_out, void* stream);
int gitcap_greedy_raw(gitcap_t* h, const uint8_t* frames_hwc_bgr, int B, int F, int H, int W, int max_len, int stop,
                      int64_t* ids_out, int32_t* steps_out, void* stream);

/* Replaces: F.log_softmax(scores) + beam_scores, view(B, beams*V), torch.topk(2*beams)
 *                                                         src/models/model.py:557-565
 * logits: device fp32 [B*beams][ld]; beam_scores: device fp32 [B*beams]; outputs: device
 * out_scores fp32 [B][K], out_idx int32 [B][K] (flat index beam*V + word), sorted descending,
 * ties by smaller flat index; K <= 16, beams <= 16.  Stateless (no handle). */
int gitcap_beam_topk(const float* logits, int ld, const float* beam_scores, int B, int beams, int V, int K,
                     float* out_scores, int32_t* out_idx, void* stream);

/* Replaces: GenerativeImageTextModel.infer + GeneratorWithBeamSearchV2.search
 *                                                         src/models/model.py:426-462, :479-678
 * (num_keep_best = 1, do_sample = False: the way GenerativeImageTextTeacher.forward drives it, :768).
 * The whole search runs on the device with no host round trip: per step decoder forward, beam top-k
 * (log-softmax + beam scores, :557-565), hypothesis/beam bookkeeping (:573-621) and the KV reorder the
 * reference leaves commented out (:623-634).  decoded_out: device int64 [B][max_steps], CLS-prefixed best
 * hypothesis padded with EOS (:671-675); logprobs_out: device fp32 [B] (:665).  The reference's early
 * `break` when every sentence is done (:640) is not needed for correctness (done sentences ignore their
 * candidates) and is not taken: all max_steps-1 steps are enqueued. */
int gitcap_beam_search(gitcap_t* h, const float* frames, int B, int F, int beams, int max_steps,
                       float length_penalty, int per_node_beam_size,
                       int64_t* decoded_out, float* logprobs_out, void* stream);

/* Pipelined form of gitcap_greedy for a stream of batches (no reference counterpart: the reference
 * processes one clip at a time, src/models/model.py:765).  submit enqueues the image pass on the
 * handle's encoder stream and the text loop on one of its two decoder streams, ordered after the work already
 * on `stream` (so `frames` may be produced there), and returns a ticket; at most FOUR submissions
 * may be in flight (four slots), so one batch's MFMA-bound image pass overlaps the latency-bound
 * token loops of the batches before it.  wait makes `stream` wait for that submission's ids_out/steps_out.
 * frames / ids_out / steps_out must stay valid until the wait. */
int gitcap_greedy_submit(gitcap_t* h, const float* frames, int B, int F, int max_len, int stop,
                         int64_t* ids_out, int32_t* steps_out, void* stream, int* ticket);
int gitcap_greedy_wait(gitcap_t* h, int ticket, void* stream);

/* The same pipelined form for gitcap_beam_search (BASELINE configs[4]; the reference's teacher runs this search one clip at a time,
 * src/models/model.py:762-768 with the defaults of :702-708): the image pass of one batch overlaps the search loops of the batches
 * before it.  Tickets of both submit forms share one sequence (at most FOUR submissions of either kind in flight);
 * gitcap_beam_search_wait is gitcap_greedy_wait under the name that pairs with this call.  visual_out (nullable): as in gitcap_encode
 * (the `visual_features` of the reference's output dict, model.py:460).  step_logits_out (nullable): device fp32
 * [max_steps - 1][B * beams][vocab], the raw logits of every search step (what model.py:521 appends to saved_logits).
 * Results are bitwise those of gitcap_beam_search. */
int gitcap_beam_search_submit(gitcap_t* h, const float* frames, int B, int F, float* visual_out, int beams, int max_steps,
                              float length_penalty, int per_node_beam_size,
                              int64_t* decoded_out, float* logprobs_out, float* step_logits_out, void* stream, int* ticket);
int gitcap_beam_search_wait(gitcap_t* h, int ticket, void* stream);

/* The two pipelined submissions for RAW camera frames in device memory (frames_hwc_bgr device uint8 [B][F][H][W][3], as in
 * gitcap_greedy_raw): the transform of src/utils/dataloader.py:18-32 / src/real_time_inference.py:16-28 fused with the patch
 * gather runs as the first launch of the image pass on the handle's encoder stream.  This is the form a HOST-fed caller uses
 * (src/real_time_inference.py:39-58 holds OpenCV frames in host memory; src/inference.py:45-51 a DataLoader's CPU tensor): the
 * caller enqueues the copy of batch i + 1 (a quarter of the bytes of the fp32 tensor) on the stream it passes as `stream`, so the
 * copy runs under batch i's compute and only the image pass waits for it (gitcap/model.py: _StagingRing is that caller; keep to
 * the caller's own stream -- with a fifth stream the runtime's four hardware queues are oversubscribed and a copy queues behind
 * a token loop: measured 1 618 against 2 004 captions/s, profiles/r06_host_fed_copy_stream.txt).  Results are bitwise those of gitcap_greedy_raw / of gitcap_preprocess +
 * gitcap_beam_search.  Tickets, slots and waits as for gitcap_greedy_submit. */
int gitcap_greedy_raw_submit(gitcap_t* h, const uint8_t* frames_hwc_bgr, int B, int F, int H, int W, int max_len, int stop,
                             int64_t* ids_out, int32_t* steps_out, void* stream, int* ticket);
int gitcap_beam_search_raw_submit(gitcap_t* h, const uint8_t* frames_hwc_bgr, int B, int F, int H, int W, float* visual_out,
                                  int beams, int max_steps, float length_penalty, int per_node_beam_size,
                                  int64_t* decoded_out, float* logprobs_out, float* step_logits_out, void* stream, int* ticket);

/* Host-side staging copy for host-fed callers (no reference counterpart): bytes from pageable memory (a DataLoader batch without
 * pin_memory, OpenCV frames) into a page-locked staging buffer, split over up to 8 threads -- as many as the process may really use
 * (affinity mask, cgroup CPU quota; GITCAP_HOST_COPY_THREADS overrides).  Plain memcpy semantics, blocking, no device work.
 * (A framework's own parallel copy may size its pool to the whole machine: under a CPU quota that costs tens of milliseconds per
 * 58 MB batch, measured; this is the copy gitcap/model.py: _StagingRing uses.) */
int gitcap_host_copy(void* dst, const void* src, int64_t bytes);

/* Health of the in-kernel statistics exchange (no reference counterpart).  The residual GEMMs that normalise their own output
 * rows exchange LayerNorm statistics between the workgroups of a row block (co-residency).  If a workgroup ever
 * gives up waiting (about 30 s: its siblings cannot become resident because another process or a CU-masked stream holds the
 * CUs) the kernel does NOT trap: it raises a host-visible flag and finishes with undefined rows.  Every entry point above
 * checks the flag first and gitcap_poll_errors checks it on demand (call it after synchronising the stream to vouch for the
 * results just produced).  Once raised: the device is drained, the flag is cleared, the handle switches for good to the
 * unfused GEMM + LayerNorm launches (bitwise the same results) and GITCAP_ERR_EXCHANGE is returned ONCE -- the caller
 * re-runs the calls whose results it had not yet vouched for.  Submissions (gitcap_greedy_submit / gitcap_beam_search_submit)
 * that were in flight at that moment are marked: their gitcap_*_wait returns GITCAP_ERR_EXCHANGE every time it is called, so a
 * retry cannot hand out their undefined ids; they must be submitted again.  Returns 0 when healthy. */
int gitcap_poll_errors(gitcap_t* h);

/* Beam reorder of the text part of the KV cache (what src/models/model.py:623-634 sketches):
 * new row r takes the cached text K/V of old row src_rows[r]; image K/V are shared. */
int gitcap_reorder_rows(gitcap_t* h, const int32_t* src_rows, int rows, int t_len, void* stream);
/* Instrumentation used by bench.py (no reference counterpart: the reference has no profiler,
 * SURVEY.md par. 5).  When enabled every launch of a kernel class is bracketed by two HIP events
 * recorded on the launch stream; gitcap_profile_read waits for them, sums the elapsed times and
 * the algorithmic flops/bytes of the bracketed launches, and resets the class. */
enum {
    GITCAP_PROF_GEMM = 0,       /* bf16 MFMA GEMM (patch-embed, qkv, proj, fc1, fc2, vproj) */
    GITCAP_PROF_ATTN_FULL = 1,  /* flash attention over frames / image prefix */
    GITCAP_PROF_SKINNY = 2,     /* text-row weight-streaming GEMMs (incl. vocabulary head) and their reduce+LayerNorm */
    GITCAP_PROF_ATTN_TEXT = 3,  /* text-row attention over the KV cache */
    GITCAP_PROF_ROWOPS = 4,     /* LayerNorm over the image rows (launches of its own) */
    GITCAP_PROF_GEMM_LN = 5,    /* the residual GEMMs that also normalise their output rows (counted here, not in class 0) */
    GITCAP_PROF_CLASSES = 6
};
int gitcap_profile_enable(gitcap_t* h, int enable);
int gitcap_profile_read(gitcap_t* h, int cls, double* ms_total, int64_t* launches,
                        double* flops_total, double* bytes_total);

/* Kernel-level test hooks (tests/test_kernels_gpu.py, tools/gemm_bench.py): run ONE kernel on
 * caller-owned device buffers.  gemm: out[m][n] = sum_k A[m][k]*W[n][k] (+epilogue `epi` of
 * csrc/kernels.h: 0 bias->bf16, 1 bias+QuickGELU->bf16, 2 bias+GELU->bf16, 3 bias+resid->f32,
 * 4 bias->f32); A [M][K] bf16, W [N][K] bf16; tile = 64, 128 or 256 (square tiles: M, N multiples of the tile). */
int gitcap_dbg_gemm(const void* A, const void* W, const float* bias, const float* resid, void* out,
                    int M, int N, int K, int epi, int tile, void* stream);
/* The fp8 tile kernel (csrc/gemm_f8.hip): A8 [M][K] and W8 [N][K] OCP e4m3 codes (M, N multiples of 256, K of 128), out = acc * ascale *
 * wscale[n] + bias; epi 4 -> fp32 [M][N], 0 -> bf16, 8 / 9 -> e4m3 codes of quick_gelu / erf_gelu (.) * out8_inv. */
int gitcap_dbg_gemm_f8(const void* A8, const void* W8, const float* wscale, float ascale, const float* bias, void* out,
                       int M, int N, int K, int epi, float out8_inv, void* stream);
/* GEMM + bias [+ resid] followed by LayerNorm of the output rows (N = 768 or 1024).  post = 0: out_f32 = x = A W^T + bias +
 * resid, out_bf16 = LN(x) (pre-LN block); post = 1: out_f32 = out_bf16 = LN(x), resid may be NULL (post-LN block).
 * fused = 1: inside the 256x256 kernel (the tiles of a 256-row block exchange segment statistics); fused = 0: the `tile` kernel,
 * then the row kernel.  Both produce the same bits (csrc/ln_canon.h). */
int gitcap_dbg_gemm_ln(const void* A, const void* W, const float* bias, const float* resid, const float* gamma,
                       const float* beta, float eps, float* out_f32, void* out_bf16, int M, int N, int K, int post,
                       int fused, int tile, void* stream);
/* Speed-only switches at run time (what the GITCAP_* environment variables set once per process; INTEGRATION.md par. 9), so that
 * one process can check that results do not depend on them.  key 0: GEMM + LayerNorm epilogue on/off; 1: one/two-row prologue
 * on/off; 2: 256x256-tile threshold; 3: 128x128-tile threshold; 4: retired (was: 224-row tiles for synchronous calls; accepted, no effect); 5: the greedy loop's
 * arg-max launch also embeds the next step's input rows on/off; 6: polls a fused GEMM + LayerNorm workgroup waits for its
 * siblings before it gives up (0 = default; 1 forces the fail-soft path of gitcap_poll_errors in a test); 7: the text rows'
 * FC1 -> GELU -> FC2 as one launch over hidden slices on/off; 8: fragment-major copies of the text-path weights at the next
 * gitcap_finalize_weights on/off; 9: 8-wave workgroups for text-attention launches of more (row, head) units than CUs on/off;
 * 10: the vocabulary head's four-tile workgroups that share the activation rows through LDS on/off; 11: three-wave workgroups
 * that share the slab reduce of the one/two-row prologue on/off.  Returns the previous value (< 0: bad key).  The switches are process-wide atomics: a call on another thread sees either value,
 * and either value gives the same bits. */
int gitcap_dbg_config(int key, int value);
/* attn_full: qkv [G*S][3*H*64] bf16 -> ctx [G*S][H*64] bf16 */
int gitcap_dbg_attn_full(const void* qkv, void* ctx, int G, int S, int H, void* stream);
/* layernorm: x fp32 [rows][D] -> out_f32 / out_bf16 (either may be NULL) */
int gitcap_dbg_layernorm(const float* x, const float* gamma, const float* beta, float eps, int rows, int D,
                         float* out_f32, void* out_bf16, void* stream);

/* Residual stream of the ViT per block (tests/test_stress_layers_gpu.py: single-block checks on the device's own inputs).
 * While `buf` is non-NULL every SYNCHRONOUS image pass (gitcap_encode / _greedy / _beam_search and their _raw forms) copies
 * the fp32 residual stream x [rows][enc_width] (rows = B * F * tokens per frame, unpadded) to buf + e * rows * enc_width:
 * e = 0 the ln_pre output, e = i the output of encoder block i - 1, for e < enc_layers (the last block's output only exists
 * behind ln_post: visual_out of gitcap_encode).  buf: device fp32 [enc_layers][rows][enc_width], caller owned; NULL disables. */
int gitcap_dbg_enc_tap(gitcap_t* h, float* buf);

/* Introspection used by tests and bench.py.  gitcap_weight_bytes counts the tensors as loaded; the fragment-major second copies of
 * the decoder / head weights that gitcap_finalize_weights makes for the token loop (+132 MB bf16 / +66 MB e4m3 at GIT-base), the
 * e4m3 staging panels and everything sized by max_batch / max_frames / max_text_len / max_beams are workspace.  The text-row
 * scratch of the four pipeline slots grows with rows = max_batch * max_beams * max_text_len: 48 fp32 FC2 slabs + 12 per-head
 * partials per row = 184 KB per row and slot (INTEGRATION.md par. 4 has the formula). */
int gitcap_workspace_bytes(const gitcap_t* h, int64_t* bytes);
int gitcap_abi_version(void);

/* ---------------------------------------------------------------------------------------------------
 * Student decoder (SURVEY.md par. 8 row f.2): StudentCandidateV1, src/models/model.py:50-187 --
 * torch.nn.TransformerDecoder (post-LN, ReLU; model.py:82-85) over the caption so far (causal mask,
 * PAD tokens masked as keys; model.py:134-136, src/utils/masking.py) with cross-attention over
 * `memory` = one token per frame (model.py:124), embedding + positional table / sqrt(d_model)
 * (model.py:140-144, :320-340) and a Linear vocabulary head (model.py:152).  The TinyViT image encoder
 * (timm, model.py:38) is NOT behind this ABI: the caller supplies memory [B][mem_tokens][d_model].
 * Tensor names are the reference's state_dict keys (embed.weight, pos_enc.pe,
 * decoder.layers.{i}.self_attn.in_proj_weight, ..., linear.weight, linear.bias); same ownership, error
 * and threading rules as the gitcap_* entry points above.
 * ------------------------------------------------------------------------------------------------- */
typedef struct gitcap_student gitcap_student_t;
struct gitcap_student_config {
    int32_t d_model, n_head, d_ffn, num_layers;      /* config.py:79-83: 576, 8, 1024, 2 */
    int32_t vocab_size, cls_token_id, sep_token_id;  /* 30522, 101, 102 */
    int32_t pad_token_id;                            /* 0: create_padding_mask default, masking.py:4 */
    int32_t mem_tokens;                              /* frames per clip (6) */
    int32_t max_pos;                                 /* rows of the positional table (500, model.py:324) */
    int32_t max_rows;                                /* largest batch the workspace is sized for */
    int32_t max_text_len;                            /* largest max_len of greedy / T-1 of forward_decoder (<= 63) */
    float ln_eps;                                    /* 1e-5 */
};
/* StudentCandidateV1.__init__ (model.py:55-106) for the decoder part */
int gitcap_student_create(const struct gitcap_student_config* cfg, int device, gitcap_student_t** out);
void gitcap_student_destroy(gitcap_student_t* h);
const char* gitcap_student_last_error(const gitcap_student_t* h);
/* load_state_dict (src/inference.py:38): host fp32 data, logical shape; GEMM weights are stored as bf16 */
int gitcap_student_load_tensor(gitcap_student_t* h, const char* name, const float* data, const int64_t* shape, int rank);
int gitcap_student_finalize(gitcap_student_t* h);
/* memory: device fp32 [B][mem_tokens][d_model] (model.py:124) -> bf16 + the cross-attention K/V of every layer */
int gitcap_student_set_memory(gitcap_student_t* h, const float* memory, int B, void* stream);
/* forward_decoder (model.py:128-154): ids device int64 [B][ld_ids] (T valid columns) -> logits device fp32
 * [B][T][vocab].  Needs a preceding set_memory with the same B.  A row whose first token is PAD yields NaN
 * for that position, as torch does (every key masked). */
int gitcap_student_forward_decoder(gitcap_student_t* h, const int64_t* ids, int ld_ids, int B, int T, float* logits, void* stream);
/* greedy_decode (model.py:156-187) from memory: ids_out device int64 [B][max_len+1], CLS-prefixed; the
 * token loop runs on the device with an exact KV cache; steps_out (device int32[1], nullable) = number of
 * generated tokens under the stop rule (enum gitcap_stop; GITCAP_STOP_ALL_SEP = model.py:184). */
int gitcap_student_greedy(gitcap_student_t* h, const float* memory, int B, int max_len, int stop,
                          int64_t* ids_out, int32_t* steps_out, void* stream);
/* StudentCandidateV1.beam_search (src/models/model.py:189-318) on the device, KV-cached, no host round trip: memory [B][F][D] fp32
 * (device), k beams (rows b * k + i; B * k <= max_rows, k <= 16), no end-of-sequence handling (as the reference);
 * ids_out [B][max_len] = the best beam of every clip, CLS first (model.py:317). */
int gitcap_student_beam_search(gitcap_student_t* h, const float* memory, int B, int k, int max_len, int64_t* ids_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GITCAP_H */
