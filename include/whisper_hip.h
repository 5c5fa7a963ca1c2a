/*
 * whisper_hip.h — C ABI of the MI355X-native Whisper hot path (libwhisper_hip.so).
 *
 * Drop-in boundary for KrArunT/whisper-rust-ort's one hot path
 *     clip -> log-mel -> encoder -> greedy decoder with KV past -> token ids.
 * The reference has no plugin API; the seam is three Rust functions plus the ONNX Runtime
 * session lifecycle in src/main.rs.  Each entry point below names the reference interface it
 * replaces (file:line in /root/reference).  INTEGRATION.md shows the Rust `extern "C"` block a
 * maintainer would add to bind them.
 *
 * Conventions
 *   - plain C: opaque handles, pointers and sizes only; no C++/torch types cross the ABI.
 *   - every function returns an int status (WH_OK == 0); nothing throws or aborts across the ABI.
 *     wh_last_error() gives the message of the last failure on a ctx (or globally for load errors).
 *     The reference's three `bail!` preconditions map to distinct codes (WH_ERR_EMPTY_AUDIO,
 *     WH_ERR_BAD_SHAPE, WH_ERR_STATE).
 *   - caller owns every host buffer in and out (outputs are caller-allocated with explicit
 *     capacities); the library owns device weights (wh_model) and per-stream workspaces + KV
 *     caches (wh_ctx).  Only the two handle types need a library-side free.
 *   - wh_model is immutable after load and may be shared by any number of wh_ctx / host threads
 *     (reference: `&Session` shared across the rayon pool, src/main.rs:890-919).  A wh_ctx is
 *     single-threaded and bound to one HIP stream (reference: per-thread IoBinding + `past` map,
 *     src/main.rs:786-791).
 *   - there is NO CPU fallback: every compute entry point needs a gfx950 device and fails with
 *     WH_ERR_HIP otherwise.
 */
#ifndef WHISPER_HIP_H
#define WHISPER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WH_ABI_VERSION 1

/* status codes */
#define WH_OK 0
#define WH_ERR_EMPTY_AUDIO 1 /* src/main.rs:414-416  bail!("Empty audio") */
#define WH_ERR_BAD_SHAPE 2   /* src/main.rs:710-716  bail!("Unexpected logits shape") / bad mel shape */
#define WH_ERR_STATE 3       /* src/main.rs:808-810  bail!("Missing cached decoder input"): decode before encode */
#define WH_ERR_ARG 4
#define WH_ERR_NOMEM 5
#define WH_ERR_HIP 6         /* no device / HIP runtime error / kernel image missing */
#define WH_ERR_IO 7          /* model dir, config.json, safetensors */
#define WH_ERR_UNSUPPORTED 8

/* arithmetic type of the Linear/conv/attention contractions */
#define WH_PREC_F32 0  /* exact-f32 MFMA (v_mfma_f32_16x16x4_f32): the token-for-token / 1e-3-logit mode */
#define WH_PREC_BF16 1 /* bf16 MFMA, f32 accumulate, f32 residual stream: the throughput mode */
#define WH_PREC_FP8 2  /* BASELINE configs[4]: Linear/QKV weights as OCP e4m3 codes with one f32 scale per output channel;
                          cross-attention K/V cache as e4m3 with one scale per (clip, layer, K|V, head).  Encoder GEMMs
                          fed by a LayerNorm or by the GELU output (Q|K, V^T, fc1, fc2, cross-K/V projection) run on the
                          fp8 matrix cores (v_mfma_scale_f32_16x16x128_f8f6f4) with MX-quantised activations — e4m3 codes
                          + one power-of-two exponent per (row, 32 columns) — wherever the model's widths allow
                          (DESIGN.md §4a); decoder GEMMs dequantise the codes in registers into the bf16 MFMA operand.
                          Convolutions, LayerNorm parameters, biases, embeddings (so the tied LM head) and the remaining
                          activations as in WH_PREC_BF16.  Reference analogue: ORT dynamic quantisation of MatMul/Gemm
                          (int8 weights, per-call quantised activations), quantize_onnx_int8.py:37-42 */
#define WH_PREC_F16X3 3 /* f32 results on the fp16 matrix cores: every operand of a contraction is the sum of two fp16
                          limbs (x = hi + lo, 22 significant bits), every product three v_mfma_f32_16x16x32_f16
                          (hi.hi + hi.lo + lo.hi, f32 accumulate); everything outside the contractions (residual stream,
                          LayerNorm, softmax, GELU, argmax) in f32 exactly as WH_PREC_F32.  Meets the reference's f32
                          results (src/main.rs:777: f32 logits from f32 ORT graphs) — tokens identical, logits within
                          1e-3 — at matrix-core speed.  Operands must lie within fp16 range (|x| < 65504),
                          as in any fp16 Whisper deployment. */

#define WH_N_FRAMES 3000      /* mel frames per 30 s window (src/main.rs:896) */
#define WH_CLIP_SAMPLES 480000 /* 30 s @ 16 kHz */

typedef struct wh_model wh_model;
typedef struct wh_ctx wh_ctx;

/* Geometry read from config.json (HF WhisperConfig field names in comments). */
typedef struct {
    int32_t n_mels;      /* num_mel_bins */
    int32_t d_model;     /* d_model */
    int32_t n_heads;     /* encoder_attention_heads == decoder_attention_heads; head_dim must be 64 */
    int32_t enc_layers;  /* encoder_layers */
    int32_t dec_layers;  /* decoder_layers */
    int32_t ffn;         /* encoder_ffn_dim == decoder_ffn_dim */
    int32_t vocab;       /* vocab_size */
    int32_t n_audio_ctx; /* max_source_positions (1500) */
    int32_t n_text_ctx;  /* max_target_positions (448) */
} wh_dims;

/* Stage timings of the last transcribe call on a ctx, in seconds — the reference's `Timing`
 * buckets (src/main.rs:1010-1016) measured with HIP events on the ctx stream. */
typedef struct {
    double preprocess_s; /* log-mel                       (src/main.rs:870-872) */
    double encode_s;     /* encoder                        } both are the reference's model_only_s */
    double decode_s;     /* cross-KV + greedy token loop   } (src/main.rs:964-967)                 */
    double total_s;      /* first H2D to last D2H, host wall clock */
    double h2d_s, d2h_s;
} wh_timing;

/* Greedy-decode parameters: the arguments of greedy_decode_with_past (src/main.rs:753-761) with
 * GenerationCfg (src/main.rs:102-106) flattened. */
typedef struct {
    const int64_t* prompt;         /* [n_prompt] — sot, lang, task, (notimestamps)  src/main.rs:851-855 */
    size_t n_prompt;
    size_t max_new_tokens;         /* --max-new-tokens; total generated ≤ this (src/main.rs:793) */
    int64_t eot;                   /* stop token (src/main.rs:781,820) */
    const int64_t* suppress;       /* generation_config.json suppress_tokens        (src/main.rs:765) */
    size_t n_suppress;
    const int64_t* begin_suppress; /* begin_suppress_tokens: first generated token only (766-768,778) */
    size_t n_begin_suppress;
    /* parity harness only (NULL/0 in production): generated token i is replaced by forced[i] as the
     * next input AFTER the argmax has been recorded, and EOT does not stop a forced step. */
    const int64_t* forced;
    size_t n_forced;
} wh_decode_params;

/* ---- model lifecycle: replaces build_session ×3 (src/main.rs:169-202, 1099-1108) ------------- */
/* `model_dir_or_spec` is what the CLI's --onnx-dir names: a directory holding config.json +
 * model.safetensors (HF layout), or the literal spec "synthetic:<preset>:<seed>" with preset in
 * {nano, micro, base, large-v3} for hash-seeded weights (whisper-rust-ort_amd/modelspec.py). */
int wh_model_load(const char* model_dir_or_spec, int device, int precision, wh_model** out);
/* Same, from a caller-held f32 blob in canonical tensor order (modelspec.tensor_table). */
int wh_model_create(const wh_dims* dims, const float* weights, size_t n_weights, int device, int precision,
                    wh_model** out);
void wh_model_free(wh_model* m);
int wh_model_get_dims(const wh_model* m, wh_dims* out);
int wh_model_precision(const wh_model* m);
/* Copies the f32 master copy of one tensor (HF state-dict name) to `out`; *n_out = element count.
 * Used to check the C++ synthetic generator against the numpy one. */
int wh_model_export_tensor(const wh_model* m, const char* name, float* out, size_t cap, size_t* n_out);

/* ---- per-stream context: workspace + KV cache for up to max_batch clips in flight ------------- */
#define WH_MAX_BATCH 2048   /* largest max_batch wh_ctx_create accepts */
int wh_ctx_create(wh_model* m, int max_batch, wh_ctx** out);
/* Chip partition (no reference counterpart; the reference overlaps windows on CPU threads, src/main.rs:884-919).
 * A context created with these options runs log-mel + encoder on one HIP stream and the cross-K/V projection + token
 * loop on another, each optionally confined to a set of compute units (hipExtStreamCreateWithCUMask), so that
 * wh_transcribe_batch_device_next can run the NEXT batch's MFMA-bound encoder beside this batch's HBM-bound token loop.
 * A CU mask is `words` 32-bit words; on MI355X bit i selects compute unit i / 8 of XCD i % 8 (256 bits), so the low n
 * bits are n compute units spread evenly over the 8 XCDs.  words == 0: the stream may use the whole chip. */
#define WH_CTX_TWO_STREAMS 1 /* separate encoder / decode streams even without CU masks */
/* Cross-attention of the token loop computed on the encoder states themselves instead of the per-layer projected K / V
 * (models of whisper-base geometry in the bf16, split-fp16 and fp8 modes — the states as bf16 rows, fp16 limb planes or e4m3 rows; algebraically
 * the same attention, half the bytes streamed per token and no cross-K/V cache).  Default: on for contexts of max_batch >= 256.  _ON on a model without the geometry is refused. */
#define WH_CTX_CROSS_ES_ON 2
#define WH_CTX_CROSS_ES_OFF 4
typedef struct {
    size_t struct_size;          /* sizeof(wh_ctx_opts): guards against a caller built for another layout */
    int max_batch;               /* as wh_ctx_create */
    int flags;                   /* WH_CTX_* */
    const uint32_t* enc_cu_mask; /* log-mel + encoder stream */
    size_t enc_cu_mask_words;
    const uint32_t* dec_cu_mask; /* cross-K/V projection + token loop stream */
    size_t dec_cu_mask_words;
} wh_ctx_opts;
int wh_ctx_create_ex(wh_model* m, const wh_ctx_opts* opts, wh_ctx** out);
void wh_ctx_free(wh_ctx* c);
/* what the token loop's cross-attention streams: 0 = the projected K / V of every decoder layer (2 Ld S d elements per clip and
 * token, the reference's present.{i}.encoder.{key,value}, src/main.rs:771-787), 1 = the encoder states (Ld S d), -1 = c is NULL */
int wh_ctx_cross_mode(const wh_ctx* c);
/* Workspace placement (contexts that run the encoder-state cross-attention at >= 1024 clips): wh_ctx_create_ex times that kernel on the fresh
 * workspace and, when it reads slow, builds further workspaces (three in all at most) while the earlier ones are held and keeps the fastest (the
 * kernel's launch time depends on where the workspace lies).  Returns the number of workspaces timed (0: step not taken; WH_PLACE=0 disables it), and the
 * microseconds per launch on the first and on the kept workspace.  No counterpart in the reference (ONNX Runtime owns its arena). */
int wh_ctx_placement(const wh_ctx* c, float* first_us, float* kept_us);
const char* wh_last_error(const wh_ctx* c); /* c == NULL: last load/create error of this thread */
int wh_get_timings(const wh_ctx* c, wh_timing* out);

/* ---- staged calls: the reference's three functions, one clip each ----------------------------- */
/* whisper_log_mel_80 (src/main.rs:407-509), n_mels taken from the model.  `pcm` is the WHOLE file
 * (any n ≥ 1): n_frames = wh_mel_frames(n); normalisation uses the global max over all frames.
 * mel_out: [n_mels][n_frames] row-major (mel-major), caller-allocated, cap_frames ≥ n_frames. */
size_t wh_mel_frames(size_t n_samples); /* src/main.rs:444-452 */
int wh_log_mel(wh_ctx* c, const float* pcm, size_t n_samples, float* mel_out, size_t cap_frames,
               size_t* n_frames_out);
/* run_encoder (src/main.rs:698-707): mel [n_mels][3000] host f32 → encoder states.  enc_out
 * ([n_audio_ctx][d_model] f32) may be NULL: the states always stay resident in the ctx for decode. */
int wh_encode(wh_ctx* c, const float* mel, float* enc_out);
/* greedy_decode_with_past (src/main.rs:753-829) on the encoder states held by the ctx.
 * tokens_out: prompt ++ generated (EOT included if hit), capacity n_prompt + max_new_tokens.
 * logits_out (optional, parity): [n_generated][vocab] f32, row i = logits that chose generated token i. */
int wh_decode_greedy(wh_ctx* c, const wh_decode_params* p, int64_t* tokens_out, size_t cap_tokens,
                     size_t* n_tokens_out, float* logits_out, size_t cap_logits_rows);

/* Parity harness (no reference counterpart: the reference decodes one window per call): the same greedy decode over
 * ALL clips whose encoder states are resident in the ctx — the n clips of the last wh_transcribe_batch* /
 * wh_transcribe_longform batch, or the one clip of wh_encode — run as ONE batch, i.e. through the batched kernel
 * variants the throughput path uses, with the logits read back.  tokens_out: [n][cap_tokens]; n_tokens_out: [n];
 * logits_out (optional): [n][cap_logits_rows][vocab] f32, row i of clip b = logits that chose its generated token i.
 * `forced` in the params applies the same token history to every clip.  *n_clips_out = n. */
int wh_decode_greedy_batch(wh_ctx* c, const wh_decode_params* p, int64_t* tokens_out, size_t cap_tokens, size_t* n_tokens_out,
                           size_t cap_clips, size_t* n_clips_out, float* logits_out, size_t cap_logits_rows);

/* The same, reading back the logits of n_rows chosen batch rows only (rows[i] in 0 .. n-1, each once): logits_out is
 * [n_rows][cap_logits_rows][vocab].  Tokens come back for every clip.  Lets the parity tests hold a 2048-clip context to the golden
 * vectors over a whole forced history (all rows' logits would be 10 GB). */
int wh_decode_greedy_rows(wh_ctx* c, const wh_decode_params* p, const int32_t* rows, size_t n_rows, int64_t* tokens_out, size_t cap_tokens,
                          size_t* n_tokens_out, size_t cap_clips, size_t* n_clips_out, float* logits_out, size_t cap_logits_rows);

/* ---- fused batch call: the body of transcribe_longform_chunked for ≤ max_batch single-window
 * clips (src/main.rs:870-915 / 946-967) run as one batch on the ctx stream -------------------- */
typedef struct {
    const float* pcm;  /* 16 kHz mono f32 */
    size_t n_samples;  /* 1 … 480000; shorter clips are zero-padded in normalised mel space (899-905) */
} wh_clip;
/* tokens_out: [n_clips][n_prompt + max_new_tokens] row-major; n_tokens_out: [n_clips]. */
int wh_transcribe_batch(wh_ctx* c, const wh_clip* clips, size_t n_clips, const wh_decode_params* p,
                        int64_t* tokens_out, size_t* n_tokens_out);
/* wh_transcribe_batch with the NEXT batch's host-to-device copy taken off the critical path (the reference's file loop loads file
 * i + 1 only after file i is done, src/main.rs:1164-1213; its `end_to_end_s` counts the load, :1190).  Transcribes `clips` like
 * wh_transcribe_batch and, if next_clips != NULL, copies their PCM into the context's second device buffer on a copy stream, beside this
 * batch's log-mel, encoder and token loop (asynchronous for page-locked host memory; pageable memory still works, staged by the
 * runtime).  A following call whose clips start at the same host address, with the same count and lengths, finds its PCM resident (or
 * on its way) and only waits for the copy's event; any other call copies as wh_transcribe_batch does.  next_clips' memory must stay
 * valid and unchanged until that call.  Results are identical to wh_transcribe_batch. */
int wh_transcribe_batch_next(wh_ctx* c, const wh_clip* clips, size_t n_clips, const wh_clip* next_clips, size_t n_next,
                             const wh_decode_params* p, int64_t* tokens_out, size_t* n_tokens_out);
/* Same with PCM already resident in device memory: d_pcm is [n_clips][480000] f32 on the ctx's
 * device (each clip exactly 30 s).  This is the entry bench.py times (inputs resident in HBM). */
int wh_transcribe_batch_device(wh_ctx* c, const float* d_pcm, size_t n_clips, const wh_decode_params* p,
                               int64_t* tokens_out, size_t* n_tokens_out);
/* Software-pipelined form of the same call (the reference's analogue is its window pool, src/main.rs:884-919: while one
 * window decodes, other threads already run the encoder of the next ones).  Transcribes the batch at d_pcm exactly like
 * wh_transcribe_batch_device and, if d_pcm_next != NULL, puts log-mel + encoder of the batch at d_pcm_next on the ctx's
 * encoder stream as soon as this batch's cross-K/V projection has been enqueued: they run beside this batch's token loop.
 * A following call whose (d_pcm, n_clips) equal that (d_pcm_next, n_clips_next) finds its encoder states resident and
 * starts with the cross-K/V projection; any other call simply recomputes them.  Results are identical to the
 * unpipelined call (same kernels, same order per clip).  d_pcm_next must stay valid and unchanged until that call. */
int wh_transcribe_batch_device_next(wh_ctx* c, const float* d_pcm, size_t n_clips, const float* d_pcm_next,
                                    size_t n_clips_next, const wh_decode_params* p, int64_t* tokens_out,
                                    size_t* n_tokens_out);

/* ---- long-form (src/main.rs:834-1008): whole-file mel once, 30 s windows every
 * (chunk_len - overlap) samples, all windows decoded as batches; returns per-window tokens ------ */
int wh_longform_plan(size_t n_samples, double chunk_length_s, double overlap_s, size_t* offsets, size_t cap,
                     size_t* n_chunks); /* chunk start samples, src/main.rs:858-882 */
int wh_transcribe_longform(wh_ctx* c, const float* pcm, size_t n_samples, double chunk_length_s, double overlap_s,
                           const wh_decode_params* p, int64_t* tokens_out /* [n_chunks][n_prompt+max_new] */,
                           size_t* n_tokens_out /* [n_chunks] */, size_t cap_chunks, size_t* n_chunks_out);

/* ---- measurement hooks (bench.py): time named kernel groups with HIP events on the ctx stream -- */
#define WH_KG_MEL 0
#define WH_KG_ENC_GEMM 1
#define WH_KG_ENC_ATTN 2
#define WH_KG_DEC_CROSS_ATTN 3
#define WH_KG_DEC_GEMM 4
#define WH_KG_DEC_OTHER 5
#define WH_KG_COUNT 6
/* group_mask bits 0-15: bit g set → launches of kernel group g are bracketed by hipEvents on the launch
 * stream (0 = off).  Bits 16-31: sampling stride S of the token loop — S <= 1: every decoder position is
 * launched eagerly and timed; S > 1: only every S-th generated position is (the others replay the captured
 * hipGraph, so the measurement barely perturbs the run).  The totals of the last transcribe call come back
 * as (milliseconds, number of timed launches) per group. */
int wh_profile_enable(wh_ctx* c, int group_mask);
int wh_profile_get(const wh_ctx* c, double* ms /* [WH_KG_COUNT] */, int64_t* launches /* [WH_KG_COUNT] */);

/* Host-only: the hash-seeded weights of "synthetic:<preset>:<seed>" as one f32 blob in canonical
 * tensor order (needs no device).  out == NULL → only *n_out is written. */
int wh_synthetic_weights(const char* preset, uint64_t seed, float* out, size_t cap, size_t* n_out);

/* OCP e4m3fn codes as WH_PREC_FP8 stores them (round to nearest even, saturating at +-448): host-side helpers for
 * the offline weight conversion (reference analogue: quantize_onnx_int8.py:31-42) and for tests. */
void wh_e4m3_quantize(const float* x, size_t n, uint8_t* codes);
void wh_e4m3_dequantize(const uint8_t* codes, size_t n, float* x);
int wh_abi_version(void);
int wh_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* WHISPER_HIP_H */
