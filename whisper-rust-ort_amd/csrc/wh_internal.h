// wh_internal.h — wh_model / wh_ctx definitions (internal to libwhisper_hip.so).
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/whisper_hip.h"
#include "wh_kernels.h"

struct EncLayerDev {
    void *qk_w, *v_w, *o_w, *fc1_w, *fc2_w;
    float *qk_b, *v_b, *o_b, *fc1_b, *fc2_b;
    float *ln1_w, *ln1_b, *ln2_w, *ln2_b;
    // WH_PREC_FP8: the matrices above hold the e4m3 code VALUES (exact in bf16); these are the per-output-channel scales
    float *qk_sc = nullptr, *v_sc = nullptr, *o_sc = nullptr, *fc1_sc = nullptr, *fc2_sc = nullptr;
    // ... and these the raw e4m3 codes [N][K] (one byte each) for the fp8-MFMA GEMMs (wh_gemm8_mx.hip)
    void *qk_w8 = nullptr, *v_w8 = nullptr, *fc1_w8 = nullptr, *fc2_w8 = nullptr;
    // WH_PREC_BF16, LayerNorm folded into the consumer GEMMs (wh_gemm8.hip, run_encoder's fold path): W (.) gamma in bf16,
    // s[n] = sum_k of the stored values, c[n] = bias[n] + sum_k beta[k] W[n][k]
    void *qk_wf = nullptr, *v_wf = nullptr, *fc1_wf = nullptr;
    float *qk_s = nullptr, *qk_c = nullptr, *v_s = nullptr, *v_c = nullptr, *fc1_s = nullptr, *fc1_c = nullptr;
};
struct DecLayerDev {
    void *qkv_w, *o_w, *cq_w, *co_w, *fc1_w, *fc2_w;
    float *qkv_b, *o_b, *cq_b, *co_b, *fc1_b, *fc2_b;
    float *ln1_w, *ln1_b, *ln2_w, *ln2_b, *ln3_w, *ln3_b;
    // LayerNorm folded into the consumer GEMMs: qkv_w/cq_w/fc1_w hold W ⊙ γ, *_b hold c[n], *_s hold s[n]
    float *qkv_s, *cq_s, *fc1_s;
    // WH_PREC_FP8: every matrix above is raw e4m3 codes [N][K] (one byte each) with these per-output-channel scales;
    // LayerNorm's γ is then applied on the activation side (the producers write x ⊙ γ_next into the slab) and
    // *_s hold sum_k γ[k] * W[n][k], *_b hold c[n], both of the dequantised weights
    float *qkv_sc = nullptr, *o_sc = nullptr, *cq_sc = nullptr, *co_sc = nullptr, *fc1_sc = nullptr, *fc2_sc = nullptr;
    // cross-attention on the encoder states (wh_cross_es.hip, wh_model::cross_es): cqx_w [H][d][64] = head h's rows of the cross W_k,
    // transposed (the query-side expansion); cv_w / cv_b = this layer's plain W_v rows and b_v inside the stacked cross-K/V projection
    void *cqx_w = nullptr, *cv_w = nullptr;
    float* cv_b = nullptr;
};

struct wh_model {
    wh_dims dims{};
    int prec = WH_PREC_BF16;
    int device = 0;
    size_t esz = 2;  // bytes per element of the compute dtype
    // f32 master copy in canonical order (modelspec.tensor_table) + name → (offset, count)
    std::vector<float> master;
    std::map<std::string, std::pair<size_t, size_t>> index;
    // device arena
    char* arena = nullptr;
    size_t arena_bytes = 0;
    int conv1_k = 0;  // 3*n_mels rounded up to 32
    void *conv1_w = nullptr, *conv2_w = nullptr, *tok_emb = nullptr, *cross_kv_w = nullptr;
    float *conv1_b = nullptr, *conv2_b = nullptr, *enc_pos = nullptr, *dec_pos = nullptr, *cross_kv_b = nullptr;
    float *enc_ln_w = nullptr, *enc_ln_b = nullptr, *dec_ln_w = nullptr, *dec_ln_b = nullptr;
    float* cross_kv_sc = nullptr;  // WH_PREC_FP8: scales of the cross K/V projection rows [Ld][2][d]
    void* cross_kv_w8 = nullptr;   // WH_PREC_FP8: the same rows as raw e4m3 codes
    void* cross_kv_wf = nullptr;   // WH_PREC_BF16: the same rows with the encoder's final LayerNorm folded in (+ s, c)
    float *cross_kv_s = nullptr, *cross_kv_c = nullptr;
    void* lm_w = nullptr;  // tied embedding ⊙ final-LN γ (LM head operand); WH_PREC_FP8: the embedding itself (γ on the activation side)
    float *lm_s = nullptr, *lm_c = nullptr;
    bool cross_es = false;   // the decoder layers carry cqx_w / cv_* (bf16, whisper-base geometry)
    std::vector<EncLayerDev> enc;
    std::vector<DecLayerDev> dec;
    // log-mel tables
    double* mel_tw = nullptr;
    float *mel_win = nullptr, *mel_fbT = nullptr;
};

struct wh_ctx {
    wh_model* m = nullptr;
    int max_batch = 1;
    int dec_cus = 256;              // compute units of the token-loop stream (its CU mask's population, else the device's count)
    hipStream_t stream = nullptr;   // cross-K/V projection + token loop (and everything else when s_enc == stream)
    // log-mel + encoder.  The same stream unless the context was created with wh_ctx_create_ex (CU masks / two streams):
    // then the next batch's encoder runs beside this batch's token loop, each on its own part of the chip.
    hipStream_t s_enc = nullptr;
    hipStream_t cur = nullptr;      // stream of the phase being launched (event brackets of the profiling hooks)
    // hand-offs between the two streams: encoder states ready (recorded on s_enc), encoder states consumed by the
    // cross-K/V projection (recorded on stream; the next encoder pass may overwrite them)
    hipEvent_t ev_enc_done = nullptr, ev_kv_done = nullptr;
    bool kv_done_armed = false;
    // stage events of an encoder pass {start, mel done, encoder done}: two sets, because a prefetched pass (the NEXT
    // batch's) is recorded while the resident pass's times have not been read yet
    hipEvent_t enc_ev[2][3] = {{nullptr}};
    int enc_set = 0;                // set of the resident encoder states
    // encoder states prefetched by wh_transcribe_batch_device_next for its `next` batch
    const float* pre_pcm = nullptr;
    int pre_n = 0;
    bool pre_valid = false;
    std::string err;
    wh_timing timing{};
    bool have_enc = false;  // encoder states of `enc_batch` clips are resident
    int enc_batch = 0;
    // profiling hooks
    bool prof = false;
    int prof_mask = 0;  // bit g set: kernel group g is bracketed by events
    int prof_stride = 0;     // > 1: only every stride-th generated position is launched eagerly with events
    bool capturing = false;  // a decode step is being captured into a hipGraph
    bool no_graph = false;  // WH_NO_GRAPH=1: launch every decode step eagerly
    bool sync_every_pos = false;  // WH_SYNC_EVERY_POS=1: host waits after every decoder position (profiler triage only)
    int prof_group = -1;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events[WH_KG_COUNT];
    size_t prof_used[WH_KG_COUNT] = {0};
    double prof_ms[WH_KG_COUNT] = {0};
    int64_t prof_launches[WH_KG_COUNT] = {0};
    hipEvent_t ev[8] = {nullptr};

    // ---- device workspace (sized for max_batch clips) ----
    char* ws = nullptr;
    size_t ws_bytes = 0;
    float* pcm = nullptr;       // [B][480000]
    size_t pcm_cap = 0;         // samples allocated behind `pcm_long` for whole-file mel
    float* pcm_long = nullptr;  // staged-API / long-form whole file (grown on demand)
    float* raw_long = nullptr;  // [n_mels][frames] raw log-mel of a whole file (grown on demand)
    size_t raw_long_cap = 0;    // frames
    float* mel_out_long = nullptr;
    int* d_nsamp = nullptr;     // [B]
    int* d_nframes = nullptr;   // [B]
    int* d_src_index = nullptr; // [B]
    int* d_frame_start = nullptr;
    unsigned* d_gmax = nullptr; // [B]
    float* raw = nullptr;       // [B][n_mels][3008]
    float* mel_stage = nullptr; // [n_mels][3000] staged-API mel upload
    void* melT = nullptr;       // [B][3002][n_mels] (+slack)
    void* h1 = nullptr;         // [B][3001][d]
    float* x = nullptr;         // [B][S][d] f32 residual stream
    void* xn = nullptr;         // [B][S][d]
    void* qk = nullptr;         // [B][S][2d]
    void* vT = nullptr;         // [B][d][ldv]
    void* att = nullptr;        // [B][S][d]
    void* hbuf = nullptr;       // [B][S][ffn]
    void* enc_out = nullptr;    // [B][S][d] compute dtype (cross-KV GEMM operand)
    // WH_PREC_FP8, MX activations (e4m3 codes + E8M0 block exponents [row][4][K/128]) when the geometry allows (mx_ok)
    bool mx_ok = false;
    unsigned char *xn8 = nullptr, *xn8_sc = nullptr;   // LayerNorm outputs [B*S][d]
    unsigned char *h8 = nullptr, *h8_sc = nullptr;     // GELU(fc1) [B*S][ffn]
    float* enc_out_f32 = nullptr;  // [B][S][d] f32 (API output)
    // WH_PREC_BF16 with the encoder LayerNorms folded into their consumer GEMMs (enc_fold): the residual stream again as bf16
    // (written by the producing GEMM's epilogue; the consumers' operand), the producers' partial sums and {mean, rstd} per row
    bool enc_fold = false;
    bool enc_mlp = false;    // ... and the feed-forward block of a layer as one launch (wh_mlp.hip) where its geometry is covered
    void* xb = nullptr;            // [B][S][d] bf16
    float* enc_part = nullptr;     // [d/64][B*S][2]
    float* enc_stat = nullptr;     // [B*S][2]
    float* enc_shift = nullptr;    // [B*S] running row offsets of the folded LayerNorms (GemmArgs::row_shift)
    float* enc_shift0 = nullptr;   // [B*S] the first producer's offsets: the row means of the positional table (conv2 adds it), fixed
    int ldv = 0;
    void* cross_kv = nullptr;   // [Ld][2][B][S][d]
    // cross-attention on the encoder states (wh_cross_es.hip): decided at creation from the model and the context's capacity.
    // es_E [B][S][d] = the encoder's final LayerNorm output in the compute dtype, decode-side storage like cross_kv (which is
    // then not allocated); dqe [B][H d] f32 expanded queries; dctx = the H d context values per clip, slab layout
    bool cross_es = false;
    void* es_E = nullptr;
    int es_rows = 0;            // rows from one clip's states to the next in es_E (>= n_audio_ctx)
    // workspace placement step of wh_ctx_create_ex: workspaces timed (0: step not taken), the cross-attention launch time on the first and on the kept one
    int place_tries = 0;
    float place_us_first = 0.0f, place_us_kept = 0.0f;
    int es_rows_cap = 0;        // rows per clip the buffer was sized for
    float* dq32 = nullptr;      // [B][d] f32 cross-attention queries (pre-scaled)
    float* dqe = nullptr;
    void* dctx = nullptr;
    void* cross_kv8 = nullptr;  // WH_PREC_FP8: the same planes as e4m3 codes (cross_kv is then the bf16 staging copy)
    float* kv_amax = nullptr;   // WH_PREC_FP8: [Ld][2][B][H] max|.| of each head's block (scale = amax / 448)
    void *self_k = nullptr, *self_v = nullptr;  // [Ld][B][H][TC][64]
    // decode step buffers
    float* dx = nullptr;        // [B][d]
    void* dxn = nullptr;        // [B][d]
    void* dxs = nullptr;        // raw residual rows, compute dtype, slab layout [d/32][mpad][32]
    float* lnpart = nullptr;    // LayerNorm partial sums [d/16][mpad][2]
    float* dshift = nullptr;    // [mpad] running row offsets of the decoder's folded LayerNorms (SkinnyArgs::row_shift / shift_io)
    void* dqkv = nullptr;       // [B][3d]
    void* datt = nullptr;       // [B][d]
    void* dq = nullptr;         // [B][d]
    void* dh = nullptr;         // [B][ffn]
    float* cpart = nullptr;     // [B][splits][d]
    float* cml = nullptr;       // [B][splits][H][2]
    float* part_val = nullptr;  // [B][n_tiles]
    int* part_idx = nullptr;
    int *feed = nullptr, *out_tokens = nullptr, *n_out = nullptr, *done = nullptr, *forced = nullptr, *pos = nullptr;
    int* step_ticket = nullptr;  // zeroed at creation, re-armed by its last arriver
    unsigned *mask_first = nullptr, *mask_base = nullptr;
    float* logits = nullptr;    // optional parity buffer (grown on demand)
    size_t logits_cap = 0;
    int* logits_sel = nullptr;  // [B]: slot of a batch row in `logits` or -1 (wh_decode_greedy_rows)
    // wh_transcribe_batch_next: the next batch's PCM copied to a second device buffer on a copy stream beside this batch's work
    float* pcm2 = nullptr;      // [B][480000], allocated at the first call of that entry
    hipStream_t s_copy = nullptr;
    hipEvent_t ev_h2d = nullptr;
    int h2d_buf = 0;            // buffer (0: pcm, 1: pcm2) that holds / will hold the prefetched batch
    const float* h2d_src = nullptr;   // host address of the prefetched batch's first clip
    int h2d_n = 0;
    bool h2d_valid = false;
    std::vector<int> h2d_ns, h2d_nf;  // its per-clip sample / frame counts
    int tok_ld = 0;
    int mpad = 16;              // row pitch of the k-slab-major decode activations (multiple of 16)
    int cross_splits = 1;
    // the decode GEMMs as LDS-DMA tile GEMMs (wh_dec_tile.hip): decided at creation from the model and the context's capacity
    bool dec_tile = false;
    // The captured decode step, kept across calls: every kernel argument the step bakes in is either a fixed workspace
    // address or one of the values below, so a call with the same key replays the instantiated graph as it is.
    // Destroyed in wh_ctx_free (after the stream has drained) or when the key changes.
    struct StepKey {
        int nb = 0, n_prompt = 0, eot = 0, n_forced = 0, logits_rows = 0;
        const float* d_logits = nullptr;
        const int* d_sel = nullptr;
        bool operator==(const StepKey& o) const {
            return nb == o.nb && n_prompt == o.n_prompt && eot == o.eot && n_forced == o.n_forced && logits_rows == o.logits_rows &&
                   d_logits == o.d_logits && d_sel == o.d_sel;
        }
    } step_key;
    hipGraph_t step_graph = nullptr;
    hipGraphExec_t step_exec = nullptr;
};

// Linear weights that arrived already quantised (F8_E4M3 + "<name>_scale" in model.safetensors, written by
// quantize_fp8.py): keyed by the tensor's offset in the f32 master blob, which then holds dequant(code) * scale.
// WH_PREC_FP8 uses these codes and scales as they are instead of quantising the master copy again.
struct WhPreQuantEntry {
    std::vector<uint8_t> codes;  // [rows][cols]
    std::vector<float> scale;    // [rows]
};
typedef std::map<size_t, WhPreQuantEntry> WhPreQuant;

// wh_model.cpp
int wh_model_build(const wh_dims& dims, std::vector<float>&& master, int device, int precision, wh_model** out,
                   const WhPreQuant* pre = nullptr);
void wh_synth_weights(const wh_dims& dims, uint64_t seed, std::vector<float>& out);
void wh_tensor_table(const wh_dims& dims, std::vector<std::pair<std::string, std::vector<int64_t>>>& out);
bool wh_preset_dims(const std::string& name, wh_dims* out);
uint8_t wh_e4m3_from_f32(float x);
float wh_e4m3_to_f32(uint8_t c);
int wh_load_model_dir(const std::string& dir, wh_dims* dims, std::vector<float>& master, WhPreQuant* pre = nullptr);
