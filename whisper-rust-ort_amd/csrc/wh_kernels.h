// wh_kernels.h — host-side launchers of the gfx950 kernels (internal to libwhisper_hip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <vector>

constexpr int WH_SMALL_CTX_CLIPS = 8;

struct GemmArgs {
    const void* A = nullptr;   // [M rows][K], row m at (m / m_per) * a_bs + (m % m_per) * lda
    long lda = 0, a_bs = 0, a_zs = 0;
    const void* W = nullptr;   // [N][K] k-contiguous
    long ldw = 0, w_zs = 0;
    void* C = nullptr;         // row m at (m / m_per) * c_bs + (m % m_per) * ldc
    long ldc = 0, c_bs = 0, c_zs = 0;
    const float* bias = nullptr;
    int bias_mode = 0;         // 0 none, 1 per column n, 2 per row m
    const float* wscale = nullptr;  // WH_PREC_FP8: accumulator scale (the weight's per-output-channel scale), indexed like bias
    const float* R = nullptr;  // f32 residual / positional table, same row addressing with ldr, r_bs
    long ldr = 0, r_bs = 0, r_zs = 0;
    int act = 0;               // 1 = erf GELU (applied before the residual add)
    int M = 0, N = 0, K = 0;
    int m_per = 1 << 30;
    int n_per = 1 << 30;       // column n stored at (n / n_per) * c_ns + (n % n_per)
    long c_ns = 0;
    int batch = 1;             // gridDim.z, pointer strides *_zs
    // the context is sized for a few clips only: 256-row tiles would leave most of the chip idle (6 row tiles per clip), so the
    // 128 x 128 kernel runs instead (1 clip: 0.57 vs 1.17 ms of encoder GEMMs, 4 clips: 0.93 vs 1.33, 16 clips: 2.48 vs 2.19).
    // Chosen from the context's capacity, never from the call's clip count: a clip decodes identically alone or in a batch.
    bool small_ctx = false;
    // WH_PREC_F16X3: operands are fp16 limb pairs (h2) unless this is set — f32 rows split at the fragment loads (conv1, whose overlapping
    // rows of the token-major log-mel start at multiples of n_mels elements, not of the 32-element h2 block)
    bool f32_operands = false;
    // wh_gemm8_mx.hip (WH_PREC_FP8): A and W hold e4m3 codes (one byte per element; lda, ldw, a_bs, *_zs in elements = bytes)
    // with E8M0 block exponents per (row, 32 consecutive k), layout [row][4][wh_mx_nkp(K)]; null = no block exponents (weights)
    const unsigned char* a_sc = nullptr;
    const unsigned char* w_sc8 = nullptr;
    long a_sc_zs = 0, w_sc_zs = 0;
    unsigned char* c_sc = nullptr;   // MX output: block exponents of C (C itself receives the codes), layout [row][4][wh_mx_nkp(N)]
    // LayerNorm folded around the GEMM (wh_gemm8.hip, bf16 encoder path): LN(x) W^T = rstd (x (gamma . W)^T - mean s) + c.
    //  consumer: y = rstd_i (acc - mean_i ln_s) + bias with {mean_i, rstd_i} = ln_stat[2 i .. 2 i + 1], i = the output ROW m
    //            (ln_mode 1: ln_s per column n, indexed like bias_mode 1) or the output COLUMN n (ln_mode 2: the per-clip V^T
    //            product, whose columns are the keys; ln_s per row m, like bias_mode 2; ln_stat advances by ln_stat_zs per batch)
    const float* ln_stat = nullptr;
    long ln_stat_zs = 0;
    const float* ln_s = nullptr;
    int ln_mode = 0;
    //  producer of a LayerNorm input (contiguous f32 rows, out_f32): besides C also write the rows as bf16 (xb_out, C's
    //            addressing) and, per (64-column group, row), the partial sums {sum v, sum v^2} into stats_out[group][stats_rows][2]
    void* xb_out = nullptr;
    float* stats_out = nullptr;
    long stats_rows = 0;
    //            Row centring (round 4): xb_out and the partial sums are taken of v - row_shift[m] — the row's running offset, kept by
    //            k_ln_stats as (previous offset + the mean measured on the shifted rows), i.e. the previous LayerNorm's true row mean.  A
    //            LayerNorm is invariant under a per-row shift of its input (the consumers' formula holds with the shifted rows' own mean),
    //            but the bf16 rounding of the raw rows and the E[x^2] - mean^2 variance are not: with |mean| >> spread they lose the
    //            row's content (measured: encoder error 0.97 against 0.07 at mean 25, spread 1).  nullptr: no shift (the first producer).
    const float* row_shift = nullptr;
};

// wh_mlp.hip: the encoder layer's feed-forward block in one launch (bf16 operands, LayerNorm fold on, d_model 512)
struct MlpArgs {
    const void* X = nullptr;        // [M][d] bf16: the residual stream's rows minus their running offset (the previous producer's xb_out)
    long ldx = 0;
    const float* ln_stat = nullptr; // [M][2] {mean, rstd} of those rows (k_ln_stats)
    const void* W1 = nullptr;       // [F][d] bf16, gamma folded in
    const float* s1 = nullptr;      // [F] row sums of the folded weights (GemmArgs::ln_s)
    const float* c1 = nullptr;      // [F] folded bias
    const void* W2 = nullptr;       // [d][F] bf16
    const float* b2 = nullptr;      // [d]
    float* Xres = nullptr;          // [M][d] f32 residual stream, updated in place
    long ldr = 0;
    void* xb_out = nullptr;         // [M][d] bf16 copy of the new rows minus row_shift (may be X: a workgroup reads its own rows before it writes them)
    float* stats_out = nullptr;     // [d / 64][stats_rows][2] partial {sum, sum of squares} per (64-column group, row)
    long stats_rows = 0;
    const float* row_shift = nullptr;
    int M = 0, d = 0, F = 0;
};
bool wh_enc_mlp_applicable(const MlpArgs& a);
int wh_launch_enc_mlp(hipStream_t s, const MlpArgs& a);

struct SkinnyArgs {
    const void* X = nullptr;   // activations, k-slab-major [K/32][x_mpad][32] (compute dtype)
    int x_mpad = 64;
    const void* W = nullptr;   // [N][K]
    const float* bias = nullptr;
    // WH_PREC_FP8: W holds e4m3 codes (one byte per element, dequantised in registers to the bf16 MFMA operand) and
    // the accumulator is multiplied by wscale[n] before everything else
    const float* wscale = nullptr;
    const float* xgamma = nullptr;  // WH_PREC_FP8 producers: xslab_out receives v * xgamma[n] (the next LayerNorm's γ)
    const float* R = nullptr;  // residual f32 [M][ldr]
    long ldr = 0;
    void* C = nullptr;         // output: row-major [M][ldc] (c_mpad == 0) or slab [N/32][c_mpad][32]
    long ldc = 0;
    int c_mpad = 0;
    int M = 0, N = 0, K = 0, act = 0;
    // LayerNorm folded into this GEMM (consumer): y = rstd[m] * (acc - mean[m] * ln_s[n]) + bias[n], with
    // mean/rstd of row m reduced from ln_tiles per-tile partial sums {sum x, sum x^2} [ln_tiles][x_mpad][2]
    const float* ln_part = nullptr;
    int ln_tiles = 0;
    const float* ln_s = nullptr;
    // X given as split cross-attention partials instead of a slab (X == nullptr): row m, column k is
    //   sum_s w_s * xpart[m][s][k] / sum_s w_s * l_s,  w_s = exp(m_s - max_s m_s),  {m_s, l_s} = xml[m][s][head(k)][0..1]
    // (the merge of the key ranges happens in the consumer: no cross-workgroup exchange inside the attention kernel)
    const float* xpart = nullptr;  // [M][x_splits][K] unnormalised partial outputs
    const float* xml = nullptr;    // [M][x_splits][x_heads][2]
    int x_splits = 0, x_heads = 0;
    // producer of the next LayerNorm's input: besides C (f32 row-major residual stream) also write the
    // raw rows in the compute dtype, slab layout, and this column tile's partial sums
    void* xslab_out = nullptr;   // [N/32][x_mpad][32]
    float* stats_out = nullptr;  // [N/16][x_mpad][2]
    // Row centring of the folded LayerNorms (see GemmArgs::row_shift): producers write the slab copy and the partial sums of
    // v - row_shift[m]; the one consumer of a LayerNorm adds the mean it measured to shift_io[m] (its first column tile's workgroups
    // only), so the next producer subtracts the row's true mean of one LayerNorm earlier.  The embedding kernels start a position with
    // the exact row mean.  Both [x_mpad] f32, the same array.
    const float* row_shift = nullptr;
    float* shift_io = nullptr;
    // grouped form (gridDim.z = zn independent products sharing M, N, K): group z reads X + z * x_zs and W + z * w_zs (elements),
    // bias + z * bias_zs, and writes C + z * c_zs (elements).  The per-head V projection of the encoder-state cross-attention:
    // X = head z's 512 context values (16 k-slabs), W = rows 64 z .. of W_v, C = columns 64 z .. of the attention output slab.
    // (plain consumers only: no LayerNorm fold, residual, statistics or position ticket)
    int zn = 1;
    long x_zs = 0, w_zs = 0, c_zs = 0, bias_zs = 0;
    // position advance by the last workgroup of the last kernel of a step
    int* ticket = nullptr;
    int* pos_w = nullptr;
    // LM-head mode
    const int* pos_p = nullptr;
    int n_prompt = 0;
    const unsigned* mask_first = nullptr;
    const unsigned* mask_base = nullptr;
    float* logits = nullptr;   // optional [M][logits_rows][N], or — logits_sel given — [selected][logits_rows][N]
    int logits_rows = 0;
    const int* logits_sel = nullptr;   // [M]: slot of row m in `logits`, -1 = this row's logits are not kept
    float* part_val = nullptr;
    int* part_idx = nullptr;
};

// embedding of the NEXT position fused into the argmax finish (tok_emb == nullptr: not fused)
struct NextEmbed {
    const void* tok_emb = nullptr;   // [vocab][d] compute dtype
    const float* pos_emb = nullptr;  // [n_text_ctx][d]
    float* x = nullptr;              // [B][d] f32 residual stream
    void* xslab = nullptr;           // slab copy (compute dtype), scaled by xgamma when given
    float* stats = nullptr;          // [mpad][2] {sum x, sum x^2}
    float* shift = nullptr;          // [mpad]: the row's mean, subtracted from the slab copy and the sums (SkinnyArgs::row_shift)
    const float* xgamma = nullptr;
    int d = 0, mpad = 0;
};

struct DecodeState {
    int* feed = nullptr;        // [B][tok_ld] input token per position
    int* out_tokens = nullptr;  // [B][tok_ld] prompt ++ generated
    int* n_out = nullptr;       // [B]
    int* done = nullptr;        // [B]
    const int* forced = nullptr;
    int n_forced = 0;
    int n_prompt = 0;
    int eot = 0;
    int tok_ld = 0;
};

void wh_build_mel_tables(int n_mels, std::vector<double>& tw, std::vector<float>& win, std::vector<float>& fbT);
void wh_launch_mel_stft(hipStream_t s, const float* pcm, long pcm_stride, const int* n_samples, int n_clips,
                        long max_frames, const double* tw, const float* win, const float* fbT, int n_mels, float* raw,
                        long raw_clip_stride, long raw_row_stride, unsigned* gmax);
void wh_launch_mel_norm(hipStream_t s, const float* raw, long raw_row_stride, const unsigned* gmax, int n_mels,
                        long n_frames, float* out);
template <typename T>
void wh_launch_mel_tokens(hipStream_t s, const float* src, long src_clip_stride, long src_row_stride,
                          const int* src_index, const int* frame_start, const int* n_frames_src, const unsigned* gmax,
                          int mode, int n_mels, int n_out, T* tok, long tok_clip_stride);

// The GEMM launchers return WH_OK or an error code with the reason in wh_set_error: a launch that cannot run as asked is
// reported to the caller, never skipped silently (run_encoder / run_decode propagate it).
int wh_launch_gemm(hipStream_t s, int prec, bool out_f32, const GemmArgs& g);
bool wh_gemm8_enabled();   // false under WH_GEMM8=0 (A/B switch): every GEMM on k_gemm, so no folded LayerNorm either
// wh_gemm8.hip: the 8-wave LDS-DMA kernel (bf16 operands) for problems with at least one full 256 x 128 tile
bool wh_gemm8_applicable(const GemmArgs& g);
// {mean, rstd} per row from the producers' partial sums: stat[row][2] <- partials[groups][rows][2] (groups added in order)
// shift (optional, [rows]): the rows' running offsets — shift[r] = shift_in[r] (the offset the producer subtracted; nullptr: none) + mean[r];
// shift_in == shift updates them in place
void wh_launch_ln_stats(hipStream_t s, const float* partials, int groups, long rows, int d, float* stat, float* shift = nullptr, const float* shift_in = nullptr);
int wh_launch_gemm8(hipStream_t s, bool out_f32, const GemmArgs& g);
// wh_gemm8x.hip: WH_PREC_F16X3 on 256 x 256 LDS-DMA tiles, h2 operands, f32 or h2 results
bool wh_gemm8x_applicable(const GemmArgs& g, bool out_h2);
int wh_launch_gemm8x(hipStream_t s, bool out_h2, const GemmArgs& g);
// MX block exponents of a [rows][K] operand: [row][4][nkp] bytes, nkp = K-steps of 128 rounded up to a multiple of 4 (from 4 on)
// so that the bytes of 16 consecutive K-steps are four aligned dwords
__host__ __device__ inline int wh_mx_nkp(int K) { const int nk = K >> 7; return nk < 4 ? nk : (nk + 3) / 4 * 4; }
inline bool wh_mx_ln_width(int d) { return d == 256 || d == 512 || d == 1024 || d == 1280 || d == 1536 || d == 2048; }   // k_layernorm_mx instantiations
// wh_gemm8_mx.hip: e4m3 x e4m3 on v_mfma_scale_f32_16x16x128_f8f6f4 (out: 0 bf16, 1 f32, 2 MX codes + exponents), and the
// LayerNorm that produces MX activations
bool wh_gemm8_mx_applicable(const GemmArgs& g);
int wh_launch_gemm8_mx(hipStream_t s, int out, const GemmArgs& g);
int wh_launch_layernorm_mx(hipStream_t s, const float* x, const float* w, const float* b, void* codes, void* exps, long rows, int d);
void wh_launch_layernorm(hipStream_t s, int prec, const float* x, const float* w, const float* b, void* y, long rows,
                         int d);
// the same with the output in blocks: input rows [i * in_blk, (i + 1) * in_blk) go to output rows i * out_blk ...  (in_blk == 0: contiguous)
void wh_launch_layernorm_blocks(hipStream_t s, int prec, const float* x, const float* w, const float* b, void* y, long rows, int d, int in_blk,
                                int out_blk);
void wh_launch_enc_attn(hipStream_t s, int prec, const void* qk, const void* vT, void* out, int n_clips, int S, int d,
                        int n_heads, int ldv);

void wh_launch_dec_gemm(hipStream_t s, int prec, bool out_f32, const SkinnyArgs& a);
// wh_dec_tile.hip: the same contract on 128 x {128, 64} LDS-DMA tiles for contexts of a thousand clips and more (bf16 / f16x3 operands; not
// bit-identical to k_dec_gemm, so chosen per context: wh_ctx::dec_tile)
bool wh_dec_tile_applicable(int prec, const SkinnyArgs& a);
void wh_launch_dec_tile(hipStream_t s, int prec, bool out_f32, const SkinnyArgs& a);
void wh_launch_dec_embed(hipStream_t s, int prec, const void* tok_emb, const float* pos_emb, const int* feed, int feed_ld,
                         const int* pos_p, float* x, void* xslab, float* stats, int rows, int d, int mpad, const float* xgamma, float* shift);
void wh_launch_lm_head(hipStream_t s, int prec, const SkinnyArgs& a);
// wh_gemm8.hip: the same contract on 256 x 256 LDS-DMA tiles for hundreds of rows (bit-identical logits; bf16 operands)
bool wh_lm_head_tile_applicable(const SkinnyArgs& a);
int wh_lm_head_tile_parts(const SkinnyArgs& a);
void wh_launch_lm_head_tile(hipStream_t s, const SkinnyArgs& a);
// wh_gemm8x.hip: the same for WH_PREC_F16X3 (h2 operands)
bool wh_lm_head_tile_x3_applicable(const SkinnyArgs& a);
int wh_lm_head_tile_x3_parts(const SkinnyArgs& a);
void wh_launch_lm_head_tile_x3(hipStream_t s, const SkinnyArgs& a);
int wh_lm_head_parts(int prec, const SkinnyArgs& a);  // argmax partials per row written by wh_launch_lm_head, layout [part][x_mpad]
void wh_launch_argmax_finish(hipStream_t s, int prec, const float* part_val, const int* part_idx, int n_parts, int mpad, int* pos_p,
                             int* ticket, const DecodeState& st, int B, const NextEmbed& ne);
void wh_launch_dec_self_attn(hipStream_t s, int prec, const void* qkv, void* kc, void* vc, void* out, const int* pos_p,
                             int d, int n_heads, int tc, int B, int mpad);
// stream_nt: non-temporal K/V loads (set when the cross K/V of all layers exceed what the Infinity Cache can keep)
// splits == 1: the normalised output goes straight to `out` (slab layout, row pitch mpad); else partials for the consumer's merge
void wh_launch_dec_cross_attn(hipStream_t s, int prec, const void* q, const void* ck, const void* cv, float* part,
                              float* ml, int S, int d, int n_heads, int splits, int B, void* out, int mpad, bool stream_nt);

// wh_cross_es.hip: the same attention computed on the encoder states themselves (bf16, whisper-base geometry): qe [B][H][d] f32
// expanded queries, E [B][S][d] bf16, out = the H * d context values per clip as a decode-GEMM operand (slab layout, pitch mpad)
bool wh_cross_es_geometry(int d, int n_heads, int S);
// expanded queries qe[m][h][j] = sum_t q[m][64 h + t] * wkT[h][j][t]: q [M][d] f32 (pre-scaled, bias included), wkT [H][d][64] bf16
// (head h's rows of W_k, transposed), qe [M][H][d] f32.  q enters the bf16 MFMAs as hi + lo, so nothing of it is rounded.
// (WH_PREC_F16X3: wkT as h2, E as fp16 limb planes [B][e_rows][hi d | lo d] written by wh_launch_layernorm_es2, out as an h2 slab)
void wh_launch_dec_qexpand(hipStream_t s, int prec, const float* q, const void* wkT, float* qe, int M, int d, int n_heads);
// e_rows >= S: rows between two clips' states; n_cus: compute units the launch stream may use (one persistent workgroup per CU)
void wh_launch_dec_cross_attn_es(hipStream_t s, int prec, const float* qe, const void* E, void* out, int S, int e_rows, int B, int mpad, bool stream_nt, int n_cus);
void wh_launch_layernorm_es2(hipStream_t s, const float* x, const float* w, const float* b, void* y, long rows, int in_blk, int out_blk);
// (WH_PREC_F16X3, wh_cross_es3.hip: E as key rows of [d fp16 | d e4m3 remainders] = 3 bytes per element, written by wh_launch_layernorm_es3; the default of the
// mode — wh_es3_enabled(), WH_ES3=0 keeps the two-fp16-limb form — and dispatched by wh_launch_dec_cross_attn_es)
bool wh_es3_enabled();
void wh_launch_dec_cross_attn_es3(hipStream_t s, const float* qe, const void* E, void* out, int S, int e_rows, int B, int mpad, bool stream_nt, int n_cus);
void wh_launch_layernorm_es3(hipStream_t s, const float* x, const float* w, const float* b, void* y, long rows, int in_blk, int out_blk);
// (WH_PREC_FP8, wh_cross_es8.hip: E as e4m3 rows [B][e_rows][d] written by wh_launch_layernorm_es8; wkT and out as in bf16 — wh_launch_dec_cross_attn_es dispatches on prec)
void wh_launch_dec_cross_attn_es8(hipStream_t s, const float* qe, const void* E, void* out, int S, int e_rows, int B, int mpad, bool stream_nt, int n_cus);
void wh_launch_layernorm_es8(hipStream_t s, const float* x, const float* gamma, const float* beta, void* out, long rows, int S, int e_rows);

// dynamic LDS to request for a cross-attention launch of total_wgs workgroups whose kernel needs own_bytes: caps the resident
// workgroups per CU at two for large launches (wh_decode.hip)
size_t wh_cross_lds_reserve(long total_wgs, size_t own_bytes);

// WH_PREC_FP8 (wh_fp8.hip)
void wh_launch_kv_quant(hipStream_t s, const void* kv_bf16, unsigned* amax, void* kv8, long planes_x_clips, int S, int d,
                        int n_heads);
void wh_launch_dec_cross_attn8(hipStream_t s, const void* q, const void* ck, const void* cv, const float* amax_k,
                               const float* amax_v, float* part, float* ml, int S, int d, int n_heads, int splits, int B, void* out,
                               int mpad, bool stream_nt);

extern int wh_dbg_cross_unroll;
extern int wh_dbg_lm_blocks_per_cu;
extern int wh_dbg_mt;
extern int wh_dbg_nw;
extern int wh_dbg_wide;
extern int wh_dbg_lm_mt;
