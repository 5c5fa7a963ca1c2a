/*
 * whisper_oracle.c — CPU restatement of the reference's hot path, fp32, plain C.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product: only
 * tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load this file's
 * shared object, and only as the checker / the timed CPU baseline.  The product path is
 * the HIP library behind include/whisper_hip.h and fails loudly without it.
 *
 * What it restates (reference = /root/reference, KrArunT/whisper-rust-ort):
 *   - orc_log_mel            src/main.rs:323-509  (hann_window, hz_to_mel_slaney,
 *                            mel_to_hz_slaney, build_mel_filterbank, whisper_log_mel_80)
 *   - orc_window_mel         src/main.rs:895-905  (caller-side zero-padded 3000-frame window)
 *   - orc_argmax_masked      src/main.rs:709-735  (argmax_last_dim_raw)
 *   - orc_decode_greedy      src/main.rs:753-829  (greedy_decode_with_past: suppress sets
 *                            765-768, step-0 decoder 771-783, with-past loop 793-826)
 *   - orc_encoder            src/main.rs:698-707  (run_encoder) — the arithmetic itself lives
 *                            in a third-party dependency that is NOT under /root/reference:
 *                            ONNX Runtime via `ort = "=2.0.0-rc.6"` (Cargo.toml:22) executing
 *                            graphs exported by optimum 2.1.0 / transformers 4.42
 *                            (Dockerfile.container:35-44, scripts/export_onnx_whisper.py:19-28).
 *                            The published algorithm restated here is the upstream Whisper
 *                            definition, transformers `modeling_whisper.py` (cited per function
 *                            as [3P] with the line numbers of the locally installed 5.15.0).
 *   - the 400-point FFT is rustfft 6.4.1 (Cargo.lock:991-992) in the reference; a DFT is a
 *     DFT, so it is restated as a direct float64 DFT rounded to f32.
 *
 * PINNING: the reference holds no golden vectors, tests or fixtures for this path (SURVEY.md §4,
 * §8c) and cannot be built here (Rust + prebuilt ORT download).  The oracle is therefore pinned
 * against vectors produced by the upstream definition itself: tests/golden/make_golden.py imports
 * the locally installed `transformers` Whisper classes (third-party, not the reference), loads
 * the same hash-seeded weights and writes tests/golden/*.npz; tests/test_oracle_golden.py holds
 * this file to those vectors.  Relative to the REFERENCE's own outputs parity stays "unpinned"
 * (see DESIGN.md).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stddef.h>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int n_mels, d_model, n_heads, enc_layers, dec_layers, ffn, vocab, n_audio_ctx, n_text_ctx;
} orc_dims;

#define ORC_OK 0
#define ORC_ERR_EMPTY_AUDIO 1   /* src/main.rs:414-416 */
#define ORC_ERR_BAD_SHAPE 2     /* src/main.rs:710-716 */
#define ORC_ERR_ARG 3
#define ORC_ERR_NOMEM 4

/* ------------------------------------------------------------------------------------------
 * log-mel front end
 * ---------------------------------------------------------------------------------------- */
#define N_FFT 400
#define HOP 160
#define N_FREQ 201

/* src/main.rs:323-330 — periodic Hann, evaluated in f32 */
static void hann_window(float* w, int n) {
    for (int i = 0; i < n; i++) {
        float x = (3.14159265358979323846f * 2.0f * (float)i) / (float)n;
        w[i] = 0.5f - 0.5f * cosf(x);
    }
}

/* src/main.rs:332-341 */
static float hz_to_mel_slaney(float hz) {
    const float min_log_hz = 1000.0f, min_log_mel = 15.0f;
    const float logstep = 27.0f / logf(6.4f);
    float mel = 3.0f * hz / 200.0f;
    if (hz >= min_log_hz) mel = min_log_mel + logf(hz / min_log_hz) * logstep;
    return mel;
}

/* src/main.rs:343-352 */
static float mel_to_hz_slaney(float mel) {
    const float min_log_hz = 1000.0f, min_log_mel = 15.0f;
    const float logstep = logf(6.4f) / 27.0f;
    float hz = 200.0f * mel / 3.0f;
    if (mel >= min_log_mel) hz = min_log_hz * expf(logstep * (mel - min_log_mel));
    return hz;
}

/* src/main.rs:354-405 — Slaney-normalised triangular filterbank, all in f32.
 * fb is [n_mels][201] row-major. sr=16000, fmin=0, fmax=8000 as at the call site (:438). */
void orc_mel_filterbank(int n_mels, float* fb) {
    const float sr_half = 8000.0f;
    float fmax = 8000.0f;
    if (fmax > sr_half) fmax = sr_half;
    float mel_min = hz_to_mel_slaney(0.0f), mel_max = hz_to_mel_slaney(fmax);
    float* fp = (float*)malloc(sizeof(float) * (size_t)(n_mels + 2));
    for (int i = 0; i < n_mels + 2; i++) {
        float m = mel_min + (mel_max - mel_min) * (float)i / (float)(n_mels + 1);
        fp[i] = mel_to_hz_slaney(m);
    }
    for (int m = 0; m < n_mels; m++) {
        float fl = fp[m], fc = fp[m + 1], fr = fp[m + 2];
        float dl = fmaxf(fc - fl, 1e-6f), dr = fmaxf(fr - fc, 1e-6f);
        float enorm = 2.0f / fmaxf(fr - fl, 1e-6f);
        for (int k = 0; k < N_FREQ; k++) {
            float f = (float)k * sr_half / (float)(N_FREQ - 1);
            float lower = (f - fl) / dl, upper = (fr - f) / dr;
            float w = fmaxf(fminf(lower, upper), 0.0f);
            fb[(size_t)m * N_FREQ + k] = w * enorm;
        }
    }
    free(fp);
}

/* src/main.rs:444-452 */
size_t orc_mel_frames(size_t n) {
    size_t padded = n + N_FFT;
    size_t nf = padded < N_FFT ? 1 : 1 + (padded - N_FFT) / HOP;
    if (nf > 1) nf -= 1;
    return nf;
}

/* src/main.rs:407-509.  out is [n_mels][n_frames] row-major (mel-major), n_frames =
 * orc_mel_frames(n).  Normalisation uses the GLOBAL max over the whole array (:494-506). */
int orc_log_mel(const float* pcm, size_t n, int n_mels, float* out) {
    if (n == 0) return ORC_ERR_EMPTY_AUDIO;
    const size_t pad = N_FFT / 2;
    const size_t plen = n + 2 * pad;
    float* padded = (float*)calloc(plen, sizeof(float));
    if (!padded) return ORC_ERR_NOMEM;
    if (n >= 2) { /* :421-431 */
        for (size_t i = 0; i < pad; i++) {
            size_t idx = pad - i;
            size_t src = idx < n - 1 ? idx : n - 1;
            padded[i] = pcm[src];
        }
        memcpy(padded + pad, pcm, n * sizeof(float));
        for (size_t i = 0; i < pad; i++) {
            size_t idx = (n >= 2 + i) ? n - (2 + i) : 0; /* saturating_sub */
            padded[pad + n + i] = pcm[idx];
        }
    } else { /* :432-435 */
        memcpy(padded, pcm, n * sizeof(float)); /* extend_from_slice THEN resize with zeros */
    }
    float window[N_FFT];
    hann_window(window, N_FFT);
    float* fb = (float*)malloc(sizeof(float) * (size_t)n_mels * N_FREQ);
    orc_mel_filterbank(n_mels, fb);
    /* DFT twiddles in float64 */
    double* cs = (double*)malloc(sizeof(double) * N_FFT * 2);
    for (int j = 0; j < N_FFT; j++) {
        cs[2 * j] = cos(2.0 * M_PI * (double)j / N_FFT);
        cs[2 * j + 1] = -sin(2.0 * M_PI * (double)j / N_FFT);
    }
    const size_t n_frames = orc_mel_frames(n);
#pragma omp parallel for schedule(static)
    for (long frame = 0; frame < (long)n_frames; frame++) {
        size_t start = (size_t)frame * HOP;
        float x[N_FFT];
        for (int i = 0; i < N_FFT; i++) { /* :463-470 */
            size_t idx = start + (size_t)i;
            float s = idx < plen ? padded[idx] : 0.0f;
            x[i] = s * window[i];
        }
        float pows[N_FREQ];
        for (int k = 0; k < N_FREQ; k++) { /* :472-481 */
            double re = 0.0, im = 0.0;
            int j = 0;
            for (int i = 0; i < N_FFT; i++) {
                re += (double)x[i] * cs[2 * j];
                im += (double)x[i] * cs[2 * j + 1];
                j += k;
                if (j >= N_FFT) j -= N_FFT;
            }
            float fre = (float)re, fim = (float)im;
            pows[k] = fre * fre + fim * fim;
        }
        for (int m = 0; m < n_mels; m++) { /* :484-490 */
            float e = 0.0f;
            const float* row = fb + (size_t)m * N_FREQ;
            for (int k = 0; k < N_FREQ; k++) e += row[k] * pows[k];
            out[(size_t)m * n_frames + (size_t)frame] = fmaxf(e, 1e-10f);
        }
    }
    float max_log = -INFINITY; /* :494-500 */
    const size_t tot = (size_t)n_mels * n_frames;
    for (size_t i = 0; i < tot; i++) {
        float lv = log10f(out[i]);
        if (lv > max_log) max_log = lv;
    }
    for (size_t i = 0; i < tot; i++) { /* :502-506 */
        float lv = log10f(out[i]);
        float c = fmaxf(lv, max_log - 8.0f);
        out[i] = (c + 4.0f) / 4.0f;
    }
    free(cs);
    free(fb);
    free(padded);
    return ORC_OK;
}

/* src/main.rs:895-905 / 950-961: zero-filled [n_mels,3000] window starting at frame_start. */
void orc_window_mel(const float* mel_full, size_t total_frames, int n_mels, size_t frame_start,
                    size_t win_frames, float* out) {
    memset(out, 0, sizeof(float) * (size_t)n_mels * win_frames);
    if (frame_start >= total_frames) return;
    size_t end = frame_start + win_frames;
    if (end > total_frames) end = total_frames;
    size_t frames = end - frame_start;
    for (int m = 0; m < n_mels; m++)
        memcpy(out + (size_t)m * win_frames, mel_full + (size_t)m * total_frames + frame_start,
               frames * sizeof(float));
}

/* ------------------------------------------------------------------------------------------
 * weight blob navigation — canonical order of whisper-rust-ort_amd/modelspec.py::tensor_table
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const float *qw, *qb, *kw, *vw, *vb, *ow, *ob;
} attn_w;
typedef struct {
    const float *w, *b;
} ln_w;
typedef struct {
    const float *w1, *b1, *w2, *b2;
} mlp_w;

static const float* take(const float** p, size_t n) {
    const float* r = *p;
    *p += n;
    return r;
}
static void take_attn(const float** p, size_t d, attn_w* a) {
    a->qw = take(p, d * d); a->qb = take(p, d);
    a->kw = take(p, d * d);
    a->vw = take(p, d * d); a->vb = take(p, d);
    a->ow = take(p, d * d); a->ob = take(p, d);
}
static void take_ln(const float** p, size_t d, ln_w* l) { l->w = take(p, d); l->b = take(p, d); }
static void take_mlp(const float** p, size_t d, size_t F, mlp_w* m) {
    m->w1 = take(p, F * d); m->b1 = take(p, F);
    m->w2 = take(p, d * F); m->b2 = take(p, d);
}

typedef struct { attn_w sa; ln_w sa_ln; mlp_w mlp; ln_w fin_ln; } enc_layer_w;
typedef struct { attn_w sa; ln_w sa_ln; attn_w ca; ln_w ca_ln; mlp_w mlp; ln_w fin_ln; } dec_layer_w;
typedef struct {
    const float *conv1w, *conv1b, *conv2w, *conv2b, *enc_pos;
    enc_layer_w* enc;
    ln_w enc_ln;
    const float *tok_emb, *dec_pos;
    dec_layer_w* dec;
    ln_w dec_ln;
} model_w;

static int map_weights(const orc_dims* c, const float* w, model_w* m) {
    const size_t d = (size_t)c->d_model, F = (size_t)c->ffn;
    const float* p = w;
    m->conv1w = take(&p, d * (size_t)c->n_mels * 3); m->conv1b = take(&p, d);
    m->conv2w = take(&p, d * d * 3); m->conv2b = take(&p, d);
    m->enc_pos = take(&p, (size_t)c->n_audio_ctx * d);
    m->enc = (enc_layer_w*)malloc(sizeof(enc_layer_w) * (size_t)c->enc_layers);
    m->dec = (dec_layer_w*)malloc(sizeof(dec_layer_w) * (size_t)c->dec_layers);
    if (!m->enc || !m->dec) return ORC_ERR_NOMEM;
    for (int i = 0; i < c->enc_layers; i++) {
        take_attn(&p, d, &m->enc[i].sa); take_ln(&p, d, &m->enc[i].sa_ln);
        take_mlp(&p, d, F, &m->enc[i].mlp); take_ln(&p, d, &m->enc[i].fin_ln);
    }
    take_ln(&p, d, &m->enc_ln);
    m->tok_emb = take(&p, (size_t)c->vocab * d);
    m->dec_pos = take(&p, (size_t)c->n_text_ctx * d);
    for (int i = 0; i < c->dec_layers; i++) {
        take_attn(&p, d, &m->dec[i].sa); take_ln(&p, d, &m->dec[i].sa_ln);
        take_attn(&p, d, &m->dec[i].ca); take_ln(&p, d, &m->dec[i].ca_ln);
        take_mlp(&p, d, F, &m->dec[i].mlp); take_ln(&p, d, &m->dec[i].fin_ln);
    }
    take_ln(&p, d, &m->dec_ln);
    return ORC_OK;
}
static void unmap_weights(model_w* m) { free(m->enc); free(m->dec); }

size_t orc_n_params(const orc_dims* c) {
    const size_t d = (size_t)c->d_model, F = (size_t)c->ffn;
    const size_t attn = 4 * d * d + 3 * d, ln = 2 * d, mlp = 2 * d * F + F + d;
    size_t n = d * (size_t)c->n_mels * 3 + d + d * d * 3 + d + (size_t)c->n_audio_ctx * d;
    n += (size_t)c->enc_layers * (attn + ln + mlp + ln) + ln;
    n += (size_t)c->vocab * d + (size_t)c->n_text_ctx * d;
    n += (size_t)c->dec_layers * (2 * attn + 3 * ln + mlp) + ln;
    return n;
}

/* ------------------------------------------------------------------------------------------
 * dense helpers (deterministic summation order; no -ffast-math)
 * ---------------------------------------------------------------------------------------- */
/* C[M,N] = A[M,K](lda) * W[N,K]^T + bias[N].  W is transposed once so the inner loop runs
 * over N with unit stride and every output is a plain k-ordered sum. */
static void gemm_nt(const float* A, size_t lda, const float* W, const float* bias, float* C,
                    size_t ldc, size_t M, size_t N, size_t K) {
    float* Wt = (float*)malloc(sizeof(float) * K * N);
#pragma omp parallel for schedule(static)
    for (long k = 0; k < (long)K; k++)
        for (size_t n = 0; n < N; n++) Wt[(size_t)k * N + n] = W[n * K + (size_t)k];
    enum { RB = 4, NB = 64 };
#pragma omp parallel for schedule(static)
    for (long ib = 0; ib < (long)((M + RB - 1) / RB); ib++) {
        size_t i0 = (size_t)ib * RB, rows = M - i0 < RB ? M - i0 : RB;
        for (size_t j0 = 0; j0 < N; j0 += NB) {
            size_t cols = N - j0 < NB ? N - j0 : NB;
            float acc[RB][NB];
            for (size_t r = 0; r < RB; r++)
                for (size_t j = 0; j < NB; j++) acc[r][j] = (bias && j < cols) ? bias[j0 + j] : 0.0f;
            if (cols == NB && rows == RB) {
                for (size_t k = 0; k < K; k++) {
                    const float* wr = Wt + k * N + j0;
                    float a0 = A[(i0 + 0) * lda + k], a1 = A[(i0 + 1) * lda + k];
                    float a2 = A[(i0 + 2) * lda + k], a3 = A[(i0 + 3) * lda + k];
                    for (size_t j = 0; j < NB; j++) {
                        float w = wr[j];
                        acc[0][j] += a0 * w; acc[1][j] += a1 * w;
                        acc[2][j] += a2 * w; acc[3][j] += a3 * w;
                    }
                }
            } else {
                for (size_t k = 0; k < K; k++) {
                    const float* wr = Wt + k * N + j0;
                    for (size_t r = 0; r < rows; r++) {
                        float a = A[(i0 + r) * lda + k];
                        for (size_t j = 0; j < cols; j++) acc[r][j] += a * wr[j];
                    }
                }
            }
            for (size_t r = 0; r < rows; r++)
                for (size_t j = 0; j < cols; j++) C[(i0 + r) * ldc + j0 + j] = acc[r][j];
        }
    }
    free(Wt);
}

/* y[N] = W[N,K] x + bias : 16 interleaved partial sums per output, combined in fixed order */
static void gemv(const float* W, const float* bias, const float* x, float* y, size_t N, size_t K) {
#pragma omp parallel for schedule(static)
    for (long n = 0; n < (long)N; n++) {
        const float* w = W + (size_t)n * K;
        float acc[16];
        for (int l = 0; l < 16; l++) acc[l] = 0.0f;
        size_t k = 0;
        for (; k + 16 <= K; k += 16)
            for (int l = 0; l < 16; l++) acc[l] += w[k + l] * x[k + l];
        for (; k < K; k++) acc[k & 15] += w[k] * x[k];
        float s = 0.0f;
        for (int l = 0; l < 16; l++) s += acc[l];
        y[n] = s + (bias ? bias[n] : 0.0f);
    }
}

/* [3P] torch LayerNorm, eps 1e-5 (modeling_whisper.py:371): biased variance */
static void layer_norm(const float* x, const ln_w* l, float* y, size_t rows, size_t d) {
#pragma omp parallel for schedule(static)
    for (long r = 0; r < (long)rows; r++) {
        const float* xr = x + (size_t)r * d;
        float* yr = y + (size_t)r * d;
        double mean = 0.0, var = 0.0;
        for (size_t i = 0; i < d; i++) mean += xr[i];
        mean /= (double)d;
        for (size_t i = 0; i < d; i++) { double t = xr[i] - mean; var += t * t; }
        var /= (double)d;
        float rstd = (float)(1.0 / sqrt(var + 1e-5));
        float fm = (float)mean;
        for (size_t i = 0; i < d; i++) yr[i] = (xr[i] - fm) * rstd * l->w[i] + l->b[i];
    }
}

/* exact (erf) GELU — activation_function "gelu" (configuration_whisper.py:140) */
static inline float gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

static void softmax_row(float* s, size_t n) {
    float mx = -INFINITY;
    for (size_t i = 0; i < n; i++) if (s[i] > mx) mx = s[i];
    float sum = 0.0f;
    for (size_t i = 0; i < n; i++) { s[i] = expf(s[i] - mx); sum += s[i]; }
    float inv = 1.0f / sum;
    for (size_t i = 0; i < n; i++) s[i] *= inv;
}

/* ------------------------------------------------------------------------------------------
 * encoder  — [3P] modeling_whisper.py WhisperEncoder.forward (:566-567, 618-645),
 *            WhisperEncoderLayer.forward (:385-399), WhisperAttention.forward (:279-357)
 * mel: [n_mels][2*n_audio_ctx] row-major;  out: [n_audio_ctx][d_model]
 * ---------------------------------------------------------------------------------------- */
static int g_act_mx;                                         /* fp8 section below */
static void fake_quant_mx(float* x, size_t rows, size_t K);

int orc_encoder(const orc_dims* c, const float* w, const float* mel, float* out) {
    model_w m;
    int rc = map_weights(c, w, &m);
    if (rc) return rc;
    const size_t d = (size_t)c->d_model, T = (size_t)c->n_audio_ctx, L = 2 * T, C = (size_t)c->n_mels;
    const size_t H = (size_t)c->n_heads, hd = d / H, F = (size_t)c->ffn;
    /* conv1 (k3,p1) + GELU as a GEMM over an im2col matrix [L][3C] with weight [d][C*3]
     * re-ordered to the same (c,k) flattening torch uses: W[o][c][k] → column c*3+k */
    float* col = (float*)calloc(L * 3 * C, sizeof(float));
    float* h1 = (float*)malloc(sizeof(float) * L * d);
    float* col2 = (float*)calloc(T * 3 * d, sizeof(float));
    float* x = (float*)malloc(sizeof(float) * T * d);
    float* xn = (float*)malloc(sizeof(float) * T * d);
    float* q = (float*)malloc(sizeof(float) * T * d);
    float* k = (float*)malloc(sizeof(float) * T * d);
    float* v = (float*)malloc(sizeof(float) * T * d);
    float* ao = (float*)malloc(sizeof(float) * T * d);
    float* hb = (float*)malloc(sizeof(float) * T * F);
    if (!col || !h1 || !col2 || !x || !xn || !q || !k || !v || !ao || !hb) return ORC_ERR_NOMEM;
    for (size_t t = 0; t < L; t++)
        for (size_t ci = 0; ci < C; ci++)
            for (int kk = 0; kk < 3; kk++) {
                long src = (long)t + kk - 1;
                col[t * 3 * C + ci * 3 + (size_t)kk] = (src >= 0 && src < (long)L) ? mel[ci * L + (size_t)src] : 0.0f;
            }
    gemm_nt(col, 3 * C, m.conv1w, m.conv1b, h1, d, L, d, 3 * C);
    for (size_t i = 0; i < L * d; i++) h1[i] = gelu(h1[i]);
    /* conv2 (k3,s2,p1) + GELU */
    for (size_t t = 0; t < T; t++)
        for (size_t ci = 0; ci < d; ci++)
            for (int kk = 0; kk < 3; kk++) {
                long src = 2 * (long)t + kk - 1;
                col2[t * 3 * d + ci * 3 + (size_t)kk] = (src >= 0 && src < (long)L) ? h1[(size_t)src * d + ci] : 0.0f;
            }
    gemm_nt(col2, 3 * d, m.conv2w, m.conv2b, x, d, T, d, 3 * d);
    for (size_t i = 0; i < T * d; i++) x[i] = gelu(x[i]) + m.enc_pos[i]; /* :619-624 */
    const float scaling = 1.0f / sqrtf((float)hd);
    for (int li = 0; li < c->enc_layers; li++) {
        const enc_layer_w* lw = &m.enc[li];
        layer_norm(x, &lw->sa_ln, xn, T, d);
        if (g_act_mx) fake_quant_mx(xn, T, d);
        gemm_nt(xn, d, lw->sa.qw, lw->sa.qb, q, d, T, d, d);
        for (size_t i = 0; i < T * d; i++) q[i] *= scaling; /* :309 scale q BEFORE QK^T */
        gemm_nt(xn, d, lw->sa.kw, NULL, k, d, T, d, d);
        gemm_nt(xn, d, lw->sa.vw, lw->sa.vb, v, d, T, d, d);
#pragma omp parallel
        {
            float* s = (float*)malloc(sizeof(float) * T);
#pragma omp for collapse(2) schedule(static)
            for (long h = 0; h < (long)H; h++)
                for (long i = 0; i < (long)T; i++) {
                    const float* qi = q + (size_t)i * d + (size_t)h * hd;
                    for (size_t j = 0; j < T; j++) {
                        const float* kj = k + j * d + (size_t)h * hd;
                        float acc = 0.0f;
                        for (size_t e = 0; e < hd; e++) acc += qi[e] * kj[e];
                        s[j] = acc;
                    }
                    softmax_row(s, T);
                    float* o = ao + (size_t)i * d + (size_t)h * hd;
                    for (size_t e = 0; e < hd; e++) o[e] = 0.0f;
                    for (size_t j = 0; j < T; j++) {
                        const float* vj = v + j * d + (size_t)h * hd;
                        float p = s[j];
                        for (size_t e = 0; e < hd; e++) o[e] += p * vj[e];
                    }
                }
            free(s);
        }
        gemm_nt(ao, d, lw->sa.ow, lw->sa.ob, xn, d, T, d, d);
        for (size_t i = 0; i < T * d; i++) x[i] += xn[i];
        layer_norm(x, &lw->fin_ln, xn, T, d);
        if (g_act_mx) fake_quant_mx(xn, T, d);
        gemm_nt(xn, d, lw->mlp.w1, lw->mlp.b1, hb, F, T, F, d);
        for (size_t i = 0; i < T * F; i++) hb[i] = gelu(hb[i]);
        if (g_act_mx) fake_quant_mx(hb, T, F);
        gemm_nt(hb, F, lw->mlp.w2, lw->mlp.b2, xn, d, T, d, F);
        for (size_t i = 0; i < T * d; i++) x[i] += xn[i];
    }
    layer_norm(x, &m.enc_ln, out, T, d); /* :642 */
    free(col); free(h1); free(col2); free(x); free(xn); free(q); free(k); free(v); free(ao); free(hb);
    unmap_weights(&m);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * masked argmax — src/main.rs:709-735.  Last row of a row-major [..., V] tensor; suppressed ids
 * are skipped; strict `>` from -inf, so the lowest index wins ties and NaN never wins; if every
 * candidate is skipped/NaN/-inf the answer is 0.
 * ---------------------------------------------------------------------------------------- */
static int in_set(int64_t v, const int64_t* s, size_t n) {
    for (size_t i = 0; i < n; i++) if (s[i] == v) return 1;
    return 0;
}
int orc_argmax_last_row(const int64_t* shape, size_t ndim, const float* data, size_t len,
                        const int64_t* suppress, size_t ns, int64_t* out) {
    if (ndim < 2) return ORC_ERR_BAD_SHAPE;
    size_t V = (size_t)shape[ndim - 1];
    if (V == 0 || len < V) return ORC_ERR_BAD_SHAPE;
    size_t rows = len / V;
    const float* row = data + (rows - 1) * V;
    size_t best_i = 0;
    float best_v = -INFINITY;
    for (size_t i = 0; i < V; i++) {
        if (ns && in_set((int64_t)i, suppress, ns)) continue;
        if (row[i] > best_v) { best_v = row[i]; best_i = i; }
    }
    *out = (int64_t)best_i;
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * greedy decode with KV past — src/main.rs:753-829.
 *   [3P] WhisperDecoder.forward (:737-795), WhisperDecoderLayer.forward (:466-500), learned
 *   positions offset by past length (:208-212), final LN (:790), tied LM head, no bias (:965,970).
 * Step 0 feeds the P prompt ids (decoder_model.onnx, :771-783); it is evaluated token by token
 * through the same cached single-position routine, which is the causal computation.
 * `forced` (optional, parity harness only): generated token i is replaced by forced[i] AFTER the
 * argmax is recorded in tokens_out — teacher forcing for logit comparisons.
 * logits_out (optional): [n_generated][V], row i = logits that produced generated token i.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const orc_dims* c;
    model_w m;
    float *selfk, *selfv;   /* [Ld][n_text_ctx][d] */
    float *crossk, *crossv; /* [Ld][T][d] */
    float *x, *xn, *q, *kv, *ao, *hb, *sc;
} dec_state;

static void dec_position(dec_state* st, int64_t token, size_t pos, float* logits /* may be NULL */) {
    const orc_dims* c = st->c;
    const size_t d = (size_t)c->d_model, T = (size_t)c->n_audio_ctx, H = (size_t)c->n_heads, hd = d / H;
    const size_t F = (size_t)c->ffn, TC = (size_t)c->n_text_ctx;
    const float scaling = 1.0f / sqrtf((float)hd);
    float* x = st->x;
    for (size_t i = 0; i < d; i++) x[i] = st->m.tok_emb[(size_t)token * d + i] + st->m.dec_pos[pos * d + i];
    for (int li = 0; li < c->dec_layers; li++) {
        const dec_layer_w* lw = &st->m.dec[li];
        float* sk = st->selfk + (size_t)li * TC * d;
        float* sv = st->selfv + (size_t)li * TC * d;
        /* self attention over positions 0..pos */
        layer_norm(x, &lw->sa_ln, st->xn, 1, d);
        gemv(lw->sa.qw, lw->sa.qb, st->xn, st->q, d, d);
        for (size_t i = 0; i < d; i++) st->q[i] *= scaling;
        gemv(lw->sa.kw, NULL, st->xn, sk + pos * d, d, d);
        gemv(lw->sa.vw, lw->sa.vb, st->xn, sv + pos * d, d, d);
        for (size_t h = 0; h < H; h++) {
            float* s = st->sc;
            for (size_t j = 0; j <= pos; j++) {
                float acc = 0.0f;
                for (size_t e = 0; e < hd; e++) acc += st->q[h * hd + e] * sk[j * d + h * hd + e];
                s[j] = acc;
            }
            softmax_row(s, pos + 1);
            float* o = st->ao + h * hd;
            for (size_t e = 0; e < hd; e++) o[e] = 0.0f;
            for (size_t j = 0; j <= pos; j++)
                for (size_t e = 0; e < hd; e++) o[e] += s[j] * sv[j * d + h * hd + e];
        }
        gemv(lw->sa.ow, lw->sa.ob, st->ao, st->xn, d, d);
        for (size_t i = 0; i < d; i++) x[i] += st->xn[i];
        /* cross attention over the T encoder positions */
        const float* ck = st->crossk + (size_t)li * T * d;
        const float* cv = st->crossv + (size_t)li * T * d;
        layer_norm(x, &lw->ca_ln, st->xn, 1, d);
        gemv(lw->ca.qw, lw->ca.qb, st->xn, st->q, d, d);
        for (size_t i = 0; i < d; i++) st->q[i] *= scaling;
#pragma omp parallel for schedule(static)
        for (long h = 0; h < (long)H; h++) {
            float* s = st->sc + (size_t)h * T;
            for (size_t j = 0; j < T; j++) {
                float acc = 0.0f;
                for (size_t e = 0; e < hd; e++) acc += st->q[(size_t)h * hd + e] * ck[j * d + (size_t)h * hd + e];
                s[j] = acc;
            }
            softmax_row(s, T);
            float* o = st->ao + (size_t)h * hd;
            for (size_t e = 0; e < hd; e++) o[e] = 0.0f;
            for (size_t j = 0; j < T; j++)
                for (size_t e = 0; e < hd; e++) o[e] += s[j] * cv[j * d + (size_t)h * hd + e];
        }
        gemv(lw->ca.ow, lw->ca.ob, st->ao, st->xn, d, d);
        for (size_t i = 0; i < d; i++) x[i] += st->xn[i];
        /* MLP */
        layer_norm(x, &lw->fin_ln, st->xn, 1, d);
        gemv(lw->mlp.w1, lw->mlp.b1, st->xn, st->hb, F, d);
        for (size_t i = 0; i < F; i++) st->hb[i] = gelu(st->hb[i]);
        gemv(lw->mlp.w2, lw->mlp.b2, st->hb, st->xn, d, F);
        for (size_t i = 0; i < d; i++) x[i] += st->xn[i];
    }
    if (logits) {
        layer_norm(x, &st->m.dec_ln, st->xn, 1, d);
        gemv(st->m.tok_emb, NULL, st->xn, logits, (size_t)c->vocab, d);
    }
}

/* ------------------------------------------------------------------------------------------
 * fp8 (OCP e4m3fn) restatement for the build's fp8 mode.  The reference's analogue is
 * weights-only QInt8 on MatMul/Gemm (quantize_onnx_int8.py:37-42); e4m3 itself is the published OCP 8-bit
 * format: 1-4-3, bias 7, max 448, no infinities, S.1111.111 = NaN.  Round to nearest even, saturating.
 * Twins: modelspec.quantize_e4m3 (numpy), wh_quantize_e4m3 (csrc/wh_model.cpp).
 * ---------------------------------------------------------------------------------------- */
static uint8_t e4m3_from_f32(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    const uint8_t sign = (uint8_t)((u >> 24) & 0x80);
    if (x != x) return 0x7F;
    float a = fabsf(x);
    if (a > 448.0f) a = 448.0f;
    if (a >= 0.015625f) { /* normal: RNE at mantissa bit 20 */
        memcpy(&u, &a, 4);
        u += 0x7FFFFu + ((u >> 20) & 1u);
        uint32_t code = (((u >> 23) - 120u) << 3) | ((u >> 20) & 7u);
        if (code > 0x7Eu) code = 0x7Eu;
        return (uint8_t)(code | sign);
    }
    return (uint8_t)((uint32_t)nearbyint((double)a * 512.0) | sign); /* multiples of 2^-9; 8 = first normal */
}
static float e4m3_to_f32(uint8_t c) {
    const int e = (c >> 3) & 15, m = c & 7;
    float mag = e == 0 ? (float)m * 0.001953125f : ldexpf((float)(8 + m), e - 10);
    if ((c & 0x7F) == 0x7F) mag = NAN;
    return (c & 0x80) ? -mag : mag;
}
void orc_e4m3_quantize(const float* x, size_t n, uint8_t* out) { for (size_t i = 0; i < n; i++) out[i] = e4m3_from_f32(x[i]); }
void orc_e4m3_dequantize(const uint8_t* c, size_t n, float* out) { for (size_t i = 0; i < n; i++) out[i] = e4m3_to_f32(c[i]); }

/* fp8 cross-attention K/V cache: one scale per (layer, K|V, head) = max|.| / 448 over the clip's 1500 x 64
 * block, values stored as e4m3 codes of value / scale — restated as quantise-dequantise in place. */
static int g_kv_fp8 = 0;
void orc_set_kv_fp8(int on) { g_kv_fp8 = on; }

/* MX activations of the build's fp8-MFMA encoder (whisper-rust-ort_amd/csrc/wh_gemm8_mx.hip): every 32 consecutive
 * elements of a row share a power-of-two scale 2^(eb-127), eb = max(0, E - 8 + (mantissa > 1.75)) from the block's largest
 * magnitude (E = its biased f32 exponent), so that the maximum maps into (224, 448]; elements are stored as e4m3 codes of
 * value * 2^(127-eb).  Restated as quantise-dequantise in place.  Applied where the HIP path quantises: the LayerNorm
 * outputs that feed the q/k/v and fc1 projections, the GELU output that feeds fc2, and the encoder output as the operand
 * of the cross-attention K/V projections.  Reference analogue: dynamic activation quantisation of MatMul/Gemm,
 * quantize_onnx_int8.py:37-42. */
static int g_act_mx = 0;
void orc_set_act_mx(int on) { g_act_mx = on; }
static void fake_quant_mx(float* x, size_t rows, size_t K) {
#pragma omp parallel for schedule(static)
    for (long r = 0; r < (long)rows; r++)
        for (size_t b0 = 0; b0 + 32 <= K; b0 += 32) {
            float* v = x + (size_t)r * K + b0;
            float am = 0.0f;
            for (int i = 0; i < 32; i++) am = fmaxf(am, fabsf(v[i]));
            uint32_t ab;
            memcpy(&ab, &am, 4);
            int eb = (int)((ab >> 23) & 0xFF) - 8 + (int)((ab & 0x7FFFFF) > 0x600000);
            if (eb < 0) eb = 0;
            const float inv = ldexpf(1.0f, 127 - eb), sc = ldexpf(1.0f, eb - 127);
            for (int i = 0; i < 32; i++) v[i] = e4m3_to_f32(e4m3_from_f32(v[i] * inv)) * sc;
        }
}
static void fake_quant_heads(float* kv /* [T][d] */, size_t T, size_t d, size_t n_heads) {
    const size_t hd = d / n_heads;
    for (size_t h = 0; h < n_heads; h++) {
        float amax = 0.0f;
        for (size_t t = 0; t < T; t++)
            for (size_t e = 0; e < hd; e++) amax = fmaxf(amax, fabsf(kv[t * d + h * hd + e]));
        const float scale = amax > 0.0f ? amax / 448.0f : 1.0f;
        for (size_t t = 0; t < T; t++)
            for (size_t e = 0; e < hd; e++) {
                float* v = &kv[t * d + h * hd + e];
                *v = e4m3_to_f32(e4m3_from_f32(*v / scale)) * scale;
            }
    }
}

int orc_decode_greedy(const orc_dims* c, const float* w, const float* enc /* [T][d] */,
                      const int64_t* prompt, size_t p, size_t max_new, int64_t eot,
                      const int64_t* suppress, size_t ns, const int64_t* begin_suppress, size_t nb,
                      const int64_t* forced, size_t n_forced,
                      int64_t* tokens_out /* cap p+max_new */, size_t* n_out,
                      float* logits_out /* optional [max_new][V] */) {
    if (p == 0 || max_new == 0 || p + max_new > (size_t)c->n_text_ctx) return ORC_ERR_ARG;
    dec_state st;
    memset(&st, 0, sizeof st);
    st.c = c;
    int rc = map_weights(c, w, &st.m);
    if (rc) return rc;
    const size_t d = (size_t)c->d_model, T = (size_t)c->n_audio_ctx, Ld = (size_t)c->dec_layers;
    const size_t V = (size_t)c->vocab, TC = (size_t)c->n_text_ctx, F = (size_t)c->ffn;
    st.selfk = (float*)calloc(Ld * TC * d, sizeof(float));
    st.selfv = (float*)calloc(Ld * TC * d, sizeof(float));
    st.crossk = (float*)malloc(sizeof(float) * Ld * T * d);
    st.crossv = (float*)malloc(sizeof(float) * Ld * T * d);
    st.x = (float*)malloc(sizeof(float) * d); st.xn = (float*)malloc(sizeof(float) * d);
    st.q = (float*)malloc(sizeof(float) * d); st.ao = (float*)malloc(sizeof(float) * d);
    st.hb = (float*)malloc(sizeof(float) * F);
    st.sc = (float*)malloc(sizeof(float) * (size_t)c->n_heads * (T > TC ? T : TC));
    float* logits = (float*)malloc(sizeof(float) * V);
    /* cross K/V once per clip: present.{i}.encoder.{key,value} of step 0 (src/main.rs:786-787) */
    float* enc_q = NULL;   /* MX form of the encoder states as the projections' operand */
    if (g_act_mx) {
        enc_q = (float*)malloc(sizeof(float) * T * d);
        memcpy(enc_q, enc, sizeof(float) * T * d);
        fake_quant_mx(enc_q, T, d);
        enc = enc_q;
    }
    for (size_t li = 0; li < Ld; li++) {
        gemm_nt(enc, d, st.m.dec[li].ca.kw, NULL, st.crossk + li * T * d, d, T, d, d);
        gemm_nt(enc, d, st.m.dec[li].ca.vw, st.m.dec[li].ca.vb, st.crossv + li * T * d, d, T, d, d);
        if (g_kv_fp8) {
            fake_quant_heads(st.crossk + li * T * d, T, d, (size_t)c->n_heads);
            fake_quant_heads(st.crossv + li * T * d, T, d, (size_t)c->n_heads);
        }
    }
    /* suppress sets, :765-768 */
    int64_t* sup_first = (int64_t*)malloc(sizeof(int64_t) * (ns + nb + 1));
    memcpy(sup_first, suppress, ns * sizeof(int64_t));
    memcpy(sup_first + ns, begin_suppress, nb * sizeof(int64_t));
    size_t n = 0;
    for (size_t i = 0; i < p; i++) tokens_out[n++] = prompt[i];
    /* step 0 */
    for (size_t i = 0; i + 1 < p; i++) dec_position(&st, prompt[i], i, NULL);
    dec_position(&st, prompt[p - 1], p - 1, logits);
    size_t gen = 0;
    int64_t shape[3] = {1, 1, (int64_t)V};
    int64_t next;
    orc_argmax_last_row(shape, 3, logits, V, sup_first, ns + nb, &next);
    if (logits_out) memcpy(logits_out, logits, V * sizeof(float));
    tokens_out[n++] = next;
    gen = 1;
    if (forced && gen <= n_forced) next = forced[gen - 1];
    if (next != eot || (forced && gen <= n_forced)) {
        /* with-past loop, :793-826 — runs max_new-1 more times at most */
        for (size_t it = 1; it < max_new; it++) {
            size_t pos = p + it - 1;
            dec_position(&st, next, pos, logits);
            orc_argmax_last_row(shape, 3, logits, V, suppress, ns, &next);
            if (logits_out) memcpy(logits_out + gen * V, logits, V * sizeof(float));
            tokens_out[n++] = next;
            gen++;
            int is_forced = forced && gen <= n_forced;
            if (is_forced) next = forced[gen - 1];
            else if (next == eot) break;
        }
    }
    *n_out = n;
    free(sup_first); free(logits); free(enc_q);
    free(st.selfk); free(st.selfv); free(st.crossk); free(st.crossv);
    free(st.x); free(st.xn); free(st.q); free(st.ao); free(st.hb); free(st.sc);
    unmap_weights(&st.m);
    return ORC_OK;
}

void orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
