// wh_decode.hip — per-token decoder kernels for gfx950: the body of the with-past loop of
// greedy_decode_with_past (reference src/main.rs:793-826) for a BATCH of independent clips.
//
// The reference runs decoder_with_past_model.onnx once per token per clip through ORT's IoBinding
// and re-binds 4*Ld past tensors every step (src/main.rs:798-812).  Here the KV past is an
// on-device cache with an append position, the token loop never leaves the device (masked argmax,
// EOT bookkeeping and the next input token are computed by kernels), and up to 64 clips advance
// together so each weight byte streamed from HBM serves all of them.
//
// Launch structure per decoder layer (8 kernels):
//   [LN1 + QKV]  [self-attn]  [out-proj + residual]  [LN2 + cross-Q]  [cross-attn + merge]
//   [cross out-proj + residual]  [LN3 + fc1 + GELU]  [fc2 + residual]
// plus per position [final LN + LM head + masked argmax partials] [argmax finish].  LayerNorm and
// the token/position embedding run in the prologue of the GEMM that consumes them (activations
// staged in LDS as MFMA operands, weight fragments already in flight), the key-range merge of the
// cross-attention runs in the last workgroup of each clip, and the position counter is advanced by
// the last workgroup of the last kernel of a step.
//
// Every kernel reads the current position from device memory (*pos), never from a kernel argument,
// so one captured hipGraph of a step can be replayed for every step.
//
// [3P] decoder definition: modeling_whisper.py WhisperDecoderLayer.forward (:466-500), learned
// positions offset by the past length (:208-212), final LN (:790), tied LM head, no bias (:965,970).
#include "wh_common.h"
#include "wh_kernels.h"

#include <algorithm>
#include <mutex>
#include <unordered_map>

namespace {

// One workgroup arrives; the last one of the grid bumps *pos (every workgroup read *pos at its
// start, so nobody can observe the new value within this launch) and re-arms the ticket.
__device__ __forceinline__ void advance_if_last(int* ticket, int* pos_p, int n_blocks) {
    if (!ticket) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = atomicAdd(ticket, 1);
        if (t == n_blocks - 1) {
            *ticket = 0;
            *pos_p += 1;
        }
    }
}

// Prologue of the decode GEMMs: rows [m0, m0 + rows_tile) of the residual stream (or of the
// token + position embedding) → LayerNorm → compute dtype → LDS tile Xs[rows_tile][ldx].
// 16 lanes share a row (lane sub = lane & 15 owns float4 columns sub, sub+16, ...), so one wave
// instruction covers 4 rows x 256 contiguous bytes; G row groups are loaded before any is reduced.
// [3P] torch LayerNorm eps 1e-5, biased variance (modeling_whisper.py:371).
template <typename T, int PRO, int NVR, int G>
__device__ __forceinline__ void ln_rows_to_lds(const SkinnyArgs& a, int pos, int m0, int rows_tile, int wave, int lane,
                                               T* Xs, int ldx) {
    const int sub = lane & 15, rg = lane >> 4, d = a.K;
    f32x4 lw[NVR], lb[NVR];  // this lane's columns of gamma / beta: loaded once, ahead of everything
#pragma unroll
    for (int j = 0; j < NVR; j++) {
        const int c = (sub + 16 * j) * 4;
        lw[j] = lb[j] = f32x4{0, 0, 0, 0};
        if (c < d) {
            lw[j] = *reinterpret_cast<const f32x4*>(a.ln_w + c);
            lb[j] = *reinterpret_cast<const f32x4*>(a.ln_b + c);
        }
    }
    for (int p0 = 0; p0 < 4; p0 += G) {
        f32x4 v[G][NVR];
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int ml = wave * 16 + (p0 + g) * 4 + rg, m = m0 + ml;
            const bool valid = ml < rows_tile && m < a.M;
            const float* xr = a.xres + (long)m * d;
            const T* er = nullptr;
            const float* pr = nullptr;
            if (PRO == 2 && valid) {
                const long tok = a.feed[m * a.feed_ld + pos];
                er = (const T*)a.tok_emb + tok * d;
                pr = a.pos_emb + (long)pos * d;
            }
#pragma unroll
            for (int j = 0; j < NVR; j++) {
                const int c = (sub + 16 * j) * 4;
                v[g][j] = f32x4{0, 0, 0, 0};
                if (valid && c < d) {
                    if (PRO == 1) {
                        v[g][j] = *reinterpret_cast<const f32x4*>(xr + c);
                    } else {
                        f32x4 pp = *reinterpret_cast<const f32x4*>(pr + c);
                        v[g][j] = f32x4{cvt_in<T>(er[c]) + pp[0], cvt_in<T>(er[c + 1]) + pp[1], cvt_in<T>(er[c + 2]) + pp[2],
                                        cvt_in<T>(er[c + 3]) + pp[3]};
                        if (blockIdx.x == 0) *reinterpret_cast<f32x4*>(a.xres_out + (long)m * d + c) = v[g][j];
                    }
                }
            }
        }
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int ml = wave * 16 + (p0 + g) * 4 + rg;
            float sum = 0.0f;
#pragma unroll
            for (int j = 0; j < NVR; j++) sum += (v[g][j][0] + v[g][j][1]) + (v[g][j][2] + v[g][j][3]);
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) sum += __shfl_xor(sum, off);
            const float mean = sum / (float)d;
            float q = 0.0f;
#pragma unroll
            for (int j = 0; j < NVR; j++) {
                if ((sub + 16 * j) * 4 < d) {
#pragma unroll
                    for (int e = 0; e < 4; e++) { float t = v[g][j][e] - mean; q += t * t; }
                }
            }
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) q += __shfl_xor(q, off);
            const float rstd = rsqrtf(q / (float)d + 1e-5f);
            const bool live = (m0 + ml) < a.M;
            if (ml < rows_tile) {
#pragma unroll
                for (int j = 0; j < NVR; j++) {
                    const int c = (sub + 16 * j) * 4;
                    if (c < d) {
                        const f32x4 ww = lw[j], bb = lb[j];
                        T* dst = Xs + (long)ml * ldx + c;
                        if (live)
                            store4(dst, (v[g][j][0] - mean) * rstd * ww[0] + bb[0], (v[g][j][1] - mean) * rstd * ww[1] + bb[1],
                                   (v[g][j][2] - mean) * rstd * ww[2] + bb[2], (v[g][j][3] - mean) * rstd * ww[3] + bb[3]);
                        else
                            store4(dst, 0.0f, 0.0f, 0.0f, 0.0f);
                    }
                }
            }
        }
    }
}

// ---- decode GEMM: C[m][n] = act(sum_k X[m][k] W[n][k] + bias[n]) (+ R[m][n]),  M <= 64 ---------
// Weight-streaming: every weight element is read once per launch straight into MFMA fragments
// (up to 8 fragments per wave in flight before the first MFMA); activations are MFMA column
// operands.
//   PRO 0: X[m][:] read from global (compute dtype).
//   PRO 1: X = LayerNorm(x f32) computed in the prologue into LDS.
//   PRO 2: x = tok_emb[feed[m][pos]] + pos_emb[pos] (workgroup 0 also writes x), then as PRO 1.
//   SPLITK=4: one 16-column tile per workgroup, the four waves split K and reduce through LDS.
//   SPLITK=1: four 16-column tiles per workgroup, one per wave (LM head, N = vocab).
//   MODE 0  : normal epilogue.   MODE 1: LM-head epilogue — optional logits store + per-tile
//             masked argmax partials (reference argmax_last_dim_raw, src/main.rs:709-735).
template <typename T, typename TO, int MT, int SPLITK, int PRO, int MODE>
__global__ __launch_bounds__(256) void k_dec_gemm(SkinnyArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fl = lane & 15, fg = lane >> 4;
    const int tile = (SPLITK == 4) ? blockIdx.x : blockIdx.x * 4 + wave;
    const int n_tiles = (a.N + 15) >> 4;
    const bool tile_ok = tile < n_tiles;
    const int n0 = tile * 16;
    const T* W = (const T*)a.W;
    int nrow = n0 + fl;
    if (nrow > a.N - 1) nrow = a.N - 1;
    const int kspan = a.K / SPLITK;
    const int kb = (SPLITK == 4) ? wave * kspan : 0;
    const T* wp = W + (long)nrow * a.K + kb + fg * 8;
    const int iters = kspan >> 5;

    // first chunk of weight fragments goes in flight before anything else
    constexpr int DEPTH = 8;
    typename FragT<T>::type wq[DEPTH];
#pragma unroll
    for (int i = 0; i < DEPTH; i++)
        if (i < iters) wq[i] = load_frag<T>(wp + i * 32);

    const int pos = a.pos_p ? *a.pos_p : 0;
    const int m0 = blockIdx.y * MT * 16;  // first row of this workgroup's row group
    // epilogue operands of the SPLITK=4 form (wave w finishes row-tile w): fetched now, used last
    f32x4 pre_bias = {0, 0, 0, 0}, pre_r = {0, 0, 0, 0};
    if (SPLITK == 4 && wave < MT) {
        const int n = n0 + 4 * fg, m = m0 + wave * 16 + fl;
        if (n < a.N && m < a.M) {
            if (a.bias) pre_bias = *reinterpret_cast<const f32x4*>(a.bias + n);
            if (a.R) pre_r = *reinterpret_cast<const f32x4*>(a.R + (long)m * a.ldr + n);
        }
    }
    const int ldx = (PRO == 0) ? (int)a.ldx : a.K + 8;
    const T* X;
    if constexpr (PRO == 0) {
        X = (const T*)a.X;
    } else {
        T* Xs = reinterpret_cast<T*>(smem_raw);
        if (a.K <= 512) ln_rows_to_lds<T, PRO, 8, 2>(a, pos, m0, MT * 16, wave, lane, Xs, ldx);
        else ln_rows_to_lds<T, PRO, 20, 1>(a, pos, m0, MT * 16, wave, lane, Xs, ldx);
        __syncthreads();
        X = Xs;
    }
    const T* xp[MT];
#pragma unroll
    for (int t = 0; t < MT; t++) {
        int m = t * 16 + fl;  // row within the LDS tile (PRO != 0) or global row (PRO == 0)
        if (PRO == 0) {
            m += m0;
            if (m > a.M - 1) m = a.M - 1;
        }
        xp[t] = X + (long)m * ldx + kb + fg * 8;
    }
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; t++) acc[t] = f32x4{0, 0, 0, 0};
    for (int c0 = 0; c0 < iters; c0 += DEPTH) {
        if (c0 > 0) {
#pragma unroll
            for (int i = 0; i < DEPTH; i++)
                if (c0 + i < iters) wq[i] = load_frag<T>(wp + (c0 + i) * 32);
        }
#pragma unroll
        for (int i = 0; i < DEPTH; i++) {
            if (c0 + i < iters) {
#pragma unroll
                for (int t = 0; t < MT; t++) {
                    typename FragT<T>::type xf = load_frag<T>(xp[t] + (c0 + i) * 32);
                    mma16(acc[t], wq[i], xf);  // D rows = n (4*fg + r), col = m (fl)
                }
            }
        }
    }
    if (SPLITK == 4) {
        // cross-wave K reduction; wave w then owns row-tile w of the epilogue
        f32x4* red = reinterpret_cast<f32x4*>(smem_raw + ((PRO == 0) ? 0 : (size_t)MT * 16 * ldx * sizeof(T)));
#pragma unroll
        for (int t = 0; t < MT; t++) red[(wave * MT + t) * 64 + lane] = acc[t];
        __syncthreads();
        if (wave < MT) {
            f32x4 s = red[(0 * MT + wave) * 64 + lane];
#pragma unroll
            for (int w = 1; w < 4; w++) {
                f32x4 o = red[(w * MT + wave) * 64 + lane];
                s[0] += o[0]; s[1] += o[1]; s[2] += o[2]; s[3] += o[3];
            }
            const int n = n0 + 4 * fg, m = m0 + wave * 16 + fl;
            if (n < a.N && m < a.M) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    v[e] = s[e] + pre_bias[e];
                    if (a.act == 1) v[e] = gelu_erf(v[e]);
                    v[e] += pre_r[e];
                }
                store4((TO*)a.C + (long)m * a.ldc + n, v[0], v[1], v[2], v[3]);
            }
        }
        advance_if_last(a.ticket, a.pos_w, gridDim.x * gridDim.y);
        return;
    }
    // SPLITK == 1
    const int n = n0 + 4 * fg;
    if (MODE == 0) {
        if (tile_ok && n < a.N) {
            f32x4 bias = {0, 0, 0, 0};
            if (a.bias) bias = *reinterpret_cast<const f32x4*>(a.bias + n);
#pragma unroll
            for (int t = 0; t < MT; t++) {
                const int m = m0 + t * 16 + fl;
                if (m >= a.M) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    v[e] = acc[t][e] + bias[e];
                    if (a.act == 1) v[e] = gelu_erf(v[e]);
                }
                if (a.R) {
                    f32x4 r = *reinterpret_cast<const f32x4*>(a.R + (long)m * a.ldr + n);
                    v[0] += r[0]; v[1] += r[1]; v[2] += r[2]; v[3] += r[3];
                }
                store4((TO*)a.C + (long)m * a.ldc + n, v[0], v[1], v[2], v[3]);
            }
        }
        advance_if_last(a.ticket, a.pos_w, gridDim.x * gridDim.y);
    } else if (tile_ok) {
        const int gen = pos - (a.n_prompt - 1);  // index of the token this row generates
        const unsigned* mask = (gen == 0) ? a.mask_first : a.mask_base;
#pragma unroll
        for (int t = 0; t < MT; t++) {
            const int m = m0 + t * 16 + fl;
            float bv = -INFINITY;
            int bi = 0x7fffffff;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int nn = n + e;
                const float v = acc[t][e];
                if (nn < a.N && m < a.M) {
                    if (a.logits && gen >= 0 && gen < a.logits_rows)
                        a.logits[((long)m * a.logits_rows + gen) * a.N + nn] = v;
                    const bool sup = (mask[nn >> 5] >> (nn & 31)) & 1u;
                    if (!sup && v > bv) { bv = v; bi = nn; }  // strict >, NaN never wins
                }
            }
#pragma unroll
            for (int off = 16; off < 64; off <<= 1) {
                float ov = __shfl_xor(bv, off);
                int oi = __shfl_xor(bi, off);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            if (fg == 0 && m < a.M) {
                a.part_val[(long)m * n_tiles + tile] = bv;
                a.part_idx[(long)m * n_tiles + tile] = bi;
            }
        }
    }
}

// ---- LM head: logits = LN(x) · E^T over the whole vocabulary ([3P] :790, :965-970) + masked argmax
// partials per 16-column tile (reference argmax_last_dim_raw, src/main.rs:709-735).  The [M][K]
// activation tile is staged in LDS once per workgroup; each wave then walks 16-row tiles of the
// embedding matrix with all of a tile's weight fragments in flight before its first MFMA.
template <typename T, int MT>
__global__ __launch_bounds__(256) void k_lm_head(SkinnyArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int EPC = 16 / (int)sizeof(T);
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fl = lane & 15, fg = lane >> 4;
    const int n_tiles = (a.N + 15) >> 4;
    const int m0 = blockIdx.y * MT * 16;
    const int ldx = a.K + 8;
    T* Xs = reinterpret_cast<T*>(smem_raw);
    const int cpr = a.K / EPC;
    for (int c = tid; c < MT * 16 * cpr; c += 256) {
        const int row = c / cpr, col = (c - row * cpr) * EPC;
        u32x4 val = {0, 0, 0, 0};
        if (m0 + row < a.M) val = *reinterpret_cast<const u32x4*>((const T*)a.X + (long)(m0 + row) * a.ldx + col);
        *reinterpret_cast<u32x4*>(Xs + (long)row * ldx + col) = val;
    }
    const int pos = *a.pos_p;
    const int gen = pos - (a.n_prompt - 1);  // index of the token this row generates
    const unsigned* mask = (gen == 0) ? a.mask_first : a.mask_base;
    __syncthreads();
    const int iters = a.K >> 5;
    constexpr int DEPTH = 8;
    for (int tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += gridDim.x * 4) {
        const int n0 = tile * 16;
        int nrow = n0 + fl;
        if (nrow > a.N - 1) nrow = a.N - 1;
        const T* wp = (const T*)a.W + (long)nrow * a.K + fg * 8;
        f32x4 acc[MT];
#pragma unroll
        for (int t = 0; t < MT; t++) acc[t] = f32x4{0, 0, 0, 0};
        for (int c0 = 0; c0 < iters; c0 += DEPTH) {
            typename FragT<T>::type wq[DEPTH];
#pragma unroll
            for (int i = 0; i < DEPTH; i++)
                if (c0 + i < iters) wq[i] = load_frag<T>(wp + (c0 + i) * 32);
#pragma unroll
            for (int i = 0; i < DEPTH; i++) {
                if (c0 + i < iters) {
#pragma unroll
                    for (int t = 0; t < MT; t++) {
                        typename FragT<T>::type xf = load_frag<T>(Xs + (long)(t * 16 + fl) * ldx + (c0 + i) * 32 + fg * 8);
                        mma16(acc[t], wq[i], xf);
                    }
                }
            }
        }
        const int n = n0 + 4 * fg;
#pragma unroll
        for (int t = 0; t < MT; t++) {
            const int m = m0 + t * 16 + fl;
            float bv = -INFINITY;
            int bi = 0x7fffffff;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int nn = n + e;
                const float v = acc[t][e];
                if (nn < a.N && m < a.M) {
                    if (a.logits && gen >= 0 && gen < a.logits_rows)
                        a.logits[((long)m * a.logits_rows + gen) * a.N + nn] = v;
                    const bool sup = (mask[nn >> 5] >> (nn & 31)) & 1u;
                    if (!sup && v > bv) { bv = v; bi = nn; }  // strict >, NaN never wins
                }
            }
#pragma unroll
            for (int off = 16; off < 64; off <<= 1) {
                float ov = __shfl_xor(bv, off);
                int oi = __shfl_xor(bi, off);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            if (fg == 0 && m < a.M) {
                a.part_val[(long)m * n_tiles + tile] = bv;
                a.part_idx[(long)m * n_tiles + tile] = bi;
            }
        }
    }
}

// Final reduce of the per-tile argmax partials + greedy bookkeeping for one clip per workgroup:
// records the generated token, EOT stop (src/main.rs:781-783, 820-822) and the next input token.
__global__ __launch_bounds__(256) void k_argmax_finish(const float* __restrict__ part_val,
                                                       const int* __restrict__ part_idx, int n_tiles, int* pos_p,
                                                       int* ticket, DecodeState st) {
    __shared__ float sv[256];
    __shared__ int si[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int pos = *pos_p;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i0 = tid; i0 < n_tiles; i0 += 256 * 8) {
        float v[8];
        int ix[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = i0 + u * 256;
            v[u] = -INFINITY;
            ix[u] = 0x7fffffff;
            if (i < n_tiles) { v[u] = part_val[(long)b * n_tiles + i]; ix[u] = part_idx[(long)b * n_tiles + i]; }
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (v[u] > bv || (v[u] == bv && ix[u] < bi)) { bv = v[u]; bi = ix[u]; }
    }
    sv[tid] = bv; si[tid] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            float ov = sv[tid + s]; int oi = si[tid + s];
            if (ov > sv[tid] || (ov == sv[tid] && oi < si[tid])) { sv[tid] = ov; si[tid] = oi; }
        }
        __syncthreads();
    }
    if (tid == 0) {
        const int gen = pos - (st.n_prompt - 1);
        // nothing beat -inf (all suppressed / NaN / -inf): the reference's best_i stays 0
        const int tok = (si[0] == 0x7fffffff) ? 0 : si[0];
        int next = tok;
        if (!st.done[b]) {
            st.out_tokens[b * st.tok_ld + st.n_prompt + gen] = tok;
            st.n_out[b] = st.n_prompt + gen + 1;
            const bool forced = gen < st.n_forced;
            if (forced) next = st.forced[gen];
            else if (tok == st.eot) st.done[b] = 1;
        }
        if (pos + 1 < st.tok_ld) st.feed[b * st.tok_ld + pos + 1] = next;
    }
    advance_if_last(ticket, pos_p, gridDim.x);
}

// ---- decoder self-attention, one position ([3P] :417-425, 468-475): one wave per (head, clip) ---
// qkv: [B][3d] (q pre-scaled | k | v) of the current position; caches [B][H][TC][64].
template <typename T>
__global__ __launch_bounds__(64) void k_dec_self_attn(const T* __restrict__ qkv, T* __restrict__ kc,
                                                      T* __restrict__ vc, T* __restrict__ out,
                                                      const int* __restrict__ pos_p, int d, int n_heads, int tc) {
    constexpr int HD = WH_HEAD_DIM;
    __shared__ float qs[HD];
    __shared__ float sc[512];
    const int h = blockIdx.x, b = blockIdx.y, lane = threadIdx.x, pos = *pos_p;
    const T* row = qkv + (long)b * 3 * d;
    T* kcb = kc + ((long)b * n_heads + h) * tc * HD;
    T* vcb = vc + ((long)b * n_heads + h) * tc * HD;
    const T kcur = row[d + h * HD + lane], vcur = row[2 * d + h * HD + lane];
    kcb[(long)pos * HD + lane] = kcur;  // append: present.{i}.decoder.{key,value}
    vcb[(long)pos * HD + lane] = vcur;
    qs[lane] = cvt_in<T>(row[h * HD + lane]);
    __syncthreads();
    // scores over the past (from the cache) — each lane owns keys lane, lane+64, ...
    float mx = -INFINITY;
    for (int j = lane; j < pos; j += 64) {
        const T* kr = kcb + (long)j * HD;
        float s = 0.0f;
#pragma unroll
        for (int e = 0; e < HD; e += 8) {
            typename FragT<T>::type kk = load_frag<T>(kr + e);
#pragma unroll
            for (int u = 0; u < 8; u++) s += qs[e + u] * (float)kk[u];
        }
        sc[j] = s;
        mx = fmaxf(mx, s);
    }
    // the current position straight from registers
    const float scur = wave_sum(qs[lane] * cvt_in<T>(kcur));
    mx = fmaxf(wave_max(mx), scur);
    __syncthreads();
    float sum = 0.0f;
    for (int j = lane; j < pos; j += 64) {
        float p = __expf(sc[j] - mx);
        sc[j] = p;
        sum += p;
    }
    const float pcur = __expf(scur - mx);
    sum = wave_sum(sum) + pcur;
    __syncthreads();
    float o0 = pcur * cvt_in<T>(vcur), o1 = 0.0f, o2 = 0.0f, o3 = 0.0f;
    int j = 0;
    for (; j + 4 <= pos; j += 4) {
        const float v0 = cvt_in<T>(vcb[(long)j * HD + lane]), v1 = cvt_in<T>(vcb[(long)(j + 1) * HD + lane]);
        const float v2 = cvt_in<T>(vcb[(long)(j + 2) * HD + lane]), v3 = cvt_in<T>(vcb[(long)(j + 3) * HD + lane]);
        o0 += sc[j] * v0; o1 += sc[j + 1] * v1; o2 += sc[j + 2] * v2; o3 += sc[j + 3] * v3;
    }
    for (; j < pos; j++) o0 += sc[j] * cvt_in<T>(vcb[(long)j * HD + lane]);
    out[(long)b * d + h * HD + lane] = cvt_out<T>(((o0 + o1) + (o2 + o3)) / sum);
}

// ---- decoder cross-attention, one position ([3P] :433-440, 478-491) -----------------------------
// The HBM-bound kernel of batched decode: per clip and layer it streams S*d K and S*d V elements
// (18.4 MB per clip per step for whisper-base in bf16) and nothing else of note.
// One workgroup = one clip x one contiguous key range, ALL heads: every key row of K (and V) is one
// contiguous d-element line read once with 16-byte lane accesses; a wave walks its keys UNROLL at a
// time with the K and V rows of all of them in flight (single pass, online softmax in registers).
// The last workgroup of a clip to finish merges the key-range partials (agent-scope release /
// acquire around an arrival ticket) and writes the attention output.
//   ck/cv: [B][S][d]  (head h at columns h*64..h*64+63),   q: [B][d] pre-scaled
//   part : [B][splits][d] unnormalised partial outputs, ml: [B][splits][H][2] (max, sum)
template <typename T, int NCH>  // NCH = ceil(d*sizeof(T)/16 / 64): 16-B chunks per lane per row
__global__ __launch_bounds__(256) void k_dec_cross_attn(const T* __restrict__ q, const T* __restrict__ ck,
                                                        const T* __restrict__ cv, float* __restrict__ part,
                                                        float* __restrict__ ml, T* __restrict__ out,
                                                        int* __restrict__ tickets, int S, int d, int n_heads,
                                                        int splits) {
    constexpr int EPC = 16 / (int)sizeof(T);   // elements per 16-B chunk: 8 (bf16) / 4 (f32)
    constexpr int LPH = WH_HEAD_DIM / EPC;     // lanes per head: 8 / 16
    constexpr int UNROLL = (NCH == 1) ? 4 : (NCH <= 3 ? 2 : 1);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef __attribute__((ext_vector_type(EPC))) T vec_t;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sp = blockIdx.x, b = blockIdx.y;
    const int per = (S + splits - 1) / splits;
    const int ks = sp * per, ke = min(S, ks + per);
    const int pw = (ke - ks + 3) >> 2;         // keys per wave, contiguous
    const int j0 = ks + wave * pw, j1 = min(ke, j0 + pw);
    const int chunks = d / EPC;                // 16-B chunks per row

    float qv[NCH][EPC], o[NCH][EPC], mrun[NCH], lrun[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const int ch = lane + 64 * c;
        mrun[c] = -INFINITY;
        lrun[c] = 0.0f;
#pragma unroll
        for (int u = 0; u < EPC; u++) {
            qv[c][u] = (ch < chunks) ? cvt_in<T>(q[(long)b * d + ch * EPC + u]) : 0.0f;
            o[c][u] = 0.0f;
        }
    }
    const T* kb = ck + (long)b * S * d;
    const T* vb = cv + (long)b * S * d;
    // Software pipeline, two register sets: the K and V rows of the next UNROLL keys are in flight
    // while the current UNROLL keys go through dot / online softmax / P·V.
    vec_t kA[UNROLL][NCH], vA[UNROLL][NCH], kB[UNROLL][NCH], vB[UNROLL][NCH];
    auto load_set = [&](vec_t (&kk)[UNROLL][NCH], vec_t (&vv)[UNROLL][NCH], int j) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const int jj = min(j + u, j1 - 1);  // tail: re-read the last key, masked in compute_set
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const int ch = lane + 64 * c;
                if (ch < chunks) {
                    kk[u][c] = *reinterpret_cast<const vec_t*>(kb + (long)jj * d + ch * EPC);
                    vv[u][c] = *reinterpret_cast<const vec_t*>(vb + (long)jj * d + ch * EPC);
                }
            }
        }
    };
    auto compute_set = [&](const vec_t (&kk)[UNROLL][NCH], const vec_t (&vv)[UNROLL][NCH], int j) {
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const int ch = lane + 64 * c;
            float s[UNROLL];
            float mx = mrun[c];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                float t = 0.0f;
                if (ch < chunks) {
#pragma unroll
                    for (int e = 0; e < EPC; e++) t += qv[c][e] * (float)kk[u][c][e];
                }
#pragma unroll
                for (int off = 1; off < LPH; off <<= 1) t += __shfl_xor(t, off);
                s[u] = (j + u < j1) ? t : -INFINITY;
                mx = fmaxf(mx, s[u]);
            }
            const float scale = __expf(mrun[c] - mx);  // first chunk: exp(-inf) = 0
            float ls = lrun[c] * scale;
#pragma unroll
            for (int e = 0; e < EPC; e++) o[c][e] *= scale;
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                const float p = __expf(s[u] - mx);
                ls += p;
                if (ch < chunks) {
#pragma unroll
                    for (int e = 0; e < EPC; e++) o[c][e] += p * (float)vv[u][c][e];
                }
            }
            mrun[c] = mx;
            lrun[c] = ls;
        }
    };
    if (j0 < j1) load_set(kA, vA, j0);
    for (int j = j0; j < j1; j += 2 * UNROLL) {
        const bool hasB = j + UNROLL < j1;
        if (hasB) load_set(kB, vB, j + UNROLL);
        compute_set(kA, vA, j);
        if (hasB) {
            if (j + 2 * UNROLL < j1) load_set(kA, vA, j + 2 * UNROLL);
            compute_set(kB, vB, j + UNROLL);
        }
    }
    // merge the four waves of this key range (LDS): wm/wl [4][H], wo [4][d]
    float* wm = smem;
    float* wl = wm + 4 * n_heads;
    float* wo = wl + 4 * n_heads;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const int ch = lane + 64 * c;
        if (ch < chunks) {
            if ((lane % LPH) == 0) {
                wm[wave * n_heads + ch / LPH] = mrun[c];
                wl[wave * n_heads + ch / LPH] = lrun[c];
            }
#pragma unroll
            for (int e = 0; e < EPC; e++) wo[wave * d + ch * EPC + e] = o[c][e];
        }
    }
    __syncthreads();
    // Publish this key range's partial WRITE-THROUGH (sc1 stores: no L2 write-back fence needed),
    // drain, arrive on the clip's ticket; the last range to arrive merges all ranges.
    float* pp = part + ((long)b * splits + sp) * d;
    float* mp = ml + ((long)b * splits + sp) * n_heads * 2;
    for (int n = tid; n < d; n += 256) {
        const int h = n / WH_HEAD_DIM;
        const float M = fmaxf(fmaxf(wm[h], wm[n_heads + h]), fmaxf(wm[2 * n_heads + h], wm[3 * n_heads + h]));
        float num = 0.0f, den = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const float mw = wm[w * n_heads + h];
            const float sc = (mw == -INFINITY) ? 0.0f : __expf(mw - M);  // a wave may own no keys
            num += sc * wo[w * d + n];
            den += sc * wl[w * n_heads + h];
        }
        __hip_atomic_store(pp + n, num, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((n % WH_HEAD_DIM) == 0) {
            __hip_atomic_store(mp + h, M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(mp + n_heads + h, den, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its stores
    __syncthreads();
    if (tid == 0) {
        const int t = __hip_atomic_fetch_add(tickets + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == splits - 1);
        if (s_last) {
            __hip_atomic_store(tickets + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (!s_last) return;
    // merge: every partial of this clip is loaded (L1-bypassing, all loads in flight) before use
    constexpr int MAXS = 32;
    const float* pb = part + (long)b * splits * d;
    const float* mb = ml + (long)b * splits * n_heads * 2;
    for (int n = tid; n < d; n += 256) {
        const int h = n / WH_HEAD_DIM;
        float mv[MAXS], lv[MAXS], pv[MAXS];
#pragma unroll
        for (int s2 = 0; s2 < MAXS; s2++) {
            mv[s2] = -INFINITY; lv[s2] = 0.0f; pv[s2] = 0.0f;
            if (s2 < splits) {
                mv[s2] = __hip_atomic_load(mb + s2 * n_heads * 2 + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lv[s2] = __hip_atomic_load(mb + s2 * n_heads * 2 + n_heads + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pv[s2] = __hip_atomic_load(pb + (long)s2 * d + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        float M = -INFINITY;
#pragma unroll
        for (int s2 = 0; s2 < MAXS; s2++) M = fmaxf(M, mv[s2]);
        float num = 0.0f, den = 0.0f;
#pragma unroll
        for (int s2 = 0; s2 < MAXS; s2++) {
            const float w = (mv[s2] == -INFINITY) ? 0.0f : __expf(mv[s2] - M);
            num += w * pv[s2];
            den += w * lv[s2];
        }
        out[(long)b * d + n] = cvt_out<T>(num / den);
    }
}

// raise a kernel's dynamic-LDS limit once per (kernel, size)
template <typename K>
void set_max_smem(K kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return;
    static std::mutex mu;
    static std::unordered_map<const void*, size_t> done;
    std::lock_guard<std::mutex> lk(mu);
    size_t& cur = done[(const void*)kernel];
    if (bytes > cur) {
        (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        cur = bytes;
    }
}

template <typename T, typename TO, int SPLITK, int PRO, int MODE>
void launch_dec_gemm_mt(hipStream_t s, const SkinnyArgs& a) {
    const int n_tiles = (a.N + 15) / 16;
    int mt = std::min(4, (a.M + 15) / 16);
    auto lds = [&](int t) {
        return ((PRO == 0) ? 0 : (size_t)t * 16 * (a.K + 8) * sizeof(T)) + (SPLITK == 4 ? (size_t)4 * t * 64 * 16 : 0);
    };
    while (mt > 1 && lds(mt) > 150 * 1024) mt--;  // row groups along grid.y when the tile would not fit LDS
    const size_t sm = lds(mt);
    dim3 grid(SPLITK == 4 ? n_tiles : (n_tiles + 3) / 4, (a.M + 16 * mt - 1) / (16 * mt));
#define WH_LAUNCH(MT_)                                                                        \
    {                                                                                         \
        auto kfn = k_dec_gemm<T, TO, MT_, SPLITK, PRO, MODE>;                                 \
        set_max_smem(kfn, sm);                                                                \
        hipLaunchKernelGGL(kfn, grid, dim3(256), sm, s, a);                                   \
    }
    switch (mt) {
        case 1: WH_LAUNCH(1) break;
        case 2: WH_LAUNCH(2) break;
        case 3: WH_LAUNCH(3) break;
        default: WH_LAUNCH(4) break;
    }
#undef WH_LAUNCH
}

template <typename T, typename TO, int PRO>
void launch_dec_gemm_split(hipStream_t s, const SkinnyArgs& a) {
    const bool split = (a.K % 128 == 0) && a.N < 8192;
    if (split) launch_dec_gemm_mt<T, TO, 4, PRO, 0>(s, a);
    else launch_dec_gemm_mt<T, TO, 1, PRO, 0>(s, a);
}

}  // namespace

// pro: 0 = X from global, 1 = fused LayerNorm of a.xres, 2 = fused embedding + LayerNorm
void wh_launch_dec_gemm(hipStream_t s, int prec, bool out_f32, int pro, const SkinnyArgs& a) {
#define WH_PRO(T_, TO_)                                             \
    switch (pro) {                                                  \
        case 0: launch_dec_gemm_split<T_, TO_, 0>(s, a); break;     \
        case 1: launch_dec_gemm_split<T_, TO_, 1>(s, a); break;     \
        default: launch_dec_gemm_split<T_, TO_, 2>(s, a); break;    \
    }
    if (prec == WH_PREC_F32) { WH_PRO(float, float) }
    else if (out_f32) { WH_PRO(bf16, float) }
    else { WH_PRO(bf16, bf16) }
#undef WH_PRO
}

template <typename T>
void launch_lm_head_t(hipStream_t s, const SkinnyArgs& a) {
    const int n_tiles = (a.N + 15) / 16;
    int mt = std::min(4, (a.M + 15) / 16);
    auto lds = [&](int t) { return (size_t)t * 16 * (a.K + 8) * sizeof(T); };
    while (mt > 1 && lds(mt) > 150 * 1024) mt--;
    const size_t sm = lds(mt);
    const int per_cu = std::max<int>(1, (int)(150 * 1024 / sm));
    dim3 grid(std::min((n_tiles + 3) / 4, 256 * std::min(per_cu, 2)), (a.M + 16 * mt - 1) / (16 * mt));
#define WH_LM(MT_)                                                \
    {                                                             \
        auto kfn = k_lm_head<T, MT_>;                             \
        set_max_smem(kfn, sm);                                    \
        hipLaunchKernelGGL(kfn, grid, dim3(256), sm, s, a);       \
    }
    switch (mt) {
        case 1: WH_LM(1) break;
        case 2: WH_LM(2) break;
        case 3: WH_LM(3) break;
        default: WH_LM(4) break;
    }
#undef WH_LM
}

// a.X = final-LayerNorm'ed rows [M][K] in the compute dtype
void wh_launch_lm_head(hipStream_t s, int prec, const SkinnyArgs& a) {
    if (prec == WH_PREC_F32) launch_lm_head_t<float>(s, a);
    else launch_lm_head_t<bf16>(s, a);
}

void wh_launch_argmax_finish(hipStream_t s, const float* part_val, const int* part_idx, int n_tiles, int* pos_p,
                             int* ticket, const DecodeState& st, int B) {
    hipLaunchKernelGGL(k_argmax_finish, dim3(B), dim3(256), 0, s, part_val, part_idx, n_tiles, pos_p, ticket, st);
}

void wh_launch_dec_self_attn(hipStream_t s, int prec, const void* qkv, void* kc, void* vc, void* out, const int* pos_p,
                             int d, int n_heads, int tc, int B) {
    dim3 grid(n_heads, B);
    if (prec == WH_PREC_F32)
        hipLaunchKernelGGL(k_dec_self_attn<float>, grid, dim3(64), 0, s, (const float*)qkv, (float*)kc, (float*)vc, (float*)out, pos_p, d, n_heads, tc);
    else
        hipLaunchKernelGGL(k_dec_self_attn<bf16>, grid, dim3(64), 0, s, (const bf16*)qkv, (bf16*)kc, (bf16*)vc, (bf16*)out, pos_p, d, n_heads, tc);
}

void wh_launch_dec_cross_attn(hipStream_t s, int prec, const void* q, const void* ck, const void* cv, float* part,
                              float* ml, void* out, int* tickets, int S, int d, int n_heads, int splits, int B) {
    dim3 grid(splits, B);
    const size_t sm = sizeof(float) * ((size_t)8 * n_heads + 4 * (size_t)d);
#define WH_CA(T_, N_) hipLaunchKernelGGL((k_dec_cross_attn<T_, N_>), grid, dim3(256), sm, s, (const T_*)q, (const T_*)ck, \
                                         (const T_*)cv, part, ml, (T_*)out, tickets, S, d, n_heads, splits)
    if (prec == WH_PREC_F32) {
        const int nch = (d / 4 + 63) / 64;  // f32: 4 elements per chunk
        if (nch == 1) WH_CA(float, 1);
        else if (nch == 2) WH_CA(float, 2);
        else WH_CA(float, 5);
    } else {
        const int nch = (d / 8 + 63) / 64;
        if (nch == 1) WH_CA(bf16, 1);
        else WH_CA(bf16, 3);
    }
#undef WH_CA
}
