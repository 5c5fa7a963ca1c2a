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
#include <stdlib.h>

#include "wh_common.h"
#include "wh_kernels.h"

#include <algorithm>
#include <mutex>
#include <unordered_map>

// tuning knobs (tools/microbench.cpp flips them; product code leaves the defaults)
int wh_dbg_cross_unroll = 4;
int wh_dbg_lm_blocks_per_cu = 2;
int wh_dbg_lm_mt = 4;
int wh_dbg_mt = 0;
int wh_dbg_nw = 0;   // 4 / 8: force the K split of the decode GEMM (0 = heuristic)
int wh_dbg_wide = -1;  // column tiles per workgroup at > 256 rows: -1 heuristic, 0 off (k_dec_gemm only), 2 / 4 forced

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

// ---- decode activation layout ("k-slab major") ------------------------------------------------
// Activations that feed a decode GEMM are stored [K/32][Mpad][32]: the 32-deep k-slab of 16
// consecutive rows is one contiguous 1 KiB block, so the MFMA column-operand load of a wave (lane
// (row fl, group fg) reads 8 consecutive k of row 16t+fl) is a single fully coalesced access — the
// same shape as the weight-fragment loads — instead of 16 rows x 64 B scattered over a row-major
// [M][K] buffer.  Producers (LayerNorm, attention, fc1 epilogue) write this layout directly.
__device__ __forceinline__ long slab_idx(int m, int k, int mpad) { return ((long)(k >> 5) * mpad + m) * 32 + (k & 31); }

// Token embedding + learned position ([3P] :737, :757-766) for the rows of this position: writes the f32
// residual stream, its raw copy in the compute dtype (slab layout) and each row's {sum x, sum x^2}
// (one "tile" of LayerNorm partials).  One wave per row.
template <typename T>
__global__ __launch_bounds__(256) void k_dec_embed(const T* __restrict__ tok_emb, const float* __restrict__ pos_emb,
                                                   const int* __restrict__ feed, int feed_ld,
                                                   const int* __restrict__ pos_p, float* __restrict__ x,
                                                   T* __restrict__ xslab, float* __restrict__ stats, int rows, int d,
                                                   int mpad, const float* __restrict__ xgamma, float* __restrict__ shift) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63, pos = *pos_p;
    const T* er = tok_emb + (long)feed[row * feed_ld + pos] * d;
    const float* pr = pos_emb + (long)pos * d;
    // pass 1: the residual row and its mean; pass 2: the slab copy and the sums of the CENTRED row (SkinnyArgs::row_shift)
    float s0 = 0.0f;
    for (int c = lane * 4; c < d; c += 256) {
        const f32x4 v = load4_f32(er + c) + *reinterpret_cast<const f32x4*>(pr + c);
        *reinterpret_cast<f32x4*>(x + (long)row * d + c) = v;
        s0 += (v[0] + v[1]) + (v[2] + v[3]);
    }
    const float mean = shift ? dpp_wave_sum(s0) / (float)d : 0.0f;
    float s1 = 0.0f, s2 = 0.0f;
    for (int c = lane * 4; c < d; c += 256) {
        f32x4 v = load4_f32(er + c) + *reinterpret_cast<const f32x4*>(pr + c);
        v -= mean;
        f32x4 g = {1, 1, 1, 1};
        if (xgamma) g = *reinterpret_cast<const f32x4*>(xgamma + c);  // fp8 mode: the first LayerNorm's γ rides on the slab copy
        store4(xslab + slab_idx(row, c, mpad), v[0] * g[0], v[1] * g[1], v[2] * g[2], v[3] * g[3]);
        s1 += (v[0] + v[1]) + (v[2] + v[3]);
        s2 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    s1 = dpp_wave_sum(s1);
    s2 = dpp_wave_sum(s2);
    if (lane == 0) {
        stats[2 * row] = s1;
        stats[2 * row + 1] = s2;
        if (shift) shift[row] = mean;
    }
}

// ---- decode GEMM: C[m][n] = act(sum_k X[m][k] W[n][k] + bias[n]) (+ R[m][n]),  M <= 64 per row group
// Weight-streaming: every weight element is read once per launch straight into MFMA fragments (up
// to 8 per wave in flight before the first MFMA); activations are read as MFMA column operands from
// the slab layout.  The waves of a workgroup split K (4 or 8 ways) and reduce through LDS; wave w
// then finishes row-tile w (bias, erf-GELU, f32 residual; output row-major or slab).
// Merge of split attention partials (one row, 4 consecutive columns per item), in two phases so that the
// caller can put every load of several items in flight before any arithmetic.  SP = compile-time bound on
// the number of key ranges (everything unrolled, selects instead of branches: a branch would end the basic
// block and serialise the memory round trips).
template <int SP>
struct PartRaw {
    float mv[SP], lv[SP];
    f32x4 p[SP];
};
template <int SP>
__device__ __forceinline__ void partials_load(const SkinnyArgs& a, int m, int k, PartRaw<SP>& r) {
    const int h = k / WH_HEAD_DIM, hs = a.x_heads * 2;
    const float* mb = a.xml + (long)m * a.x_splits * hs + h;
    const float* pb = a.xpart + (long)m * a.x_splits * a.K + k;
#pragma unroll
    for (int s2 = 0; s2 < SP; s2++) {
        const int sc = s2 < a.x_splits ? s2 : 0;  // clamped re-read, weight forced to 0 in partials_merge
        r.mv[s2] = mb[sc * hs];
        r.lv[s2] = mb[sc * hs + a.x_heads];
        r.p[s2] = *reinterpret_cast<const f32x4*>(pb + (long)sc * a.K);
    }
}
template <int SP>
__device__ __forceinline__ f32x4 partials_merge(const SkinnyArgs& a, PartRaw<SP>& r, bool live) {
    float M = -INFINITY;
#pragma unroll
    for (int s2 = 0; s2 < SP; s2++) {
        r.mv[s2] = (s2 < a.x_splits) ? r.mv[s2] : -INFINITY;
        M = fmaxf(M, r.mv[s2]);
    }
    M = (M == -INFINITY) ? 0.0f : M;  // every range empty: exp(-inf - 0) = 0 below, never inf - inf
    f32x4 acc = {0, 0, 0, 0};
    float den = 0.0f;
#pragma unroll
    for (int s2 = 0; s2 < SP; s2++) {
        const float w = __builtin_amdgcn_exp2f((r.mv[s2] - M) * 1.44269504088896341f);  // empty key range: m = -inf, weight 0
        den += w * r.lv[s2];
#pragma unroll
        for (int e = 0; e < 4; e++) acc[e] += w * r.p[s2][e];
    }
    const float inv = live ? __builtin_amdgcn_rcpf(den) : 0.0f;
#pragma unroll
    for (int e = 0; e < 4; e++) acc[e] *= inv;
    return acc;
}

// Weight operand of one k-step.  Native dtype: 8 consecutive k per lane (one MFMA per load).  e4m3 codes: 8 per lane
// (fp8x8_t, one MFMA) or 16 per lane (fp8x16_t: one 16-byte load feeds two MFMAs — half the load instructions for
// the same bytes); codes are dequantised in registers with v_cvt_scalef32_pk_bf16_fp8 (byte j of the dword is
// element j, exact: tools/fp8_check.hip).  A k-step covers 4 * KW k-values; within it lane group fg holds
// k = fg*KW .. fg*KW+KW-1, sub-fragment j the eight starting at 8j — the activation operand follows the same map.
struct fp8x8_t { unsigned char v; };
struct fp8x16_t { unsigned char v; };
template <typename T, typename TW> struct WTraits { static constexpr int KW = 8; };
template <> struct WTraits<bf16, fp8x16_t> { static constexpr int KW = 16; };

__device__ __forceinline__ bf16x8 cvt8_e4m3_bf16(unsigned lo, unsigned hi) {
    const bf16x2 a = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, false), b = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, true);
    const bf16x2 c = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, false), d = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, true);
    return bf16x8{a.x, a.y, b.x, b.y, c.x, c.y, d.x, d.y};
}
template <typename T, typename TW>
__device__ __forceinline__ void load_wfrags(const TW* p, typename FragT<T>::type (&out)[WTraits<T, TW>::KW / 8]) {
    if constexpr (sizeof(TW) == sizeof(T)) {
        out[0] = load_frag<T>(reinterpret_cast<const T*>(p));
    } else if constexpr (WTraits<T, TW>::KW == 8) {
        typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
        const u32x2 u = *reinterpret_cast<const u32x2*>(p);
        out[0] = cvt8_e4m3_bf16(u.x, u.y);
    } else {
        const wh_u32x4 u = *reinterpret_cast<const wh_u32x4*>(p);
        out[0] = cvt8_e4m3_bf16(u.x, u.y);
        out[1] = cvt8_e4m3_bf16(u.z, u.w);
    }
}

// grouped launches (SkinnyArgs::zn): group blockIdx.z works on its own X, W, bias and C
template <typename T, typename TO, typename TW>
__device__ __forceinline__ void dec_gemm_group(SkinnyArgs& a) {
    const long z = blockIdx.z;
    if (z == 0) return;
    a.X = (const T*)a.X + z * a.x_zs;
    a.W = (const TW*)a.W + z * a.w_zs;
    a.C = a.c_mpad ? (void*)((T*)a.C + z * a.c_zs) : (void*)((TO*)a.C + z * a.c_zs);
    if (a.bias) a.bias += z * a.bias_zs;
}

template <typename T, typename TO, int MT, int NW, int XP, typename TW = T>  // XP > 0: X from XP-bounded attention partials
__global__ __launch_bounds__(NW * 64) void k_dec_gemm(SkinnyArgs a) {
    extern __shared__ __attribute__((aligned(128))) char smem_raw[];   // 128: h2 tiles find their 32-blocks from the address
    dec_gemm_group<T, TO, TW>(a);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fl = lane & 15, fg = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int m0 = blockIdx.y * MT * 16;  // first row of this workgroup's row group
    int nrow = n0 + fl;
    if (nrow > a.N - 1) nrow = a.N - 1;
    constexpr int KW = WTraits<T, TW>::KW, SUB = KW / 8, KSTEP = 4 * KW;  // k-values per lane / MFMAs / k-values per step
    const int kspan = a.K / NW, kb = wave * kspan, iters = kspan / KSTEP;
    const TW* wp = (const TW*)a.W + (long)nrow * a.K + kb + fg * KW;
    // activation operand of (step i, sub-fragment j): slab (kb + i*KSTEP + fg*KW) / 32, offset (fg*KW) % 32 + 8j
    const long xstep = (long)a.x_mpad * 32 * (KSTEP / 32);
    const T* xp = XP > 0 ? nullptr : (const T*)a.X + ((long)((kb + fg * KW) >> 5) * a.x_mpad + m0 + fl) * 32 + ((fg * KW) & 31);
    // k-steps in flight per wave: 8 for 16-byte fragments; 32-byte ones (f32, fp16 limbs) keep the register budget of MT = 1
    // (8 x (1 + MT) fragments would spill from MT = 2 on: 1,074 spilled registers at MT = 4 before this rule)
    constexpr int FRB = (int)sizeof(typename FragT<T>::type);
    constexpr int DEPTH = FRB <= 16 ? 8 / SUB : (MT == 1 ? 8 : MT == 2 ? 4 : 2);
    typename FragT<T>::type wq[DEPTH][SUB], xq[DEPTH][SUB][MT];
#pragma unroll
    for (int i = 0; i < DEPTH; i++)
        if (i < iters) {
            load_wfrags<T, TW>(wp + i * KSTEP, wq[i]);
            if constexpr (XP == 0) {
#pragma unroll
                for (int j = 0; j < SUB; j++)
#pragma unroll
                    for (int t = 0; t < MT; t++) xq[i][j][t] = load_frag<T>(xp + i * xstep + t * 512 + 8 * j);
            }
        }
    // epilogue operands (wave w finishes row-tile w): fetched now, used last
    f32x4 pre_bias = {0, 0, 0, 0}, pre_r = {0, 0, 0, 0}, pre_ws = {1, 1, 1, 1}, pre_g = {1, 1, 1, 1};
    float pre_sh = 0.0f;   // the row's running offset (producers: SkinnyArgs::row_shift)
    const int en = n0 + 4 * fg, em = m0 + wave * 16 + fl;
    const bool ep_ok = wave < MT && en < a.N && em < a.M;
    if (ep_ok) {
        if (a.bias) pre_bias = *reinterpret_cast<const f32x4*>(a.bias + en);
        if (a.wscale) pre_ws = *reinterpret_cast<const f32x4*>(a.wscale + en);
        if (a.xgamma) pre_g = *reinterpret_cast<const f32x4*>(a.xgamma + en);
        if (a.R) pre_r = *reinterpret_cast<const f32x4*>(a.R + (long)em * a.ldr + en);
        if (a.row_shift) pre_sh = a.row_shift[em];
    }
    // LayerNorm folded in: reduce the producer's per-tile partial sums of this row group to mean / rstd
    // (while the weight and activation fragments above are in flight)
    float* lnred = reinterpret_cast<float*>(smem_raw) + (size_t)NW * MT * 64 * 4;  // [4][MT*16][2], then stat[MT*16][2]
    float ln_mean = 0.0f, ln_rstd = 1.0f, ln_sv[4] = {0, 0, 0, 0};
    if (a.ln_part) {
        constexpr int ROWS = MT * 16;
        if (tid < 4 * ROWS) {
            const int r = tid % ROWS, q = tid / ROWS;
            float s1, s2;
            ln_partial_sum(a.ln_part, a.ln_tiles, a.x_mpad, m0 + r, q, 4, s1, s2);
            lnred[(q * ROWS + r) * 2] = s1;
            lnred[(q * ROWS + r) * 2 + 1] = s2;
        }
        __syncthreads();
        if (ep_ok) {
            const int r = wave * 16 + fl;
            const float s1 = (lnred[r * 2] + lnred[(ROWS + r) * 2]) + (lnred[(2 * ROWS + r) * 2] + lnred[(3 * ROWS + r) * 2]);
            const float s2 = (lnred[r * 2 + 1] + lnred[(ROWS + r) * 2 + 1]) + (lnred[(2 * ROWS + r) * 2 + 1] + lnred[(3 * ROWS + r) * 2 + 1]);
            wh_ln_mean_rstd(s1, s2, (float)a.K, false, ln_mean, ln_rstd);
            const f32x4 sv = *reinterpret_cast<const f32x4*>(a.ln_s + en);
            ln_sv[0] = sv[0]; ln_sv[1] = sv[1]; ln_sv[2] = sv[2]; ln_sv[3] = sv[3];
        }
    }
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; t++) acc[t] = f32x4{0, 0, 0, 0};
    if constexpr (XP > 0) {
        // X = merge of the key ranges' partials, built once per workgroup in LDS (slab layout [K/32][16][32]) with
        // row-contiguous 16-byte loads; the weight fragments above are in flight meanwhile (iters <= DEPTH here)
        static_assert(MT == 1, "merged-X variant runs 16-row groups");
        T* Xs = reinterpret_cast<T*>(smem_raw + (size_t)NW * 64 * 16 + 4 * 16 * 2 * 4);
        constexpr int NT = NW * 64, IF = XP <= 4 ? 4 : (XP <= 8 ? 2 : 1);    // items in flight per thread
        // 4-column chunks per row; only live rows are merged (a dead row of X feeds only its own, never stored,
        // output column of the MFMA tile)
        const int cpr = a.K >> 2, items = min(16, a.M - m0) * cpr;
        for (int it0 = tid; it0 < items; it0 += NT * IF) {
            PartRaw<XP> raw[IF];
            int row[IF], ch[IF];
            bool live[IF];
#pragma unroll
            for (int u = 0; u < IF; u++) {
                const int item = min(it0 + u * NT, items - 1);
                row[u] = item / cpr;
                ch[u] = item - row[u] * cpr;
                live[u] = it0 + u * NT < items;
                partials_load<XP>(a, m0 + row[u], ch[u] * 4, raw[u]);
            }
            __builtin_amdgcn_sched_barrier(0);  // all loads above, all arithmetic below
#pragma unroll
            for (int u = 0; u < IF; u++) {
                const f32x4 v = partials_merge<XP>(a, raw[u], live[u]);
                const int k = ch[u] * 4;
                if (live[u]) store4(Xs + ((long)(k >> 5) * 16 + row[u]) * 32 + (k & 31), v[0], v[1], v[2], v[3]);
            }
        }
        __syncthreads();
        const T* xl = Xs + ((long)((kb + fg * KW) >> 5) * 16 + fl) * 32 + ((fg * KW) & 31);
        constexpr int lstep = 512 * (KSTEP / 32);
#pragma unroll
        for (int i = 0; i < DEPTH; i++)
            if (i < iters) {
#pragma unroll
                for (int j = 0; j < SUB; j++) mma16(acc[0], wq[i][j], load_frag<T>(xl + i * lstep + 8 * j));
            }
        for (int i = DEPTH; i < iters; i++) {  // deep K tail
            typename FragT<T>::type wt[SUB];
            load_wfrags<T, TW>(wp + i * KSTEP, wt);
#pragma unroll
            for (int j = 0; j < SUB; j++) mma16(acc[0], wt[j], load_frag<T>(xl + i * lstep + 8 * j));
        }
    } else
    for (int c0 = 0; c0 < iters; c0 += DEPTH) {
        if (c0 > 0) {
#pragma unroll
            for (int i = 0; i < DEPTH; i++)
                if (c0 + i < iters) {
                    load_wfrags<T, TW>(wp + (c0 + i) * KSTEP, wq[i]);
#pragma unroll
                    for (int j = 0; j < SUB; j++)
#pragma unroll
                        for (int t = 0; t < MT; t++) xq[i][j][t] = load_frag<T>(xp + (c0 + i) * xstep + t * 512 + 8 * j);
                }
        }
#pragma unroll
        for (int i = 0; i < DEPTH; i++)
            if (c0 + i < iters) {
#pragma unroll
                for (int j = 0; j < SUB; j++)
#pragma unroll
                    for (int t = 0; t < MT; t++) mma16(acc[t], wq[i][j], xq[i][j][t]);  // D rows = n (4*fg + r), col = m (fl)
            }
    }
    f32x4* red = reinterpret_cast<f32x4*>(smem_raw);
#pragma unroll
    for (int t = 0; t < MT; t++) red[(wave * MT + t) * 64 + lane] = acc[t];
    __syncthreads();
    if (ep_ok) {
        f32x4 s = red[(0 * MT + wave) * 64 + lane];
#pragma unroll
        for (int w = 1; w < NW; w++) {
            f32x4 o = red[(w * MT + wave) * 64 + lane];
            s[0] += o[0]; s[1] += o[1]; s[2] += o[2]; s[3] += o[3];
        }
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            s[e] = wh_scale(s[e], pre_ws[e]);  // fp8 weights: the channel scale (1 otherwise)
            v[e] = a.ln_part ? wh_ln_fold(s[e], ln_mean, ln_rstd, ln_sv[e], pre_bias[e]) : wh_add(s[e], pre_bias[e]);
            if (a.act == 1) v[e] = gelu_erf(v[e]);
            v[e] += pre_r[e];
        }
        // a slab output is another GEMM's operand: the compute type T (== TO wherever both are 2-byte); row-major outputs are TO
        if (a.c_mpad) store4((T*)a.C + slab_idx(em, en, a.c_mpad), v[0], v[1], v[2], v[3]);
        else store4((TO*)a.C + (long)em * a.ldc + en, v[0], v[1], v[2], v[3]);
        if (a.xslab_out) store4((T*)a.xslab_out + slab_idx(em, en, a.x_mpad), wh_sub(v[0], pre_sh) * pre_g[0], wh_sub(v[1], pre_sh) * pre_g[1],
                                wh_sub(v[2], pre_sh) * pre_g[2], wh_sub(v[3], pre_sh) * pre_g[3]);
        // the one consumer of a LayerNorm keeps the rows' running offsets up to date (first column tile only)
        if (a.shift_io && blockIdx.x == 0 && blockIdx.z == 0 && fg == 0) a.shift_io[em] += ln_mean;
    }
    if (a.stats_out && wave < MT) {
        // this column tile's {sum x, sum x^2} per row: 4 values per lane, then the 4 lane groups of the row
        float s1 = 0.0f, s2 = 0.0f;
        if (ep_ok) {
            f32x4 s = red[(0 * MT + wave) * 64 + lane];
#pragma unroll
            for (int w = 1; w < NW; w++) {
                f32x4 o = red[(w * MT + wave) * 64 + lane];
                s[0] += o[0]; s[1] += o[1]; s[2] += o[2]; s[3] += o[3];
            }
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float v = wh_sub(wh_add(wh_add(wh_scale(s[e], pre_ws[e]), pre_bias[e]), pre_r[e]), pre_sh);  // producers have no activation
                s1 += v;
                s2 = __builtin_fmaf(v, v, s2);   // explicit: the same bits in k_dec_gemm and k_dec_gemm_wide
            }
        }
        s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
        if (fg == 0 && em < a.M) {
            a.stats_out[((long)blockIdx.x * a.x_mpad + em) * 2] = s1;
            a.stats_out[((long)blockIdx.x * a.x_mpad + em) * 2 + 1] = s2;
        }
    }
    advance_if_last(a.ticket, a.pos_w, gridDim.x * gridDim.y);
}

// Batches of hundreds of rows: a workgroup owns NT 16-column tiles x MT 16-row tiles, so every activation fragment a wave
// loads feeds NT MFMAs and every weight fragment MT (k_dec_gemm at NT = 1 pulls the activations through L2 once per 16
// output columns: 123 MB for the 1536 x 512 QKV projection at 1024 rows, 49 MB here).  Same K split over the waves, same
// k order inside a wave and the same order of the cross-wave sum as k_dec_gemm: an output element is bit-identical
// whichever of the two computes it, so the choice may follow the batch.  Wave w finishes tiles w, w + NW, ...
// (tile id = nt * MT + mt).  No merged-X (attention partials) variant: large batches run one key range per clip.
template <typename T, typename TO, int MT, int NT, int NW, typename TW = T>
__global__ __launch_bounds__(NW * 64, 2) void k_dec_gemm_wide(SkinnyArgs a) {
    extern __shared__ __attribute__((aligned(128))) char smem_raw[];   // 128: h2 tiles find their 32-blocks from the address
    dec_gemm_group<T, TO, TW>(a);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fl = lane & 15, fg = lane >> 4;
    const int n0 = blockIdx.x * 16 * NT;
    const int m0 = blockIdx.y * MT * 16;
    constexpr int KW = WTraits<T, TW>::KW, SUB = KW / 8, KSTEP = 4 * KW;
    constexpr int TILES = MT * NT, TPW = (TILES + NW - 1) / NW;
    const int kspan = a.K / NW, kb = wave * kspan, iters = kspan / KSTEP;
    const TW* wp[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) wp[nt] = (const TW*)a.W + (long)min(n0 + nt * 16 + fl, a.N - 1) * a.K + kb + fg * KW;
    const long xstep = (long)a.x_mpad * 32 * (KSTEP / 32);
    const T* xp = (const T*)a.X + ((long)((kb + fg * KW) >> 5) * a.x_mpad + m0 + fl) * 32 + ((fg * KW) & 31);
    // four 32-deep k-slabs in flight per wave: (NT + MT) KiB each; two with 32-byte fragments (the register file holds NT + MT of them per slab)
    constexpr int DEPTH = sizeof(typename FragT<T>::type) <= 16 ? 4 / SUB : 2;
    typename FragT<T>::type wq[DEPTH][NT][SUB], xq[DEPTH][SUB][MT];
#pragma unroll
    for (int i = 0; i < DEPTH; i++)
        if (i < iters) {
#pragma unroll
            for (int nt = 0; nt < NT; nt++) load_wfrags<T, TW>(wp[nt] + i * KSTEP, wq[i][nt]);
#pragma unroll
            for (int j = 0; j < SUB; j++)
#pragma unroll
                for (int t = 0; t < MT; t++) xq[i][j][t] = load_frag<T>(xp + i * xstep + t * 512 + 8 * j);
        }
    // the tiles this wave finishes; their residual rows are fetched now, the per-column operands after the main loop
    int en[TPW], em[TPW];
    bool ep_ok[TPW];
    f32x4 pre_r[TPW];
    float pre_sh[TPW];   // the rows' running offsets (producers: SkinnyArgs::row_shift)
#pragma unroll
    for (int j = 0; j < TPW; j++) {
        const int t = wave + j * NW;
        en[j] = n0 + (t / MT) * 16 + 4 * fg;
        em[j] = m0 + (t % MT) * 16 + fl;
        ep_ok[j] = t < TILES && en[j] < a.N && em[j] < a.M;
        pre_r[j] = f32x4{0, 0, 0, 0};
        pre_sh[j] = 0.0f;
        if (ep_ok[j] && a.R) pre_r[j] = *reinterpret_cast<const f32x4*>(a.R + (long)em[j] * a.ldr + en[j]);
        if (ep_ok[j] && a.row_shift) pre_sh[j] = a.row_shift[em[j]];
    }
    float* lnred = reinterpret_cast<float*>(smem_raw) + (size_t)NW * TILES * 64 * 4;  // [4][MT*16][2]
    if (a.ln_part) {
        constexpr int ROWS = MT * 16;
        static_assert(4 * ROWS <= NW * 64, "one thread per (quarter, row)");
        if (tid < 4 * ROWS) {
            const int r = tid % ROWS, q = tid / ROWS;
            float s1, s2;
            ln_partial_sum(a.ln_part, a.ln_tiles, a.x_mpad, m0 + r, q, 4, s1, s2);
            lnred[(q * ROWS + r) * 2] = s1;
            lnred[(q * ROWS + r) * 2 + 1] = s2;
        }
    }
    // per-column operands of the tiles this wave finishes: requested AFTER the main loop.  Requesting them before it where the
    // register file has room (two column tiles per workgroup) was measured slower in round 3 — the trace of the whole step reads
    // 9.68 instead of 8.74 us for the QKV / cross-Q launches, 7.17 instead of 6.97 us for the out-projections
    // (profiles/r03_kernel_stats_base_bf16_b1024.txt history) — they only lengthen the queue in front of the first fragments.
    constexpr bool EARLY_COLS = false;
    f32x4 pre_bias[TPW], pre_ws[TPW], pre_g[TPW], pre_sv[TPW];
    auto load_cols = [&]() {
#pragma unroll
        for (int j = 0; j < TPW; j++) {
            pre_bias[j] = f32x4{0, 0, 0, 0}; pre_ws[j] = f32x4{1, 1, 1, 1}; pre_g[j] = f32x4{1, 1, 1, 1}; pre_sv[j] = f32x4{0, 0, 0, 0};
            if (ep_ok[j]) {
                if (a.bias) pre_bias[j] = *reinterpret_cast<const f32x4*>(a.bias + en[j]);
                if (a.wscale) pre_ws[j] = *reinterpret_cast<const f32x4*>(a.wscale + en[j]);
                if (a.xgamma) pre_g[j] = *reinterpret_cast<const f32x4*>(a.xgamma + en[j]);
                if (a.ln_part) pre_sv[j] = *reinterpret_cast<const f32x4*>(a.ln_s + en[j]);
            }
        }
    };
    if constexpr (EARLY_COLS) load_cols();
    f32x4 acc[NT][MT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int t = 0; t < MT; t++) acc[nt][t] = f32x4{0, 0, 0, 0};
    // rolling prefetch: a slot is refilled with the k-slab DEPTH ahead as soon as its MFMAs are issued, so the next round's
    // loads fly under this round's arithmetic instead of starting after it
    for (int c0 = 0; c0 < iters; c0 += DEPTH) {
#pragma unroll
        for (int i = 0; i < DEPTH; i++) {
            if (c0 + i < iters) {
#pragma unroll
                for (int j = 0; j < SUB; j++)
#pragma unroll
                    for (int nt = 0; nt < NT; nt++)
#pragma unroll
                        for (int t = 0; t < MT; t++) mma16(acc[nt][t], wq[i][nt][j], xq[i][j][t]);
            }
            if (c0 + DEPTH + i < iters) {
#pragma unroll
                for (int nt = 0; nt < NT; nt++) load_wfrags<T, TW>(wp[nt] + (c0 + DEPTH + i) * KSTEP, wq[i][nt]);
#pragma unroll
                for (int j = 0; j < SUB; j++)
#pragma unroll
                    for (int t = 0; t < MT; t++) xq[i][j][t] = load_frag<T>(xp + (c0 + DEPTH + i) * xstep + t * 512 + 8 * j);
            }
        }
    }
    if constexpr (!EARLY_COLS) load_cols();
    f32x4* red = reinterpret_cast<f32x4*>(smem_raw);
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int t = 0; t < MT; t++) red[(wave * TILES + nt * MT + t) * 64 + lane] = acc[nt][t];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TPW; j++) {
        const int t = wave + j * NW;
        if (t >= TILES) break;  // wave-uniform
        f32x4 s = {0, 0, 0, 0};
        float v[4] = {0, 0, 0, 0};
        if (ep_ok[j]) {
            s = red[(0 * TILES + t) * 64 + lane];
#pragma unroll
            for (int w = 1; w < NW; w++) {
                f32x4 o = red[(w * TILES + t) * 64 + lane];
                s[0] += o[0]; s[1] += o[1]; s[2] += o[2]; s[3] += o[3];
            }
            float ln_mean = 0.0f, ln_rstd = 1.0f;
            if (a.ln_part) {
                constexpr int ROWS = MT * 16;
                const int r = (t % MT) * 16 + fl;
                const float s1 = (lnred[r * 2] + lnred[(ROWS + r) * 2]) + (lnred[(2 * ROWS + r) * 2] + lnred[(3 * ROWS + r) * 2]);
                const float s2 = (lnred[r * 2 + 1] + lnred[(ROWS + r) * 2 + 1]) + (lnred[(2 * ROWS + r) * 2 + 1] + lnred[(3 * ROWS + r) * 2 + 1]);
                wh_ln_mean_rstd(s1, s2, (float)a.K, false, ln_mean, ln_rstd);
            }
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float sc = wh_scale(s[e], pre_ws[j][e]);  // fp8 weights: the channel scale (1 otherwise)
                v[e] = a.ln_part ? wh_ln_fold(sc, ln_mean, ln_rstd, pre_sv[j][e], pre_bias[j][e]) : wh_add(sc, pre_bias[j][e]);
                if (a.act == 1) v[e] = gelu_erf(v[e]);
                v[e] += pre_r[j][e];
            }
            if (a.c_mpad) store4((T*)a.C + slab_idx(em[j], en[j], a.c_mpad), v[0], v[1], v[2], v[3]);
            else store4((TO*)a.C + (long)em[j] * a.ldc + en[j], v[0], v[1], v[2], v[3]);
            if (a.xslab_out) store4((T*)a.xslab_out + slab_idx(em[j], en[j], a.x_mpad), wh_sub(v[0], pre_sh[j]) * pre_g[j][0], wh_sub(v[1], pre_sh[j]) * pre_g[j][1],
                                    wh_sub(v[2], pre_sh[j]) * pre_g[j][2], wh_sub(v[3], pre_sh[j]) * pre_g[j][3]);
            if (a.shift_io && blockIdx.x == 0 && blockIdx.z == 0 && t / MT == 0 && fg == 0) a.shift_io[em[j]] += ln_mean;   // (first column tile only)
        }
        if (a.stats_out) {
            // this column tile's {sum x, sum x^2} per row (producers have no activation: v = s*ws + bias + r, as in k_dec_gemm)
            float s1 = 0.0f, s2 = 0.0f;
            if (ep_ok[j]) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float u = wh_sub(wh_add(wh_add(wh_scale(s[e], pre_ws[j][e]), pre_bias[j][e]), pre_r[j][e]), pre_sh[j]);
                    s1 += u;
                    s2 = __builtin_fmaf(u, u, s2);
                }
            }
            s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
            s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
            if (fg == 0 && em[j] < a.M) {
                const long tile = (long)blockIdx.x * NT + t / MT;
                a.stats_out[(tile * a.x_mpad + em[j]) * 2] = s1;
                a.stats_out[(tile * a.x_mpad + em[j]) * 2 + 1] = s2;
            }
        }
    }
    advance_if_last(a.ticket, a.pos_w, gridDim.x * gridDim.y);
}

// ---- LM head: logits = LN(x) · E^T over the whole vocabulary ([3P] :790, :965-970) + masked argmax
// partials per 16-column tile (reference argmax_last_dim_raw, src/main.rs:709-735).  The [M][K]
// activation tile is staged in LDS once per workgroup; each wave then walks 16-row tiles of the
// embedding matrix with all of a tile's weight fragments in flight before its first MFMA.
template <typename T, int MT>
__global__ __launch_bounds__(256) void k_lm_head(SkinnyArgs a) {
    extern __shared__ __attribute__((aligned(128))) char smem_raw[];   // 128: h2 tiles find their 32-blocks from the address
    constexpr int EPC = 16 / (int)sizeof(T);
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fl = lane & 15, fg = lane >> 4;
    const int n_tiles = (a.N + 15) >> 4;
    const int m0 = blockIdx.y * MT * 16;
    // activation tile: the row group's rows of every k-slab, LDS layout [K/32][MT*16][32]
    T* Xs = reinterpret_cast<T*>(smem_raw);
    constexpr int ROWS = MT * 16;
    const int nslab = a.K >> 5, cps = ROWS * 32 / EPC;  // 16-B chunks per slab of this row group
    // sixteen 16-byte loads in flight per thread, then the LDS stores (a load -> store loop costs a round trip per
    // chunk); the first batch's stores wait until the first weight unit is in flight too (see below)
    u32x4 xr[16];
    int xo[16];
    auto stage_load = [&](int c0) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int c = min(c0 + i * 256, nslab * cps - 1), sl = c / cps, o = (c - sl * cps) * EPC;
            xo[i] = sl * ROWS * 32 + o;
            xr[i] = *reinterpret_cast<const u32x4*>((const T*)a.X + ((long)sl * a.x_mpad + m0) * 32 + o);
        }
    };
    auto stage_store = [&](int c0) {
#pragma unroll
        for (int i = 0; i < 16; i++)
            if (c0 + i * 256 < nslab * cps) *reinterpret_cast<u32x4*>(Xs + xo[i]) = xr[i];
    };
    stage_load(tid);
    // final LayerNorm folded in: mean / rstd of this row group from the producer's per-tile partial sums
    float* lnstat = reinterpret_cast<float*>(smem_raw + (size_t)nslab * ROWS * 32 * sizeof(T));  // [ROWS][2]
    if (a.ln_part) {
        // all 256 threads: thread (row r, quarter q) sums every 4th tile, the four quarters meet through LDS
        float* lnq = lnstat + 2 * ROWS;  // [4][ROWS][2]
        if (tid < 4 * ROWS) {
            const int r = tid % ROWS, q = tid / ROWS;
            float s1, s2;
            ln_partial_sum(a.ln_part, a.ln_tiles, a.x_mpad, m0 + r, q, 4, s1, s2);
            lnq[(q * ROWS + r) * 2] = s1;
            lnq[(q * ROWS + r) * 2 + 1] = s2;
        }
        __syncthreads();
        if (tid < ROWS) {
            const float s1 = (lnq[tid * 2] + lnq[(ROWS + tid) * 2]) + (lnq[(2 * ROWS + tid) * 2] + lnq[(3 * ROWS + tid) * 2]);
            const float s2 = (lnq[tid * 2 + 1] + lnq[(ROWS + tid) * 2 + 1]) + (lnq[(2 * ROWS + tid) * 2 + 1] + lnq[(3 * ROWS + tid) * 2 + 1]);
            float mean, rstd;
            wh_ln_mean_rstd(s1, s2, (float)a.K, false, mean, rstd);
            lnstat[2 * tid] = mean;
            lnstat[2 * tid + 1] = rstd;
        }
    }
    const int pos = *a.pos_p;
    const int gen = pos - (a.n_prompt - 1);  // index of the token this row generates
    const unsigned* mask = (gen == 0) ? a.mask_first : a.mask_base;
    // Pipeline unit = (tile, 16-fragment chunk of K).  Two register sets: the next unit's weight
    // fragments are in flight while the current unit's MFMAs issue.
    const int iters = a.K >> 5;
    constexpr int DEPTH = sizeof(typename FragT<T>::type) <= 16 ? 16 : 8;   // two register sets of DEPTH weight fragments
    const int nchunk = (iters + DEPTH - 1) / DEPTH;
    const int stride = gridDim.x * 4, first = blockIdx.x * 4 + wave;
    const int my_tiles = first < n_tiles ? (n_tiles - first + stride - 1) / stride : 0;
    const int units = my_tiles * nchunk;
    typedef typename FragT<T>::type frag_t;
    frag_t wA[DEPTH], wB[DEPTH];
    f32x4 acc[MT];
    // masked argmax, lane-local across ALL of this wave's tiles (they are visited in increasing column order, so
    // strict > keeps the lowest index on ties): the cross-lane reduce and the partial store happen once per wave
    float bvw[MT];
    int biw[MT];
#pragma unroll
    for (int t = 0; t < MT; t++) { bvw[t] = -INFINITY; biw[t] = 0x7fffffff; }
    auto load_unit = [&](frag_t (&wq)[DEPTH], int u) {
        const int tile = first + (u / nchunk) * stride, c0 = (u % nchunk) * DEPTH;
        int nrow = tile * 16 + fl;
        if (nrow > a.N - 1) nrow = a.N - 1;
        const T* wp = (const T*)a.W + (long)nrow * a.K + fg * 8;
#pragma unroll
        for (int i = 0; i < DEPTH; i++)
            if (c0 + i < iters) wq[i] = load_frag<T>(wp + (c0 + i) * 32);
    };
    auto compute_unit = [&](const frag_t (&wq)[DEPTH], int u) {
        const int tile = first + (u / nchunk) * stride, ck = u % nchunk, c0 = ck * DEPTH;
        if (ck == 0) {
#pragma unroll
            for (int t = 0; t < MT; t++) acc[t] = f32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < DEPTH; i++) {
            if (c0 + i < iters) {
#pragma unroll
                for (int t = 0; t < MT; t++) {
                    frag_t xf = load_frag<T>(Xs + ((long)(c0 + i) * ROWS + t * 16 + fl) * 32 + fg * 8);
                    mma16(acc[t], wq[i], xf);
                }
            }
        }
        if (ck != nchunk - 1) return;
        const int n = tile * 16 + 4 * fg;
        float sv[4] = {0, 0, 0, 0}, cv[4] = {0, 0, 0, 0};
        if (a.ln_part) {
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (n + e < a.N) { sv[e] = a.ln_s[n + e]; cv[e] = a.bias[n + e]; }
        }
        unsigned mbits = 0;  // suppress bits of this lane's 4 columns (one mask word covers a 16-column tile)
        if (n < a.N) mbits = mask[n >> 5] >> (n & 31);
#pragma unroll
        for (int t = 0; t < MT; t++) {
            const int m = m0 + t * 16 + fl;
            const float mean = a.ln_part ? lnstat[2 * (t * 16 + fl)] : 0.0f, rstd = a.ln_part ? lnstat[2 * (t * 16 + fl) + 1] : 1.0f;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int nn = n + e;
                const float v = a.ln_part ? wh_ln_fold(acc[t][e], mean, rstd, sv[e], cv[e]) : acc[t][e];
                if (nn < a.N && m < a.M) {
                    if (a.logits && gen >= 0 && gen < a.logits_rows) {
                        const int slot = a.logits_sel ? a.logits_sel[m] : m;
                        if (slot >= 0) a.logits[((long)slot * a.logits_rows + gen) * a.N + nn] = v;
                    }
                    const bool sup = (mbits >> e) & 1u;
                    if (!sup && v > bvw[t]) { bvw[t] = v; biw[t] = nn; }  // strict >, NaN never wins
                }
            }
        }
    };
    if (units > 0) load_unit(wA, 0);   // in flight while the activation tile is staged
    stage_store(tid);
    for (int c0 = tid + 256 * 16; c0 < nslab * cps; c0 += 256 * 16) {  // K > 512
        stage_load(c0);
        stage_store(c0);
    }
    __syncthreads();
    for (int u = 0; u < units; u += 2) {
        const bool hasB = u + 1 < units;
        if (hasB) load_unit(wB, u + 1);
        compute_unit(wA, u);
        if (hasB) {
            if (u + 2 < units) load_unit(wA, u + 2);
            compute_unit(wB, u + 1);
        }
    }
    // one partial per (wave, row): argmax over the four lane groups of a row on v_permlane*_swap, then 16 consecutive
    // rows leave as one 64-byte store — layout [part][x_mpad], part = blockIdx.x * 4 + wave (k_argmax_finish reads
    // gridDim.x * 4 parts per row)
    const int part = blockIdx.x * 4 + wave;
#pragma unroll
    for (int t = 0; t < MT; t++) {
        float bv = bvw[t];
        int bi = biw[t];
        wh_u32x2 tv = __builtin_amdgcn_permlane16_swap(__float_as_uint(bv), __float_as_uint(bv), false, false);
        wh_u32x2 ti = __builtin_amdgcn_permlane16_swap((unsigned)bi, (unsigned)bi, false, false);
        float v0 = __uint_as_float(tv.x), v1 = __uint_as_float(tv.y);
        int i0 = (int)ti.x, i1 = (int)ti.y;
        bool take1 = v1 > v0 || (v1 == v0 && i1 < i0);
        bv = take1 ? v1 : v0;
        bi = take1 ? i1 : i0;
        tv = __builtin_amdgcn_permlane32_swap(__float_as_uint(bv), __float_as_uint(bv), false, false);
        ti = __builtin_amdgcn_permlane32_swap((unsigned)bi, (unsigned)bi, false, false);
        v0 = __uint_as_float(tv.x); v1 = __uint_as_float(tv.y);
        i0 = (int)ti.x; i1 = (int)ti.y;
        take1 = v1 > v0 || (v1 == v0 && i1 < i0);
        bv = take1 ? v1 : v0;
        bi = take1 ? i1 : i0;
        const int m = m0 + t * 16 + fl;
        if (fg == 0 && m < a.M) {
            a.part_val[(long)part * a.x_mpad + m] = bv;
            a.part_idx[(long)part * a.x_mpad + m] = bi;
        }
    }
}

// Final reduce of the per-tile argmax partials + greedy bookkeeping for one clip per workgroup:
// records the generated token, EOT stop (src/main.rs:781-783, 820-822) and the next input token.
template <typename T>
__global__ __launch_bounds__(256) void k_argmax_finish(const float* __restrict__ part_val,
                                                       const int* __restrict__ part_idx, int n_tiles, int mpad, int* pos_p,
                                                       int* ticket, DecodeState st, NextEmbed ne) {
    __shared__ float sv[256];
    __shared__ int si[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int pos = *pos_p;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i0 = tid; i0 < n_tiles; i0 += 256 * 8) {   // n_tiles = number of partials per row, layout [part][mpad]
        float v[8];
        int ix[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = i0 + u * 256;
            v[u] = -INFINITY;
            ix[u] = 0x7fffffff;
            if (i < n_tiles) { v[u] = part_val[(long)i * mpad + b]; ix[u] = part_idx[(long)i * mpad + b]; }
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
        si[0] = next;
    }
    if (ne.tok_emb) {
        // token + position embedding of the next position for this clip's row (k_dec_embed's work, one launch
        // and one kernel boundary per token saved): f32 residual row, slab copy, row sums
        __syncthreads();
        const int next = si[0];
        __syncthreads();
        const T* er = (const T*)ne.tok_emb + (long)next * ne.d;
        const float* pr = ne.pos_emb + (long)(pos + 1) * ne.d;
        // the row's mean first (k_dec_embed's two passes): the slab copy and the sums are of the centred row
        float mean = 0.0f;
        if (ne.shift) {
            float s0 = 0.0f;
            for (int c = tid * 4; c < ne.d; c += 1024) {
                const f32x4 v = load4_f32(er + c) + *reinterpret_cast<const f32x4*>(pr + c);
                s0 += (v[0] + v[1]) + (v[2] + v[3]);
            }
            s0 = dpp_wave_sum(s0);
            if ((tid & 63) == 0) sv[8 + (tid >> 6)] = s0;
            __syncthreads();
            mean = ((sv[8] + sv[9]) + (sv[10] + sv[11])) / (float)ne.d;
        }
        float s1 = 0.0f, s2 = 0.0f;
        for (int c = tid * 4; c < ne.d; c += 1024) {
            const f32x4 p4 = *reinterpret_cast<const f32x4*>(pr + c);
            f32x4 v = load4_f32(er + c) + p4;
            *reinterpret_cast<f32x4*>(ne.x + (long)b * ne.d + c) = v;
            v -= mean;
            f32x4 g = {1, 1, 1, 1};
            if (ne.xgamma) g = *reinterpret_cast<const f32x4*>(ne.xgamma + c);
            store4((T*)ne.xslab + slab_idx(b, c, ne.mpad), v[0] * g[0], v[1] * g[1], v[2] * g[2], v[3] * g[3]);
            s1 += (v[0] + v[1]) + (v[2] + v[3]);
            s2 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        }
        s1 = dpp_wave_sum(s1);
        s2 = dpp_wave_sum(s2);
        if ((tid & 63) == 0) { sv[tid >> 6] = s1; sv[4 + (tid >> 6)] = s2; }
        __syncthreads();
        if (tid == 0) {
            ne.stats[2 * b] = (sv[0] + sv[1]) + (sv[2] + sv[3]);
            ne.stats[2 * b + 1] = (sv[4] + sv[5]) + (sv[6] + sv[7]);
            if (ne.shift) ne.shift[b] = mean;
        }
    }
    advance_if_last(ticket, pos_p, gridDim.x);
}

// ---- decoder self-attention, one position ([3P] :417-425, 468-475): one wave per (head, clip) ---
// qkv: [B][3d] (q pre-scaled | k | v) of the current position.
// K cache [B][H][TC][64]  (lane j scores key j: its row is 8 x 16-byte loads, issued before anything else)
// V cache [B][H][TC][64]  (lane e reads column e of every cached row: one 128-byte line per wave load)
// The new k / v are appended (present.{i}.decoder.{key,value}) and used straight from registers.
template <typename T, typename TO = T>   // TO: the attention output as the out-projection's operand (slab layout)
__global__ __launch_bounds__(64) void k_dec_self_attn(const T* __restrict__ qkv, T* __restrict__ kc,
                                                      T* __restrict__ vc, TO* __restrict__ out,
                                                      const int* __restrict__ pos_p, int d, int n_heads, int tc,
                                                      int mpad) {
    constexpr int HD = WH_HEAD_DIM;
    typedef typename FragT<T>::type frag_t;
    __shared__ __attribute__((aligned(16))) float qs[HD];
    __shared__ __attribute__((aligned(16))) float sc[512];
    const int h = blockIdx.x, b = blockIdx.y, lane = threadIdx.x, pos = *pos_p;
    const T* row = qkv + (long)b * 3 * d;
    T* kcb = kc + ((long)b * n_heads + h) * tc * HD;
    T* vcb = vc + ((long)b * n_heads + h) * tc * HD;
    // 1. the cached K rows of this lane's first two keys go in flight before anything else
    frag_t kr[2][HD / 8];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int j = lane + 64 * i;
        if (j < pos) {
#pragma unroll
            for (int e = 0; e < HD / 8; e++) kr[i][e] = load_frag<T>(kcb + (long)j * HD + e * 8);
        }
    }
    // ... and so do the first NPRE * RPI cached V rows, 16 bytes per lane: lane = (row slot, 16-byte chunk of the row), one
    // wave instruction covers RPI whole rows (8 for bf16, 4 for f32) instead of one row as 64 two-byte loads
    constexpr int EPC = 16 / (int)sizeof(T), CPR = HD / EPC, RPI = 64 / CPR, NPRE = 8;
    typedef __attribute__((ext_vector_type(EPC))) T vrow_t;
    const int rslot = lane / CPR, chunk = lane % CPR;
    vrow_t vpre[NPRE];
#pragma unroll
    for (int u = 0; u < NPRE; u++)  // unconditional: rows >= pos are masked below
        vpre[u] = *reinterpret_cast<const vrow_t*>(vcb + (long)min(u * RPI + rslot, tc - 1) * HD + chunk * EPC);
    const vrow_t vcur_v = *reinterpret_cast<const vrow_t*>(row + 2 * d + h * HD + chunk * EPC);
    const T kcur = row[d + h * HD + lane], vcur = row[2 * d + h * HD + lane];
    qs[lane] = cvt_in<T>(row[h * HD + lane]);
    kcb[(long)pos * HD + lane] = kcur;
    vcb[(long)pos * HD + lane] = vcur;
    __syncthreads();
    // 2. scores of the past: lane owns keys lane, lane+64, ... (the first two already loaded)
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int j = lane + 64 * i;
        if (j < pos) {
            float s = 0.0f;
#pragma unroll
            for (int e = 0; e < HD / 8; e++)
#pragma unroll
                for (int u = 0; u < 8; u++) s += qs[e * 8 + u] * (float)kr[i][e][u];
            sc[j] = s;
            mx = fmaxf(mx, s);
        }
    }
    for (int j = lane + 128; j < pos; j += 64) {
        float s = 0.0f;
#pragma unroll
        for (int e = 0; e < HD / 8; e++) {
            const frag_t kk = load_frag<T>(kcb + (long)j * HD + e * 8);
#pragma unroll
            for (int u = 0; u < 8; u++) s += qs[e * 8 + u] * (float)kk[u];
        }
        sc[j] = s;
        mx = fmaxf(mx, s);
    }
    const float scur = dpp_wave_sum(qs[lane] * cvt_in<T>(kcur));
    mx = fmaxf(dpp_wave_max(mx), scur);
    float sum = 0.0f;
    for (int j = lane; j < pos; j += 64) {
        const float p = __expf(sc[j] - mx);
        sc[j] = p;
        sum += p;
    }
    const float pcur = __expf(scur - mx);
    sum = dpp_wave_sum(sum) + pcur;
    __syncthreads();
    // 3. P·V: lane accumulates its chunk's EPC columns over the rows of its slot; the RPI slots are summed at the end
    float o[EPC];
#pragma unroll
    for (int e = 0; e < EPC; e++) o[e] = (rslot == 0) ? pcur * cvt_in<T>(vcur_v[e]) : 0.0f;
#pragma unroll
    for (int u = 0; u < NPRE; u++) {
        const int r = u * RPI + rslot;
        const float pr = (r < pos) ? sc[min(r, 511)] : 0.0f;  // select: a row beyond pos contributes exactly 0
#pragma unroll
        for (int e = 0; e < EPC; e++) o[e] += (r < pos) ? pr * cvt_in<T>(vpre[u][e]) : 0.0f;
    }
    for (int j = NPRE * RPI; j < pos; j += NPRE * RPI) {  // later rows: NPRE wide loads in flight, clamped and masked
        vrow_t v[NPRE];
#pragma unroll
        for (int u = 0; u < NPRE; u++)
            v[u] = *reinterpret_cast<const vrow_t*>(vcb + (long)min(j + u * RPI + rslot, tc - 1) * HD + chunk * EPC);
#pragma unroll
        for (int u = 0; u < NPRE; u++) {
            const int r = j + u * RPI + rslot;
            const float pr = (r < pos) ? sc[min(r, 511)] : 0.0f;
#pragma unroll
            for (int e = 0; e < EPC; e++) o[e] += (r < pos) ? pr * cvt_in<T>(v[u][e]) : 0.0f;
        }
    }
    // sum over the row slots (lanes with equal chunk): lane ^ 8 inside a 16-lane row by DPP (bf16 only: 8 chunks per
    // row), then across the four rows of the wave
#pragma unroll
    for (int e = 0; e < EPC; e++) {
        float t = o[e];
        if (CPR == 8) t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x128, 0xF, 0xF, true));  // row_ror:8
        o[e] = xrow_sum(t);
    }
    if (rslot == 0) {
        const float inv = 1.0f / sum;
        if constexpr (EPC == 4) {   // f32 caches: 4 columns per lane, in the operand type of the consumer
            store4(out + slab_idx(b, h * HD + chunk * EPC, mpad), o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
        } else {
            vrow_t ov;
#pragma unroll
            for (int e = 0; e < EPC; e++) ov[e] = cvt_out<T>(o[e] * inv);
            *reinterpret_cast<vrow_t*>(out + slab_idx(b, h * HD + chunk * EPC, mpad)) = ov;
        }
    }
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
template <typename T, int NCH, int UNROLL, bool NT, typename TO = T>  // NCH = ceil(d*sizeof(T)/16 / 64): 16-B chunks per lane per row; NT: non-temporal K/V loads; TO: type of the slab output
__global__ __launch_bounds__(256) void k_dec_cross_attn(const T* __restrict__ q, const T* __restrict__ ck,
                                                        const T* __restrict__ cv, float* __restrict__ part,
                                                        float* __restrict__ ml, int S, int d, int n_heads,
                                                        int splits, TO* __restrict__ out, int mpad) {
    constexpr int EPC = 16 / (int)sizeof(T);   // elements per 16-B chunk: 8 (bf16) / 4 (f32)
    constexpr int LPH = WH_HEAD_DIM / EPC;     // lanes per head: 8 / 16
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef __attribute__((ext_vector_type(EPC))) T vec_t;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sp = blockIdx.x, b = blockIdx.y;
    const int per = (S + splits - 1) / splits;
    const int ks = sp * per, ke = min(S, ks + per);
    // Key groups of UNROLL consecutive keys are dealt round-robin to the four waves, so at any moment
    // the workgroup reads 4*UNROLL adjacent key rows (one 16 KiB run of K and one of V for bf16 base).
    const int j1 = ke;
    const int chunks = d / EPC;                // 16-B chunks per row

    constexpr bool PACKED = sizeof(T) == 2;  // bf16: packed-pair arithmetic (v_dot2c_f32_bf16, v_perm_b32)
    float qv[NCH][EPC], o[NCH][EPC], mrun[NCH], lrun[NCH];
    wh_u32x4 qd[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const int ch = lane + 64 * c;
        mrun[c] = -INFINITY;
        lrun[c] = 0.0f;
        qd[c] = wh_u32x4{0, 0, 0, 0};
        if (PACKED && ch < chunks) qd[c] = *reinterpret_cast<const wh_u32x4*>(q + (long)b * d + ch * EPC);
#pragma unroll
        for (int u = 0; u < EPC; u++) {
            qv[c][u] = (!PACKED && ch < chunks) ? cvt_in<T>(q[(long)b * d + ch * EPC + u]) : 0.0f;
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
                    // read-once stream: non-temporal, so that the K/V bytes of a launch (hundreds of MB) do not evict the
                    // decode weights and activations from L2 / the Infinity Cache between the surrounding small GEMMs
                    // (only when the whole cross K/V set is larger than the Infinity Cache: a small batch re-reads it from there)
                    if constexpr (NT) {
                        kk[u][c] = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(kb + (long)jj * d + ch * EPC));
                        vv[u][c] = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(vb + (long)jj * d + ch * EPC));
                    } else {
                        kk[u][c] = *reinterpret_cast<const vec_t*>(kb + (long)jj * d + ch * EPC);
                        vv[u][c] = *reinterpret_cast<const vec_t*>(vb + (long)jj * d + ch * EPC);
                    }
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
                    if constexpr (PACKED) {  // 4 x v_dot2c_f32_bf16 on the packed pairs, no conversions
                        t = dot8_bf16(as_u32x4(kk[u][c]), qd[c], t);
                    } else {
#pragma unroll
                        for (int e = 0; e < EPC; e++) t += qv[c][e] * (float)kk[u][c][e];
                    }
                }
                t = dpp_group_sum<LPH>(t);  // every lane of the head's lane group gets the full dot product
                s[u] = (j + u < j1) ? t : -INFINITY;
                mx = fmaxf(mx, s[u]);
            }
            const float scale = __expf(mrun[c] - mx);  // first chunk: exp(-inf) = 0
            float ls = lrun[c] * scale;
#pragma unroll
            for (int e = 0; e < EPC; e++) o[c][e] *= scale;
            if constexpr (PACKED && (UNROLL % 2 == 0)) {
                // P·V two keys at a time: pair the same element of both keys with v_perm_b32, then one
                // v_dot2c_f32_bf16 against the packed (p_u, p_u+1) per output element
#pragma unroll
                for (int u = 0; u < UNROLL; u += 2) {
                    const float p0 = __expf(s[u] - mx), p1 = __expf(s[u + 1] - mx);
                    ls += p0 + p1;
                    const bf16x2 pp = bf16x2{(bf16)p0, (bf16)p1};
                    if (ch < chunks) {
                        const wh_u32x4 va = as_u32x4(vv[u][c]), vb = as_u32x4(vv[u + 1][c]);
                        pv2_bf16(va.x, vb.x, pp, o[c][0], o[c][1]);
                        pv2_bf16(va.y, vb.y, pp, o[c][2], o[c][3]);
                        pv2_bf16(va.z, vb.z, pp, o[c][4], o[c][5]);
                        pv2_bf16(va.w, vb.w, pp, o[c][6], o[c][7]);
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < UNROLL; u++) {
                    const float p = __expf(s[u] - mx);
                    ls += p;
                    if (ch < chunks) {
#pragma unroll
                        for (int e = 0; e < EPC; e++) o[c][e] += p * (float)vv[u][c][e];
                    }
                }
            }
            mrun[c] = mx;
            lrun[c] = ls;
        }
    };
    constexpr int GS = 4 * UNROLL;             // stride between this wave's consecutive key groups
    const int j0 = ks + wave * UNROLL;
    if (j0 < j1) load_set(kA, vA, j0);
    for (int j = j0; j < j1; j += 2 * GS) {
        const bool hasB = j + GS < j1;
        if (hasB) load_set(kB, vB, j + GS);
        compute_set(kA, vA, j);
        if (hasB) {
            if (j + 2 * GS < j1) load_set(kA, vA, j + 2 * GS);
            compute_set(kB, vB, j + GS);
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
    // this key range's partial (unnormalised output, running max, running sum); the consumer GEMM merges
    // the ranges of a clip while it builds its MFMA operand (frag_from_partials), so nothing is exchanged
    // between workgroups here and the kernel ends with its last store
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
        if (out) {  // one key range per clip: this IS the attention output (slab layout, compute dtype)
            store1(out + slab_idx(b, n, mpad), num / den);
            continue;
        }
        pp[n] = num;
        if ((n % WH_HEAD_DIM) == 0) {
            mp[h] = M;
            mp[n_heads + h] = den;
        }
    }
}

// ---- the same for wide models (d_model > 512, bf16): one workgroup = one clip x one key range x one COLUMN GROUP of
// 256 columns (4 heads).  With all heads in one workgroup a d = 1280 row needs 2.5 wave-instructions (three chunks per
// lane, a sixth of the lanes idle in the last) and the register sets allow only two keys in flight per wave: 0.42 of the
// HBM roof at whisper-large-v3 width.  Here a wave-instruction reads the 512-byte segments of TWO key rows (lanes 0-31 the
// even key, 32-63 the odd key of a pair), UNROLL pairs per register set, two sets: the base kernel's 16 KiB in flight per
// wave, every lane busy.  The two parities of a wave keep separate online-softmax states; 4 waves x 2 parities merge
// through LDS.  Output layout unchanged (partials [B][splits][d], {max, sum} [B][splits][H][2], or the slab when splits == 1).
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k_dec_cross_attn_cg(const bf16* __restrict__ q, const bf16* __restrict__ ck,
                                                           const bf16* __restrict__ cv, float* __restrict__ part,
                                                           float* __restrict__ ml, int S, int d, int n_heads,
                                                           int splits, bf16* __restrict__ out, int mpad) {
    constexpr int CGW = 256, HPG = CGW / WH_HEAD_DIM;   // columns / heads per group
    __shared__ __attribute__((aligned(16))) float wm[8][HPG], wl[8][HPG], wo[8][CGW];
    typedef __attribute__((ext_vector_type(8))) bf16 vec_t;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int par = lane >> 5, chunk = lane & 31;
    const int sp = blockIdx.x, b = blockIdx.y, cg = blockIdx.z;
    const int per = (S + splits - 1) / splits;
    const int ks = sp * per, j1 = min(S, ks + per);
    const int col0 = cg * CGW + chunk * 8;
    const wh_u32x4 qd = *reinterpret_cast<const wh_u32x4*>(q + (long)b * d + col0);
    float o[8], mrun = -INFINITY, lrun = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = 0.0f;
    const bf16* kb = ck + (long)b * S * d + col0;
    const bf16* vb = cv + (long)b * S * d + col0;
    vec_t kA[UNROLL], vA[UNROLL], kB[UNROLL], vB[UNROLL];
    auto load_set = [&](vec_t (&kk)[UNROLL], vec_t (&vv)[UNROLL], int j) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const int jj = min(j + 2 * u + par, j1 - 1);   // tail: re-read the last key, masked in compute_set
            if constexpr (NT) {
                kk[u] = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(kb + (long)jj * d));
                vv[u] = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(vb + (long)jj * d));
            } else {
                kk[u] = *reinterpret_cast<const vec_t*>(kb + (long)jj * d);
                vv[u] = *reinterpret_cast<const vec_t*>(vb + (long)jj * d);
            }
        }
    };
    auto compute_set = [&](const vec_t (&kk)[UNROLL], const vec_t (&vv)[UNROLL], int j) {
        float s[UNROLL];
        float mx = mrun;
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            float t = dot8_bf16(as_u32x4(kk[u]), qd, 0.0f);
            t = dpp_group_sum<8>(t);
            s[u] = (j + 2 * u + par < j1) ? t : -INFINITY;
            mx = fmaxf(mx, s[u]);
        }
        const float scale = (mx == -INFINITY) ? 1.0f : __expf(mrun - mx);   // nothing seen yet (this parity's keys all masked): keep zeros
        float ls = lrun * scale;
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] *= scale;
#pragma unroll
        for (int u = 0; u < UNROLL; u += 2) {
            const float p0 = (s[u] == -INFINITY) ? 0.0f : __expf(s[u] - mx), p1 = (s[u + 1] == -INFINITY) ? 0.0f : __expf(s[u + 1] - mx);
            ls += p0 + p1;
            const bf16x2 pp = bf16x2{(bf16)p0, (bf16)p1};
            const wh_u32x4 va = as_u32x4(vv[u]), vb2 = as_u32x4(vv[u + 1]);
            pv2_bf16(va.x, vb2.x, pp, o[0], o[1]);
            pv2_bf16(va.y, vb2.y, pp, o[2], o[3]);
            pv2_bf16(va.z, vb2.z, pp, o[4], o[5]);
            pv2_bf16(va.w, vb2.w, pp, o[6], o[7]);
        }
        mrun = mx;
        lrun = ls;
    };
    constexpr int GS = 4 * UNROLL * 2;   // keys per round of the four waves
    const int j0 = ks + wave * UNROLL * 2;
    if (j0 < j1) load_set(kA, vA, j0);
    for (int j = j0; j < j1; j += 2 * GS) {
        const bool hasB = j + GS < j1;
        if (hasB) load_set(kB, vB, j + GS);
        compute_set(kA, vA, j);
        if (hasB) {
            if (j + 2 * GS < j1) load_set(kA, vA, j + 2 * GS);
            compute_set(kB, vB, j + GS);
        }
    }
    // merge the 4 waves x 2 parities of this key range (LDS)
    const int st = wave * 2 + par;
    if ((chunk & 7) == 0) { wm[st][chunk >> 3] = mrun; wl[st][chunk >> 3] = lrun; }
#pragma unroll
    for (int e = 0; e < 8; e++) wo[st][chunk * 8 + e] = o[e];
    __syncthreads();
    float* pp = part + ((long)b * splits + sp) * d + cg * CGW;
    float* mp = ml + ((long)b * splits + sp) * n_heads * 2;
    {
        const int n = tid, hl = n / WH_HEAD_DIM, h = cg * HPG + hl;   // 256 threads = 256 columns of the group
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < 8; w++) M = fmaxf(M, wm[w][hl]);
        float num = 0.0f, den = 0.0f;
#pragma unroll
        for (int w = 0; w < 8; w++) {
            const float mw = wm[w][hl];
            const float sc = (mw == -INFINITY) ? 0.0f : __expf(mw - M);   // a stream may own no keys
            num += sc * wo[w][n];
            den += sc * wl[w][hl];
        }
        if (out) {
            out[slab_idx(b, cg * CGW + n, mpad)] = (bf16)(num / den);
        } else {
            pp[n] = num;
            if ((n % WH_HEAD_DIM) == 0) { mp[h] = M; mp[n_heads + h] = den; }
        }
    }
}

// raise a kernel's dynamic-LDS limit once per (device, kernel, size)
template <typename K>
void set_max_smem(K kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return;
    wh_ensure_dyn_lds((const void*)kernel, bytes);
}

template <typename T, typename TO, int NW, typename TW>
void launch_dec_gemm_mt(hipStream_t s, const SkinnyArgs& a) {
    const int n_tiles = (a.N + 15) / 16;
    // rows per workgroup: 64, or 32 when K is deep (every workgroup pulls its rows of X through L2; halving
    // them costs a second read of the 16-column weight tile but doubles the workgroups sharing the load)
    // measured (tools/microbench.cpp, M = 64): N = 512 → 16 rows best (2.6 vs 3.7 us), N = 1536 / 2048 → 32 rows
    // (2.95 vs 3.5 us), K = 2048 → 16 rows (4.6 vs 7.5 us): aim for >= 128 workgroups
    int mt_cap = (NW == 8 || n_tiles <= 48) ? 1 : 2;
    // wide models (K >= 1024: weights stream from HBM, profiles/r02_dec_gemm_wide_m32.txt): 32-row groups only for the 8-way
    // split of a wide N (QKV: 6.9 vs 7.6 us); everything else 16 rows (fc1 9.9 vs 10.5 us, N = 1280 4.6 vs 5.3 us)
    if (a.K >= 1024) mt_cap = (NW == 8 && n_tiles > 96) ? 2 : 1;
    if (wh_dbg_mt > 0) mt_cap = wh_dbg_mt;  // microbench override
    // batches beyond 64 rows (measured at 256 clips): 64-row groups for the 4-way K split — the row groups already give
    // hundreds of workgroups, and each weight fragment then feeds four MFMAs instead of one or two (-0.7 % step time)
    if (wh_dbg_mt <= 0 && a.M > 64 && NW == 4) mt_cap = n_tiles <= 32 ? 2 : 4;   // (narrow N at 256 rows: 4.1 vs 4.8 us)
    if (wh_dbg_mt <= 0 && a.M > 64 && NW == 8 && !a.xpart) mt_cap = 2;  // fc2: 32-row groups (-0.5 %; 64 is slower again)
    // hundreds of rows (one key range per clip, so no merged X): NT column tiles per workgroup while >= 256 workgroups remain
    // (k_dec_gemm_wide; bit-identical results, so following the batch is allowed)
    if constexpr (!__is_same(T, float) && !__is_same(T, xf32))   // (f32 rows keep k_dec_gemm: the exact-f32 mode and the small-context form of the split-fp16 mode)
    if (a.M > 64 && !a.xpart && wh_dbg_mt <= 0 && a.K % (NW * 4 * WTraits<T, TW>::KW) == 0) {
        constexpr int WMT = 4;
        int wide = wh_dbg_wide;
        if (const char* e = getenv("WH_DEC_WIDE")) wide = atoi(e);   // A/B switch for the parity test (0: k_dec_gemm only)
        const int rg = (a.M + 16 * WMT - 1) / (16 * WMT);
        // measured (tools/dec_gemm_sweep.cpp, profiles/r02_dec_gemm_sweep.txt): four column tiles while >= 512 workgroups
        // remain (fc1 at 1024 rows: 11.7 vs 13.2 us), else two while >= 128 remain (1280 x 1280 at 256 rows: 8.5 vs 11.3 us;
        // 512 x 512 at 256 rows would leave 64: 5.7 vs 4.1 us), else k_dec_gemm.  Eight waves x 16 tiles of partial sums
        // would take 128 KB of LDS: two tiles at most there.
        int nt = 1;
        if (NW == 4 && n_tiles % 4 == 0 && (n_tiles / 4) * rg * a.zn >= 512) nt = 4;
        else if (n_tiles % 2 == 0 && (n_tiles / 2) * rg * a.zn >= 128) nt = 2;
        if (wide == 2 || (wide == 4 && NW == 4)) nt = wide;
        if (wide != 0 && nt > 1 && n_tiles % nt == 0) {
            const size_t smw = (size_t)NW * WMT * nt * 64 * 16 + (size_t)4 * WMT * 16 * 2 * 4;
            dim3 gw(n_tiles / nt, rg, a.zn);
            if (nt == 4) { set_max_smem(k_dec_gemm_wide<T, TO, WMT, 4, NW, TW>, smw); hipLaunchKernelGGL((k_dec_gemm_wide<T, TO, WMT, 4, NW, TW>), gw, dim3(NW * 64), smw, s, a); }
            else { set_max_smem(k_dec_gemm_wide<T, TO, WMT, 2, NW, TW>, smw); hipLaunchKernelGGL((k_dec_gemm_wide<T, TO, WMT, 2, NW, TW>), gw, dim3(NW * 64), smw, s, a); }
            return;
        }
    }
    const int mt = std::min(mt_cap, (a.M + 15) / 16);
    const size_t sm = (size_t)NW * mt * 64 * 16 + (size_t)4 * mt * 16 * 2 * 4;
    dim3 grid(n_tiles, (a.M + 16 * mt - 1) / (16 * mt), a.zn);
    if (a.xpart) {  // X merged from attention partials: 16-row groups only (the merge is per-lane work)
        dim3 g1(n_tiles, (a.M + 15) / 16);
        const size_t sm1 = (size_t)NW * 64 * 16 + 4 * 16 * 2 * 4 + (size_t)16 * a.K * sizeof(T);  // + merged X tile
        if (a.x_splits <= 4) { set_max_smem(k_dec_gemm<T, TO, 1, NW, 4, TW>, sm1); hipLaunchKernelGGL((k_dec_gemm<T, TO, 1, NW, 4, TW>), g1, dim3(NW * 64), sm1, s, a); }
        else if (a.x_splits <= 8) { set_max_smem(k_dec_gemm<T, TO, 1, NW, 8, TW>, sm1); hipLaunchKernelGGL((k_dec_gemm<T, TO, 1, NW, 8, TW>), g1, dim3(NW * 64), sm1, s, a); }
        else if (a.x_splits <= 16) { set_max_smem(k_dec_gemm<T, TO, 1, NW, 16, TW>, sm1); hipLaunchKernelGGL((k_dec_gemm<T, TO, 1, NW, 16, TW>), g1, dim3(NW * 64), sm1, s, a); }
        else { set_max_smem(k_dec_gemm<T, TO, 1, NW, 32, TW>, sm1); hipLaunchKernelGGL((k_dec_gemm<T, TO, 1, NW, 32, TW>), g1, dim3(NW * 64), sm1, s, a); }
        return;
    }
    switch (mt) {
        case 1: hipLaunchKernelGGL((k_dec_gemm<T, TO, 1, NW, 0, TW>), grid, dim3(NW * 64), sm, s, a); break;
        case 2: hipLaunchKernelGGL((k_dec_gemm<T, TO, 2, NW, 0, TW>), grid, dim3(NW * 64), sm, s, a); break;
        case 3: hipLaunchKernelGGL((k_dec_gemm<T, TO, 3, NW, 0, TW>), grid, dim3(NW * 64), sm, s, a); break;
        default: hipLaunchKernelGGL((k_dec_gemm<T, TO, 4, NW, 0, TW>), grid, dim3(NW * 64), sm, s, a); break;
    }
}

// K split over the waves of a workgroup: 8 ways when K is deep, when X is merged from attention partials, or when the
// launch would otherwise put fewer than ~128 workgroups on the chip (small batch x narrow N: more waves per
// workgroup keep more weight bytes in flight per CU); else 4.  Needs K % 256 == 0.
static inline bool dec_gemm_8way(const SkinnyArgs& a) {
    // (K >= 1024, N <= 4096: eight waves per workgroup keep twice the weight bytes in flight per CU — 4.6 vs 5.8 us for
    // N = K = 1280 at 32 rows, 5.7 vs 6.5 us for the QKV projection; the widest N has enough workgroups without it.
    // The choice never depends on the batch: a batch must not change the summation order of a row)
    return (a.K >= 2048 || a.xpart || (a.K >= 1024 && a.N <= 4096)) && a.K % 256 == 0;
}

template <typename T, typename TO, typename TW>
void launch_dec_gemm_split(hipStream_t s, const SkinnyArgs& a) {
    // K is split over the waves of a workgroup: 8 ways when it is deep, else 4 (K % 128 == 0 always
    // holds: d_model and ffn are multiples of 128, checked at model load)
    // (X from attention partials: 8 ways too — fewer fragments to merge per lane)
    const bool eight = wh_dbg_nw ? (wh_dbg_nw == 8 && a.K % 256 == 0) : dec_gemm_8way(a);
    if (eight) launch_dec_gemm_mt<T, TO, 8, TW>(s, a);
    else launch_dec_gemm_mt<T, TO, 4, TW>(s, a);
}

}  // namespace

void wh_launch_dec_gemm(hipStream_t s, int prec, bool out_f32, const SkinnyArgs& a) {
    if (prec == WH_PREC_F32) launch_dec_gemm_split<float, float, float>(s, a);
    else if (prec == WH_PREC_F16X3) launch_dec_gemm_split<h2, float, h2>(s, a);   // operands as fp16 limbs (weights, slabs), f32 row-major results
    else if (prec == WH_PREC_FP8 && a.wscale) {  // e4m3 weight codes, bf16 activations
        // 16 codes per lane per load when every wave's K share is a multiple of 64, else 8
        const int nw = dec_gemm_8way(a) ? 8 : 4;
        const bool wide = (a.K / nw) % 64 == 0;
        if (wide) { if (out_f32) launch_dec_gemm_split<bf16, float, fp8x16_t>(s, a); else launch_dec_gemm_split<bf16, bf16, fp8x16_t>(s, a); }
        else { if (out_f32) launch_dec_gemm_split<bf16, float, fp8x8_t>(s, a); else launch_dec_gemm_split<bf16, bf16, fp8x8_t>(s, a); }
    } else if (out_f32) launch_dec_gemm_split<bf16, float, bf16>(s, a);
    else launch_dec_gemm_split<bf16, bf16, bf16>(s, a);
}

void wh_launch_dec_embed(hipStream_t s, int prec, const void* tok_emb, const float* pos_emb, const int* feed, int feed_ld,
                         const int* pos_p, float* x, void* xslab, float* stats, int rows, int d, int mpad, const float* xgamma, float* shift) {
    dim3 grid((rows + 3) / 4);
    if (prec == WH_PREC_F16X3)
        hipLaunchKernelGGL(k_dec_embed<h2>, grid, dim3(256), 0, s, (const h2*)tok_emb, pos_emb, feed, feed_ld, pos_p, x, (h2*)xslab, stats, rows, d, mpad, xgamma, shift);
    else if (prec == WH_PREC_F32)
        hipLaunchKernelGGL(k_dec_embed<float>, grid, dim3(256), 0, s, (const float*)tok_emb, pos_emb, feed, feed_ld, pos_p, x, (float*)xslab, stats, rows, d, mpad, xgamma, shift);
    else
        hipLaunchKernelGGL(k_dec_embed<bf16>, grid, dim3(256), 0, s, (const bf16*)tok_emb, pos_emb, feed, feed_ld, pos_p, x, (bf16*)xslab, stats, rows, d, mpad, xgamma, shift);
}

template <typename T>
void launch_lm_head_t(hipStream_t s, const SkinnyArgs& a, int* n_parts_out = nullptr) {
    const int n_tiles = (a.N + 15) / 16;
    int mt = std::min(wh_dbg_lm_mt, (a.M + 15) / 16);
    auto lds = [&](int t) { return (size_t)t * 16 * a.K * sizeof(T) + (size_t)t * 16 * 2 * 4 * 5; };  // X tile + LN stats (+ 4 quarter sums)
    while (mt > 1 && lds(mt) > 150 * 1024) mt--;
    // (128-row groups — two instead of four at 256 clips, half the passes over the 53 MB embedding but one workgroup per CU —
    // measured no faster: 189.5 vs 188.0 ms per 256-clip step)
    if (mt > 4) mt = 4;
    const size_t sm = lds(mt);
    const int per_cu = std::max<int>(1, (int)(150 * 1024 / sm));
    dim3 grid(std::min((n_tiles + 3) / 4, 256 * std::min(per_cu, wh_dbg_lm_blocks_per_cu) / ((a.M + 16 * mt - 1) / (16 * mt))), (a.M + 16 * mt - 1) / (16 * mt));
    if (n_parts_out) { *n_parts_out = (int)grid.x * 4; return; }  // query only: partials per row = waves per row group
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
    else if (prec == WH_PREC_F16X3) { if (wh_lm_head_tile_x3_applicable(a)) wh_launch_lm_head_tile_x3(s, a); else launch_lm_head_t<h2>(s, a); }
    else if (wh_lm_head_tile_applicable(a)) wh_launch_lm_head_tile(s, a);   // hundreds of rows: 256 x 256 tiles (wh_gemm8.hip), same logits
    else launch_lm_head_t<bf16>(s, a);
}

// number of argmax partials per row the LM head writes for this shape (its layout is [part][x_mpad])
int wh_lm_head_parts(int prec, const SkinnyArgs& a) {
    int n = 0;
    if (prec == WH_PREC_F16X3 && wh_lm_head_tile_x3_applicable(a)) n = wh_lm_head_tile_x3_parts(a);
    else if (prec == WH_PREC_F32 || prec == WH_PREC_F16X3) launch_lm_head_t<float>(nullptr, a, &n);
    else if (wh_lm_head_tile_applicable(a)) n = wh_lm_head_tile_parts(a);
    else launch_lm_head_t<bf16>(nullptr, a, &n);
    return n;
}

void wh_launch_argmax_finish(hipStream_t s, int prec, const float* part_val, const int* part_idx, int n_parts, int mpad, int* pos_p,
                             int* ticket, const DecodeState& st, int B, const NextEmbed& ne) {
    if (prec == WH_PREC_F16X3) hipLaunchKernelGGL(k_argmax_finish<h2>, dim3(B), dim3(256), 0, s, part_val, part_idx, n_parts, mpad, pos_p, ticket, st, ne);
    else if (prec == WH_PREC_F32) hipLaunchKernelGGL(k_argmax_finish<float>, dim3(B), dim3(256), 0, s, part_val, part_idx, n_parts, mpad, pos_p, ticket, st, ne);
    else hipLaunchKernelGGL(k_argmax_finish<bf16>, dim3(B), dim3(256), 0, s, part_val, part_idx, n_parts, mpad, pos_p, ticket, st, ne);
}

void wh_launch_dec_self_attn(hipStream_t s, int prec, const void* qkv, void* kc, void* vc, void* out, const int* pos_p,
                             int d, int n_heads, int tc, int B, int mpad) {
    dim3 grid(n_heads, B);
    if (prec == WH_PREC_F16X3)   // f32 q/k/v and caches (no matrix-core work here), the output as the out-projection's fp16-limb operand
        hipLaunchKernelGGL((k_dec_self_attn<float, h2>), grid, dim3(64), 0, s, (const float*)qkv, (float*)kc, (float*)vc, (h2*)out, pos_p, d, n_heads, tc, mpad);
    else if (prec == WH_PREC_F32)
        hipLaunchKernelGGL(k_dec_self_attn<float>, grid, dim3(64), 0, s, (const float*)qkv, (float*)kc, (float*)vc, (float*)out, pos_p, d, n_heads, tc, mpad);
    else
        hipLaunchKernelGGL(k_dec_self_attn<bf16>, grid, dim3(64), 0, s, (const bf16*)qkv, (bf16*)kc, (bf16*)vc, (bf16*)out, pos_p, d, n_heads, tc, mpad);
}

// Occupancy cap of the cross-attention stream.  With more than two workgroups resident per CU (1024 clips = 4 per CU) every one
// keeps 32 KB of K/V rows in flight and the memory system queues far more requests than the ~7 MB its latency x bandwidth
// needs; measured at 1024 clips (profiles/r03_cross_attn_occupancy.txt): 474.5 us per launch with 4 resident workgroups per
// CU, 493.4 with 3, 464.2 with 2, 477.0 with 1.  The cap is an LDS reservation: a launch with more than 2 x 256 workgroups
// asks for 72 KB of dynamic LDS per workgroup (the kernels use only their own first bytes), so two fit a CU's 160 KB and a
// third does not.  WH_CROSS_WGS_PER_CU=0 removes the cap, 1 reserves for one workgroup per CU.  Results do not depend on it.
size_t wh_cross_lds_reserve(long total_wgs, size_t own_bytes) {
    static const int cap = [] { const char* e = getenv("WH_CROSS_WGS_PER_CU"); return e ? atoi(e) : 2; }();
    if (cap <= 0 || cap > 2 || total_wgs <= (long)cap * 256) return own_bytes;
    const size_t target = cap == 1 ? (size_t)150 * 1024 : (size_t)72 * 1024;
    return own_bytes > target ? own_bytes : target;
}

void wh_launch_dec_cross_attn(hipStream_t s, int prec, const void* q, const void* ck, const void* cv, float* part,
                              float* ml, int S, int d, int n_heads, int splits, int B, void* out, int mpad, bool stream_nt) {
    dim3 grid(splits, B);
    const bool f32_layout = prec == WH_PREC_F32 || prec == WH_PREC_F16X3;   // (no matrix-core work in this kernel: the f32 form serves both)
    const long wgs = (long)splits * B * ((!f32_layout && d > 512 && d % 256 == 0) ? d / 256 : 1);
    const size_t sm = wh_cross_lds_reserve(wgs, sizeof(float) * ((size_t)8 * n_heads + 4 * (size_t)d));
#define WH_CA2(T_, N_, U_, NT_, TO_) do { set_max_smem(k_dec_cross_attn<T_, N_, U_, NT_, TO_>, sm);                                                 \
                                     hipLaunchKernelGGL((k_dec_cross_attn<T_, N_, U_, NT_, TO_>), grid, dim3(256), sm, s, (const T_*)q, (const T_*)ck, \
                                             (const T_*)cv, part, ml, S, d, n_heads, splits, (TO_*)(splits == 1 ? out : nullptr), mpad); } while (0)
#define WH_CA1(T_, N_, U_, NT_) WH_CA2(T_, N_, U_, NT_, T_)
    static const int nt_env = [] { const char* e = getenv("WH_CROSS_NT"); return e ? atoi(e) : -1; }();   // (A/B runs: force the non-temporal K/V loads off / on)
    if (nt_env >= 0) stream_nt = nt_env != 0;
#define WH_CA(T_, N_, U_) do { if (stream_nt) WH_CA1(T_, N_, U_, true); else WH_CA1(T_, N_, U_, false); } while (0)
    if (prec == WH_PREC_F16X3) {   // the f32 kernel (no matrix-core work); one key range per clip writes the out-projection's fp16-limb operand
#define WH_CAX(N_, U_) do { if (stream_nt) WH_CA2(float, N_, U_, true, h2); else WH_CA2(float, N_, U_, false, h2); } while (0)
        const int nch = (d / 4 + 63) / 64;
        if (nch == 1) WH_CAX(1, 4);
        else if (nch == 2) WH_CAX(2, 2);
        else WH_CAX(5, 1);
#undef WH_CAX
    } else if (f32_layout) {
        const int nch = (d / 4 + 63) / 64;  // f32: 4 elements per chunk
        if (nch == 1) WH_CA(float, 1, 4);
        else if (nch == 2) WH_CA(float, 2, 2);
        else WH_CA(float, 5, 1);
    } else {
        const int nch = (d / 8 + 63) / 64;
        static const int unroll_env = [] { const char* e = getenv("WH_CROSS_UNROLL"); return e ? atoi(e) : 0; }();   // (A/B runs)
        if (nch == 1) {
            const int un = unroll_env ? unroll_env : wh_dbg_cross_unroll;
            if (un == 8) WH_CA(bf16, 1, 8);
            else if (un == 2) WH_CA(bf16, 1, 2);
            else WH_CA(bf16, 1, 4);
        }
        else if (d % 256 == 0 && getenv("WH_CROSS_ALLHEADS") == nullptr) {   // wide models: one workgroup per 256-column group
            dim3 g3(splits, B, d / 256);
            bf16* o = (bf16*)(splits == 1 ? out : nullptr);
            const size_t smcg = wh_cross_lds_reserve(wgs, 12 * 1024) - 12 * 1024;   // (the kernel's own ~10 KB are static: only the reserve is dynamic)
            if (stream_nt) { set_max_smem(k_dec_cross_attn_cg<4, true>, smcg); hipLaunchKernelGGL((k_dec_cross_attn_cg<4, true>), g3, dim3(256), smcg, s, (const bf16*)q, (const bf16*)ck, (const bf16*)cv, part, ml, S, d, n_heads, splits, o, mpad); }
            else { set_max_smem(k_dec_cross_attn_cg<4, false>, smcg); hipLaunchKernelGGL((k_dec_cross_attn_cg<4, false>), g3, dim3(256), smcg, s, (const bf16*)q, (const bf16*)ck, (const bf16*)cv, part, ml, S, d, n_heads, splits, o, mpad); }
        } else WH_CA(bf16, 3, 2);
    }
#undef WH_CA
#undef WH_CA1
#undef WH_CA2
}
