// wh_decode.hip — per-token decoder kernels for gfx950: the body of the with-past loop of
// greedy_decode_with_past (reference src/main.rs:793-826) for a BATCH of independent clips.
//
// The reference runs decoder_with_past_model.onnx once per token per clip through ORT's IoBinding
// and re-binds 4*Ld past tensors every step (src/main.rs:798-812).  Here the KV past is an
// on-device cache with an append position, the token loop never leaves the device (masked argmax,
// EOT bookkeeping and the next input token are computed by kernels), and up to 64 clips advance
// together so each weight byte streamed from HBM serves all of them.
//
// Every kernel reads the current position from device memory (*pos), never from a kernel argument,
// so one captured hipGraph of a step can be replayed for every step.
//
// [3P] decoder definition: modeling_whisper.py WhisperDecoderLayer.forward (:466-500), learned
// positions offset by the past length (:208-212), final LN (:790), tied LM head, no bias (:965,970).
#include "wh_common.h"
#include "wh_kernels.h"

namespace {

// x[b][:] = tok_emb[feed[b][pos]][:] + pos_emb[pos][:]            ([3P] :737, :757-766)
template <typename T>
__global__ void k_dec_embed(const T* __restrict__ tok_emb, const float* __restrict__ pos_emb,
                            const int* __restrict__ feed, int feed_ld, const int* __restrict__ pos_p,
                            float* __restrict__ x, int d) {
    const int b = blockIdx.x, pos = *pos_p;
    const long tok = feed[b * feed_ld + pos];
    for (int i = threadIdx.x; i < d; i += blockDim.x)
        x[(long)b * d + i] = cvt_in<T>(tok_emb[tok * d + i]) + pos_emb[(long)pos * d + i];
}

// ---- skinny GEMM: C[m][n] = act(sum_k X[m][k] W[n][k] + bias[n]) (+ R[m][n]),  M <= 64 ---------
// Weight-streaming: every weight element is read once per launch straight into MFMA fragments
// (no LDS round trip: nothing is shared between waves), activations come from L2.
//   SPLITK=4: one 16-column tile per workgroup, the four waves split K and reduce through LDS
//             (layers with N = d .. ffn: enough workgroups to cover the chip).
//   SPLITK=1: four 16-column tiles per workgroup, one per wave (LM head, N = vocab).
//   MODE 0  : normal epilogue.   MODE 1: LM-head epilogue — optional logits store + per-tile
//             masked argmax partials (reference argmax_last_dim_raw, src/main.rs:709-735).
template <typename T, typename TO, int MT, int SPLITK, int MODE>
__global__ __launch_bounds__(256) void k_skinny(SkinnyArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fl = lane & 15, fg = lane >> 4;
    const int tile = (SPLITK == 4) ? blockIdx.x : blockIdx.x * 4 + wave;
    const int n_tiles = (a.N + 15) >> 4;
    const bool tile_ok = tile < n_tiles;
    const int n0 = tile * 16;
    const T* W = (const T*)a.W;
    const T* X = (const T*)a.X;
    int nrow = n0 + fl;
    if (nrow > a.N - 1) nrow = a.N - 1;
    const T* wp = W + (long)nrow * a.K + fg * 8;
    const T* xp[MT];
#pragma unroll
    for (int t = 0; t < MT; t++) {
        int m = t * 16 + fl;
        if (m > a.M - 1) m = a.M - 1;
        xp[t] = X + (long)m * a.ldx + fg * 8;
    }
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; t++) acc[t] = f32x4{0, 0, 0, 0};
    const int kspan = a.K / SPLITK;
    const int kb = (SPLITK == 4) ? wave * kspan : 0;
    if (tile_ok) {
#pragma unroll 4
        for (int k = kb; k < kb + kspan; k += 32) {
            typename FragT<T>::type wf = load_frag<T>(wp + k);
#pragma unroll
            for (int t = 0; t < MT; t++) {
                typename FragT<T>::type xf = load_frag<T>(xp[t] + k);
                mma16(acc[t], wf, xf);  // D rows = n (4*fg + r), col = m (fl)
            }
        }
    }
    if (SPLITK == 4) {
        __shared__ f32x4 red[3][MT][64];
        if (wave > 0) {
#pragma unroll
            for (int t = 0; t < MT; t++) red[wave - 1][t][lane] = acc[t];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int t = 0; t < MT; t++)
#pragma unroll
            for (int w = 0; w < 3; w++) {
                f32x4 o = red[w][t][lane];
                acc[t][0] += o[0]; acc[t][1] += o[1]; acc[t][2] += o[2]; acc[t][3] += o[3];
            }
    }
    if (!tile_ok) return;
    const int n = n0 + 4 * fg;
    if (MODE == 0) {
        if (n >= a.N) return;
        f32x4 bias = {0, 0, 0, 0};
        if (a.bias) bias = *reinterpret_cast<const f32x4*>(a.bias + n);
#pragma unroll
        for (int t = 0; t < MT; t++) {
            const int m = t * 16 + fl;
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
    } else {
        const int pos = *a.pos_p;
        const int gen = pos - (a.n_prompt - 1);  // index of the token this row generates
        const unsigned* mask = (gen == 0) ? a.mask_first : a.mask_base;
#pragma unroll
        for (int t = 0; t < MT; t++) {
            const int m = t * 16 + fl;
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
                                                       const int* __restrict__ part_idx, int n_tiles,
                                                       const int* __restrict__ pos_p, DecodeState st) {
    __shared__ float sv[256];
    __shared__ int si[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < n_tiles; i += 256) {
        float v = part_val[(long)b * n_tiles + i];
        int ix = part_idx[(long)b * n_tiles + i];
        if (v > bv || (v == bv && ix < bi)) { bv = v; bi = ix; }
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
        const int pos = *pos_p;
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
}

__global__ void k_step_advance(int* pos_p) { *pos_p += 1; }

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
    float o = pcur * cvt_in<T>(vcur);
    for (int j = 0; j < pos; j++) o += sc[j] * cvt_in<T>(vcb[(long)j * HD + lane]);
    out[(long)b * d + h * HD + lane] = cvt_out<T>(o / sum);
}

// ---- decoder cross-attention, one position ([3P] :433-440, 478-491) -----------------------------
// The HBM-bound kernel of batched decode: per clip and layer it streams S*d K and S*d V elements
// (18.4 MB per clip per step for whisper-base in bf16) and nothing else of note.
// One workgroup = one clip x one contiguous key range, ALL heads: every key row of K (and V) is
// one contiguous d-element line, read once with 16-byte lane accesses, in order.
//   ck/cv: [B][S][d]  (head h at columns h*64..h*64+63)
//   q    : [B][d]  pre-scaled
//   part : [B][splits][d] unnormalised outputs, ml: [B][splits][H][2] (max, sum)
template <typename T, int NCH>  // NCH = ceil(d*sizeof(T)/16 / 64): 16-B chunks per lane per row
__global__ __launch_bounds__(256) void k_dec_cross_attn(const T* __restrict__ q, const T* __restrict__ ck,
                                                        const T* __restrict__ cv, float* __restrict__ part,
                                                        float* __restrict__ ml, int S, int d, int n_heads,
                                                        int splits) {
    constexpr int EPC = 16 / (int)sizeof(T);   // elements per 16-B chunk: 8 (bf16) / 4 (f32)
    constexpr int LPH = WH_HEAD_DIM / EPC;     // lanes per head: 8 / 16
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sp = blockIdx.x, b = blockIdx.y;
    const int per = (S + splits - 1) / splits;
    const int ks = sp * per, ke = min(S, ks + per), nk = ke - ks;
    const int chunks = d / EPC;                 // 16-B chunks per row
    float* sc = smem;                           // [n_heads][per]
    float* red = smem + n_heads * per;          // [4][d] cross-wave reduction of the output
    float* hm = red + 4 * d;                    // [n_heads] max, [n_heads] sum

    typedef typename FragT<T>::type frag_t;
    // q chunk(s) owned by this lane
    float qv[NCH][EPC];
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const int ch = lane + 64 * c;
#pragma unroll
        for (int u = 0; u < EPC; u++) qv[c][u] = (ch < chunks) ? cvt_in<T>(q[(long)b * d + ch * EPC + u]) : 0.0f;
    }
    const T* kb = ck + ((long)b * S + ks) * d;
    const T* vb = cv + ((long)b * S + ks) * d;
    typedef __attribute__((ext_vector_type(EPC))) T vec_t;
    // pass 1: scores.  wave w takes keys w, w+4, ...; lane owns chunk(s) of the row
    for (int j = wave; j < nk; j += 4) {
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const int ch = lane + 64 * c;
            float s = 0.0f;
            if (ch < chunks) {
                vec_t kk = *reinterpret_cast<const vec_t*>(kb + (long)j * d + ch * EPC);
#pragma unroll
                for (int u = 0; u < EPC; u++) s += qv[c][u] * (float)kk[u];
            }
#pragma unroll
            for (int off = 1; off < LPH; off <<= 1) s += __shfl_xor(s, off);
            if (ch < chunks && (lane % LPH) == 0) sc[(ch / LPH) * per + j] = s;
        }
    }
    __syncthreads();
    // per-head max and sum over this key range (wave w handles heads w, w+4, ...)
    for (int h = wave; h < n_heads; h += 4) {
        float mx = -INFINITY;
        for (int j = lane; j < nk; j += 64) mx = fmaxf(mx, sc[h * per + j]);
        mx = wave_max(mx);
        float sum = 0.0f;
        for (int j = lane; j < nk; j += 64) {
            float p = __expf(sc[h * per + j] - mx);
            sc[h * per + j] = p;
            sum += p;
        }
        sum = wave_sum(sum);
        if (lane == 0) { hm[h] = mx; hm[n_heads + h] = sum; }
    }
    __syncthreads();
    // pass 2: P·V
    float o[NCH][EPC];
#pragma unroll
    for (int c = 0; c < NCH; c++)
#pragma unroll
        for (int u = 0; u < EPC; u++) o[c][u] = 0.0f;
    for (int j = wave; j < nk; j += 4) {
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const int ch = lane + 64 * c;
            if (ch < chunks) {
                const float p = sc[(ch / LPH) * per + j];
                vec_t vv = *reinterpret_cast<const vec_t*>(vb + (long)j * d + ch * EPC);
#pragma unroll
                for (int u = 0; u < EPC; u++) o[c][u] += p * (float)vv[u];
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const int ch = lane + 64 * c;
        if (ch < chunks)
#pragma unroll
            for (int u = 0; u < EPC; u++) red[wave * d + ch * EPC + u] = o[c][u];
    }
    __syncthreads();
    float* pp = part + ((long)b * splits + sp) * d;
    for (int i = tid; i < d; i += 256) pp[i] = red[i] + red[d + i] + red[2 * d + i] + red[3 * d + i];
    float* mp = ml + ((long)b * splits + sp) * n_heads * 2;
    for (int i = tid; i < 2 * n_heads; i += 256) mp[i] = hm[i];
}

// merge the key-range partials: out[b][n] = sum_s e^{m_s-M} o_s[n] / sum_s e^{m_s-M} l_s
template <typename T>
__global__ void k_cross_combine(const float* __restrict__ part, const float* __restrict__ ml, T* __restrict__ out,
                                int d, int n_heads, int splits) {
    const int b = blockIdx.x;
    for (int n = threadIdx.x; n < d; n += blockDim.x) {
        const int h = n / WH_HEAD_DIM;
        float M = -INFINITY;
        for (int s = 0; s < splits; s++) M = fmaxf(M, ml[((long)b * splits + s) * n_heads * 2 + h]);
        float num = 0.0f, den = 0.0f;
        for (int s = 0; s < splits; s++) {
            const float* mp = ml + ((long)b * splits + s) * n_heads * 2;
            const float w = __expf(mp[h] - M);
            num += w * part[((long)b * splits + s) * d + n];
            den += w * mp[n_heads + h];
        }
        out[(long)b * d + n] = cvt_out<T>(num / den);
    }
}

template <typename T, typename TO, int SPLITK, int MODE>
void launch_skinny_mt(hipStream_t s, const SkinnyArgs& a) {
    const int n_tiles = (a.N + 15) / 16;
    dim3 grid(SPLITK == 4 ? n_tiles : (n_tiles + 3) / 4);
    const int mt = (a.M + 15) / 16;
    switch (mt) {
        case 1: hipLaunchKernelGGL((k_skinny<T, TO, 1, SPLITK, MODE>), grid, dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL((k_skinny<T, TO, 2, SPLITK, MODE>), grid, dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL((k_skinny<T, TO, 3, SPLITK, MODE>), grid, dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((k_skinny<T, TO, 4, SPLITK, MODE>), grid, dim3(256), 0, s, a); break;
    }
}

}  // namespace

void wh_launch_dec_embed(hipStream_t s, int prec, const void* tok_emb, const float* pos_emb, const int* feed,
                         int feed_ld, const int* pos_p, float* x, int d, int B) {
    if (prec == WH_PREC_F32)
        hipLaunchKernelGGL(k_dec_embed<float>, dim3(B), dim3(256), 0, s, (const float*)tok_emb, pos_emb, feed, feed_ld, pos_p, x, d);
    else
        hipLaunchKernelGGL(k_dec_embed<bf16>, dim3(B), dim3(256), 0, s, (const bf16*)tok_emb, pos_emb, feed, feed_ld, pos_p, x, d);
}

void wh_launch_skinny(hipStream_t s, int prec, bool out_f32, const SkinnyArgs& a) {
    // K must split into 32-deep slabs per wave
    const bool split = (a.K % 128 == 0) && a.N < 8192;
    if (prec == WH_PREC_F32) {
        if (split) launch_skinny_mt<float, float, 4, 0>(s, a);
        else launch_skinny_mt<float, float, 1, 0>(s, a);
    } else if (out_f32) {
        if (split) launch_skinny_mt<bf16, float, 4, 0>(s, a);
        else launch_skinny_mt<bf16, float, 1, 0>(s, a);
    } else {
        if (split) launch_skinny_mt<bf16, bf16, 4, 0>(s, a);
        else launch_skinny_mt<bf16, bf16, 1, 0>(s, a);
    }
}

void wh_launch_lm_head(hipStream_t s, int prec, const SkinnyArgs& a) {
    if (prec == WH_PREC_F32) launch_skinny_mt<float, float, 1, 1>(s, a);
    else launch_skinny_mt<bf16, float, 1, 1>(s, a);
}

void wh_launch_argmax_finish(hipStream_t s, const float* part_val, const int* part_idx, int n_tiles, const int* pos_p,
                             const DecodeState& st, int B) {
    hipLaunchKernelGGL(k_argmax_finish, dim3(B), dim3(256), 0, s, part_val, part_idx, n_tiles, pos_p, st);
}

void wh_launch_step_advance(hipStream_t s, int* pos_p) { hipLaunchKernelGGL(k_step_advance, dim3(1), dim3(1), 0, s, pos_p); }

void wh_launch_dec_self_attn(hipStream_t s, int prec, const void* qkv, void* kc, void* vc, void* out, const int* pos_p,
                             int d, int n_heads, int tc, int B) {
    dim3 grid(n_heads, B);
    if (prec == WH_PREC_F32)
        hipLaunchKernelGGL(k_dec_self_attn<float>, grid, dim3(64), 0, s, (const float*)qkv, (float*)kc, (float*)vc, (float*)out, pos_p, d, n_heads, tc);
    else
        hipLaunchKernelGGL(k_dec_self_attn<bf16>, grid, dim3(64), 0, s, (const bf16*)qkv, (bf16*)kc, (bf16*)vc, (bf16*)out, pos_p, d, n_heads, tc);
}

size_t wh_cross_attn_smem(int S, int d, int n_heads, int splits) {
    const int per = (S + splits - 1) / splits;
    return sizeof(float) * ((size_t)n_heads * per + 4 * (size_t)d + 2 * (size_t)n_heads);
}

void wh_launch_dec_cross_attn(hipStream_t s, int prec, const void* q, const void* ck, const void* cv, float* part,
                              float* ml, int S, int d, int n_heads, int splits, int B) {
    dim3 grid(splits, B);
    const size_t sm = wh_cross_attn_smem(S, d, n_heads, splits);
    if (prec == WH_PREC_F32) {
        const int nch = (d / 4 + 63) / 64;  // f32: 4 elements per chunk
        switch (nch) {
            case 1: hipLaunchKernelGGL((k_dec_cross_attn<float, 1>), grid, dim3(256), sm, s, (const float*)q, (const float*)ck, (const float*)cv, part, ml, S, d, n_heads, splits); break;
            case 2: hipLaunchKernelGGL((k_dec_cross_attn<float, 2>), grid, dim3(256), sm, s, (const float*)q, (const float*)ck, (const float*)cv, part, ml, S, d, n_heads, splits); break;
            default: hipLaunchKernelGGL((k_dec_cross_attn<float, 5>), grid, dim3(256), sm, s, (const float*)q, (const float*)ck, (const float*)cv, part, ml, S, d, n_heads, splits); break;
        }
    } else {
        const int nch = (d / 8 + 63) / 64;
        switch (nch) {
            case 1: hipLaunchKernelGGL((k_dec_cross_attn<bf16, 1>), grid, dim3(256), sm, s, (const bf16*)q, (const bf16*)ck, (const bf16*)cv, part, ml, S, d, n_heads, splits); break;
            default: hipLaunchKernelGGL((k_dec_cross_attn<bf16, 3>), grid, dim3(256), sm, s, (const bf16*)q, (const bf16*)ck, (const bf16*)cv, part, ml, S, d, n_heads, splits); break;
        }
    }
}

void wh_launch_cross_combine(hipStream_t s, int prec, const float* part, const float* ml, void* out, int d, int n_heads,
                             int splits, int B) {
    if (prec == WH_PREC_F32)
        hipLaunchKernelGGL(k_cross_combine<float>, dim3(B), dim3(256), 0, s, part, ml, (float*)out, d, n_heads, splits);
    else
        hipLaunchKernelGGL(k_cross_combine<bf16>, dim3(B), dim3(256), 0, s, part, ml, (bf16*)out, d, n_heads, splits);
}
