// wh_fp8.hip — WH_PREC_FP8 pieces that have no bf16 twin: the e4m3 cross-attention K/V cache.
//
//   k_kv_absmax / k_kv_quant   present.{i}.encoder.{key,value} (reference src/main.rs:786-787), as produced in
//                              bf16 by the projection GEMM, become e4m3 codes with one scale per
//                              (clip, layer, K|V, head): scale = max|.| / 448 over the head's 1500 x 64 block
//                              (oracle: fake_quant_heads in oracle/whisper_oracle.c)
//   k_dec_cross_attn8          the per-token cross attention over that cache: half the HBM bytes of the bf16
//                              kernel per launch.  K's scale is folded into q, V's into the partial output, so
//                              the key loop carries no per-key scale work.
//
// All three are HBM-streaming kernels: 16-byte loads per lane, whole rows per wave instruction.
#include "wh_common.h"
#include "wh_kernels.h"

namespace {

typedef __attribute__((ext_vector_type(2))) float f32x2;

// ---- per-(plane, clip, head) absolute maximum of a bf16 [planes][B][S][d] cache ----------------------------
// grid (row chunks, planes * B); amax holds the maxima as raw f32 bit patterns (non-negative: the unsigned
// order is the float order), zeroed by the caller.
__global__ __launch_bounds__(256) void k_kv_absmax(const bf16* __restrict__ kv, unsigned* __restrict__ amax, int S, int d,
                                                   int n_heads, int rows_per_wg) {
    __shared__ unsigned hm[32];
    const int tid = threadIdx.x;
    if (tid < 32) hm[tid] = 0u;
    __syncthreads();
    const long pb = blockIdx.y;
    const int r0 = blockIdx.x * rows_per_wg, r1 = min(S, r0 + rows_per_wg);
    const int cpr = d >> 3;  // 16-byte chunks per row
    const bf16* base = kv + pb * (long)S * d;
    const int it1 = r1 * cpr;
    for (int it0 = r0 * cpr + tid; it0 < it1; it0 += 256 * 8) {  // eight 16-byte loads in flight per thread
        wh_u32x4 u[8];
#pragma unroll
        for (int i = 0; i < 8; i++) u[i] = *reinterpret_cast<const wh_u32x4*>(base + (long)min(it0 + i * 256, it1 - 1) * 8);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int it = min(it0 + i * 256, it1 - 1);  // a clamped re-read only repeats a value already counted
            // |bf16| as an integer compare: clear the sign bits, take the larger halfword
            const unsigned a0 = u[i].x & 0x7FFF7FFFu, a1 = u[i].y & 0x7FFF7FFFu, a2 = u[i].z & 0x7FFF7FFFu, a3 = u[i].w & 0x7FFF7FFFu;
            unsigned mx = max(max(a0 & 0xFFFFu, a0 >> 16), max(a1 & 0xFFFFu, a1 >> 16));
            mx = max(mx, max(max(a2 & 0xFFFFu, a2 >> 16), max(a3 & 0xFFFFu, a3 >> 16)));
            atomicMax(&hm[((it % cpr) * 8) / WH_HEAD_DIM], mx << 16);  // bf16 bits -> f32 bits
        }
    }
    __syncthreads();
    if (tid < n_heads && hm[tid]) atomicMax(amax + pb * n_heads + tid, hm[tid]);
}

// ---- bf16 -> e4m3 codes of value / scale, scale = amax / 448 (1 for an all-zero head) ------------------------
__global__ __launch_bounds__(256) void k_kv_quant(const bf16* __restrict__ kv, const float* __restrict__ amax,
                                                  unsigned char* __restrict__ out, long n_chunks, int S, int d, int n_heads) {
    const long it = (long)blockIdx.x * 256 + threadIdx.x;
    if (it >= n_chunks) return;
    const int cpr = d >> 3;
    const long row = it / cpr;
    const int ch = (int)(it - row * cpr);
    const long pb = row / S;
    const float am = amax[pb * n_heads + (ch * 8) / WH_HEAD_DIM];
    const float sc = am > 0.0f ? am / 448.0f : 1.0f;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(kv + it * 8);
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; e++) f[e] = fminf(fmaxf((float)v[e] / sc, -448.0f), 448.0f);  // the instruction returns NaN above 448
    int lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
    int hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
    typedef __attribute__((ext_vector_type(2))) int i32x2;
    *reinterpret_cast<i32x2*>(out + it * 8) = i32x2{lo, hi};
}

// ---- cross attention over the e4m3 cache -------------------------------------------------------------------
// One workgroup = (clip, key range), 4 waves.  A lane owns 16 consecutive dims (one 16-byte chunk of a key row,
// 4 lanes per head); a wave instruction covers KPI whole key rows (KPI = 64 / lanes per row when that divides,
// else 1 with NCH chunks per lane).  Online softmax per (lane group, key slot); the KPI slots and the 4 waves are
// merged through LDS, the key ranges of a clip by the consumer GEMM (frag merge in k_dec_gemm).
//   ck/cv: [B][S][d] e4m3 codes,  q: [B][d] bf16 pre-scaled by head_dim^-0.5,  amax_k/amax_v: [B][H]
template <int KPI, int NCH, int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k_dec_cross_attn8(const bf16* __restrict__ q, const unsigned char* __restrict__ ck,
                                                         const unsigned char* __restrict__ cv,
                                                         const float* __restrict__ amax_k, const float* __restrict__ amax_v,
                                                         float* __restrict__ part, float* __restrict__ ml, int S, int d,
                                                         int n_heads, int splits, bf16* __restrict__ out, int mpad) {
    static_assert(KPI == 1 || NCH == 1, "several keys per instruction only when a row fits a wave");
    constexpr int LPK = 64 / KPI;  // lanes per key row (KPI > 1)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sp = blockIdx.x, b = blockIdx.y;
    const int per = (S + splits - 1) / splits;
    const int ks = sp * per, j1 = min(S, ks + per);
    const int chunks = d >> 4;                     // 16-byte chunks per row
    const int slot = KPI > 1 ? lane / LPK : 0;     // which of the instruction's keys this lane reads
    const int lch = KPI > 1 ? lane % LPK : lane;   // chunk (first of NCH)

    float qf[NCH][16], o[NCH][16], mrun[NCH], lrun[NCH];
    bool live[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const int ch = lch + 64 * c;
        live[c] = ch < chunks;
        mrun[c] = -INFINITY;
        lrun[c] = 0.0f;
        const int chc = live[c] ? ch : 0;
        const float am = amax_k[b * n_heads + (chc * 16) / WH_HEAD_DIM];
        const float sk = am > 0.0f ? am / 448.0f : 1.0f;  // K's scale rides on q
        const bf16x8 q0 = *reinterpret_cast<const bf16x8*>(q + (long)b * d + chc * 16);
        const bf16x8 q1 = *reinterpret_cast<const bf16x8*>(q + (long)b * d + chc * 16 + 8);
#pragma unroll
        for (int e = 0; e < 8; e++) {
            qf[c][e] = live[c] ? (float)q0[e] * sk : 0.0f;
            qf[c][8 + e] = live[c] ? (float)q1[e] * sk : 0.0f;
        }
#pragma unroll
        for (int e = 0; e < 16; e++) o[c][e] = 0.0f;
    }
    const unsigned char* kb = ck + (long)b * S * d;
    const unsigned char* vb = cv + (long)b * S * d;
    wh_u32x4 kA[UNROLL][NCH], vA[UNROLL][NCH], kB[UNROLL][NCH], vB[UNROLL][NCH];
    auto load_set = [&](wh_u32x4 (&kk)[UNROLL][NCH], wh_u32x4 (&vv)[UNROLL][NCH], int j) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const int jj = min(j + u * KPI + slot, j1 - 1);  // tail: re-read the last key, masked in compute_set
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const int ch = live[c] ? lch + 64 * c : 0;
                // read-once stream: non-temporal (keeps the decode weights resident in L2 / the Infinity Cache)
                if constexpr (NT) {
                    kk[u][c] = __builtin_nontemporal_load(reinterpret_cast<const wh_u32x4*>(kb + (long)jj * d + ch * 16));
                    vv[u][c] = __builtin_nontemporal_load(reinterpret_cast<const wh_u32x4*>(vb + (long)jj * d + ch * 16));
                } else {
                    kk[u][c] = *reinterpret_cast<const wh_u32x4*>(kb + (long)jj * d + ch * 16);
                    vv[u][c] = *reinterpret_cast<const wh_u32x4*>(vb + (long)jj * d + ch * 16);
                }
            }
        }
    };
    auto compute_set = [&](const wh_u32x4 (&kk)[UNROLL][NCH], const wh_u32x4 (&vv)[UNROLL][NCH], int j) {
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            float s[UNROLL];
            float mx = mrun[c];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                const unsigned w[4] = {kk[u][c].x, kk[u][c].y, kk[u][c].z, kk[u][c].w};
                f32x2 acc = {0.0f, 0.0f};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8(w[i], false), hi = __builtin_amdgcn_cvt_pk_f32_fp8(w[i], true);
                    acc += lo * f32x2{qf[c][4 * i], qf[c][4 * i + 1]};
                    acc += hi * f32x2{qf[c][4 * i + 2], qf[c][4 * i + 3]};
                }
                float t = dpp_group_sum<4>(acc.x + acc.y);  // the 4 lanes of a head
                s[u] = (j + u * KPI + slot < j1) ? t : -INFINITY;
                mx = fmaxf(mx, s[u]);
            }
            const float msafe = (mx == -INFINITY) ? 0.0f : mx;  // a slot may own no key of this set
            const float scale = __expf(mrun[c] - msafe);        // first set: exp(-inf) = 0
            float ls = lrun[c] * scale;
#pragma unroll
            for (int e = 0; e < 16; e++) o[c][e] *= scale;
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                const float p = __expf(s[u] - msafe);
                ls += p;
                const f32x2 pp = {p, p};
                const unsigned w[4] = {vv[u][c].x, vv[u][c].y, vv[u][c].z, vv[u][c].w};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8(w[i], false), hi = __builtin_amdgcn_cvt_pk_f32_fp8(w[i], true);
                    const f32x2 a = f32x2{o[c][4 * i], o[c][4 * i + 1]} + pp * lo;
                    const f32x2 bb = f32x2{o[c][4 * i + 2], o[c][4 * i + 3]} + pp * hi;
                    o[c][4 * i] = a.x; o[c][4 * i + 1] = a.y; o[c][4 * i + 2] = bb.x; o[c][4 * i + 3] = bb.y;
                }
            }
            mrun[c] = mx;
            lrun[c] = ls;
        }
    };
    constexpr int GK = UNROLL * KPI;  // keys per register set
    constexpr int GS = 4 * GK;        // stride between this wave's consecutive key groups
    const int j0 = ks + wave * GK;
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
    // merge the 4 * KPI partial states of this key range (LDS): wm/wl [sets][H], wo [sets][d]
    constexpr int SETS = 4 * KPI;
    float* wm = smem;
    float* wl = wm + SETS * n_heads;
    float* wo = wl + SETS * n_heads;
    const int set = wave * KPI + slot;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        if (live[c]) {
            const int ch = lch + 64 * c;
            if ((lane & 3) == 0) {
                wm[set * n_heads + ch / 4] = mrun[c];
                wl[set * n_heads + ch / 4] = lrun[c];
            }
#pragma unroll
            for (int e = 0; e < 16; e += 4)
                *reinterpret_cast<f32x4*>(wo + (long)set * d + ch * 16 + e) = f32x4{o[c][e], o[c][e + 1], o[c][e + 2], o[c][e + 3]};
        }
    }
    __syncthreads();
    float* pp = part + ((long)b * splits + sp) * d;
    float* mp = ml + ((long)b * splits + sp) * n_heads * 2;
    for (int n = tid; n < d; n += 256) {
        const int h = n / WH_HEAD_DIM;
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < SETS; w++) M = fmaxf(M, wm[w * n_heads + h]);
        const float am = amax_v[b * n_heads + h];
        const float sv = am > 0.0f ? am / 448.0f : 1.0f;  // V's scale rides on the partial output
        float num = 0.0f, den = 0.0f;
#pragma unroll
        for (int w = 0; w < SETS; w++) {
            const float mw = wm[w * n_heads + h];
            const float sc = (mw == -INFINITY) ? 0.0f : __expf(mw - M);  // a wave / slot may own no keys
            num += sc * wo[(long)w * d + n];
            den += sc * wl[w * n_heads + h];
        }
        if (out) {  // one key range per clip: this IS the attention output (slab layout [d/32][mpad][32])
            out[((long)(n >> 5) * mpad + b) * 32 + (n & 31)] = (bf16)(num * sv / den);
            continue;
        }
        pp[n] = num * sv;
        if ((n % WH_HEAD_DIM) == 0) {
            mp[h] = M;
            mp[n_heads + h] = den;
        }
    }
}

template <int KPI, int NCH, int UNROLL>
void launch_ca8(hipStream_t s, const void* q, const void* ck, const void* cv, const float* amax_k, const float* amax_v, float* part,
                float* ml, int S, int d, int n_heads, int splits, int B, void* out, int mpad, bool stream_nt) {
    const size_t sm = wh_cross_lds_reserve((long)splits * B, sizeof(float) * ((size_t)2 * 4 * KPI * n_heads + (size_t)4 * KPI * d));
    if (sm > 48 * 1024) {
        wh_ensure_dyn_lds((const void*)k_dec_cross_attn8<KPI, NCH, UNROLL, true>, sm);
        wh_ensure_dyn_lds((const void*)k_dec_cross_attn8<KPI, NCH, UNROLL, false>, sm);
    }
    if (stream_nt)
        hipLaunchKernelGGL((k_dec_cross_attn8<KPI, NCH, UNROLL, true>), dim3(splits, B), dim3(256), sm, s, (const bf16*)q,
                           (const unsigned char*)ck, (const unsigned char*)cv, amax_k, amax_v, part, ml, S, d, n_heads, splits,
                           (bf16*)(splits == 1 ? out : nullptr), mpad);
    else
        hipLaunchKernelGGL((k_dec_cross_attn8<KPI, NCH, UNROLL, false>), dim3(splits, B), dim3(256), sm, s, (const bf16*)q,
                           (const unsigned char*)ck, (const unsigned char*)cv, amax_k, amax_v, part, ml, S, d, n_heads, splits,
                           (bf16*)(splits == 1 ? out : nullptr), mpad);
}

}  // namespace

void wh_launch_kv_quant(hipStream_t s, const void* kv_bf16, unsigned* amax, void* kv8, long planes_x_clips, int S, int d,
                        int n_heads) {
    const int rows_per_wg = 125;  // 12 workgroups per 1500-row plane
    hipLaunchKernelGGL(k_kv_absmax, dim3((S + rows_per_wg - 1) / rows_per_wg, (unsigned)planes_x_clips), dim3(256), 0, s,
                       (const bf16*)kv_bf16, amax, S, d, n_heads, rows_per_wg);
    const long n_chunks = planes_x_clips * S * (d / 8);
    hipLaunchKernelGGL(k_kv_quant, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, s, (const bf16*)kv_bf16,
                       (const float*)amax, (unsigned char*)kv8, n_chunks, S, d, n_heads);
}

void wh_launch_dec_cross_attn8(hipStream_t s, const void* q, const void* ck, const void* cv, const float* amax_k,
                               const float* amax_v, float* part, float* ml, int S, int d, int n_heads, int splits, int B, void* out,
                               int mpad, bool stream_nt) {
    const int chunks = d / 16;
    if (chunks == 32) launch_ca8<2, 1, 4>(s, q, ck, cv, amax_k, amax_v, part, ml, S, d, n_heads, splits, B, out, mpad, stream_nt);       // d = 512
    else if (chunks == 16) launch_ca8<4, 1, 2>(s, q, ck, cv, amax_k, amax_v, part, ml, S, d, n_heads, splits, B, out, mpad, stream_nt);  // d = 256
    else if (chunks == 8) launch_ca8<8, 1, 1>(s, q, ck, cv, amax_k, amax_v, part, ml, S, d, n_heads, splits, B, out, mpad, stream_nt);   // d = 128
    else if (chunks <= 64) launch_ca8<1, 1, 4>(s, q, ck, cv, amax_k, amax_v, part, ml, S, d, n_heads, splits, B, out, mpad, stream_nt);  // d <= 1024
    else launch_ca8<1, 2, 2>(s, q, ck, cv, amax_k, amax_v, part, ml, S, d, n_heads, splits, B, out, mpad, stream_nt);                    // d = 1280
}
