// wh_attn.hip — encoder self-attention (non-causal, no mask), flash-style, head_dim 64, gfx950.
//
// Stands in for the MatMul→Softmax→MatMul subgraph of every encoder layer of encoder_model.onnx
// (run via reference src/main.rs:703; definition [3P] modeling_whisper.py eager_attention_forward
// :215-238 with q pre-scaled by head_dim^-0.5 at :309 — the scale is folded into W_q/b_q at load).
// The [S,S] score matrix is never written.  One workgroup = 128 query rows of one (clip, head): four
// waves x 32 rows.  Per 64-key tile (K and V^T staged in LDS, double buffered, one barrier per tile):
//   S^T = K · Q^T      (MFMA row operand = K rows from LDS, column operand = Q fragments in registers)
//         → every lane holds 16 scores of ONE query column: the online-softmax state (max, sum) is
//           lane-local apart from one cross-lane max over the four lane groups;
//   O^T += V^T · P^T   (row operand = V^T from LDS, column operand = P^T straight from the score
//           registers: the k-slot ↔ key assignment of the MFMA is permuted identically on both
//           operands, so P never goes through LDS and needs no transpose)
//         → the rescale factor of a query is lane-local too, and each lane finally stores 4 consecutive
//           features of its query row.
//
// Layouts (T = bf16 or f32):
//   qk : [clip][S][2*d]      q at column h*64, k at column d + h*64           (QK projection output)
//   vT : [clip][d][ldv]      row h*64+e holds V[:, e] over keys (key-contiguous, zero beyond S)
//   out: [clip][S][d]        column h*64+e
#include <type_traits>

#include "wh_common.h"
#include "wh_kernels.h"

namespace {

template <typename T> struct HalfFrag;  // 4 consecutive elements
template <> struct HalfFrag<bf16> { typedef bf16x4 type; };
template <> struct HalfFrag<float> { typedef f32x4 type; };
template <> struct HalfFrag<xf32> { typedef f32x4 type; };
template <> struct HalfFrag<h2> { typedef f32x4 type; };   // (unused: h2 rows are read limb-wise, see vt_frag)

__device__ __forceinline__ bf16x8 join_half(bf16x4 a, bf16x4 b) { return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }
__device__ __forceinline__ f32x8 join_half(f32x4 a, f32x4 b) { return f32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }
__device__ __forceinline__ void pack_p(bf16x8& f, const f32x4& a, const f32x4& b) {
    f = bf16x8{(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3], (bf16)b[0], (bf16)b[1], (bf16)b[2], (bf16)b[3]};
}
__device__ __forceinline__ void pack_p(f32x8& f, const f32x4& a, const f32x4& b) { f = f32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }
// WH_PREC_F16X3: the probabilities enter the second product as two fp16 limbs like any other operand
__device__ __forceinline__ void pack_p(xfrag& f, const f32x4& a, const f32x4& b) { f = x3_split(f32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}); }
template <typename T> struct Joiner {
    template <typename H> static __device__ __forceinline__ auto join(H a, H b) { return join_half(a, b); }
};
template <> struct Joiner<xf32> {
    static __device__ __forceinline__ xfrag join(f32x4 a, f32x4 b) { return x3_split(join_half(a, b)); }
};
// V^T row operand of contraction step ks: k-slots 8 fg + j <-> keys 32 ks + 4 fg + j (j < 4) and 32 ks + 16 + 4 fg + j - 4 (j >= 4) of the 64-key tile
template <typename T>
__device__ __forceinline__ typename FragT<T>::type vt_frag(const T* row, int ks, int fg) {
    typedef typename HalfFrag<T>::type half_t;
    const T* vp = row + 32 * ks + 4 * fg;
    return Joiner<T>::join(*reinterpret_cast<const half_t*>(vp), *reinterpret_cast<const half_t*>(vp + 16));
}
template <>
__device__ __forceinline__ xfrag vt_frag<h2>(const h2* row, int ks, int fg) {   // 4 consecutive keys = 8 bytes of hi limbs + 8 bytes of lo limbs of the 32-key block
    const char* b = reinterpret_cast<const char*>(row) + 128 * ks + 8 * fg;
    const f16x4 h0 = *reinterpret_cast<const f16x4*>(b), h1 = *reinterpret_cast<const f16x4*>(b + 32);
    const f16x4 l0 = *reinterpret_cast<const f16x4*>(b + 64), l1 = *reinterpret_cast<const f16x4*>(b + 96);
    xfrag r;
    r.hi = f16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
    r.lo = f16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
    return r;
}

template <typename T, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void k_enc_attn(const T* __restrict__ qk, const T* __restrict__ vT,
                                                          T* __restrict__ out, int S, int d, int ldv, int n_heads) {
    constexpr int HD = WH_HEAD_DIM, KV = 64, QB = NWAVES * 32, NT_ = NWAVES * 64;
    // padded LDS row of the K tile (elements): +16 B, conflict-free for 16-B reads; h2 rows (two 128-byte blocks: hi and lo chunks 64 bytes
    // apart) take +32 B: 72 dwords per row put the 16 rows of a fragment read on 16 distinct 4-bank groups
    constexpr int LD = __is_same(T, h2) ? HD + 8 : HD + 16 / (int)sizeof(T);
    // V^T tile: read 8 bytes per lane (bf16), 16 lanes per LDS cycle — rows 144 B apart put lanes fl and fl+8 on the
    // same banks (2-way conflict on every read); 136-B rows spread the 16 lanes over all 32 banks.  Its staging
    // stores are then 8-byte ones (rows are only 8-byte aligned).
    constexpr int LDV = (sizeof(T) == 2 || __is_same(T, h2)) ? HD + 4 : LD;   // (h2: 8-byte limb reads like bf16's — 68-dword rows spread 16 rows x 2 groups over all 64 banks)
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr int CPR = HD / EPC;                 // 16-B chunks per 64-element row
    constexpr int NCH = KV * CPR / NT_;           // staging chunks per thread per operand: 2 (bf16) / 4 (f32) with 4 waves, half with 8
    typedef typename FragT<T>::type frag_t;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* Ks = reinterpret_cast<T*>(smem_raw);       // [2][KV][LD]   rows = keys
    T* Vs = Ks + 2 * KV * LD;                     // [2][HD][LDV]  rows = features, columns = keys

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fl = lane & 15, fg = lane >> 4;
    // XCD-aware work mapping (speed only): workgroups are dealt round-robin over the 8 XCDs, so linear ids
    // with equal (id % 8) share an L2; the query blocks of one (clip, head) pair re-read the same K / V.
    int qt, pair;
    {
        const int nq = (S + QB - 1) / QB, npair = gridDim.x / nq, id = blockIdx.x;
        if ((npair & 7) == 0) {
            const int xcd = id & 7, slot = id >> 3;
            pair = (slot / nq) * 8 + xcd;
            qt = slot % nq;
        } else {
            pair = id / nq;
            qt = id % nq;
        }
    }
    const int h = pair % n_heads;
    const long clip = pair / n_heads;
    const int q0 = qt * QB + wave * 32;
    const T* qkc = qk + clip * (long)S * 2 * d;
    const T* vc = vT + clip * (long)d * ldv + (long)h * HD * ldv;

    // Q^T column operands: lane (fl = query, fg) holds Q[q][32*ks + 8*fg ..]
    frag_t qf[2][2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        int q = q0 + 16 * u + fl;
        if (q > S - 1) q = S - 1;
        const T* qp = qkc + (long)q * 2 * d + h * HD;
        qf[u][0] = load_frag<T>(qp + fg * 8);
        qf[u][1] = load_frag<T>(qp + 32 + fg * 8);
    }
    f32x4 o[2][4];   // O^T: rows e = 16*te + 4*fg + r, column q = fl
    float mrow[2], lsum[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        mrow[u] = -INFINITY;
        lsum[u] = 0.0f;
#pragma unroll
        for (int te = 0; te < 4; te++) o[u][te] = f32x4{0, 0, 0, 0};
    }

    // staging: thread → (row, 16-B chunk) of the K tile and of the V^T tile
    int st_row[NCH], st_col[NCH];
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        const int c = tid + i * NT_;
        st_row[i] = c / CPR;
        st_col[i] = (c % CPR) * EPC;
    }
    u32x4 kreg[NCH], vreg[NCH];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            int key = k0 + st_row[i];
            if (key > S - 1) key = S - 1;
            kreg[i] = *reinterpret_cast<const u32x4*>(qkc + (long)key * 2 * d + d + h * HD + st_col[i]);
            vreg[i] = *reinterpret_cast<const u32x4*>(vc + (long)st_row[i] * ldv + k0 + st_col[i]);  // ldv covers k0+63, zero padded
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            *reinterpret_cast<u32x4*>(&Ks[(buf * KV + st_row[i]) * LD + st_col[i]]) = kreg[i];
            if constexpr (sizeof(T) == 2) {
                typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
                u32x2* vd = reinterpret_cast<u32x2*>(&Vs[(buf * HD + st_row[i]) * LDV + st_col[i]]);
                vd[0] = u32x2{vreg[i].x, vreg[i].y};
                vd[1] = u32x2{vreg[i].z, vreg[i].w};
            } else {
                *reinterpret_cast<u32x4*>(&Vs[(buf * HD + st_row[i]) * LDV + st_col[i]]) = vreg[i];
            }
        }
    };
    load_tile(0);
    store_tile(0);
    __syncthreads();
    const int nt = (S + KV - 1) / KV;
    // One key tile.  LAST = the final tile: the only one that may hold keys >= S (masked) and the only one with no
    // successor to stage — peeled so that the steady-state body carries neither the mask selects nor those branches.
    auto tile_body = [&](int it, auto last_c) {
        constexpr bool LAST = decltype(last_c)::value;
        const int cur = it & 1, k0 = it * KV;
        if (!LAST) load_tile(k0 + KV);  // flies under this tile's MFMAs
        const T* Kc = Ks + cur * KV * LD;
        const T* Vc = Vs + cur * HD * LDV;
        // ---- S^T[key][q]: rows key = 16*t + 4*fg + r, column q = fl ------------------------------------
        f32x4 sc[2][4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const frag_t k0f = slab_frag<T>(&Kc[(t * 16 + fl) * LD], fg);
            const frag_t k1f = slab_frag<T>(&Kc[(t * 16 + fl) * LD + 32], fg);
#pragma unroll
            for (int u = 0; u < 2; u++) {
                sc[u][t] = f32x4{0, 0, 0, 0};
                mma16(sc[u][t], k0f, qf[u][0]);
                mma16(sc[u][t], k1f, qf[u][1]);
            }
        }
        // ---- online softmax, lane-local per query column -----------------------------------------------
        // VALU budget (this loop is VALU-bound, not MFMA-bound: 32 MFMAs = 512 cycles per tile and wave): the key
        // mask only on the tail tile, scores taken to the exp2 domain by one packed FMA per pair, packed f32 math
        // for the rescale and the row sum, the cross-row max on v_permlane*_swap instead of the LDS crossbar.
        typedef __attribute__((ext_vector_type(2))) float f32x2;
        constexpr float LOG2E = 1.44269504088896341f;
        frag_t pf[2][2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            if (LAST) {
#pragma unroll
                for (int t = 0; t < 4; t++)
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (k0 + t * 16 + 4 * fg + r >= S) sc[u][t][r] = -INFINITY;
            }
            // a chain of max(max(m, a), b): one v_max3_f32 per two scores (16 per query column instead of the 24 a balanced tree takes)
            float mx = fmaxf(sc[u][0][0], sc[u][0][1]);
            mx = fmaxf(fmaxf(mx, sc[u][0][2]), sc[u][0][3]);
#pragma unroll
            for (int t = 1; t < 4; t++) {
                mx = fmaxf(fmaxf(mx, sc[u][t][0]), sc[u][t][1]);
                mx = fmaxf(fmaxf(mx, sc[u][t][2]), sc[u][t][3]);
            }
            mx = xrow_max(mx);
            const float mn = fmaxf(mrow[u], mx);
            const float alpha = __builtin_amdgcn_exp2f((mrow[u] - mn) * LOG2E);
            const float nm2 = -mn * LOG2E;
            f32x2 rs2 = {0.0f, 0.0f};
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const f32x2 e = f32x2{sc[u][t][r], sc[u][t][r + 1]} * f32x2{LOG2E, LOG2E} + f32x2{nm2, nm2};  // v_pk_fma_f32
                    const f32x2 p = {__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
                    sc[u][t][r] = p.x;
                    sc[u][t][r + 1] = p.y;
                    rs2 += p;
                }
            lsum[u] = lsum[u] * alpha + (rs2.x + rs2.y);  // this lane group's share; the four groups are added at the end
            mrow[u] = mn;
            const f32x2 a2 = {alpha, alpha};
#pragma unroll
            for (int te = 0; te < 4; te++) {
                const f32x2 lo = f32x2{o[u][te][0], o[u][te][1]} * a2, hi = f32x2{o[u][te][2], o[u][te][3]} * a2;  // v_pk_mul_f32
                o[u][te] = f32x4{lo.x, lo.y, hi.x, hi.y};
            }
            // P^T column operands: k-slot 8*fg + j ↔ key 16*t0 + 4*fg + j (j < 4), 16*t1 + 4*fg + j - 4 (j >= 4)
            pack_p(pf[u][0], sc[u][0], sc[u][1]);
            pack_p(pf[u][1], sc[u][2], sc[u][3]);
        }
        // ---- O^T[e][q] += V^T[e][keys] P^T[keys][q], same k-slot ↔ key assignment on the row operand ----
#pragma unroll
        for (int te = 0; te < 4; te++) {
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const frag_t vf = vt_frag<T>(&Vc[(te * 16 + fl) * LDV], ks, fg);
                mma16(o[0][te], vf, pf[0][ks]);
                mma16(o[1][te], vf, pf[1][ks]);
            }
        }
        if (!LAST) {
            store_tile(cur ^ 1);
            __syncthreads();
        }
    };
    for (int it = 0; it + 1 < nt; it++) tile_body(it, std::false_type{});
    tile_body(nt - 1, std::true_type{});
    T* oc = out + clip * (long)S * d;
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const float l = xrow_sum(lsum[u]);
        const int q = q0 + 16 * u + fl;
        if (q >= S) continue;
        const float inv = 1.0f / l;
#pragma unroll
        for (int te = 0; te < 4; te++)
            store4(oc + (long)q * d + h * HD + te * 16 + 4 * fg, o[u][te][0] * inv, o[u][te][1] * inv, o[u][te][2] * inv, o[u][te][3] * inv);
    }
}

}  // namespace

void wh_launch_enc_attn(hipStream_t s, int prec, const void* qk, const void* vT, void* out, int n_clips, int S, int d,
                        int n_heads, int ldv) {
    if (prec == WH_PREC_F32) {
        dim3 grid(((S + 127) / 128) * n_heads * n_clips);
        const size_t sm = (size_t)2 * 2 * 64 * (64 + 4) * 4;  // 69.6 KB: above the default dynamic-LDS limit
        wh_ensure_dyn_lds((const void*)k_enc_attn<float, 4>, sm);
        hipLaunchKernelGGL((k_enc_attn<float, 4>), grid, dim3(256), sm, s, (const float*)qk, (const float*)vT, (float*)out, S, d, ldv, n_heads);
    } else if (prec == WH_PREC_F16X3) {   // Q, K, V^T and the output as fp16 limb pairs (h2); the probabilities are split in registers
        // (4 waves, two workgroups per CU; the 8-wave form that pays in bf16 measured 190 vs 185 ms per 2048-clip step here)
        dim3 grid(((S + 127) / 128) * n_heads * n_clips);
        const size_t sm = (size_t)2 * 2 * 64 * (64 + 8) * 4;
        wh_ensure_dyn_lds((const void*)k_enc_attn<h2, 4>, sm);
        hipLaunchKernelGGL((k_enc_attn<h2, 4>), grid, dim3(256), sm, s, (const h2*)qk, (const h2*)vT, (h2*)out, S, d, ldv, n_heads);
    } else {
        const size_t sm = (size_t)2 * 2 * 64 * (64 + 8) * 2;
        // 8 waves = 256 query rows per workgroup: each K / V^T tile is staged once per 256 queries instead of once per 128
        // (a per-query result does not depend on the grouping).  WH_ENC_ATTN_W4=1: the 4-wave form, for A/B runs.
        static const bool w4 = getenv("WH_ENC_ATTN_W4") != nullptr && atoi(getenv("WH_ENC_ATTN_W4")) != 0;
        if (w4) {
            dim3 grid(((S + 127) / 128) * n_heads * n_clips);
            hipLaunchKernelGGL((k_enc_attn<bf16, 4>), grid, dim3(256), sm, s, (const bf16*)qk, (const bf16*)vT, (bf16*)out, S, d, ldv, n_heads);
        } else {
            dim3 grid(((S + 255) / 256) * n_heads * n_clips);
            hipLaunchKernelGGL((k_enc_attn<bf16, 8>), grid, dim3(512), sm, s, (const bf16*)qk, (const bf16*)vT, (bf16*)out, S, d, ldv, n_heads);
        }
    }
}
