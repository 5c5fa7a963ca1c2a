// wh_attn.hip — encoder self-attention (non-causal, no mask), flash-style, head_dim 64, gfx950.
//
// Stands in for the MatMul→Softmax→MatMul subgraph of every encoder layer of encoder_model.onnx
// (run via reference src/main.rs:703; definition [3P] modeling_whisper.py eager_attention_forward
// :215-238 with q pre-scaled by head_dim^-0.5 at :309 — the scale is folded into W_q/b_q at load).
// The [S,S] score matrix is never written: one workgroup owns 64 query rows of one (clip, head),
// four waves x 16 rows; K and V^T tiles of 64 keys are staged in LDS once per workgroup and
// consumed as MFMA operands; softmax runs online in f32 registers.
//
// Layouts (T = bf16 or f32):
//   qk : [clip][S][2*d]      q at column h*64, k at column d + h*64           (QK projection output)
//   vT : [clip][d][ldv]      row h*64+e holds V[:, e] over keys (key-contiguous, zero beyond S)
//   out: [clip][S][d]        column h*64+e
#include "wh_common.h"
#include "wh_kernels.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void k_enc_attn(const T* __restrict__ qk, const T* __restrict__ vT,
                                                  T* __restrict__ out, int S, int d, int ldv, int n_heads) {
    constexpr int HD = WH_HEAD_DIM, KV = 64;
    constexpr int LD = HD + 16 / (int)sizeof(T);  // padded LDS row (elements): +16 B
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr int CPR = HD / EPC;                 // 16-B chunks per 64-element row
    __shared__ __attribute__((aligned(16))) T Ks[KV * LD];       // [key][e]
    __shared__ __attribute__((aligned(16))) T Vs[HD * LD];       // [e][key]
    __shared__ __attribute__((aligned(16))) T Ps[4 * 16 * LD];   // per wave [q][key]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fl = lane & 15, fg = lane >> 4;
    // XCD-aware work mapping (speed only): workgroups are dealt round-robin over the 8 XCDs, so linear
    // ids with equal (id % 8) share an L2.  All query tiles of one (clip, head) pair re-read the same K / V,
    // so a pair's tiles are given ids with the same residue: id = slot * 8 + xcd, pair = (slot / nq) * 8 + xcd.
    int qt, pair;
    {
        const int nq = (S + 63) / 64, npair = gridDim.x / nq, id = blockIdx.x;
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
    const int q0 = qt * 64 + wave * 16;
    const T* qkc = qk + clip * (long)S * 2 * d;
    const T* vc = vT + clip * (long)d * ldv + (long)h * HD * ldv;

    // Q fragments: A[i = q (fl)][k = e]
    typename FragT<T>::type qf[2];
    {
        int q = q0 + fl;
        if (q > S - 1) q = S - 1;
        const T* qp = qkc + (long)q * 2 * d + h * HD;
        qf[0] = load_frag<T>(qp + fg * 8);
        qf[1] = load_frag<T>(qp + 32 + fg * 8);
    }
    f32x4 o[4];
#pragma unroll
    for (int t = 0; t < 4; t++) o[t] = f32x4{0, 0, 0, 0};
    float mrow[4], lrow[4];
#pragma unroll
    for (int r = 0; r < 4; r++) { mrow[r] = -INFINITY; lrow[r] = 0.0f; }

    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    T* Pw = Ps + wave * 16 * LD;
    for (int k0 = 0; k0 < S; k0 += KV) {
        __syncthreads();  // previous tile fully consumed
        for (int c = tid; c < KV * CPR; c += 256) {
            int row = c / CPR, col = (c % CPR) * EPC;
            int key = k0 + row;
            if (key > S - 1) key = S - 1;
            *reinterpret_cast<u32x4*>(&Ks[row * LD + col]) =
                *reinterpret_cast<const u32x4*>(qkc + (long)key * 2 * d + d + h * HD + col);
            // V^T rows are e, columns keys k0..k0+63 (ldv >= S rounded up to 64, zero padded)
            *reinterpret_cast<u32x4*>(&Vs[row * LD + col]) =
                *reinterpret_cast<const u32x4*>(vc + (long)row * ldv + k0 + col);
        }
        __syncthreads();
        // S tile: D[i = q][j = key]; rows i = 4*fg + r, col j = fl
        f32x4 sc[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            sc[t] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                typename FragT<T>::type kf = load_frag<T>(&Ks[(t * 16 + fl) * LD + ks * 32 + fg * 8]);
                mma16(sc[t], qf[ks], kf);
            }
        }
        float alpha[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                if (k0 + t * 16 + fl >= S) sc[t][r] = -INFINITY;
                mx = fmaxf(mx, sc[t][r]);
            }
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
            const float mn = fmaxf(mrow[r], mx);
            alpha[r] = __expf(mrow[r] - mn);
            float rs = 0.0f;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                float p = __expf(sc[t][r] - mn);
                sc[t][r] = p;
                rs += p;
            }
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) rs += __shfl_xor(rs, off);
            lrow[r] = lrow[r] * alpha[r] + rs;
            mrow[r] = mn;
        }
        // P to LDS as [q][key], then back as the row operand of P·V
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) Pw[(4 * fg + r) * LD + t * 16 + fl] = cvt_out<T>(sc[t][r]);
        __syncthreads();
        typename FragT<T>::type pf[2];
        pf[0] = load_frag<T>(&Pw[fl * LD + fg * 8]);
        pf[1] = load_frag<T>(&Pw[fl * LD + 32 + fg * 8]);
#pragma unroll
        for (int t = 0; t < 4; t++) {
#pragma unroll
            for (int r = 0; r < 4; r++) o[t][r] *= alpha[r];
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                typename FragT<T>::type vf = load_frag<T>(&Vs[(t * 16 + fl) * LD + ks * 32 + fg * 8]);
                mma16(o[t], pf[ks], vf);  // D[i = q][j = e]
            }
        }
    }
    T* oc = out + clip * (long)S * d;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int q = q0 + 4 * fg + r;
        if (q >= S) continue;
        const float inv = 1.0f / lrow[r];
#pragma unroll
        for (int t = 0; t < 4; t++) oc[(long)q * d + h * HD + t * 16 + fl] = cvt_out<T>(o[t][r] * inv);
    }
}

}  // namespace

void wh_launch_enc_attn(hipStream_t s, int prec, const void* qk, const void* vT, void* out, int n_clips, int S, int d,
                        int n_heads, int ldv) {
    dim3 grid(((S + 63) / 64) * n_heads * n_clips);
    if (prec == WH_PREC_F32)
        hipLaunchKernelGGL(k_enc_attn<float>, grid, dim3(256), 0, s, (const float*)qk, (const float*)vT, (float*)out, S, d, ldv, n_heads);
    else
        hipLaunchKernelGGL(k_enc_attn<bf16>, grid, dim3(256), 0, s, (const bf16*)qk, (const bf16*)vT, (bf16*)out, S, d, ldv, n_heads);
}
