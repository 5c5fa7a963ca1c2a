// wh_cross_es.hip — decoder cross-attention of one position computed ON THE ENCODER STATES (gfx950, bf16, whisper-base
// geometry: d_model 512, 8 heads) — the HBM-bound kernel of batched decode, streaming half the bytes of k_dec_cross_attn.
//
// The reference's decoder graphs project the encoder states once per clip and layer to K = E Wk^T and V = E Wv^T + bv
// (present.{i}.encoder.{key,value}, reference src/main.rs:771-787) and every later token reads both (:798-812): 2 S d
// elements per clip, layer and position.  Both products are linear in E, so for head h
//     score_h[key] = q_h . K_h[key]          = (Wk_h^T q_h) . E[key]              =: qe_h . E[key]        (qe_h: d values)
//     out_h        = sum_key p_h[key] V_h[key] = Wv_h (sum_key p_h[key] E[key]) + bv_h =: Wv_h ctx_h + bv_h  (ctx_h: d values)
// (sum_key p = 1).  The kernel therefore streams E itself — S d elements per clip, ONE array shared by every decoder layer
// — and uses each key row twice while it sits in LDS: once against the 8 expanded queries, once weighted by the 8
// probability rows.  The expansion q -> qe and the contraction ctx -> out live in the weights of the two decode GEMMs
// around the kernel (wh_model.cpp: G_l = blockdiag(Wk_h^T) Wq, Wvo_l = Wo blockdiag(Wv_h)); the cross-K/V projection and
// its 2 Ld S d cache disappear.  Per key row the work is 8 x more multiply-adds than the projected form (8 heads x 512
// instead of 8 x 64) — 25 GFLOP per launch at 1024 clips, which is why it runs on the matrix cores:
//
//   scores  D[m][key] = sum_dim  QE[m][dim] * E[key][dim]     mfma 16x16x32 bf16, rows m = {hi(qe_h) : h} ++ {lo(qe_h) : h}
//   output  C[m][dim] = sum_key  P[m][key]  * E[key][dim]     mfma 16x16x32 bf16, rows m = {hi(p_h)} ++ {lo(p_h)}
//
// Both row operands are split into a bf16 head and a bf16 remainder (x = hi + lo to ~16 mantissa bits): the 8 heads fill
// only half of a 16-row MFMA tile, the other half carries the remainders for free, and neither the expanded query nor the
// probabilities lose precision to bf16 — the only rounded quantity is E, which the projected form rounds as its GEMM operand
// as well.
//
// One workgroup (4 waves) per clip.  E streams global -> LDS on the LDS-DMA path (global_load_lds_dwordx4, no registers)
// through a ring of four 32-key tiles (32 KiB each; three in flight while one is consumed); per tile:
//   A: counted vmcnt wait + barrier (tile visible)            | issue the tile three ahead into the slot just freed
//   scores: wave w -> keys 16 (w / 2) .. + 15, dims 256 (w % 2) .. + 255: 8 ds_read_b128 + 8 MFMA, hi + lo rows added across
//           lanes (v_permlane32_swap), partial scores to LDS
//   B: barrier
//   every wave: all 8 x 32 scores (two dim halves added), online softmax (running max / sum per head; identical in the four
//           waves), P operand built in registers, accumulators rescaled
//   output: wave w -> dims 128 w .. + 127: a lane reads 8 keys x 8 dims as 8 ds_read_b128 and transposes the 8 x 8 block in
//           registers (32 v_perm_b32) into the 8 key-contiguous column operands; 8 MFMA
// Bank conflicts: LDS is written linearly by the DMA, so the swizzle is on the source side — LDS chunk p of tile row r holds
// dim-chunk p ^ (r & 15); both read patterns then hit 16 distinct 16-byte slots per service group (the key order inside a
// 32-deep contraction step is (fg & 1) * 16 + (fg >> 1) * 8 + j for the same reason).
// No cross-wave merge at the end: every wave has seen every key and owns its own 128 output dims.
#include <stdlib.h>

#include "wh_common.h"
#include "wh_kernels.h"

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int ES_D = 512, ES_H = 8, ES_TK = 32, ES_NSTAGE = 4;
constexpr int ES_ROWB = ES_D * 2;                      // bytes per key row
constexpr int ES_TILEB = ES_TK * ES_ROWB;              // 32 KiB
constexpr int ES_SCP = 36;                             // floats per (dim half, head) row of the score exchange (32 keys + pad)
constexpr int ES_LDS = ES_NSTAGE * ES_TILEB + 2 * ES_H * ES_SCP * 4 + 4 * 16 * 4;

template <int N> __device__ __forceinline__ void es_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int AUX>
__device__ __forceinline__ void es_glds16(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, AUX);
}

__device__ __forceinline__ float es_sum32(float v) {   // v(lane) + v(lane ^ 32), in every lane
    const wh_u32x2 t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(t.x) + __uint_as_float(t.y);
}

// qe : [B][8][512] f32 expanded queries (natural-log score units)        E: [B][S][512] bf16 encoder states (final LayerNorm applied)
// out: ctx as the decode GEMM's operand, slab layout [8 * 512 / 32][mpad][32] bf16, column h * 512 + dim
template <int AUX>
__global__ __launch_bounds__(256, 1) void k_dec_cross_attn_es(const float* __restrict__ qe, const bf16* __restrict__ E,
                                                              bf16* __restrict__ out, int S, int mpad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sc = reinterpret_cast<float*>(smem + ES_NSTAGE * ES_TILEB);   // [2 dim halves][8 heads][ES_SCP]
    float* wx = sc + 2 * ES_H * ES_SCP;                                   // [4 waves][16]: wave-private lane exchange
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fl = lane & 15, fg = lane >> 4;
    const int b = blockIdx.x;
    const bf16* Eb = E + (long)b * S * ES_D;
    const int ntile = (S + ES_TK - 1) / ES_TK;
    const int hf = wave & 1, kt = wave >> 1;

    // ---- expanded queries of this wave's dim half as the MFMA row operand: row fl -> head fl & 7, rows 8..15 the remainders
    // (issued by hand: the compiler would sink plain loads below the ring's first stages and then wait for everything)
    f32x4 qraw[16];
    {
        const float* qp = qe + ((long)b * ES_H + (fl & 7)) * ES_D + 256 * hf + 8 * fg;
#define WH_ES_QLD(I, OFF) asm volatile("global_load_dwordx4 %0, %1, off offset:" #OFF : "=&v"(qraw[I]) : "v"(qp) : "memory")
        WH_ES_QLD(0, 0);    WH_ES_QLD(1, 16);   WH_ES_QLD(2, 128);  WH_ES_QLD(3, 144);
        WH_ES_QLD(4, 256);  WH_ES_QLD(5, 272);  WH_ES_QLD(6, 384);  WH_ES_QLD(7, 400);
        WH_ES_QLD(8, 512);  WH_ES_QLD(9, 528);  WH_ES_QLD(10, 640); WH_ES_QLD(11, 656);
        WH_ES_QLD(12, 768); WH_ES_QLD(13, 784); WH_ES_QLD(14, 896); WH_ES_QLD(15, 912);
#undef WH_ES_QLD
    }
    // ---- the ring: wave w brings rows 8 w .. 8 w + 7 of a tile, one wave-instruction = one 1 KiB key row
    auto stage = [&](int t) {
        char* base = smem + (t & (ES_NSTAGE - 1)) * ES_TILEB;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int r = wave * 8 + j;
            const int key = min(t * ES_TK + r, S - 1);   // tail rows: the last key again (finite; masked below)
            es_glds16<AUX>(Eb + (long)key * ES_D + ((lane ^ (r & 15)) << 3), base + r * ES_ROWB);
        }
    };
    stage(0);
    stage(1);
    stage(2);
    // the 16 query loads were issued first: done when at most the 24 ring loads are outstanding (vmcnt retires in order)
    asm volatile("s_waitcnt vmcnt(24)"
                 : "+v"(qraw[0]), "+v"(qraw[1]), "+v"(qraw[2]), "+v"(qraw[3]), "+v"(qraw[4]), "+v"(qraw[5]), "+v"(qraw[6]), "+v"(qraw[7]),
                   "+v"(qraw[8]), "+v"(qraw[9]), "+v"(qraw[10]), "+v"(qraw[11]), "+v"(qraw[12]), "+v"(qraw[13]), "+v"(qraw[14]), "+v"(qraw[15])
                 :: "memory");
    bf16x8 qa[8];
    {
        const bool lo = fl >= 8;
#pragma unroll
        for (int s = 0; s < 8; s++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const float v = qraw[2 * s + (u >> 2)][u & 3] * 1.44269504088896341f;   // scores in log2 units: p = exp2(s - m)
                const bf16 h = (bf16)v;
                qa[s][u] = lo ? (bf16)(v - (float)h) : h;
            }
        }
    }

    f32x4 acc[8];
#pragma unroll
    for (int e = 0; e < 8; e++) acc[e] = f32x4{0, 0, 0, 0};
    float m_run = -INFINITY, l_run = 0.0f;
    const int kb = 16 * (fg & 1) + 8 * (fg >> 1);   // first key (within a tile) of this lane's contraction slots

    for (int t = 0; t < ntile; t++) {
        {   // tile t has landed (this wave's share); the younger tiles stay in flight
            const int newer = min(2, ntile - 1 - t);
            if (newer == 2) es_wait_vm<16>();
            else if (newer == 1) es_wait_vm<8>();
            else es_wait_vm<0>();
        }
        __builtin_amdgcn_s_barrier();   // A: tile t visible to all; every wave is done with tile t-1
        if (t + 3 < ntile) stage(t + 3);
        const char* tb = smem + (t & (ES_NSTAGE - 1)) * ES_TILEB;
        // ---- scores of keys 16 kt + fl over dims 256 hf ..: rows 4 fg + i of D
        {
            const int r = 16 * kt + fl;
            const char* rp = tb + r * ES_ROWB;
            f32x4 d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0};
            bf16x8 ef[8];
#pragma unroll
            for (int s = 0; s < 8; s++) {
                const int c = 32 * hf + 4 * s + fg;
                ef[s] = *reinterpret_cast<const bf16x8*>(rp + ((c ^ (r & 15)) << 4));
            }
            __builtin_amdgcn_sched_barrier(0);   // all eight reads in flight before the first MFMA
#pragma unroll
            for (int s = 0; s < 8; s += 2) {
                d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[s], ef[s], d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[s + 1], ef[s + 1], d1, 0, 0, 0);
            }
            // rows 0-7 (lane groups 0, 1) carry hi(qe), rows 8-15 (groups 2, 3) lo(qe): add lane and lane ^ 32
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = es_sum32(d0[i] + d1[i]);
            if (fg < 2) {
#pragma unroll
                for (int i = 0; i < 4; i++) sc[(hf * ES_H + 4 * fg + i) * ES_SCP + 16 * kt + fl] = v[i];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // B: all 8 x 32 partial scores are in LDS
        // ---- online softmax: lane -> head fl & 7, keys kb .. kb + 7 of the tile (identical in the four waves)
        bf16x8 pa;
        {
            const int h = fl & 7;
            const float* s0 = sc + h * ES_SCP + kb;
            const float* s1 = sc + (ES_H + h) * ES_SCP + kb;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(s0), a1 = *reinterpret_cast<const f32x4*>(s0 + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(s1), b1 = *reinterpret_cast<const f32x4*>(s1 + 4);
            float sv[8];
            float tmax = -INFINITY;
            const int key0 = t * ES_TK + kb;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const float x = (u < 4) ? a0[u & 3] + b0[u & 3] : a1[u & 3] + b1[u & 3];
                sv[u] = (key0 + u < S) ? x : -INFINITY;
                tmax = fmaxf(tmax, sv[u]);
            }
            tmax = xrow_max(tmax);   // over the four lane groups: all 32 keys of the tile
            const float m_new = fmaxf(m_run, tmax);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // first tile: exp2(-inf) = 0
            float ps = 0.0f;
            const bool lo = fl >= 8;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const float p = __builtin_amdgcn_exp2f(sv[u] - m_new);   // masked key: exp2(-inf) = 0
                ps += p;
                const bf16 ph = (bf16)p;
                pa[u] = lo ? (bf16)(p - (float)ph) : ph;
            }
            l_run = l_run * alpha + ps;
            m_run = m_new;
            // the accumulators hold rows 4 fg + i = heads 4 (fg & 1) + i: fetch their factors through the wave's LDS words
            if (lane < 8) wx[wave * 16 + lane] = alpha;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(wx + wave * 16 + 4 * (fg & 1));
#pragma unroll
            for (int e = 0; e < 8; e++) {
#pragma unroll
                for (int i = 0; i < 4; i++) acc[e][i] *= a4[i];
            }
        }
        // ---- output: dims 128 wave + 8 fl + e, contraction over the tile's 32 keys
        {
            wh_u32x4 blk[8];
            const int slot = (16 * wave + fl);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int r = kb + j;
                blk[j] = *reinterpret_cast<const wh_u32x4*>(tb + r * ES_ROWB + ((slot ^ (r & 15)) << 4));
            }
#pragma unroll
            for (int e = 0; e < 8; e++) {
                wh_u32x4 op;
#pragma unroll
                for (int dq = 0; dq < 4; dq++) {
                    const unsigned ka = blk[2 * dq][e >> 1], kbv = blk[2 * dq + 1][e >> 1];
                    op[dq] = (e & 1) ? __builtin_amdgcn_perm(kbv, ka, 0x07060302u) : __builtin_amdgcn_perm(kbv, ka, 0x05040100u);
                }
                bf16x8 ob;
                __builtin_memcpy(&ob, &op, 16);
                acc[e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, ob, acc[e], 0, 0, 0);
            }
        }
    }
    // ---- normalise and store: hi + lo rows, 1 / sum p of the row's head
    {
        const float l_tot = xrow_sum(l_run);
        if (lane < 8) wx[wave * 16 + lane] = 1.0f / l_tot;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const f32x4 inv4 = *reinterpret_cast<const f32x4*>(wx + wave * 16 + 4 * (fg & 1));
        float o[4][8];
#pragma unroll
        for (int e = 0; e < 8; e++)
#pragma unroll
            for (int i = 0; i < 4; i++) o[i][e] = es_sum32(acc[e][i]) * inv4[i];
        if (fg < 2) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int k = (4 * fg + i) * ES_D + 128 * wave + 8 * fl;
                bf16x8 ov;
#pragma unroll
                for (int e = 0; e < 8; e++) ov[e] = (bf16)o[i][e];
                *reinterpret_cast<bf16x8*>(out + ((long)(k >> 5) * mpad + b) * 32 + (k & 31)) = ov;
            }
        }
    }
}

}  // namespace

bool wh_cross_es_geometry(int d, int n_heads, int S) { return d == ES_D && n_heads == ES_H && S >= 3 * ES_TK; }

void wh_launch_dec_cross_attn_es(hipStream_t s, const float* qe, const void* E, void* out, int S, int B, int mpad, bool stream_nt) {
    static const int nt_env = [] { const char* e = getenv("WH_CROSS_NT"); return e ? atoi(e) : -1; }();
    if (nt_env >= 0) stream_nt = nt_env != 0;
    if (stream_nt) {
        wh_ensure_dyn_lds((const void*)k_dec_cross_attn_es<2>, ES_LDS);
        hipLaunchKernelGGL(k_dec_cross_attn_es<2>, dim3(B), dim3(256), ES_LDS, s, qe, (const bf16*)E, (bf16*)out, S, mpad);
    } else {
        wh_ensure_dyn_lds((const void*)k_dec_cross_attn_es<0>, ES_LDS);
        hipLaunchKernelGGL(k_dec_cross_attn_es<0>, dim3(B), dim3(256), ES_LDS, s, qe, (const bf16*)E, (bf16*)out, S, mpad);
    }
}
