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

constexpr int ES_D = 512, ES_H = 8, ES_TK = 32;
constexpr int ES_ROWB = ES_D * 2;                      // bytes per key row
constexpr int ES_TILEB = ES_TK * ES_ROWB;              // 32 KiB
constexpr int ES_SCP = 36;                             // floats per (dim half, head) row of the score exchange (32 keys + pad)
constexpr int es_lds(int nstage, bool persist) { return nstage * ES_TILEB + 2 * 2 * ES_H * ES_SCP * 4 + (persist ? ES_H * ES_D * 4 : 0); }

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
// NSTAGE ring slots: NSTAGE - 1 tiles in flight while one is consumed (4: 130 KiB, one workgroup per CU; 2: 66 KiB, two per CU).
// PERSIST: a workgroup walks clips blockIdx.x, + gridDim.x, ... with ONE tile sequence over all of them — the ring keeps
// streaming across a clip boundary (the next clip's first tiles are in flight while this clip's last are consumed) and the
// next clip's expanded queries arrive through LDS (a 16 KiB DMA issued a clip ahead), so a CU's stream never drains between
// clips; without it every workgroup pays its own start-up (query loads, first-tile latency) and its tail.
template <int AUX, int NSTAGE, bool PERSIST, int ABL = 0>   // ABL (tools/es_bench.hip only): 1 = ring, waits and barriers only
__global__ __launch_bounds__(256, 1) void k_dec_cross_attn_es(const float* __restrict__ qe, const bf16* __restrict__ E,
                                                                                bf16* __restrict__ out, int S, int mpad, int B) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sc = reinterpret_cast<float*>(smem + NSTAGE * ES_TILEB);   // [2 tiles][2 dim halves][8 heads][ES_SCP]
    float* Qs = sc + 2 * 2 * ES_H * ES_SCP;                               // PERSIST: [8][512] f32, the next clip's expanded queries
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fl = lane & 15, fg = lane >> 4;
    const int ntile = (S + ES_TK - 1) / ES_TK;
    const int hf = wave & 1, kt = wave >> 1;
    const int G = gridDim.x;
    const int n_my = PERSIST ? (B - (int)blockIdx.x + G - 1) / G : 1;   // clips of this workgroup
    const int total = n_my * ntile;                                      // tiles of this workgroup
    static_assert(NSTAGE >= 3, "the pipeline holds two tiles (scored / consumed) and needs at least one in flight");
    constexpr int LA = NSTAGE - 1;   // tiles staged ahead of the one being consumed

    // ---- the ring: wave w brings rows 8 w .. 8 w + 7 of a tile, one wave-instruction = one 1 KiB key row.  Tiles are staged
    // strictly in sequence, so the (clip, tile-in-clip) of the next one to stage is carried along instead of divided out.
    int st_clip = blockIdx.x, st_t = 0, st_slot = 0;
    int voff[8];   // byte offset of this lane's 16 bytes of row 8 wave + j inside a tile's 32 KiB of E (the same for every tile: 32 % 16 == 0)
#pragma unroll
    for (int j = 0; j < 8; j++) voff[j] = (wave * 8 + j) * ES_ROWB + ((lane ^ ((wave * 8 + j) & 15)) << 4);
    auto stage_next = [&]() {
        char* base = smem + st_slot * ES_TILEB;
        const char* Et = reinterpret_cast<const char*>(E) + ((long)st_clip * S + (long)st_t * ES_TK) * ES_ROWB;   // wave-uniform
        if (st_t * ES_TK + ES_TK <= S) {
#pragma unroll
            for (int j = 0; j < 8; j++) es_glds16<AUX>(Et + voff[j], base + (wave * 8 + j) * ES_ROWB);
        } else {   // the clip's last tile: rows past the end re-read the last key (finite; their scores are masked)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int r = wave * 8 + j;
                const int key = min(st_t * ES_TK + r, S - 1);
                es_glds16<AUX>(reinterpret_cast<const char*>(E) + ((long)st_clip * S + key) * ES_ROWB + ((lane ^ (r & 15)) << 4), base + r * ES_ROWB);
            }
        }
        st_slot = st_slot + 1 == NSTAGE ? 0 : st_slot + 1;
        if (++st_t == ntile) { st_t = 0; st_clip += G; }
    };
    // expanded queries of clip `clip` into Qs: 16 KiB, four wave-instructions per wave
    auto stage_q = [&](int clip) {
        const float* src = qe + (long)clip * (ES_H * ES_D);
#pragma unroll
        for (int j = 0; j < 4; j++) es_glds16<0>(src + ((wave * 4 + j) * 64 + lane) * 4, reinterpret_cast<char*>(Qs) + (wave * 4 + j) * 1024);
    };

    // ---- expanded queries of this wave's dim half as the MFMA row operand: row fl -> head fl & 7, rows 8..15 the remainders
    bf16x8 qa[8];
    f32x4 qraw[16];
    auto build_qa = [&]() {
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
    };
    auto qa_from_lds = [&]() {
        const float* qp = Qs + (fl & 7) * ES_D + 256 * hf + 8 * fg;
#pragma unroll
        for (int s = 0; s < 8; s++) {
            qraw[2 * s] = *reinterpret_cast<const f32x4*>(qp + 32 * s);
            qraw[2 * s + 1] = *reinterpret_cast<const f32x4*>(qp + 32 * s + 4);
        }
        build_qa();
    };
    if constexpr (PERSIST) {
        stage_q(blockIdx.x);
#pragma unroll
        for (int t = 0; t < LA; t++)
            if (t < total) stage_next();
        if (total >= LA) es_wait_vm<8 * LA>(); else es_wait_vm<0>();   // this wave's share of the queries (issued first: vmcnt retires in order)
        __builtin_amdgcn_s_barrier();                                    // ... and everybody else's
        qa_from_lds();
    } else {
        // (issued by hand: the compiler would sink plain loads below the ring's first stages and then wait for everything)
        const float* qp = qe + ((long)blockIdx.x * ES_H + (fl & 7)) * ES_D + 256 * hf + 8 * fg;
#define WH_ES_QLD(I, OFF) asm volatile("global_load_dwordx4 %0, %1, off offset:" #OFF : "=&v"(qraw[I]) : "v"(qp) : "memory")
        WH_ES_QLD(0, 0);    WH_ES_QLD(1, 16);   WH_ES_QLD(2, 128);  WH_ES_QLD(3, 144);
        WH_ES_QLD(4, 256);  WH_ES_QLD(5, 272);  WH_ES_QLD(6, 384);  WH_ES_QLD(7, 400);
        WH_ES_QLD(8, 512);  WH_ES_QLD(9, 528);  WH_ES_QLD(10, 640); WH_ES_QLD(11, 656);
        WH_ES_QLD(12, 768); WH_ES_QLD(13, 784); WH_ES_QLD(14, 896); WH_ES_QLD(15, 912);
#undef WH_ES_QLD
#pragma unroll
        for (int t = 0; t < LA; t++) stage_next();
        // the 16 query loads were issued first: done when at most the 8 LA ring loads are outstanding
        asm volatile("s_waitcnt vmcnt(%16)"
                     : "+v"(qraw[0]), "+v"(qraw[1]), "+v"(qraw[2]), "+v"(qraw[3]), "+v"(qraw[4]), "+v"(qraw[5]), "+v"(qraw[6]), "+v"(qraw[7]),
                       "+v"(qraw[8]), "+v"(qraw[9]), "+v"(qraw[10]), "+v"(qraw[11]), "+v"(qraw[12]), "+v"(qraw[13]), "+v"(qraw[14]), "+v"(qraw[15])
                     : "n"(8 * LA) : "memory");
        build_qa();
    }

    // ---- scores of tile (slot `sl`) for keys 16 kt + fl over dims 256 hf ..: rows 4 fg + i of D; partials to sc buffer `buf`
    auto score_reads = [&](int sl, bf16x8 (&ef)[8]) {
        const int r = 16 * kt + fl;
        const char* rp = smem + sl * ES_TILEB + r * ES_ROWB;
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const int c = 32 * hf + 4 * s + fg;
            ef[s] = *reinterpret_cast<const bf16x8*>(rp + ((c ^ (r & 15)) << 4));
        }
    };
    auto score_mfma = [&](const bf16x8 (&ef)[8], int buf) {
        f32x4 d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0};
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
            float* dst = sc + buf * (2 * ES_H * ES_SCP) + (hf * ES_H + 4 * fg) * ES_SCP + 16 * kt + fl;
#pragma unroll
            for (int i = 0; i < 4; i++) dst[i * ES_SCP] = v[i];
        }
    };

    f32x4 acc[8];
    float m_run = -INFINITY, l_run = 0.0f;
    const int kb = 16 * (fg & 1) + 8 * (fg >> 1);   // first key (within a tile) of this lane's contraction slots
    int clip = blockIdx.x, t = 0, slot = 0;          // the tile being consumed: tile t of `clip`, ring slot `slot`

    // Software pipeline, one barrier per tile: iteration g computes the SCORES of tile g + 1 and the softmax + output of tile g,
    // so the LDS round trips and MFMA chains of the two halves overlap inside a wave (one wave per SIMD: nobody else hides them)
    // and the score exchange needs no barrier of its own — the barrier at the top of iteration g + 1 publishes it.
    // Tile g + 1 must therefore have landed at the top of iteration g: LA - 1 tiles stay in flight.
    {   // scores of tile 0
        bf16x8 ef[8];
        if (total > 1) es_wait_vm<8 * (LA - 1)>(); else es_wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if constexpr (!(ABL & 1)) { score_reads(0, ef); score_mfma(ef, 0); }
    }
#pragma unroll
    for (int e = 0; e < 8; e++) acc[e] = f32x4{0, 0, 0, 0};
    for (int g = 0; g < total; g++) {
        const bool more = g + 1 < total;
        if (more) {   // tile g + 1 has landed (this wave's share); the younger tiles stay in flight.  Conservative where other loads
                      // (the next clip's queries) are among the younger ones: the count then also covers a few loads of tile g + 2.
            const int newer = min(LA - 2, total - 2 - g);
            if (newer >= 1) es_wait_vm<8>();
            else es_wait_vm<0>();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // tile g + 1 and the scores of tile g visible to all; every wave is done with tile g - 1
        if (g + LA < total) stage_next();
        if (PERSIST && t == 0 && clip + G < B) stage_q(clip + G);
        const int nslot = slot + 1 == NSTAGE ? 0 : slot + 1;
        if constexpr (ABL & 1) {
            slot = nslot;
            if (++t == ntile) { t = 0; clip += G; }
            continue;
        }
        // the next clip's first tile is scored with the next clip's queries (in Qs since a clip ago)
        if (PERSIST && t == ntile - 1 && more) qa_from_lds();
        const char* tb = smem + slot * ES_TILEB;
        // ---- every LDS read of this iteration up front: score operands of tile g + 1, scores of tile g, the 8 x 8 blocks of tile g
        bf16x8 ef[8];
        if (more) score_reads(nslot, ef);
        const int h = fl & 7;
        const float* s0 = sc + (g & 1) * (2 * ES_H * ES_SCP) + h * ES_SCP + kb;
        const float* s1 = s0 + ES_H * ES_SCP;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(s0), a1 = *reinterpret_cast<const f32x4*>(s0 + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(s1), b1 = *reinterpret_cast<const f32x4*>(s1 + 4);
        wh_u32x4 blk[8];
        {
            const int cs = (16 * wave + fl);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int r = kb + j;
                blk[j] = *reinterpret_cast<const wh_u32x4*>(tb + r * ES_ROWB + ((cs ^ (r & 15)) << 4));
            }
        }
        // ---- online softmax of tile g: lane -> head fl & 7, keys kb .. kb + 7 of the tile (identical in the four waves)
        bf16x8 pa;
        {
            float sv[8];
            float tmax = -INFINITY;
            const int key0 = t * ES_TK + kb;
#pragma unroll
            for (int u = 0; u < 8; u++) sv[u] = (u < 4) ? a0[u & 3] + b0[u & 3] : a1[u & 3] + b1[u & 3];
            if (t == ntile - 1) {   // (wave-uniform) keys past the end of the clip
#pragma unroll
                for (int u = 0; u < 8; u++) sv[u] = (key0 + u < S) ? sv[u] : -INFINITY;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) tmax = fmaxf(tmax, sv[u]);
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
            // the accumulators hold rows 4 fg + i = heads 4 (fg & 1) + i; head h's factor sits in lane h: through SGPRs (v_readlane),
            // and only when some running maximum moved (wave-uniform)
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
                float ah[8];
#pragma unroll
                for (int q = 0; q < 8; q++) ah[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, alpha), q));
                const bool up = fg & 1;
                const float a4[4] = {up ? ah[4] : ah[0], up ? ah[5] : ah[1], up ? ah[6] : ah[2], up ? ah[7] : ah[3]};
#pragma unroll
                for (int e = 0; e < 8; e++) {
#pragma unroll
                    for (int i = 0; i < 4; i++) acc[e][i] *= a4[i];
                }
            }
        }
        // ---- scores of tile g + 1 (independent of everything above: fills the matrix pipe while the VALU transposes)
        if (more) score_mfma(ef, (g + 1) & 1);
        // ---- output of tile g: dims 128 wave + 8 fl + e, contraction over the tile's 32 keys
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
        slot = nslot;
        if (++t < ntile) continue;
        // ---- the clip ends: normalise and store — hi + lo rows, 1 / sum p of the row's head
        {
            const float inv = 1.0f / xrow_sum(l_run);
            float ih[8];
#pragma unroll
            for (int q = 0; q < 8; q++) ih[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, inv), q));
            const bool up = fg & 1;
            const float inv4[4] = {up ? ih[4] : ih[0], up ? ih[5] : ih[1], up ? ih[6] : ih[2], up ? ih[7] : ih[3]};
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
                    *reinterpret_cast<bf16x8*>(out + ((long)(k >> 5) * mpad + clip) * 32 + (k & 31)) = ov;
                }
            }
        }
        // the next clip starts from nothing
#pragma unroll
        for (int e = 0; e < 8; e++) acc[e] = f32x4{0, 0, 0, 0};
        m_run = -INFINITY;
        l_run = 0.0f;
        t = 0;
        clip += G;
    }
}

}  // namespace

bool wh_cross_es_geometry(int d, int n_heads, int S) { return d == ES_D && n_heads == ES_H && S >= 4 * ES_TK; }

void wh_launch_dec_cross_attn_es(hipStream_t s, const float* qe, const void* E, void* out, int S, int B, int mpad, bool stream_nt) {
    static const int nt_env = [] { const char* e = getenv("WH_CROSS_NT"); return e ? atoi(e) : -1; }();
    static const int nstage = [] { const char* e = getenv("WH_ES_NSTAGE"); return e ? atoi(e) : 4; }();     // (A/B runs)
    static const int persist = [] { const char* e = getenv("WH_ES_PERSIST"); return e ? atoi(e) : 1; }();   // (A/B runs) 0: one workgroup per clip
    static const int n_cus = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256; return n; }();
    if (nt_env >= 0) stream_nt = nt_env != 0;
#define WH_ES_LAUNCH(AUX_, NS_, P_, GRID_, ...)                                                                                        \
    do {                                                                                                                            \
        wh_ensure_dyn_lds((const void*)k_dec_cross_attn_es<AUX_, NS_, P_>, es_lds(NS_, P_));                                        \
        hipLaunchKernelGGL((k_dec_cross_attn_es<AUX_, NS_, P_, ##__VA_ARGS__>), dim3(GRID_), dim3(256), es_lds(NS_, P_), s, qe, (const bf16*)E, (bf16*)out, S, mpad, B); \
    } while (0)
    // persistent form: one workgroup per CU walks its clips (only worth it when a CU gets more than one)
#ifdef WH_ES_BENCH
    if (getenv("WH_ES_ABL")) { WH_ES_LAUNCH(2, 4, true, n_cus, 1); return; }
#endif
    if (persist && B > n_cus) {
        if (nstage == 3) { if (stream_nt) WH_ES_LAUNCH(2, 3, true, n_cus); else WH_ES_LAUNCH(0, 3, true, n_cus); }
        else { if (stream_nt) WH_ES_LAUNCH(2, 4, true, n_cus); else WH_ES_LAUNCH(0, 4, true, n_cus); }
        return;
    }
    if (nstage == 3) { if (stream_nt) WH_ES_LAUNCH(2, 3, false, B); else WH_ES_LAUNCH(0, 3, false, B); }
    else { if (stream_nt) WH_ES_LAUNCH(2, 4, false, B); else WH_ES_LAUNCH(0, 4, false, B); }
#undef WH_ES_LAUNCH
}
