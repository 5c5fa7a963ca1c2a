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
//   output  C[m][dim] = sum_key  P[m][key]  * E[key][dim]     mfma 16x16x32 bf16, rows m = {bf16(p_h) : h} ++ zeros
//
// The expanded queries are split into a bf16 head and a bf16 remainder (qe = hi + lo to ~16 mantissa bits): the 8 heads fill
// only half of a 16-row MFMA tile, the other half carries the remainders for free, so the scores see no bf16 rounding of the
// query side; the only rounded operand is E, which the projected form rounds as its GEMM operand as well.  The probabilities go
// in as bf16 (their remainder would vanish in the bf16 rounding of the output).
//
// One workgroup per CU (persistent, walks its clips): four computing waves + loader wave(s).  E streams global -> LDS on the
// LDS-DMA path (global_load_lds_dwordx4, no registers) through a ring of four 32-key tiles (32 KiB each), every piece issued by
// a loader wave; one barrier per tile; iteration g (tile g + 1 landed, two tiles in flight behind it):
//   scores of tile g + 1: wave w -> keys 16 (w / 2) .. + 15, dims 256 (w % 2) .. + 255: 8 ds_read_b128 + 8 MFMA, hi + lo rows added
//           across lanes (v_permlane32_swap), partial scores to the LDS exchange buffer of tile g + 1
//   softmax of tile g (every wave, identically — each needs the whole 16 x 32 probability operand): the two dim halves of the
//           8 x 32 scores added, running max / sum per head, operand built in registers, accumulators rescaled when a maximum
//           moved (factors through v_readlane)
//   output of tile g: wave w -> dims 128 w .. + 127: a lane reads 8 keys x 8 dims as 8 ds_read_b128 and transposes the 8 x 8 block
//           in registers (32 v_perm_b32) into the 8 key-contiguous column operands; 8 MFMA
// Bank conflicts: LDS is written linearly by the DMA, so the swizzle is on the source side — LDS chunk p of tile row r holds
// dim-chunk p ^ (r & 15); both read patterns then hit 16 distinct 16-byte slots per ds_read_b128 service group (the key order
// inside a 32-deep contraction step is (fg & 1) * 16 + (fg >> 1) * 8 + j for the same reason).
// No cross-wave merge at the end of a clip: every wave has seen every key and owns its own 128 output dims.
//
// Measured (MI355X, 1024 clips per launch, inside bench.py's step): 250 us = 6.3 TB/s = 0.79 of the HBM roof; the ring alone
// (no arithmetic) streams at 6.7 TB/s = 236 us.  The difference is not the computing waves' instruction issue (~1,630 cycles per tile
// against ~1,650 for the stream at the ~1.4 GHz the chip holds here): a form with a quarter of that work moved to a sixth wave
// (tools/cross_es_smw.patch.txt) runs no faster; the LDS-DMA pieces land more slowly while the computing waves read LDS, and the ring
// cannot keep more of them in flight.  Records: DESIGN.md section 5d, tools/cross_es2_proto.hip.txt (64-key steps, softmax shared through
// LDS with three barriers per step: no faster either).
#include <stdlib.h>

#include <algorithm>

#include "wh_common.h"
#include "wh_kernels.h"

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int ES_D = 512, ES_H = 8, ES_TK = 32, ES_NSTAGE = 4;
constexpr int ES_ROWB = ES_D * 2;                      // bytes per key row
constexpr int ES_TILEB = ES_TK * ES_ROWB;              // 32 KiB
constexpr int ES_SCP = 36;                             // floats per (dim half, head) row of the score exchange (32 keys + pad)
constexpr int ES_SCB = 4 * ES_H * ES_SCP;             // floats per score-exchange buffer: [dim half][hi | lo of the query][head][ES_SCP]
constexpr int ES_LDS = ES_NSTAGE * ES_TILEB + 2 * ES_SCB * 4 + ES_H * ES_D * 4;   // ring + score exchange + next queries = 153 KiB

template <int N> __device__ __forceinline__ void es_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int AUX>
__device__ __forceinline__ void es_glds16(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, AUX);
}

__device__ __forceinline__ float es_sum32(float v) {   // v(lane) + v(lane ^ 32), in every lane (tools/cross_es2_proto.hip.txt)
    const wh_u32x2 t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(t.x) + __uint_as_float(t.y);
}

__device__ __forceinline__ float es_ror8(float v) {    // v of lane ^ 8 (same 16-lane row): DPP row_ror:8
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
}

// qe : [B][8][512] f32 expanded queries (natural-log score units)        E: [B][e_rows][512] bf16 encoder states (final LayerNorm applied),
// e_rows >= S: the clips' states sit e_rows rows apart (20 rows of padding take the lock-step streams off a common 4 KiB phase, wh_api.cpp)
// out: ctx as the decode GEMM's operand, slab layout [8 * 512 / 32][mpad][32] bf16, column h * 512 + dim
//
// Roles.  Waves 0-3 compute; waves 4 .. 3 + NL only load: every LDS-DMA piece of the ring and of the query prefetch is issued by
// a loader wave.  An LDS-DMA wave-instruction costs its issuer 60-185 cycles when it sits among ds_reads and MFMAs
// (MI355X_MICROARCH.md, "LDS-DMA piece issue cost"): with the four computing waves issuing their own eight pieces per tile the
// kernel ran at 253 us per 1024-clip launch against 236 us for the ring alone.  All waves meet at the same barriers.
//
// A workgroup walks clips blockIdx.x, + gridDim.x, ... with ONE tile sequence over all of them: the ring keeps streaming across a
// clip boundary (the next clip's first tiles are in flight while this clip's last are consumed) and the next clip's expanded
// queries arrive through LDS a clip ahead, so a CU's stream never drains between clips.
//
// Software pipeline, one barrier per tile: iteration g computes the SCORES of tile g + 1 and the softmax + output of tile g, so
// the LDS round trips and MFMA chains of the two halves overlap inside a wave and the score exchange needs no barrier of its
// own.  Tile g + 1 must therefore have landed at the top of iteration g: NSTAGE - 2 tiles stay in flight.
template <int AUX, int NL, int ABL = 0>   // ABL (tools/es_bench.hip only): 1 = ring, waits and barriers only; 2 = phase stamps (s_memtime) of workgroup 0 into dbg
__global__ __launch_bounds__(256 + 64 * NL, 1) void k_dec_cross_attn_es(const float* __restrict__ qe, const bf16* __restrict__ E,
                                                                         bf16* __restrict__ out, int S, int e_rows, int mpad, int B, unsigned long long* dbg) {
    constexpr int NSTAGE = ES_NSTAGE, LA = NSTAGE - 1;   // LA tiles staged ahead of the one being consumed
    static_assert(NSTAGE == 4 && (NL == 1 || NL == 2), "ring of four 32-key slots; one or two loader waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sc = reinterpret_cast<float*>(smem + NSTAGE * ES_TILEB);   // [2 tiles][ES_SCB]
    float* Qs = sc + 2 * ES_SCB;                               // [8][512] f32: the next clip's expanded queries
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntile = (S + ES_TK - 1) / ES_TK;
    const int G = gridDim.x;
    const int n_my = (B - (int)blockIdx.x + G - 1) / G;   // clips of this workgroup
    const int total = n_my * ntile;                        // tiles of this workgroup

    if (wave >= 4) {
        // ================================ loader ================================
        constexpr int PPT = ES_TK / NL;          // pieces (1 KiB key rows) per loader wave and tile
        constexpr int QPP = 16 / NL;             // pieces of a clip's 16 KiB of queries per loader wave
        const int lw = wave - 4;
        int voff[PPT];   // byte offset of this lane's 16 bytes of row lw * PPT + j inside a tile (the same for every tile: 32 % 16 == 0)
#pragma unroll
        for (int j = 0; j < PPT; j++) voff[j] = (lw * PPT + j) * ES_ROWB + ((lane ^ ((lw * PPT + j) & 15)) << 4);
        int st_clip = blockIdx.x, st_t = 0, st_slot = 0;   // tiles are staged strictly in sequence
        auto stage_next = [&]() {
            char* base = smem + st_slot * ES_TILEB;
            const char* Et = reinterpret_cast<const char*>(E) + ((long)st_clip * e_rows + (long)st_t * ES_TK) * ES_ROWB;   // wave-uniform
            if (st_t * ES_TK + ES_TK <= S) {
#pragma unroll
                for (int j = 0; j < PPT; j++) es_glds16<AUX>(Et + voff[j], base + (lw * PPT + j) * ES_ROWB);
            } else {   // the clip's last tile: rows past the end re-read the last key (finite; their scores are masked)
#pragma unroll
                for (int j = 0; j < PPT; j++) {
                    const int r = lw * PPT + j;
                    const int key = min(st_t * ES_TK + r, S - 1);
                    es_glds16<AUX>(reinterpret_cast<const char*>(E) + ((long)st_clip * e_rows + key) * ES_ROWB + ((lane ^ (r & 15)) << 4), base + r * ES_ROWB);
                }
            }
            st_slot = st_slot + 1 == NSTAGE ? 0 : st_slot + 1;
            if (++st_t == ntile) { st_t = 0; st_clip += G; }
        };
        auto stage_q = [&](int clip) {
            const float* src = qe + (long)clip * (ES_H * ES_D);
#pragma unroll
            for (int j = 0; j < QPP; j++) es_glds16<0>(src + ((lw * QPP + j) * 64 + lane) * 4, reinterpret_cast<char*>(Qs) + (lw * QPP + j) * 1024);
        };
        stage_q(blockIdx.x);
#pragma unroll
        for (int t = 0; t < LA; t++)
            if (t < total) stage_next();
        // vmcnt retires in issue order (and holds at most 63): "all but the last two tiles' pieces" covers the queries and tile 0
        if (total >= LA) es_wait_vm<(PPT * (LA - 1) < 63 ? PPT * (LA - 1) : 63)>(); else es_wait_vm<0>();
        __builtin_amdgcn_s_barrier();   // P1: the first clip's queries are in Qs
        __builtin_amdgcn_s_barrier();   // P2: tile 0 is in the ring
        int clip = blockIdx.x, t = 0;
        unsigned long long lt[4] = {0, 0, 0, 0};
        for (int g = 0; g < total; g++) {
            unsigned long long c0 = 0, c1 = 0, c2 = 0;
            if constexpr (ABL & 2) c0 = __builtin_amdgcn_s_memtime();
            if (g + 1 < total) {   // tile g + 1 has landed; the younger tiles stay in flight.  Conservative where the next clip's
                                   // queries are among the younger loads: the count then also covers a few pieces of tile g + 2.
                if (total - 2 - g >= LA - 2) es_wait_vm<PPT*(LA - 2)>(); else es_wait_vm<0>();
            }
            if constexpr (ABL & 2) c1 = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_barrier();
            if constexpr (ABL & 2) c2 = __builtin_amdgcn_s_memtime();
            if (g + LA < total) stage_next();
            if (t == 0 && clip + G < B) stage_q(clip + G);   // Qs was read (if at all) before this barrier
            if (++t == ntile) { t = 0; clip += G; }
            if constexpr (ABL & 2) { lt[0] += c1 - c0; lt[1] += c2 - c1; lt[2] += __builtin_amdgcn_s_memtime() - c2; lt[3] += 1; }
        }
        if constexpr (ABL & 2) {
            if (blockIdx.x == 0 && lane == 0 && dbg) { for (int i = 0; i < 4; i++) dbg[8 + 4 * (wave - 4) + i] = lt[i]; }
        }
        return;
    }

    // ================================ compute ================================
    const int fl = lane & 15, fg = lane >> 4;
    const int hf = wave & 1, kt = wave >> 1;
    // ---- expanded queries of this wave's dim half as the MFMA row operand: row fl -> head fl & 7, rows 0-7 the bf16 heads,
    // rows 8-15 the bf16 remainders (qe = hi + lo to ~16 mantissa bits)
    bf16x8 qa[8];
    auto qa_from_lds = [&]() {
        const float* qp = Qs + (fl & 7) * ES_D + 256 * hf + 8 * fg;
        const bool lo = fl >= 8;
#pragma unroll
        for (int s0 = 0; s0 < 8; s0 += 4) {   // eight reads in flight at a time (left alone, the compiler keeps two)
            f32x4 q[4][2];
#pragma unroll
            for (int s = 0; s < 4; s++) {
                q[s][0] = *reinterpret_cast<const f32x4*>(qp + 32 * (s0 + s));
                q[s][1] = *reinterpret_cast<const f32x4*>(qp + 32 * (s0 + s) + 4);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const float v = q[s][u >> 2][u & 3] * 1.44269504088896341f;   // scores in log2 units: p = exp2(s - m)
                    const bf16 h = (bf16)v;
                    qa[s0 + s][u] = lo ? (bf16)(v - (float)h) : h;
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // ---- scores of the tile in slot `sl` for keys 16 kt + fl over dims 256 hf ..: rows 4 fg + i of D; partials to sc buffer `buf`
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
        // rows 0-7 (lane groups 0, 1) carry hi(qe), rows 8-15 (groups 2, 3) lo(qe): both go to the exchange buffer as they are (the
        // readers add the four partials of a score: two dim halves x {hi, lo}) — cheaper than adding lane and lane ^ 32 here
        float* dst = sc + buf * ES_SCB + ((hf * 2 + (fg >> 1)) * ES_H + 4 * (fg & 1)) * ES_SCP + 16 * kt + fl;
#pragma unroll
        for (int i = 0; i < 4; i++) dst[i * ES_SCP] = d0[i] + d1[i];
    };

    f32x4 acc[8];   // rows 4 fg + i: heads 4 fg + i in lane groups 0 and 1 (the P operand's rows 8-15 are zero)
    float m_run = -INFINITY, l_run = 0.0f;
    const int kb = 16 * (fg & 1) + 8 * (fg >> 1);   // first key (within a tile) of this lane's contraction slots
    int clip = blockIdx.x, t = 0, slot = 0;          // the tile being consumed: tile t of `clip`, ring slot `slot`

    __builtin_amdgcn_s_barrier();   // P1
    qa_from_lds();
    __builtin_amdgcn_s_barrier();   // P2
    if constexpr (!(ABL & 1)) {     // scores of tile 0
        bf16x8 ef[8];
        score_reads(0, ef);
        score_mfma(ef, 0);
    }
#pragma unroll
    for (int e = 0; e < 8; e++) acc[e] = f32x4{0, 0, 0, 0};
    unsigned long long ct[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int g = 0; g < total; g++) {
        const bool more = g + 1 < total;
        unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0;
        if constexpr (ABL & 2) c0 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // tile g + 1 and the scores of tile g visible to all; every wave is done with tile g - 1
        if constexpr (ABL & 2) c1 = __builtin_amdgcn_s_memtime();
        const int nslot = slot + 1 == NSTAGE ? 0 : slot + 1;
        if constexpr (ABL & 1) {
            slot = nslot;
            if (++t == ntile) { t = 0; clip += G; }
            continue;
        }
        // the next clip's first tile is scored with the next clip's queries (in Qs since a clip ago)
        if (t == ntile - 1 && more) qa_from_lds();
        const char* tb = smem + slot * ES_TILEB;
        // ---- every LDS read of this iteration up front, in the order of use (LDS returns in order, so each consumer waits only
        // for what it needs): scores of tile g (softmax), score operands of tile g + 1, the 8 x 8 blocks of tile g (output)
        // lanes fl and fl + 8 share a head (the operand's rows 8-15 are zero): each takes four of the lane group's eight keys
        const int h = fl & 7, kq = kb + 4 * (fl >> 3);
        const float* s0 = sc + (g & 1) * ES_SCB + h * ES_SCP + kq;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(s0), a1 = *reinterpret_cast<const f32x4*>(s0 + ES_H * ES_SCP);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(s0 + 2 * ES_H * ES_SCP), b1 = *reinterpret_cast<const f32x4*>(s0 + 3 * ES_H * ES_SCP);
        bf16x8 ef[8];
        if (more) score_reads(nslot, ef);
        wh_u32x4 blk[8];
        {
            const int cs = (16 * wave + fl);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int r = kb + j;
                blk[j] = *reinterpret_cast<const wh_u32x4*>(tb + r * ES_ROWB + ((cs ^ (r & 15)) << 4));
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the reads above the arithmetic below
        if constexpr (ABL & 2) c2 = __builtin_amdgcn_s_memtime();
        // ---- online softmax of tile g: lane -> head fl & 7, keys kb .. kb + 7 of the tile (identical in the four waves)
        bf16x8 pa;
        {
            float sv[4];
            float tmax = -INFINITY;
            const int key0 = t * ES_TK + kq;
#pragma unroll
            for (int u = 0; u < 4; u++) sv[u] = (a0[u] + a1[u]) + (b0[u] + b1[u]);
            if (t == ntile - 1) {   // (wave-uniform) keys past the end of the clip
#pragma unroll
                for (int u = 0; u < 4; u++) sv[u] = (key0 + u < S) ? sv[u] : -INFINITY;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) tmax = fmaxf(tmax, sv[u]);
            tmax = fmaxf(tmax, es_ror8(tmax));   // the head's other four keys of this lane group
            tmax = xrow_max(tmax);               // over the four lane groups: all 32 keys of the tile
            const float m_new = fmaxf(m_run, tmax);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // first tile: exp2(-inf) = 0
            float ps = 0.0f;
            float pv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                pv[u] = __builtin_amdgcn_exp2f(sv[u] - m_new);   // masked key: exp2(-inf) = 0
                ps += pv[u];
            }
            // rows 0-7 carry bf16(p) of keys kb .. kb + 7: this lane's four and, through DPP, its partner's; rows 8-15 stay zero (a
            // remainder row would be lost in the bf16 rounding of the output anyway)
            {
                const bf16x2 p01 = {(bf16)pv[0], (bf16)pv[1]}, p23 = {(bf16)pv[2], (bf16)pv[3]};
                unsigned own0, own1;
                __builtin_memcpy(&own0, &p01, 4);
                __builtin_memcpy(&own1, &p23, 4);
                const unsigned oth0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)own0, 0x128, 0xF, 0xF, true);   // row_ror:8 = lane fl ^ 8 of the row
                const unsigned oth1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)own1, 0x128, 0xF, 0xF, true);
                const bool hi = fl < 8;
                const wh_u32x4 pw = {hi ? own0 : 0u, hi ? own1 : 0u, hi ? oth0 : 0u, hi ? oth1 : 0u};
                __builtin_memcpy(&pa, &pw, 16);
            }
            l_run = l_run * alpha + ps;
            m_run = m_new;
            // the accumulators hold rows 4 fg + i = heads 4 fg + i (fg < 2); head h's factor sits in lane h: through SGPRs (v_readlane),
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
        if constexpr (ABL & 2) { asm volatile("" :: "v"(pa) : "memory"); c3 = __builtin_amdgcn_s_memtime(); }
        // ---- scores of tile g + 1 (independent of everything above: fills the matrix pipe while the VALU transposes)
        if (more) score_mfma(ef, (g + 1) & 1);
        if constexpr (ABL & 2) { asm volatile("" ::: "memory"); c4 = __builtin_amdgcn_s_memtime(); }
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
        if constexpr (ABL & 2) {
            asm volatile("" :: "v"(acc[7]) : "memory");
            ct[0] += c1 - c0; ct[1] += c2 - c1; ct[2] += c3 - c2; ct[3] += c4 - c3; ct[4] += __builtin_amdgcn_s_memtime() - c4; ct[5] += 1;
        }
        slot = nslot;
        if (++t < ntile) continue;
        // ---- the clip ends: normalise and store (rows 0-7 = lane groups 0 and 1)
        {
            const float lh = l_run + es_ror8(l_run);   // the head's two key quartets
            const float inv = 1.0f / xrow_sum(lh);
            float ih[8];
#pragma unroll
            for (int q = 0; q < 8; q++) ih[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, inv), q));
            if (fg < 2) {
                const bool up = fg & 1;
                const float inv4[4] = {up ? ih[4] : ih[0], up ? ih[5] : ih[1], up ? ih[6] : ih[2], up ? ih[7] : ih[3]};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int k = (4 * fg + i) * ES_D + 128 * wave + 8 * fl;
                    bf16x8 ov;
#pragma unroll
                    for (int e = 0; e < 8; e++) ov[e] = (bf16)(acc[e][i] * inv4[i]);
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
    if constexpr (ABL & 2) {
        if (blockIdx.x == 0 && tid == 0 && dbg) { for (int i = 0; i < 6; i++) dbg[i] = ct[i]; }
    }
}



// ---- expanded queries: qe[m][h][j] = sum_t q[m][64 h + t] * wkT[h][j][t]  (t < 64) ------------------------------------------------
// The K projection moved to the query side: 8 small products [rows x 64] x [64 x 512] per launch.  One workgroup = 64 rows x one head
// x 128 columns; the weight tile is the MFMA row operand (a lane ends with 4 consecutive columns of one row: 16-byte stores), q goes
// in as bf16 hi + lo (two MFMAs per step) so the f32 query is not rounded.  16 KiB of q and 16 KiB of weights per workgroup.
__global__ __launch_bounds__(256) void k_dec_qexpand(const float* __restrict__ q, const bf16* __restrict__ wkT, float* __restrict__ qe,
                                                     int M, int d, int n_heads) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fl = lane & 15, fg = lane >> 4;
    const int h = blockIdx.y, n0 = blockIdx.z * 128;
    const int m = blockIdx.x * 64 + wave * 16 + fl, mc = min(m, M - 1);
    bf16x8 xh[2], xl[2];
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
        const float* qp = q + (long)mc * d + h * WH_HEAD_DIM + 32 * ks + 8 * fg;
        const f32x4 a = *reinterpret_cast<const f32x4*>(qp), b = *reinterpret_cast<const f32x4*>(qp + 4);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const float v = u < 4 ? a[u & 3] : b[u & 3];
            const bf16 hi = (bf16)v;
            xh[ks][u] = hi;
            xl[ks][u] = (bf16)(v - (float)hi);
        }
    }
    const bf16* wp = wkT + ((long)h * d + n0 + fl) * WH_HEAD_DIM + 8 * fg;
    bf16x8 wf[8][2];
#pragma unroll
    for (int nt = 0; nt < 8; nt++)
#pragma unroll
        for (int ks = 0; ks < 2; ks++) wf[nt][ks] = *reinterpret_cast<const bf16x8*>(wp + (long)nt * 16 * WH_HEAD_DIM + 32 * ks);
#pragma unroll
    for (int nt = 0; nt < 8; nt++) {
        f32x4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][ks], xl[ks], acc, 0, 0, 0);   // D rows = columns n (4 fg + r), D column = row m (fl)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][ks], xh[ks], acc, 0, 0, 0);
        }
        if (m < M) *reinterpret_cast<f32x4*>(qe + ((long)m * n_heads + h) * d + n0 + 16 * nt + 4 * fg) = acc;
    }
}


// =====================================================================================================================================
// WH_PREC_F16X3: the same attention with every operand as two fp16 limbs (wh_common.h).  The encoder states arrive as
//   E2 [B][e_rows][hi 512 | lo 512] fp16     (2 KiB per key row: the bytes of the f32 states, written by k_layernorm_es2)
// and the kernel is the one above with the key row read as 1,024 "dims": scores = QE . E_hi + QE . E_lo with the query operand
// repeated for the second plane; output = sum over (key, plane) of P x E_plane — a 16-key tile fills the 32 contraction slots of
// one MFMA (slot 8 fg + j <-> key 8 (fg >> 1) + j of plane fg & 1).  Both operand tiles carry a limb pair in their 16 rows
// (rows 0-7 the hi limbs of the 8 heads, rows 8-15 the lo limbs), so a product costs two fp16 MFMAs, not three, and even holds the
// lo.lo term; rows h and h + 8 are added where they leave the matrix core.  Per 32 KiB tile (16 keys) the work of a wave is what the
// bf16 kernel does per 32 KiB tile (32 keys): 8 + 8 MFMAs, 4 exp2, the same LDS reads — twice the time per clip for twice the bytes.
// =====================================================================================================================================
constexpr int E2_TK = 16;
constexpr int E2_ROWB = 2 * ES_D * 2;                  // bytes per key row: hi plane, lo plane
constexpr int E2_TILEB = E2_TK * E2_ROWB;              // 32 KiB
constexpr int E2_SCP = 20;                             // floats per (wave, head) row of the score exchange (16 keys + pad)
constexpr int E2_SCB = 4 * ES_H * E2_SCP;              // floats per score-exchange buffer: [wave = quarter of the 1,024 dims][head][E2_SCP]
constexpr int E2_LDS = ES_NSTAGE * E2_TILEB + 2 * E2_SCB * 4 + ES_H * ES_D * 4;   // ring + score exchange + next queries = 149 KiB

template <int AUX, int NL>
__global__ __launch_bounds__(256 + 64 * NL, 1) void k_dec_cross_attn_es2(const float* __restrict__ qe, const _Float16* __restrict__ E,
                                                                          h2* __restrict__ out, int S, int e_rows, int mpad, int B) {
    constexpr int NSTAGE = ES_NSTAGE, LA = NSTAGE - 1;
    static_assert(NSTAGE == 4 && (NL == 1 || NL == 2), "ring of four 16-key slots; one or two loader waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sc = reinterpret_cast<float*>(smem + NSTAGE * E2_TILEB);   // [2 tiles][E2_SCB]
    float* Qs = sc + 2 * E2_SCB;                                      // [8][512] f32: the next clip's expanded queries
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntile = (S + E2_TK - 1) / E2_TK;
    const int G = gridDim.x;
    const int n_my = (B - (int)blockIdx.x + G - 1) / G;
    const int total = n_my * ntile;

    if (wave >= 4) {
        // ================================ loader: 32 pieces of 1 KiB (half a key row) per tile ================================
        constexpr int PPT = 2 * E2_TK / NL;
        constexpr int QPP = 16 / NL;
        const int lw = wave - 4;
        int voff[PPT];   // byte offset of this lane's 16 bytes inside a tile: LDS chunk p of tile row r holds chunk p ^ r of the key row
#pragma unroll
        for (int j = 0; j < PPT; j++) {
            const int pi = lw * PPT + j, r = pi >> 1, p = (pi & 1) * 64 + lane;
            voff[j] = r * E2_ROWB + ((p ^ r) << 4);
        }
        int st_clip = blockIdx.x, st_t = 0, st_slot = 0;
        auto stage_next = [&]() {
            char* base = smem + st_slot * E2_TILEB;
            const char* Et = reinterpret_cast<const char*>(E) + ((long)st_clip * e_rows + (long)st_t * E2_TK) * E2_ROWB;   // wave-uniform
            if (st_t * E2_TK + E2_TK <= S) {
#pragma unroll
                for (int j = 0; j < PPT; j++) es_glds16<AUX>(Et + voff[j], base + (lw * PPT + j) * 1024);
            } else {   // the clip's last tile: rows past the end re-read the last key (finite; their scores are masked)
#pragma unroll
                for (int j = 0; j < PPT; j++) {
                    const int pi = lw * PPT + j, r = pi >> 1, p = (pi & 1) * 64 + lane;
                    const int key = min(st_t * E2_TK + r, S - 1);
                    es_glds16<AUX>(reinterpret_cast<const char*>(E) + ((long)st_clip * e_rows + key) * E2_ROWB + ((p ^ r) << 4), base + pi * 1024);
                }
            }
            st_slot = st_slot + 1 == NSTAGE ? 0 : st_slot + 1;
            if (++st_t == ntile) { st_t = 0; st_clip += G; }
        };
        auto stage_q = [&](int clip) {
            const float* src = qe + (long)clip * (ES_H * ES_D);
#pragma unroll
            for (int j = 0; j < QPP; j++) es_glds16<0>(src + ((lw * QPP + j) * 64 + lane) * 4, reinterpret_cast<char*>(Qs) + (lw * QPP + j) * 1024);
        };
        stage_q(blockIdx.x);
#pragma unroll
        for (int t = 0; t < LA; t++)
            if (t < total) stage_next();
        if (total >= LA) es_wait_vm<(PPT * (LA - 1) < 63 ? PPT * (LA - 1) : 63)>(); else es_wait_vm<0>();
        __builtin_amdgcn_s_barrier();   // P1: the first clip's queries are in Qs
        __builtin_amdgcn_s_barrier();   // P2: tile 0 is in the ring
        int clip = blockIdx.x, t = 0;
        for (int g = 0; g < total; g++) {
            if (g + 1 < total) {   // tile g + 1 has landed; the younger tiles stay in flight
                if (total - 2 - g >= LA - 2) es_wait_vm<PPT*(LA - 2)>(); else es_wait_vm<0>();
            }
            __builtin_amdgcn_s_barrier();
            if (g + LA < total) stage_next();
            if (t == 0 && clip + G < B) stage_q(clip + G);   // Qs was read (if at all) before this barrier
            if (++t == ntile) { t = 0; clip += G; }
        }
        return;
    }

    // ================================ compute ================================
    const int fl = lane & 15, fg = lane >> 4;
    const int hf = wave & 1;   // this wave's half of the 512 query dims (waves 0, 1: against the hi plane; 2, 3: the lo plane)
    // expanded queries as the MFMA row operand: row fl -> head fl & 7, rows 0-7 the fp16 hi limbs, rows 8-15 the lo limbs
    f16x8 qa[8];
    auto qa_from_lds = [&]() {
        const float* qp = Qs + (fl & 7) * ES_D + 256 * hf + 8 * fg;
        const bool lo = fl >= 8;
#pragma unroll
        for (int s0 = 0; s0 < 8; s0 += 4) {
            f32x4 q[4][2];
#pragma unroll
            for (int s = 0; s < 4; s++) {
                q[s][0] = *reinterpret_cast<const f32x4*>(qp + 32 * (s0 + s));
                q[s][1] = *reinterpret_cast<const f32x4*>(qp + 32 * (s0 + s) + 4);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const float v = q[s][u >> 2][u & 3] * 1.44269504088896341f;   // scores in log2 units: p = exp2(s - m)
                    const _Float16 h = (_Float16)v;
                    qa[s0 + s][u] = lo ? (_Float16)(v - (float)h) : h;
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // scores of the tile in slot `sl`: keys fl, the 256 "dims" 256 wave .. of the 1,024 (8 contraction steps)
    auto score_reads = [&](int sl, f16x8 (&ef)[8]) {
        const char* rp = smem + sl * E2_TILEB + fl * E2_ROWB;
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const int c = 32 * wave + 4 * s + fg;
            ef[s] = *reinterpret_cast<const f16x8*>(rp + ((c ^ fl) << 4));
        }
    };
    auto score_mfma = [&](const f16x8 (&ef)[8], int buf) {
        f32x4 d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 8; s += 2) {
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa[s], ef[s], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa[s + 1], ef[s + 1], d1, 0, 0, 0);
        }
        // rows 4 fg + i: the hi-limb rows of heads 0-7 in lane groups 0, 1, the lo-limb rows in groups 2, 3 (lane + 32): added here, so the
        // exchange buffer holds one partial per (wave, head, key) and the softmax adds four
        float* dst = sc + buf * E2_SCB + (wave * ES_H + 4 * (fg & 1)) * E2_SCP + fl;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float v = es_sum32(d0[i] + d1[i]);
            if (fg < 2) dst[i * E2_SCP] = v;
        }
    };

    f32x4 acc[8];   // rows 4 fg + i (hi-limb probability rows in groups 0, 1; lo-limb rows in groups 2, 3), column fl <-> dim 128 wave + 8 fl + e
    float m_run = -INFINITY, l_run = 0.0f;
    const int kb = 8 * (fg >> 1), pl = fg & 1;        // this lane group's contraction slots: keys kb .. kb + 7 of plane pl
    int clip = blockIdx.x, t = 0, slot = 0;

    __builtin_amdgcn_s_barrier();   // P1
    qa_from_lds();
    __builtin_amdgcn_s_barrier();   // P2
    {
        f16x8 ef[8];
        score_reads(0, ef);
        score_mfma(ef, 0);
    }
#pragma unroll
    for (int e = 0; e < 8; e++) acc[e] = f32x4{0, 0, 0, 0};
    for (int g = 0; g < total; g++) {
        const bool more = g + 1 < total;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // tile g + 1 and the scores of tile g visible to all; every wave is done with tile g - 1
        const int nslot = slot + 1 == NSTAGE ? 0 : slot + 1;
        if (t == ntile - 1 && more) qa_from_lds();   // the next clip's first tile is scored with the next clip's queries
        const char* tb = smem + slot * E2_TILEB;
        // every LDS read of this iteration up front, in the order of use
        const int h = fl & 7, kq = kb + 4 * (fl >> 3);   // lanes fl and fl + 8 share a head: each takes four of the lane group's eight keys
        const float* s0 = sc + (g & 1) * E2_SCB + h * E2_SCP + kq;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(s0), a1 = *reinterpret_cast<const f32x4*>(s0 + ES_H * E2_SCP);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(s0 + 2 * ES_H * E2_SCP), b1 = *reinterpret_cast<const f32x4*>(s0 + 3 * ES_H * E2_SCP);
        f16x8 ef[8];
        if (more) score_reads(nslot, ef);
        wh_u32x4 blk[8];
        {
            const int cs = pl * 64 + 16 * wave + fl;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int r = kb + j;
                blk[j] = *reinterpret_cast<const wh_u32x4*>(tb + r * E2_ROWB + ((cs ^ r) << 4));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- online softmax of tile g: lane -> head fl & 7, keys kq .. kq + 3 (lane groups fg and fg ^ 1 repeat each other)
        f16x8 pa;
        {
            float sv[4];
            float tmax = -INFINITY;
            const int key0 = t * E2_TK + kq;
#pragma unroll
            for (int u = 0; u < 4; u++) sv[u] = (a0[u] + a1[u]) + (b0[u] + b1[u]);
            if (t == ntile - 1) {
#pragma unroll
                for (int u = 0; u < 4; u++) sv[u] = (key0 + u < S) ? sv[u] : -INFINITY;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) tmax = fmaxf(tmax, sv[u]);
            tmax = fmaxf(tmax, es_ror8(tmax));
            tmax = xrow_max(tmax);               // over the lane groups: all 16 keys of the tile
            const float m_new = fmaxf(m_run, tmax);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            float ps = 0.0f;
            float pv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                pv[u] = __builtin_amdgcn_exp2f(sv[u] - m_new);
                ps += pv[u];
            }
            // the probability operand, rows 0-7 the hi limbs, rows 8-15 the lo limbs of keys kb .. kb + 7: a lane's own four keys are
            // slots 0-3 (fl < 8) or 4-7 (fl >= 8) of its row; the other four come from lane fl ^ 8, which holds the other limb row of
            // the same head — it sends the limb this row wants
            {
                typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
                _Float16 ph[4], pq[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { ph[u] = (_Float16)pv[u]; pq[u] = (_Float16)(pv[u] - (float)ph[u]); }
                const f16x2 h01 = {ph[0], ph[1]}, h23 = {ph[2], ph[3]}, l01 = {pq[0], pq[1]}, l23 = {pq[2], pq[3]};
                unsigned uh0, uh1, ul0, ul1;
                __builtin_memcpy(&uh0, &h01, 4); __builtin_memcpy(&uh1, &h23, 4);
                __builtin_memcpy(&ul0, &l01, 4); __builtin_memcpy(&ul1, &l23, 4);
                const bool hi = fl < 8;
                const unsigned send0 = hi ? ul0 : uh0, send1 = hi ? ul1 : uh1;
                const unsigned recv0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send0, 0x128, 0xF, 0xF, true);   // row_ror:8 = lane fl ^ 8
                const unsigned recv1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send1, 0x128, 0xF, 0xF, true);
                const wh_u32x4 pw = {hi ? uh0 : recv0, hi ? uh1 : recv1, hi ? recv0 : ul0, hi ? recv1 : ul1};
                __builtin_memcpy(&pa, &pw, 16);
            }
            l_run = l_run * alpha + ps;
            m_run = m_new;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {   // a running maximum moved: rescale (head q's factor sits in lane q)
                float ah[8];
#pragma unroll
                for (int q = 0; q < 8; q++) ah[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, alpha), q));
                const bool up = fg & 1;   // rows 4 fg + i belong to head 4 (fg & 1) + i in either limb half
                const float a4[4] = {up ? ah[4] : ah[0], up ? ah[5] : ah[1], up ? ah[6] : ah[2], up ? ah[7] : ah[3]};
#pragma unroll
                for (int e = 0; e < 8; e++) {
#pragma unroll
                    for (int i = 0; i < 4; i++) acc[e][i] *= a4[i];
                }
            }
        }
        if (more) score_mfma(ef, (g + 1) & 1);
        // ---- output of tile g: dims 128 wave + 8 fl + e, contraction over the tile's 16 keys x 2 planes
#pragma unroll
        for (int e = 0; e < 8; e++) {
            wh_u32x4 op;
#pragma unroll
            for (int dq = 0; dq < 4; dq++) {
                const unsigned ka = blk[2 * dq][e >> 1], kbv = blk[2 * dq + 1][e >> 1];
                op[dq] = (e & 1) ? __builtin_amdgcn_perm(kbv, ka, 0x07060302u) : __builtin_amdgcn_perm(kbv, ka, 0x05040100u);
            }
            f16x8 ob;
            __builtin_memcpy(&ob, &op, 16);
            acc[e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pa, ob, acc[e], 0, 0, 0);
        }
        slot = nslot;
        if (++t < ntile) continue;
        // ---- the clip ends: add the two limb rows of every head, normalise, store as the V projection's fp16-limb operand
        {
            const float lh = l_run + es_ror8(l_run);          // the head's two key quartets of this lane group
            const float inv = 1.0f / (xrow_sum(lh) * 0.5f);   // lane groups fg and fg ^ 1 hold the same eight keys: every key counted twice, exactly
            float ih[8];
#pragma unroll
            for (int q = 0; q < 8; q++) ih[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, inv), q));
            const bool up = fg & 1;
            const float inv4[4] = {up ? ih[4] : ih[0], up ? ih[5] : ih[1], up ? ih[6] : ih[2], up ? ih[7] : ih[3]};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                f32x8 ov;
#pragma unroll
                for (int e = 0; e < 8; e++) ov[e] = es_sum32(acc[e][i]) * inv4[i];   // row 4 fg + i (hi limbs) + row 8 + 4 fg + i (lo limbs)
                if (fg < 2) {
                    const int k = (4 * fg + i) * ES_D + 128 * wave + 8 * fl;
                    store8(out + ((long)(k >> 5) * mpad + clip) * 32 + (k & 31), ov);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 8; e++) acc[e] = f32x4{0, 0, 0, 0};
        m_run = -INFINITY;
        l_run = 0.0f;
        t = 0;
        clip += G;
    }
}

// The encoder's final LayerNorm into the limb planes E2 [clip][es_rows][hi 512 | lo 512] (d_model 512): one wave per row, 8 columns per lane.
// [3P] torch LayerNorm eps 1e-5, biased variance, two-pass in f32 — k_layernorm's arithmetic (wh_gemm.hip).
__global__ __launch_bounds__(256) void k_layernorm_es2(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                       _Float16* __restrict__ y, long rows, int in_blk, int out_blk) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63, c = lane * 8;
    const float* xr = x + row * ES_D;
    const f32x4 v0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr + c)), v1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr + c + 4));
    const float mean = dpp_wave_sum((v0[0] + v0[1] + v0[2] + v0[3]) + (v1[0] + v1[1] + v1[2] + v1[3])) / (float)ES_D;
    float q = 0.0f;
#pragma unroll
    for (int e = 0; e < 4; e++) { const float t0 = v0[e] - mean, t1 = v1[e] - mean; q += t0 * t0; q += t1 * t1; }
    const float rstd = rsqrtf(dpp_wave_sum(q) / (float)ES_D + 1e-5f);
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(w + c), w1 = *reinterpret_cast<const f32x4*>(w + c + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(b + c), b1 = *reinterpret_cast<const f32x4*>(b + c + 4);
    f32x8 o;
#pragma unroll
    for (int e = 0; e < 4; e++) { o[e] = (v0[e] - mean) * rstd * w0[e] + b0[e]; o[4 + e] = (v1[e] - mean) * rstd * w1[e] + b1[e]; }
    const xfrag f = x3_split(o);
    const long orow = in_blk > 0 ? (row / in_blk) * out_blk + row % in_blk : row;
    _Float16* yr = y + orow * (2 * ES_D) + c;
    *reinterpret_cast<f16x8*>(yr) = f.hi;
    *reinterpret_cast<f16x8*>(yr + ES_D) = f.lo;
}

// expanded queries, WH_PREC_F16X3: the same products with both operands as fp16 limbs (wkT stored as h2, q split in registers)
__global__ __launch_bounds__(256) void k_dec_qexpand_x3(const float* __restrict__ q, const h2* __restrict__ wkT, float* __restrict__ qe,
                                                        int M, int d, int n_heads) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fl = lane & 15, fg = lane >> 4;
    const int h = blockIdx.y, n0 = blockIdx.z * 128;
    const int m = blockIdx.x * 64 + wave * 16 + fl, mc = min(m, M - 1);
    xfrag xs[2];
#pragma unroll
    for (int ks = 0; ks < 2; ks++) xs[ks] = x3_split(*reinterpret_cast<const f32x8*>(q + (long)mc * d + h * WH_HEAD_DIM + 32 * ks + 8 * fg));
    const h2* wp = wkT + ((long)h * d + n0 + fl) * WH_HEAD_DIM + 8 * fg;
    xfrag wf[8][2];
#pragma unroll
    for (int nt = 0; nt < 8; nt++)
#pragma unroll
        for (int ks = 0; ks < 2; ks++) wf[nt][ks] = load_frag<h2>(wp + (long)nt * 16 * WH_HEAD_DIM + 32 * ks);
#pragma unroll
    for (int nt = 0; nt < 8; nt++) {
        f32x4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 2; ks++) mma16(acc, wf[nt][ks], xs[ks]);   // D rows = columns n (4 fg + r), D column = row m (fl)
        if (m < M) *reinterpret_cast<f32x4*>(qe + ((long)m * n_heads + h) * d + n0 + 16 * nt + 4 * fg) = acc;
    }
}

}  // namespace

#ifdef WH_ES_BENCH
unsigned long long* wh_es_bench_dbg = nullptr;
#endif

bool wh_cross_es_geometry(int d, int n_heads, int S) { return d == ES_D && n_heads == ES_H && S >= 4 * ES_TK; }

void wh_launch_layernorm_es2(hipStream_t s, const float* x, const float* w, const float* b, void* y, long rows, int in_blk, int out_blk) {
    hipLaunchKernelGGL(k_layernorm_es2, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, w, b, (_Float16*)y, rows, in_blk, out_blk);
}

void wh_launch_dec_cross_attn_es(hipStream_t s, int prec, const float* qe, const void* E, void* out, int S, int e_rows, int B, int mpad, bool stream_nt, int n_cus) {
    static const int nt_env = [] { const char* e = getenv("WH_CROSS_NT"); return e ? atoi(e) : -1; }();
    static const int nl = [] { const char* e = getenv("WH_ES_LOADERS"); return e ? atoi(e) : 1; }();        // (A/B runs) loader waves per workgroup
    static const int persist = [] { const char* e = getenv("WH_ES_PERSIST"); return e ? atoi(e) : 1; }();   // (A/B runs) 0: one workgroup per clip
    if (prec == WH_PREC_FP8) { wh_launch_dec_cross_attn_es8(s, qe, E, out, S, e_rows, B, mpad, stream_nt, n_cus); return; }   // e4m3 states (wh_cross_es8.hip)
    if (prec == WH_PREC_F16X3 && wh_es3_enabled()) { wh_launch_dec_cross_attn_es3(s, qe, E, out, S, e_rows, B, mpad, stream_nt, n_cus); return; }   // fp16 + e4m3 remainder (wh_cross_es3.hip)
    if (n_cus <= 0) n_cus = 256;   // (the context passes the population of its decode stream's CU mask, else its device's CU count)
    if (nt_env >= 0) stream_nt = nt_env != 0;
    const int grid = persist ? std::min(B, n_cus) : B;   // one workgroup per CU walks its clips
    if (prec == WH_PREC_F16X3) {   // fp16 limb planes (k_dec_cross_attn_es2)
#define WH_ES2_LAUNCH(AUX_, NL_)                                                                                                                       \
        do {                                                                                                                                           \
            wh_ensure_dyn_lds((const void*)k_dec_cross_attn_es2<AUX_, NL_>, E2_LDS);                                                                   \
            hipLaunchKernelGGL((k_dec_cross_attn_es2<AUX_, NL_>), dim3(grid), dim3(256 + 64 * NL_), E2_LDS, s, qe, (const _Float16*)E, (h2*)out, S, e_rows, mpad, B); \
        } while (0)
        if (nl == 2) { if (stream_nt) WH_ES2_LAUNCH(2, 2); else WH_ES2_LAUNCH(0, 2); }
        else { if (stream_nt) WH_ES2_LAUNCH(2, 1); else WH_ES2_LAUNCH(0, 1); }
#undef WH_ES2_LAUNCH
        return;
    }
    unsigned long long* es_dbg = nullptr;
#ifdef WH_ES_BENCH
    es_dbg = wh_es_bench_dbg;
#endif
#define WH_ES_LAUNCH(AUX_, NL_, ...)                                                                                                \
    do {                                                                                                                            \
        wh_ensure_dyn_lds((const void*)k_dec_cross_attn_es<AUX_, NL_, ##__VA_ARGS__>, ES_LDS);                                      \
        hipLaunchKernelGGL((k_dec_cross_attn_es<AUX_, NL_, ##__VA_ARGS__>), dim3(grid), dim3(256 + 64 * NL_), ES_LDS, s, qe, (const bf16*)E, (bf16*)out, S, e_rows, mpad, B, es_dbg); \
    } while (0)
#ifdef WH_ES_BENCH
    if (const char* e = getenv("WH_ES_ABL")) { if (atoi(e) == 2) WH_ES_LAUNCH(2, 1, 2); else WH_ES_LAUNCH(2, 1, 1); return; }
#endif
    if (nl == 2) { if (stream_nt) WH_ES_LAUNCH(2, 2); else WH_ES_LAUNCH(0, 2); }
    else { if (stream_nt) WH_ES_LAUNCH(2, 1); else WH_ES_LAUNCH(0, 1); }
#undef WH_ES_LAUNCH
}

void wh_launch_dec_qexpand(hipStream_t s, int prec, const float* q, const void* wkT, float* qe, int M, int d, int n_heads) {
    if (prec == WH_PREC_F16X3) {
        hipLaunchKernelGGL(k_dec_qexpand_x3, dim3((M + 63) / 64, n_heads, d / 128), dim3(256), 0, s, q, (const h2*)wkT, qe, M, d, n_heads);
        return;
    }
    hipLaunchKernelGGL(k_dec_qexpand, dim3((M + 63) / 64, n_heads, d / 128), dim3(256), 0, s, q, (const bf16*)wkT, qe, M, d, n_heads);
}
