// wh_cross_es8.hip — WH_PREC_FP8: decoder cross-attention of one position on the encoder states held as e4m3 (gfx950, whisper-base geometry:
// d_model 512, 8 heads).  The algebra and the structure are wh_cross_es.hip's (reference src/main.rs:771-787, 798-812: the decoder graphs'
// cross-attention over present.{i}.encoder.{key,value}):
//     score_h[key] = (Wk_h^T q_h) . E[key] =: qe_h . E[key]          out_h = Wv_h (sum_key p_h[key] E[key]) + bv_h
// with E streamed ONCE per (layer, position) for all 8 heads — here at one byte per element, half of what the fp8 mode's e4m3 K + V of every
// layer stream (2 S d bytes) and half of the bf16 encoder-state form.  Both products run on the fp8 matrix cores:
//
//   scores  D[m][key] = sum_dim QE[m][dim] * E[key][dim]     v_mfma_f32_16x16x32_fp8_fp8, rows m = {fp8(qe_h) : h} ++ {fp8(16 (qe_h - hi)) : h}
//   output  C[m][dim] = sum_key P[m][key]  * E[key][dim]     the same instruction,        rows m = {fp8(p_h) : h}  ++ {fp8(16 (p_h - hi)) : h}
//
// The 8 heads fill half of a 16-row tile; the other half carries the e4m3 remainders (scaled by 16 so that they stay normal numbers), and the
// two halves are added — the second divided by 16 — where they leave the matrix core: queries and probabilities enter with ~8 significant bits
// (what bf16 gives them in wh_cross_es.hip); the only operand at e4m3 precision is E, as K and V are in the projected form of this mode.
//
// One workgroup per CU (persistent, walks its clips): four computing waves + one loader wave; ring of five 32-key tiles of 16 KiB (512-byte key
// rows; an LDS-DMA piece of 1 KiB is two rows), one barrier per tile, the software pipeline of wh_cross_es.hip (scores of tile g + 1, softmax and
// output of tile g in one iteration).  Bank conflicts: LDS is written linearly by the DMA, so the swizzle is on the source side — 16-byte chunk p
// of tile row r holds chunk p ^ swz(r), swz(r) = (r & 15) ^ 8 ((r >> 4) & 1): both read patterns (ds_read_b64: 8 dims of a key per lane) then
// touch 32 distinct 8-byte units in each of the instruction's two 32-lane service groups.
#include <stdlib.h>

#include <algorithm>

#include "wh_common.h"
#include "wh_es_fp8.h"
#include "wh_kernels.h"

namespace {

using namespace wh_es_fp8;

constexpr int E8_D = 512, E8_H = 8, E8_TK = 32;
constexpr int E8_ROWB = E8_D;                          // bytes per key row
constexpr int E8_TILEB = E8_TK * E8_ROWB;              // 16 KiB
constexpr int E8_SCP = 36;                             // floats per (dim half, limb, head) row of the score exchange (32 keys + pad)
constexpr int E8_SCB = 4 * E8_H * E8_SCP;              // floats per score-exchange buffer: [dim half][hi | lo of the query][head][E8_SCP]
constexpr int e8_lds(int nstage) { return nstage * E8_TILEB + 2 * E8_SCB * 4 + E8_H * E8_D * 4; }   // ring + score exchange + next queries

// qe : [B][8][512] f32 expanded queries (natural-log score units)        E: [B][e_rows][512] e4m3 encoder states (final LayerNorm applied)
// out: ctx as the decode GEMM's operand, slab layout [8 * 512 / 32][mpad][32] bf16, column h * 512 + dim
template <int AUX, int NSTAGE, int NL>
__global__ __launch_bounds__(256 + 64 * NL, 1) void k_dec_cross_attn_es8(const float* __restrict__ qe, const unsigned char* __restrict__ E, bf16* __restrict__ out,
                                                                int S, int e_rows, int mpad, int B) {
    constexpr int LA = NSTAGE - 1;   // LA tiles staged ahead of the one being consumed
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sc = reinterpret_cast<float*>(smem + NSTAGE * E8_TILEB);   // [2 tiles][E8_SCB]
    float* Qs = sc + 2 * E8_SCB;                                       // [8][512] f32: the next clip's expanded queries
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntile = (S + E8_TK - 1) / E8_TK;
    const int G = gridDim.x;
    const int n_my = (B - (int)blockIdx.x + G - 1) / G;   // clips of this workgroup
    const int total = n_my * ntile;                        // tiles of this workgroup

    if (wave >= 4) {
        // ================================ loader ================================
        constexpr int PPT = E8_TILEB / 1024 / NL;   // pieces of 1 KiB = 2 key rows per tile and loader wave (an LDS-DMA piece costs its issuer ~100 cycles:
                                                    // one wave issuing all 16 takes longer than the 1,350 cycles a tile has at the HBM rate)
        const int lw = wave - 4;
        const int rsub = lane >> 5, pc = lane & 31;   // row inside a piece, physical 16-byte chunk of the row
        int st_clip = blockIdx.x, st_t = 0, st_slot = 0;   // tiles are staged strictly in sequence
        auto stage_next = [&]() {
            char* base = smem + st_slot * E8_TILEB;
            const unsigned char* Ec = E + (long)st_clip * e_rows * E8_ROWB;
#pragma unroll
            for (int jj = 0; jj < PPT; jj++) {
                const int j = lw * PPT + jj;
                const int r = 2 * j + rsub;
                const int key = min(st_t * E8_TK + r, S - 1);   // rows past the clip's end re-read its last key (finite; their scores are masked)
                glds16<AUX>(Ec + (long)key * E8_ROWB + ((pc ^ swz8(r)) << 4), base + j * 1024);
            }
            st_slot = st_slot + 1 == NSTAGE ? 0 : st_slot + 1;
            if (++st_t == ntile) { st_t = 0; st_clip += G; }
        };
        constexpr int QPP = 16 / NL;
        auto stage_q = [&](int clip) {
            const float* src = qe + (long)clip * (E8_H * E8_D);
#pragma unroll
            for (int jj = 0; jj < QPP; jj++) {
                const int j = lw * QPP + jj;
                glds16<0>(src + (j * 64 + lane) * 4, reinterpret_cast<char*>(Qs) + j * 1024);
            }
        };
        stage_q(blockIdx.x);
#pragma unroll
        for (int t = 0; t < LA; t++)
            if (t < total) stage_next();
        // vmcnt retires in issue order (and holds at most 63): "all but the last LA - 1 tiles' pieces" covers the queries and tile 0
        if (total >= LA) wait_vm<(PPT * (LA - 1) < 63 ? PPT * (LA - 1) : 63)>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();   // P1: the first clip's queries are in Qs
        __builtin_amdgcn_s_barrier();   // P2: tile 0 is in the ring
        int clip = blockIdx.x, t = 0;
        for (int g = 0; g < total; g++) {
            if (g + 1 < total) {   // tile g + 1 has landed; the younger tiles stay in flight (conservative where the next clip's queries are among them)
                if (total - 2 - g >= LA - 2) wait_vm<PPT*(LA - 2)>(); else wait_vm<0>();
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
    const int hf = wave & 1, kt = wave >> 1;
    // ---- expanded queries of this wave's dim half as the MFMA row operand: row fl -> head fl & 7, rows 0-7 the e4m3 heads, rows 8-15 the remainders
    long qa[8];
    auto qa_from_lds = [&]() {
        const float* qp = Qs + (fl & 7) * E8_D + 256 * hf + 8 * fg;
        const bool lo = fl >= 8;
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(qp + 32 * s), b = *reinterpret_cast<const f32x4*>(qp + 32 * s + 4);
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = (u < 4 ? a[u & 3] : b[u & 3]) * 1.44269504088896341f;   // scores in log2 units: p = exp2(s - m)
            const unsigned h0 = pack4(v[0], v[1], v[2], v[3]), h1 = pack4(v[4], v[5], v[6], v[7]);
            const unsigned r0 = rem4(h0, v[0], v[1], v[2], v[3]), r1 = rem4(h1, v[4], v[5], v[6], v[7]);
            qa[s] = lo ? join(r0, r1) : join(h0, h1);
        }
    };
    // ---- scores of the tile in slot `sl` for keys 16 kt + fl over dims 256 hf ..: rows 4 fg + i of D; partials to sc buffer `buf`
    auto score_reads = [&](int sl, long (&ef)[8]) {
        const int r = 16 * kt + fl;
        const char* rp = smem + sl * E8_TILEB + r * E8_ROWB + (fg & 1) * 8;
        const int sw = swz8(r);
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const int c = 16 * hf + 2 * s + (fg >> 1);
            ef[s] = *reinterpret_cast<const long*>(rp + ((c ^ sw) << 4));
        }
    };
    auto score_mfma = [&](const long (&ef)[8], int buf) {
        f32x4 d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 8; s += 2) {
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(qa[s], ef[s], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(qa[s + 1], ef[s + 1], d1, 0, 0, 0);
        }
        // rows 0-7 (lane groups 0, 1) carry the head limbs, rows 8-15 (groups 2, 3) the remainders: both go to the exchange buffer as they are
        float* dst = sc + buf * E8_SCB + ((hf * 2 + (fg >> 1)) * E8_H + 4 * (fg & 1)) * E8_SCP + 16 * kt + fl;
#pragma unroll
        for (int i = 0; i < 4; i++) dst[i * E8_SCP] = d0[i] + d1[i];
    };

    f32x4 acc[8];   // rows 4 fg + i: heads 4 (fg & 1) + i; lane groups 0, 1 from the probabilities' head limbs, 2, 3 from their remainders
    float m_run = -INFINITY, l_run = 0.0f;
    const int kb = 16 * (fg & 1) + 8 * (fg >> 1);   // first key (within a tile) of this lane's contraction slots
    int clip = blockIdx.x, t = 0, slot = 0;          // the tile being consumed: tile t of `clip`, ring slot `slot`

    __builtin_amdgcn_s_barrier();   // P1
    qa_from_lds();
    __builtin_amdgcn_s_barrier();   // P2
    {   // scores of tile 0
        long ef[8];
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
        // the next clip's first tile is scored with the next clip's queries (in Qs since a clip ago)
        if (t == ntile - 1 && more) qa_from_lds();
        const char* tb = smem + slot * E8_TILEB;
        // ---- every LDS read of this iteration up front, in the order of use: scores of tile g (softmax), score operands of tile g + 1,
        // the 8 x 8 blocks of tile g (output).  Lanes fl and fl + 8 share a head: each takes four of the lane group's eight keys
        const int h = fl & 7, kq = kb + 4 * (fl >> 3);
        const float* s0 = sc + (g & 1) * E8_SCB + h * E8_SCP + kq;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(s0), a1 = *reinterpret_cast<const f32x4*>(s0 + E8_H * E8_SCP);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(s0 + 2 * E8_H * E8_SCP), b1 = *reinterpret_cast<const f32x4*>(s0 + 3 * E8_H * E8_SCP);
        long ef[8];
        if (more) score_reads(nslot, ef);
        wh_u32x2 blk[8];
        {
            const int c = 8 * wave + (fl >> 1);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int r = kb + j;
                blk[j] = *reinterpret_cast<const wh_u32x2*>(tb + r * E8_ROWB + ((c ^ swz8(r)) << 4) + (fl & 1) * 8);
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the reads above the arithmetic below
        // ---- online softmax of tile g: lane -> head fl & 7, keys kb .. kb + 7 of the tile (identical in the four waves)
        long pa;
        {
            float sv[4];
            float tmax = -INFINITY;
            const int key0 = t * E8_TK + kq;
#pragma unroll
            for (int u = 0; u < 4; u++) sv[u] = (a0[u] + b0[u]) + (a1[u] + b1[u]) * REM_INV;   // two dim halves x {head limb, remainder / 16}
            if (t == ntile - 1) {   // (wave-uniform) keys past the end of the clip
#pragma unroll
                for (int u = 0; u < 4; u++) sv[u] = (key0 + u < S) ? sv[u] : -INFINITY;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) tmax = fmaxf(tmax, sv[u]);
            tmax = fmaxf(tmax, ror8(tmax));   // the head's other four keys of this lane group
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
            // operand rows 0-7: e4m3(p) of keys kb .. kb + 7 (this lane's four and, through DPP, its partner's); rows 8-15: their remainders x 16
            {
                const unsigned own_hi = pack4(pv[0], pv[1], pv[2], pv[3]);
                const unsigned own_lo = rem4(own_hi, pv[0], pv[1], pv[2], pv[3]);
                const unsigned oth_hi = ror8u(own_hi), oth_lo = ror8u(own_lo);
                // lane fl < 8 (row h) owns keys kb .. kb + 3, lane fl + 8 (row 8 + h) keys kb + 4 .. kb + 7
                pa = fl < 8 ? join(own_hi, oth_hi) : join(oth_lo, own_lo);
            }
            l_run = l_run * alpha + ps;
            m_run = m_new;
            // head h's factor sits in lane h: through SGPRs (v_readlane), and only when some running maximum moved (wave-uniform)
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
        // ---- output of tile g: dims 128 wave + 8 fl + e, contraction over the tile's 32 keys.  The 8 keys x 8 dims block of bytes is
        // transposed in registers: byte pairs of key pairs first (16 v_perm_b32), then the four keys of a contraction half (16 more)
        {
            unsigned w[4][4];   // w[q][m]: {key 2q dim 2m, key 2q+1 dim 2m, key 2q dim 2m+1, key 2q+1 dim 2m+1}
#pragma unroll
            for (int q = 0; q < 4; q++) {
                w[q][0] = __builtin_amdgcn_perm(blk[2 * q + 1].x, blk[2 * q].x, 0x05010400u);
                w[q][1] = __builtin_amdgcn_perm(blk[2 * q + 1].x, blk[2 * q].x, 0x07030602u);
                w[q][2] = __builtin_amdgcn_perm(blk[2 * q + 1].y, blk[2 * q].y, 0x05010400u);
                w[q][3] = __builtin_amdgcn_perm(blk[2 * q + 1].y, blk[2 * q].y, 0x07030602u);
            }
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const unsigned sel = (e & 1) ? 0x07060302u : 0x05040100u;
                const unsigned k03 = __builtin_amdgcn_perm(w[1][e >> 1], w[0][e >> 1], sel);   // keys kb .. kb + 3 of dim e
                const unsigned k47 = __builtin_amdgcn_perm(w[3][e >> 1], w[2][e >> 1], sel);   // keys kb + 4 .. kb + 7
                acc[e] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(pa, join(k03, k47), acc[e], 0, 0, 0);
            }
        }
        slot = nslot;
        if (++t < ntile) continue;
        // ---- the clip ends: rows h and 8 + h (lanes l and l ^ 32) are the head limb's and the remainder's share of head h — add, normalise, store
        {
            const float lh = l_run + ror8(l_run);   // the head's two key quartets
            const float inv = 1.0f / xrow_sum(lh);
            float ih[8];
#pragma unroll
            for (int q = 0; q < 8; q++) ih[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, inv), q));
            const bool up = fg & 1;
            const float inv4[4] = {up ? ih[4] : ih[0], up ? ih[5] : ih[1], up ? ih[6] : ih[2], up ? ih[7] : ih[3]};
            const float wgt = fg < 2 ? 1.0f : REM_INV;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                bf16x8 ov;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float mine = acc[e][i] * wgt;
                    const wh_u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mine), __float_as_uint(mine), false, false);
                    ov[e] = (bf16)((__uint_as_float(sw.x) + __uint_as_float(sw.y)) * inv4[i]);
                }
                if (fg < 2) {
                    const int k = (4 * fg + i) * E8_D + 128 * wave + 8 * fl;
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

// encoder final LayerNorm -> e4m3 rows [B][e_rows][512] (the states as the kernel above streams them): one wave per row
__global__ __launch_bounds__(256) void k_layernorm_es8(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       unsigned char* __restrict__ out, long rows, int S, int e_rows) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* xr = x + r * E8_D + lane * 8;
    const f32x4 a = *reinterpret_cast<const f32x4*>(xr), b = *reinterpret_cast<const f32x4*>(xr + 4);
    float s = 0.0f;
#pragma unroll
    for (int u = 0; u < 4; u++) s += a[u] + b[u];
    const float mean = wave_sum(s) * (1.0f / E8_D);
    float q = 0.0f;
#pragma unroll
    for (int u = 0; u < 4; u++) { q += (a[u] - mean) * (a[u] - mean); q += (b[u] - mean) * (b[u] - mean); }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / E8_D) + 1e-5f);
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + lane * 8), g1 = *reinterpret_cast<const f32x4*>(gamma + lane * 8 + 4);
    const f32x4 c0 = *reinterpret_cast<const f32x4*>(beta + lane * 8), c1 = *reinterpret_cast<const f32x4*>(beta + lane * 8 + 4);
    float v[8];
#pragma unroll
    for (int u = 0; u < 4; u++) { v[u] = (a[u] - mean) * rstd * g0[u] + c0[u]; v[4 + u] = (b[u] - mean) * rstd * g1[u] + c1[u]; }
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = fminf(fmaxf(v[u], -448.0f), 448.0f);   // e4m3's range (a LayerNorm output of that size would be an outlier of outliers; never a NaN code)
    const wh_u32x2 o = {pack4(v[0], v[1], v[2], v[3]), pack4(v[4], v[5], v[6], v[7])};
    const long clip = r / S, key = r % S;
    *reinterpret_cast<wh_u32x2*>(out + (clip * e_rows + key) * E8_D + lane * 8) = o;
}

}  // namespace

// (two workgroups per CU were measured twice: with rings of three tiles 419 vs 324 us per 2048-clip launch on random data — the shallow rings starve the stream;
// with rings of four and the queries read from global memory instead of an LDS prefetch buffer 215 vs 200 ms of cross-attention per step in the pipeline,
// tools/runs/gpu_r04as.sh — one workgroup per CU stays)
void wh_launch_dec_cross_attn_es8(hipStream_t s, const float* qe, const void* E, void* out, int S, int e_rows, int B, int mpad, bool stream_nt, int n_cus) {
    // (A/B runs) loader waves per workgroup — measured in the pipeline at 2048 clips (tools/runs/gpu_r04ao.sh): no difference (214.6 vs 215.6 ms of cross-attention), so one
    static const int nl = [] { const char* e = getenv("WH_ES8_LOADERS"); return e ? atoi(e) : 1; }();
    if (n_cus <= 0) n_cus = 256;
    const int grid = std::min(B, n_cus);   // one workgroup per CU walks its clips
#define WH_ES8_LAUNCH(AUX_, NL_)                                                                                                                         \
    do {                                                                                                                                                 \
        wh_ensure_dyn_lds((const void*)k_dec_cross_attn_es8<AUX_, 5, NL_>, e8_lds(5));                                                                   \
        hipLaunchKernelGGL((k_dec_cross_attn_es8<AUX_, 5, NL_>), dim3(grid), dim3(256 + 64 * NL_), e8_lds(5), s, qe, (const unsigned char*)E, (bf16*)out, S, e_rows, mpad, B); \
    } while (0)
    if (stream_nt) { if (nl == 1) WH_ES8_LAUNCH(2, 1); else WH_ES8_LAUNCH(2, 2); }
    else { if (nl == 1) WH_ES8_LAUNCH(0, 1); else WH_ES8_LAUNCH(0, 2); }
#undef WH_ES8_LAUNCH
}

void wh_launch_layernorm_es8(hipStream_t s, const float* x, const float* gamma, const float* beta, void* out, long rows, int S, int e_rows) {
    hipLaunchKernelGGL(k_layernorm_es8, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, gamma, beta, (unsigned char*)out, rows, S, e_rows);
}
