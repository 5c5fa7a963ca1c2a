// wh_cross_es3.hip — WH_PREC_F16X3: decoder cross-attention of one position on the encoder states held as an fp16 limb plus an E4M3 REMAINDER
// (gfx950, whisper-base geometry: d_model 512, 8 heads) — three bytes per element where wh_cross_es.hip's k_dec_cross_attn_es2 streams four.
//
//     E[key][dim] = hi + 2^-12 lo8,     hi = fp16(E),   lo8 = e4m3(2^12 (E - hi))
//
// A CPU experiment on HF whisper-base dims (teacher-forced logits against f32) gives 2.0e-3 for fp16 states alone, 4.7e-6 for two fp16 limbs and
// 5.0e-5 for this form: fifteen significant bits of E, well inside the mode's 1e-3 tolerance (reference src/main.rs:777 runs f32; tests hold the mode
// to token-exact / 1e-3 against the golden vectors).  The algebra is wh_cross_es.hip's (src/main.rs:771-787, 798-812):
//     score_h[key] = qe_h . E[key]          ctx_h = sum_key p_h[key] E[key]
// and each product is computed twice, once per plane, on the matrix cores of the plane's type:
//     hi plane   v_mfma_f32_16x16x32_f16       operand rows 0-7: fp16(x_h), rows 8-15: fp16(x_h - hi)      (x = expanded queries / probabilities:
//     lo plane   v_mfma_f32_16x16x32_fp8_fp8   operand rows 0-7: e4m3(x_h), rows 8-15: e4m3(16 (x_h - hi))   22 resp. 8 significant bits)
// the row halves and the two planes are added where they leave the matrix core (lo plane x 2^-12, its remainder rows / 16 more).
//
// Structure: wh_cross_es8.hip's (persistent workgroup per CU, four computing waves + a loader wave, LDS-DMA ring, one barrier per tile, scores of
// tile g + 1 beside softmax and output of tile g) with 32-key tiles of 48 KiB — the keys' 1 KiB fp16 rows first (the bf16 kernel's layout and
// swizzle), then their 512-byte e4m3 rows (wh_cross_es8.hip's) — in a ring of three.  With all but 7 KiB of the LDS in the ring there is no query
// prefetch buffer: a computing wave reads its slice of the next clip's expanded queries from global memory at the clip boundary.
#include <stdlib.h>

#include <algorithm>

#include "wh_common.h"
#include "wh_es_fp8.h"
#include "wh_kernels.h"

namespace {

using namespace wh_es_fp8;

constexpr int E3_D = 512, E3_H = 8, E3_TK = 32, E3_NSTAGE = 3;
constexpr int E3_ROWB = E3_D * 3;                      // bytes per key row in memory: 1 KiB of fp16, 512 B of e4m3
constexpr int E3_LO = E3_TK * E3_D * 2;                // offset of the e4m3 rows inside an LDS tile
constexpr int E3_TILEB = E3_TK * E3_ROWB;              // 48 KiB
constexpr int E3_SCP = 36;                             // floats per (dim half, limb, head) row of the score exchange (32 keys + pad)
constexpr int E3_SCB = 4 * E3_H * E3_SCP;              // floats per score-exchange buffer
constexpr int E3_LDS = E3_NSTAGE * E3_TILEB + 2 * E3_SCB * 4;   // 153 KiB
constexpr float E3_S8 = 1.0f / 4096.0f;                // the e4m3 plane's weight (states' remainders are stored x 2^12)

// two fp16 as one dword (lo half = a)
__device__ __forceinline__ unsigned e3_h2(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
    const f16x2 v = {(_Float16)a, (_Float16)b};
    unsigned u;
    __builtin_memcpy(&u, &v, 4);
    return u;
}

// qe : [B][8][512] f32 expanded queries (natural-log score units)     E: [B][e_rows] key rows of 1,536 bytes: 512 fp16, then 512 e4m3 (k_layernorm_es3)
// out: ctx as the decode GEMM's operand, h2 slab [8 * 512 / 32][mpad][32], column h * 512 + dim
template <int AUX, int NL>
__global__ __launch_bounds__(256 + 64 * NL, 1) void k_dec_cross_attn_es3(const float* __restrict__ qe, const unsigned char* __restrict__ E, h2* __restrict__ out,
                                                                int S, int e_rows, int mpad, int B) {
    constexpr int NSTAGE = E3_NSTAGE, LA = NSTAGE - 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sc = reinterpret_cast<float*>(smem + NSTAGE * E3_TILEB);   // [2 tiles][E3_SCB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntile = (S + E3_TK - 1) / E3_TK;
    const int G = gridDim.x;
    const int n_my = (B - (int)blockIdx.x + G - 1) / G;
    const int total = n_my * ntile;

    if (wave >= 4) {
        // ================================ loaders: 32 + 16 pieces of 1 KiB per tile, shared by NL waves ================================
        // (an LDS-DMA piece costs its issuer ~100 cycles: one wave issuing all 48 takes longer than the tile's 4,000 cycles of HBM time)
        const int lw = wave - 4;
        int st_clip = blockIdx.x, st_t = 0, st_slot = 0;
        auto stage_next = [&]() {
            char* base = smem + st_slot * E3_TILEB;
            const unsigned char* Ec = E + (long)st_clip * e_rows * E3_ROWB;
#pragma unroll
            for (int jj = 0; jj < E3_TK / NL; jj++) {   // fp16 rows: a piece is one key's 1 KiB, LDS chunk p holds dim-chunk p ^ (r & 15)
                const int j = lw * (E3_TK / NL) + jj;
                const int key = min(st_t * E3_TK + j, S - 1);   // rows past the clip's end re-read its last key (finite; their scores are masked)
                glds16<AUX>(Ec + (long)key * E3_ROWB + ((lane ^ (j & 15)) << 4), base + j * 1024);
            }
#pragma unroll
            for (int jj = 0; jj < E3_TK / 2 / NL; jj++) {   // e4m3 rows: a piece is two keys' 512 bytes
                const int j = lw * (E3_TK / 2 / NL) + jj;
                const int r = 2 * j + (lane >> 5);
                const int key = min(st_t * E3_TK + r, S - 1);
                glds16<AUX>(Ec + (long)key * E3_ROWB + 2 * E3_D + (((lane & 31) ^ swz8(r)) << 4), base + E3_LO + j * 1024);
            }
        st_slot = st_slot + 1 == NSTAGE ? 0 : st_slot + 1;
            if (++st_t == ntile) { st_t = 0; st_clip += G; }
        };
#pragma unroll
        for (int t = 0; t < LA; t++)
            if (t < total) stage_next();
        if (total >= LA) wait_vm<48 / NL>(); else wait_vm<0>();   // tile 0 (the older of two) has landed
        __builtin_amdgcn_s_barrier();   // P: tile 0 is in the ring
        for (int g = 0; g < total; g++) {
            if (g + 1 < total) wait_vm<0>();   // tile g + 1 has landed (nothing younger is in flight at this point)
            __builtin_amdgcn_s_barrier();
            if (g + LA < total) stage_next();
        }
        return;
    }

    // ================================ compute ================================
    const int fl = lane & 15, fg = lane >> 4;
    const int hf = wave & 1, kt = wave >> 1;
    // ---- expanded queries of this wave's dim half as the row operands of both planes: row fl -> head fl & 7, rows 0-7 head limbs, rows 8-15 remainders
    f16x8 qa16[8];
    long qa8[8];
    // (read from global memory where they are needed — once per clip, at its predecessor's last tile: the registers a prefetch would hold, 64 per lane,
    // are not there beside two planes' accumulators, operands and blocks)
    auto q_load = [&](int clip) {
        const float* qp = qe + (long)clip * (E3_H * E3_D) + (fl & 7) * E3_D + 256 * hf + 8 * fg;
        const bool lo = fl >= 8;
#pragma unroll
        for (int s0 = 0; s0 < 8; s0 += 4) {
            f32x4 q[4][2];
#pragma unroll
            for (int s = 0; s < 4; s++) {
                q[s][0] = *reinterpret_cast<const f32x4*>(qp + 32 * (s0 + s));
                q[s][1] = *reinterpret_cast<const f32x4*>(qp + 32 * (s0 + s) + 4);
            }
#pragma unroll
            for (int s = 0; s < 4; s++) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = q[s][u >> 2][u & 3] * 1.44269504088896341f;   // log2 units: p = exp2(s - m)
                f16x8 h, r;
#pragma unroll
                for (int u = 0; u < 8; u++) { h[u] = (_Float16)v[u]; r[u] = (_Float16)(v[u] - (float)h[u]); }
                qa16[s0 + s] = lo ? r : h;
                const unsigned h0 = pack4(v[0], v[1], v[2], v[3]), h1 = pack4(v[4], v[5], v[6], v[7]);
                const unsigned r0 = rem4(h0, v[0], v[1], v[2], v[3]), r1 = rem4(h1, v[4], v[5], v[6], v[7]);
                qa8[s0 + s] = lo ? join(r0, r1) : join(h0, h1);
            }
        }
    };
    // ---- scores of the tile in slot `sl` for keys 16 kt + fl over dims 256 hf ..
    auto score_reads = [&](int sl, f16x8 (&e16)[8]) {   // the fp16 plane's operands (the e4m3 plane's are read inside score_mfma, once these are spent)
        const int r = 16 * kt + fl;
        const char* rp16 = smem + sl * E3_TILEB + r * 1024;
#pragma unroll
        for (int s = 0; s < 8; s++) e16[s] = *reinterpret_cast<const f16x8*>(rp16 + (((32 * hf + 4 * s + fg) ^ (r & 15)) << 4));
    };
    const float w8 = fg < 2 ? E3_S8 : E3_S8 * REM_INV;   // weight of this lane's rows of an e4m3-plane product
    auto score_mfma = [&](int sl, const f16x8 (&e16)[8], int buf) {
        f32x4 d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0}, c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 8; s += 2) {
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa16[s], e16[s], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa16[s + 1], e16[s + 1], d1, 0, 0, 0);
        }
        {
            const int r = 16 * kt + fl;
            const char* rp8 = smem + sl * E3_TILEB + E3_LO + r * 512 + (fg & 1) * 8;
            const int sw = swz8(r);
            long e8[8];
#pragma unroll
            for (int s = 0; s < 8; s++) e8[s] = *reinterpret_cast<const long*>(rp8 + (((16 * hf + 2 * s + (fg >> 1)) ^ sw) << 4));
#pragma unroll
            for (int s = 0; s < 8; s += 2) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(qa8[s], e8[s], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(qa8[s + 1], e8[s + 1], c1, 0, 0, 0);
            }
        }
        // rows 0-7 (lane groups 0, 1) and rows 8-15 (groups 2, 3) both go to the exchange buffer; the readers add the four partials of a score
        float* dst = sc + buf * E3_SCB + ((hf * 2 + (fg >> 1)) * E3_H + 4 * (fg & 1)) * E3_SCP + 16 * kt + fl;
#pragma unroll
        for (int i = 0; i < 4; i++) dst[i * E3_SCP] = (d0[i] + d1[i]) + (c0[i] + c1[i]) * w8;
    };

    f32x4 acc16[8], acc8[8];   // rows 4 fg + i: heads 4 (fg & 1) + i; lane groups 0, 1 from the head limbs of p, 2, 3 from its remainders
    float m_run = -INFINITY, l_run = 0.0f;
    const int kb = 16 * (fg & 1) + 8 * (fg >> 1);   // first key (within a tile) of this lane's contraction slots
    int clip = blockIdx.x, t = 0, slot = 0;

    q_load(blockIdx.x);
    __builtin_amdgcn_s_barrier();   // P
    {
        f16x8 e16[8];
        score_reads(0, e16);
        score_mfma(0, e16, 0);
    }
#pragma unroll
    for (int e = 0; e < 8; e++) { acc16[e] = f32x4{0, 0, 0, 0}; acc8[e] = f32x4{0, 0, 0, 0}; }
    for (int g = 0; g < total; g++) {
        const bool more = g + 1 < total;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // tile g + 1 and the scores of tile g visible to all; every wave is done with tile g - 1
        const int nslot = slot + 1 == NSTAGE ? 0 : slot + 1;
        if (t == ntile - 1 && more) q_load(clip + G);   // the next clip's first tile is scored with the next clip's queries
        const char* tb = smem + slot * E3_TILEB;
        const int h = fl & 7, kq = kb + 4 * (fl >> 3);
        const float* s0 = sc + (g & 1) * E3_SCB + h * E3_SCP + kq;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(s0), a1 = *reinterpret_cast<const f32x4*>(s0 + E3_H * E3_SCP);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(s0 + 2 * E3_H * E3_SCP), b1 = *reinterpret_cast<const f32x4*>(s0 + 3 * E3_H * E3_SCP);
        f16x8 e16[8];
        if (more) score_reads(nslot, e16);
        __builtin_amdgcn_sched_barrier(0);
        // ---- online softmax of tile g: lane -> head fl & 7, keys kb .. kb + 7 of the tile (identical in the four waves)
        f16x8 pa16;
        long pa8;
        {
            float sv[4];
            float tmax = -INFINITY;
            const int key0 = t * E3_TK + kq;
#pragma unroll
            for (int u = 0; u < 4; u++) sv[u] = (a0[u] + a1[u]) + (b0[u] + b1[u]);
            if (t == ntile - 1) {
#pragma unroll
                for (int u = 0; u < 4; u++) sv[u] = (key0 + u < S) ? sv[u] : -INFINITY;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) tmax = fmaxf(tmax, sv[u]);
            tmax = fmaxf(tmax, ror8(tmax));
            tmax = xrow_max(tmax);
            const float m_new = fmaxf(m_run, tmax);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            float ps = 0.0f;
            float pv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                pv[u] = __builtin_amdgcn_exp2f(sv[u] - m_new);
                ps += pv[u];
            }
            {   // fp16 operand: rows 0-7 fp16(p) of keys kb .. kb + 7, rows 8-15 fp16(p - hi); lane fl < 8 owns keys kb .. kb + 3, lane fl + 8 keys kb + 4 ..
                const _Float16 q0 = (_Float16)pv[0], q1 = (_Float16)pv[1], q2 = (_Float16)pv[2], q3 = (_Float16)pv[3];
                const unsigned oh0 = e3_h2((float)q0, (float)q1), oh1 = e3_h2((float)q2, (float)q3);
                const unsigned ol0 = e3_h2(pv[0] - (float)q0, pv[1] - (float)q1), ol1 = e3_h2(pv[2] - (float)q2, pv[3] - (float)q3);
                const unsigned xh0 = ror8u(oh0), xh1 = ror8u(oh1), xl0 = ror8u(ol0), xl1 = ror8u(ol1);
                const wh_u32x4 w = fl < 8 ? wh_u32x4{oh0, oh1, xh0, xh1} : wh_u32x4{xl0, xl1, ol0, ol1};
                __builtin_memcpy(&pa16, &w, 16);
                const unsigned own_hi = pack4(pv[0], pv[1], pv[2], pv[3]);
                const unsigned own_lo = rem4(own_hi, pv[0], pv[1], pv[2], pv[3]);
                const unsigned oth_hi = ror8u(own_hi), oth_lo = ror8u(own_lo);
                pa8 = fl < 8 ? join(own_hi, oth_hi) : join(oth_lo, own_lo);
            }
            l_run = l_run * alpha + ps;
            m_run = m_new;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
                float ah[8];
#pragma unroll
                for (int q = 0; q < 8; q++) ah[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, alpha), q));
                const bool up = fg & 1;
                const float a4[4] = {up ? ah[4] : ah[0], up ? ah[5] : ah[1], up ? ah[6] : ah[2], up ? ah[7] : ah[3]};
#pragma unroll
                for (int e = 0; e < 8; e++) {
#pragma unroll
                    for (int i = 0; i < 4; i++) { acc16[e][i] *= a4[i]; acc8[e][i] *= a4[i]; }
                }
            }
        }
        // ---- scores of tile g + 1 (their operands leave the registers before the output blocks are read: both sets at once do not fit 256 registers)
        if (more) score_mfma(nslot, e16, (g + 1) & 1);
        // ---- output of tile g: dims 128 wave + 8 fl + e, contraction over the tile's 32 keys; the 8 keys x 8 dims blocks are transposed in registers.
        // One plane after the other: the blocks of both at once do not fit the register budget either.
        {   // fp16 plane (wh_cross_es.hip's transposition of 2-byte elements)
            wh_u32x4 blk16[8];
            const int cs = 16 * wave + fl;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int r = kb + j;
                blk16[j] = *reinterpret_cast<const wh_u32x4*>(tb + r * 1024 + ((cs ^ (r & 15)) << 4));
            }
#pragma unroll
            for (int e = 0; e < 8; e++) {
                wh_u32x4 op;
#pragma unroll
                for (int dq = 0; dq < 4; dq++) {
                    const unsigned ka = blk16[2 * dq][e >> 1], kbv = blk16[2 * dq + 1][e >> 1];
                    op[dq] = (e & 1) ? __builtin_amdgcn_perm(kbv, ka, 0x07060302u) : __builtin_amdgcn_perm(kbv, ka, 0x05040100u);
                }
                f16x8 ob;
                __builtin_memcpy(&ob, &op, 16);
                acc16[e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pa16, ob, acc16[e], 0, 0, 0);
            }
        }
        {   // e4m3 plane (wh_cross_es8.hip's byte transposition)
            wh_u32x2 blk8[8];
            const int c8 = 8 * wave + (fl >> 1);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int r = kb + j;
                blk8[j] = *reinterpret_cast<const wh_u32x2*>(tb + E3_LO + r * 512 + ((c8 ^ swz8(r)) << 4) + (fl & 1) * 8);
            }
            unsigned w[4][4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                w[q][0] = __builtin_amdgcn_perm(blk8[2 * q + 1].x, blk8[2 * q].x, 0x05010400u);
                w[q][1] = __builtin_amdgcn_perm(blk8[2 * q + 1].x, blk8[2 * q].x, 0x07030602u);
                w[q][2] = __builtin_amdgcn_perm(blk8[2 * q + 1].y, blk8[2 * q].y, 0x05010400u);
                w[q][3] = __builtin_amdgcn_perm(blk8[2 * q + 1].y, blk8[2 * q].y, 0x07030602u);
            }
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const unsigned sel = (e & 1) ? 0x07060302u : 0x05040100u;
                const unsigned k03 = __builtin_amdgcn_perm(w[1][e >> 1], w[0][e >> 1], sel);
                const unsigned k47 = __builtin_amdgcn_perm(w[3][e >> 1], w[2][e >> 1], sel);
                acc8[e] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(pa8, join(k03, k47), acc8[e], 0, 0, 0);
            }
        }
        slot = nslot;
        if (++t < ntile) continue;
        // ---- the clip ends: rows h and 8 + h (lanes l and l ^ 32) and the two planes are added, normalised, stored as an h2 slab
        {
            const float lh = l_run + ror8(l_run);
            const float inv = 1.0f / xrow_sum(lh);
            float ih[8];
#pragma unroll
            for (int q = 0; q < 8; q++) ih[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, inv), q));
            const bool up = fg & 1;
            const float inv4[4] = {up ? ih[4] : ih[0], up ? ih[5] : ih[1], up ? ih[6] : ih[2], up ? ih[7] : ih[3]};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                f32x8 ov;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float mine = acc16[e][i] + acc8[e][i] * w8;
                    const wh_u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mine), __float_as_uint(mine), false, false);
                    ov[e] = (__uint_as_float(sw.x) + __uint_as_float(sw.y)) * inv4[i];
                }
                if (fg < 2) {
                    const int k = (4 * fg + i) * E3_D + 128 * wave + 8 * fl;
                    store8(out + ((long)(k >> 5) * mpad + clip) * 32 + (k & 31), ov);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 8; e++) { acc16[e] = f32x4{0, 0, 0, 0}; acc8[e] = f32x4{0, 0, 0, 0}; }
        m_run = -INFINITY;
        l_run = 0.0f;
        t = 0;
        clip += G;
    }
}

// The encoder's final LayerNorm into key rows of [512 fp16 | 512 e4m3 remainders x 2^12]: one wave per row, 8 columns per lane
// ([3P] torch LayerNorm eps 1e-5, biased variance, two-pass in f32 — k_layernorm_es2's arithmetic)
__global__ __launch_bounds__(256) void k_layernorm_es3(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                       unsigned char* __restrict__ y, long rows, int in_blk, int out_blk) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63, c = lane * 8;
    const float* xr = x + row * E3_D;
    const f32x4 v0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr + c)), v1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr + c + 4));
    const float mean = dpp_wave_sum((v0[0] + v0[1] + v0[2] + v0[3]) + (v1[0] + v1[1] + v1[2] + v1[3])) / (float)E3_D;
    float q = 0.0f;
#pragma unroll
    for (int e = 0; e < 4; e++) { const float t0 = v0[e] - mean, t1 = v1[e] - mean; q += t0 * t0; q += t1 * t1; }
    const float rstd = rsqrtf(dpp_wave_sum(q) / (float)E3_D + 1e-5f);
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(w + c), w1 = *reinterpret_cast<const f32x4*>(w + c + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(b + c), b1 = *reinterpret_cast<const f32x4*>(b + c + 4);
    float o[8];
#pragma unroll
    for (int e = 0; e < 4; e++) { o[e] = (v0[e] - mean) * rstd * w0[e] + b0[e]; o[4 + e] = (v1[e] - mean) * rstd * w1[e] + b1[e]; }
    f16x8 hi;
    float rm[8];
#pragma unroll
    for (int e = 0; e < 8; e++) { hi[e] = (_Float16)o[e]; rm[e] = fminf(fmaxf((o[e] - (float)hi[e]) * 4096.0f, -448.0f), 448.0f); }   // (|o| < 256 never reaches the clamp)
    const long orow = in_blk > 0 ? (row / in_blk) * out_blk + row % in_blk : row;
    unsigned char* yr = y + orow * E3_ROWB;
    *reinterpret_cast<f16x8*>(yr + c * 2) = hi;
    *reinterpret_cast<wh_u32x2*>(yr + 2 * E3_D + c) = wh_u32x2{pack4(rm[0], rm[1], rm[2], rm[3]), pack4(rm[4], rm[5], rm[6], rm[7])};
}

}  // namespace

// WH_ES3=0: the split-fp16 mode keeps its encoder states as two fp16 limbs (k_dec_cross_attn_es2) — read once per process, by the writer and the kernel alike
bool wh_es3_enabled() {
    static const bool on = [] { const char* e = getenv("WH_ES3"); return !(e && atoi(e) == 0); }();
    return on;
}

void wh_launch_dec_cross_attn_es3(hipStream_t s, const float* qe, const void* E, void* out, int S, int e_rows, int B, int mpad, bool stream_nt, int n_cus) {
    static const int nl = [] { const char* e = getenv("WH_ES3_LOADERS"); return e ? atoi(e) : 2; }();   // (A/B runs) loader waves per workgroup
    if (n_cus <= 0) n_cus = 256;
    const int grid = std::min(B, n_cus);   // one workgroup per CU walks its clips
#define WH_ES3_LAUNCH(AUX_, NL_)                                                                                                                      \
    do {                                                                                                                                              \
        wh_ensure_dyn_lds((const void*)k_dec_cross_attn_es3<AUX_, NL_>, E3_LDS);                                                                      \
        hipLaunchKernelGGL((k_dec_cross_attn_es3<AUX_, NL_>), dim3(grid), dim3(256 + 64 * NL_), E3_LDS, s, qe, (const unsigned char*)E, (h2*)out, S, e_rows, mpad, B); \
    } while (0)
    if (stream_nt) { if (nl == 1) WH_ES3_LAUNCH(2, 1); else WH_ES3_LAUNCH(2, 2); }
    else { if (nl == 1) WH_ES3_LAUNCH(0, 1); else WH_ES3_LAUNCH(0, 2); }
#undef WH_ES3_LAUNCH
}

void wh_launch_layernorm_es3(hipStream_t s, const float* x, const float* w, const float* b, void* y, long rows, int in_blk, int out_blk) {
    hipLaunchKernelGGL(k_layernorm_es3, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, w, b, (unsigned char*)y, rows, in_blk, out_blk);
}
