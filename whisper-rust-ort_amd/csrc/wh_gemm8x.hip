// wh_gemm8x.hip — the encoder-side GEMM of WH_PREC_F16X3 for gfx950: both operands as fp16 limb pairs (`h2`, wh_common.h),
// 256 x 256 x 32 tiles, 8 waves, operands streamed global -> LDS by the LDS-DMA path, three fp16 MFMAs per product.
//
//   C[m][n] = act( sum_k A[m][k] * W[n][k] + bias ) + R[m][n]          (the contract of k_gemm / k_gemm8, wh_gemm.hip)
//
// It stands in for ONNX Runtime's MLAS GEMM / Conv nodes behind run_encoder (reference src/main.rs:698-707) in the mode that
// reproduces their f32 results (src/main.rs:777) on the matrix cores: x = hi + lo with two fp16 limbs (22 significant bits),
// a . w ~= a.lo w.hi + a.hi w.lo + a.hi w.hi with f32 accumulation.  Against k_gemm8 (bf16) a k-step moves twice the operand
// bytes and issues three times the MFMAs, so the loop is bound by the matrix pipe, not by the L2 -> LDS stream:
//   * an operand row of a k-step is one 128-byte h2 block [32 x hi | 32 x lo]; a wave-instruction of the LDS-DMA writes 1 KiB =
//     8 such rows, lane i -> row i / 8, 16-byte chunk i % 8.  LDS is written linearly, so the bank swizzle is on the source side:
//     LDS chunk p of row r holds chunk p ^ ((r >> 1) & 7) of the row's block.  Fragment reads (ds_read_b128: lane l -> row l & 15,
//     hi chunk l >> 4, lo chunk 4 + (l >> 4)) then hit 16 distinct 16-byte slots in every service group of MI355X_MICROARCH.md
//     §LDS (rows alternate between the two halves of a 256-byte bank row; the XOR spreads row pairs over the eight chunk slots).
//   * slot = A tile (256 rows) + W tile (256 rows) = 64 KiB; two slots.  k-step t: this wave's stage-t loads have landed
//     (s_waitcnt vmcnt(0)), s_barrier (all of stage t visible, every wave done with stage t - 1), issue stage t + 1 into the other
//     slot, then the 96 MFMAs of stage t — a stage has one whole k-step (~3,000 SIMD cycles at two waves per SIMD) to land.
//   * waves 2 x 4, 128 x 64 per wave.  The four weight fragments stay in registers for the k-step; activation fragments are read
//     two row tiles at a time between the MFMA groups, so the 32 accumulators (128 registers) and the limbs fit 256 registers.
//   * epilogue as k_gemm8: wave-private passes through the idle ring, row-contiguous 8-column groups per lane — f32 rows
//     (residual stream, f32 consumers) or h2 blocks (the next GEMM's / the attention's operand: two 16-byte stores per group).
//   * XCD-aware tile order as in k_gemm: one XCD walks a contiguous run of tiles, n fastest.
#include <stdlib.h>

#include "wh_common.h"
#include "wh_kernels.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 32;
constexpr int ROWB = BK * 4;                        // 128 bytes per LDS row: 32 hi limbs, 32 lo limbs
constexpr int SLOT_A = BM * ROWB, SLOT = SLOT_A + BN * ROWB;   // 64 KiB
constexpr int NSLOT = 2;
constexpr int WN = 4, WM = 2;                       // waves along n / m
constexpr int TM = BM / WM / 16, TN = BN / WN / 16; // 8 x 4 MFMA tiles of 16 x 16 per wave
constexpr int EP_PITCH = 68;                        // floats per staged output row (64 + 4: conflict-free both ways)
constexpr int EP_ROWS = 32;                         // rows of a wave's sub-tile staged per pass (8.5 KiB per wave)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

template <typename TO>
__global__ __launch_bounds__(512, 2) void k_gemm8x(GemmArgs g) {
    extern __shared__ __attribute__((aligned(128))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int fl = lane & 15, fg = lane >> 4;
    const int nk = g.K / BK;

    const int nbn = (g.N + BN - 1) / BN;
    const int total = nbn * ((g.M + BM - 1) / BM);
    int tile = blockIdx.x;
    {
        const int q = total >> 3, r = total & 7, xcd = tile & 7, idx = tile >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (tile / nbn) * BM, n0 = (tile % nbn) * BN;
    const long z = blockIdx.z;
    const char* A = reinterpret_cast<const char*>((const h2*)g.A + z * g.a_zs);
    const char* W = reinterpret_cast<const char*>((const h2*)g.W + z * g.w_zs);

    // per-lane source addresses of this wave's share of a stage: one wave-instruction = 1 KiB = 8 rows x 128 bytes, lane i -> row i / 8,
    // LDS chunk i % 8; 4 instructions of A (32 rows) and 4 of W (32 rows) per wave
    const int rl = lane >> 3, ps = lane & 7;
    const char* a_src[4];
    const char* w_src[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int row = wave * 32 + j * 8 + rl;
        const int m = min(m0 + row, g.M - 1), n = min(n0 + row, g.N - 1);
        a_src[j] = A + ((long)(m / g.m_per) * g.a_bs + (long)(m % g.m_per) * g.lda) * 4 + ((ps ^ swz(row)) << 4);
        w_src[j] = W + (long)n * g.ldw * 4 + ((ps ^ swz(row)) << 4);
    }
    auto stage = [&](int slot, int kt) {
        char* base = smem + slot * SLOT;
#pragma unroll
        for (int j = 0; j < 4; j++) glds16(a_src[j] + (long)kt * ROWB, base + (wave * 32 + j * 8) * ROWB);
#pragma unroll
        for (int j = 0; j < 4; j++) glds16(w_src[j] + (long)kt * ROWB, base + SLOT_A + (wave * 32 + j * 8) * ROWB);
    };
    auto stage_half = [&](int slot, int kt, int half) {   // half 0: this wave's A rows, half 1: its W rows
        char* base = smem + slot * SLOT;
        if (half == 0) {
#pragma unroll
            for (int j = 0; j < 4; j++) glds16(a_src[j] + (long)kt * ROWB, base + (wave * 32 + j * 8) * ROWB);
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) glds16(w_src[j] + (long)kt * ROWB, base + SLOT_A + (wave * 32 + j * 8) * ROWB);
        }
    };

    f32x4 acc[TM][TN];   // [m tile][n tile]
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = f32x4{0, 0, 0, 0};

    // fragment byte offsets inside a slot: row fl of a 16-row tile (tiles start at multiples of 16 rows, so swz(row) = swz(fl))
    const int ch_hi = (fg ^ swz(fl)) << 4, ch_lo = ((4 + fg) ^ swz(fl)) << 4;
    const int a_row = (wm * (TM * 16) + fl) * ROWB, w_row = SLOT_A + (wn * (TN * 16) + fl) * ROWB;
    auto frag = [&](const char* sb, int row_off) {
        xfrag f;
        f.hi = *reinterpret_cast<const f16x8*>(sb + row_off + ch_hi);
        f.lo = *reinterpret_cast<const f16x8*>(sb + row_off + ch_lo);
        return f;
    };

    stage(0, 0);
    for (int kt = 0; kt < nk; kt++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of stage kt has landed
        __builtin_amdgcn_s_barrier();                       // stage kt visible to all; every wave has read the fragments of kt - 1
        const char* sb = smem + (kt & 1) * SLOT;
        xfrag wf[TN];
#pragma unroll
        for (int j = 0; j < TN; j++) wf[j] = frag(sb, w_row + j * 16 * ROWB);
#pragma unroll
        for (int i0 = 0; i0 < TM; i0 += 2) {
            xfrag af[2];
#pragma unroll
            for (int u = 0; u < 2; u++) af[u] = frag(sb, a_row + (i0 + u) * 16 * ROWB);
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
                for (int j = 0; j < TN; j++) mma16(acc[i0 + u][j], wf[j], af[u]);   // D rows = n, cols = m
            // the next stage's eight LDS-DMA instructions go out behind the first two MFMA groups (24 MFMAs = ~400 cycles of matrix work each):
            // an LDS-DMA instruction costs its issuer ~100 cycles and a wave issues in order — issued right after the barrier, ahead of the
            // MFMAs, they held every wave of the workgroup in DMA issue while the matrix pipe idled
            if (kt + 1 < nk && i0 < 4) {
                __builtin_amdgcn_sched_barrier(0);
                stage_half((kt + 1) & 1, kt + 1, i0 >> 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (all fragment reads of this wave have left LDS before the next barrier)
    }
    __builtin_amdgcn_s_barrier();   // every wave is done with the ring: it becomes the output staging area

    // ---- epilogue: passes of 32 rows per wave through the wave's own 8.5 KiB of the idle ring ------------------------
    float* stg = reinterpret_cast<float*>(smem) + wave * (EP_ROWS * EP_PITCH);
    const int nw0 = n0 + wn * 64;            // first column of this wave's sub-tile
    const int mw0 = m0 + wm * (TM * 16);
    TO* C = (TO*)g.C + z * g.c_zs;
    const float* R = g.R ? g.R + z * g.r_zs : nullptr;
    const long nc0 = (long)(nw0 / g.n_per) * g.c_ns + (nw0 % g.n_per);   // n_per is a multiple of 64 or >= N (checked at launch)
    f32x4 pb[TN];
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int n = nw0 + j * 16 + fg * 4;
        pb[j] = f32x4{0, 0, 0, 0};
        if (n < g.N && g.bias_mode == 1 && g.bias) pb[j] = *reinterpret_cast<const f32x4*>(g.bias + n);
    }
    float rowb[TM];
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int m = min(mw0 + i * 16 + fl, g.M - 1);
        rowb[i] = (g.bias_mode == 2 && g.bias) ? g.bias[m] : 0.0f;
    }
    const int c8 = (lane & 7) * 8, r8 = lane >> 3;   // row-contiguous read-back: 8 lanes x 8 columns per row, 8 rows per wave-instruction
    const int n_st = nw0 + c8;
#pragma unroll
    for (int pass = 0; pass < TM / 2; pass++) {
        // the residual rows of this pass, requested before its staging writes
        f32x4 rr0[EP_ROWS / 8], rr1[EP_ROWS / 8];
        if (R) {
            const int mq0 = mw0 + pass * EP_ROWS + r8;
            long qb = mq0 / g.m_per, qi = mq0 % g.m_per;
#pragma unroll
            for (int it = 0; it < EP_ROWS / 8; it++) {
                rr0[it] = f32x4{0, 0, 0, 0};
                rr1[it] = f32x4{0, 0, 0, 0};
                if (mq0 + it * 8 < g.M && n_st < g.N) {
                    const float* rp = R + qb * g.r_bs + qi * g.ldr + n_st;
                    rr0[it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(rp));  // read once
                    rr1[it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(rp + 4));
                }
                qi += 8;
                if (qi >= g.m_per) { qi -= g.m_per; qb += 1; }
            }
        }
#pragma unroll
        for (int ii = 0; ii < 2; ii++) {
            const int i = pass * 2 + ii;
            const float bm = rowb[i];
#pragma unroll
            for (int j = 0; j < TN; j++) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = acc[i][j][e] + (pb[j][e] + bm);
                if (g.act == 1) {
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = gelu_erf(v[e]);
                }
                *reinterpret_cast<f32x4*>(&stg[(ii * 16 + fl) * EP_PITCH + j * 16 + fg * 4]) = f32x4{v[0], v[1], v[2], v[3]};
            }
        }
        // wave-private staging: the wave's own LDS writes are ordered before its reads by the lgkmcnt wait the compiler inserts
        const int mp0 = mw0 + pass * EP_ROWS + r8;
        long mb = mp0 / g.m_per, mi = mp0 % g.m_per;
#pragma unroll
        for (int it = 0; it < EP_ROWS / 8; it++) {
            const int lr = it * 8 + r8, m = mp0 + it * 8;
            if (m < g.M && n_st < g.N) {
                f32x4 v0 = *reinterpret_cast<const f32x4*>(&stg[lr * EP_PITCH + c8]);
                f32x4 v1 = *reinterpret_cast<const f32x4*>(&stg[lr * EP_PITCH + c8 + 4]);
                if (R) { v0 += rr0[it]; v1 += rr1[it]; }
                TO* cp = C + mb * g.c_bs + mi * g.ldc + nc0 + c8;
                if (n_st + 8 <= g.N) store8(cp, f32x8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]});
                else store4(cp, v0[0], v0[1], v0[2], v0[3]);          // N % 8 == 4: the last group holds 4 valid columns
            }
            mi += 8;
            if (mi >= g.m_per) { mi -= g.m_per; mb += 1; }
        }
    }
}


// ---- the LM head at hundreds of rows, WH_PREC_F16X3: k_lm_head_tile's contract (wh_gemm8.hip) on this file's main loop ----------------------
// logits = LN(x) . E^T over the whole vocabulary + masked argmax partials (reference argmax_last_dim_raw, src/main.rs:709-735).  A workgroup
// owns a 256 x 256 logit tile; the activation rows come from the decode slab layout [K/32][mpad][32] h2 (a k-step's 256 rows are 256
// consecutive 128-byte blocks), the tied embedding with the final LayerNorm's gamma folded in is the weight operand.  Same MFMA chain over k
// as k_lm_head<h2> (three fp16 MFMAs per k-step in mma16's order, one accumulator), LayerNorm partial sums reduced in the same order, the
// same epilogue expression: logits bit-identical, so the launcher may pick by the call's row count.
__global__ __launch_bounds__(512, 2) void k_lm_head_tile_x3(SkinnyArgs a) {
    extern __shared__ __attribute__((aligned(128))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int fl = lane & 15, fg = lane >> 4;
    const int nk = a.K / BK;
    const int nbn = (a.N + BN - 1) / BN;
    const int total = nbn * ((a.M + BM - 1) / BM);
    int tile = blockIdx.x;
    {
        const int q = total >> 3, r = total & 7, xcd = tile & 7, idx = tile >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int ct = tile % nbn, m0 = (tile / nbn) * BM, n0 = ct * BN;

    // final LayerNorm: quarter sums of the producer's per-tile partials, two quarters per thread (row tid & 255)
    float* lnstat = reinterpret_cast<float*>(smem + (size_t)NSLOT * SLOT);   // [256][2] mean, rstd
    float* lnq = lnstat + 2 * BM;                                            // [4][256][2]
    if (a.ln_part) {
        const int r = tid & (BM - 1), h = tid >> 8, row = min(m0 + r, a.x_mpad - 1);
        float s1a, s2a, s1b, s2b;
        ln_partial_sum(a.ln_part, a.ln_tiles, a.x_mpad, row, h, 4, s1a, s2a);
        ln_partial_sum(a.ln_part, a.ln_tiles, a.x_mpad, row, h + 2, 4, s1b, s2b);
        lnq[(h * BM + r) * 2] = s1a;
        lnq[(h * BM + r) * 2 + 1] = s2a;
        lnq[((h + 2) * BM + r) * 2] = s1b;
        lnq[((h + 2) * BM + r) * 2 + 1] = s2b;
    }

    const int rl = lane >> 3, ps = lane & 7;
    const char* a_src[4];
    const char* w_src[4];
    const long a_kstep = (long)a.x_mpad * ROWB;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int row = wave * 32 + j * 8 + rl;
        a_src[j] = reinterpret_cast<const char*>(a.X) + (long)min(m0 + row, a.x_mpad - 1) * ROWB + ((ps ^ swz(row)) << 4);
        w_src[j] = reinterpret_cast<const char*>(a.W) + (long)min(n0 + row, a.N - 1) * a.K * 4 + ((ps ^ swz(row)) << 4);
    }
    auto stage = [&](int slot, int kt) {
        char* base = smem + slot * SLOT;
#pragma unroll
        for (int j = 0; j < 4; j++) glds16(a_src[j] + kt * a_kstep, base + (wave * 32 + j * 8) * ROWB);
#pragma unroll
        for (int j = 0; j < 4; j++) glds16(w_src[j] + (long)kt * ROWB, base + SLOT_A + (wave * 32 + j * 8) * ROWB);
    };
    auto stage_half = [&](int slot, int kt, int half) {
        char* base = smem + slot * SLOT;
        if (half == 0) {
#pragma unroll
            for (int j = 0; j < 4; j++) glds16(a_src[j] + kt * a_kstep, base + (wave * 32 + j * 8) * ROWB);
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) glds16(w_src[j] + (long)kt * ROWB, base + SLOT_A + (wave * 32 + j * 8) * ROWB);
        }
    };
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = f32x4{0, 0, 0, 0};
    const int ch_hi = (fg ^ swz(fl)) << 4, ch_lo = ((4 + fg) ^ swz(fl)) << 4;
    const int a_row = (wm * (TM * 16) + fl) * ROWB, w_row = SLOT_A + (wn * (TN * 16) + fl) * ROWB;
    auto frag = [&](const char* sb, int row_off) {
        xfrag f;
        f.hi = *reinterpret_cast<const f16x8*>(sb + row_off + ch_hi);
        f.lo = *reinterpret_cast<const f16x8*>(sb + row_off + ch_lo);
        return f;
    };
    stage(0, 0);
    for (int kt = 0; kt < nk; kt++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (first step: this wave's quarter sums have left for LDS)
        __builtin_amdgcn_s_barrier();
        if (kt == 0 && a.ln_part && tid < BM) {   // the quarter sums are in LDS since the barrier above
            const float s1 = (lnq[tid * 2] + lnq[(BM + tid) * 2]) + (lnq[(2 * BM + tid) * 2] + lnq[(3 * BM + tid) * 2]);
            const float s2 = (lnq[tid * 2 + 1] + lnq[(BM + tid) * 2 + 1]) + (lnq[(2 * BM + tid) * 2 + 1] + lnq[(3 * BM + tid) * 2 + 1]);
            float mean, rstd;
            wh_ln_mean_rstd(s1, s2, (float)a.K, false, mean, rstd);
            lnstat[2 * tid] = mean;
            lnstat[2 * tid + 1] = rstd;
        }
        const char* sb = smem + (kt & 1) * SLOT;
        xfrag wf[TN];
#pragma unroll
        for (int j = 0; j < TN; j++) wf[j] = frag(sb, w_row + j * 16 * ROWB);
#pragma unroll
        for (int i0 = 0; i0 < TM; i0 += 2) {
            xfrag af[2];
#pragma unroll
            for (int u = 0; u < 2; u++) af[u] = frag(sb, a_row + (i0 + u) * 16 * ROWB);
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
                for (int j = 0; j < TN; j++) mma16(acc[i0 + u][j], wf[j], af[u]);
            if (kt + 1 < nk && i0 < 4) {   // the next stage's LDS-DMA behind the first MFMA groups (k_gemm8x's order)
                __builtin_amdgcn_sched_barrier(0);
                stage_half((kt + 1) & 1, kt + 1, i0 >> 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    __syncthreads();   // lnstat visible to everyone; the ring is idle

    // ---- epilogue: final LayerNorm fold + masked argmax, one partial per (column tile, row) ----------------------------------------------
    const int pos = *a.pos_p;
    const int gen = pos - (a.n_prompt - 1);
    const unsigned* mask = (gen == 0) ? a.mask_first : a.mask_base;
    const int nw0 = n0 + wn * 64;
    float sv[TN][4], cv[TN][4];
    unsigned mbits[TN];
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int n = nw0 + j * 16 + 4 * fg;
#pragma unroll
        for (int e = 0; e < 4; e++) { sv[j][e] = 0.0f; cv[j][e] = 0.0f; }
        if (a.ln_part) {
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (n + e < a.N) { sv[j][e] = a.ln_s[n + e]; cv[j][e] = a.bias[n + e]; }
        }
        mbits[j] = 0;
        if (n < a.N) mbits[j] = mask[n >> 5] >> (n & 31);
    }
    float* red_v = reinterpret_cast<float*>(smem);            // [4][256]
    int* red_i = reinterpret_cast<int*>(smem) + WN * BM;
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int rloc = wm * (TM * 16) + i * 16 + fl, m = m0 + rloc;
        const float mean = a.ln_part ? lnstat[2 * rloc] : 0.0f, rstd = a.ln_part ? lnstat[2 * rloc + 1] : 1.0f;
        float bv = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int n = nw0 + j * 16 + 4 * fg;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int nn = n + e;
                const float v = a.ln_part ? wh_ln_fold(acc[i][j][e], mean, rstd, sv[j][e], cv[j][e]) : acc[i][j][e];
                if (nn < a.N && m < a.M) {
                    if (a.logits && gen >= 0 && gen < a.logits_rows) {
                        const int slot = a.logits_sel ? a.logits_sel[m] : m;
                        if (slot >= 0) a.logits[((long)slot * a.logits_rows + gen) * a.N + nn] = v;
                    }
                    const bool sup = (mbits[j] >> e) & 1u;
                    if (!sup && v > bv) { bv = v; bi = nn; }  // strict >, columns ascending: lowest index on ties, NaN never wins
                }
            }
        }
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
        if (fg == 0) {
            red_v[wn * BM + rloc] = bv;
            red_i[wn * BM + rloc] = bi;
        }
    }
    __syncthreads();
    if (tid < BM && m0 + tid < a.M) {
        float bv = red_v[tid];
        int bi = red_i[tid];
#pragma unroll
        for (int w = 1; w < WN; w++) {
            const float v1 = red_v[w * BM + tid];
            const int i1 = red_i[w * BM + tid];
            const bool take1 = v1 > bv || (v1 == bv && i1 < bi);
            bv = take1 ? v1 : bv;
            bi = take1 ? i1 : bi;
        }
        a.part_val[(long)ct * a.x_mpad + m0 + tid] = bv;
        a.part_idx[(long)ct * a.x_mpad + m0 + tid] = bi;
    }
}

template <typename TO>
void launch8x(hipStream_t s, const GemmArgs& g) {
    const size_t sm = (size_t)NSLOT * SLOT;
    static_assert((size_t)8 * EP_ROWS * EP_PITCH * 4 <= (size_t)NSLOT * SLOT, "output staging must fit the ring");
    dim3 grid(((g.N + BN - 1) / BN) * ((g.M + BM - 1) / BM), 1, g.batch);
    wh_ensure_dyn_lds((const void*)k_gemm8x<TO>, sm);
    hipLaunchKernelGGL((k_gemm8x<TO>), grid, dim3(512), sm, s, g);
}

}  // namespace

// h2 operands: every row start 128-byte aligned (lda, ldw, a_bs multiples of 32 elements), K a multiple of 32; h2 outputs need column groups
// of 8 (a last one of 4) inside one 32-block (ldc, c_bs, c_ns multiples of 32)
bool wh_gemm8x_applicable(const GemmArgs& g, bool out_h2) {
    const bool rows_ok = (g.lda % 32) == 0 && (g.ldw % 32) == 0 && (g.a_bs % 32) == 0 && (g.a_zs % 32) == 0 && (g.w_zs % 32) == 0;
    const bool out_ok = !out_h2 || ((g.ldc % 32) == 0 && (g.c_bs % 32) == 0 && (g.c_ns % 32) == 0 && (g.c_zs % 32) == 0);   // (N % 4 == 0: a last group of four columns)
    return g.M >= BM && g.N >= 128 && (g.K % BK) == 0 && (g.N % 4) == 0 && (g.n_per >= g.N || (g.n_per % 64) == 0) && g.m_per >= 8 && rows_ok && out_ok &&
           !g.wscale && !g.ln_mode && !g.xb_out && !g.stats_out;
}

int wh_launch_gemm8x(hipStream_t s, bool out_h2, const GemmArgs& g) {
    if (!wh_gemm8x_applicable(g, out_h2)) {
        wh_set_error("k_gemm8x: geometry M %d N %d K %d (lda %ld ldw %ld ldc %ld) not covered", g.M, g.N, g.K, g.lda, g.ldw, g.ldc);
        return WH_ERR_UNSUPPORTED;
    }
    if (out_h2) launch8x<h2>(s, g);
    else launch8x<float>(s, g);
    return WH_OK;
}

// LM head at hundreds of rows, h2 operands: one argmax partial per (256-column tile, row) — layout [part][x_mpad]
bool wh_lm_head_tile_x3_applicable(const SkinnyArgs& a) {
    const char* e = getenv("WH_LM_TILE_MIN_ROWS");   // (0 disables: A/B runs and the parity test flip it between contexts)
    const int min_rows = e ? atoi(e) : 256;
    return min_rows > 0 && a.M >= min_rows && (a.K % BK) == 0 && a.X != nullptr && a.xpart == nullptr && a.wscale == nullptr;
}
int wh_lm_head_tile_x3_parts(const SkinnyArgs& a) { return (a.N + BN - 1) / BN; }
void wh_launch_lm_head_tile_x3(hipStream_t s, const SkinnyArgs& a) {
    const size_t sm = (size_t)NSLOT * SLOT + (size_t)BM * 2 * 4 * 5;   // ring + LayerNorm statistics ([256][2] + four quarter sums)
    dim3 grid(((a.N + BN - 1) / BN) * ((a.M + BM - 1) / BM));
    wh_ensure_dyn_lds((const void*)k_lm_head_tile_x3, sm);
    hipLaunchKernelGGL(k_lm_head_tile_x3, grid, dim3(512), sm, s, a);
}
