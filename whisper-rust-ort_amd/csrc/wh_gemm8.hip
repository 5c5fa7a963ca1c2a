// wh_gemm8.hip — the encoder-side bf16 GEMM for gfx950: 8 waves, 256 x {128, 256} x 32 tiles, operands streamed
// global -> LDS by the LDS-DMA path (global_load_lds_dwordx4) through a ring of slots with counted waits.
//
//   C[m][n] = act( wscale[n] * sum_k A[m][k] * W[n][k] + bias ) + R[m][n]          (same contract as k_gemm, wh_gemm.hip)
//
// It stands in for ONNX Runtime's MLAS GEMM / Conv nodes behind run_encoder (reference src/main.rs:698-707) and the
// step-0 cross-attention K/V projection (:771-787) wherever the problem has at least one full tile; k_gemm keeps the
// small shapes and the exact-f32 mode.
//
// Structure:
//   * slot = A tile [256 rows][32 k] + W tile [BN rows][32 k], 64-byte rows.  A wave-instruction of the LDS-DMA writes
//     1 KiB = 16 whole rows, lane i -> row i/4, 16-byte chunk i%4.  LDS is written linearly by construction, so the bank
//     swizzle sits on the SOURCE side: LDS chunk p of row r holds k-chunk p ^ swz(r); the permutation stays inside the
//     row's own 64 bytes, so global reads stay whole lines.  Fragment reads (ds_read_b128: lane l -> row l & 15,
//     k-chunk l >> 4) apply the same XOR and are conflict-free (see swz()).
//   * k-step t:  wait until this wave's slot-t loads have landed (counted s_waitcnt vmcnt: the newer stages stay in
//     flight), raw s_barrier (every wave's slot-t data is there, and every wave has finished reading slot t-1), issue
//     the stage that reuses the slot step t-1 read, then the MFMAs (16x16x32 bf16) on slot t.  One barrier per k-step,
//     never vmcnt(0) inside the loop.
//   * two geometries.  BN = 256: waves 2 x 4, 128 x 64 per wave, 4 slots x 32 KiB, one workgroup per CU, fragments
//     double-buffered in registers (241 VGPRs) — the default: half the L2 -> LDS traffic per flop.  BN = 128: waves 4 x 2,
//     64 x 64 per wave, 3 slots x 24 KiB = 72 KiB, two workgroups per CU (116 VGPRs) — used for the GELU GEMM, whose
//     VALU-heavy epilogue a second workgroup hides under its own MFMAs.
//     Measured on MI355X at 256 clips (tools/gemm8_ablate.hip, which also times ablated variants): main loop alone
//     ~1.0 PF/s in either geometry (BN = 128 is bound by the L2 -> LDS stream, 12-14 TB/s chip-wide; BN = 256 by the
//     barrier-paced MFMA phases), epilogue +50...100 %: a CU's vector-memory path is shared by the LDS-DMA loads, the
//     residual reads and the stores, so a co-resident workgroup's loads queue behind the other's store burst — staggering
//     the two workgroups of a CU by half a tile changed nothing (measured), and was dropped.
//   * the weight tile is the MFMA row operand: a lane ends up with 4 consecutive n of one output row, which makes bias /
//     channel-scale loads and the LDS staging writes 16-byte.
//   * epilogue: each wave parks 32 rows of its sub-tile at a time in its own 8.5 KiB of the (now idle) ring and stores
//     them row-contiguously with 16 bytes per lane in either output type (8-byte-per-lane bf16 stores ran at 3 TB/s,
//     16-byte ones at 5 TB/s); the f32 residual is read the same way.  No workgroup barrier in the epilogue.
//   * XCD-aware tile order as in k_gemm: one XCD walks a contiguous run of tiles, n fastest, so the column tiles that share
//     an activation row panel reuse it from that XCD's L2.
#include <stdlib.h>

#include "wh_common.h"
#include "wh_kernels.h"

namespace {

constexpr int BM = 256, BK = 32;
constexpr int ROWB = BK * 2;                        // 64 bytes per LDS row
constexpr int EP_PITCH = 68;                        // floats per staged output row (64 + 4: conflict-free both ways)
constexpr int EP_ROWS = 32;                         // rows of a wave's sub-tile staged per pass (8.5 KiB per wave)

template <int BN> struct Geo {
    static constexpr int WN = BN / 64, WM = 8 / WN;              // waves along n / m
    static constexpr int TM = BM / WM / 16, TN = 4;              // 16 x 16 MFMA tiles per wave
    static constexpr int SLOT_A = BM * ROWB, SLOT = SLOT_A + BN * ROWB;
    static constexpr int NSLOT = BN == 128 ? 3 : 4;                // ring slots: one being read, the others in flight
    static constexpr int W_INSTR = BN / 128;                     // LDS-DMA instructions of W per wave and stage
    static constexpr int PER_STAGE = 2 + W_INSTR;                // ... of A and W together
    static constexpr int WAVES_PER_SIMD = BN == 128 ? 4 : 2;
    static constexpr bool PIPE = BN == 256;                      // fragment double buffering needs ~240 VGPRs: one workgroup per CU only
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}
// LDS chunk (16 bytes) p of tile row r holds k-chunk p ^ swz(r): with 64-byte rows a 256-byte bank row holds 4 tile rows
// and the 16-lane service groups of ds_read_b128 ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32) mix the k-chunks c and c+1 of
// row quadruples {0,3} and {1,2}: XOR-ing bit 1 of the chunk with bit 3 of the row makes every group hit 16 distinct slots.
__device__ __forceinline__ int swz(int row) { return (row >> 2) & 2; }

#ifdef WH_GEMM8_STAMPS   // tools/gemm8_ablate.hip only (never defined in the library build): s_memtime stamps of lane 0 of every wave into a buffer of their own
__device__ unsigned long long* g_gemm8_stamps;   // [workgroup][wave][16]
#define WH_STAMP(i) do { if (lane == 0) g_gemm8_stamps[((long)blockIdx.x * 8 + wave) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define WH_STAMP_ID() do { if (lane == 0) g_gemm8_stamps[((long)blockIdx.x * 8 + wave) * 16 + 15] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492); } while (0)
#else
#define WH_STAMP(i) do {} while (0)
#define WH_STAMP_ID() do {} while (0)
#endif

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void store8(float* p, const f32x4& a, const f32x4& b) {   // 8 consecutive outputs
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
}
__device__ __forceinline__ void store8(bf16* p, const f32x4& a, const f32x4& b) {
    *reinterpret_cast<bf16x8*>(p) = bf16x8{(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3], (bf16)b[0], (bf16)b[1], (bf16)b[2], (bf16)b[3]};
}

// ABL (tools/gemm8_ablate.hip only; 0 in the library): 1 = no MFMA (fragment reads kept), 2 = no LDS-DMA inside the loop, 32 = no W stage,
// 4 = no epilogue, 8 = no fragment reads and no MFMA (loads + barriers only), 16 = epilogue without its global stores
// LN (compile-time, so that the register budget of the 128-wide geometry holds): 0 = plain epilogue (bias, channel scale);
// 1 / 2 = consumer of a folded LayerNorm with the statistics per output row / per output column (GemmArgs::ln_mode)
template <typename TO, int BN, int ABL = 0, int LN = 0>
__global__ __launch_bounds__(512, Geo<BN>::WAVES_PER_SIMD) void k_gemm8(GemmArgs g) {
    typedef Geo<BN> G;
    constexpr int TM = G::TM, TN = G::TN, NSLOT = G::NSLOT, SLOT = G::SLOT, SLOT_A = G::SLOT_A;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / G::WN, wn = wave % G::WN;
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
    WH_STAMP(0);
    WH_STAMP_ID();
    const bf16* A = (const bf16*)g.A + z * g.a_zs;
    const bf16* W = (const bf16*)g.W + z * g.w_zs;

    // per-lane source pointers of this wave's share of a stage: one wave-instruction = 1 KiB = 16 rows x 64 bytes,
    // lane i -> row i / 4, chunk i % 4; 2 instructions of A (32 rows), 1 or 2 of W (16 or 32 rows) per wave
    const int rl = lane >> 2, ps = lane & 3;
    const bf16* a_src[2];
    const bf16* w_src[G::W_INSTR];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int row = wave * 32 + j * 16 + rl;
        int m = m0 + row;
        if (m > g.M - 1) m = g.M - 1;
        a_src[j] = A + (long)(m / g.m_per) * g.a_bs + (long)(m % g.m_per) * g.lda + ((ps ^ swz(row)) << 3);
    }
#pragma unroll
    for (int j = 0; j < G::W_INSTR; j++) {
        const int row = wave * (16 * G::W_INSTR) + j * 16 + rl;
        int n = n0 + row;
        if (n > g.N - 1) n = g.N - 1;
        w_src[j] = W + (long)n * g.ldw + ((ps ^ swz(row)) << 3);
    }
    auto stage = [&](int slot, int kt) {
        char* base = smem + slot * SLOT;
#pragma unroll
        for (int j = 0; j < 2; j++) glds16(a_src[j] + (long)kt * BK, base + (wave * 32 + j * 16) * ROWB);
        if (ABL & 32) return;   // (ablation: no W stage — what the loop would cost with the weight fragments coming from elsewhere)
#pragma unroll
        for (int j = 0; j < G::W_INSTR; j++) glds16(w_src[j] + (long)kt * BK, base + SLOT_A + (wave * (16 * G::W_INSTR) + j * 16) * ROWB);
    };

    f32x4 acc[TM][TN];   // [m tile][n tile]
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = f32x4{0, 0, 0, 0};

    // fragment byte offsets inside a slot (row-local swizzle: rows of a 16-row tile start at a multiple of 16)
    const int ch = (fg ^ swz(fl)) << 4;
    const int a_off = (wm * (TM * 16) + fl) * ROWB + ch;
    const int w_off = SLOT_A + (wn * 64 + fl) * ROWB + ch;

    // Software pipeline (BN = 256): the fragments of k-step t+1 are read from LDS into a second register set while the
    // MFMAs of k-step t run on the first, so a wave does not stand between a barrier and its MFMAs waiting for ds_reads.
    //   top of step t:  stage t+1 has landed (counted wait) | barrier | issue stage t+NSLOT into the slot whose
    //   fragments were read during step t-1 | read fragments of t+1 | MFMAs of t.
    // BN = 128 (two workgroups per CU, 128 VGPRs each) reads the fragments of step t right before its MFMAs.
    bf16x8 af[G::PIPE ? 2 : 1][TM], wf[G::PIPE ? 2 : 1][TN];
    auto read_frags = [&](int set, int kt) {
        const char* sb = smem + (kt % NSLOT) * SLOT;
#pragma unroll
        for (int i = 0; i < TM; i++) af[set][i] = *reinterpret_cast<const bf16x8*>(sb + a_off + i * 16 * ROWB);
#pragma unroll
        for (int j = 0; j < TN; j++) wf[set][j] = *reinterpret_cast<const bf16x8*>(sb + w_off + j * 16 * ROWB);
    };
    auto mfmas = [&](int set) {
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) {
                if (!(ABL & 1)) mma16(acc[i][j], wf[set][j], af[set][i]);   // D rows = n, cols = m
                else asm volatile("" :: "v"(wf[set][j]), "v"(af[set][i]));
            }
    };
    // wait until this wave's stage `kt` has landed: the younger stages issued so far (at most `cap`) stay outstanding.
    // vmcnt retires in issue order, so "at most newer * PER_STAGE outstanding" == "stage kt and everything older is done".
    auto wait_stage = [&](int kt, int cap) {
        const int newer = (ABL & 2) ? 0 : min(cap, nk - 1 - kt);
        constexpr int PS = (ABL & 32) ? 2 : G::PER_STAGE;
        if (newer >= 3) wait_vm<3 * PS>();
        else if (newer == 2) wait_vm<2 * PS>();
        else if (newer == 1) wait_vm<PS>();
        else wait_vm<0>();
    };
    if (G::PIPE) {
#pragma unroll
        for (int t = 0; t < NSLOT; t++)
            if (t < nk) stage(t, t);
        wait_stage(0, NSLOT - 1);           // the whole ring was just issued: slots 1 .. NSLOT-1 may still be in flight
        __builtin_amdgcn_s_barrier();
        if (!(ABL & 8)) read_frags(0, 0);
        WH_STAMP(12);
        // two k-steps per iteration so the register sets are named statically
        for (int kt = 0; kt < nk; kt += 2) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int t = kt + h;
                if (t >= nk) break;
                // Order inside a k-step (round 4): the MFMAs of step t (operands in registers since step t - 1) are issued FIRST, the LDS-DMA
                // instructions of the next stage and the fragment reads of t + 1 behind them.  An LDS-DMA wave-instruction costs its issuer
                // ~100 cycles (MI355X_MICROARCH.md, "LDS-DMA piece issue cost") and a wave issues in order: with the four of them ahead of
                // the MFMAs both waves of a SIMD stood in DMA issue right after every barrier while the matrix pipe idled.
                if (t + 1 < nk) {
                    wait_stage(t + 1, NSLOT - 2);       // in flight here: stages t+1 .. t+NSLOT-1 (t+NSLOT is issued below)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's fragment reads of step t have left LDS
                    __builtin_amdgcn_s_barrier();       // stage t+1 visible to all; every wave has read the fragments of t
                }
                if (!(ABL & 8)) mfmas(h & (G::PIPE ? 1 : 0));
                if (t + 1 < nk) {
                    __builtin_amdgcn_sched_barrier(0);  // keep the MFMAs above ahead of the DMA issue below
                    if (t + NSLOT < nk) {               // slot of stage t is free again
                        if (!(ABL & 2)) stage(t % NSLOT, t + NSLOT);
                        else asm volatile("s_nop 0" ::: "memory");
                    }
                    if (!(ABL & 8)) read_frags((h ^ 1) & (G::PIPE ? 1 : 0), t + 1);
                }
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < NSLOT - 1; t++)
            if (t < nk) stage(t, t);
        WH_STAMP(12);
        for (int kt = 0; kt < nk; kt++) {
            wait_stage(kt, NSLOT - 2);          // in flight: stages kt .. kt+NSLOT-2
            __builtin_amdgcn_s_barrier();       // stage kt visible to all; every wave has consumed the fragments of kt-1
            if (!(ABL & 8)) {
                read_frags(0, kt);
                mfmas(0);                       // (issued before the DMA below: its issue cost then falls under the MFMAs' execution)
                __builtin_amdgcn_sched_barrier(0);
            }
            if (kt + NSLOT - 1 < nk) {
                if (!(ABL & 2)) stage((kt + NSLOT - 1) % NSLOT, kt + NSLOT - 1);
                else asm volatile("s_nop 0" ::: "memory");
            }
        }
    }
    WH_STAMP(1);
    __builtin_amdgcn_s_barrier();   // every wave is done with the ring: it becomes the output staging area
    WH_STAMP(2);
    if ((ABL & 4) && g.M > 0) {     // keep the accumulators alive, store nothing
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) asm volatile("" :: "v"(acc[i][j]));
        return;
    }

    // ---- epilogue: passes of 32 rows per wave through the wave's own 8.5 KiB of the idle ring ------------------------
    float* stg = reinterpret_cast<float*>(smem) + wave * (EP_ROWS * EP_PITCH);
    const int nw0 = n0 + wn * 64;            // first column of this wave's sub-tile
    const int mw0 = m0 + wm * (TM * 16);
    TO* C = (TO*)g.C + z * g.c_zs;
    const float* R = g.R ? g.R + z * g.r_zs : nullptr;
    const long nc0 = (long)(nw0 / g.n_per) * g.c_ns + (nw0 % g.n_per);   // n_per is a multiple of 64 or >= N (checked at launch)
    // per-column operands of this lane's 4 consecutive n per tile — LN 0: bias, channel scale; LN 1: c[n] (as bias), s[n];
    // LN 2: mean[n], rstd[n] (bias and s are per row there)
    f32x4 pb[TN], pw[TN];
    const float* lstat = LN ? g.ln_stat + z * g.ln_stat_zs : nullptr;
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int n = nw0 + j * 16 + fg * 4;
        pb[j] = f32x4{0, 0, 0, 0};
        pw[j] = f32x4{1, 1, 1, 1};
        if (LN == 2) {
            if (n < g.N) {   // {mean, rstd} interleaved per column (N % 4 == 0: whole groups of four)
                const f32x4 u0 = *reinterpret_cast<const f32x4*>(lstat + 2 * (long)n), u1 = *reinterpret_cast<const f32x4*>(lstat + 2 * (long)n + 4);
                pb[j] = f32x4{u0[0], u0[2], u1[0], u1[2]};
                pw[j] = f32x4{u0[1], u0[3], u1[1], u1[3]};
            }
        } else if (n < g.N && g.bias_mode == 1) {
            if (g.bias) pb[j] = *reinterpret_cast<const f32x4*>(g.bias + n);
            if (LN == 1) pw[j] = *reinterpret_cast<const f32x4*>(g.ln_s + n);
            else if (g.wscale) pw[j] = *reinterpret_cast<const f32x4*>(g.wscale + n);
        }
    }
    // per-ROW operands of all of this lane's TM rows, requested together (one round trip instead of one per 16-row tile):
    // LN 1: rowa = mean[m], rowb = rstd[m];  per-row bias (bias_mode 2): rowb = bias[m] / c[m], rowa = channel scale or (LN 2) s[m]
    float rowa[TM], rowb[TM];
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int m = min(mw0 + i * 16 + fl, g.M - 1);
        rowa[i] = LN == 1 ? 0.0f : 1.0f;
        rowb[i] = LN == 1 ? 1.0f : 0.0f;
        if (LN == 1) {
            typedef __attribute__((ext_vector_type(2))) float f32x2;
            const f32x2 st = *reinterpret_cast<const f32x2*>(lstat + 2 * (long)m);
            rowa[i] = st.x;
            rowb[i] = st.y;
        } else if (g.bias_mode == 2) {
            if (g.bias) rowb[i] = g.bias[m];
            if (LN == 2) rowa[i] = g.ln_s[m];
            else if (g.wscale) rowa[i] = g.wscale[m];
        }
    }
    WH_STAMP(3);
    // row-contiguous read-back: 8 lanes x 8 columns per row, 8 rows per wave-instruction
    const int c8 = (lane & 7) * 8, r8 = lane >> 3;
    const int n_st = nw0 + c8;
#pragma unroll
    for (int pass = 0; pass < TM / 2; pass++) {
        // the residual rows of this pass, requested before its staging writes: four independent loads in flight under the
        // VALU / LDS work below instead of one memory round trip per 8-row group (each behind the previous group's store)
        // (256-wide geometry only: the 128-wide one runs two workgroups per CU on a 128-register budget that has no room for it)
        constexpr bool RHOIST = BN == 256;
        f32x4 rr0[RHOIST ? EP_ROWS / 8 : 1], rr1[RHOIST ? EP_ROWS / 8 : 1];
        if (RHOIST && R) {
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
            const float bm = rowb[i], wmul = rowa[i];   // LN 1: bm = rstd[m], wmul = mean[m];  LN 2: bm = c[m], wmul = s[m]
#pragma unroll
            for (int j = 0; j < TN; j++) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    if (LN == 1) v[e] = bm * (acc[i][j][e] - wmul * pw[j][e]) + pb[j][e];          // rstd (acc - mean s[n]) + c[n]
                    else if (LN == 2) v[e] = pw[j][e] * (acc[i][j][e] - pb[j][e] * wmul) + bm;     // rstd[n] (acc - mean[n] s[m]) + c[m]
                    else v[e] = acc[i][j][e] * (pw[j][e] * wmul) + (pb[j][e] + bm);
                }
                if (g.act == 1) {
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = gelu_erf(v[e]);
                }
                *reinterpret_cast<f32x4*>(&stg[(ii * 16 + fl) * EP_PITCH + j * 16 + fg * 4]) = f32x4{v[0], v[1], v[2], v[3]};
            }
        }
        WH_STAMP(4 + 2 * pass);
        // wave-private staging: the wave's own LDS writes are ordered before its reads by the lgkmcnt wait the compiler inserts
        const int mp0 = mw0 + pass * EP_ROWS + r8;       // this lane's first row of the pass; its rows are mp0 + 8 * it
        long mb = mp0 / g.m_per, mi = mp0 % g.m_per;     // row -> (block, row in block), advanced without dividing again
#pragma unroll
        for (int it = 0; it < EP_ROWS / 8; it++) {
            const int lr = it * 8 + r8, m = mp0 + it * 8;
            float s1 = 0.0f, s2 = 0.0f;   // LayerNorm partial sums of this lane's 8 values (producers)
            if (m < g.M && n_st < g.N) {
                f32x4 v0 = *reinterpret_cast<const f32x4*>(&stg[lr * EP_PITCH + c8]);
                f32x4 v1 = *reinterpret_cast<const f32x4*>(&stg[lr * EP_PITCH + c8 + 4]);
                if (R) {
                    if constexpr (RHOIST) {
                        v0 += rr0[it];
                        v1 += rr1[it];
                    } else {
                        const float* rp = R + mb * g.r_bs + mi * g.ldr + n_st;
                        v0 += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(rp));  // read once
                        v1 += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(rp + 4));
                    }
                }
                TO* cp = C + mb * g.c_bs + mi * g.ldc + nc0 + c8;
                if (ABL & 16) asm volatile("" :: "v"(v0), "v"(v1));   // everything but the store
                else if (n_st + 8 <= g.N) store8(cp, v0, v1);
                else store4(cp, v0[0], v0[1], v0[2], v0[3]);          // N % 8 == 4: the last group holds 4 valid columns
                if constexpr (sizeof(TO) == 4) {
                    // producer of a LayerNorm input (N % 64 == 0 and contiguous rows checked at launch): the row segment again as
                    // bf16 — the consumer GEMMs' operand — and its contribution to the row's {sum, sum of squares}; both of the row
                    // minus its running offset (GemmArgs::row_shift)
                    if (g.row_shift) {
                        const float sh = g.row_shift[m];
                        v0 -= sh;
                        v1 -= sh;
                    }
                    if (g.xb_out) store8((bf16*)g.xb_out + z * g.c_zs + mb * g.c_bs + mi * g.ldc + nc0 + c8, v0, v1);
                    if (g.stats_out) {
#pragma unroll
                        for (int e = 0; e < 4; e++) { s1 += v0[e]; s2 += v0[e] * v0[e]; }
#pragma unroll
                        for (int e = 0; e < 4; e++) { s1 += v1[e]; s2 += v1[e] * v1[e]; }
                    }
                }
            }
            if constexpr (sizeof(TO) == 4) {
                if (g.stats_out) {    // the 8 lanes of a row -> one partial per (64-column group, row): DPP only, fixed order
                    s1 = dpp_group_sum<8>(s1);
                    s2 = dpp_group_sum<8>(s2);
                    if ((lane & 7) == 0 && m < g.M && n_st < g.N) {
                        float* sp = g.stats_out + ((long)(nw0 >> 6) * g.stats_rows + m) * 2;
                        sp[0] = s1;
                        sp[1] = s2;
                    }
                }
            }
            mi += 8;
            if (mi >= g.m_per) { mi -= g.m_per; mb += 1; }
        }
        WH_STAMP(5 + 2 * pass);
    }
}


// ---- the LM head at hundreds of rows on the same tile structure ------------------------------------------------------------
// logits = LN(x) · E^T over the whole vocabulary + masked argmax partials (k_lm_head's contract, wh_decode.hip; reference
// argmax_last_dim_raw, src/main.rs:709-735).  k_lm_head stages a 64-row activation tile per workgroup and streams the 53 MB
// tied embedding once per 64 rows: 850 MB through L2 per launch at 1024 rows.  Here a workgroup owns a 256 x 256 logit tile
// (k_gemm8's BN = 256 geometry and LDS-DMA ring; the activation rows come from the decode slab layout [K/32][mpad][32], which
// is an LDS tile per k-step as it stands), so the embedding passes L2 once per 256 rows; the epilogue never stores the tile —
// it folds the final LayerNorm (rstd (acc - mean s[n]) + c[n]), applies the suppress mask and keeps one (max, index) per row and
// wave, merged over the four waves of a row through LDS: one partial per (column tile, row).
// Bit-identical logits: the same MFMA chain over k as k_lm_head (weights as the row operand, k ascending, one accumulator),
// LayerNorm partial sums reduced in the same order, the same epilogue expression — so the launcher may pick by the call's
// row count (tests/test_hip_parity.py::test_wide_batch_decode_gemm_is_bit_identical runs both).
__global__ __launch_bounds__(512, 2) void k_lm_head_tile(SkinnyArgs a) {
    typedef Geo<256> G;
    constexpr int BN = 256, TM = G::TM, TN = G::TN, NSLOT = G::NSLOT, SLOT = G::SLOT, SLOT_A = G::SLOT_A;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / G::WN, wn = wave % G::WN;
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

    // final LayerNorm: quarter sums of the producer's per-tile partials, two quarters per thread (row tid & 255) — requested
    // before the ring, so they are the oldest vector-memory requests
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

    const int rl = lane >> 2, ps = lane & 3;
    const bf16* a_src[2];
    const bf16* w_src[2];
    const long a_kstep = (long)a.x_mpad * 32;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int row = wave * 32 + j * 16 + rl;
        a_src[j] = (const bf16*)a.X + (long)min(m0 + row, a.x_mpad - 1) * 32 + ((ps ^ swz(row)) << 3);
        w_src[j] = (const bf16*)a.W + (long)min(n0 + row, a.N - 1) * a.K + ((ps ^ swz(row)) << 3);
    }
    auto stage = [&](int slot, int kt) {
        char* base = smem + slot * SLOT;
#pragma unroll
        for (int j = 0; j < 2; j++) glds16(a_src[j] + kt * a_kstep, base + (wave * 32 + j * 16) * ROWB);
#pragma unroll
        for (int j = 0; j < 2; j++) glds16(w_src[j] + (long)kt * BK, base + SLOT_A + (wave * 32 + j * 16) * ROWB);
    };
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = f32x4{0, 0, 0, 0};
    const int ch = (fg ^ swz(fl)) << 4;
    const int a_off = (wm * (TM * 16) + fl) * ROWB + ch;
    const int w_off = SLOT_A + (wn * 64 + fl) * ROWB + ch;
    bf16x8 af[2][TM], wf[2][TN];
    auto read_frags = [&](int set, int kt) {
        const char* sb = smem + (kt % NSLOT) * SLOT;
#pragma unroll
        for (int i = 0; i < TM; i++) af[set][i] = *reinterpret_cast<const bf16x8*>(sb + a_off + i * 16 * ROWB);
#pragma unroll
        for (int j = 0; j < TN; j++) wf[set][j] = *reinterpret_cast<const bf16x8*>(sb + w_off + j * 16 * ROWB);
    };
    auto mfmas = [&](int set) {
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) mma16(acc[i][j], wf[set][j], af[set][i]);   // D rows = n, cols = m
    };
    auto wait_stage = [&](int kt, int cap) {
        const int newer = min(cap, nk - 1 - kt);
        constexpr int PS = G::PER_STAGE;
        if (newer >= 3) wait_vm<3 * PS>();
        else if (newer == 2) wait_vm<2 * PS>();
        else if (newer == 1) wait_vm<PS>();
        else wait_vm<0>();
    };
#pragma unroll
    for (int t = 0; t < NSLOT; t++)
        if (t < nk) stage(t, t);
    wait_stage(0, NSLOT - 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's quarter sums have left for LDS
    __builtin_amdgcn_s_barrier();          // stage 0 visible; the LayerNorm quarter sums are in LDS
    if (a.ln_part && tid < BM) {
        const float s1 = (lnq[tid * 2] + lnq[(BM + tid) * 2]) + (lnq[(2 * BM + tid) * 2] + lnq[(3 * BM + tid) * 2]);
        const float s2 = (lnq[tid * 2 + 1] + lnq[(BM + tid) * 2 + 1]) + (lnq[(2 * BM + tid) * 2 + 1] + lnq[(3 * BM + tid) * 2 + 1]);
        float mean, rstd;
        wh_ln_mean_rstd(s1, s2, (float)a.K, false, mean, rstd);
        lnstat[2 * tid] = mean;
        lnstat[2 * tid + 1] = rstd;
    }
    read_frags(0, 0);
    for (int kt = 0; kt < nk; kt += 2) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int t = kt + h;
            if (t >= nk) break;
            if (t + 1 < nk) {
                wait_stage(t + 1, NSLOT - 2);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            mfmas(h);   // ahead of the DMA issue (k_gemm8's order)
            if (t + 1 < nk) {
                __builtin_amdgcn_sched_barrier(0);
                if (t + NSLOT < nk) stage(t % NSLOT, t + NSLOT);
                read_frags(h ^ 1, t + 1);
            }
        }
    }
    __syncthreads();   // lnstat written by the first 256 threads is visible to everyone (and every MFMA has its operands)

    // ---- epilogue: final LayerNorm fold + masked argmax, one partial per (wave, row) ------------------------------------
    const int pos = *a.pos_p;
    const int gen = pos - (a.n_prompt - 1);  // index of the token this row generates
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
        if (n < a.N) mbits[j] = mask[n >> 5] >> (n & 31);   // suppress bits of this lane's 4 columns
    }
    float* red_v = reinterpret_cast<float*>(smem);            // [4][256] — every wave is past the last barrier of the main loop
    int* red_i = reinterpret_cast<int*>(smem) + G::WN * BM;
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
        // argmax over the four lane groups of the row (k_lm_head's reduction)
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
        if (fg == 0) {   // this wave's (max, index) of row rloc: the four waves that share the row meet in LDS (the ring is idle now)
            red_v[wn * BM + rloc] = bv;
            red_i[wn * BM + rloc] = bi;
        }
    }
    __syncthreads();
    // one partial per (column tile, row): 203 instead of 812 partials per row for k_argmax_finish to read (strided by the row pitch)
    if (tid < BM && m0 + tid < a.M) {
        float bv = red_v[tid];
        int bi = red_i[tid];
#pragma unroll
        for (int w = 1; w < G::WN; w++) {
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


template <typename TO, int BN, int LN = 0>
void launch8(hipStream_t s, const GemmArgs& g) {
    typedef Geo<BN> G;
    const size_t sm = (size_t)G::NSLOT * G::SLOT;
    static_assert((size_t)8 * EP_ROWS * EP_PITCH * 4 <= (size_t)G::NSLOT * G::SLOT, "output staging must fit the ring");
    dim3 grid(((g.N + BN - 1) / BN) * ((g.M + BM - 1) / BM), 1, g.batch);
    wh_ensure_dyn_lds((const void*)k_gemm8<TO, BN, 0, LN>, sm);
    hipLaunchKernelGGL((k_gemm8<TO, BN, 0, LN>), grid, dim3(512), sm, s, g);
}

}  // namespace

bool wh_gemm8_applicable(const GemmArgs& g) {
    // N tails are handled by clamping + masking in 8-column groups (a last group of 4); the column-plane mapping needs 64-column granularity;
    // the row -> (block, row) walk of the epilogue advances by 8 rows at a time
    return g.M >= BM && g.N >= 128 && (g.K % BK) == 0 && (g.N % 4) == 0 && (g.n_per >= g.N || (g.n_per % 64) == 0) && g.m_per >= 8;
}

// {mean, rstd} per row from the producers' per-(64-column group, row) partial sums; groups are added in order (deterministic)
__global__ __launch_bounds__(256) void k_ln_stats(const float* __restrict__ partials, int groups, long rows, float inv_d, float* __restrict__ stat,
                                                  float* __restrict__ shift, const float* __restrict__ shift_in) {
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    float s1 = 0.0f, s2 = 0.0f;
    for (int g0 = 0; g0 < groups; g0 += 8) {   // eight loads in flight
        f32x2 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const f32x2*>(partials + ((long)min(g0 + u, groups - 1) * rows + r) * 2);
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (g0 + u < groups) { s1 += v[u].x; s2 += v[u].y; }
    }
    float mean, rstd;
    wh_ln_mean_rstd(s1, s2, inv_d, true, mean, rstd);
    *reinterpret_cast<f32x2*>(stat + 2 * r) = f32x2{mean, rstd};
    if (shift) shift[r] = (shift_in ? shift_in[r] : 0.0f) + mean;   // offset the producer used + the mean measured on its shifted rows = the row's true mean
}

// LM head at hundreds of rows (bf16 operands): argmax partials per row = column tiles x 4 (layout [part][x_mpad])
bool wh_lm_head_tile_applicable(const SkinnyArgs& a) {
    const char* e = getenv("WH_LM_TILE_MIN_ROWS");   // (0 disables: A/B runs and the parity test flip it between contexts)
    const int min_rows = e ? atoi(e) : 256;
    return min_rows > 0 && a.M >= min_rows && (a.K % BK) == 0 && a.K / BK >= 4 && a.X != nullptr && a.xpart == nullptr && a.wscale == nullptr;
}
int wh_lm_head_tile_parts(const SkinnyArgs& a) { return (a.N + 255) / 256; }
void wh_launch_lm_head_tile(hipStream_t s, const SkinnyArgs& a) {
    typedef Geo<256> G;
    const size_t sm = (size_t)G::NSLOT * G::SLOT + (size_t)BM * 2 * 4 * 5;   // ring + LayerNorm statistics ([256][2] + four quarter sums)
    dim3 grid(((a.N + 255) / 256) * ((a.M + BM - 1) / BM));
    wh_ensure_dyn_lds((const void*)k_lm_head_tile, sm);
    hipLaunchKernelGGL(k_lm_head_tile, grid, dim3(512), sm, s, a);
}

void wh_launch_ln_stats(hipStream_t s, const float* partials, int groups, long rows, int d, float* stat, float* shift, const float* shift_in) {
    hipLaunchKernelGGL(k_ln_stats, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, partials, groups, rows, 1.0f / (float)d, stat, shift, shift_in);
}

int wh_launch_gemm8(hipStream_t s, bool out_f32, const GemmArgs& g) {
    if (!wh_gemm8_applicable(g)) {
        wh_set_error("k_gemm8: geometry M %d N %d K %d not covered", g.M, g.N, g.K);
        return WH_ERR_UNSUPPORTED;
    }
    if ((g.xb_out || g.stats_out) && (!out_f32 || (g.N % 64) != 0 || g.n_per < g.N || (g.m_per < g.M && g.c_bs != (long)g.m_per * g.ldc) || g.batch != 1)) {
        wh_set_error("k_gemm8: LayerNorm-producer outputs need an f32 result with contiguous rows and N a multiple of 64");
        return WH_ERR_UNSUPPORTED;
    }
    // measured per shape (tools/gemm8_ablate.hip): the GELU epilogue is VALU work that a second workgroup on the CU hides
    // under its own MFMAs (fc1: 1.30 ms with BN = 128 against 1.43 ms); everything else is faster or equal with the larger
    // tile's halved L2 -> LDS traffic
    const bool wide = g.act == 0 && g.N >= 256;
    if (g.ln_mode) {   // consumers of a folded LayerNorm: bf16 results only, no channel scale (the bf16 encoder path)
        if (out_f32 || g.wscale || !g.ln_stat || !g.ln_s || (g.ln_mode == 1 && g.bias_mode != 1) || (g.ln_mode == 2 && (g.bias_mode != 2 || (g.N % 4) != 0))) {
            wh_set_error("k_gemm8: unsupported LayerNorm-fold arguments (mode %d)", g.ln_mode);
            return WH_ERR_UNSUPPORTED;
        }
        if (g.ln_mode == 1) { if (wide) launch8<bf16, 256, 1>(s, g); else launch8<bf16, 128, 1>(s, g); }
        else { if (wide) launch8<bf16, 256, 2>(s, g); else launch8<bf16, 128, 2>(s, g); }
        return WH_OK;
    }
    if (out_f32) { if (wide) launch8<float, 256>(s, g); else launch8<float, 128>(s, g); }
    else { if (wide) launch8<bf16, 256>(s, g); else launch8<bf16, 128>(s, g); }
    return WH_OK;
}
