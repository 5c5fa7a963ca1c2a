// wh_mlp.hip — the encoder layer's feed-forward block as ONE kernel for gfx950 (bf16 operands, d_model 512):
//
//   x[m][:] += W2 · gelu( LN(x[m][:]) · W1^T + b1 ) + b2          and, for the next LayerNorm: bf16 copy of the new row + its partial sums
//
// It stands in for the MatMul / Gelu / MatMul / Add nodes of an encoder layer behind run_encoder (reference src/main.rs:698-707,
// encoder_model.onnx) and replaces two k_gemm8 launches (fc1 + GELU, fc2 + residual) wherever the LayerNorm fold is on (wh_api.cpp).
// Why one kernel: the two GEMMs on their own are bound by the L2 -> LDS stream of 256 x 256 tiles (64 flop per streamed byte), by an
// epilogue that stores while nothing else runs on the CU (tools/gemm8_stamps.hip: 31-53 % of a workgroup's life), and they move the
// 2048-wide hidden activations through HBM twice (12.6 GB written + read per layer at 2048 clips).  Here a workgroup owns 128 rows and
// ALL 512 output columns: the hidden activations never leave the CU, there is one epilogue per 128 x 512 outputs instead of two per
// 256 x 256, and a streamed byte feeds 89 flop.
//
// Structure (8 waves, one workgroup per CU, all 160 KiB of LDS):
//   * the hidden dimension is walked in chunks of 128 units.  Per chunk: GEMM1  h[128][128] = x_tile[128][512] · W1_chunk^T in 8 k-steps of
//     64 (waves 2 x 4, 64 x 32 per wave, accumulators acc1: 32 registers), the LayerNorm fold + bias + erf GELU on the accumulators
//     (rstd_m (acc - mean_m s_n) + c_n, as k_gemm8<LN = 1>), h rounded to bf16 into an LDS tile H (32 KiB, the layout of an MFMA operand),
//     then GEMM2  acc2[128][512] += H · W2[:, chunk]^T in 4 k-steps of 32 (waves 2 x 4, 64 x 128 per wave, acc2: 128 registers, alive for
//     the whole workgroup).  h is rounded to bf16 exactly where the two-kernel path rounds it (fc1's output), so the numerics are the same.
//   * every k-step of either GEMM is one 32 KiB STAGE of a four-slot ring fed by the LDS-DMA path (global_load_lds_dwordx4), four
//     wave-instructions per wave and stage in both kinds — GEMM1: x rows [128][64 k] + W1 rows [128][64 k] as 128-byte LDS rows (source-side
//     swizzle chunk ^ ((row >> 1) & 7), k_gemm8x's geometry); GEMM2: W2 rows [512 n][32 k] as 64-byte rows (swizzle (row >> 2) & 2, k_gemm8's) —
//     so one counted s_waitcnt vmcnt scheme covers the whole sequence of 12 stages per chunk, three stages in flight across the chunk and
//     GEMM boundaries, one s_barrier per stage.
//   * the per-column fold operands s_n, c_n of a chunk (1 KiB) ride the same path: one extra LDS-DMA instruction of wave 0 per chunk into the head
//     of H while H is idle (GEMM1 phase), counted in that wave's vmcnt waits; read back right before H is rewritten.
//   * measured on MI355X at 256 clips (tools/mlp_check.hip, profiles/r04_enc_mlp.txt): 2,150-2,280 us per launch; without GELU and epilogue 1,511 us =
//     6 MB per tile at 11.9 TB/s, the chip-wide L2 -> LDS rate of this access pattern (the weights hit L2, the 16-fold re-read of the x tile does
//     not: 12 MB pass an XCD's 4 MB L2 between two reads of a line) — the main loop sits on that stream; GELU adds ~250 us, the epilogue ~310.
//     Built, measured and not adopted (tools/attic/enc_mlp_gelu_pieces.hip.txt): GEMM1 as 16 rows x 128 hidden units per wave with the GELU in
//     four pieces beside GEMM2's stages (2,377 us: nine fragment reads per eight MFMAs, and the GELU's VALU time does not hide); GEMM2 stages as
//     whole 128-byte lines [256 n][64 k] (2,384 us); the next stage's LDS-DMA issue ahead of the MFMAs, and the epilogue's residual rows requested
//     two row tiles ahead of their use (both equal within the run-to-run spread: the epilogue waits on the issue of its stores, not on its loads).
//   * epilogue: bias + f32 residual (in place on the residual stream), the row again as bf16 minus its running offset, partial {sum, sum of
//     squares} per (64-column group, row) for k_ln_stats — k_gemm8's LayerNorm-producer contract (GemmArgs::xb_out / stats_out / row_shift).
#include <stdlib.h>

#include "wh_common.h"
#include "wh_kernels.h"

namespace {

constexpr int FM = 128;                 // rows per workgroup
constexpr int FC = 128;                 // hidden units per chunk
constexpr int FD = 512;                 // d_model (compile-time: the accumulator tile is the whole output row)
constexpr int SLOT = 32768, NSLOT = 4;
constexpr int H_OFF = NSLOT * SLOT, H_ROWB = FC * 2;      // H: [128 rows][128 k] bf16, 256-byte rows, 16-byte chunk p of row r at p ^ (r & 15)
constexpr int LDS_BYTES = H_OFF + FM * H_ROWB;            // 163,840 = all of a CU's LDS
constexpr int K1 = FD / 64, K2 = FC / 32, PER_CH = K1 + K2;   // 8 + 4 stages per chunk
constexpr int TM1 = 4, TN1 = 2;         // GEMM1 wave tile 64 x 32
constexpr int TM2 = 4, TN2 = 8;         // GEMM2 wave tile 64 x 128
constexpr int W1_OFF = FM * 128;        // GEMM1 stage: x rows first, then the W1 rows

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ int swz_a(int row) { return (row >> 1) & 7; }   // 128-byte rows (wh_gemm8x.hip)
__device__ __forceinline__ int swz_b(int row) { return (row >> 2) & 2; }   // 64-byte rows (wh_gemm8.hip)
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ABL (tools/mlp_check.hip only; 0 in the library): 1 = no epilogue (accumulators kept alive), 2 = no GELU / fold arithmetic (plain conversion),
// 4 = the bf16 copy as 8-byte stores (the first form of the epilogue)
template <int ABL = 0>
__global__ __launch_bounds__(512, 2) void k_enc_mlp(MlpArgs a) {
    extern __shared__ __attribute__((aligned(128))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int fl = lane & 15, fg = lane >> 4;
    const int total = (a.M + FM - 1) / FM;
    int tile = blockIdx.x;
    {   // XCD-aware order (as k_gemm8): one XCD walks a contiguous run of row tiles
        const int q = total >> 3, r = total & 7, xcd = tile & 7, idx = tile >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = tile * FM;
    const int nch = a.F / FC, Q = nch * PER_CH;
    const bf16* Xt = (const bf16*)a.X;
    const bf16* W1 = (const bf16*)a.W1;
    const bf16* W2 = (const bf16*)a.W2;

    // LayerNorm statistics of this lane's GEMM1 rows (oldest vector-memory operations of the wave: covered by the first counted wait)
    float mean[TM1], rstd[TM1];
#pragma unroll
    for (int i = 0; i < TM1; i++) {
        const int m = min(m0 + wm * 64 + i * 16 + fl, a.M - 1);
        const f32x2 st = *reinterpret_cast<const f32x2*>(a.ln_stat + 2 * (long)m);
        mean[i] = st.x;
        rstd[i] = st.y;
    }
#pragma unroll
    for (int i = 0; i < TM1; i++) asm volatile("" :: "v"(mean[i]), "v"(rstd[i]));   // (used here, so that the compiler's wait for them sits here and not inside the loop)

    // per-lane sources of this wave's share of a stage, recomputed from the lane id at every issue (a dozen VALU instructions against the
    // ~100 cycles an LDS-DMA instruction costs its issuer): kept in registers across the loop they were what the allocator spilled, and a
    // scratch reload in the loop waits for vmcnt(0) — the whole ring.
    //   GEMM1 stage: one wave-instruction = 8 rows x 128 bytes, lane i -> row i / 8, LDS chunk i % 8; 2 instructions of x, 2 of W1
    //   GEMM2 stage: one wave-instruction = 16 rows x 64 bytes, lane i -> row i / 4, LDS chunk i % 4; 4 instructions of W2
    const char* Xb = reinterpret_cast<const char*>(Xt + (long)m0 * a.ldx);   // (uniform bases + 32-bit lane offsets)
    const int rows_here = min(FM, a.M - m0);
    int ic = 0, ir = 0, islot = 0;   // chunk / step / slot of the next stage to issue
    auto issue = [&]() {
        char* base = smem + islot * SLOT;
        int ln = lane;
        asm volatile("" : "+v"(ln));   // (opaque: nothing below is loop-invariant to the compiler)
        if (ir < K1) {
            const char* xs = Xb + ir * 128;
            const char* ws = reinterpret_cast<const char*>(W1 + (long)ic * (FC * FD)) + ir * 128;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int row = wave * 16 + j * 8 + (ln >> 3);
                const unsigned ch = (unsigned)(((ln & 7) ^ swz_a(row)) << 4);
                glds16(xs + (unsigned)(min(row, rows_here - 1) * (int)a.ldx * 2) + ch, base + (wave * 16 + j * 8) * 128);
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int row = wave * 16 + j * 8 + (ln >> 3);
                const unsigned ch = (unsigned)(((ln & 7) ^ swz_a(row)) << 4);
                glds16(ws + (unsigned)(row * FD * 2) + ch, base + W1_OFF + (wave * 16 + j * 8) * 128);
            }
        } else {
            const char* ws = reinterpret_cast<const char*>(W2 + (long)ic * FC + (ir - K1) * 32);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int row = wave * 64 + j * 16 + (ln >> 2);
                glds16(ws + (unsigned)(row * a.F * 2) + (unsigned)(((ln & 3) ^ swz_b(row)) << 4), base + (wave * 64 + j * 16) * 64);
            }
        }
        if (++ir == PER_CH) { ir = 0; ic++; }
        islot = (islot + 1) & (NSLOT - 1);
    };
    f32x4 acc2[TM2][TN2];
#pragma unroll
    for (int i = 0; i < TM2; i++)
#pragma unroll
        for (int j = 0; j < TN2; j++) acc2[i][j] = f32x4{0, 0, 0, 0};

    // fragment byte offsets (row fl of a 16-row tile; tiles start at multiples of 16 rows, so the swizzles depend on fl only)
    const int a1_off = (wm * 64 + fl) * 128, w1f_off = W1_OFF + (wn * 32 + fl) * 128;
    const int ca0 = ((fg) ^ swz_a(fl)) << 4, ca1 = ((4 + fg) ^ swz_a(fl)) << 4;          // the two 32-k halves of a GEMM1 stage
    const int w2f_off = (wn * 128 + fl) * 64 + ((fg ^ swz_b(fl)) << 4);
    const int h_rd = H_OFF + (wm * 64 + fl) * H_ROWB;                                    // + ((kk * 4 + fg) ^ fl) << 4
    char* const h_wr = smem + H_OFF + (wm * 64 + fl) * H_ROWB + (fg & 1) * 8;             // + ((wn * 4 + j * 2 + (fg >> 1)) ^ fl) << 4, + i * 16 rows

    for (int t = 0; t < NSLOT - 1; t++)
        if (t < Q) issue();

    int t = 0, slot = 0;
    // top of a stage: wait until this wave's share of stage t has landed.  Younger operations still allowed in flight: the stages t+1, t+2
    // (four instructions each; nothing else of this wave is a vector-memory operation inside the loop).
    auto stage_top = [&](int r) -> const char* {
        const int out = 4 * min(2, Q - 1 - t);
        if (out == 8) {
            if (wave == 0 && r >= 1 && r <= 3) wait_vm<9>();   // wave 0's fold-operand piece (issued behind stage 12 c + 3) is younger than stages 12 c + 1 .. + 3
            else wait_vm<8>();
        }
        else if (out == 4) wait_vm<4>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();   // stage t visible to all; every wave is done with the slot of stage t - 1 (and, at r = 8, H is written)
        return smem + slot * SLOT;
    };
    auto stage_end = [&](int c, int r) {
        __builtin_amdgcn_sched_barrier(0);   // the MFMAs above are issued ahead of the LDS-DMA instructions below (~100 cycles of issue each)
        if (t + NSLOT - 1 < Q) issue();      // into the slot of stage t - 1
        if (r == 0 && wave == 0) {
            // the chunk's fold operands s_n | c_n (2 x 512 bytes) into the head of H: idle since the previous chunk's last GEMM2 stage, which every
            // wave has left (barrier of this stage); read in the chunk's GELU pass seven stages on
            const float* src = (lane < 32 ? a.s1 : a.c1) + c * FC + (lane & 31) * 4;
            glds16(src, smem + H_OFF);
        }
        slot = (slot + 1) & (NSLOT - 1);
        t++;
    };
    for (int c = 0; c < nch; c++) {
        f32x4 acc1[TM1][TN1];
#pragma unroll
        for (int i = 0; i < TM1; i++)
#pragma unroll
            for (int j = 0; j < TN1; j++) acc1[i][j] = f32x4{0, 0, 0, 0};
        for (int r = 0; r < K1; r++) {
            const char* sb = stage_top(r);
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const int cc = ks ? ca1 : ca0;
                bf16x8 af[TM1], wf[TN1];
#pragma unroll
                for (int i = 0; i < TM1; i++) af[i] = *reinterpret_cast<const bf16x8*>(sb + a1_off + i * 16 * 128 + cc);
#pragma unroll
                for (int j = 0; j < TN1; j++) wf[j] = *reinterpret_cast<const bf16x8*>(sb + w1f_off + j * 16 * 128 + cc);
#pragma unroll
                for (int i = 0; i < TM1; i++)
#pragma unroll
                    for (int j = 0; j < TN1; j++) mma16(acc1[i][j], wf[j], af[i]);   // D rows = hidden unit n, cols = row m
            }
            stage_end(c, r);
        }
        // the chunk's hidden activations: fold + bias + GELU on the accumulators, bf16 into H as GEMM2's activation operand
        // (H was last read four barriers ago, by the previous chunk's GEMM2)
        // The per-column fold operands s_n, c_n of the chunk (1 KiB) were dropped into the first rows of H by wave 0 at the chunk's first stage
        // (see stage_end); every wave takes its 2 x 8 values, then one extra barrier before H is overwritten.
        f32x4 sq[TN1], cq[TN1];
#pragma unroll
        for (int j = 0; j < TN1; j++) {
            sq[j] = *reinterpret_cast<const f32x4*>(smem + H_OFF + (wn * 32 + j * 16 + fg * 4) * 4);
            cq[j] = *reinterpret_cast<const f32x4*>(smem + H_OFF + 512 + (wn * 32 + j * 16 + fg * 4) * 4);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int j = 0; j < TN1; j++) {
            float sv[4], cv[4];
#pragma unroll
            for (int e = 0; e < 4; e++) { sv[e] = sq[j][e]; cv[e] = cq[j][e]; }
#pragma unroll
            for (int i = 0; i < TM1; i++) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = (ABL & 2) ? acc1[i][j][e] + sv[e] : gelu_erf(rstd[i] * (acc1[i][j][e] - mean[i] * sv[e]) + cv[e]);
                *reinterpret_cast<bf16x4*>(h_wr + i * 16 * H_ROWB + (((wn * 4 + j * 2 + (fg >> 1)) ^ fl) << 4)) =
                    bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // H is in LDS before this wave reaches the next barrier
        for (int kk = 0; kk < K2; kk++) {
            const char* sb = stage_top(K1 + kk);
            bf16x8 af[TM2];
#pragma unroll
            for (int i = 0; i < TM2; i++) af[i] = *reinterpret_cast<const bf16x8*>(smem + h_rd + i * 16 * H_ROWB + (((kk * 4 + fg) ^ fl) << 4));
#pragma unroll
            for (int jh = 0; jh < TN2; jh += 4) {   // the weight fragments four at a time: 32 fragment registers instead of 48
                bf16x8 wf[4];
#pragma unroll
                for (int j = 0; j < 4; j++) wf[j] = *reinterpret_cast<const bf16x8*>(sb + w2f_off + (jh + j) * 16 * 64);
#pragma unroll
                for (int i = 0; i < TM2; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) mma16(acc2[i][jh + j], wf[j], af[i]);   // D rows = output column n, cols = row m
            }
            stage_end(c, K1 + kk);
        }
    }

    // ---- epilogue: + bias + residual (in place), bf16 copy minus the row's offset, LayerNorm partial sums ----------------------------------
    if ((ABL & 1) && a.M > 0) {
#pragma unroll
        for (int i = 0; i < TM2; i++)
#pragma unroll
            for (int j = 0; j < TN2; j++) asm volatile("" :: "v"(acc2[i][j]));
        return;
    }
    f32x4 b2v[TN2];
#pragma unroll
    for (int j = 0; j < TN2; j++) b2v[j] = *reinterpret_cast<const f32x4*>(a.b2 + wn * 128 + j * 16 + fg * 4);
#pragma unroll
    for (int i = 0; i < TM2; i++) {
        const int m = m0 + wm * 64 + i * 16 + fl;
        const bool ok = m < a.M;
        const long mr = ok ? m : a.M - 1;
        float* xr = a.Xres + mr * a.ldr + wn * 128 + fg * 4;
        f32x4 rr[TN2];
#pragma unroll
        for (int j = 0; j < TN2; j++) rr[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr + j * 16));
        const float sh = a.row_shift ? a.row_shift[mr] : 0.0f;
        bf16* xbr = (bf16*)a.xb_out + mr * a.ldx + wn * 128 + fg * 4;
        float s1[2] = {0.0f, 0.0f}, s2[2] = {0.0f, 0.0f};
        // the bf16 copy leaves as 16-byte stores: a lane holds 4 columns (8 bytes) of column tile j and of tile j + 1; v_permlane16_swap trades the
        // odd lane groups' tile-j halves for the even groups' tile-(j + 1) halves, after which lane group fg holds 8 consecutive columns of tile
        // j + (fg & 1) (columns 8 (fg >> 1) ..): half the store instructions of the 8-byte form (ABL & 4), whose issue is what the epilogue waits on
        wh_u32x2 pk[TN2];
#pragma unroll
        for (int j = 0; j < TN2; j++) {
            f32x4 v = acc2[i][j] + b2v[j] + rr[j];
            if (ok) *reinterpret_cast<f32x4*>(xr + j * 16) = v;
            v -= sh;
            const bf16x4 b = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
            if (ABL & 4) { if (ok) *reinterpret_cast<bf16x4*>(xbr + j * 16) = b; }
            else __builtin_memcpy(&pk[j], &b, 8);
#pragma unroll
            for (int e = 0; e < 4; e++) { s1[j >> 2] += v[e]; s2[j >> 2] += v[e] * v[e]; }
        }
        if (!(ABL & 4)) {
#pragma unroll
            for (int j = 0; j < TN2; j += 2) {
                // D = tile j, S = tile j + 1: new D = [D.row0, S.row0, D.row2, S.row2], new S = [D.row1, S.row1, D.row3, S.row3] (rows = lane groups)
                const wh_u32x2 lo = __builtin_amdgcn_permlane16_swap(pk[j].x, pk[j + 1].x, false, false);
                const wh_u32x2 hi = __builtin_amdgcn_permlane16_swap(pk[j].y, pk[j + 1].y, false, false);
                // even groups: (D', S') = (own tile-j columns 4 fg .., the next group's 4 (fg + 1) ..); odd groups: (the previous group's tile-(j+1) columns, own)
                const wh_u32x4 w = {lo.x, hi.x, lo.y, hi.y};
                bf16* dst = (bf16*)a.xb_out + mr * a.ldx + wn * 128 + (j + (fg & 1)) * 16 + (fg >> 1) * 8;
                if (ok) *reinterpret_cast<wh_u32x4*>(dst) = w;
            }
        }
#pragma unroll
        for (int g = 0; g < 2; g++) {   // the four lanes (fg) of a row -> one partial per (64-column group, row)
            const float t1 = xrow_sum(s1[g]), t2 = xrow_sum(s2[g]);
            if (fg == 0 && ok) {
                float* sp = a.stats_out + ((long)(wn * 2 + g) * a.stats_rows + m) * 2;
                *reinterpret_cast<f32x2*>(sp) = f32x2{t1, t2};
            }
        }
    }
}

}  // namespace

bool wh_enc_mlp_applicable(const MlpArgs& a) {
    return a.d == FD && a.F >= FC && (a.F % FC) == 0 && a.M >= FM && a.ldx == FD && a.ldr == FD && a.X && a.ln_stat && a.W1 && a.s1 && a.c1 && a.W2 && a.b2 &&
           a.Xres && a.xb_out && a.stats_out && (long)a.F * FD < (1L << 31);
}

int wh_launch_enc_mlp(hipStream_t s, const MlpArgs& a) {
    if (!wh_enc_mlp_applicable(a)) {
        wh_set_error("k_enc_mlp: geometry M %d d %d F %d not covered", a.M, a.d, a.F);
        return WH_ERR_UNSUPPORTED;
    }
    if (!wh_ensure_dyn_lds((const void*)k_enc_mlp<0>, LDS_BYTES)) return WH_ERR_HIP;
    hipLaunchKernelGGL(k_enc_mlp<0>, dim3((unsigned)((a.M + FM - 1) / FM)), dim3(512), LDS_BYTES, s, a);
    return WH_OK;
}
