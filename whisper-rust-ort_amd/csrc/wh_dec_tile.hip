// wh_dec_tile.hip — the decode GEMMs of contexts with a thousand clips and more as LDS-DMA tile GEMMs (gfx950; bf16 and
// WH_PREC_F16X3 operands).
//
//   C[m][n] = act( LN-fold( sum_k X[m][k] W[n][k] ) + bias[n] ) (+ R[m][n])          (the contract of k_dec_gemm, wh_decode.hip)
//
// The with-past loop of greedy_decode_with_past (reference src/main.rs:793-826) runs seven such products per decoder layer and
// position.  k_dec_gemm / k_dec_gemm_wide stream weights and activations straight from L2 into MFMA fragments: right for tens of
// rows, where a launch is one weight pass, but at 2048 rows (one row per clip of the batch) they are ordinary GEMMs of 1-4 GFLOP
// and every 64 x 64 output block pulls its own 2 x 64 x K operand bytes through L2 (QKV: 20 us in bf16 for 3.2 GFLOP).  Here a
// workgroup owns a 128 x {128, 64} output tile, both operands go global -> LDS on the LDS-DMA path (global_load_lds_dwordx4) through a
// ring of slots — the decode slab layout [K/32][mpad][32] is an LDS tile per k-step as it stands — and four waves (2 x 2) run the
// MFMAs on 64 x {64, 32} sub-tiles from LDS fragments; one barrier per k-step, counted vmcnt waits (k_gemm8's loop, wh_gemm8.hip).
// Epilogue per lane, as in k_dec_gemm (the weight tile is the MFMA row operand, so a lane holds 4 consecutive columns of one row):
// LayerNorm fold from the producer's partial sums, bias, erf-GELU, f32 residual; row-major or slab output; the raw slab copy and the
// per-16-column LayerNorm partial sums for the next consumer; the position ticket.
//
// Results are NOT bit-identical to k_dec_gemm (one accumulator over all of K instead of a K split over waves), so the choice
// belongs to the CONTEXT (wh_ctx::dec_tile, from its capacity), never to a call: a clip decodes identically whatever shares its batch.
#include <stdlib.h>

#include "wh_common.h"
#include "wh_kernels.h"

namespace {

constexpr int BM = 128, BK = 32;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// operand-type traits: bytes of a tile row per k-step, rows per LDS-DMA wave-instruction (1 KiB), bank swizzle, fragment read
template <typename T> struct OT;
template <> struct OT<bf16> {
    static constexpr int ROWB = 64, RPI = 16, CPR = 4;
    static __device__ __forceinline__ int swz(int row) { return (row >> 2) & 2; }   // k_gemm8's (wh_gemm8.hip)
    static __device__ __forceinline__ bf16x8 frag(const char* rowp, int fl, int fg) {
        return *reinterpret_cast<const bf16x8*>(rowp + ((fg ^ swz(fl)) << 4));
    }
};
template <> struct OT<h2> {
    static constexpr int ROWB = 128, RPI = 8, CPR = 8;
    static __device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }    // k_gemm8x's (wh_gemm8x.hip)
    static __device__ __forceinline__ xfrag frag(const char* rowp, int fl, int fg) {
        xfrag f;
        f.hi = *reinterpret_cast<const f16x8*>(rowp + ((fg ^ swz(fl)) << 4));
        f.lo = *reinterpret_cast<const f16x8*>(rowp + (((4 + fg) ^ swz(fl)) << 4));
        return f;
    }
};

__device__ __forceinline__ long slab_idx(int m, int k, int mpad) { return ((long)(k >> 5) * mpad + m) * 32 + (k & 31); }

template <typename T, typename TO, int BN, int NSLOT>
__global__ __launch_bounds__(256) void k_dec_tile(SkinnyArgs a) {
    typedef OT<T> O;
    typedef typename FragT<T>::type frag_t;
    constexpr int ROWB = O::ROWB, RPI = O::RPI, CPR = O::CPR;
    constexpr int SLOT_A = BM * ROWB, SLOT = SLOT_A + BN * ROWB;
    constexpr int A_INSTR = BM / RPI / 4, W_INSTR = BN / RPI / 4;   // LDS-DMA wave-instructions per wave and stage
    constexpr int PER_STAGE = A_INSTR + W_INSTR;
    constexpr int TM = 4, TN = BN / 32;                              // 16 x 16 tiles per wave: 64 rows x BN / 2 columns
    extern __shared__ __attribute__((aligned(128))) char smem[];
    {   // grouped launches (SkinnyArgs::zn): group blockIdx.z works on its own X, W, bias and C (a slab: type T)
        const long z = blockIdx.z;
        if (z) {
            a.X = (const T*)a.X + z * a.x_zs;
            a.W = (const T*)a.W + z * a.w_zs;
            a.C = a.c_mpad ? (void*)((T*)a.C + z * a.c_zs) : (void*)((TO*)a.C + z * a.c_zs);
            if (a.bias) a.bias += z * a.bias_zs;
        }
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fl = lane & 15, fg = lane >> 4;
    const int nk = a.K / BK;
    const int nbn = (a.N + BN - 1) / BN;
    const int total = nbn * ((a.M + BM - 1) / BM);
    int tile = blockIdx.x;
    {   // XCD-aware order: one XCD walks a contiguous run of tiles, n fastest (the column tiles of a row panel share it through one L2)
        const int q = total >> 3, r = total & 7, xcd = tile & 7, idx = tile >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (tile / nbn) * BM, n0 = (tile % nbn) * BN;

    // LayerNorm folded in: {mean, rstd} of this tile's 128 rows from the producer's per-tile partial sums (two halves of the tiles per row,
    // met through LDS) — requested before the ring, so they are the oldest vector-memory requests
    float* lnstat = reinterpret_cast<float*>(smem + (size_t)NSLOT * SLOT);   // [128][2] mean, rstd
    float* lnq = lnstat + 2 * BM;                                            // [2][128][2]
    if (a.ln_part) {
        const int r = tid & (BM - 1), h = tid >> 7, row = min(m0 + r, a.x_mpad - 1);
        float s1, s2;
        ln_partial_sum(a.ln_part, a.ln_tiles, a.x_mpad, row, h, 2, s1, s2);
        lnq[(h * BM + r) * 2] = s1;
        lnq[(h * BM + r) * 2 + 1] = s2;
    }

    // per-lane source addresses of this wave's share of a stage: lane i -> row i / CPR of the instruction's RPI rows, LDS chunk i % CPR
    const int rl = lane / CPR, ps = lane % CPR;
    const char* a_src[A_INSTR];
    const char* w_src[W_INSTR];
    const long a_kstep = (long)a.x_mpad * ROWB;
#pragma unroll
    for (int j = 0; j < A_INSTR; j++) {
        const int row = (wave * A_INSTR + j) * RPI + rl;
        a_src[j] = reinterpret_cast<const char*>(a.X) + (long)min(m0 + row, a.x_mpad - 1) * ROWB + ((ps ^ O::swz(row)) << 4);
    }
#pragma unroll
    for (int j = 0; j < W_INSTR; j++) {
        const int row = (wave * W_INSTR + j) * RPI + rl;
        w_src[j] = reinterpret_cast<const char*>(a.W) + ((long)min(n0 + row, a.N - 1) * a.K) * (ROWB / 32) + ((ps ^ O::swz(row)) << 4);
    }
    auto stage = [&](int slot, int kt) {
        char* base = smem + slot * SLOT;
#pragma unroll
        for (int j = 0; j < A_INSTR; j++) glds16(a_src[j] + kt * a_kstep, base + (wave * A_INSTR + j) * RPI * ROWB);
#pragma unroll
        for (int j = 0; j < W_INSTR; j++) glds16(w_src[j] + (long)kt * ROWB, base + SLOT_A + (wave * W_INSTR + j) * RPI * ROWB);
    };
    auto wait_stage = [&](int kt) {   // this wave's stage kt has landed; the younger stages (at most NSLOT - 2) stay in flight
        const int newer = min(NSLOT - 2, nk - 1 - kt);
        if (newer >= 2) wait_vm<2 * PER_STAGE>();
        else if (newer == 1) wait_vm<PER_STAGE>();
        else wait_vm<0>();
    };
    static_assert(NSLOT >= 2 && NSLOT <= 4, "ring of two to four slots");

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = f32x4{0, 0, 0, 0};

    const int a_row = (wm * 64 + fl) * ROWB, w_row = SLOT_A + (wn * (BN / 2) + fl) * ROWB;
#pragma unroll
    for (int t = 0; t < NSLOT - 1; t++)
        if (t < nk) stage(t, t);
    for (int kt = 0; kt < nk; kt++) {
        wait_stage(kt);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's fragment reads of kt - 1 (and, first, its LayerNorm sums) have left LDS
        __builtin_amdgcn_s_barrier();                          // stage kt visible to all; every wave is done with stage kt - 1
        if (kt == 0 && a.ln_part && tid < BM) {
            const float s1 = lnq[tid * 2] + lnq[(BM + tid) * 2], s2 = lnq[tid * 2 + 1] + lnq[(BM + tid) * 2 + 1];
            float mean, rstd;
            wh_ln_mean_rstd(s1, s2, (float)a.K, false, mean, rstd);
            lnstat[2 * tid] = mean;
            lnstat[2 * tid + 1] = rstd;
            // the one consumer of a LayerNorm keeps the rows' running offsets up to date (first column tile only)
            if (a.shift_io && n0 == 0 && blockIdx.z == 0 && m0 + tid < a.M) a.shift_io[m0 + tid] += mean;
        }
        const char* sb = smem + (kt % NSLOT) * SLOT;
        frag_t wf[TN];
#pragma unroll
        for (int j = 0; j < TN; j++) wf[j] = O::frag(sb + w_row + j * 16 * ROWB, fl, fg);
#pragma unroll
        for (int i0 = 0; i0 < TM; i0 += 2) {
            frag_t af[2];
#pragma unroll
            for (int u = 0; u < 2; u++) af[u] = O::frag(sb + a_row + (i0 + u) * 16 * ROWB, fl, fg);
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
                for (int j = 0; j < TN; j++) mma16(acc[i0 + u][j], wf[j], af[u]);   // D rows = n (4 fg + e), column = m (fl)
            if (i0 == 0 && kt + NSLOT - 1 < nk) {   // the LDS-DMA of the stage that reuses the freed slot: behind the first MFMA group (its issue cost
                __builtin_amdgcn_sched_barrier(0);   // — ~100 cycles per instruction, in order — then falls under the MFMAs' execution)
                stage((kt + NSLOT - 1) % NSLOT, kt + NSLOT - 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    __syncthreads();   // lnstat (written during step 0 by the first 128 threads) visible to everyone

    // ---- epilogue, per lane: row m = fl of row tile i, columns n = 4 fg .. + 3 of column tile j --------------------------------------------
    const int mw0 = m0 + wm * 64, nw0 = n0 + wn * (BN / 2);
    f32x4 pb[TN], ps4[TN];
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int n = nw0 + j * 16 + 4 * fg;
        pb[j] = f32x4{0, 0, 0, 0};
        ps4[j] = f32x4{0, 0, 0, 0};
        if (n < a.N) {
            if (a.bias) pb[j] = *reinterpret_cast<const f32x4*>(a.bias + n);
            if (a.ln_part) ps4[j] = *reinterpret_cast<const f32x4*>(a.ln_s + n);
        }
    }
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int rloc = wm * 64 + i * 16 + fl, m = mw0 + i * 16 + fl;
        const float mean = a.ln_part ? lnstat[2 * rloc] : 0.0f, rstd = a.ln_part ? lnstat[2 * rloc + 1] : 1.0f;
        const float sh = (a.row_shift && m < a.M) ? a.row_shift[m] : 0.0f;   // producers: the row's running offset
        f32x4 rr[TN];
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int n = nw0 + j * 16 + 4 * fg;
            rr[j] = f32x4{0, 0, 0, 0};
            if (a.R && m < a.M && n < a.N) rr[j] = *reinterpret_cast<const f32x4*>(a.R + (long)m * a.ldr + n);
        }
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int n = nw0 + j * 16 + 4 * fg;
            const bool ok = m < a.M && n < a.N;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                v[e] = a.ln_part ? wh_ln_fold(acc[i][j][e], mean, rstd, ps4[j][e], pb[j][e]) : wh_add(acc[i][j][e], pb[j][e]);
                if (a.act == 1) v[e] = gelu_erf(v[e]);
                v[e] += rr[j][e];
            }
            if (ok) {
                if (a.c_mpad) store4((T*)a.C + slab_idx(m, n, a.c_mpad), v[0], v[1], v[2], v[3]);
                else store4((TO*)a.C + (long)m * a.ldc + n, v[0], v[1], v[2], v[3]);
                if (a.xslab_out) store4((T*)a.xslab_out + slab_idx(m, n, a.x_mpad), v[0] - sh, v[1] - sh, v[2] - sh, v[3] - sh);
            }
            if (a.stats_out) {   // this 16-column tile's {sum x, sum x^2} of the centred row m: 4 values per lane, then the row's four lane groups
                float s1 = 0.0f, s2 = 0.0f;
                if (ok) {
#pragma unroll
                    for (int e = 0; e < 4; e++) { const float u = v[e] - sh; s1 += u; s2 = __builtin_fmaf(u, u, s2); }
                }
                s1 = xrow_sum(s1);
                s2 = xrow_sum(s2);
                if (fg == 0 && m < a.M && n < a.N) {
                    float* sp = a.stats_out + ((long)((nw0 + j * 16) >> 4) * a.x_mpad + m) * 2;
                    sp[0] = s1;
                    sp[1] = s2;
                }
            }
        }
    }
    if (a.ticket) {   // the last workgroup of the last kernel of a prompt step advances the device-side position
        __syncthreads();
        if (tid == 0) {
            const int t = atomicAdd(a.ticket, 1);
            if (t == (int)(gridDim.x * gridDim.z) - 1) {
                *a.ticket = 0;
                *a.pos_w += 1;
            }
        }
    }
}

template <typename T, typename TO, int BN, int NSLOT>
void launch_tile(hipStream_t s, const SkinnyArgs& a) {
    constexpr int SLOT = (BM + BN) * OT<T>::ROWB;
    const size_t sm = (size_t)NSLOT * SLOT + (size_t)BM * 2 * 4 * 3;   // ring + {mean, rstd} + two half sums per row
    dim3 grid(((a.N + BN - 1) / BN) * ((a.M + BM - 1) / BM), 1, a.zn);
    wh_ensure_dyn_lds((const void*)k_dec_tile<T, TO, BN, NSLOT>, sm);
    hipLaunchKernelGGL((k_dec_tile<T, TO, BN, NSLOT>), grid, dim3(256), sm, s, a);
}

template <typename T, typename TO>
void launch_by_shape(hipStream_t s, const SkinnyArgs& a) {
    // Column-tile width.  A 2048-row call has 16 row tiles, so its grid is 16 x N / BN workgroups of four waves: fp16 limb pairs (measured at 2048 clips,
    // tools/runs/gpu_r04y.sh: decode GEMM group 197.3 ms with the bf16 rule below, 189.1 with 64 columns everywhere, 172.5 with 32 columns for
    // N <= 512 and 64 beyond) take the narrowest tile that still fills the chip — N = d_model: 256 workgroups instead of 128 on 256 CUs;
    // bf16 (64-byte LDS rows: a 32-column tile would be half an LDS-DMA instruction per wave): 128 columns while they give ~128 workgroups, else 64.
    static const int force = getenv("WH_DEC_TILE_BN") ? atoi(getenv("WH_DEC_TILE_BN")) : 0;   // (A/B runs: 64 / 128 everywhere)
    constexpr int NS = sizeof(typename FragT<T>::type) <= 16 ? 4 : 3;   // ring slots: 16 / 12 KiB per slot in bf16, 32 / 24 KiB with fp16 limbs
    const long wg128 = (long)((a.N + 127) / 128) * ((a.M + BM - 1) / BM) * a.zn;
    if (force == 64) { launch_tile<T, TO, 64, NS>(s, a); return; }
    if (force == 128 && a.N >= 128) { launch_tile<T, TO, 128, NS>(s, a); return; }
    if constexpr (sizeof(typename FragT<T>::type) > 16) {
        if ((long)((a.N + 63) / 64) * ((a.M + BM - 1) / BM) * a.zn < 256) launch_tile<T, TO, 32, NS>(s, a);
        else launch_tile<T, TO, 64, NS>(s, a);
    } else {
        if (wg128 >= 128 && a.N >= 128) launch_tile<T, TO, 128, NS>(s, a);
        else launch_tile<T, TO, 64, NS>(s, a);
    }
}

}  // namespace

// X as a slab (no merged attention partials), native weights (no fp8 codes / channel scales / activation-side gamma), K a multiple of 32,
// N a multiple of 16 (whole 16-column LayerNorm-partial tiles), at least one k-step per ring slot
bool wh_dec_tile_applicable(int prec, const SkinnyArgs& a) {
    return (prec == WH_PREC_BF16 || prec == WH_PREC_F16X3) && a.X != nullptr && a.xpart == nullptr && a.wscale == nullptr && a.xgamma == nullptr &&
           (a.K % BK) == 0 && a.K >= 4 * BK && (a.N % 16) == 0 && a.logits == nullptr && a.part_val == nullptr;
}

void wh_launch_dec_tile(hipStream_t s, int prec, bool out_f32, const SkinnyArgs& a) {
    if (prec == WH_PREC_F16X3) launch_by_shape<h2, float>(s, a);          // row-major results f32, slabs h2
    else if (out_f32) launch_by_shape<bf16, float>(s, a);
    else launch_by_shape<bf16, bf16>(s, a);
}
