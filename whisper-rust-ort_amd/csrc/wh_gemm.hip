// wh_gemm.hip — LDS-tiled MFMA GEMM with fused epilogues, and row LayerNorm, for gfx950.
//
//   C[m][n] = act( sum_k A[m][k] * W[n][k] + bias ) + R[m][n]
//
// K must be a multiple of 64 (bf16) / 32 (f32).
// Both operands are k-contiguous (activations row-major, weights in torch Linear [out][in]
// layout), which is exactly the 8-consecutive-k-per-lane MFMA fragment, so tiles go global → LDS →
// registers with 16-byte accesses and no transposes.  The MFMA is issued with the WEIGHT tile as
// the row operand, so every lane ends up with 4 consecutive n of one output row: bias/residual
// loads and the output store are 8/16-byte vector accesses.
//
// Row m of A lives at A + (m / m_per) * a_bs + (m % m_per) * lda — this lets the same kernel run
//   - plain Linear layers over all clips of a batch (m_per = rows per clip),
//   - Conv1d(k3,p1) and Conv1d(k3,s2,p1) as GEMMs over OVERLAPPING rows of a zero-padded
//     token-major activation buffer (lda = C_in resp. 2*C_in, K = 3*C_in): no im2col buffer
//     (encoder ops K2/K3 of SURVEY §2d; [3P] modeling_whisper.py:566-567,618-624),
//   - per-clip batched products via blockIdx.z strides (V^T projection).
// It stands in for ONNX Runtime's MLAS GEMM/Conv nodes behind run_encoder
// (reference src/main.rs:698-707) and the step-0 cross-attention K/V projection (:771-787).
#include <stdlib.h>

#include "wh_common.h"
#include "wh_kernels.h"

namespace {

template <typename T> struct Tile;  // K-step depth and LDS row stride (elements) per compute dtype
template <> struct Tile<bf16> { static constexpr int BK = 64, LDK = 72; };   // 144 B rows
template <> struct Tile<float> { static constexpr int BK = 32, LDK = 36; };  // 144 B rows
template <> struct Tile<xf32> { static constexpr int BK = 32, LDK = 36; };   // WH_PREC_F16X3: f32 tiles, split into fp16 limbs at the fragment load
template <> struct Tile<h2> { static constexpr int BK = 32, LDK = 36; };     // ... or already split: one 128-byte block [hi | lo] per row and k-step

template <typename T, typename TO, int BM, int BN>
__global__ __launch_bounds__(256) void k_gemm(GemmArgs g) {
    constexpr int BK = Tile<T>::BK, LDK = Tile<T>::LDK;
    constexpr int CPR = BK * (int)sizeof(T) / 16;        // 16-B chunks per tile row (8)
    constexpr int EPC = 16 / (int)sizeof(T);             // elements per chunk
    constexpr int A_CH = BM * CPR / 256, W_CH = BN * CPR / 256;
    constexpr int TM = BM / 32, TN = BN / 32;            // 16x16 tiles per wave (2x2 waves)
    // two LDS buffers per operand: slab k+1 is written while slab k is being consumed, one barrier per slab
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* As = reinterpret_cast<T*>(smem_raw);              // [2][BM][LDK]
    T* Ws = As + 2 * BM * LDK;                           // [2][BN][LDK]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int fl = lane & 15, fg = lane >> 4;
    // XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one, each XCD has its
    // own L2): linear id b runs tile (b % 8) * chunk + b / 8, so one XCD walks a CONTIGUOUS run of tiles, n fastest —
    // the column tiles that share an activation row panel follow each other on one L2 and the panel leaves HBM once
    // instead of once per XCD (bijective for any tile count; placement only affects speed, never results).
    const int nbn = (g.N + BN - 1) / BN;
    const int total = nbn * ((g.M + BM - 1) / BM);
    int tile = blockIdx.x;
    {
        const int q = total >> 3, r = total & 7, xcd = tile & 7, idx = tile >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (tile / nbn) * BM, n0 = (tile % nbn) * BN;
    const long z = blockIdx.z;
    const T* A = (const T*)g.A + z * g.a_zs;
    const T* W = (const T*)g.W + z * g.w_zs;

    // per-thread source pointers for the staging chunks
    const T* a_src[A_CH];
    const T* w_src[W_CH];
    int a_dst[A_CH], w_dst[W_CH];
#pragma unroll
    for (int i = 0; i < A_CH; i++) {
        int c = tid + i * 256, row = c / CPR, col = (c % CPR) * EPC;
        int m = m0 + row;
        if (m > g.M - 1) m = g.M - 1;
        a_src[i] = A + (long)(m / g.m_per) * g.a_bs + (long)(m % g.m_per) * g.lda + col;
        a_dst[i] = row * LDK + col;
    }
#pragma unroll
    for (int i = 0; i < W_CH; i++) {
        int c = tid + i * 256, row = c / CPR, col = (c % CPR) * EPC;
        int n = n0 + row;
        if (n > g.N - 1) n = g.N - 1;
        w_src[i] = W + (long)n * g.ldw + col;
        w_dst[i] = row * LDK + col;
    }

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = f32x4{0, 0, 0, 0};

    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    u32x4 a_reg[A_CH], w_reg[W_CH];
#pragma unroll
    for (int i = 0; i < A_CH; i++) a_reg[i] = *reinterpret_cast<const u32x4*>(a_src[i]);
#pragma unroll
    for (int i = 0; i < W_CH; i++) w_reg[i] = *reinterpret_cast<const u32x4*>(w_src[i]);
#pragma unroll
    for (int i = 0; i < A_CH; i++) *reinterpret_cast<u32x4*>(&As[a_dst[i]]) = a_reg[i];
#pragma unroll
    for (int i = 0; i < W_CH; i++) *reinterpret_cast<u32x4*>(&Ws[w_dst[i]]) = w_reg[i];
    __syncthreads();

    // per-column epilogue operands (bias, fp8 channel scale) fetched under the main loop
    f32x4 pre_b[TN], pre_w[TN];
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int n = n0 + wc * (BN / 2) + j * 16 + fg * 4;
        pre_b[j] = f32x4{0, 0, 0, 0};
        pre_w[j] = f32x4{1, 1, 1, 1};
        if (n < g.N && g.bias_mode == 1) {
            if (g.bias) pre_b[j] = *reinterpret_cast<const f32x4*>(g.bias + n);
            if (g.wscale) pre_w[j] = *reinterpret_cast<const f32x4*>(g.wscale + n);
        }
    }

    const int nk = g.K / BK;
    for (int kt = 0; kt < nk; kt++) {
        const T* Ac = As + (kt & 1) * BM * LDK;
        const T* Wc = Ws + (kt & 1) * BN * LDK;
        const bool more = kt + 1 < nk;
        if (more) {  // next slab's global loads fly under this slab's MFMAs
#pragma unroll
            for (int i = 0; i < A_CH; i++) a_reg[i] = *reinterpret_cast<const u32x4*>(a_src[i] + (long)(kt + 1) * BK);
#pragma unroll
            for (int i = 0; i < W_CH; i++) w_reg[i] = *reinterpret_cast<const u32x4*>(w_src[i] + (long)(kt + 1) * BK);
        }
#pragma unroll
        for (int ks = 0; ks < BK / 32; ks++) {
            typename FragT<T>::type af[TM], wf[TN];
#pragma unroll
            for (int i = 0; i < TM; i++) af[i] = slab_frag<T>(&Ac[(wr * (BM / 2) + i * 16 + fl) * LDK + ks * 32], fg);
#pragma unroll
            for (int j = 0; j < TN; j++) wf[j] = slab_frag<T>(&Wc[(wc * (BN / 2) + j * 16 + fl) * LDK + ks * 32], fg);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) mma16(acc[i][j], wf[j], af[i]);  // D rows = n, cols = m
        }
        if (more) {
            T* An = As + ((kt + 1) & 1) * BM * LDK;
            T* Wn = Ws + ((kt + 1) & 1) * BN * LDK;
#pragma unroll
            for (int i = 0; i < A_CH; i++) *reinterpret_cast<u32x4*>(&An[a_dst[i]]) = a_reg[i];
#pragma unroll
            for (int i = 0; i < W_CH; i++) *reinterpret_cast<u32x4*>(&Wn[w_dst[i]]) = w_reg[i];
        }
        __syncthreads();
    }

    // epilogue, two passes through LDS (the operand buffers are free after the last barrier):
    //  1. every lane applies channel scale + bias (+ GELU) to its 4-column groups and parks them in an f32 image
    //     of the BM x BN tile;
    //  2. the tile leaves row-contiguously — 32 lanes per 128-column row, 16 B (f32) / 8 B (bf16) per lane — so the f32
    //     residual is read and the output written in whole 128-byte lines.  Per-lane stores straight from the MFMA
    //     layout touch 16 rows x 64 B per instruction; with the f32 residual stream those GEMMs are HBM-bound and ran
    //     at 2.8 TB/s.
    constexpr int LDT = BN + 4;  // f32 tile row pitch: +16 B keeps both passes bank-conflict-free
    float* Ts = reinterpret_cast<float*>(smem_raw);
    static_assert((size_t)BM * LDT * 4 <= (size_t)2 * (BM + BN) * LDK * sizeof(T), "output tile must fit the operand buffers");
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int lr = wr * (BM / 2) + i * 16 + fl;
        const int m = m0 + lr;
        const float bm = (g.bias && g.bias_mode == 2 && m < g.M) ? g.bias[m] : 0.0f;
        const float wm = (g.wscale && g.bias_mode == 2 && m < g.M) ? g.wscale[m] : 1.0f;
#pragma unroll
        for (int j = 0; j < TN; j++) {
            float v[4];
            // fp8 weights: code-valued operand, the channel scale (1 otherwise) is applied here in f32; then the bias
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = acc[i][j][e] * (pre_w[j][e] * wm) + (pre_b[j][e] + bm);
            if (g.act == 1) {
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = gelu_erf(v[e]);
            }
            *reinterpret_cast<f32x4*>(&Ts[lr * LDT + wc * (BN / 2) + j * 16 + fg * 4]) = f32x4{v[0], v[1], v[2], v[3]};
        }
    }
    __syncthreads();
    TO* C = (TO*)g.C + z * g.c_zs;
    const float* R = g.R ? g.R + z * g.r_zs : nullptr;
    // the tile's columns stay inside one n_per block (n_per is a multiple of BN or >= N: checked at launch)
    const long nc0 = (long)(n0 / g.n_per) * g.c_ns + (n0 % g.n_per);
    constexpr int CPRO = BN / 4;                 // 4-column chunks per tile row
    constexpr int RPP = 256 / CPRO;              // tile rows per pass
    const int c4 = (tid % CPRO) * 4, r_in = tid / CPRO;
#pragma unroll 4
    for (int r0 = 0; r0 < BM; r0 += RPP) {
        const int lr = r0 + r_in, m = m0 + lr, n = n0 + c4;
        if (m >= g.M || n >= g.N) continue;
        const long mb = m / g.m_per, mi = m % g.m_per;
        f32x4 v = *reinterpret_cast<const f32x4*>(&Ts[lr * LDT + c4]);
        if (R) {
            const f32x4 r = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(R + mb * g.r_bs + mi * g.ldr + n));  // read once
            v[0] += r[0]; v[1] += r[1]; v[2] += r[2]; v[3] += r[3];
        }
        store4(C + mb * g.c_bs + mi * g.ldc + nc0 + c4, v[0], v[1], v[2], v[3]);
    }
}

// ---- LayerNorm over rows of length d (f32 in, T out); one wave per row -------------------------
// output row of input row `row`: contiguous, or — in_blk > 0 — blocks of in_blk rows (a clip's states) placed out_blk rows apart
__device__ __forceinline__ long ln_out_row(long row, int in_blk, int out_blk) {
    return in_blk > 0 ? (row / in_blk) * out_blk + row % in_blk : row;
}

// [3P] torch LayerNorm eps 1e-5, biased variance (modeling_whisper.py:371,377,642,790).
template <typename TO>
__global__ __launch_bounds__(256) void k_layernorm(const float* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ b, TO* __restrict__ y, long rows, int d, int in_blk, int out_blk) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + row * d;
    constexpr int MAXV = 5;  // d <= 1280: 64 lanes * 4 floats * 5
    f32x4 v[MAXV];
    float s = 0.0f;
    const int nv = d >> 8;  // d / 256 full sweeps
#pragma unroll
    for (int i = 0; i < MAXV; i++) {
        if (i < nv) {
            v[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr + (i * 64 + lane) * 4));  // read once
            s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        }
    }
    // tail (d % 256, multiple of 4): used by d = 128 (nano) and 1280 = 5*256 has none
    const int rem = d & 255;
    f32x4 tv = {0, 0, 0, 0};
    const bool has_tail = lane * 4 < rem;
    if (has_tail) {
        tv = *reinterpret_cast<const f32x4*>(xr + nv * 256 + lane * 4);
        s += tv[0] + tv[1] + tv[2] + tv[3];
    }
    const float mean = dpp_wave_sum(s) / (float)d;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXV; i++) {
        if (i < nv) {
#pragma unroll
            for (int e = 0; e < 4; e++) { float t = v[i][e] - mean; q += t * t; }
        }
    }
    if (has_tail) {
#pragma unroll
        for (int e = 0; e < 4; e++) { float t = tv[e] - mean; q += t * t; }
    }
    const float rstd = rsqrtf(dpp_wave_sum(q) / (float)d + 1e-5f);
    TO* yr = y + ln_out_row(row, in_blk, out_blk) * d;
#pragma unroll
    for (int i = 0; i < MAXV; i++) {
        if (i < nv) {
            const int c = (i * 64 + lane) * 4;
            f32x4 ww = *reinterpret_cast<const f32x4*>(w + c), bb = *reinterpret_cast<const f32x4*>(b + c);
            store4(yr + c, (v[i][0] - mean) * rstd * ww[0] + bb[0], (v[i][1] - mean) * rstd * ww[1] + bb[1],
                   (v[i][2] - mean) * rstd * ww[2] + bb[2], (v[i][3] - mean) * rstd * ww[3] + bb[3]);
        }
    }
    if (has_tail) {
        const int c = nv * 256 + lane * 4;
        f32x4 ww = *reinterpret_cast<const f32x4*>(w + c), bb = *reinterpret_cast<const f32x4*>(b + c);
        store4(yr + c, (tv[0] - mean) * rstd * ww[0] + bb[0], (tv[1] - mean) * rstd * ww[1] + bb[1],
               (tv[2] - mean) * rstd * ww[2] + bb[2], (tv[3] - mean) * rstd * ww[3] + bb[3]);
    }
}

// d % 512 == 0, 2-byte output: a lane owns 8 consecutive columns per 512-column sweep (two 16-byte loads, one 16-byte
// store — 8-byte bf16 stores ran at 3 TB/s against 5 for 16-byte ones in the GEMM epilogue) and a wave carries two rows at
// once, so four loads per lane are in flight instead of two.
template <int SWEEPS>
__global__ __launch_bounds__(256) void k_layernorm_w8(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ b, bf16* __restrict__ y, long rows, int d, int in_blk, int out_blk) {
    const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
    if (row0 >= rows) return;
    const int lane = threadIdx.x & 63;
    const bool two = row0 + 1 < rows;
    const float* xr[2] = {x + row0 * d, x + (two ? row0 + 1 : row0) * d};
    f32x4 v[2][SWEEPS][2];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int i = 0; i < SWEEPS; i++)
#pragma unroll
            for (int h = 0; h < 2; h++) v[r][i][h] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr[r] + i * 512 + lane * 8 + h * 4));
    float mean[2], rstd[2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < SWEEPS; i++)
#pragma unroll
            for (int h = 0; h < 2; h++) s += (v[r][i][h][0] + v[r][i][h][1]) + (v[r][i][h][2] + v[r][i][h][3]);
        mean[r] = dpp_wave_sum(s) / (float)d;
        float q = 0.0f;
#pragma unroll
        for (int i = 0; i < SWEEPS; i++)
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int e = 0; e < 4; e++) { const float t = v[r][i][h][e] - mean[r]; q += t * t; }
        rstd[r] = rsqrtf(dpp_wave_sum(q) / (float)d + 1e-5f);
    }
#pragma unroll
    for (int i = 0; i < SWEEPS; i++) {
        const int c = i * 512 + lane * 8;
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(w + c), w1 = *reinterpret_cast<const f32x4*>(w + c + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(b + c), b1 = *reinterpret_cast<const f32x4*>(b + c + 4);
#pragma unroll
        for (int r = 0; r < 2; r++) {
            if (r == 1 && !two) break;
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                o[e] = (bf16)((v[r][i][0][e] - mean[r]) * rstd[r] * w0[e] + b0[e]);
                o[4 + e] = (bf16)((v[r][i][1][e] - mean[r]) * rstd[r] * w1[e] + b1[e]);
            }
            *reinterpret_cast<bf16x8*>(y + ln_out_row(row0 + r, in_blk, out_blk) * d + c) = o;
        }
    }
}

template <typename T, typename TO>
void launch_gemm_t(hipStream_t s, const GemmArgs& g) {
    const long blocks128 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128) * g.batch;
    constexpr int LDK = Tile<T>::LDK;
    if (blocks128 >= 192) {
        dim3 grid(((g.N + 127) / 128) * ((g.M + 127) / 128), 1, g.batch);
        const size_t sm = (size_t)2 * (128 + 128) * LDK * sizeof(T);
        wh_ensure_dyn_lds((const void*)k_gemm<T, TO, 128, 128>, sm);
        hipLaunchKernelGGL((k_gemm<T, TO, 128, 128>), grid, dim3(256), sm, s, g);
    } else {
        dim3 grid(((g.N + 63) / 64) * ((g.M + 63) / 64), 1, g.batch);
        const size_t sm = (size_t)2 * (64 + 64) * LDK * sizeof(T);
        hipLaunchKernelGGL((k_gemm<T, TO, 64, 64>), grid, dim3(256), sm, s, g);
    }
}

}  // namespace

bool wh_gemm8_enabled() {
    static const bool use8 = getenv("WH_GEMM8") == nullptr || atoi(getenv("WH_GEMM8")) != 0;   // WH_GEMM8=0: A/B against k_gemm
    return use8;
}

int wh_launch_gemm(hipStream_t s, int prec, bool out_f32, const GemmArgs& g) {
    if (prec != WH_PREC_F32 && prec != WH_PREC_F16X3 && wh_gemm8_enabled() && !g.small_ctx && wh_gemm8_applicable(g)) return wh_launch_gemm8(s, out_f32, g);
    if (prec == WH_PREC_F16X3 && !g.f32_operands && wh_gemm8_enabled() && !g.small_ctx && wh_gemm8x_applicable(g, !out_f32)) return wh_launch_gemm8x(s, !out_f32, g);
    if (g.ln_mode || g.xb_out || g.stats_out) {   // only k_gemm8 implements the LayerNorm fold: never drop it silently
        wh_set_error("GEMM with a folded LayerNorm (M %d N %d K %d) must run on k_gemm8", g.M, g.N, g.K);
        return WH_ERR_UNSUPPORTED;
    }
    const int bk = (prec == WH_PREC_F32 || prec == WH_PREC_F16X3) ? Tile<float>::BK : Tile<bf16>::BK;
    if ((g.K % bk) != 0 || (g.n_per < g.N && (g.n_per % 128) != 0)) {
        wh_set_error("k_gemm: K %d must be a multiple of %d and a column plane (%d) a multiple of the 128-column tile", g.K, bk, g.n_per);
        return WH_ERR_UNSUPPORTED;
    }
    if (prec == WH_PREC_F32) {
        launch_gemm_t<float, float>(s, g);
    } else if (prec == WH_PREC_F16X3) {   // results: f32 rows (residual stream, f32 consumers) or h2 (the next matrix-core consumer's operand)
        if (g.f32_operands) { if (out_f32) launch_gemm_t<xf32, float>(s, g); else launch_gemm_t<xf32, h2>(s, g); }
        else if (out_f32) launch_gemm_t<h2, float>(s, g);
        else launch_gemm_t<h2, h2>(s, g);
    } else {
        if (out_f32) launch_gemm_t<bf16, float>(s, g);
        else launch_gemm_t<bf16, bf16>(s, g);
    }
    return WH_OK;
}

void wh_launch_layernorm_blocks(hipStream_t s, int prec, const float* x, const float* w, const float* b, void* y, long rows, int d, int in_blk,
                                int out_blk) {
    dim3 grid((unsigned)((rows + 3) / 4));
    const bool f32_layout = prec == WH_PREC_F32 || prec == WH_PREC_F16X3;
    if (!f32_layout && (d == 512 || d == 1024) && getenv("WH_LN_W8_OFF") == nullptr) {
        dim3 g8((unsigned)((rows + 7) / 8));
        if (d == 512) hipLaunchKernelGGL(k_layernorm_w8<1>, g8, dim3(256), 0, s, x, w, b, (bf16*)y, rows, d, in_blk, out_blk);
        else hipLaunchKernelGGL(k_layernorm_w8<2>, g8, dim3(256), 0, s, x, w, b, (bf16*)y, rows, d, in_blk, out_blk);
        return;
    }
    if (prec == WH_PREC_F16X3) hipLaunchKernelGGL(k_layernorm<h2>, grid, dim3(256), 0, s, x, w, b, (h2*)y, rows, d, in_blk, out_blk);   // the consumers' fp16-limb operand
    else if (f32_layout) hipLaunchKernelGGL(k_layernorm<float>, grid, dim3(256), 0, s, x, w, b, (float*)y, rows, d, in_blk, out_blk);
    else hipLaunchKernelGGL(k_layernorm<bf16>, grid, dim3(256), 0, s, x, w, b, (bf16*)y, rows, d, in_blk, out_blk);
}

void wh_launch_layernorm(hipStream_t s, int prec, const float* x, const float* w, const float* b, void* y, long rows, int d) {
    wh_launch_layernorm_blocks(s, prec, x, w, b, y, rows, d, 0, 0);
}
