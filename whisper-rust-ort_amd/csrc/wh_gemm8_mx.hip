// wh_gemm8_mx.hip — WH_PREC_FP8 encoder GEMMs on the fp8 matrix cores (BASELINE configs[4]: "CDNA4 fp8 MFMA").
//
//   C[m][n] = act( wscale[n] * sum_k A8[m][k] 2^(ea[m][k/32]) * W8[n][k] 2^(ew[n][k/32]) + bias ) + R[m][n]
//
// A8 / W8 are OCP e4m3 codes, ea / ew E8M0 block exponents (one per 32 consecutive k: the MX layout).  Weights carry
// their f32 per-output-channel scale (wscale, applied in the epilogue) and no block exponents; ACTIVATIONS are quantised
// by their producers (k_layernorm_mx, this kernel's own MX epilogue for the GELU output) with a power-of-two scale per
// (row, 32-block) chosen so that the block's largest magnitude lands in (224, 448] — the hardware applies it for free:
// v_mfma_scale_f32_16x16x128_f8f6f4 multiplies each lane's 32 products by 2^(ea - 127) 2^(ew - 127).  That instruction
// contracts K = 128 per issue at twice the bf16 MFMA rate and halves the operand bytes through L2 -> LDS -> registers.
// Reference analogue of the quantised leg: quantize_onnx_int8.py:37-42 (dynamic quantisation of MatMul/Gemm: int8
// weights AND per-call quantised activations); restated for the oracle in oracle/whisper_oracle.c (fake_quant_mx).
//
// Operand layout of the instruction (measured: tools/mx_scale_probe.hip, tools/mx_mfma_check.hip): lane l, row r = l & 15,
// group g = l >> 4 holds k = 16 g .. 16 g + 15 in VGPRs 0-3 and k = 64 + 16 g .. + 15 in VGPRs 4-7; its scale byte applies
// to (row r, block g).  So a lane reads the 16-byte chunks g and 4 + g of its row's 128-byte k-step from LDS.
//
// Kernel structure = k_gemm8's 128-wide geometry with K-steps of 128 bytes: 8 waves (4 x 2), 256 x 128 tile, slot =
// A [256][128 B] + W [128][128 B] = 48 KiB, three slots, LDS-DMA with the bank swizzle on the source side
// (chunk p of row r holds k-chunk p ^ ((r >> 1) & 7): conflict-free for the chunk pairs (g, 4 + g)), counted vmcnt, one
// barrier per K-step, wave-private LDS staging of the output, 16-byte row-contiguous stores.  Block exponents are laid
// out [row][4][nkp] (nkp = K/128 rounded up to a multiple of 4 from 4 on: wh_mx_nkp) so that a lane's bytes for consecutive
// K-steps are adjacent dwords: the exponents of 16 K-steps (a segment) sit in four registers per tile row, loaded before the
// ring starts.  Contractions longer than 16 K-steps (whisper-large-v3's ffn 5120: 40) reload them at each segment's end,
// behind that step's MFMAs, and the next step waits for vmcnt(0) once — the only ordinary loads that share the loop with the
// LDS-DMA, one partial drain per 16 K-steps.
#include <type_traits>

#include "wh_common.h"
#include "wh_kernels.h"

namespace {

constexpr int BM = 256, BN = 128, BKB = 128;
constexpr int ROWB = BKB;
constexpr int SLOT_A = BM * ROWB, SLOT = SLOT_A + BN * ROWB;   // 32 + 16 KiB
constexpr int NSLOT = 3, PER_STAGE = 6;
constexpr int TM = 4, TN = 4;
constexpr int EP_PITCH = 68, EP_ROWS = 32;
constexpr int MAX_NK = 16;                                     // K-steps per exponent segment (four dwords per tile row)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;

__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// the exponents of one row, lane group fg, K-steps 16 seg .. 16 seg + 15 as up to four dwords (row pitch 4 * nkp bytes)
__device__ __forceinline__ void load_exps(const unsigned char* base, long row, int fg, int nkp, int seg, unsigned (&dw)[MAX_NK / 4]) {
    const unsigned char* p = base + (row * 4 + fg) * nkp + seg * MAX_NK;
#pragma unroll
    for (int i = 0; i < MAX_NK / 4; i++) dw[i] = 0x7F7F7F7Fu;
    if (nkp == 2) dw[0] = *reinterpret_cast<const unsigned short*>(p) | 0x7F7F0000u;
    else {
#pragma unroll
        for (int i = 0; i < MAX_NK / 4; i++)
            if (seg * MAX_NK + 4 * i < nkp) dw[i] = *reinterpret_cast<const unsigned*>(p + 4 * i);
    }
}

// max |.| over the 4 lanes of a quad (32 consecutive columns at 8 per lane)
__device__ __forceinline__ float quad_max(float v) {
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
    return v;
}

// ---- epilogue (k_gemm8's): 32 rows per wave and pass through the wave's own 8.5 KiB of the idle ring; shared by both tile geometries
struct MxOut {};   // tag: C = e4m3 codes + block exponents
template <typename TO, int TM, int TN>
__device__ __forceinline__ void mx_epilogue(const GemmArgs& g, f32x4 (&acc)[TM][TN], char* smem, int wave, int lane, int mw0, int nw0, long z) {
    const int fl = lane & 15, fg = lane >> 4;
    float* stg = reinterpret_cast<float*>(smem) + wave * (EP_ROWS * EP_PITCH);
    constexpr bool MX = __is_same(TO, MxOut);
    typedef typename std::conditional<MX, unsigned char, TO>::type TC;
    TC* C = (TC*)g.C + z * g.c_zs;
    const float* R = g.R ? g.R + z * g.r_zs : nullptr;
    const long nc0 = (long)(nw0 / g.n_per) * g.c_ns + (nw0 % g.n_per);
    f32x4 pb[TN], pw[TN];
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int n = nw0 + j * 16 + fg * 4;
        pb[j] = f32x4{0, 0, 0, 0};
        pw[j] = f32x4{1, 1, 1, 1};
        if (n < g.N && g.bias_mode == 1) {
            if (g.bias) pb[j] = *reinterpret_cast<const f32x4*>(g.bias + n);
            if (g.wscale) pw[j] = *reinterpret_cast<const f32x4*>(g.wscale + n);
        }
    }
    // per-row bias / channel scale (the per-clip V^T product) of all of this lane's rows, requested together
    float rowb[TM], roww[TM];
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int m = min(mw0 + i * 16 + fl, g.M - 1);
        rowb[i] = 0.0f;
        roww[i] = 1.0f;
        if (g.bias_mode == 2) {
            if (g.bias) rowb[i] = g.bias[m];
            if (g.wscale) roww[i] = g.wscale[m];
        }
    }
    const int c8 = (lane & 7) * 8, r8 = lane >> 3;
    const int n_st = nw0 + c8;
#pragma unroll
    for (int pass = 0; pass < TM / 2; pass++) {
#pragma unroll
        for (int ii = 0; ii < 2; ii++) {
            const int i = pass * 2 + ii;
            const float bm = rowb[i], wmul = roww[i];
#pragma unroll
            for (int j = 0; j < TN; j++) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = acc[i][j][e] * (pw[j][e] * wmul) + (pb[j][e] + bm);
                if (g.act == 1) {
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = gelu_erf(v[e]);
                }
                *reinterpret_cast<f32x4*>(&stg[(ii * 16 + fl) * EP_PITCH + j * 16 + fg * 4]) = f32x4{v[0], v[1], v[2], v[3]};
            }
        }
        if constexpr (MX) {
            // MX output (the GELU activations that feed fc2): 16 columns per lane -> a lane pair owns one 32-column block and
            // a lane stores 16 bytes of codes; exponent of a block from its largest magnitude (byte = E - 8, + 1 when the
            // mantissa exceeds 1.75: the block maximum then maps into (224, 448]), codes = e4m3(value * 2^-e).
            // N % 128 == 0 and contiguous rows are checked at launch.
            const int c16 = (lane & 3) * 16, r16 = lane >> 2;
#pragma unroll
            for (int it = 0; it < EP_ROWS / 16; it++) {
                const int lr = it * 16 + r16, m = mw0 + pass * EP_ROWS + lr, n = nw0 + c16;
                f32x4 q[4];
#pragma unroll
                for (int u = 0; u < 4; u++) q[u] = *reinterpret_cast<const f32x4*>(&stg[lr * EP_PITCH + c16 + 4 * u]);
                float am = 0.0f;
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int e = 0; e < 4; e++) am = fmaxf(am, fabsf(q[u][e]));
                am = fmaxf(am, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, am), 0xB1, 0xF, 0xF, true)));  // lane ^ 1
                const unsigned ab = __float_as_uint(am);
                const int eb = max(0, (int)((ab >> 23) & 0xFF) - 8 + (int)((ab & 0x7FFFFF) > 0x600000));
                const float inv = __uint_as_float((unsigned)(254 - eb) << 23);
                i32x4 pk;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    int t = __builtin_amdgcn_cvt_pk_fp8_f32(q[u][0] * inv, q[u][1] * inv, 0, false);
                    pk[u] = __builtin_amdgcn_cvt_pk_fp8_f32(q[u][2] * inv, q[u][3] * inv, t, true);
                }
                if (m < g.M && n < g.N) {
                    *reinterpret_cast<i32x4*>(C + (long)m * g.ldc + n) = pk;
                    if ((lane & 1) == 0) {
                        const int blk = n >> 5;
                        g.c_sc[((long)m * 4 + (blk & 3)) * wh_mx_nkp(g.N) + (blk >> 2)] = (unsigned char)eb;
                    }
                }
            }
        } else {
            const int mp0 = mw0 + pass * EP_ROWS + r8;
            long mb = mp0 / g.m_per, mi = mp0 % g.m_per;
            // the pass's residual rows first, all four in flight (k_gemm8's epilogue: one round trip per pass, not per 8-row group)
            f32x4 rr0[EP_ROWS / 8], rr1[EP_ROWS / 8];
            if (R) {
                long qb = mb, qi = mi;
#pragma unroll
                for (int it = 0; it < EP_ROWS / 8; it++) {
                    rr0[it] = f32x4{0, 0, 0, 0};
                    rr1[it] = f32x4{0, 0, 0, 0};
                    if (mp0 + it * 8 < g.M && n_st < g.N) {
                        const float* rp = R + qb * g.r_bs + qi * g.ldr + n_st;
                        rr0[it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(rp));
                        rr1[it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(rp + 4));
                    }
                    qi += 8;
                    if (qi >= g.m_per) { qi -= g.m_per; qb += 1; }
                }
            }
#pragma unroll
            for (int it = 0; it < EP_ROWS / 8; it++) {
                const int lr = it * 8 + r8, m = mp0 + it * 8;
                if (m < g.M && n_st < g.N) {
                    f32x4 v0 = *reinterpret_cast<const f32x4*>(&stg[lr * EP_PITCH + c8]);
                    f32x4 v1 = *reinterpret_cast<const f32x4*>(&stg[lr * EP_PITCH + c8 + 4]);
                    if (R) {
                        v0 += rr0[it];
                        v1 += rr1[it];
                    }
                    TC* cp = C + mb * g.c_bs + mi * g.ldc + nc0 + c8;
                    if (n_st + 8 <= g.N) {
                        if constexpr (sizeof(TC) == 4) {
                            *reinterpret_cast<f32x4*>(cp) = v0;
                            *reinterpret_cast<f32x4*>(cp + 4) = v1;
                        } else {
                            *reinterpret_cast<bf16x8*>(cp) = bf16x8{(bf16)v0[0], (bf16)v0[1], (bf16)v0[2], (bf16)v0[3], (bf16)v1[0], (bf16)v1[1], (bf16)v1[2], (bf16)v1[3]};
                        }
                    } else {
                        store4(cp, v0[0], v0[1], v0[2], v0[3]);   // N % 8 == 4: the last group holds 4 valid columns
                    }
                }
                mi += 8;
                if (mi >= g.m_per) { mi -= g.m_per; mb += 1; }
            }
        }
    }
}

template <typename TO>
__global__ __launch_bounds__(512, 2) void k_gemm8_mx(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fl = lane & 15, fg = lane >> 4;
    const int nk = g.K / BKB, nkp = wh_mx_nkp(g.K);

    const int nbn = (g.N + BN - 1) / BN;
    const int total = nbn * ((g.M + BM - 1) / BM);
    int tile = blockIdx.x;
    {
        const int q = total >> 3, r = total & 7, xcd = tile & 7, idx = tile >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (tile / nbn) * BM, n0 = (tile % nbn) * BN;
    const long z = blockIdx.z;
    const unsigned char* A = (const unsigned char*)g.A + z * g.a_zs;
    const unsigned char* W = (const unsigned char*)g.W + z * g.w_zs;
    const int mw0 = m0 + wm * 64, nw0 = n0 + wn * 64;

    // block exponents of this wave's rows, one segment of 16 K-steps at a time: the first before anything else, complete
    // before the ring starts
    unsigned ea[TM][MAX_NK / 4], ew[TN][MAX_NK / 4];
    auto load_segment = [&](int seg) {
#pragma unroll
        for (int i = 0; i < TM; i++) {
#pragma unroll
            for (int q = 0; q < MAX_NK / 4; q++) ea[i][q] = 0x7F7F7F7Fu;
            if (g.a_sc) load_exps(g.a_sc + z * g.a_sc_zs, min(mw0 + i * 16 + fl, g.M - 1), fg, nkp, seg, ea[i]);
        }
#pragma unroll
        for (int j = 0; j < TN; j++) {
#pragma unroll
            for (int q = 0; q < MAX_NK / 4; q++) ew[j][q] = 0x7F7F7F7Fu;
            if (g.w_sc8) load_exps(g.w_sc8 + z * g.w_sc_zs, min(nw0 + j * 16 + fl, g.N - 1), fg, nkp, seg, ew[j]);
        }
    };
    load_segment(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // per-lane source pointers of this wave's share of a stage: one wave-instruction = 1 KiB = 8 rows x 128 bytes
    const int rl = lane >> 3, ps = lane & 7;
    const unsigned char* a_src[4];
    const unsigned char* w_src[2];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int row = wave * 32 + j * 8 + rl;
        const int m = min(m0 + row, g.M - 1);
        a_src[j] = A + (long)(m / g.m_per) * g.a_bs + (long)(m % g.m_per) * g.lda + ((ps ^ swz(row)) << 4);
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int row = wave * 16 + j * 8 + rl;
        const int n = min(n0 + row, g.N - 1);
        w_src[j] = W + (long)n * g.ldw + ((ps ^ swz(row)) << 4);
    }
    auto stage = [&](int slot, int kt) {
        char* base = smem + slot * SLOT;
#pragma unroll
        for (int j = 0; j < 4; j++) glds16(a_src[j] + (long)kt * BKB, base + (wave * 32 + j * 8) * ROWB);
#pragma unroll
        for (int j = 0; j < 2; j++) glds16(w_src[j] + (long)kt * BKB, base + SLOT_A + (wave * 16 + j * 8) * ROWB);
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = f32x4{0, 0, 0, 0};

    const int c0 = (fg ^ swz(fl)) << 4, c1 = ((4 + fg) ^ swz(fl)) << 4;
    const int a_row = (wm * 64 + fl) * ROWB, w_row = SLOT_A + (wn * 64 + fl) * ROWB;

    stage(0, 0);
    if (nk > 1) stage(1, 1);
    for (int kt = 0; kt < nk; kt++) {
        // (first step of a later segment: the exponent loads issued behind the previous step's MFMAs are the youngest requests)
        if (kt + 1 < nk && ((kt & (MAX_NK - 1)) != 0 || kt == 0)) wait_vm<PER_STAGE>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) stage((kt + 2) % NSLOT, kt + 2);
        const char* sb = smem + (kt % NSLOT) * SLOT;
        i32x8 af[TM], wf[TN];
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const i32x4 lo = *reinterpret_cast<const i32x4*>(sb + a_row + i * 16 * ROWB + c0), hi = *reinterpret_cast<const i32x4*>(sb + a_row + i * 16 * ROWB + c1);
            af[i] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const i32x4 lo = *reinterpret_cast<const i32x4*>(sb + w_row + j * 16 * ROWB + c0), hi = *reinterpret_cast<const i32x4*>(sb + w_row + j * 16 * ROWB + c1);
            wf[j] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
        // this K-step's exponent byte of every tile (byte kt & 3 of dword kt >> 2, moved to byte 0: opsel 0)
        const int sh = 8 * (kt & 3), qd = (kt & (MAX_NK - 1)) >> 2;
        int sa[TM], sw[TN];
#pragma unroll
        for (int i = 0; i < TM; i++) sa[i] = (int)((qd == 0 ? ea[i][0] : qd == 1 ? ea[i][1] : qd == 2 ? ea[i][2] : ea[i][3]) >> sh);
#pragma unroll
        for (int j = 0; j < TN; j++) sw[j] = (int)((qd == 0 ? ew[j][0] : qd == 1 ? ew[j][1] : qd == 2 ? ew[j][2] : ew[j][3]) >> sh);
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)   // D rows = n (first operand), cols = m
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[j], af[i], acc[i][j], 0, 0, 0, sw[j], 0, sa[i]);
        if ((kt & (MAX_NK - 1)) == MAX_NK - 1 && kt + 1 < nk) load_segment((kt + 1) / MAX_NK);   // next segment's exponents
    }
    __builtin_amdgcn_s_barrier();   // every wave is done with the ring: it becomes the output staging area
    mx_epilogue<TO, TM, TN>(g, acc, smem, wave, lane, mw0, nw0, z);
}

// ---- LayerNorm with MX output: e4m3 codes [rows][d] + block exponents [rows][4][d/128]; one wave per row --------------
template <int NV>
__global__ __launch_bounds__(256) void k_layernorm_mx(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                      unsigned char* __restrict__ codes, unsigned char* __restrict__ exps, long rows) {
    constexpr int d = 256 * NV, nk = d >> 7, NB = d >> 5, nkp = nk < 4 ? nk : (nk + 3) / 4 * 4, NE = 4 * nkp;   // NE exponent bytes per row
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + row * d;
    f32x4 v[NV];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        v[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr + (i * 64 + lane) * 4));
        s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
    const float mean = dpp_wave_sum(s) / (float)d;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; i++) {
#pragma unroll
        for (int e = 0; e < 4; e++) { const float t = v[i][e] - mean; q += t * t; }
    }
    const float rstd = rsqrtf(dpp_wave_sum(q) / (float)d + 1e-5f);
    int ebs[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const int c = (i * 64 + lane) * 4;
        const f32x4 ww = *reinterpret_cast<const f32x4*>(w + c), bb = *reinterpret_cast<const f32x4*>(b + c);
        float y[4], am = 0.0f;
#pragma unroll
        for (int e = 0; e < 4; e++) { y[e] = (v[i][e] - mean) * rstd * ww[e] + bb[e]; am = fmaxf(am, fabsf(y[e])); }
        am = quad_max(am);   // 8 lanes x 4 columns = one 32-block: quad, then the neighbouring quad
        am = fmaxf(am, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, am), 0x141, 0xF, 0xF, true)));  // row_half_mirror
        const unsigned ab = __float_as_uint(am);
        const int eb = max(0, (int)((ab >> 23) & 0xFF) - 8 + (int)((ab & 0x7FFFFF) > 0x600000));
        const float inv = __uint_as_float((unsigned)(254 - eb) << 23);
        int pk = __builtin_amdgcn_cvt_pk_fp8_f32(y[0] * inv, y[1] * inv, 0, false);
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(y[2] * inv, y[3] * inv, pk, true);
        *reinterpret_cast<int*>(codes + row * d + c) = pk;
        ebs[i] = eb;
    }
    // exponent byte at layout position j = (blk & 3) * nkp + (blk >> 2)  <=>  blk = (j / nkp) + 4 * (j % nkp), a padding byte
    // where j % nkp >= nk (written as 2^0, never used); block blk was computed in sweep blk >> 3 by the lanes 8 * (blk & 7) .. + 7
    static_assert(NE <= 64, "one lane per exponent byte");
    const int j = lane < NE ? lane : 0, kt = j % nkp, blk = min((j / nkp) + 4 * kt, NB - 1), src = 8 * (blk & 7), sweep = blk >> 3;
    int mine = 127;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const int got = __shfl(ebs[i], src);
        mine = (sweep == i && kt < nk) ? got : mine;
    }
    if (lane < NE) exps[row * NE + lane] = (unsigned char)mine;
}

template <typename TO>
void launch_mx(hipStream_t s, const GemmArgs& g) {
    const size_t sm = (size_t)NSLOT * SLOT;
    static_assert((size_t)8 * EP_ROWS * EP_PITCH * 4 <= (size_t)NSLOT * SLOT, "output staging must fit the ring");
    dim3 grid(((g.N + BN - 1) / BN) * ((g.M + BM - 1) / BM), 1, g.batch);
    wh_ensure_dyn_lds((const void*)k_gemm8_mx<TO>, sm);
    hipLaunchKernelGGL(k_gemm8_mx<TO>, grid, dim3(512), sm, s, g);
}

}  // namespace

bool wh_gemm8_mx_applicable(const GemmArgs& g) {
    const int nk = g.K / BKB;
    return g.M >= BM && g.N >= BN && (g.K % BKB) == 0 && nk >= 2 && (g.N % 4) == 0 &&
           (g.n_per >= g.N || (g.n_per % 64) == 0) && g.m_per >= 8 && (!g.c_sc || ((g.N % 128) == 0 && g.m_per >= g.M && g.n_per >= g.N));
}

// out: 0 = bf16, 1 = f32, 2 = MX (codes + exponents in g.c_sc)
int wh_launch_gemm8_mx(hipStream_t s, int out, const GemmArgs& g) {
    if (!wh_gemm8_mx_applicable(g)) {   // the geometry guarantees live in one place; callers decide from the context (mx_ok)
        wh_set_error("k_gemm8_mx: geometry M %d N %d K %d not covered", g.M, g.N, g.K);
        return WH_ERR_UNSUPPORTED;
    }
    if (out == 2) launch_mx<MxOut>(s, g);
    else if (out == 1) launch_mx<float>(s, g);
    else launch_mx<bf16>(s, g);
    return WH_OK;
}

int wh_launch_layernorm_mx(hipStream_t s, const float* x, const float* w, const float* b, void* codes, void* exps, long rows, int d) {
    dim3 grid((unsigned)((rows + 3) / 4));
    unsigned char *cp = (unsigned char*)codes, *ep = (unsigned char*)exps;
    switch (d) {   // the MX path exists for these widths only (wh_api.cpp: mx_ok)
        case 256: hipLaunchKernelGGL(k_layernorm_mx<1>, grid, dim3(256), 0, s, x, w, b, cp, ep, rows); break;
        case 512: hipLaunchKernelGGL(k_layernorm_mx<2>, grid, dim3(256), 0, s, x, w, b, cp, ep, rows); break;
        case 1024: hipLaunchKernelGGL(k_layernorm_mx<4>, grid, dim3(256), 0, s, x, w, b, cp, ep, rows); break;
        case 1280: hipLaunchKernelGGL(k_layernorm_mx<5>, grid, dim3(256), 0, s, x, w, b, cp, ep, rows); break;
        case 1536: hipLaunchKernelGGL(k_layernorm_mx<6>, grid, dim3(256), 0, s, x, w, b, cp, ep, rows); break;
        case 2048: hipLaunchKernelGGL(k_layernorm_mx<8>, grid, dim3(256), 0, s, x, w, b, cp, ep, rows); break;
        default: wh_set_error("k_layernorm_mx: unsupported width %d", d); return WH_ERR_UNSUPPORTED;
    }
    return WH_OK;
}
