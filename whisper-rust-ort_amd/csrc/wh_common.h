// wh_common.h — shared device helpers for the gfx950 kernels (wave64, MFMA 16x16 tiles).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/whisper_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(4))) double f64x4;

#define WH_WAVE 64
#define WH_HEAD_DIM 64

// ---------------------------------------------------------------------------------------------
// One "fragment" = 8 consecutive k-elements per lane.  Lane l of a wave holds, for the row/column
// index (l & 15), the k-range 8*(l>>4) .. 8*(l>>4)+7 of a 32-deep k-slab.
//   mma16(acc, a, b):  D[i][j] += sum_k A[i][k] * B[k][j]
//     a = A[i = l&15][k-range],  b = B[k-range][j = l&15]
//     D layout (all dtypes): column j = l & 15, rows i = 4*(l>>4) + r for r = 0..3 (acc[r]).
// bf16: one v_mfma_f32_16x16x32_bf16.  f32: eight v_mfma_f32_16x16x4_f32 (exact f32 fma chain);
// the per-instruction k assignment {8g+e : g=0..3} is a permutation of the slab, identical for
// both operands, so the contraction is the same sum.
// ---------------------------------------------------------------------------------------------
template <typename T> struct FragT;
template <> struct FragT<bf16> { typedef bf16x8 type; };
template <> struct FragT<float> { typedef f32x8 type; };

template <typename T>
__device__ __forceinline__ typename FragT<T>::type load_frag(const T* p) {
    return *reinterpret_cast<const typename FragT<T>::type*>(p);
}

__device__ __forceinline__ void mma16(f32x4& acc, const bf16x8& a, const bf16x8& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma16(f32x4& acc, const f32x8& a, const f32x8& b) {
#pragma unroll
    for (int e = 0; e < 8; e++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// WH_PREC_F16X3: f32 storage, contractions on the fp16 matrix cores.  An operand value x is carried as two fp16 limbs,
// hi = f16(x), lo = f16(x - hi) (x - hi is exact in f32; 11 + 11 significant bits), and a product is three MFMAs:
//   sum_k a_k b_k  ~=  sum a.lo b.hi + sum a.hi b.lo + sum a.hi b.hi      (the lo.lo term, 2^-22 relative, is dropped)
// accumulated in f32 — measured against exact f32 on whisper-base: max |d logit| 6e-5, the distance between two f32
// implementations being 3e-5 (tools/x3_numerics.py; with bf16 limbs the same scheme measures 4e-4).
// `xf32` is the element-type tag of such an operand in memory (the bytes of a float); its fragment holds the limbs, so a
// fragment is split once when it is loaded and reused by every MFMA it feeds.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
struct xf32 { float v; };
struct xfrag { f16x8 hi, lo; };
template <> struct FragT<xf32> { typedef xfrag type; };
__device__ __forceinline__ xfrag x3_split(const f32x8& x) {
    xfrag r;
    r.hi = __builtin_convertvector(x, f16x8);                    // 4 x v_cvt_pk_f16_f32 (RNE)
    r.lo = __builtin_convertvector(x - __builtin_convertvector(r.hi, f32x8), f16x8);
    return r;
}
template <>
__device__ __forceinline__ xfrag load_frag<xf32>(const xf32* p) {
    return x3_split(*reinterpret_cast<const f32x8*>(p));
}
__device__ __forceinline__ void mma16(f32x4& acc, const xfrag& a, const xfrag& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.lo, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b.lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b.hi, acc, 0, 0, 0);
}

// `h2`: the same operand already split, in memory.  32 consecutive k of a row take 128 bytes, [32 x fp16 hi | 32 x fp16 lo] — as many
// bytes as the f32 values, so every pitch and offset of an h2 array is the f32 one (an h2* advances 4 bytes per element) and
// the limbs of element k of a 128-byte-aligned 32-block sit at block + 2 (k % 32) and block + 64 + 2 (k % 32).  A fragment (8
// consecutive k) is two 16-byte loads and no arithmetic; producers write the limbs (store4 / store8 / store1 below).  The
// accessors take the ELEMENT address and find the block from its low seven bits, so kernels index h2 arrays exactly like
// float arrays; every h2 array (global or LDS) therefore starts 128-byte aligned and keeps row pitches that are multiples of 32.
struct h2 { unsigned v; };
template <> struct FragT<h2> { typedef xfrag type; };
__device__ __forceinline__ const char* h2_limb(const void* elem) {   // address of the element's hi limb (its lo limb: + 64)
    const uintptr_t a = (uintptr_t)elem;
    return reinterpret_cast<const char*>((a & ~(uintptr_t)127) + ((a & 127) >> 1));
}
template <>
__device__ __forceinline__ xfrag load_frag<h2>(const h2* p) {   // p = element address of the fragment's first k (a multiple of 8)
    const char* b = h2_limb(p);
    xfrag r;
    r.hi = *reinterpret_cast<const f16x8*>(b);
    r.lo = *reinterpret_cast<const f16x8*>(b + 64);
    return r;
}
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
__device__ __forceinline__ f32x4 load4_f32(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 load4_f32(const xf32* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 load4_f32(const bf16* p) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ f32x4 load4_f32(const h2* p) {   // 4 consecutive elements as f32 (hi + lo: exact)
    const char* b = h2_limb(p);
    const f16x4 hi = *reinterpret_cast<const f16x4*>(b), lo = *reinterpret_cast<const f16x4*>(b + 64);
    return f32x4{(float)hi[0] + (float)lo[0], (float)hi[1] + (float)lo[1], (float)hi[2] + (float)lo[2], (float)hi[3] + (float)lo[3]};
}

template <typename T> __device__ __forceinline__ T cvt_out(float v);
template <> __device__ __forceinline__ xf32 cvt_out<xf32>(float v) { return xf32{v}; }
template <> __device__ __forceinline__ float cvt_out<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 cvt_out<bf16>(float v) { return (bf16)v; }  // v_cvt_pk_bf16_f32: RNE, NaN-safe

template <typename T> __device__ __forceinline__ float cvt_in(T v) { return (float)v; }
template <> __device__ __forceinline__ float cvt_in<xf32>(xf32 v) { return v.v; }

// store 4 consecutive outputs
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<f32x4*>(p) = f32x4{a, b, c, d};
}
__device__ __forceinline__ void store4(xf32* p, float a, float b, float c, float d) {
    *reinterpret_cast<f32x4*>(p) = f32x4{a, b, c, d};
}
__device__ __forceinline__ void store4(h2* p, float a, float b, float c, float d) {   // the limbs of 4 consecutive elements: two 8-byte stores
    char* q = const_cast<char*>(h2_limb(p));
    const f32x4 v = {a, b, c, d};
    const f16x4 hi = __builtin_convertvector(v, f16x4);
    const f16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), f16x4);
    *reinterpret_cast<f16x4*>(q) = hi;
    *reinterpret_cast<f16x4*>(q + 64) = lo;
}
__device__ __forceinline__ void store8(float* p, const f32x8& v) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
// fragment of k-group fg (8 consecutive k) of the 32-deep slab that starts at `slab` — for LDS tiles whose rows are not 128-byte
// aligned (padded pitches), where an h2 block cannot be found from the element address: the slab start is given instead
template <typename T>
__device__ __forceinline__ typename FragT<T>::type slab_frag(const T* slab, int fg) { return load_frag<T>(slab + 8 * fg); }
template <>
__device__ __forceinline__ xfrag slab_frag<h2>(const h2* slab, int fg) {
    const char* b = reinterpret_cast<const char*>(slab) + 16 * fg;
    xfrag r;
    r.hi = *reinterpret_cast<const f16x8*>(b);
    r.lo = *reinterpret_cast<const f16x8*>(b + 64);
    return r;
}
__device__ __forceinline__ void store8(h2* p, const f32x8& v) {   // 8 consecutive elements: two 16-byte stores
    char* q = const_cast<char*>(h2_limb(p));
    const xfrag f = x3_split(v);
    *reinterpret_cast<f16x8*>(q) = f.hi;
    *reinterpret_cast<f16x8*>(q + 64) = f.lo;
}
__device__ __forceinline__ void store1(h2* p, float v) {
    char* q = const_cast<char*>(h2_limb(p));
    const _Float16 hi = (_Float16)v;
    *reinterpret_cast<_Float16*>(q) = hi;
    *reinterpret_cast<_Float16*>(q + 64) = (_Float16)(v - (float)hi);
}
__device__ __forceinline__ void store1(float* p, float v) { *p = v; }
__device__ __forceinline__ void store1(bf16* p, float v) { *p = (bf16)v; }
__device__ __forceinline__ void store4(bf16* p, float a, float b, float c, float d) {
    *reinterpret_cast<bf16x4*>(p) = bf16x4{(bf16)a, (bf16)b, (bf16)c, (bf16)d};
}

// exact (erf) GELU — activation_function "gelu" ([3P] configuration_whisper.py:140)
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7): branch-free, one v_exp + one v_rcp, about half the
// VALU work of libm erff.  The 1e-3 logit budget of the f32 mode is four orders of magnitude above it.
__device__ __forceinline__ float erf_as(float x) {
#pragma clang fp contract(off)   // (only the fmaf calls below fuse: the same bits in every kernel that inlines this)
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = 1.0f - p * t * __expf(-ax * ax);
    return copysignf(e, x);
}
__device__ __forceinline__ float gelu_erf(float x) {
#pragma clang fp contract(off)
    return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f));
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Sum over aligned groups of 8 or 16 consecutive lanes with DPP only (no LDS crossbar, no waits):
// quad_perm[1,0,3,2], quad_perm[2,3,0,1], then row_half_mirror (lane i <-> 7-i pairs the two quads,
// which already hold equal sums), then row_mirror for 16 lanes.  Every lane ends with the group sum.
template <int LANES>
__device__ __forceinline__ float dpp_group_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    if (LANES >= 8) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    if (LANES >= 16) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

// Full-wave (64-lane) sum / max: DPP inside each 16-lane row, then the four row results through SGPRs
// (v_readlane) — no LDS crossbar (ds_bpermute) round trips.
__device__ __forceinline__ float dpp_wave_sum(float v) {
    v = dpp_group_sum<16>(v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)) +
            __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32)) +
            __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48)));
}
__device__ __forceinline__ float dpp_max16(float v) {
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)));
    return v;
}
__device__ __forceinline__ float dpp_wave_max(float v) {
    v = dpp_max16(v);
    return fmaxf(fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)),
                       __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16))),
                 fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32)),
                       __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48))));
}

// All-reduce over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48) without the LDS crossbar:
// v_permlane16_swap (x, x) -> {[r0 r0 r2 r2], [r1 r1 r3 r3]}, v_permlane32_swap (x, x) -> {[r0 r1 r0 r1], [r2 r3 r2 r3]}
// (gfx950; semantics checked in tools/permlane_check.hip).  `__shfl_xor(v, 16)` lowers to ds_bpermute + a wait.
typedef __attribute__((ext_vector_type(2))) unsigned wh_u32x2;
__device__ __forceinline__ float xrow_max(float v) {
    wh_u32x2 t = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(t.x), __uint_as_float(t.y));
    t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(t.x), __uint_as_float(t.y));
}
__device__ __forceinline__ float xrow_sum(float v) {
    wh_u32x2 t = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(t.x) + __uint_as_float(t.y);
    t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(t.x) + __uint_as_float(t.y);
}

typedef __attribute__((ext_vector_type(4))) unsigned wh_u32x4;
// NOTE: hipcc (ROCm 7.2) miscompiles __builtin_bit_cast(bf16x2, <element of a uint vector>) — every
// element resolves to element 0.  Going through memcpy produces the intended register moves.
__device__ __forceinline__ bf16x2 as_bf16x2(unsigned u) { bf16x2 r; __builtin_memcpy(&r, &u, 4); return r; }
template <typename V> __device__ __forceinline__ wh_u32x4 as_u32x4(const V& v) { wh_u32x4 r; __builtin_memcpy(&r, &v, 16); return r; }
// dot product of 8 packed bf16 pairs (two 16-byte chunks) accumulated into t: 4 x v_dot2c_f32_bf16
__device__ __forceinline__ float dot8_bf16(wh_u32x4 a, wh_u32x4 b, float t) {
    t = __builtin_amdgcn_fdot2_f32_bf16(as_bf16x2(a.x), as_bf16x2(b.x), t, false);
    t = __builtin_amdgcn_fdot2_f32_bf16(as_bf16x2(a.y), as_bf16x2(b.y), t, false);
    t = __builtin_amdgcn_fdot2_f32_bf16(as_bf16x2(a.z), as_bf16x2(b.z), t, false);
    t = __builtin_amdgcn_fdot2_f32_bf16(as_bf16x2(a.w), as_bf16x2(b.w), t, false);
    return t;
}
// o0, o1 += p0 * va.{lo,hi} + p1 * vb.{lo,hi} for one packed dword of two keys
__device__ __forceinline__ void pv2_bf16(unsigned va, unsigned vb, bf16x2 pp, float& o0, float& o1) {
    const unsigned lo = __builtin_amdgcn_perm(vb, va, 0x05040100u);  // (va.lo, vb.lo)
    const unsigned hi = __builtin_amdgcn_perm(vb, va, 0x07060302u);  // (va.hi, vb.hi)
    o0 = __builtin_amdgcn_fdot2_f32_bf16(as_bf16x2(lo), pp, o0, false);
    o1 = __builtin_amdgcn_fdot2_f32_bf16(as_bf16x2(hi), pp, o1, false);
}

// order-preserving float <-> uint map for atomicMax on floats of either sign
__device__ __forceinline__ unsigned f2ord(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __host__ __forceinline__ float ord2f(unsigned u) {
    unsigned v = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
#ifdef __HIP_DEVICE_COMPILE__
    return __uint_as_float(v);
#else
    float f;
    __builtin_memcpy(&f, &v, 4);
    return f;
#endif
}

// Sum of the LayerNorm partials {sum x, sum x^2} of row `row` over tiles q, q + step, q + 2 step, ... < n_tiles,
// eight 8-byte loads in flight at a time (clamped re-reads are masked): a run-time loop with one load per
// iteration costs one memory round trip per tile.
__device__ __forceinline__ void ln_partial_sum(const float* __restrict__ part, int n_tiles, int x_mpad, int row, int q, int step,
                                               float& s1, float& s2) {
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    s1 = 0.0f;
    s2 = 0.0f;
    for (int t0 = q; t0 < n_tiles; t0 += 8 * step) {
        f32x2 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int tl = min(t0 + u * step, n_tiles - 1);
            v[u] = *reinterpret_cast<const f32x2*>(part + ((long)tl * x_mpad + row) * 2);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const bool ok = t0 + u * step < n_tiles;
            s1 += ok ? v[u].x : 0.0f;
            s2 += ok ? v[u].y : 0.0f;
        }
    }
}

// Epilogue arithmetic of kernels that must agree BIT FOR BIT whichever of them computes a row (k_dec_gemm / k_dec_gemm_wide,
// k_lm_head / k_lm_head_tile: the choice follows the batch, a clip's result must not): explicit, un-contracted operations, so
// the value does not depend on how the compiler happens to fuse multiplies and adds in each kernel (moving a load in
// k_dec_gemm_wide once changed an fma choice and with it a token of the fp8 path).
__device__ __forceinline__ float wh_ln_fold(float acc, float mean, float rstd, float s, float c) {   // rstd (acc - mean s) + c
#pragma clang fp contract(off)
    const float p = mean * s;
    const float t = acc - p;
    const float u = rstd * t;
    return u + c;
}
// {mean, rstd} of a row from its {sum x, sum x^2}: biased variance, eps 1e-5 (torch LayerNorm)
__device__ __forceinline__ void wh_ln_mean_rstd(float s1, float s2, float inv_or_k, bool is_inv, float& mean, float& rstd) {
#pragma clang fp contract(off)
    mean = is_inv ? s1 * inv_or_k : s1 / inv_or_k;
    const float ex2 = is_inv ? s2 * inv_or_k : s2 / inv_or_k;
    const float var = ex2 - mean * mean;
    rstd = rsqrtf(fmaxf(var, 0.0f) + 1e-5f);
}
__device__ __forceinline__ float wh_scale(float acc, float ws) {   // acc * ws (fp8 weights: the channel scale; exact for ws = 1)
#pragma clang fp contract(off)
    return acc * ws;
}
__device__ __forceinline__ float wh_add(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float wh_sub(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}

#define WH_HIP_CHECK(expr)                                                                      \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) return wh_fail_hip(_e, #expr, __FILE__, __LINE__);                \
    } while (0)

int wh_fail_hip(hipError_t e, const char* what, const char* file, int line);
// Dynamic-LDS opt-in above the 64 KB default, once per (current device, kernel, size): hipFuncSetAttribute applies to the
// device that is current when it is called, so a process that opens several devices needs it on each.  A failure is
// recorded through wh_set_error (the launch that follows then fails and surfaces at the call's hipGetLastError).
bool wh_ensure_dyn_lds(const void* kernel, size_t bytes);
void wh_set_error(const char* fmt, ...);
