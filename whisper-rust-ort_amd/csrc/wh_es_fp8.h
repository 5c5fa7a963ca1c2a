// wh_es_fp8.h — device helpers shared by the encoder-state cross-attention kernels that run on the fp8 matrix cores
// (wh_cross_es8.hip: e4m3 states; wh_cross_es3.hip: fp16 states + e4m3 remainder): LDS-DMA pieces, the swizzle of 512-byte e4m3 rows,
// e4m3 head + remainder operands, lane ^ 8 exchanges.
#pragma once
#include "wh_common.h"

namespace wh_es_fp8 {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr float REM = 16.0f, REM_INV = 1.0f / 16.0f;   // scale of the e4m3 remainder rows of queries and probabilities

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int AUX>
__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {   // one LDS-DMA piece: 64 lanes x 16 bytes, written linearly from lds_wave_base
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, AUX);
}
// 512-byte e4m3 rows: LDS chunk (16 bytes) p of tile row r holds chunk p ^ swz8(r) — both ds_read_b64 patterns of the kernels (8 dims of a key per
// lane: keys across fl for the scores, dims across fl for the output blocks) then touch 32 distinct 8-byte units per 32-lane service group
__device__ __forceinline__ int swz8(int r) { return (r & 15) ^ (((r >> 4) & 1) << 3); }
__device__ __forceinline__ float ror8(float v) {    // v of lane ^ 8 (same 16-lane row): DPP row_ror:8
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
}
__device__ __forceinline__ unsigned ror8u(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true); }
// four f32 -> four e4m3 bytes (byte u = v[u])
__device__ __forceinline__ unsigned pack4(float a, float b, float c, float d) {
    int p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(c, d, p, true);
}
// the remainder limb of four values whose head limb is `hi`: e4m3(16 (v - hi))
__device__ __forceinline__ unsigned rem4(unsigned hi, float a, float b, float c, float d) {
    const float h0 = __builtin_amdgcn_cvt_f32_fp8((int)hi, 0), h1 = __builtin_amdgcn_cvt_f32_fp8((int)hi, 1);
    const float h2_ = __builtin_amdgcn_cvt_f32_fp8((int)hi, 2), h3 = __builtin_amdgcn_cvt_f32_fp8((int)hi, 3);
    return pack4((a - h0) * REM, (b - h1) * REM, (c - h2_) * REM, (d - h3) * REM);
}
__device__ __forceinline__ long join(unsigned lo, unsigned hi) { return (long)(((unsigned long long)hi << 32) | lo); }

}  // namespace wh_es_fp8
