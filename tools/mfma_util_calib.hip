// mfma_util_calib.hip — a kernel whose MFMA pipe is busy all the time (four independent v_mfma_f32_16x16x32_bf16 chains per wave,
// four waves per SIMD), to learn how rocprofv3's SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE aggregate on gfx950 before they
// are turned into "MFMA utilisation" for the product's kernels (profiles/mfma_util.py).
// Build: hipcc --offload-arch=gfx950 -O2 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 tools/mfma_util_calib.hip -o tools/mfma_util_calib
// (the library's flag: accumulators in VGPRs.  Without it the compiler keeps them in AGPRs and the 16x16x32 loop drops from 1.89 to 1.45 PFLOP/s.)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ __launch_bounds__(256) void k_mfma_busy(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x ^ i)); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_mfma_busy32(float* out, int iters) {   // the 32x32x16 shape: 32 cycles of issue each
    bf16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x ^ i)); }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + (threadIdx.x & 255)] = c0[0] + c1[1] + c2[2] + c3[3];
}
// the same with half the waves idle in a scalar sleep loop for as long: expected utilisation one half per SIMD-time
__global__ __launch_bounds__(256) void k_mfma_half(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x ^ i)); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0;
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
        __builtin_amdgcn_s_sleep(2);   // 128 cycles asleep per 2 MFMAs (~32 busy cycles): mostly idle
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1];
}
int main() {
    float* out; hipMalloc(&out, 1024 * 256 * 4);
    const int iters = 20000;
    for (int r = 0; r < 2; r++) {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_mfma_busy, dim3(1024), dim3(256), 0, 0, out, iters);
        hipDeviceSynchronize();
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const double n = 1024.0 * 4 * iters * 4;
        printf("k_mfma_busy: %.3f ms, %.0f MFMAs, %.1f TFLOP/s, %.2f cycles per MFMA per SIMD at 2.4 GHz\n", s * 1e3, n, n * 16384 / s / 1e12, s * 2.4e9 / (n / 1024));
    }
    for (int waves : {4, 8}) for (int r = 0; r < 2; r++) {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        if (waves == 4) hipLaunchKernelGGL(k_mfma_busy32<4>, dim3(1024), dim3(256), 0, 0, out, iters / 2);
        else hipLaunchKernelGGL(k_mfma_busy32<8>, dim3(512), dim3(512), 0, 0, out, iters / 2);
        hipDeviceSynchronize();
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const double n = 1024.0 * 4 * (iters / 2) * 4;
        printf("k_mfma_busy32 (%d-wave workgroups): %.3f ms, %.0f MFMAs, %.1f TFLOP/s, %.2f cycles per MFMA per SIMD at 2.4 GHz\n", waves, s * 1e3, n, n * 32768 / s / 1e12, s * 2.4e9 / (n / 1024));
    }
    hipLaunchKernelGGL(k_mfma_half, dim3(1024), dim3(256), 0, 0, out, iters / 4);
    hipDeviceSynchronize();
    return 0;
}
