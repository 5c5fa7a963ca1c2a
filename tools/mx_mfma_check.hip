// mx_mfma_check.hip — operand layout and scale semantics of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands, checked
// against a host sum with exactly representable values (integers * powers of two: the f32 accumulation is exact).
//   lane l supplies, for its row (l & 15) and g = l >> 4, the codes k = 16 g .. 16 g + 15 (VGPRs 0-3) and k = 64 + 16 g .. + 15
//   (VGPRs 4-7), and ONE E8M0 scale byte (byte `opsel` of the scale register) for the 32-block k / 32 = g of that row
//   (tools/mx_scale_probe.hip found this mapping);  D[i][j] += sum_k A[i][k] 2^(sa-127) B[j][k] 2^(sb-127),
//   D row i = 4 (l >> 4) + r, column j = l & 15 (first operand = rows of D).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int MODE>
__global__ void k(const unsigned char* A, const unsigned char* B, const unsigned char* sa, const unsigned char* sb, float* D) {
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    typedef __attribute__((ext_vector_type(4))) int i32x4;
    const i32x4 a0 = *reinterpret_cast<const i32x4*>(A + r * 128 + g * 16), a1 = *reinterpret_cast<const i32x4*>(A + r * 128 + 64 + g * 16);
    const i32x4 b0 = *reinterpret_cast<const i32x4*>(B + r * 128 + g * 16), b1 = *reinterpret_cast<const i32x4*>(B + r * 128 + 64 + g * 16);
    const i32x8 a = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]}, b = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    f32x4 c = {0, 0, 0, 0};
    if (MODE == 0) {   // scale in byte 0, opsel 0
        c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, (int)sa[r * 4 + g], 0, (int)sb[r * 4 + g]);
    } else if (MODE == 1) {   // scale in byte 1 of a, byte 2 of b: opsel 1 / 2
        c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 1, (int)sa[r * 4 + g] << 8, 2, (int)sb[r * 4 + g] << 16);
    } else {   // same scale replicated in every byte
        c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, (int)sa[r * 4 + g] * 0x01010101, 0, (int)sb[r * 4 + g] * 0x01010101);
    }
    for (int e = 0; e < 4; e++) D[(4 * g + e) * 16 + r] = c[e];
}
static float e4m3(unsigned char c) { int e = (c >> 3) & 15, m = c & 7; float v = e == 0 ? m * ldexpf(1.0f, -9) : (8 + m) * ldexpf(1.0f, e - 10); return (c & 0x80) ? -v : v; }
int main() {
    std::vector<unsigned char> A(16 * 128), B(16 * 128), sa(64), sb(64);
    unsigned x = 7;
    auto rnd = [&]() { x = x * 1664525u + 1013904223u; return x >> 8; };
    for (auto& v : A) { unsigned c = rnd() % 0x50; v = (unsigned char)((c + 0x20) | ((rnd() & 1) << 7)); }   // moderate magnitudes, both signs
    for (auto& v : B) { unsigned c = rnd() % 0x50; v = (unsigned char)((c + 0x20) | ((rnd() & 1) << 7)); }
    for (auto& v : sa) v = (unsigned char)(127 - 3 + rnd() % 7);
    for (auto& v : sb) v = (unsigned char)(127 - 2 + rnd() % 5);
    unsigned char *dA, *dB, *dsa, *dsb; float* dD;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dsa, 64); hipMalloc(&dsb, 64); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    hipMemcpy(dsa, sa.data(), 64, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 64, hipMemcpyHostToDevice);
    int bad = 0;
    for (int uniform = 1; uniform >= 0; uniform--) {
        if (uniform) { for (auto& v : sa) v = 127; for (auto& v : sb) v = 127; }
        else { for (auto& v : sa) v = (unsigned char)(127 - 3 + rnd() % 7); for (auto& v : sb) v = (unsigned char)(127 - 2 + rnd() % 5); }
        hipMemcpy(dsa, sa.data(), 64, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 64, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 3; mode++) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
            else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
            else hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
            float D[256]; hipMemcpy(D, dD, 1024, hipMemcpyDeviceToHost);
            double worst = 0, worstT = 0;
            for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
                double s = 0;
                for (int kk = 0; kk < 128; kk++) s += (double)e4m3(A[i * 128 + kk]) * ldexp(1.0, sa[i * 4 + kk / 32] - 127) * (double)e4m3(B[j * 128 + kk]) * ldexp(1.0, sb[j * 4 + kk / 32] - 127);
                worst = fmax(worst, fabs(s - D[i * 16 + j]) / fmax(1.0, fabs(s)));
                worstT = fmax(worstT, fabs(s - D[j * 16 + i]) / fmax(1.0, fabs(s)));
            }
            printf("scales %s, mode %d: max relative deviation %.3g (transposed D: %.3g)\n", uniform ? "uniform" : "varied", mode, worst, worstT);
            if (mode == 0 && worst > 1e-3) bad = 1;   // the instruction accumulates with slightly less than f32 precision: 1e-4 seen
        }
    }
    return bad;
}
