// mx_scale_probe.hip — which (row, 32-block of K) does a lane's E8M0 scale byte apply to in v_mfma_scale_f32_16x16x128_f8f6f4,
// and which k do a lane's two 16-byte halves hold?  (Result, MI355X / ROCm 7.2: lane (row r = l & 15, group g = l >> 4) holds
// k = 16 g .. 16 g + 15 in its first four VGPRs and k = 64 + 16 g .. 64 + 16 g + 15 in the last four; its scale byte applies
// to row r, k-block g = k / 32.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
// A all ones; B ones only in half hB (0: VGPRs 0-3, 1: VGPRs 4-7) of lane group gB; A scale x2 only in lanes (row rS or any, group gS)
__global__ void k(int gB, int hB, int gS, int rS, float* D) {
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    i32x8 a, b;
    for (int e = 0; e < 8; e++) { a[e] = 0x38383838; b[e] = (g == gB && (e >> 2) == hB) ? 0x38383838 : 0; }
    const int sa = (g == gS && (rS < 0 || r == rS)) ? 128 : 127;
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa, 0, 127);
    for (int e = 0; e < 4; e++) D[(4 * g + e) * 16 + r] = c[e];
}
int main() {
    float* dD; hipMalloc(&dD, 1024); float D[256];
    printf("D[0][0]: B non-zero in (lane group gB, half hB) (rows), A scale doubled in lane group gS (columns); 16 = untouched, 32 = scaled\n");
    for (int gB = 0; gB < 4; gB++)
        for (int hB = 0; hB < 2; hB++) {
            printf("gB %d half %d:", gB, hB);
            for (int gS = 0; gS < 4; gS++) {
                hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, gB, hB, gS, -1, dD);
                hipMemcpy(D, dD, 1024, hipMemcpyDeviceToHost);
                printf(" %4.0f", D[0]);
            }
            printf("\n");
        }
    printf("A scale doubled only in lane (row 3, group 0), B = ones in (group 0, half 0): D[i][0] for i = 0..15:");
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, 0, 0, 0, 3, dD);
    hipMemcpy(D, dD, 1024, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; i++) printf(" %.0f", D[i * 16]);
    printf("\n");
    return 0;
}
