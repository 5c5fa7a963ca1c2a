// permlane_check.hip — semantics of v_permlane16_swap / v_permlane32_swap on gfx950 (for cross-row reductions
// without the LDS crossbar).  Prints, for input lane ids, what each lane holds after the swap.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) unsigned u2;
__global__ void k(unsigned* out) {
    const unsigned l = threadIdx.x;
    u2 a = __builtin_amdgcn_permlane16_swap(l, l + 100, false, false);
    u2 b = __builtin_amdgcn_permlane32_swap(l, l + 100, false, false);
    out[l] = a.x; out[64 + l] = a.y; out[128 + l] = b.x; out[192 + l] = b.y;
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* nm[4] = {"permlane16_swap.x", "permlane16_swap.y", "permlane32_swap.x", "permlane32_swap.y"};
    for (int r = 0; r < 4; r++) { printf("%s:", nm[r]); for (int i = 0; i < 64; i += 4) printf(" %u", h[r * 64 + i]); printf("\n"); }
    return 0;
}
