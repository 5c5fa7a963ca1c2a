#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__global__ void k(const unsigned* a, const unsigned* b, float* out, unsigned* pout) {
    int i = threadIdx.x;
    out[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a[i]), __builtin_bit_cast(bf16x2, b[i]), 100.0f, false);
    pout[2 * i] = __builtin_amdgcn_perm(b[i], a[i], 0x05040100u);
    pout[2 * i + 1] = __builtin_amdgcn_perm(b[i], a[i], 0x07060302u);
    bf16x2 pp = bf16x2{(__bf16)1.5f, (__bf16)-2.0f};
    if (i == 0) pout[200] = __builtin_bit_cast(unsigned, pp);
}
static unsigned short f2b(float f) { unsigned u; memcpy(&u, &f, 4); return u >> 16; }
int main() {
    unsigned ha[4], hb[4];
    float av[4][2] = {{1, 2}, {0.5f, -3}, {1e-3f, 7}, {100, 0.25f}}, bv[4][2] = {{3, 4}, {2, 2}, {1000, 0.5f}, {0.01f, 8}};
    for (int i = 0; i < 4; i++) { ha[i] = f2b(av[i][0]) | (f2b(av[i][1]) << 16); hb[i] = f2b(bv[i][0]) | (f2b(bv[i][1]) << 16); }
    unsigned *da, *db, *dp; float* dout;
    hipMalloc(&da, 16); hipMalloc(&db, 16); hipMalloc(&dout, 16); hipMalloc(&dp, 1024);
    hipMemcpy(da, ha, 16, hipMemcpyHostToDevice); hipMemcpy(db, hb, 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, da, db, dout, dp);
    float ho[4]; unsigned hp[256];
    hipMemcpy(ho, dout, 16, hipMemcpyDeviceToHost); hipMemcpy(hp, dp, 1024, hipMemcpyDeviceToHost);
    for (int i = 0; i < 4; i++) printf("dot2: got %.6f expect %.6f | a=%08x b=%08x perm_lo=%08x perm_hi=%08x\n", ho[i], 100 + av[i][0] * bv[i][0] + av[i][1] * bv[i][1], ha[i], hb[i], hp[2 * i], hp[2 * i + 1]);
    printf("pack {1.5,-2.0} = %08x (1.5=%04x -2.0=%04x)\n", hp[200], f2b(1.5f), f2b(-2.0f));
    return 0;
}
