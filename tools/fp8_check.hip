// fp8_check.hip — gfx950 fp8 conversion instructions against the host e4m3 tables (exact comparison).
// Build: hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/fp8_check.hip -o tools/fp8_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(2))) float f2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;

static uint8_t q_host(float x) {
    uint32_t u; memcpy(&u, &x, 4);
    const uint8_t sign = (uint8_t)((u >> 24) & 0x80);
    if (x != x) return 0x7F;
    float a = fabsf(x); if (a > 448.0f) a = 448.0f;
    if (a >= 0.015625f) { memcpy(&u, &a, 4); u += 0x7FFFFu + ((u >> 20) & 1u); uint32_t c = (((u >> 23) - 120u) << 3) | ((u >> 20) & 7u); if (c > 0x7E) c = 0x7E; return (uint8_t)(c | sign); }
    return (uint8_t)((uint32_t)nearbyint((double)a * 512.0) | sign);
}
static float dq_host(uint8_t c) { int e = (c >> 3) & 15, m = c & 7; float mag = e == 0 ? m * 0.001953125f : ldexpf((float)(8 + m), e - 10); if ((c & 0x7F) == 0x7F) mag = NAN; return (c & 0x80) ? -mag : mag; }

__global__ void k_dec(float* f32out, float* bfout) {  // thread t: code t in every byte position
    const unsigned c = threadIdx.x;
    const unsigned u = c | ((c ^ 0x55u) << 8) | ((c ^ 0xAAu) << 16) | ((c ^ 0xFFu) << 24);
    f2 lo = __builtin_amdgcn_cvt_pk_f32_fp8(u, false), hi = __builtin_amdgcn_cvt_pk_f32_fp8(u, true);
    f32out[c * 4 + 0] = lo.x; f32out[c * 4 + 1] = lo.y; f32out[c * 4 + 2] = hi.x; f32out[c * 4 + 3] = hi.y;
    bf2 bl = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(u, 1.0f, false), bh = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(u, 1.0f, true);
    bfout[c * 4 + 0] = (float)bl.x; bfout[c * 4 + 1] = (float)bl.y; bfout[c * 4 + 2] = (float)bh.x; bfout[c * 4 + 3] = (float)bh.y;
}
__global__ void k_enc(const float* x, int n, uint8_t* out) {
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 >= n) return;
    auto cl = [](float v) { return fminf(fmaxf(v, -448.0f), 448.0f); };  // the instruction returns NaN above 448: clamp first
    int p = __builtin_amdgcn_cvt_pk_fp8_f32(cl(x[i]), cl(x[i + 1]), 0, false);
    p = __builtin_amdgcn_cvt_pk_fp8_f32(cl(x[i + 2]), cl(x[i + 3]), p, true);
    *reinterpret_cast<int*>(out + i) = p;
}
int main() {
    float *d32, *dbf; hipMalloc(&d32, 4096); hipMalloc(&dbf, 4096);
    hipLaunchKernelGGL(k_dec, dim3(1), dim3(256), 0, 0, d32, dbf);
    float h32[1024], hbf[1024]; hipMemcpy(h32, d32, 4096, hipMemcpyDeviceToHost); hipMemcpy(hbf, dbf, 4096, hipMemcpyDeviceToHost);
    int bad32 = 0, badbf = 0;
    for (int c = 0; c < 256; c++) {
        const uint8_t b[4] = {(uint8_t)c, (uint8_t)(c ^ 0x55), (uint8_t)(c ^ 0xAA), (uint8_t)(c ^ 0xFF)};
        for (int j = 0; j < 4; j++) {
            const float r = dq_host(b[j]);
            if (r != r) continue;
            if (h32[c * 4 + j] != r) { if (bad32 < 5) printf("f32 code %02x pos %d: got %g want %g\n", b[j], j, h32[c * 4 + j], r); bad32++; }
            if (hbf[c * 4 + j] != r) { if (badbf < 5) printf("bf16 code %02x pos %d: got %g want %g\n", b[j], j, hbf[c * 4 + j], r); badbf++; }
        }
    }
    printf("decode: cvt_pk_f32_fp8 mismatches %d, cvt_scalef32_pk_bf16_fp8 mismatches %d (byte j of the dword = element j)\n", bad32, badbf);
    const int n = 1 << 20;
    std::vector<float> x(n);
    uint64_t s = 12345;
    for (int i = 0; i < n; i++) { s = s * 6364136223846793005ull + 1442695040888963407ull; const float u = (float)((s >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f; const int e = (int)((s >> 20) & 31) - 20; x[i] = ldexpf(u, e) * 448.0f; }
    x[0] = 448.0f; x[1] = -448.0f; x[2] = 0.0f; x[3] = 447.9f; x[4] = 0.001953125f; x[5] = 0.0009765625f; x[6] = 0.0029296875f; x[7] = 0.015625f;
    float* dx; uint8_t* dq; hipMalloc(&dx, n * 4); hipMalloc(&dq, n);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_enc, dim3(n / 4 / 256), dim3(256), 0, 0, dx, n, dq);
    std::vector<uint8_t> q(n); hipMemcpy(q.data(), dq, n, hipMemcpyDeviceToHost);
    int badq = 0;
    for (int i = 0; i < n; i++) if (q[i] != q_host(x[i])) { if (badq < 8) printf("encode x=%g: hw %02x host %02x\n", x[i], q[i], q_host(x[i])); badq++; }
    printf("encode: clamp + cvt_pk_fp8_f32 mismatches %d of %d\n", badq, n);
    return 0;
}
