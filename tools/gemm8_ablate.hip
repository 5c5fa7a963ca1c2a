// gemm8_ablate.hip — where does k_gemm8 spend its time?  Times the library kernel and its ablated variants on the
// encoder's GEMM shapes (256 clips): full / no MFMA / no LDS-DMA in the loop / no epilogue / loads + barriers only.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I whisper-rust-ort_amd/csrc tools/gemm8_ablate.hip -o tools/gemm8_ablate
#include "../whisper-rust-ort_amd/csrc/wh_gemm8.hip"
#include <cstdio>
#include <vector>
bool wh_ensure_dyn_lds(const void* k, size_t b) { return hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b) == hipSuccess; }
void wh_set_error(const char*, ...) {}
template <typename TO, int BN, int ABL> float run(const GemmArgs& g, int reps) {
    typedef Geo<BN> G;
    const size_t sm = (size_t)G::NSLOT * G::SLOT;
    (void)hipFuncSetAttribute((const void*)k_gemm8<TO, BN, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    dim3 grid(((g.N + BN - 1) / BN) * ((g.M + BM - 1) / BM), 1, g.batch);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k_gemm8<TO, BN, ABL>), grid, dim3(512), sm, 0, g);
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_gemm8<TO, BN, ABL>), grid, dim3(512), sm, 0, g);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    return ms / reps * 1e3f;
}
template <typename TO, int BN> void all(const char* name, const GemmArgs& g, double gf) {
    const float t0 = run<TO, BN, 0>(g, 5), t1 = run<TO, BN, 1>(g, 5), t4 = run<TO, BN, 4>(g, 5), t12 = run<TO, BN, 8 | 4>(g, 5), t16 = run<TO, BN, 16>(g, 5);
    printf("%-26s BN %3d: full %7.1f us (%5.0f TF/s) | no-mfma %7.1f | no-epilogue %7.1f | loads+barriers only %7.1f | no-stores %7.1f\n", name, BN, t0,
           gf / t0 * 1e3, t1, t4, t12, t16);
}
int main() {
    const long M = 256L * 1500;
    struct Shape { const char* name; int N, K; bool f32out, resid, act; } shapes[] = {
        {"QK   N1024 K512 ", 1024, 512, false, false, false}, {"fc1  N2048 K512 gelu", 2048, 512, false, false, true},
        {"fc2  N512 K2048 f32+res", 512, 2048, true, true, false}, {"O    N512 K512 f32+res", 512, 512, true, true, false}};
    bf16 *A, *W; float *R, *bias; void* C;
    hipMalloc(&A, M * 2048 * 2); hipMalloc(&W, 2048L * 2048 * 2); hipMalloc(&C, M * 2048 * 2); hipMalloc(&R, M * 512 * 4); hipMalloc(&bias, 8192);
    std::vector<unsigned short> h(1 << 24);
    unsigned x = 12345; for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3c00 + ((x >> 20) & 0x1ff) + ((x >> 31) << 15)); }  // ~+-1
    for (long off = 0; off < M * 2048 * 2; off += (long)h.size() * 2) hipMemcpy((char*)A + off, h.data(), std::min<long>(h.size() * 2, M * 2048 * 2 - off), hipMemcpyHostToDevice);
    hipMemcpy(W, h.data(), 2048L * 2048 * 2, hipMemcpyHostToDevice);
    hipMemset(R, 0, M * 512 * 4); hipMemset(bias, 0, 8192);
    for (auto& sh : shapes) {
        GemmArgs g; g.A = A; g.lda = sh.K; g.W = W; g.ldw = sh.K; g.C = sh.f32out ? (void*)R : C; g.ldc = sh.N; g.bias = bias; g.bias_mode = 1;
        g.act = sh.act; g.M = (int)M; g.N = sh.N; g.K = sh.K;
        if (sh.resid) { g.R = R; g.ldr = sh.N; }
        const double gf = 2.0 * M * sh.N * sh.K * 1e-9;
        if (sh.f32out) { all<float, 128>(sh.name, g, gf); all<float, 256>(sh.name, g, gf); }
        else { all<bf16, 128>(sh.name, g, gf); all<bf16, 256>(sh.name, g, gf); }
    }
    return 0;
}
