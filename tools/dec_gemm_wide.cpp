// dec_gemm_wide.cpp — the decode GEMMs at whisper-large-v3 width (d 1280, ffn 5120) with weights streaming from HBM: every
// launch of a chain reads a different layer's matrix out of a 1.6 GB pool.  Sweeps rows-per-workgroup and the K split.
// Build: hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/dec_gemm_wide.cpp -Lwhisper-rust-ort_amd -lwhisper_hip -Wl,-rpath,$PWD/whisper-rust-ort_amd -o tools/dec_gemm_wide
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>
#include "../whisper-rust-ort_amd/csrc/wh_kernels.h"
#include "../include/whisper_hip.h"
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void* dmalloc(size_t b) { void* p; if (hipMalloc(&p, b) != hipSuccess) { printf("alloc failed\n"); exit(1); } hipMemset(p, 0, b); return p; }
static double time_chain(hipStream_t s, int reps, const std::function<void(int)>& launch) {
    launch(0); hipStreamSynchronize(s);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < reps; i++) launch(i);
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    double best = 1e9;
    for (int r = 0; r < 3; r++) { double t0 = now(); hipGraphLaunch(ge, s); hipStreamSynchronize(s); best = std::min(best, (now() - t0) / reps * 1e6); }
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return best;
}
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32;
    const int d = 1280, F = 5120, L = 32;
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const size_t per_layer = (size_t)F * d * 2;                // the largest matrix
    char* pool = (char*)dmalloc(per_layer * L);                // 32 x 13 MB: consecutive launches never share weights
    void* X = dmalloc((size_t)64 * F * 2); float* xres = (float*)dmalloc((size_t)64 * d * 4); float* bias = (float*)dmalloc(F * 4 * 4);
    void* C1 = dmalloc((size_t)64 * F * 4); float* part = (float*)dmalloc((size_t)(d / 16) * 64 * 2 * 4); float* sv = (float*)dmalloc(F * 4 * 4);
    void* xs = dmalloc((size_t)64 * d * 2); float* st = (float*)dmalloc((size_t)(d / 16) * 64 * 2 * 4);
    struct Case { const char* name; int N, K; bool ln, res; } cases[] = {
        {"LN+QKV   N3840 K1280", 3 * d, d, true, false}, {"o-proj   N1280 K1280 +res+stats", d, d, false, true},
        {"LN+q     N1280 K1280", d, d, true, false}, {"LN+fc1   N5120 K1280", F, d, true, false}, {"fc2      N1280 K5120 +res+stats", d, F, false, true}};
    for (auto& c : cases) {
        const double mb = (double)c.N * c.K * 2 / 1e6;
        for (int nw : {0, 4, 8}) for (int mt : {0, 1, 2}) {
            wh_dbg_nw = nw; wh_dbg_mt = mt;
            SkinnyArgs a; a.bias = bias; a.M = B; a.N = c.N; a.K = c.K; a.X = X; a.x_mpad = 64;
            if (c.ln) { a.ln_part = part; a.ln_tiles = d / 16; a.ln_s = sv; a.C = C1; a.ldc = c.N; }
            if (c.res) { a.R = xres; a.ldr = d; a.C = xres; a.ldc = d; a.xslab_out = xs; a.stats_out = st; }
            const double us = time_chain(s, 64, [&](int i) { SkinnyArgs b = a; b.W = pool + (size_t)(i % L) * per_layer; wh_launch_dec_gemm(s, WH_PREC_BF16, c.res, b); });
            printf("%-34s M=%2d nw=%d mt=%d : %6.2f us  %5.2f TB/s of %5.1f MB\n", c.name, B, nw, mt, us, mb / us, mb);
        }
    }
    return 0;
}
