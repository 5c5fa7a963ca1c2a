// dec_gemm_sweep.cpp — the six decode GEMM shapes of a layer at a given width and row count, weights cycling through a pool
// larger than the caches, over the K split (4 / 8 waves), the rows per workgroup of k_dec_gemm and the column tiles per
// workgroup of k_dec_gemm_wide.  Usage: dec_gemm_sweep <rows> <d_model> <ffn> [matrices in the weight pool]
// Build: hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/dec_gemm_sweep.cpp -Lwhisper-rust-ort_amd -lwhisper_hip -Wl,-rpath,$PWD/whisper-rust-ort_amd -o tools/dec_gemm_sweep
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
    const int B = argc > 1 ? atoi(argv[1]) : 256, d = argc > 2 ? atoi(argv[2]) : 1280, F = argc > 3 ? atoi(argv[3]) : 5120;
    const int MP = (B + 15) / 16 * 16, L = argc > 4 ? atoi(argv[4]) : (d >= 1024 ? 32 : 6);   // matrices in the pool (32 x 13 MB: HBM; 4: Infinity Cache; 1: L2)
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const size_t per_layer = (size_t)F * d * 2;
    char* pool = (char*)dmalloc(per_layer * L);
    void* X = dmalloc((size_t)MP * F * 2); float* xres = (float*)dmalloc((size_t)MP * d * 4); float* bias = (float*)dmalloc(F * 4 * 4);
    void* C1 = dmalloc((size_t)MP * F * 4); float* part = (float*)dmalloc((size_t)(d / 16) * MP * 2 * 4); float* sv = (float*)dmalloc(F * 4 * 4);
    void* xs = dmalloc((size_t)MP * d * 2); float* st = (float*)dmalloc((size_t)(d / 16) * MP * 2 * 4);
    struct Case { const char* name; int N, K; bool ln, res; } cases[] = {
        {"LN+QKV", 3 * d, d, true, false}, {"o-proj +res+stats", d, d, false, true},
        {"LN+q", d, d, true, false}, {"LN+fc1", F, d, true, false}, {"fc2 +res+stats", d, F, false, true}};
    printf("rows %d, d_model %d, ffn %d, pool of %d matrices  (us per launch in a replayed chain, boundary included)\n", B, d, F, L);
    for (auto& c : cases) {
        printf("%-18s N%5d K%5d :", c.name, c.N, c.K);
        double best = 1e9; char bestn[32] = "";
        for (int nw : {4, 8}) for (int cfg : {1, 2, 4, -2, -4}) {   // > 0: rows/16 per workgroup of k_dec_gemm; < 0: column tiles of k_dec_gemm_wide
            if (cfg < 0 && (B <= 64 || (cfg == -4 && nw == 8))) continue;
            if (c.K % (nw * 32) != 0) continue;
            wh_dbg_nw = nw; wh_dbg_mt = cfg > 0 ? cfg : 0; wh_dbg_wide = cfg > 0 ? 0 : -cfg;
            SkinnyArgs a; a.bias = bias; a.M = B; a.N = c.N; a.K = c.K; a.X = X; a.x_mpad = MP;
            if (c.ln) { a.ln_part = part; a.ln_tiles = d / 16; a.ln_s = sv; a.C = C1; a.ldc = c.N; }
            if (c.res) { a.R = xres; a.ldr = d; a.C = xres; a.ldc = d; a.xslab_out = xs; a.stats_out = st; }
            const double us = time_chain(s, 48, [&](int i) { SkinnyArgs b = a; b.W = pool + (size_t)(i % L) * per_layer; wh_launch_dec_gemm(s, WH_PREC_BF16, c.res, b); });
            char nm[32]; snprintf(nm, sizeof nm, "nw%d %s%d", nw, cfg > 0 ? "mt" : "nt", cfg > 0 ? cfg : -cfg);
            printf("  %s %6.2f", nm, us);
            if (us < best) { best = us; snprintf(bestn, sizeof bestn, "%s", nm); }
        }
        printf("   | best %s %.2f us = %.2f TB/s weights, %.0f TF/s\n", bestn, best, (double)c.N * c.K * 2 / 1e6 / best, 2.0 * B * c.N * c.K / 1e6 / best);
    }
    return 0;
}
