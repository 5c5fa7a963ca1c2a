// overlap.cpp — do an HBM-bound kernel chain (cross attention) and a latency-bound chain (decode GEMMs)
// on two streams overlap on this part?  Prints each chain alone and both together.
// Build: hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/overlap.cpp -Lwhisper-rust-ort_amd -lwhisper_hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include "../whisper-rust-ort_amd/csrc/wh_kernels.h"
#include "../include/whisper_hip.h"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void* dmalloc(size_t b) { void* p; hipMalloc(&p, b); hipMemset(p, 0, b); return p; }
static hipGraphExec_t capture(hipStream_t s, int reps, const std::function<void()>& f) {
    f(); hipStreamSynchronize(s);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < reps; i++) f();
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    return ge;
}
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32;
    const int d = 512, S = 1500, H = 8, L = 6, splits = 256 / B;
    hipStream_t sa, sb; hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    const size_t plane = (size_t)B * S * d;
    void* kv = dmalloc(plane * 2 * L * 2); void* q = dmalloc((size_t)B * d * 2); void* out = dmalloc((size_t)B * d * 2);
    int* tickets = (int*)dmalloc(64 * 4);
    float* part = (float*)dmalloc((size_t)B * splits * d * 4); float* ml = (float*)dmalloc((size_t)B * splits * H * 2 * 4);
    int l = 0;
    const int nc = 60, ng = 480;
    hipGraphExec_t gc = capture(sa, nc, [&]() { wh_launch_dec_cross_attn(sa, WH_PREC_BF16, q, (char*)kv + (size_t)(2 * l) * plane * 2, (char*)kv + (size_t)(2 * l + 1) * plane * 2, part, ml, S, d, H, splits, B, out, 64, true); l = (l + 1) % L; });
    void* W = dmalloc((size_t)2048 * 2048 * 2); void* X = dmalloc((size_t)64 * 2048 * 2); float* bias = (float*)dmalloc(2048 * 4 * 4); void* C1 = dmalloc((size_t)64 * 2048 * 4 * 3);
    SkinnyArgs a; a.W = W; a.bias = bias; a.M = B; a.N = 1536; a.K = 512; a.X = X; a.x_mpad = 64; a.C = C1; a.c_mpad = 64;
    hipGraphExec_t gg = capture(sb, ng, [&]() { wh_launch_dec_gemm(sb, WH_PREC_BF16, false, a); });
    auto run = [&](bool c, bool g) {
        double best = 1e9;
        for (int r = 0; r < 5; r++) {
            double t0 = now();
            if (c) hipGraphLaunch(gc, sa);
            if (g) hipGraphLaunch(gg, sb);
            if (c) hipStreamSynchronize(sa);
            if (g) hipStreamSynchronize(sb);
            best = std::min(best, (now() - t0) * 1e3);
        }
        return best;
    };
    double tc = run(true, false), tg = run(false, true), tb = run(true, true);
    printf("B=%d  cross x%d alone %.3f ms (%.1f us each) | gemm x%d alone %.3f ms (%.2f us each) | both %.3f ms  (sum %.3f, max %.3f)\n",
           B, nc, tc, tc / nc * 1e3, ng, tg, tg / ng * 1e3, tb, tc + tg, tc > tg ? tc : tg);
    return 0;
}
