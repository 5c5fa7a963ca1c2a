// microbench.cpp — isolated timings of the decode kernels (graph-replayed chains of N launches).
// Build: hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/microbench.cpp -Lwhisper-rust-ort_amd -lwhisper_hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>
#include "../whisper-rust-ort_amd/csrc/wh_kernels.h"
#include "../include/whisper_hip.h"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void* dmalloc(size_t b, int fill = 0) { void* p; hipMalloc(&p, b); hipMemset(p, fill, b); return p; }

static double time_chain(hipStream_t s, int reps, const std::function<void()>& launch_once) {
    launch_once(); hipStreamSynchronize(s);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < reps; i++) launch_once();
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    double best = 1e9;
    for (int r = 0; r < 5; r++) { double t0 = now(); hipGraphLaunch(ge, s); hipStreamSynchronize(s); best = std::min(best, (now() - t0) / reps * 1e6); }
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return best;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64;
    setvbuf(stdout, nullptr, _IOLBF, 0);
    if (B < 1 || B > 64) { fprintf(stderr, "B must be 1..64 (buffers are sized for 64 rows)\n"); return 2; }
    const int d = 512, F = 2048, S = 1500, H = 8, V = 51865;
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const int prec = WH_PREC_BF16;
    // buffers (bf16 = 2 bytes); contents zero → finite everywhere
    void* W = dmalloc((size_t)V * d * 2); void* X = dmalloc((size_t)64 * F * 2); float* xres = (float*)dmalloc((size_t)64 * d * 4);
    float* lnw = (float*)dmalloc(d * 4); float* lnb = (float*)dmalloc(d * 4); float* bias = (float*)dmalloc((size_t)52000 * 4);
    void* C1 = dmalloc((size_t)64 * 3 * F * 4); int* pos = (int*)dmalloc(4);
    for (auto nk : std::vector<std::pair<int, int>>{{512, 512}, {1536, 512}, {2048, 512}, {512, 2048}}) {
        SkinnyArgs a; a.W = W; a.bias = bias; a.M = B; a.N = nk.first; a.K = nk.second; a.X = X; a.x_mpad = 64;
        const bool res = nk.first == d;
        if (res) { a.R = xres; a.ldr = d; a.C = xres; a.ldc = d; } else { a.C = C1; a.c_mpad = 64; }
        for (int dm : {4, 2, 1}) {
            wh_dbg_mt = dm;
            double us = time_chain(s, 200, [&]() { wh_launch_dec_gemm(s, prec, res, a); });
            printf("dec_gemm M=%d N=%4d K=%4d rows/wg=%d : %.2f us\n", B, nk.first, nk.second, dm * 16, us);
        }
        wh_dbg_mt = 0;
    }
    for (auto nk : std::vector<std::pair<int, int>>{{512, 512}, {1536, 512}, {2048, 512}, {512, 2048}}) {   // e4m3 weight codes
        SkinnyArgs a; a.W = W; a.bias = bias; a.wscale = bias; a.M = B; a.N = nk.first; a.K = nk.second; a.X = X; a.x_mpad = 64;
        const bool res = nk.first == d;
        if (res) { a.R = xres; a.ldr = d; a.C = xres; a.ldc = d; } else { a.C = C1; a.c_mpad = 64; }
        printf("dec_gemm fp8 weights M=%d N=%4d K=%4d : %.2f us\n", B, nk.first, nk.second, time_chain(s, 200, [&]() { wh_launch_dec_gemm(s, WH_PREC_FP8, res, a); }));
    }
    {   // LN-folded consumer + stats-producing residual GEMM
        float* part = (float*)dmalloc(32 * 64 * 2 * 4); float* sv = (float*)dmalloc(F * 4);
        SkinnyArgs a; a.W = W; a.bias = bias; a.M = B; a.N = 3 * d; a.K = d; a.X = X; a.x_mpad = 64; a.C = C1; a.ldc = 3 * d; a.ln_part = part; a.ln_tiles = 32; a.ln_s = sv;
        printf("dec_gemm LN-folded consumer N=1536 : %.2f us\n", time_chain(s, 200, [&]() { wh_launch_dec_gemm(s, prec, false, a); }));
        SkinnyArgs p; p.W = W; p.bias = bias; p.M = B; p.N = d; p.K = d; p.X = X; p.x_mpad = 64; p.R = xres; p.ldr = d; p.C = xres; p.ldc = d; p.xslab_out = C1; p.stats_out = part;
        printf("dec_gemm stats producer N=512 : %.2f us\n", time_chain(s, 200, [&]() { wh_launch_dec_gemm(s, prec, true, p); }));
        float* cp = (float*)dmalloc((size_t)64 * 32 * d * 4); float* cm = (float*)dmalloc((size_t)64 * 32 * 8 * 2 * 4);
        p.X = nullptr; p.xpart = cp; p.xml = cm; p.x_heads = 8;
        for (int sp : {1, 2, 4, 8, 16, 32}) {
            p.x_splits = sp;
            printf("dec_gemm stats producer N=512, X merged from %d attention partials : %.2f us\n", p.x_splits, time_chain(s, 200, [&]() { wh_launch_dec_gemm(s, prec, true, p); }));
        }
    }
    {   // encoder-shaped GEMMs: M = B*1500 rows
        const int M = B * 1500;
        void* A = dmalloc((size_t)M * 2048 * 2, 0); void* Wt = dmalloc((size_t)2048 * 2048 * 2); void* Cc = dmalloc((size_t)M * 2048 * 4);
        for (auto nk : std::vector<std::pair<int, int>>{{512, 512}, {1024, 512}, {2048, 512}, {512, 2048}}) {
            GemmArgs g; g.A = A; g.lda = nk.second; g.W = Wt; g.ldw = nk.second; g.C = Cc; g.ldc = nk.first; g.bias = bias; g.bias_mode = 1;
            g.M = M; g.N = nk.first; g.K = nk.second;
            double us = time_chain(s, 10, [&]() { wh_launch_gemm(s, prec, false, g); });
            printf("enc gemm M=%d N=%4d K=%4d : %.1f us  %.0f TF/s\n", M, nk.first, nk.second, us, 2.0 * M * nk.first * nk.second / us / 1e6);
            if (nk.first == 2048) {
                g.act = 1;
                us = time_chain(s, 10, [&]() { wh_launch_gemm(s, prec, false, g); });
                printf("enc gemm M=%d N=%4d K=%4d + GELU : %.1f us  %.0f TF/s\n", M, nk.first, nk.second, us, 2.0 * M * nk.first * nk.second / us / 1e6);
                g.act = 0;
            }
            if (nk.first == 512) {
                g.R = (const float*)Cc; g.ldr = 512; g.C = (char*)Cc + (size_t)M * 512 * 4; 
                us = time_chain(s, 10, [&]() { wh_launch_gemm(s, prec, true, g); });
                printf("enc gemm M=%d N=%4d K=%4d + f32 residual, f32 out : %.1f us  %.0f TF/s\n", M, nk.first, nk.second, us, 2.0 * M * nk.first * nk.second / us / 1e6);
                g.R = nullptr; g.C = Cc;
            }
        }
    }
    {   // the LM-head shape through the K-split weight-streaming GEMM (no argmax): what would that structure cost?
        void* Cb = dmalloc((size_t)64 * 51872 * 2 + 4096);
        for (int dm : {4, 2, 1}) {
            SkinnyArgs a; a.W = W; a.bias = bias; a.M = B; a.N = V; a.K = d; a.X = X; a.x_mpad = 64; a.C = Cb; a.ldc = 51872;
            wh_dbg_mt = dm;
            printf("dec_gemm at the LM-head shape N=%d rows/wg=%d : %.2f us\n", V, dm * 16, time_chain(s, 50, [&]() { wh_launch_dec_gemm(s, prec, false, a); }));
        }
        wh_dbg_mt = 0;
    }
    {   // LM head
        SkinnyArgs a; a.W = W; a.X = X; a.x_mpad = 64; a.M = B; a.N = V; a.K = d; a.pos_p = pos; a.n_prompt = 1;
        a.mask_first = (unsigned*)dmalloc(V / 8 + 64); a.mask_base = a.mask_first; a.part_val = (float*)dmalloc((size_t)64 * 4096 * 4); a.part_idx = (int*)dmalloc((size_t)64 * 4096 * 4);
        for (int lmt : {4, 2, 1}) for (int bpc : {1, 2, 4, 8}) { wh_dbg_lm_blocks_per_cu = bpc; wh_dbg_lm_mt = lmt; printf("lm_head M=%d rows/wg=%d blocks/cu=%d : %.2f us\n", B, lmt * 16, bpc, time_chain(s, 50, [&]() { wh_launch_lm_head(s, prec, a); })); }
        wh_dbg_lm_blocks_per_cu = 2; wh_dbg_lm_mt = 4;
    }
    {   // cross attention: distinct K/V planes per "layer" so nothing is cache resident
        const int L = 6; const size_t plane = (size_t)B * S * d;
        void* kv = dmalloc(plane * 2 * L * 2); void* q = dmalloc((size_t)64 * d * 4); void* out = dmalloc((size_t)64 * d * 4);  // slab layouts are padded to 64 rows
        int* tickets = (int*)dmalloc(64 * 4);
        for (int un : {4, 8}) for (int splits : {1, 2, 4, 8, 16, 32}) {
            wh_dbg_cross_unroll = un;
            float* part = (float*)dmalloc((size_t)B * 32 * d * 4); float* ml = (float*)dmalloc((size_t)B * 32 * H * 2 * 4);
            int l = 0;
            double us = time_chain(s, 60, [&]() { wh_launch_dec_cross_attn(s, prec, q, (char*)kv + (size_t)(2 * l) * plane * 2, (char*)kv + (size_t)(2 * l + 1) * plane * 2, part, ml, S, d, H, splits, B, out, 64, true); l = (l + 1) % L; });
            printf("cross_attn B=%d unroll=%d splits=%2d : %.2f us  (%.2f TB/s)\n", B, un, splits, us, 2.0 * S * d * 2 * B / us / 1e6);
        }
        void* qkv = dmalloc((size_t)64 * 3 * d * 4); void* kc = dmalloc((size_t)B * H * 448 * 64 * 2); void* vc = dmalloc((size_t)B * H * 448 * 64 * 2);
        int hp = 100; hipMemcpy(pos, &hp, 4, hipMemcpyHostToDevice);
        printf("self_attn pos=100 : %.2f us\n", time_chain(s, 100, [&]() { wh_launch_dec_self_attn(s, prec, qkv, kc, vc, out, pos, d, H, 448, B, 64); }));
    }
    return 0;
}
