// two_stream.cpp — do two contexts (HIP streams) of the library overlap when driven from two host threads?
// Times S contexts x (64/S) clips, whisper-base bf16, 128 new tokens, PCM resident in HBM.
// Build: hipcc --offload-arch=gfx950 -O2 -std=c++17 -pthread tools/two_stream.cpp -Lwhisper-rust-ort_amd -lwhisper_hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include "../include/whisper_hip.h"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const int S = argc > 1 ? atoi(argv[1]) : 2, total = 64, per = total / S, NEW = 128;
    const int prec = argc > 2 ? atoi(argv[2]) : WH_PREC_BF16;
    wh_model* m = nullptr;
    if (wh_model_load("synthetic:base:1234", 0, prec, &m)) { fprintf(stderr, "load: %s\n", wh_last_error(nullptr)); return 1; }
    std::vector<wh_ctx*> ctx(S);
    for (auto& c : ctx) if (wh_ctx_create(m, per, &c)) { fprintf(stderr, "ctx: %s\n", wh_last_error(nullptr)); return 1; }
    float* d_pcm; hipMalloc(&d_pcm, (size_t)total * 480000 * 4);
    std::vector<float> h((size_t)480000);
    for (int i = 0; i < total; i++) {
        unsigned s = 1234u + i;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) % 2001 - 1000) * 2e-4f; }
        hipMemcpy(d_pcm + (size_t)i * 480000, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    }
    const int64_t prompt[4] = {50258, 50259, 50359, 50363}, sup[1] = {50257};
    wh_decode_params p{}; p.prompt = prompt; p.n_prompt = 4; p.max_new_tokens = NEW; p.eot = 50257; p.suppress = sup; p.n_suppress = 1;
    std::vector<std::vector<int64_t>> toks(S, std::vector<int64_t>((size_t)per * (4 + NEW)));
    std::vector<std::vector<size_t>> nt(S, std::vector<size_t>(per));
    auto run_all = [&]() {
        std::vector<std::thread> th;
        for (int s = 0; s < S; s++)
            th.emplace_back([&, s]() {
                hipSetDevice(0);
                if (wh_transcribe_batch_device(ctx[s], d_pcm + (size_t)s * per * 480000, per, &p, toks[s].data(), nt[s].data()))
                    fprintf(stderr, "run: %s\n", wh_last_error(ctx[s]));
            });
        for (auto& t : th) t.join();
    };
    run_all(); run_all();
    double best = 1e9;
    for (int r = 0; r < 5; r++) { double t0 = now(); run_all(); best = std::min(best, now() - t0); }
    printf("streams=%d clips/stream=%d prec=%d : %.2f ms per 64 clips  (%.0f x real time)\n", S, per, prec, best * 1e3, 64 * 30.0 / best);
    return 0;
}
