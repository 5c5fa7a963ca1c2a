// es_bench.hip — k_dec_cross_attn_es alone: B clips of random encoder states and expanded queries, launch time and stream rate.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I whisper-rust-ort_amd/csrc tools/es_bench.hip -o tools/es_bench
//   WH_ES_NSTAGE=2|3|4 WH_CROSS_NT=0|1 tools/es_bench [clips] [reps]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "wh_cross_es.hip"

bool wh_ensure_dyn_lds(const void* kernel, size_t bytes) {
    return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 1024, reps = argc > 2 ? atoi(argv[2]) : 30, S = 1500, d = 512, H = 8;
    const int mpad = (B + 63) / 64 * 64;
    std::vector<unsigned short> hE((size_t)B * S * d);
    unsigned x = 12345;
    for (auto& v : hE) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3c00 + ((x >> 16) & 0x3ff)) ^ (unsigned short)((x >> 3) & 0x8000); }   // +-[0.0078, 0.03)
    std::vector<float> hq((size_t)B * H * d);
    for (auto& v : hq) { x = x * 1664525u + 1013904223u; v = ((int)(x >> 8) % 2001 - 1000) * 1e-3f; }
    void *E, *out; float* q;
    hipMalloc(&E, hE.size() * 2); hipMalloc((void**)&q, hq.size() * 4); hipMalloc(&out, (size_t)mpad * H * d * 2);
    hipMemcpy(E, hE.data(), hE.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(q, hq.data(), hq.size() * 4, hipMemcpyHostToDevice);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipMalloc((void**)&wh_es_bench_dbg, 64 * 8);
    hipMemset(wh_es_bench_dbg, 0, 64 * 8);
    for (int nt = 1; nt >= 0; nt--) {
        for (int i = 0; i < 3; i++) wh_launch_dec_cross_attn_es(s, WH_PREC_BF16, q, E, out, S, S, B, mpad, nt, 256);
        hipEventRecord(e0, s);
        for (int i = 0; i < reps; i++) wh_launch_dec_cross_attn_es(s, WH_PREC_BF16, q, E, out, S, S, B, mpad, nt, 256);
        hipEventRecord(e1, s);
        hipStreamSynchronize(s);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / reps, bytes = (double)B * S * d * 2;
        printf("clips %d nt %d: %.1f us per launch, %.2f TB/s (%s)\n", B, nt, us, bytes / us * 1e-6, hipGetErrorString(hipGetLastError()));
    }
    if ((getenv("WH_ES_ABL") && atoi(getenv("WH_ES_ABL")) == 2) || (getenv("WH_ES_ABL2") && (atoi(getenv("WH_ES_ABL2")) & 2))) {
        unsigned long long h[64];
        hipMemcpy(h, wh_es_bench_dbg, sizeof h, hipMemcpyDeviceToHost);
        const double n = (double)h[5];
        printf("consumer wave 0 of workgroup 0, cycles per tile over %.0f tiles: wait+barrier %.0f | LDS reads landed %.0f | softmax %.0f | score MFMA + exchange %.0f | transpose + output MFMA %.0f\n",
               n, h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n);
        const double nl = (double)h[11];
        printf("loader wave: vmcnt wait %.0f | barrier %.0f | issue %.0f cycles per tile\n", h[8] / nl, h[9] / nl, h[10] / nl);
    }
    return 0;
}
