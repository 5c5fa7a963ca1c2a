// gemm8_stamps.hip — where inside a k_gemm8 workgroup do the cycles go?  The library kernel built with WH_GEMM8_STAMPS: lane 0 of every wave writes
// s_memtime at entry / ring issued / loop end / after the closing barrier / operands requested / per epilogue pass (staged, stored), plus its HW_ID and
// XCC_ID, so that the workgroups of one CU can be put in order and the gap between one's last stamp and the next one's first be read.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I whisper-rust-ort_amd/csrc tools/gemm8_stamps.hip -o tools/gemm8_stamps
#define WH_GEMM8_STAMPS 1
#include "../whisper-rust-ort_amd/csrc/wh_gemm8.hip"
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>
bool wh_ensure_dyn_lds(const void* k, size_t b) { return hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b) == hipSuccess; }
void wh_set_error(const char*, ...) {}

static double med(std::vector<double>& v) { if (v.empty()) return 0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

template <typename TO, int BN> void stamp_run(const char* name, const GemmArgs& g) {
    typedef Geo<BN> G;
    const size_t sm = (size_t)G::NSLOT * G::SLOT;
    (void)hipFuncSetAttribute((const void*)k_gemm8<TO, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    const int nwg = ((g.N + BN - 1) / BN) * ((g.M + BM - 1) / BM);
    unsigned long long* d; hipMalloc(&d, (size_t)nwg * 8 * 16 * 8); hipMemset(d, 0, (size_t)nwg * 8 * 16 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_gemm8_stamps), &d, sizeof(d));
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_gemm8<TO, BN>), dim3(nwg), dim3(512), sm, 0, g);
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL((k_gemm8<TO, BN>), dim3(nwg), dim3(512), sm, 0, g);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h((size_t)nwg * 8 * 16);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    const int NP = Geo<BN>::TM / 2;
    const int last = 5 + 2 * (NP - 1);
    // segments per wave
    const char* segn[] = {"entry->ring issued", "main loop", "closing barrier", "operand requests", "pass0 compute+stage", "pass0 read-back+store", "pass1 c+s", "pass1 r+s", "pass2 c+s", "pass2 r+s", "pass3 c+s", "pass3 r+s"};
    std::vector<std::vector<double>> seg(12);
    std::vector<double> whole, loop_wg, epi_wg;
    std::map<unsigned long long, std::vector<std::pair<unsigned long long, unsigned long long>>> per_cu;   // CU key -> (first stamp, last stamp) of its workgroups
    for (int w = 0; w < nwg; w++) {
        unsigned long long t0 = ~0ull, t1 = 0, lend = 0, bend = 0;
        for (int v = 0; v < 8; v++) {
            const unsigned long long* s = &h[((size_t)w * 8 + v) * 16];
            seg[0].push_back((double)(s[12] - s[0])); seg[1].push_back((double)(s[1] - s[12])); seg[2].push_back((double)(s[2] - s[1])); seg[3].push_back((double)(s[3] - s[2]));
            unsigned long long prev = s[3];
            for (int p = 0; p < NP; p++) { seg[4 + 2 * p].push_back((double)(s[4 + 2 * p] - prev)); seg[5 + 2 * p].push_back((double)(s[5 + 2 * p] - s[4 + 2 * p])); prev = s[5 + 2 * p]; }
            t0 = std::min(t0, s[0]); t1 = std::max(t1, s[last]); lend = std::max(lend, s[1]); bend = std::max(bend, s[2]);
        }
        whole.push_back((double)(t1 - t0)); loop_wg.push_back((double)(bend - t0)); epi_wg.push_back((double)(t1 - bend));
        const unsigned long long id = h[((size_t)w * 8) * 16 + 15];
        const unsigned long long key = ((id >> 32) << 32) | (id & 0xff00);   // XCC_ID | {cu, sh, se}
        per_cu[key].push_back({t0, t1});
    }
    std::vector<double> gaps, per_cu_n;
    for (auto& kv : per_cu) {
        auto& v = kv.second; std::sort(v.begin(), v.end());
        per_cu_n.push_back((double)v.size());
        for (size_t i = 1; i < v.size(); i++) gaps.push_back((double)((long long)v[i].first - (long long)v[i - 1].second));
    }
    printf("%s  BN %d: launch %.1f us, %d workgroups on %zu CU keys (median %.0f per key)\n", name, BN, ms * 1e3, nwg, per_cu.size(), med(per_cu_n));
    printf("   per workgroup (cycles, median): first stamp -> last stamp %.0f | entry -> closing barrier passed %.0f | epilogue %.0f | gap to the next workgroup's entry on the same CU %.0f\n",
           med(whole), med(loop_wg), med(epi_wg), med(gaps));
    printf("   per wave (cycles, median):");
    for (int i = 0; i < 4 + 2 * NP; i++) printf(" %s %.0f |", segn[i], med(seg[i]));
    printf("\n");
    hipFree(d);
}

int main() {
    const long M = 256L * 1500;
    bf16 *A, *W; float *R, *bias; void* C;
    hipMalloc(&A, M * 2048 * 2); hipMalloc(&W, 2048L * 2048 * 2); hipMalloc(&C, M * 2048 * 2); hipMalloc(&R, M * 512 * 4); hipMalloc(&bias, 8192);
    std::vector<unsigned short> h(1 << 24);
    unsigned x = 12345; for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3c00 + ((x >> 20) & 0x1ff) + ((x >> 31) << 15)); }
    for (long off = 0; off < M * 2048 * 2; off += (long)h.size() * 2) hipMemcpy((char*)A + off, h.data(), std::min<long>(h.size() * 2, M * 2048 * 2 - off), hipMemcpyHostToDevice);
    hipMemcpy(W, h.data(), 2048L * 2048 * 2, hipMemcpyHostToDevice);
    hipMemset(R, 0, M * 512 * 4); hipMemset(bias, 0, 8192);
    auto mk = [&](int N, int K, bool f32out, bool resid, bool act) {
        GemmArgs g; g.A = A; g.lda = K; g.W = W; g.ldw = K; g.C = f32out ? (void*)R : C; g.ldc = N; g.bias = bias; g.bias_mode = 1; g.act = act; g.M = (int)M; g.N = N; g.K = K;
        if (resid) { g.R = R; g.ldr = N; }
        return g;
    };
    stamp_run<bf16, 256>("QK   N1024 K512 bf16 out", mk(1024, 512, false, false, false));
    stamp_run<float, 256>("O    N512 K512 f32 + residual", mk(512, 512, true, true, false));
    stamp_run<float, 256>("fc2  N512 K2048 f32 + residual", mk(512, 2048, true, true, false));
    stamp_run<bf16, 128>("fc1  N2048 K512 gelu bf16 out", mk(2048, 512, false, false, true));
    stamp_run<bf16, 256>("fc1  N2048 K512 gelu bf16 out", mk(2048, 512, false, false, true));
    return 0;
}
