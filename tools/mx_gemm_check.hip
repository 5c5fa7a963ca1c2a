// mx_gemm_check.hip — k_layernorm_mx + k_gemm8_mx against a host restatement (f64 accumulation over the dequantised operands).
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -w -I whisper-rust-ort_amd/csrc tools/mx_gemm_check.hip -o tools/mx_gemm_check
#include "../whisper-rust-ort_amd/csrc/wh_gemm8_mx.hip"
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>
bool wh_ensure_dyn_lds(const void* k, size_t b) { return hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b) == hipSuccess; }
void wh_set_error(const char*, ...) {}
static float e4m3(unsigned char c) { int e = (c >> 3) & 15, m = c & 7; float v = e == 0 ? m * ldexpf(1.0f, -9) : (8 + m) * ldexpf(1.0f, e - 10); return (c & 0x80) ? -v : v; }
static unsigned char to_e4m3(float x) {   // RNE, saturating: brute force over the 127 non-negative codes
    float a = fminf(fabsf(x), 448.0f); int best = 0; float bd = 1e30f;
    for (int c = 0; c < 0x7F; c++) { float d = fabsf(e4m3((unsigned char)c) - a); if (d < bd || (d == bd && !(c & 1))) { bd = d; best = c; } }
    return (unsigned char)(best | (x < 0 ? 0x80 : 0));
}
static float deq(const unsigned char* codes, const unsigned char* exps, long row, int k, int K) {
    const int blk = k >> 5, nk = wh_mx_nkp(K);
    return e4m3(codes[row * K + k]) * ldexpf(1.0f, (int)exps[(row * 4 + (blk & 3)) * nk + (blk >> 2)] - 127);
}
int main() {
    int bad = 0;
    // K = 1280 and 5120: whisper-large-v3's d_model and ffn (10 and 40 K-steps: a padded exponent row, resp. three exponent segments);
    // 5120 has no LayerNorm (it is the GELU output's width): its activations are quantised to MX on the host
    for (int K : {256, 512, 2048, 1280, 5120}) {
        const int M = 700, N = K >= 2048 ? 256 : 640;   // M tail (700 = 2 * 256 + 188), N a multiple of 128
        const bool has_ln = K != 5120;
        std::vector<float> X((size_t)M * K), lw(K), lb(K), bias(N), ws(N);
        unsigned s = 99 + K;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
        for (auto& v : X) v = 4.0f * rnd() * (1.0f + 3.0f * (rnd() > 0.45f));
        for (int k = 0; k < K; k++) { lw[k] = 1.0f + 0.2f * rnd(); lb[k] = 0.2f * rnd(); }
        std::vector<unsigned char> W((size_t)N * K);
        for (auto& c : W) { float v = 200.0f * rnd(); c = to_e4m3(v); }
        for (int n = 0; n < N; n++) { bias[n] = rnd(); ws[n] = 0.002f * (1.0f + rnd()); }
        float *dX, *dlw, *dlb, *dbias, *dws; unsigned char *dA8, *dAs, *dW, *dC8, *dCs; bf16* dC;
        hipMalloc(&dX, X.size() * 4); hipMalloc(&dlw, K * 4); hipMalloc(&dlb, K * 4); hipMalloc(&dbias, N * 4); hipMalloc(&dws, N * 4);
        const size_t as_bytes = (size_t)M * 4 * wh_mx_nkp(K), cs_bytes = (size_t)M * 4 * wh_mx_nkp(N);
        hipMalloc(&dA8, (size_t)M * K); hipMalloc(&dAs, as_bytes); hipMalloc(&dW, W.size()); hipMalloc(&dC, (size_t)M * N * 2);
        hipMalloc(&dC8, (size_t)M * N); hipMalloc(&dCs, cs_bytes);
        hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dlw, lw.data(), K * 4, hipMemcpyHostToDevice); hipMemcpy(dlb, lb.data(), K * 4, hipMemcpyHostToDevice);
        hipMemcpy(dbias, bias.data(), N * 4, hipMemcpyHostToDevice); hipMemcpy(dws, ws.data(), N * 4, hipMemcpyHostToDevice); hipMemcpy(dW, W.data(), W.size(), hipMemcpyHostToDevice);
        if (has_ln) wh_launch_layernorm_mx(0, dX, dlw, dlb, dA8, dAs, M, K);
        else {   // MX quantisation of X itself on the host, by the rule of the kernels
            std::vector<unsigned char> hA((size_t)M * K), hS(as_bytes, 127);
            for (long r = 0; r < M; r++)
                for (int b0 = 0; b0 < K; b0 += 32) {
                    float am = 0;
                    for (int i = 0; i < 32; i++) am = fmaxf(am, fabsf(X[r * K + b0 + i]));
                    unsigned ab; memcpy(&ab, &am, 4);
                    int eb = (int)((ab >> 23) & 0xFF) - 8 + (int)((ab & 0x7FFFFF) > 0x600000); if (eb < 0) eb = 0;
                    const int blk = b0 >> 5;
                    hS[(r * 4 + (blk & 3)) * wh_mx_nkp(K) + (blk >> 2)] = (unsigned char)eb;
                    for (int i = 0; i < 32; i++) hA[r * K + b0 + i] = to_e4m3(X[r * K + b0 + i] * ldexpf(1.0f, 127 - eb));
                }
            hipMemcpy(dA8, hA.data(), hA.size(), hipMemcpyHostToDevice); hipMemcpy(dAs, hS.data(), hS.size(), hipMemcpyHostToDevice);
        }
        GemmArgs g; g.A = dA8; g.lda = K; g.a_sc = dAs; g.W = dW; g.ldw = K; g.C = dC; g.ldc = N; g.bias = dbias; g.bias_mode = 1; g.wscale = dws; g.M = M; g.N = N; g.K = K;
        if (!wh_gemm8_mx_applicable(g)) { printf("K %d not applicable\n", K); bad = 1; continue; }
        wh_launch_gemm8_mx(0, 0, g);
        GemmArgs g2 = g; g2.C = dC8; g2.c_sc = dCs; g2.act = 1;
        wh_launch_gemm8_mx(0, 2, g2);
        hipDeviceSynchronize();
        std::vector<unsigned char> A8((size_t)M * K), As(as_bytes), C8((size_t)M * N), Cs(cs_bytes);
        std::vector<unsigned short> C((size_t)M * N);
        hipMemcpy(A8.data(), dA8, A8.size(), hipMemcpyDeviceToHost); hipMemcpy(As.data(), dAs, As.size(), hipMemcpyDeviceToHost);
        hipMemcpy(C.data(), dC, C.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(C8.data(), dC8, C8.size(), hipMemcpyDeviceToHost); hipMemcpy(Cs.data(), dCs, Cs.size(), hipMemcpyDeviceToHost);
        // (1) LayerNorm + MX quantisation vs host
        double ln_worst = 0; long exp_mismatch = 0, code_mismatch = 0;
        for (long r = 0; r < (has_ln ? M : 0); r++) {
            double mean = 0, var = 0;
            for (int k = 0; k < K; k++) mean += X[r * K + k];
            mean /= K;
            for (int k = 0; k < K; k++) { double t = X[r * K + k] - mean; var += t * t; }
            const double rstd = 1.0 / sqrt(var / K + 1e-5);
            for (int b0 = 0; b0 < K; b0 += 32) {
                float y[32], am = 0;
                for (int i = 0; i < 32; i++) { y[i] = (float)((X[r * K + b0 + i] - mean) * rstd) * lw[b0 + i] + lb[b0 + i]; am = fmaxf(am, fabsf(y[i])); }
                unsigned ab; memcpy(&ab, &am, 4);
                int eb = (int)((ab >> 23) & 0xFF) - 8 + (int)((ab & 0x7FFFFF) > 0x600000); if (eb < 0) eb = 0;
                const int blk = b0 >> 5, nk = wh_mx_nkp(K);
                if (As[(r * 4 + (blk & 3)) * nk + (blk >> 2)] != eb) exp_mismatch++;
                for (int i = 0; i < 32; i++) {
                    if (A8[r * K + b0 + i] != to_e4m3(y[i] * ldexpf(1.0f, 127 - eb))) code_mismatch++;
                    ln_worst = fmax(ln_worst, fabs(deq(A8.data(), As.data(), r, b0 + i, K) - y[i]) / fmax(1e-3, am));
                }
            }
        }
        // (2) GEMM (bf16 out) vs host sum over the DEVICE's quantised activations
        double g_worst = 0, g_scale = 0;
        for (long r = 0; r < M; r += 7)
            for (int n = 0; n < N; n += 5) {
                double acc = 0;
                for (int k = 0; k < K; k++) acc += (double)deq(A8.data(), As.data(), r, k, K) * (double)e4m3(W[(size_t)n * K + k]);
                const double ref = acc * ws[n] + bias[n];
                unsigned u = (unsigned)C[r * N + n] << 16; float got; memcpy(&got, &u, 4);
                g_worst = fmax(g_worst, fabs(got - ref)); g_scale = fmax(g_scale, fabs(ref));
                // (3) MX output of GELU(ref)
                const double ge = 0.5 * ref * (1.0 + erf(ref * 0.70710678118654752440));
                const int blk = n >> 5, nkN = wh_mx_nkp(N);
                const float got8 = e4m3(C8[r * N + n]) * ldexpf(1.0f, (int)Cs[(r * 4 + (blk & 3)) * nkN + (blk >> 2)] - 127);
                (void)ge; (void)got8;
            }
        double mx_worst = 0;
        for (long r = 0; r < M; r += 7)
            for (int b0 = 0; b0 < N; b0 += 32) {
                float ref[32], am = 0;
                for (int i = 0; i < 32; i++) {
                    double acc = 0;
                    for (int k = 0; k < K; k++) acc += (double)deq(A8.data(), As.data(), r, k, K) * (double)e4m3(W[(size_t)(b0 + i) * K + k]);
                    const double v = acc * ws[b0 + i] + bias[b0 + i];
                    ref[i] = (float)(0.5 * v * (1.0 + erf(v * 0.70710678118654752440))); am = fmaxf(am, fabsf(ref[i]));
                }
                const int blk = b0 >> 5, nkN = wh_mx_nkp(N);
                for (int i = 0; i < 32; i++) {
                    const float got8 = e4m3(C8[r * N + b0 + i]) * ldexpf(1.0f, (int)Cs[(r * 4 + (blk & 3)) * nkN + (blk >> 2)] - 127);
                    mx_worst = fmax(mx_worst, fabs(got8 - ref[i]) / fmax(1e-3, am));
                }
            }
        printf("K %4d: LN-mx exponent mismatches %ld, code mismatches %ld of %ld, worst |deq - y| / block max %.4f | GEMM bf16 out: worst abs err %.4f of max |ref| %.2f | MX out: worst err / block max %.4f\n",
               K, exp_mismatch, code_mismatch, (long)M * K, ln_worst, g_worst, g_scale, mx_worst);
        if (exp_mismatch > M / 50 || ln_worst > 0.07 || g_worst > 0.02 * g_scale || mx_worst > 0.08) bad = 1;
    }
    printf(bad ? "MISMATCH\n" : "ok\n");
    return bad;
}
