// es3_check.hip — k_dec_cross_attn_es3 (wh_cross_es3.hip: encoder states as fp16 + e4m3 remainder) against a host restatement on random data, then its
// launch time at 2048 clips.  The states are packed on the device with the library's own conversion and decoded exactly on the host.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I whisper-rust-ort_amd/csrc tools/es3_check.hip -o tools/es3_check
#include "../whisper-rust-ort_amd/csrc/wh_cross_es3.hip"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
bool wh_ensure_dyn_lds(const void* k, size_t b) { return hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b) == hipSuccess; }
void wh_set_error(const char* f, ...) { fprintf(stderr, "error: %s\n", f); }
static float e4m3(unsigned char b) {
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v = e == 0 ? ldexpf((float)m / 8.0f, -6) : ldexpf(1.0f + (float)m / 8.0f, e - 7);
    return s ? -v : v;
}
static unsigned rs = 777;
static unsigned rnd() { rs = rs * 1664525u + 1013904223u; return rs >> 8; }
static float frand(float a) { return ((int)(rnd() & 0xffff) - 32768) / 32768.0f * a; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// f32 rows [rows][512] -> E3 rows (the conversion of k_layernorm_es3 without the LayerNorm)
__global__ void k_pack_es3(const float* __restrict__ x, unsigned char* __restrict__ y, long rows) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int c = (threadIdx.x & 63) * 8;
    f16x8 hi; float rm[8];
    for (int e = 0; e < 8; e++) { const float o = x[row * 512 + c + e]; hi[e] = (_Float16)o; rm[e] = (o - (float)hi[e]) * 4096.0f; }
    unsigned char* yr = y + row * E3_ROWB;
    *reinterpret_cast<f16x8*>(yr + c * 2) = hi;
    *reinterpret_cast<wh_u32x2*>(yr + 1024 + c) = wh_u32x2{pack4(rm[0], rm[1], rm[2], rm[3]), pack4(rm[4], rm[5], rm[6], rm[7])};
}

static int check(int B, int S, int n_cus) {
    const int e_rows = S + 20, mpad = ((B + 63) / 64) * 64, H = 8, D = 512;
    const long rows = (long)B * e_rows;
    std::vector<float> Ef((size_t)rows * D), qe((size_t)B * H * D);
    for (auto& v : Ef) v = frand(2.5f);
    for (auto& v : qe) v = frand(0.12f);
    float *dEf, *dq; unsigned char* dE; unsigned char* dout;
    const size_t out_b = (size_t)(H * D / 32) * mpad * 32 * 4;
    CK(hipMalloc(&dEf, Ef.size() * 4)); CK(hipMalloc(&dE, (size_t)rows * E3_ROWB)); CK(hipMalloc(&dq, qe.size() * 4)); CK(hipMalloc(&dout, out_b));
    CK(hipMemcpy(dEf, Ef.data(), Ef.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dq, qe.data(), qe.size() * 4, hipMemcpyHostToDevice)); CK(hipMemset(dout, 0, out_b));
    hipLaunchKernelGGL(k_pack_es3, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, 0, dEf, dE, rows);
    wh_launch_dec_cross_attn_es3(0, dq, dE, dout, S, e_rows, B, mpad, true, n_cus);
    CK(hipDeviceSynchronize());
    std::vector<unsigned char> E((size_t)rows * E3_ROWB), out(out_b);
    CK(hipMemcpy(E.data(), dE, E.size(), hipMemcpyDeviceToHost)); CK(hipMemcpy(out.data(), dout, out_b, hipMemcpyDeviceToHost));
    auto Eval = [&](long r, int d) { _Float16 h; memcpy(&h, &E[(size_t)r * E3_ROWB + 2 * d], 2); return (double)(float)h + (double)e4m3(E[(size_t)r * E3_ROWB + 1024 + d]) / 4096.0; };
    double worst = 0, scale = 0, quant = 0;
    std::vector<double> sc(S), ctx(D), Er((size_t)S * D);
    for (int b = 0; b < B; b++) {
        for (int k = 0; k < S; k++) for (int d = 0; d < D; d++) { Er[(size_t)k * D + d] = Eval((long)b * e_rows + k, d); quant = fmax(quant, fabs(Er[(size_t)k * D + d] - Ef[((size_t)b * e_rows + k) * D + d])); }
        for (int h = 0; h < H; h++) {
            double mx = -1e30;
            for (int k = 0; k < S; k++) {
                double s = 0;
                for (int d = 0; d < D; d++) s += (double)qe[((size_t)b * H + h) * D + d] * Er[(size_t)k * D + d];
                sc[k] = s; mx = fmax(mx, s);
            }
            double l = 0;
            for (int d = 0; d < D; d++) ctx[d] = 0;
            for (int k = 0; k < S; k++) { const double p = exp(sc[k] - mx); l += p; for (int d = 0; d < D; d++) ctx[d] += p * Er[(size_t)k * D + d]; }
            for (int d = 0; d < D; d++) {
                const int kcol = h * D + d;
                const size_t byte = (((size_t)(kcol >> 5) * mpad + b) * 32 + (kcol & 31)) * 4, blk = byte & ~(size_t)127, off = (byte & 127) >> 1;
                _Float16 oh, ol; memcpy(&oh, &out[blk + off], 2); memcpy(&ol, &out[blk + 64 + off], 2);
                const double got = (double)(float)oh + (double)(float)ol, want = ctx[d] / l;
                if (!(fabs(got - want) <= 1e30)) worst = 1e30;
                worst = fmax(worst, fabs(got - want)); scale = fmax(scale, fabs(want));
            }
        }
    }
    const bool ok = worst <= 2e-4 * scale + 1e-6;
    printf("check B %3d S %4d on %3d workgroups: max |ctx - host| %.3e (|ctx| up to %.3f; the states' own rounding: %.2e)  %s\n", B, S, std::min(B, n_cus), worst, scale, quant, ok ? "ok" : "MISMATCH");
    hipFree(dEf); hipFree(dE); hipFree(dq); hipFree(dout);
    return ok ? 0 : 1;
}

int main(int argc, char** argv) {
    if (check(3, 64, 256) | check(5, 1500, 256) | check(7, 1500, 2) | check(4, 333, 3) | check(2, 20, 256)) return 1;
    const int B = argc > 1 ? atoi(argv[1]) : 2048, S = 1500, e_rows = 1520, mpad = B;
    unsigned char *dE, *dout; float* dq;
    CK(hipMalloc(&dE, (size_t)B * e_rows * E3_ROWB)); CK(hipMalloc(&dq, (size_t)B * 4096 * 4)); CK(hipMalloc(&dout, (size_t)128 * mpad * 32 * 4));
    {   // random states through the pack kernel, 64 clips at a time
        std::vector<float> hf((size_t)64 * e_rows * 512);
        for (auto& v : hf) v = frand(2.5f);
        float* df; CK(hipMalloc(&df, hf.size() * 4)); CK(hipMemcpy(df, hf.data(), hf.size() * 4, hipMemcpyHostToDevice));
        for (int b0 = 0; b0 < B; b0 += 64) {
            const long rows = (long)std::min(64, B - b0) * e_rows;
            hipLaunchKernelGGL(k_pack_es3, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, 0, df, dE + (size_t)b0 * e_rows * E3_ROWB, rows);
        }
        CK(hipDeviceSynchronize()); hipFree(df);
    }
    std::vector<float> hq((size_t)B * 4096);
    for (auto& v : hq) v = frand(0.12f);
    CK(hipMemcpy(dq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) wh_launch_dec_cross_attn_es3(0, dq, dE, dout, S, e_rows, B, mpad, true, 256);
    CK(hipEventRecord(e0, 0));
    const int reps = 10;
    for (int i = 0; i < reps; i++) wh_launch_dec_cross_attn_es3(0, dq, dE, dout, S, e_rows, B, mpad, true, 256);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms / reps * 1e3, bytes = (double)B * S * 1536;
    printf("k_dec_cross_attn_es3, %d clips: %.1f us per launch, %.2f TB/s of fp16 + e4m3 encoder states (k_dec_cross_attn_es2: 930-940 us for 4 bytes per element)\n", B, us, bytes / us * 1e-6);
    return 0;
}
