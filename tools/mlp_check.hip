// mlp_check.hip — k_enc_mlp (wh_mlp.hip) against a host restatement in double precision, then its launch time on the encoder's shape.
//   h = bf16( gelu( rstd_m (x_m . W1_n - mean_m s_n) + c_n ) ),  out = h . W2^T + b2 + R  (f32, in place),  xb = bf16(out - shift_m),
//   partial {sum, sum of squares} of (out - shift_m) per (64-column group, row)
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I whisper-rust-ort_amd/csrc tools/mlp_check.hip -o tools/mlp_check
#include "../whisper-rust-ort_amd/csrc/wh_mlp.hip"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
bool wh_ensure_dyn_lds(const void* k, size_t b) { return hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b) == hipSuccess; }
void wh_set_error(const char* f, ...) { fprintf(stderr, "error: %s\n", f); }
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
static unsigned rng_state = 99991;
static float rnd() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xffff) / 32768.0f - 1.0f; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static int check(int M, int F) {
    const int d = 512;
    std::vector<unsigned short> X((size_t)M * d), W1((size_t)F * d), W2((size_t)d * F);
    std::vector<float> stat((size_t)M * 2), s1(F), c1(F), b2(d), R((size_t)M * d), shift(M);
    for (auto& v : X) v = f2bf(rnd());
    for (auto& v : W1) v = f2bf(rnd() * 0.06f);
    for (auto& v : W2) v = f2bf(rnd() * 0.04f);
    for (int m = 0; m < M; m++) { stat[2 * m] = rnd() * 0.1f; stat[2 * m + 1] = 1.0f + rnd() * 0.3f; shift[m] = rnd(); }
    for (int n = 0; n < F; n++) { s1[n] = rnd(); c1[n] = rnd() * 0.5f; }
    for (auto& v : b2) v = rnd() * 0.1f;
    for (auto& v : R) v = rnd() * 2.0f;
    void *dX, *dW1, *dW2, *dxb; float *dstat, *ds1, *dc1, *db2, *dR, *dshift, *dpart;
    CK(hipMalloc(&dX, X.size() * 2)); CK(hipMalloc(&dW1, W1.size() * 2)); CK(hipMalloc(&dW2, W2.size() * 2)); CK(hipMalloc(&dxb, X.size() * 2));
    CK(hipMalloc(&dstat, stat.size() * 4)); CK(hipMalloc(&ds1, F * 4)); CK(hipMalloc(&dc1, F * 4)); CK(hipMalloc(&db2, d * 4)); CK(hipMalloc(&dR, R.size() * 4));
    CK(hipMalloc(&dshift, M * 4)); CK(hipMalloc(&dpart, (size_t)8 * M * 2 * 4));
    CK(hipMemcpy(dX, X.data(), X.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW1, W1.data(), W1.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW2, W2.data(), W2.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dstat, stat.data(), stat.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(ds1, s1.data(), F * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dc1, c1.data(), F * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db2, b2.data(), d * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dR, R.data(), R.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dshift, shift.data(), M * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dxb, 0, X.size() * 2)); CK(hipMemset(dpart, 0, (size_t)8 * M * 2 * 4));
    MlpArgs a; a.X = dX; a.ldx = d; a.ln_stat = dstat; a.W1 = dW1; a.s1 = ds1; a.c1 = dc1; a.W2 = dW2; a.b2 = db2; a.Xres = dR; a.ldr = d; a.xb_out = dxb;
    a.stats_out = dpart; a.stats_rows = M; a.row_shift = dshift; a.M = M; a.d = d; a.F = F;
    if (wh_launch_enc_mlp(0, a) != WH_OK) return 1;
    CK(hipDeviceSynchronize());
    std::vector<float> out((size_t)M * d), part((size_t)8 * M * 2); std::vector<unsigned short> xb((size_t)M * d);
    CK(hipMemcpy(out.data(), dR, out.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(xb.data(), dxb, xb.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(part.data(), dpart, part.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, worst_xb = 0, worst_part = 0, scale = 0;
    std::vector<float> h(F);
    const int step = M > 2000 ? 37 : 1;   // every row of the small cases, a sample of the large one
    for (int m = 0; m < M; m += step) {
        for (int n = 0; n < F; n++) {
            double acc = 0;
            for (int k = 0; k < d; k++) acc += (double)bf2f(X[(size_t)m * d + k]) * bf2f(W1[(size_t)n * d + k]);
            const double v = stat[2 * m + 1] * (acc - (double)stat[2 * m] * s1[n]) + c1[n];
            h[n] = bf2f(f2bf((float)(0.5 * v * (1.0 + erf(v * 0.70710678118654752440)))));
        }
        double g1[8] = {0}, g2[8] = {0};
        for (int n = 0; n < d; n++) {
            double acc = 0;
            for (int k = 0; k < F; k++) acc += (double)h[k] * bf2f(W2[(size_t)n * F + k]);
            const double v = acc + b2[n] + R[(size_t)m * d + n];
            scale = fmax(scale, fabs(v));
            worst = fmax(worst, fabs(v - out[(size_t)m * d + n]));
            const double vs = out[(size_t)m * d + n] - shift[m];   // (of the device's own f32 row: the copy and the sums are checked on their own)
            worst_xb = fmax(worst_xb, fabs(bf2f(xb[(size_t)m * d + n]) - vs) / fmax(1.0, fabs(vs)));
            g1[n >> 6] += vs; g2[n >> 6] += vs * vs;
        }
        for (int g = 0; g < 8; g++) {
            worst_part = fmax(worst_part, fabs(part[((size_t)g * M + m) * 2] - g1[g]) / fmax(1.0, fabs(g1[g])));
            worst_part = fmax(worst_part, fabs(part[((size_t)g * M + m) * 2 + 1] - g2[g]) / fmax(1.0, fabs(g2[g])));
        }
    }
    const bool ok = worst <= 4e-3 * scale && worst_xb <= 4.0e-3 && worst_part <= 1e-4;
    printf("check M %6d F %4d: max |out - host| %.3e (outputs up to %.2f), bf16 copy rel %.2e, partial sums rel %.2e  %s\n", M, F, worst, scale, worst_xb, worst_part, ok ? "ok" : "MISMATCH");
    hipFree(dX); hipFree(dW1); hipFree(dW2); hipFree(dxb); hipFree(dstat); hipFree(ds1); hipFree(dc1); hipFree(db2); hipFree(dR); hipFree(dshift); hipFree(dpart);
    return ok ? 0 : 1;
}

int main(int argc, char** argv) {
    if (check(128, 128) | check(300, 512) | check(1500, 2048) | check(13500, 2048)) return 1;
    // launch time on the encoder's shape: 256 clips x 1500 rows, d 512, F 2048
    const int M = argc > 1 ? atoi(argv[1]) * 1500 : 256 * 1500, d = 512, F = 2048;
    void *dX, *dW1, *dW2; float *dstat, *ds1, *dc1, *db2, *dR, *dshift, *dpart;
    CK(hipMalloc(&dX, (size_t)M * d * 2)); CK(hipMalloc(&dW1, (size_t)F * d * 2)); CK(hipMalloc(&dW2, (size_t)F * d * 2));
    CK(hipMalloc(&dstat, (size_t)M * 8)); CK(hipMalloc(&ds1, F * 4)); CK(hipMalloc(&dc1, F * 4)); CK(hipMalloc(&db2, d * 4)); CK(hipMalloc(&dR, (size_t)M * d * 4));
    CK(hipMalloc(&dshift, (size_t)M * 4)); CK(hipMalloc(&dpart, (size_t)8 * M * 8));
    std::vector<unsigned short> hb(1 << 22);
    for (auto& v : hb) v = f2bf(rnd() * 0.05f);
    for (size_t off = 0; off < (size_t)M * d * 2; off += hb.size() * 2) CK(hipMemcpy((char*)dX + off, hb.data(), std::min(hb.size() * 2, (size_t)M * d * 2 - off), hipMemcpyHostToDevice));
    CK(hipMemcpy(dW1, hb.data(), (size_t)F * d * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW2, hb.data() + 12345, (size_t)F * d * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dstat, 0, (size_t)M * 8)); CK(hipMemset(ds1, 0, F * 4)); CK(hipMemset(dc1, 0, F * 4)); CK(hipMemset(db2, 0, d * 4)); CK(hipMemset(dR, 0, (size_t)M * d * 4)); CK(hipMemset(dshift, 0, (size_t)M * 4));
    MlpArgs a; a.X = dX; a.ldx = d; a.ln_stat = dstat; a.W1 = dW1; a.s1 = ds1; a.c1 = dc1; a.W2 = dW2; a.b2 = db2; a.Xres = dR; a.ldr = d; a.xb_out = dX;
    a.stats_out = dpart; a.stats_rows = M; a.row_shift = dshift; a.M = M; a.d = d; a.F = F;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; i++) if (wh_launch_enc_mlp(0, a) != WH_OK) return 1;
    CK(hipEventRecord(e0, 0));
    const int reps = 5;
    for (int i = 0; i < reps; i++) wh_launch_enc_mlp(0, a);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms / reps * 1e3, tf = 4.0 * M * d * F / us * 1e-6;
    printf("k_enc_mlp, %d rows: %.1f us per launch (%.0f TFLOP/s); the two k_gemm8 launches it replaces: tools/gemm8_ablate (fc1 BN 128 + fc2 BN 256)\n", M, us, tf);
    auto time_variant = [&](auto kern, const char* what) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        const dim3 grid((unsigned)((M + FM - 1) / FM));
        hipLaunchKernelGGL(kern, grid, dim3(512), LDS_BYTES, 0, a);
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < reps; i++) hipLaunchKernelGGL(kern, grid, dim3(512), LDS_BYTES, 0, a);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float t = 0; (void)hipEventElapsedTime(&t, e0, e1);
        printf("   %s: %.1f us\n", what, t / reps * 1e3);
    };
    time_variant(k_enc_mlp<0>, "the library's form again");
    time_variant(k_enc_mlp<1>, "no epilogue");
    time_variant(k_enc_mlp<2>, "no GELU / fold arithmetic");
    time_variant(k_enc_mlp<3>, "neither");
    time_variant(k_enc_mlp<4>, "bf16 copy as 8-byte stores (the first form of the epilogue)");
    time_variant(k_enc_mlp<0>, "the library's form, third reading");
    return 0;
}
