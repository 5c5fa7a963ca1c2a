// gemm8q_proto.hip — prototype of the next encoder GEMM geometry: 4 waves, one per SIMD, each owning a 128 x 128
// corner of the 256 x 256 tile as sixteen 32 x 32 blocks (v_mfma_f32_32x32x16_bf16, 256 accumulator registers per lane: AGPRs),
// the same LDS-DMA ring as k_gemm8.  Fragment traffic per MFMA cycle halves against the 128 x 64 wave tiles (16 KB per wave and
// k-step of 1,024 MFMA cycles: 64 B per cycle and CU instead of 96).  Checked against a host product, timed against k_gemm8.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 [-mllvm -amdgpu-mfma-vgpr-form=1] -I whisper-rust-ort_amd/csrc tools/gemm8q_proto.hip -o tools/gemm8q_proto
#include "../whisper-rust-ort_amd/csrc/wh_gemm8.hip"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
bool wh_ensure_dyn_lds(const void* k, size_t b) { return hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b) == hipSuccess; }
void wh_set_error(const char*, ...) {}
namespace {
typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int Q_SLOT_A = BM * ROWB, Q_SLOT = 2 * Q_SLOT_A, Q_NSLOT = 4, Q_PER_STAGE = 8, Q_PITCH = 132;
__device__ __forceinline__ int swzq(int row) { return (row >> 2) & 3; }
__device__ __forceinline__ void mma32(f32x16& acc, const bf16x8& a, const bf16x8& b) { acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0); }

template <typename TO, int ABL = 0>   // ABL 4: no epilogue
__global__ __launch_bounds__(256) void k_gemm8q(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, kh = lane >> 5;
    const int nk = g.K / BK;
    const int nbn = (g.N + 255) / 256, total = nbn * ((g.M + BM - 1) / BM);
    int tile = blockIdx.x;
    { const int q = total >> 3, r = total & 7, xcd = tile & 7, idx = tile >> 3; tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx; }
    const int m0 = (tile / nbn) * BM, n0 = (tile % nbn) * 256;
    const bf16* A = (const bf16*)g.A;
    const bf16* W = (const bf16*)g.W;
    const int rl = lane >> 2, ps = lane & 3;
    const bf16* a_src[4];
    const bf16* w_src[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int row = wave * 64 + j * 16 + rl;
        const int m = min(m0 + row, g.M - 1), n = min(n0 + row, g.N - 1);
        a_src[j] = A + (long)(m / g.m_per) * g.a_bs + (long)(m % g.m_per) * g.lda + ((ps ^ swzq(row)) << 3);
        w_src[j] = W + (long)n * g.ldw + ((ps ^ swzq(row)) << 3);
    }
    auto stage = [&](int slot, int kt) {
        char* base = smem + slot * Q_SLOT;
#pragma unroll
        for (int j = 0; j < 4; j++) glds16(a_src[j] + (long)kt * BK, base + (wave * 64 + j * 16) * ROWB);
#pragma unroll
        for (int j = 0; j < 4; j++) glds16(w_src[j] + (long)kt * BK, base + Q_SLOT_A + (wave * 64 + j * 16) * ROWB);
    };
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int v = 0; v < 16; v++) acc[i][j][v] = 0.0f;
    const int ch0 = (kh ^ swzq(r32)) << 4, ch1 = ((2 + kh) ^ swzq(r32)) << 4;
    const int a_row = (wm * 128 + r32) * ROWB, w_row = Q_SLOT_A + (wn * 128 + r32) * ROWB;
    // fragments of the two 16-deep halves of a k-step in separate registers: half 0 of step t+1 is read while the MFMAs of half 1
    // of step t run, half 1 of step t+1 while the MFMAs of its half 0 run — one set of 64 VGPRs, LDS latency always covered
    bf16x8 af[2][4], wf[2][4];   // [half][block]
    auto read_half = [&](int h, int kt) {
        const char* sb = smem + (kt % Q_NSLOT) * Q_SLOT;
        const int ch = h ? ch1 : ch0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            af[h][i] = *reinterpret_cast<const bf16x8*>(sb + a_row + i * 32 * ROWB + ch);
            wf[h][i] = *reinterpret_cast<const bf16x8*>(sb + w_row + i * 32 * ROWB + ch);
        }
    };
    auto mfma_half = [&](int h) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) mma32(acc[i][j], wf[h][j], af[h][i]);   // D rows = n, cols = m
    };
    auto wait_stage = [&](int kt, int cap) {
        const int newer = min(cap, nk - 1 - kt);
        if (newer >= 3) wait_vm<3 * Q_PER_STAGE>();
        else if (newer == 2) wait_vm<2 * Q_PER_STAGE>();
        else if (newer == 1) wait_vm<Q_PER_STAGE>();
        else wait_vm<0>();
    };
#pragma unroll
    for (int t = 0; t < Q_NSLOT; t++)
        if (t < nk) stage(t, t);
    wait_stage(0, Q_NSLOT - 1);
    __builtin_amdgcn_s_barrier();
    read_half(0, 0);
    read_half(1, 0);
    for (int t = 0; t < nk; t++) {
        mfma_half(0);
        if (t + 1 < nk) {
            wait_stage(t + 1, Q_NSLOT - 2);      // in flight here: stages t+1 .. t+NSLOT-1
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's fragment reads of step t have left LDS
            __builtin_amdgcn_s_barrier();       // stage t+1 visible to all; every wave holds step t's fragments in registers
            if (t + Q_NSLOT < nk) stage(t % Q_NSLOT, t + Q_NSLOT);
            read_half(0, t + 1);                // (overwrites the half-0 registers the MFMAs above have consumed)
        }
        mfma_half(1);
        if (t + 1 < nk) read_half(1, t + 1);
    }
    __builtin_amdgcn_s_barrier();
    if (ABL & 4) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int v = 0; v < 16; v++) asm volatile("" :: "v"(acc[i][j][v]));
        return;
    }
    // ---- epilogue: one 32-row block of the wave's 128 x 128 corner per pass through the wave's own 16.5 KiB of the idle ring
    float* stg = reinterpret_cast<float*>(smem) + wave * (32 * Q_PITCH);
    const int nw0 = n0 + wn * 128, mw0 = m0 + wm * 128;
    TO* C = (TO*)g.C;
    const float* R = g.R;
    const bool col_bias = g.bias_mode == 1 && g.bias, col_scale = g.bias_mode == 1 && g.wscale;
    const int c8 = (lane & 15) * 8, r4 = lane >> 4;
    const int n_st = nw0 + c8;
#pragma unroll
    for (int pass = 0; pass < 4; pass++) {
        {
            const int m = mw0 + pass * 32 + r32;
            float bm = 0.0f, wmul = 1.0f;
            if (g.bias_mode == 2 && m < g.M) {
                if (g.bias) bm = g.bias[m];
                if (g.wscale) wmul = g.wscale[m];
            }
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int n = nw0 + q * 8 + kh * 4;
                f32x4 pb = {0, 0, 0, 0}, pw = {1, 1, 1, 1};
                if (n < g.N) {
                    if (col_bias) pb = *reinterpret_cast<const f32x4*>(g.bias + n);
                    if (col_scale) pw = *reinterpret_cast<const f32x4*>(g.wscale + n);
                }
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = acc[pass][q >> 2][(q & 3) * 4 + e] * (pw[e] * wmul) + (pb[e] + bm);
                if (g.act == 1) {
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = gelu_erf(v[e]);
                }
                *reinterpret_cast<f32x4*>(&stg[r32 * Q_PITCH + q * 8 + kh * 4]) = f32x4{v[0], v[1], v[2], v[3]};
            }
        }
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int lr = it * 4 + r4, m = mw0 + pass * 32 + lr;
            if (m < g.M && n_st < g.N) {
                const long mb = m / g.m_per, mi = m % g.m_per;
                f32x4 v0 = *reinterpret_cast<const f32x4*>(&stg[lr * Q_PITCH + c8]);
                f32x4 v1 = *reinterpret_cast<const f32x4*>(&stg[lr * Q_PITCH + c8 + 4]);
                if (R) {
                    const float* rp = R + mb * g.r_bs + mi * g.ldr + n_st;
                    v0 += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(rp));
                    v1 += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(rp + 4));
                }
                TO* cp = C + mb * g.c_bs + mi * g.ldc + n_st;
                if (n_st + 8 <= g.N) store8(cp, v0, v1);
                else store4(cp, v0[0], v0[1], v0[2], v0[3]);
            }
        }
    }
}
}  // namespace

template <typename K> static float time_kernel(K kern, dim3 grid, dim3 block, size_t sm, const GemmArgs& g, int reps) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(kern, grid, block, sm, 0, g);
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(kern, grid, block, sm, 0, g);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    return ms / reps * 1e3f;
}
template <typename TO> static double check_q(int M, int N, int K, bool resid, bool act, int bias_mode) {
    std::vector<unsigned short> ha((size_t)M * K), hw((size_t)N * K);
    std::vector<float> hr((size_t)M * N), hb(std::max(M, N)), hs(std::max(M, N));
    unsigned x = 777u + M + N * 3 + K * 7;
    auto rnd = [&]() { x = x * 1664525u + 1013904223u; return ((int)(x >> 9) % 2001 - 1000) * 1e-3f; };
    auto tobf = [](float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); };
    auto frombf = [](unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; };
    for (auto& v : ha) v = tobf(rnd());
    for (auto& v : hw) v = tobf(rnd());
    for (auto& v : hr) v = rnd();
    for (auto& v : hb) v = rnd();
    for (auto& v : hs) v = 0.5f + 0.25f * rnd();
    bf16 *A, *W; float *R, *bias, *ws; TO* C;
    hipMalloc(&A, ha.size() * 2); hipMalloc(&W, hw.size() * 2); hipMalloc(&R, hr.size() * 4); hipMalloc(&bias, hb.size() * 4); hipMalloc(&ws, hs.size() * 4);
    hipMalloc(&C, (size_t)M * N * sizeof(TO)); hipMemset(C, 0xff, (size_t)M * N * sizeof(TO));
    hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice); hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(R, hr.data(), hr.size() * 4, hipMemcpyHostToDevice); hipMemcpy(bias, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(ws, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
    GemmArgs g; g.A = A; g.lda = K; g.W = W; g.ldw = K; g.C = C; g.ldc = N; g.bias = bias; g.wscale = ws; g.bias_mode = bias_mode; g.act = act; g.M = M; g.N = N; g.K = K;
    if (resid) { g.R = R; g.ldr = N; }
    const size_t sm = (size_t)Q_NSLOT * Q_SLOT;
    (void)hipFuncSetAttribute((const void*)k_gemm8q<TO, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    hipLaunchKernelGGL((k_gemm8q<TO, 0>), dim3(((N + 255) / 256) * ((M + BM - 1) / BM)), dim3(256), sm, 0, g);
    std::vector<TO> hc((size_t)M * N);
    hipMemcpy(hc.data(), C, hc.size() * sizeof(TO), hipMemcpyDeviceToHost);
    double worst = 0;
    for (int m = 0; m < M; m++)
        for (int n = 0; n < N; n++) {
            double acc = 0;
            for (int k = 0; k < K; k++) acc += (double)frombf(ha[(size_t)m * K + k]) * frombf(hw[(size_t)n * K + k]);
            const int bi = bias_mode == 2 ? m : n;
            double v = acc * hs[bi] + hb[bi];
            if (act) v = 0.5 * v * (1.0 + erf(v * 0.7071067811865476));
            if (resid) v += hr[(size_t)m * N + n];
            float got;
            if constexpr (sizeof(TO) == 4) got = hc[(size_t)m * N + n]; else { unsigned short b; memcpy(&b, &hc[(size_t)m * N + n], 2); got = frombf(b); }
            const double tol = sizeof(TO) == 4 ? 0 : 0.004 * fabs(v);
            worst = std::max(worst, fabs(got - v) - tol);
        }
    hipFree(A); hipFree(W); hipFree(R); hipFree(bias); hipFree(ws); hipFree(C);
    return worst;
}
int main() {
    struct Case { int M, N, K; bool resid, act; int bm; } cases[] = {
        {512, 256, 128, false, false, 1}, {300, 384, 96, true, false, 1}, {777, 1500, 64, false, false, 2}, {256, 512, 512, true, true, 1}, {1000, 132, 32, false, true, 1}};
    int bad = 0;
    for (auto& c : cases) {
        const double e1 = check_q<float>(c.M, c.N, c.K, c.resid, c.act, c.bm), e2 = check_q<bf16>(c.M, c.N, c.K, c.resid, c.act, c.bm);
        const bool ok = e1 < 2e-4 && e2 < 2e-4;
        printf("check M%4d N%4d K%3d res%d act%d bias%d: excess error f32 %.1e bf16 %.1e  %s\n", c.M, c.N, c.K, c.resid, c.act, c.bm, e1, e2, ok ? "ok" : "MISMATCH");
        bad += !ok;
    }
    if (bad) return 1;
    const long M = 256L * 1500;
    struct Shape { const char* name; int N, K; bool f32out, resid, act; } shapes[] = {
        {"QK   N1024 K512 ", 1024, 512, false, false, false}, {"fc1  N2048 K512 gelu", 2048, 512, false, false, true},
        {"fc2  N512 K2048 f32+res", 512, 2048, true, true, false}, {"O    N512 K512 f32+res", 512, 512, true, true, false}};
    bf16 *A, *W; float *R, *bias; void* C;
    hipMalloc(&A, M * 2048 * 2); hipMalloc(&W, 2048L * 2048 * 2); hipMalloc(&C, M * 2048 * 2); hipMalloc(&R, M * 512 * 4); hipMalloc(&bias, 8192);
    std::vector<unsigned short> h(1 << 24);
    unsigned x = 12345; for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3c00 + ((x >> 20) & 0x1ff) + ((x >> 31) << 15)); }
    for (long off = 0; off < M * 2048 * 2; off += (long)h.size() * 2) hipMemcpy((char*)A + off, h.data(), std::min<long>(h.size() * 2, M * 2048 * 2 - off), hipMemcpyHostToDevice);
    hipMemcpy(W, h.data(), 2048L * 2048 * 2, hipMemcpyHostToDevice);
    hipMemset(R, 0, M * 512 * 4); hipMemset(bias, 0, 8192);
    const size_t smq = (size_t)Q_NSLOT * Q_SLOT, sm8 = (size_t)Geo<256>::NSLOT * Geo<256>::SLOT;
    for (auto& sh : shapes) {
        GemmArgs g; g.A = A; g.lda = sh.K; g.W = W; g.ldw = sh.K; g.C = sh.f32out ? (void*)R : C; g.ldc = sh.N; g.bias = bias; g.bias_mode = 1;
        g.act = sh.act; g.M = (int)M; g.N = sh.N; g.K = sh.K;
        if (sh.resid) { g.R = R; g.ldr = sh.N; }
        const double gf = 2.0 * M * sh.N * sh.K * 1e-9;
        dim3 grid(((sh.N + 255) / 256) * ((M + BM - 1) / BM));
        float tq, tq4, t8, t84;
        if (sh.f32out) {
            (void)hipFuncSetAttribute((const void*)k_gemm8q<float, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smq);
            (void)hipFuncSetAttribute((const void*)k_gemm8q<float, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smq);
            (void)hipFuncSetAttribute((const void*)k_gemm8<float, 256, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm8);
            (void)hipFuncSetAttribute((const void*)k_gemm8<float, 256, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm8);
            tq = time_kernel(k_gemm8q<float, 0>, grid, dim3(256), smq, g, 5); tq4 = time_kernel(k_gemm8q<float, 4>, grid, dim3(256), smq, g, 5);
            t8 = time_kernel(k_gemm8<float, 256, 0>, grid, dim3(512), sm8, g, 5); t84 = time_kernel(k_gemm8<float, 256, 4>, grid, dim3(512), sm8, g, 5);
        } else {
            (void)hipFuncSetAttribute((const void*)k_gemm8q<bf16, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smq);
            (void)hipFuncSetAttribute((const void*)k_gemm8q<bf16, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smq);
            (void)hipFuncSetAttribute((const void*)k_gemm8<bf16, 256, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm8);
            (void)hipFuncSetAttribute((const void*)k_gemm8<bf16, 256, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm8);
            tq = time_kernel(k_gemm8q<bf16, 0>, grid, dim3(256), smq, g, 5); tq4 = time_kernel(k_gemm8q<bf16, 4>, grid, dim3(256), smq, g, 5);
            t8 = time_kernel(k_gemm8<bf16, 256, 0>, grid, dim3(512), sm8, g, 5); t84 = time_kernel(k_gemm8<bf16, 256, 4>, grid, dim3(512), sm8, g, 5);
        }
        printf("%-26s 4-wave 128x128 prototype: full %7.1f us (%5.0f TF/s), main loop %7.1f | k_gemm8<256>: full %7.1f us (%5.0f TF/s), main loop %7.1f\n", sh.name, tq, gf / tq * 1e3, tq4, t8,
               gf / t8 * 1e3, t84);
    }
    return 0;
}
