// gemm8_ablate.hip — where does k_gemm8 spend its time?  Times the library kernel and its ablated variants on the
// encoder's GEMM shapes (256 clips): full / no MFMA / no LDS-DMA in the loop / no epilogue / loads + barriers only.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I whisper-rust-ort_amd/csrc tools/gemm8_ablate.hip -o tools/gemm8_ablate
#include "../whisper-rust-ort_amd/csrc/wh_gemm8.hip"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
bool wh_ensure_dyn_lds(const void* k, size_t b) { return hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b) == hipSuccess; }
void wh_set_error(const char*, ...) {}
template <typename TO, int BN, int ABL> float run(const GemmArgs& g, int reps) {
    typedef Geo<BN> G;
    const size_t sm = (size_t)G::NSLOT * G::SLOT;
    (void)hipFuncSetAttribute((const void*)k_gemm8<TO, BN, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    dim3 grid(((g.N + BN - 1) / BN) * ((g.M + BM - 1) / BM), 1, g.batch);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k_gemm8<TO, BN, ABL>), grid, dim3(512), sm, 0, g);
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_gemm8<TO, BN, ABL>), grid, dim3(512), sm, 0, g);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    return ms / reps * 1e3f;
}
template <typename TO, int BN> void all(const char* name, const GemmArgs& g, double gf) {
    const float t0 = run<TO, BN, 0>(g, 5), t1 = run<TO, BN, 1>(g, 5), t4 = run<TO, BN, 4>(g, 5), t12 = run<TO, BN, 8 | 4>(g, 5), t16 = run<TO, BN, 16>(g, 5);
    const float t36 = run<TO, BN, 32 | 4>(g, 5), t44 = run<TO, BN, 32 | 8 | 4>(g, 5);
    printf("%-26s BN %3d: full %7.1f us (%5.0f TF/s) | no-mfma %7.1f | no-epilogue %7.1f | loads+barriers only %7.1f | no-stores %7.1f | no-epilogue, A stages only %7.1f | loads+barriers, A only %7.1f\n", name, BN, t0,
           gf / t0 * 1e3, t1, t4, t12, t16, t36, t44);
}
// The global -> LDS stream alone, two ways: MODE 0 = LDS-DMA as in k_gemm8 (global_load_lds_dwordx4, counted vmcnt waits, one barrier
// per k-step); MODE 1 = register-staged (global_load_dwordx4 into VGPRs, ds_write_b128 one k-step before the data is due).  Same
// tiles, same slot ring (256 x 256 x 32, four 32 KiB slots, three stages in flight), same source swizzle; nothing reads the slots.
template <int MODE>
__global__ __launch_bounds__(512, 2) void k_stream_skeleton(GemmArgs g, unsigned* sink) {
    constexpr int BN = 256, NSLOT = 4, SLOT_A = BM * ROWB, SLOT = SLOT_A + BN * ROWB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = g.K / BK, nbn = (g.N + BN - 1) / BN, total = nbn * ((g.M + BM - 1) / BM);
    int tile = blockIdx.x;
    { const int q = total >> 3, r = total & 7, xcd = tile & 7, idx = tile >> 3; tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx; }
    const int m0 = (tile / nbn) * BM, n0 = (tile % nbn) * BN;
    const int rl = lane >> 2, ps = lane & 3;
    const bf16* src[4];
    int dst[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int row = wave * 32 + (j & 1) * 16 + rl;
        const bool isw = j >= 2;
        const long r = min(isw ? n0 + row : m0 + row, (isw ? g.N : g.M) - 1);
        src[j] = (const bf16*)(isw ? g.W : g.A) + r * (isw ? g.ldw : g.lda) + ((ps ^ swz(row)) << 3);
        dst[j] = (isw ? SLOT_A : 0) + (wave * 32 + (j & 1) * 16) * ROWB;     // wave-instruction base; lane i lands at + i * 16
    }
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    u4 regs[NSLOT - 1][4];
    auto issue = [&](int kt, int set) {      // set = kt % 3, passed as a compile-time constant of the unrolled callers
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (MODE == 0) glds16(src[j] + (long)kt * BK, smem + (kt % NSLOT) * SLOT + dst[j]);
            else regs[set][j] = *reinterpret_cast<const u4*>(src[j] + (long)kt * BK);
        }
    };
#pragma unroll
    for (int t = 0; t < NSLOT - 1; t++) if (t < nk) issue(t, t);
    for (int kt = 0; kt < nk; kt += 3) {            // three k-steps per iteration: the register sets are named statically
#pragma unroll
        for (int h = 0; h < 3; h++) {
            const int t = kt + h;
            if (t >= nk) break;
            const int newer = min(NSLOT - 2, nk - 1 - t);
            if (newer >= 2) wait_vm<8>(); else if (newer == 1) wait_vm<4>(); else wait_vm<0>();
            if (MODE == 1) {
#pragma unroll
                for (int j = 0; j < 4; j++) *reinterpret_cast<u4*>(smem + (t % NSLOT) * SLOT + dst[j] + lane * 16) = regs[h][j];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            if (t + NSLOT - 1 < nk) issue(t + NSLOT - 1, h);
        }
    }
    __builtin_amdgcn_s_barrier();
    if (sink && tid == 0 && blockIdx.x == 0) *sink = reinterpret_cast<unsigned*>(smem)[lane];
}
// The same stream with 128-byte LDS rows (64 k per stage, 8 whole rows per wave-instruction: every request a full cache line):
// 256 x 128 tiles, three 48 KiB slots, two stages in flight (96 KB, as above).  Reports bytes moved so the rates compare.
template <int BN, int NSLOT>
__global__ __launch_bounds__(512, 2) void k_stream_skeleton128(GemmArgs g, unsigned* sink) {
    constexpr int BK2 = 64, ROWB2 = 128, SLOT_A = BM * ROWB2, SLOT = SLOT_A + BN * ROWB2, WI = BN / 64, NI = 4 + WI;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = g.K / BK2, nbn = (g.N + BN - 1) / BN, total = nbn * ((g.M + BM - 1) / BM);
    int tile = blockIdx.x;
    { const int q = total >> 3, r = total & 7, xcd = tile & 7, idx = tile >> 3; tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx; }
    const int m0 = (tile / nbn) * BM, n0 = (tile % nbn) * BN;
    const int rl = lane >> 3, ps = lane & 7;
    const bf16* src[NI];
    int dst[NI];
#pragma unroll
    for (int j = 0; j < NI; j++) {
        const bool isw = j >= 4;
        const int row = isw ? wave * (8 * WI) + (j - 4) * 8 + rl : wave * 32 + j * 8 + rl;
        const long r = min(isw ? n0 + row : m0 + row, (isw ? g.N : g.M) - 1);
        src[j] = (const bf16*)(isw ? g.W : g.A) + r * (isw ? g.ldw : g.lda) + ((ps ^ ((row >> 1) & 7)) << 3);
        dst[j] = (isw ? SLOT_A : 0) + (isw ? wave * (8 * WI) + (j - 4) * 8 : wave * 32 + j * 8) * ROWB2;
    }
    auto issue = [&](int kt) {
#pragma unroll
        for (int j = 0; j < NI; j++) glds16(src[j] + (long)kt * BK2, smem + (kt % NSLOT) * SLOT + dst[j]);
    };
#pragma unroll
    for (int t = 0; t < NSLOT - 1; t++) if (t < nk) issue(t);
    for (int t = 0; t < nk; t++) {
        const int newer = min(NSLOT - 2, nk - 1 - t);
        if (newer >= 1) wait_vm<NI>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (t + NSLOT - 1 < nk) issue(t + NSLOT - 1);
    }
    __builtin_amdgcn_s_barrier();
    if (sink && tid == 0 && blockIdx.x == 0) *sink = reinterpret_cast<unsigned*>(smem)[lane];
}
template <int BN, int NSLOT> static float run_skeleton128(const GemmArgs& g, int reps) {
    const size_t sm = (size_t)NSLOT * (BM + BN) * 128;
    (void)hipFuncSetAttribute((const void*)k_stream_skeleton128<BN, NSLOT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    dim3 grid(((g.N + BN - 1) / BN) * ((g.M + BM - 1) / BM));
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k_stream_skeleton128<BN, NSLOT>), grid, dim3(512), sm, 0, g, (unsigned*)nullptr);
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_stream_skeleton128<BN, NSLOT>), grid, dim3(512), sm, 0, g, (unsigned*)nullptr);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    return ms / reps * 1e3f;
}
template <int MODE> float run_skeleton(const GemmArgs& g, int reps) {
    const size_t sm = (size_t)4 * (BM + 256) * ROWB;
    (void)hipFuncSetAttribute((const void*)k_stream_skeleton<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    dim3 grid(((g.N + 255) / 256) * ((g.M + BM - 1) / BM));
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k_stream_skeleton<MODE>), grid, dim3(512), sm, 0, g, (unsigned*)nullptr);
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_stream_skeleton<MODE>), grid, dim3(512), sm, 0, g, (unsigned*)nullptr);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    return ms / reps * 1e3f;
}

// correctness first: every geometry / output type / tail against a host double-precision product
template <typename TO, int BN> double check_one(int M, int N, int K, bool resid, bool act, int bias_mode) {
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
    run<TO, BN, 0>(g, 1);
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
            const double tol = sizeof(TO) == 4 ? 0 : 0.004 * fabs(v);      // bf16 output rounding
            worst = std::max(worst, fabs(got - v) - tol);
        }
    hipFree(A); hipFree(W); hipFree(R); hipFree(bias); hipFree(ws); hipFree(C);
    return worst;
}
static int check_all() {
    int bad = 0;
    struct Case { int M, N, K; bool resid, act; int bm; } cases[] = {
        {512, 256, 128, false, false, 1}, {300, 384, 96, true, false, 1}, {777, 1500, 64, false, false, 2}, {256, 512, 512, true, true, 1}, {1000, 132, 32, false, true, 1}};
    for (auto& c : cases) {
        const double e1 = check_one<float, 128>(c.M, c.N, c.K, c.resid, c.act, c.bm), e2 = check_one<float, 256>(c.M, c.N, c.K, c.resid, c.act, c.bm);
        const double e3 = check_one<bf16, 128>(c.M, c.N, c.K, c.resid, c.act, c.bm), e4 = check_one<bf16, 256>(c.M, c.N, c.K, c.resid, c.act, c.bm);
        const bool ok = e1 < 2e-4 && e2 < 2e-4 && e3 < 2e-4 && e4 < 2e-4;
        printf("check M%4d N%4d K%3d res%d act%d bias%d: excess error f32 %.1e %.1e  bf16 %.1e %.1e  %s\n", c.M, c.N, c.K, c.resid, c.act, c.bm, e1, e2, e3, e4, ok ? "ok" : "MISMATCH");
        bad += !ok;
    }
    return bad;
}
int main() {
    if (check_all()) return 1;
    const long M = 256L * 1500;
    struct Shape { const char* name; int N, K; bool f32out, resid, act; } shapes[] = {
        {"QK   N1024 K512 ", 1024, 512, false, false, false}, {"fc1  N2048 K512 gelu", 2048, 512, false, false, true},
        {"fc2  N512 K2048 f32+res", 512, 2048, true, true, false}, {"O    N512 K512 f32+res", 512, 512, true, true, false}};
    bf16 *A, *W; float *R, *bias; void* C;
    hipMalloc(&A, M * 2048 * 2); hipMalloc(&W, 2048L * 2048 * 2); hipMalloc(&C, M * 2048 * 2); hipMalloc(&R, M * 512 * 4); hipMalloc(&bias, 8192);
    std::vector<unsigned short> h(1 << 24);
    unsigned x = 12345; for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3c00 + ((x >> 20) & 0x1ff) + ((x >> 31) << 15)); }  // ~+-1
    for (long off = 0; off < M * 2048 * 2; off += (long)h.size() * 2) hipMemcpy((char*)A + off, h.data(), std::min<long>(h.size() * 2, M * 2048 * 2 - off), hipMemcpyHostToDevice);
    hipMemcpy(W, h.data(), 2048L * 2048 * 2, hipMemcpyHostToDevice);
    hipMemset(R, 0, M * 512 * 4); hipMemset(bias, 0, 8192);
    for (auto& sh : shapes) {
        GemmArgs g; g.A = A; g.lda = sh.K; g.W = W; g.ldw = sh.K; g.C = sh.f32out ? (void*)R : C; g.ldc = sh.N; g.bias = bias; g.bias_mode = 1;
        g.act = sh.act; g.M = (int)M; g.N = sh.N; g.K = sh.K;
        if (sh.resid) { g.R = R; g.ldr = sh.N; }
        const double gf = 2.0 * M * sh.N * sh.K * 1e-9;
        {
            const float t0 = run_skeleton<0>(g, 5), t1 = run_skeleton<1>(g, 5), t2 = run_skeleton128<128, 3>(g, 5), t3 = run_skeleton128<256, 2>(g, 5);
            const double b256 = (double)((M + 255) / 256) * ((sh.N + 255) / 256) * (sh.K / 32) * 32768.0, b128 = (double)((M + 255) / 256) * ((sh.N + 127) / 128) * (sh.K / 64) * 49152.0;
            printf("%-26s global -> LDS stream alone: 256 x 256 tiles, 64-byte rows, 3 x 32 KB in flight: LDS-DMA %7.1f us (%5.2f TB/s) | register-staged %7.1f us | 128-byte rows: 256 x 128 tiles, 2 x 48 KB in flight %7.1f us (%5.2f TB/s); 256 x 256 tiles, 1 x 64 KB in flight %7.1f us (%5.2f TB/s)\n",
                   sh.name, t0, b256 / t0 * 1e-6, t1, t2, b128 / t2 * 1e-6, t3, b256 / t3 * 1e-6);
        }
        if (sh.f32out) { all<float, 128>(sh.name, g, gf); all<float, 256>(sh.name, g, gf); }
        else { all<bf16, 128>(sh.name, g, gf); all<bf16, 256>(sh.name, g, gf); }
    }
    return 0;
}
