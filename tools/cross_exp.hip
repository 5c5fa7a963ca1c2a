// cross_exp.hip — where do the ~10 us between the streaming floor and the measured cross-attention launch go?
// Variants of k_dec_cross_attn (bf16, d = 512): MODE 0 full, 1 no global epilogue (partials stored, no ticket /
// merge), 2 loads only.  NW waves per workgroup, UNROLL keys per register set.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/cross_exp.hip -o tools/cross_exp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include "../whisper-rust-ort_amd/csrc/wh_common.h"

int wh_fail_hip(hipError_t, const char*, const char*, int) { return 1; }
void wh_set_error(const char*, ...) {}

template <int NW, int UNROLL, int MODE>
__global__ __launch_bounds__(NW * 64) void k_cross(const bf16* __restrict__ q, const bf16* __restrict__ ck, const bf16* __restrict__ cv,
                                                   float* __restrict__ part, float* __restrict__ ml, bf16* __restrict__ out,
                                                   int* __restrict__ tickets, int S, int d, int n_heads, int splits) {
    constexpr int EPC = 8, LPH = 8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef bf16x8 vec_t;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sp = blockIdx.x, b = blockIdx.y;
    const int per = (S + splits - 1) / splits;
    const int ks = sp * per, ke = min(S, ks + per);
    const int j1 = ke;
    float o[EPC], mrun = -INFINITY, lrun = 0.0f;
    wh_u32x4 qd = *reinterpret_cast<const wh_u32x4*>(q + (long)b * d + lane * EPC);
#pragma unroll
    for (int u = 0; u < EPC; u++) o[u] = 0.0f;
    const bf16* kb = ck + (long)b * S * d;
    const bf16* vb = cv + (long)b * S * d;
    vec_t kA[UNROLL], vA[UNROLL], kB[UNROLL], vB[UNROLL];
    auto load_set = [&](vec_t (&kk)[UNROLL], vec_t (&vv)[UNROLL], int j) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const int jj = min(j + u, j1 - 1);
            kk[u] = *reinterpret_cast<const vec_t*>(kb + (long)jj * d + lane * EPC);
            vv[u] = *reinterpret_cast<const vec_t*>(vb + (long)jj * d + lane * EPC);
        }
    };
    unsigned xacc = 0;
    auto compute_set = [&](const vec_t (&kk)[UNROLL], const vec_t (&vv)[UNROLL], int j) {
        if constexpr (MODE == 2) {
#pragma unroll
            for (int u = 0; u < UNROLL; u++) { wh_u32x4 a = as_u32x4(kk[u]), c = as_u32x4(vv[u]); xacc ^= a.x ^ a.y ^ a.z ^ a.w ^ c.x ^ c.y ^ c.z ^ c.w; }
            return;
        }
        float s[UNROLL];
        float mx = mrun;
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            float t = dot8_bf16(as_u32x4(kk[u]), qd, 0.0f);
            t = dpp_group_sum<LPH>(t);
            s[u] = (j + u < j1) ? t : -INFINITY;
            mx = fmaxf(mx, s[u]);
        }
        const float scale = __expf(mrun - mx);
        float ls = lrun * scale;
#pragma unroll
        for (int e = 0; e < EPC; e++) o[e] *= scale;
#pragma unroll
        for (int u = 0; u < UNROLL; u += 2) {
            const float p0 = __expf(s[u] - mx), p1 = __expf(s[u + 1] - mx);
            ls += p0 + p1;
            const bf16x2 pp = bf16x2{(bf16)p0, (bf16)p1};
            const wh_u32x4 va = as_u32x4(vv[u]), vb2 = as_u32x4(vv[u + 1]);
            pv2_bf16(va.x, vb2.x, pp, o[0], o[1]);
            pv2_bf16(va.y, vb2.y, pp, o[2], o[3]);
            pv2_bf16(va.z, vb2.z, pp, o[4], o[5]);
            pv2_bf16(va.w, vb2.w, pp, o[6], o[7]);
        }
        mrun = mx;
        lrun = ls;
    };
    constexpr int GS = NW * UNROLL;
    const int j0 = ks + wave * UNROLL;
    if (j0 < j1) load_set(kA, vA, j0);
    for (int j = j0; j < j1; j += 2 * GS) {
        const bool hasB = j + GS < j1;
        if (hasB) load_set(kB, vB, j + GS);
        compute_set(kA, vA, j);
        if (hasB) {
            if (j + 2 * GS < j1) load_set(kA, vA, j + 2 * GS);
            compute_set(kB, vB, j + GS);
        }
    }
    if constexpr (MODE == 2) { if (xacc == 0x12345678u) out[tid] = (bf16)1.0f; return; }
    float* wm = smem;
    float* wl = wm + NW * n_heads;
    float* wo = wl + NW * n_heads;
    if ((lane % LPH) == 0) { wm[wave * n_heads + lane / LPH] = mrun; wl[wave * n_heads + lane / LPH] = lrun; }
#pragma unroll
    for (int e = 0; e < EPC; e++) wo[wave * d + lane * EPC + e] = o[e];
    __syncthreads();
    float* pp = part + ((long)b * splits + sp) * d;
    float* mp = ml + ((long)b * splits + sp) * n_heads * 2;
    for (int n = tid; n < d; n += NW * 64) {
        const int h = n / 64;
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < NW; w++) M = fmaxf(M, wm[w * n_heads + h]);
        float num = 0.0f, den = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const float mw = wm[w * n_heads + h];
            const float sc = (mw == -INFINITY) ? 0.0f : __expf(mw - M);
            num += sc * wo[w * d + n];
            den += sc * wl[w * n_heads + h];
        }
        __hip_atomic_store(pp + n, num, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((n % 64) == 0) {
            __hip_atomic_store(mp + h, M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(mp + n_heads + h, den, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if constexpr (MODE == 1) return;
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const int t = __hip_atomic_fetch_add(tickets + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == splits - 1);
        if (s_last) {
            __hip_atomic_store(tickets + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (!s_last) return;
    constexpr int MAXS = 16;
    const float* pb = part + (long)b * splits * d;
    const float* mb = ml + (long)b * splits * n_heads * 2;
    for (int n = tid; n < d; n += NW * 64) {
        const int h = n / 64;
        float mv[MAXS], lv[MAXS], pv[MAXS];
#pragma unroll
        for (int s2 = 0; s2 < MAXS; s2++) {
            mv[s2] = -INFINITY; lv[s2] = 0.0f; pv[s2] = 0.0f;
            if (s2 < splits) {
                mv[s2] = __hip_atomic_load(mb + s2 * n_heads * 2 + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lv[s2] = __hip_atomic_load(mb + s2 * n_heads * 2 + n_heads + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pv[s2] = __hip_atomic_load(pb + (long)s2 * d + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        float M = -INFINITY;
#pragma unroll
        for (int s2 = 0; s2 < MAXS; s2++) M = fmaxf(M, mv[s2]);
        float num = 0.0f, den = 0.0f;
#pragma unroll
        for (int s2 = 0; s2 < MAXS; s2++) {
            const float w = (mv[s2] == -INFINITY) ? 0.0f : __expf(mv[s2] - M);
            num += w * pv[s2];
            den += w * lv[s2];
        }
        out[(long)b * d + n] = (bf16)(num / den);
    }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void* dmalloc(size_t b) { void* p; hipMalloc(&p, b); hipMemset(p, 0, b); return p; }
static double time_chain(hipStream_t s, int reps, const std::function<void()>& f) {
    f(); hipStreamSynchronize(s);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < reps; i++) f();
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    double best = 1e9;
    for (int r = 0; r < 5; r++) { double t0 = now(); hipGraphLaunch(ge, s); hipStreamSynchronize(s); best = std::min(best, (now() - t0) / reps * 1e6); }
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return best;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64;
    const int d = 512, S = 1500, H = 8, L = 6;
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const size_t plane = (size_t)B * S * d;
    bf16* kv = (bf16*)dmalloc(plane * 2 * L * 2); bf16* q = (bf16*)dmalloc((size_t)B * d * 2); bf16* out = (bf16*)dmalloc((size_t)B * d * 2);
    int* tickets = (int*)dmalloc(256 * 4);
    float* part = (float*)dmalloc((size_t)B * 16 * d * 4); float* ml = (float*)dmalloc((size_t)B * 16 * H * 2 * 4);
    int l = 0;
#define RUN(NW_, U_, MODE_, SPLITS_)                                                                                               \
    {                                                                                                                              \
        const int splits = SPLITS_;                                                                                                \
        const size_t sm = sizeof(float) * ((size_t)2 * NW_ * H + (size_t)NW_ * d);                                                 \
        double us = time_chain(s, 60, [&]() {                                                                                      \
            hipLaunchKernelGGL((k_cross<NW_, U_, MODE_>), dim3(splits, B), dim3(NW_ * 64), sm, s, q, kv + (size_t)(2 * l) * plane,  \
                               kv + (size_t)(2 * l + 1) * plane, part, ml, out, tickets, S, d, H, splits);                         \
            l = (l + 1) % L;                                                                                                       \
        });                                                                                                                        \
        printf("NW=%d U=%d mode=%d splits=%2d : %.2f us (%.2f TB/s)\n", NW_, U_, MODE_, splits, us, 2.0 * S * d * 2 * B / us / 1e6); \
    }
    if (B >= 128) {  // large batches: one or two key ranges per clip
        RUN(4, 4, 0, 1) RUN(4, 4, 1, 1) RUN(4, 4, 2, 1) RUN(8, 4, 1, 1) RUN(8, 2, 1, 1) RUN(4, 8, 1, 1) RUN(8, 4, 2, 1) RUN(4, 4, 1, 2) RUN(4, 4, 2, 2) RUN(16, 2, 1, 1) RUN(16, 2, 2, 1)
        return 0;
    }
    for (int sp : {4, 8}) {
        if (sp == 4) { RUN(4, 4, 0, 4) RUN(4, 4, 1, 4) RUN(4, 4, 2, 4) RUN(8, 4, 0, 4) RUN(8, 4, 1, 4) RUN(8, 4, 2, 4) RUN(8, 2, 0, 4) RUN(8, 2, 2, 4) RUN(4, 8, 2, 4) }
        else { RUN(4, 4, 0, 8) RUN(4, 4, 1, 8) RUN(4, 4, 2, 8) RUN(8, 2, 0, 8) RUN(8, 2, 2, 8) }
    }
    RUN(4, 4, 2, 16) RUN(4, 4, 2, 2) RUN(8, 4, 2, 2) RUN(16, 2, 2, 2) RUN(16, 2, 0, 2) RUN(16, 2, 2, 4) RUN(16, 2, 0, 4)
    return 0;
}
