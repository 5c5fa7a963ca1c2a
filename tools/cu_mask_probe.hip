// cu_mask_probe.hip — what a hipExtStreamCreateWithCUMask mask selects on this part, and what a CU subset can do.
//
//   1. mapping: for a set of masks, a grid of small workgroups records (XCC_ID, SE_ID, SH_ID, CU_ID) of the compute unit
//      each one ran on (s_getreg HW_ID / XCC_ID) -> compute units used per XCD;
//   2. HBM stream vs compute units: a 16-byte-per-lane non-temporal read of 3 GiB (the access pattern of
//      k_dec_cross_attn at 1024 clips) on the low n mask bits, n = 32 .. 256;
//   3. isolation: that stream on one part of the chip beside an MFMA loop on the rest, each alone and both together.
//
// hipcc --offload-arch=gfx950 -O3 -o tools/cu_mask_probe tools/cu_mask_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <set>
#include <vector>

#define CK(x)                                                                                    \
    do {                                                                                         \
        hipError_t e_ = (x);                                                                     \
        if (e_ != hipSuccess) {                                                                  \
            fprintf(stderr, "HIP error %s at %s:%d: %s\n", hipGetErrorString(e_), __FILE__, __LINE__, #x); \
            exit(1);                                                                             \
        }                                                                                        \
    } while (0)

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__global__ void k_where(unsigned* out, int spin) {
    // keep the workgroup resident for a moment so that the grid spreads over every available compute unit
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));    // HW_REG_HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));  // HW_REG_XCC_ID
        out[blockIdx.x * 2] = hw;
        out[blockIdx.x * 2 + 1] = xcc;
    }
}

__global__ __launch_bounds__(256) void k_stream(const f32x4* __restrict__ p, long n16, float* sink) {
    f32x4 acc = {0, 0, 0, 0};
    const long stride = (long)gridDim.x * blockDim.x;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const f32x4 a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + stride);
        const f32x4 c = __builtin_nontemporal_load(p + i + 2 * stride), d = __builtin_nontemporal_load(p + i + 3 * stride);
        acc += a + b + c + d;
    }
    for (; i < n16; i += stride) acc += __builtin_nontemporal_load(p + i);
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345f) sink[0] = acc[0];
}

__global__ __launch_bounds__(256) void k_mfma(float* sink, int iters) {
    typedef __attribute__((ext_vector_type(4))) float v4;
    bf16x8 a, b;
    for (int e = 0; e < 8; e++) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(e * 0.5f); }
    v4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
    }
    if (c0[0] + c1[0] + c2[0] + c3[0] == 1.2345f) sink[0] = c0[0];
}

static std::vector<uint32_t> mask_range(int first, int count) {
    std::vector<uint32_t> w(8, 0u);
    for (int i = first; i < first + count; i++) w[i >> 5] |= 1u << (i & 31);
    return w;
}

static hipStream_t masked_stream(const std::vector<uint32_t>& m) {
    hipStream_t s;
    size_t bits = 0;
    for (uint32_t w : m) bits += (size_t)__builtin_popcount(w);
    if (bits >= 256) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));   // the whole chip: no mask
    else CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)m.size(), m.data()));
    return s;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);   // a run that is killed at its time limit still shows how far it got
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, %d CUs\n", prop.name, prop.multiProcessorCount);
    const int NWG = 4096;
    unsigned* d_where;
    CK(hipMalloc(&d_where, NWG * 8));
    std::vector<unsigned> h(NWG * 2);

    // ---- 1. mapping --------------------------------------------------------------------------------
    struct Case { const char* name; int first, count; };
    const Case cases[] = {{"bits 0..7", 0, 8},     {"bits 0..15", 0, 16},   {"bits 0..31", 0, 32},    {"bits 0..63", 0, 64},
                          {"bits 0..127", 0, 128}, {"bits 64..255", 64, 192}, {"bits 96..255", 96, 160}, {"bits 128..255", 128, 128},
                          {"bits 8..15", 8, 8},    {"bit 0", 0, 1},          {"bit 1", 1, 1},           {"bit 8", 8, 1}};
    // (a mask with all 256 bits set never got past this point on the test box — the run was killed by its time limit; the
    // library turns such a mask into an ordinary stream.  An XCD whose 32 bits are all clear is not restricted at all: "bit 0"
    // alone selects one compute unit of XCD 0 and every compute unit of the other seven.)
    for (const Case& cs : cases) {
        hipStream_t s = masked_stream(mask_range(cs.first, cs.count));
        CK(hipMemsetAsync(d_where, 0xFF, NWG * 8, s));
        hipLaunchKernelGGL(k_where, dim3(NWG), dim3(64), 0, s, d_where, 2000);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), d_where, NWG * 8, hipMemcpyDeviceToHost));
        std::map<int, std::set<int>> per_xcc;   // xcc -> {se * 64 + sh * 16 + cu}
        for (int i = 0; i < NWG; i++) {
            const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xF;
            const int cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            per_xcc[(int)xcc].insert(se * 64 + sh * 16 + cu);
        }
        int total = 0;
        printf("%-14s (%3d bits): CUs per XCC:", cs.name, cs.count);
        for (auto& kv : per_xcc) { printf(" x%d=%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
        printf("  total %d\n", total);
        if (cs.count <= 8) {
            for (auto& kv : per_xcc) {
                printf("    xcc %d:", kv.first);
                for (int id : kv.second) printf(" se%d.sh%d.cu%d", id / 64, (id / 16) & 1, id & 15);
                printf("\n");
            }
        }
        CK(hipStreamDestroy(s));
    }

    // ---- 2. HBM stream vs compute units -----------------------------------------------------------
    const long bytes = 3L << 30, n16 = bytes / 16;
    f32x4* buf;
    float* sink;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time_stream = [&](hipStream_t s, int wgs, int reps) {
        hipLaunchKernelGGL(k_stream, dim3(wgs), dim3(256), 0, s, buf, n16, sink);
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_stream, dim3(wgs), dim3(256), 0, s, buf, n16, sink);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms / reps;
    };
    printf("\nnon-temporal 16 B/lane stream of 3 GiB, low n mask bits (n/8 CUs per XCD):\n");
    for (int n : {32, 64, 96, 128, 160, 192, 224, 256}) {
        hipStream_t s = masked_stream(mask_range(0, n));
        float best = 1e9f;
        int best_wg = 0;
        for (int wgs : {n * 2, n * 4, n * 8}) {
            const float ms = time_stream(s, wgs, 3);
            if (ms < best) { best = ms; best_wg = wgs; }
        }
        printf("  %3d CUs: %.3f ms = %.2f TB/s (best grid %d workgroups)\n", n, best, bytes / (best * 1e-3) / 1e12, best_wg);
        CK(hipStreamDestroy(s));
    }

    // ---- 3. isolation ------------------------------------------------------------------------------
    printf("\nstream on the upper CUs beside an MFMA loop on the lower ones (ms, each alone / together):\n");
    for (int split : {32, 64, 96, 128}) {
        hipStream_t sd = masked_stream(mask_range(split, 256 - split)), se = masked_stream(mask_range(0, split));
        const float t_stream = time_stream(sd, (256 - split) * 4, 3);
        const int iters = 200000;
        auto run_mfma = [&]() { hipLaunchKernelGGL(k_mfma, dim3(split * 4), dim3(256), 0, se, sink + 8, iters); };
        run_mfma();
        CK(hipStreamSynchronize(se));
        hipEvent_t m0, m1;
        CK(hipEventCreate(&m0));
        CK(hipEventCreate(&m1));
        CK(hipEventRecord(m0, se));
        run_mfma();
        CK(hipEventRecord(m1, se));
        CK(hipStreamSynchronize(se));
        float t_mfma;
        CK(hipEventElapsedTime(&t_mfma, m0, m1));
        // together: the MFMA kernel is launched first, the stream kernels run under it
        const int reps = std::max(1, (int)(t_mfma / t_stream));
        CK(hipEventRecord(m0, se));
        run_mfma();
        CK(hipEventRecord(m1, se));
        CK(hipEventRecord(e0, sd));
        for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_stream, dim3((256 - split) * 4), dim3(256), 0, sd, buf, n16, sink);
        CK(hipEventRecord(e1, sd));
        CK(hipDeviceSynchronize());
        float t_mfma2, t_stream2;
        CK(hipEventElapsedTime(&t_mfma2, m0, m1));
        CK(hipEventElapsedTime(&t_stream2, e0, e1));
        printf("  MFMA on %3d CUs / stream on %3d: stream %.3f -> %.3f ms per pass (%.2f -> %.2f TB/s), MFMA loop %.2f -> %.2f ms\n", split,
               256 - split, t_stream, t_stream2 / reps, bytes / (t_stream * 1e-3) / 1e12, bytes / (t_stream2 / reps * 1e-3) / 1e12, t_mfma, t_mfma2);
        CK(hipStreamDestroy(sd));
        CK(hipStreamDestroy(se));
    }
    return 0;
}
