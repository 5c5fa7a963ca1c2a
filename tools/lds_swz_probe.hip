// lds_swz_probe.hip — bank conflicts of ds_read_b128 fragment reads from 64-byte LDS rows under candidate chunk swizzles.
// Pattern A (16x16x32 operand): lane l -> row l & 15, chunk l >> 4.  Pattern B (32x32x16 operand): lane l -> row l & 31,
// chunk (l >> 5) + 2h.  LDS chunk = chunk ^ f(row), f(row) = bit a of row | bit b of row << 1.  Prints cycles per read
// (4 waves of one workgroup reading concurrently, 8 independent reads in flight per wave).
// Build: hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/lds_swz_probe.hip -o tools/lds_swz_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__global__ __launch_bounds__(256) void k_probe(const int* offs, unsigned* out, long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<unsigned*>(smem)[i] = i;
    __syncthreads();
    const int off = offs[threadIdx.x & 63];
    u32x4 acc = {0, 0, 0, 0};
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(smem + off + u * 2048 + (i & 3) * 16384);
            acc += v;
        }
    }
    const long long t1 = clock64();
    out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
int main() {
    int* d_off; unsigned* d_out; long long* d_cyc;
    hipMalloc(&d_off, 256); hipMalloc(&d_out, 1024); hipMalloc(&d_cyc, 32);
    hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const int iters = 2000;
    auto run = [&](const std::vector<int>& offs) {
        hipMemcpy(d_off, offs.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(256), 65536, 0, d_off, d_out, d_cyc, iters);
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(256), 65536, 0, d_off, d_out, d_cyc, iters);
        long long c[4]; hipMemcpy(c, d_cyc, 32, hipMemcpyDeviceToHost);
        return (double)(c[0] + c[1] + c[2] + c[3]) / 4 / (iters * 8.0);
    };
    std::vector<int> lin(64);
    for (int l = 0; l < 64; l++) lin[l] = l * 16;
    printf("linear 1 KiB (conflict-free reference): %.2f cycles per read per wave\n", run(lin));
    for (int pat = 0; pat < 2; pat++) {
        printf("pattern %s\n", pat ? "B: row l&31, chunk (l>>5) [+2h]" : "A: row l&15, chunk l>>4");
        for (int a = -1; a < 5; a++) for (int b = -1; b < 5; b++) {
            double worst = 0;
            for (int h = 0; h < (pat ? 2 : 1); h++) {
                std::vector<int> offs(64);
                for (int l = 0; l < 64; l++) {
                    const int row = pat ? (l & 31) : (l & 15), chunk = pat ? ((l >> 5) + 2 * h) : (l >> 4);
                    const int f = (a >= 0 ? ((row >> a) & 1) : 0) | ((b >= 0 ? ((row >> b) & 1) : 0) << 1);
                    offs[l] = row * 64 + ((chunk ^ f) << 4);
                }
                worst = std::max(worst, run(offs));
            }
            printf("  f = bit%2d | bit%2d << 1 : %.2f\n", a, b, worst);
        }
    }
    return 0;
}
