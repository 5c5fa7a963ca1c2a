// grid_barrier_bench.hip — cost of a device-wide barrier inside one persistent launch on gfx950 (agent-scope release/acquire
// on a monotonic counter, every wait bounded by a wall-clock limit), against the boundary between two dependent kernels
// of a replayed hipGraph.  Also checks that data written before the barrier by a workgroup on another XCD is seen after it.
// Build: hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/grid_barrier_bench.hip -o tools/grid_barrier_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ bool grid_sync(unsigned* cnt, unsigned target, int* err) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 20000000LL) { *err = 1; break; }   // 0.2 s at 100 MHz
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return true;
}

template <int MODE>  // 0 barrier only, 1 + exchange of one 16-byte item per thread with a workgroup half a grid away
__global__ __launch_bounds__(512) void k_persist(unsigned* cnt, int reps, float* buf, int* err, int* bad) {
    const int G = gridDim.x, b = blockIdx.x, peer = (b + G / 2 + 1) % G;
    int nbad = 0;
    for (int r = 0; r < reps; r++) {
        if (MODE == 1) {
            float4 v = {(float)r, (float)b, (float)threadIdx.x, 1.0f};
            reinterpret_cast<float4*>(buf)[((size_t)(r & 1) * G + b) * 512 + threadIdx.x] = v;
        }
        grid_sync(cnt, (unsigned)(r + 1) * G, err);
        if (*err) return;
        if (MODE == 1) {
            float4 v = reinterpret_cast<float4*>(buf)[((size_t)(r & 1) * G + peer) * 512 + threadIdx.x];
            if (v.x != (float)r || v.y != (float)peer || v.z != (float)threadIdx.x) nbad++;
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}
__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) *p = 1; }

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned* cnt; int *err, *bad; float* buf;
    CK(hipMalloc(&cnt, 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&buf, (size_t)2 * 1024 * 512 * 16));
    const int reps = 2000;
    for (int G : {128, 256, 512}) for (int mode : {0, 1}) {
        int occ = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, mode ? k_persist<1> : k_persist<0>, 512, 0));
        if (G > occ * 256) { printf("G=%d skipped (occupancy %d per CU)\n", G, occ); continue; }
        double best = 1e9; int herr = 0, hbad = 0;
        for (int t = 0; t < 3; t++) {
            CK(hipMemsetAsync(cnt, 0, 4, s)); CK(hipMemsetAsync(err, 0, 4, s)); CK(hipMemsetAsync(bad, 0, 4, s));
            CK(hipStreamSynchronize(s));
            const double t0 = now();
            if (mode) hipLaunchKernelGGL(k_persist<1>, dim3(G), dim3(512), 0, s, cnt, reps, buf, err, bad);
            else hipLaunchKernelGGL(k_persist<0>, dim3(G), dim3(512), 0, s, cnt, reps, buf, err, bad);
            CK(hipStreamSynchronize(s));
            best = std::min(best, (now() - t0) / reps * 1e6);
            CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
            if (herr) break;
        }
        printf("persistent G=%3d x 512 threads, %s: %6.2f us per barrier  timeout=%d stale_reads=%d\n", G, mode ? "barrier + exchange" : "barrier only     ", best, herr, hbad);
        if (herr) return 1;
    }
    for (int G : {256, 1024}) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k_empty, dim3(G), dim3(512), 0, s, (int*)nullptr);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        double best = 1e9;
        for (int t = 0; t < 3; t++) { const double t0 = now(); CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); best = std::min(best, (now() - t0) / reps * 1e6); }
        printf("graph of %d dependent empty kernels, grid %4d x 512: %6.2f us per kernel\n", reps, G, best);
    }
    return 0;
}
