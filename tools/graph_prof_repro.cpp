// graph_prof_repro.cpp — does `rocprofv3 --kernel-trace` survive the replay of a captured hipGraph on this image?
// Minimal stand-in for the decode step of wh_api.cpp (no library code): N kernel nodes captured from one stream, a
// memset node optional, kernels taking either two scalars or a by-value struct of BYTES bytes (the decode kernels take
// ~300-byte argument structs), REPLAYS launches of the instantiated graph, then destroy after a stream sync.
//   usage: graph_prof_repro <nodes> <replays> <struct bytes: 0 | 64..2048> <memset node: 0|1>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int BYTES> struct Big { float* p; int n; char pad[BYTES - 12]; };
__global__ void k_small(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0f; }
template <int BYTES> __global__ void k_big(Big<BYTES> a) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < a.n) a.p[i] += 1.0f + a.pad[0]; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 2; } } while (0)
int main(int argc, char** argv) {
    const int nodes = argc > 1 ? atoi(argv[1]) : 4, replays = argc > 2 ? atoi(argv[2]) : 3, bytes = argc > 3 ? atoi(argv[3]) : 0;
    const int with_memset = argc > 4 ? atoi(argv[4]) : 0;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float* p; CK(hipMalloc(&p, 1 << 20)); CK(hipMemset(p, 0, 1 << 20));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    if (with_memset) CK(hipMemsetAsync(p, 0, 4096, s));
    for (int i = 0; i < nodes; i++) {
        if (bytes == 0) hipLaunchKernelGGL(k_small, dim3(4), dim3(256), 0, s, p, 1024);
        else if (bytes <= 64) { Big<64> a{}; a.p = p; a.n = 1024; hipLaunchKernelGGL(k_big<64>, dim3(4), dim3(256), 0, s, a); }
        else if (bytes <= 320) { Big<320> a{}; a.p = p; a.n = 1024; hipLaunchKernelGGL(k_big<320>, dim3(4), dim3(256), 0, s, a); }
        else { Big<2048> a{}; a.p = p; a.n = 1024; hipLaunchKernelGGL(k_big<2048>, dim3(4), dim3(256), 0, s, a); }
    }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    printf("instantiated %d nodes (arg bytes %d, memset %d)\n", nodes, bytes, with_memset); fflush(stdout);
    for (int r = 0; r < replays; r++) { CK(hipGraphLaunch(ge, s)); printf("launched %d\n", r); fflush(stdout); }
    CK(hipStreamSynchronize(s));
    float h = 0; CK(hipMemcpy(&h, p, 4, hipMemcpyDeviceToHost));
    printf("done: p[0] = %.0f (expected %d)\n", h, nodes * replays);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return 0;
}
