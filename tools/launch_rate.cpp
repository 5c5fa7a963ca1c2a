// launch_rate.cpp — what does one dependent kernel boundary cost on this box?  (eager vs hipGraph,
// empty kernel vs a kernel that reads what the previous one wrote)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_empty() {}
__global__ void k_touch(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.0f; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    float* p; hipMalloc(&p, 1 << 22); hipMemset(p, 0, 1 << 22);
    const int N = 2000;
    for (int mode = 0; mode < 4; mode++) {
        const int blocks = (mode & 1) ? 128 : 1; const bool touch = mode >= 2;
        auto run = [&]() { for (int i = 0; i < N; i++) { if (touch) hipLaunchKernelGGL(k_touch, dim3(blocks), dim3(256), 0, s, p, blocks * 256); else hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, s); } };
        run(); hipStreamSynchronize(s);
        double t0 = now(); run(); hipStreamSynchronize(s); double eager = (now() - t0) / N * 1e6;
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal); run(); hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        t0 = now(); hipGraphLaunch(ge, s); hipStreamSynchronize(s); double graph = (now() - t0) / N * 1e6;
        printf("blocks=%3d touch=%d : eager %.2f us/kernel, graph %.2f us/kernel\n", blocks, (int)touch, eager, graph);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
