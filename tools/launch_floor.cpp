// Micro-benchmark: what does one dependent tiny kernel cost on this box (eager vs hipGraph), by block size / LDS / grid.
// Build: hipcc --offload-arch=gfx950 -O3 tools/launch_floor.cpp -o /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int LDS>
__global__ void k_chain(const float *in, float *out, int n) {
    __shared__ float s[LDS > 0 ? LDS : 1];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (LDS > 0) { s[threadIdx.x % LDS] = in[i % n]; __syncthreads(); }
    if (i < n) out[i] = in[i] + (LDS > 0 ? s[0] * 0.f : 0.f) + 1.f;
}

template <typename F>
double time_chain(F launch, int iters, hipStream_t s, bool graph) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipGraphExec_t ex = nullptr;
    if (graph) {
        hipGraph_t g;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < iters; ++i) launch(i);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
        hipGraphDestroy(g);
        hipGraphLaunch(ex, s);   // warm
        hipStreamSynchronize(s);
    } else {
        for (int i = 0; i < 20; ++i) launch(i);
        hipStreamSynchronize(s);
    }
    hipEventRecord(a, s);
    if (graph) hipGraphLaunch(ex, s); else for (int i = 0; i < iters; ++i) launch(i);
    hipEventRecord(b, s);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (ex) hipGraphExecDestroy(ex);
    return ms * 1e3 / iters;
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int n = 1 << 16;
    float *x, *y; CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4));
    CK(hipMemset(x, 0, n * 4));
    int iters = 2000;
    struct Cfg { int grid, block; const char *name; };
    Cfg cfgs[] = {{1, 64, "1x64"}, {16, 256, "16x256"}, {32, 1024, "32x1024"}, {256, 256, "256x256"}, {64, 512, "64x512"}};
    for (auto &c : cfgs) {
        for (int g = 0; g < 2; ++g) {
            float *p = x, *q = y;
            double us0 = time_chain([&](int i) { hipLaunchKernelGGL(k_chain<0>, dim3(c.grid), dim3(c.block), 0, s, (i & 1) ? q : p, (i & 1) ? p : q, n); }, iters, s, g);
            double us1 = time_chain([&](int i) { hipLaunchKernelGGL(k_chain<16384>, dim3(c.grid), dim3(c.block), 0, s, (i & 1) ? q : p, (i & 1) ? p : q, n); }, iters, s, g);
            printf("%-8s %-6s  no-LDS %.2f us/kernel   64KB-LDS %.2f us/kernel\n", c.name, g ? "graph" : "eager", us0, us1);
        }
    }
    return 0;
}
