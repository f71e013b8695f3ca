// Micro-benchmark: cost of a software grid barrier (all workgroups of one persistent kernel) on MI355X, with the data
// exchange a PixelCNN chain stage needs (every workgroup publishes a few hundred floats that every other one may read).
// Build: hipcc --offload-arch=gfx950 -O3 tools/grid_barrier.cpp -o /tmp/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Sync { unsigned count; unsigned abort; };

__device__ __forceinline__ bool grid_barrier(Sync *s, unsigned target) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&s->count, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(&s->count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 100000000ull) { __hip_atomic_store(&s->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}

// mode 0: plain stores/loads + agent-scope release/acquire fences (L2 write-back / invalidate)
// mode 1: data moved with agent-scope relaxed atomic dword stores/loads (sc1), no cache-wide fences on the data path
template <int MODE>
__global__ __launch_bounds__(512) void k_persist(float *buf, Sync *s, int iters, int *errs) {
    const int G = gridDim.x, b = blockIdx.x, t = threadIdx.x;
    int bad = 0;
    for (int i = 0; i < iters; ++i) {
        float *w = buf + (size_t)(i & 1) * G * 512;
        const float v = (float)(i * 7 + b);
        if (MODE == 0) w[b * 512 + t] = v;
        else __hip_atomic_store(&w[b * 512 + t], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (!grid_barrier(s, (unsigned)(i + 1) * G)) return;
        const int src = (b + 1 + (i % 7) * 9) % G;
        float r;
        if (MODE == 0) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); r = w[src * 512 + t]; }
        else r = __hip_atomic_load(&w[src * 512 + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (r != (float)(i * 7 + src)) ++bad;
    }
    if (bad) atomicAdd(errs, bad);
}

int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float *buf; Sync *s; int *errs;
    CK(hipMalloc(&buf, 2 * 1024 * 512 * 4)); CK(hipMalloc(&s, sizeof(Sync))); CK(hipMalloc(&errs, 4));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 4000;
    for (int mode = 0; mode < 2; ++mode)
        for (int G : {32, 64, 128, 256, 512}) {
            CK(hipMemsetAsync(s, 0, sizeof(Sync), st)); CK(hipMemsetAsync(errs, 0, 4, st));
            hipEventRecord(a, st);
            if (mode == 0) hipLaunchKernelGGL(k_persist<0>, dim3(G), dim3(512), 0, st, buf, s, iters, errs);
            else hipLaunchKernelGGL(k_persist<1>, dim3(G), dim3(512), 0, st, buf, s, iters, errs);
            hipEventRecord(b, st);
            CK(hipEventSynchronize(b));
            float ms; hipEventElapsedTime(&ms, a, b);
            Sync hs; int he; CK(hipMemcpy(&hs, s, sizeof(hs), hipMemcpyDeviceToHost)); CK(hipMemcpy(&he, errs, 4, hipMemcpyDeviceToHost));
            printf("mode %d  G=%3d x512: %.2f us / barrier+exchange   errors %d  abort %u\n", mode, G, ms * 1e3 / iters, he, hs.abort);
        }
    return 0;
}
