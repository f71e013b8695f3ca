// Micro-benchmark: does a long-lived kernel that holds one workgroup on every CU (a persistent PixelCNN chain would) let the
// workgroups of another stream's kernel run BESIDE it?  P: 256 workgroups x 512 threads, 35 KB of LDS, sleeps ~20 ms of wall clock
// (or issues MFMAs at a low duty cycle); C: 4096 workgroups x 256 threads, 82 KB of LDS (one per CU), ~5 us of MFMAs each.
//   hipcc --offload-arch=gfx950 -O3 tools/corun_micro.cpp -o tools/corun_micro.bin && tools/corun_micro.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k_persist(float *out, unsigned long long ticks, int duty) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {
        for (int i = 0; i < duty; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, lds[threadIdx.x & 63], acc, 0, 0, 0);
        __builtin_amdgcn_s_sleep(32);
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc[0];
}
__global__ __launch_bounds__(256) void k_conv(float *out, int iters) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = threadIdx.x * 0.5f;
    __syncthreads();
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    const float x = lds[threadIdx.x & 63];
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, 1.0f, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, 2.0f, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, 3.0f, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, 4.0f, a3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[0] + a2[0] + a3[0];
}
int main() {
    hipStream_t sa, sb;
    hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    float *oa, *ob; hipMalloc(&oa, 256 * 512 * 4); hipMalloc(&ob, 4096 * 256 * 4);
    hipFuncSetAttribute((const void *)k_conv, hipFuncAttributeMaxDynamicSharedMemorySize, 84 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto wall = [&](auto fn) {
        hipDeviceSynchronize();
        hipEventRecord(e0, 0); hipStreamWaitEvent(sa, e0, 0); hipStreamWaitEvent(sb, e0, 0);
        fn();
        hipEventRecord(e1, sa); hipStreamWaitEvent(0, e1, 0); hipEventRecord(e1, sb); hipStreamWaitEvent(0, e1, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
    };
    for (int duty : {0, 8}) {
        auto P = [&] { hipLaunchKernelGGL(k_persist, dim3(256), dim3(512), 35 * 1024, sa, oa, 2000000ull, duty); };      // 20 ms
        auto C = [&] { for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(k_conv, dim3(4096), dim3(256), 82 * 1024, sb, ob, 400); };
        wall(P); wall(C);
        const float p = wall(P), c = wall(C), pc = wall([&] { P(); C(); }), cp = wall([&] { C(); P(); });
        printf("persistent kernel (MFMA duty %d): alone %.2f ms; 8 conv-like launches alone %.2f ms; P then C submitted %.2f ms; C then P %.2f ms (serial %.2f)\n",
               duty, p, c, pc, cp, p + c);
    }
    return 0;
}
