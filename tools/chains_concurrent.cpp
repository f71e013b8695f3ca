// Micro-benchmark: how do K independent chains of DEPENDENT launches share the chip?  K = 1..4 streams, each replays a hipGraph of N
// dependent kernels (256 workgroups x 512 threads, like a chain stage at 256 clips).  Kernel bodies:
//   trivial    one load + one store per thread
//   latency    a dependent chain of 3 cold 16-byte loads per thread (descriptor -> operand -> epilogue term) from a 256 MB buffer, then a store
//   mfma       latency + ~6 us of v_mfma_f32_16x16x4_f32 per wave (the wide kernel's work at the pipe's rate)
//   mfma128    the same with 128 KB of LDS per workgroup (one workgroup per CU: the wide kernel's footprint)
// Reported: us per dependent launch seen by each chain, and the aggregate launch rate, for K chains in flight.
// Build: hipcc --offload-arch=gfx950 -O3 tools/chains_concurrent.cpp -o tools/chains_concurrent.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int LDSKB>
__global__ __launch_bounds__(512) void k_stage(const f32x4 *big, const float *in, float *out, unsigned salt, int mfma_iters) {
    extern __shared__ float lds[];
    const unsigned gid = blockIdx.x * 512 + threadIdx.x;
    float v = in[gid];
    if (MODE >= 1) {
        unsigned idx = (gid * 2654435761u + salt * 40503u) & ((1u << 24) - 1);   // 16 M x 16 B = 256 MB
        f32x4 a = big[idx];
        idx = (idx + (unsigned)(a[0] != 12345.f) * 7919u * (salt + 1)) & ((1u << 24) - 1);
        f32x4 b = big[idx];
        idx = (idx + (unsigned)(b[1] != 12345.f) * 104729u) & ((1u << 24) - 1);
        f32x4 c = big[idx];
        v += a[0] + b[1] + c[2];
    }
    if (MODE >= 2) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < mfma_iters; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v, 1.0f, acc, 0, 0, 0);
        v += acc[0] * 1e-30f;
    }
    if (LDSKB > 0) { lds[threadIdx.x] = v; __syncthreads(); v += lds[(threadIdx.x + 1) & 511] * 0.f; }
    out[gid] = v + 1.f;
}

int main(int argc, char **argv) {
    const int N = 600, WG = 256, maxK = 4;
    hipStream_t st[maxK];
    for (int k = 0; k < maxK; ++k) CK(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking));
    f32x4 *big; CK(hipMalloc(&big, (size_t)256 << 20)); CK(hipMemset(big, 0, (size_t)256 << 20));
    float *buf[maxK][2];
    for (int k = 0; k < maxK; ++k) for (int j = 0; j < 2; ++j) { CK(hipMalloc(&buf[k][j], WG * 512 * 4)); CK(hipMemset(buf[k][j], 0, WG * 512 * 4)); }
    struct Mode { const char *name; int mode, ldskb, iters; } modes[] = {
        {"trivial", 0, 0, 0}, {"latency", 1, 0, 0}, {"mfma(6us)", 2, 0, 220}, {"mfma(6us)+128KB LDS", 2, 128, 220}, {"mfma(2us)", 2, 0, 73}};
    for (auto &m : modes) {
        hipGraphExec_t ex[maxK];
        for (int k = 0; k < maxK; ++k) {
            hipGraph_t g;
            CK(hipStreamBeginCapture(st[k], hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < N; ++i) {
                const float *in = buf[k][i & 1]; float *out = buf[k][(i + 1) & 1];
                const unsigned salt = i * 17 + k;
                if (m.mode == 0) hipLaunchKernelGGL((k_stage<0, 0>), dim3(WG), dim3(512), 0, st[k], big, in, out, salt, 0);
                else if (m.mode == 1) hipLaunchKernelGGL((k_stage<1, 0>), dim3(WG), dim3(512), 0, st[k], big, in, out, salt, 0);
                else if (m.ldskb == 0) hipLaunchKernelGGL((k_stage<2, 0>), dim3(WG), dim3(512), 0, st[k], big, in, out, salt, m.iters);
                else hipLaunchKernelGGL((k_stage<2, 128>), dim3(WG), dim3(512), m.ldskb * 1024, st[k], big, in, out, salt, m.iters);
            }
            CK(hipStreamEndCapture(st[k], &g));
            CK(hipGraphInstantiate(&ex[k], g, nullptr, nullptr, 0));
            CK(hipGraphDestroy(g));
            CK(hipGraphLaunch(ex[k], st[k]));
        }
        CK(hipDeviceSynchronize());
        double one = 0;
        for (int K = 1; K <= maxK; ++K) {
            const int reps = 4;
            auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < reps; ++r) for (int k = 0; k < K; ++k) CK(hipGraphLaunch(ex[k], st[k]));
            CK(hipDeviceSynchronize());
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
            if (K == 1) one = us;
            printf("%-22s K=%d: %7.2f us per dependent launch per chain, aggregate %6.2f us per launch (%.2fx one chain's rate)\n", m.name, K, us / N,
                   us / N / K, K * one / us);
        }
        for (int k = 0; k < maxK; ++k) hipGraphExecDestroy(ex[k]);
    }
    return 0;
}
