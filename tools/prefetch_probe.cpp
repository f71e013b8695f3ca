// Does data touched by launch k stay in an XCD's L2 for launch k+1?  (tuning aid, no product code)
// A chain of dependent launches; launch k's workgroup b pulls its own 32 KB slice of region k (cold: 16 regions x 8 MB cycle
// through 128 MB) and optionally touches one dword per 128-byte line of the slice that workgroup (b + shift) of launch k+1
// will pull.  shift = 0: same workgroup id -> same XCD (ids are dealt round-robin over the 8 XCDs); shift = 1: a neighbour XCD.
//   hipcc --offload-arch=gfx950 -O3 tools/prefetch_probe.cpp -o tools/prefetch_probe.bin && ./tools/prefetch_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int SLICE_FLOATS = 32 * 256;   // 32 KB per workgroup

template <bool PF>
__global__ __launch_bounds__(512) void stage(const float *__restrict__ cur, const float *__restrict__ nxt, int shift, int nwg, int pfs, float *sink,
                                             unsigned long long *clk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long t0 = wall_clock64();
    const float *base = cur + (long)blockIdx.x * SLICE_FLOATS;
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4 *>(base + ((wave * 4 + u) * 64 + lane) * 4);   // 8 waves x 4 KB
    float t = 0.f;
    __builtin_amdgcn_sched_barrier(0);   // own loads go out first (vmcnt retires in order)
    float t2 = 0.f;
    if (PF) {   // straight-line: one dword per 64 B (pfs = 16) or per 32 B (pfs = 8: two loads) of the 32 KB slice
        const int b2 = (blockIdx.x + shift) % nwg;
        t = nxt[(long)b2 * SLICE_FLOATS + threadIdx.x * 16];
        if (pfs == 8) t2 = nxt[(long)b2 * SLICE_FLOATS + threadIdx.x * 16 + 8];
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc = (v[0] + v[1]) + (v[2] + v[3]);
    asm volatile("" :: "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]));   // own loads are back
    if (threadIdx.x == 0) atomicAdd(clk, wall_clock64() - t0);
    if (acc[0] + acc[1] + acc[2] + acc[3] + t + t2 == 123.456f) sink[threadIdx.x] = acc[0];
}

int main() {
    const int nwg = 256, nreg = 16;
    const size_t region = (size_t)nwg * SLICE_FLOATS;   // floats: 8 MB
    float *buf, *sink; unsigned long long *clk, hclk; CK(hipMalloc(&clk, 8));
    CK(hipMalloc(&buf, region * (nreg + 1) * 4)); CK(hipMemset(buf, 0, region * nreg * 4)); CK(hipMalloc(&sink, 4096));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int pfs : {16, 8})
    for (int mode = 0; mode < 2; ++mode) {   // 0 no prefetch, 1 prefetch same id, 2 prefetch id+1, 3 prefetch id+8 (same XCD, other CU)
        const int shift = mode == 2 ? 1 : (mode == 3 ? 8 : 0);
        auto go = [&](int k) {
            const float *cur = buf + (size_t)(k % nreg) * region, *nxt = mode ? buf + (size_t)((k + 1) % nreg) * region : nullptr;
            if (mode) hipLaunchKernelGGL(stage<true>, dim3(nwg), dim3(512), 0, 0, cur, nxt, shift, nwg, pfs, sink, clk);
            else hipLaunchKernelGGL(stage<false>, dim3(nwg), dim3(512), 0, 0, cur, nxt, shift, nwg, pfs, sink, clk);
        };
        for (int k = 0; k < 64; ++k) go(k);
        CK(hipDeviceSynchronize());
        const int it = 1600;
        CK(hipMemset(clk, 0, 8));
        CK(hipEventRecord(a, 0));
        for (int k = 0; k < it; ++k) go(k);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        CK(hipMemcpy(&hclk, clk, 8, hipMemcpyDeviceToHost));
        printf("  entry -> own loads back: %.3f us (mean over workgroups)\n", hclk * 0.01 / it / nwg);
        printf("prefetch stride %d B, mode %d (%s): %.3f us per launch\n", pfs * 4, mode, mode == 0 ? "no prefetch" : mode == 1 ? "prefetch, same workgroup id" : mode == 2 ? "prefetch, id + 1 (other XCD)" : "prefetch, id + 8 (same XCD)", ms * 1e3 / it);
    }
    for (int rep = 2; rep <= 2; rep *= 2)
    for (int sh = 0; sh < 2; ++sh) {   // every region is pulled in `rep` consecutive launches (no prefetch); sh: the repeats shift the workgroup -> slice map by one
        hipLaunchKernelGGL(stage<false>, dim3(nwg), dim3(512), 0, 0, buf, (const float *)nullptr, 0, nwg, 32, sink, clk);
        CK(hipDeviceSynchronize());
        const int it = 1600;
        CK(hipMemset(clk, 0, 8));
        for (int k = 0; k < it; ++k) {
            const int reg = (k / rep) % nreg, shift = sh ? (k % rep) : 0;
            hipLaunchKernelGGL(stage<false>, dim3(nwg), dim3(512), 0, 0, buf + (size_t)reg * region + (size_t)shift * SLICE_FLOATS, (const float *)nullptr, 0, nwg, 32, sink, clk);
        }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(&hclk, clk, 8, hipMemcpyDeviceToHost));
        printf("each region pulled %d x in a row%s: entry -> own loads back %.3f us\n", rep, sh ? " (slices shifted by one workgroup between repeats)" : "", hclk * 0.01 / it / nwg);
    }
    // reference: all launches read the same region (8 MB: fits the 8 x 4 MB of L2 when each XCD keeps only its own slices)
    {
        auto go = [&]() { hipLaunchKernelGGL(stage<false>, dim3(nwg), dim3(512), 0, 0, buf, (const float *)nullptr, 0, nwg, 32, sink, clk); };
        for (int k = 0; k < 64; ++k) go();
        CK(hipDeviceSynchronize());
        const int it = 1600;
        CK(hipMemset(clk, 0, 8));
        CK(hipEventRecord(a, 0));
        for (int k = 0; k < it; ++k) go();
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        CK(hipMemcpy(&hclk, clk, 8, hipMemcpyDeviceToHost));
        printf("  entry -> own loads back: %.3f us (mean over workgroups)\n", hclk * 0.01 / it / nwg);
        printf("same region every launch (L2-resident if L2 survives the kernel boundary): %.3f us per launch\n", ms * 1e3 / it);
    }
    return 0;
}
