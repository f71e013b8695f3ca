// Micro-benchmark for the one lever on the chain that was never measured: letting stage n+1's workgroups start BEFORE
// stage n has finished — its launch, descriptor decode and weight prefetch run under stage n's MFMAs and tail — with the true
// dependency (stage n's outputs) enforced inside the kernel by an arrival counter instead of by the kernel boundary.
//
//   boundary   N dependent launches in one hipGraph chain (what the library does today)
//   overlap    the same kernels as two interleaved chains of one graph (edges n -> n + 2 only): stage n + 1 is dispatched while stage n
//              runs (both fit: 64 KB of LDS, 256 threads per workgroup, 2 per CU), prefetches its weights, then lane 0 polls stage n's
//              arrival counter (relaxed, agent scope, bounded) and the workgroup reads stage n's outputs with sc1 loads; outputs
//              are written with 16-byte sc0 sc1 (write-through) stores, drained, then the arrival is counted.
//
// A stage = 256 workgroups; each streams `wkb` KB of cold weights (independent of its predecessor), reads 64 KB of the predecessor's
// 1 MB output (written by 16 different workgroups), runs `iters` dependent v_mfma_f32_16x16x4_f32 and writes 4 KB.  Every value read
// is CHECKED against the stage number, so a stale or early read poisons the result, which is verified at the end.
// Build: hipcc --offload-arch=gfx950 -O3 tools/overlap_probe.cpp -o tools/overlap_probe.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int G = 256, T = 256, ACT_F4 = G * T;   // activation buffer: one f32x4 per thread of the stage = 1 MB

__device__ __forceinline__ f32x4 load_sc1(const f32x4 *p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void store_sc1(f32x4 *p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }

template <int MODE>
__global__ __launch_bounds__(T) void k_stage(const f32x4 *__restrict__ W, const f32x4 *act_in, f32x4 *act_out, unsigned *counters, int n, int wkb,
                                             int iters, unsigned target, unsigned *err, unsigned long long *ts) {
    constexpr bool FLAG = MODE != 0;
    extern __shared__ float lds[];
    const int b = blockIdx.x, t = threadIdx.x;
    if (ts && b == 0 && t == 0) ts[n * 4 + 0] = wall_clock64();
    // ---- prologue: nothing here depends on the predecessor — this stage's weights (cold: 16 regions of the buffer take turns) ----
    const int nw = wkb * 1024 / 16 / T;   // f32x4 per thread
    const f32x4 *wp = W + ((size_t)(n & 15) * G + b) * (size_t)(wkb * 64) + t;
    f32x4 ws = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < nw; ++i) ws += wp[(size_t)i * T];
    lds[t] = ws[0] + ws[1] + ws[2] + ws[3];
    // ---- the dependency ----
    if (FLAG && n > 0) {
        if (t == 0) {
            int spins = 0;
            while (__hip_atomic_load(&counters[n - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1 << 14)) { atomicAdd(err, 1u); break; }   // bounded: a lost arrival must not hang the box
            }
            if (ts && b == 0) { ts[n * 4 + 1] = wall_clock64(); ts[n * 4 + 3] = spins; }
            if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    __syncthreads();
    const f32x4 *ap = act_in + (size_t)(b & 15) * (ACT_F4 / 16) + t;
    f32x4 a[16];
    if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = load_sc1(ap + i * T);
        // the loads are invisible to the compiler's vmcnt bookkeeping: tie every destination to the wait
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]),
                       "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
                     :: "memory");
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = ap[i * T];
    }
    bool ok = true;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        ok = ok && a[i][0] == (float)n && a[i][1] == (float)n && a[i][2] == (float)n && a[i][3] == (float)n;
        s += a[i][0];
    }
    // ---- the stage's work at the matrix pipe's rate ----
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float v = s * 1e-30f + lds[(t + 1) & (T - 1)] * 1e-30f;
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v, 1.0f, acc, 0, 0, 0);
    const float o = ok ? (float)(n + 1) + acc[0] * 1e-30f : -1e30f;
    const f32x4 ov = {o, o, o, o};
    f32x4 *op = act_out + (size_t)b * T + t;
    if (MODE == 1) {
        store_sc1(op, ov);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) __hip_atomic_fetch_add(&counters[n], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (MODE == 2) {
        *op = ov;
        __syncthreads();
        if (t == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(&counters[n], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        *op = ov;
    }
    if (ts && b == 0 && t == 0) ts[n * 4 + 2] = wall_clock64();
}

template <int MODE>
static void launch(hipStream_t s, const f32x4 *W, const f32x4 *in, f32x4 *out, unsigned *c, int n, int wkb, int iters, unsigned *err, unsigned long long *ts) {
    hipLaunchKernelGGL((k_stage<MODE>), dim3(G), dim3(T), 64 * 1024, s, W, in, out, c, n, wkb, iters, (unsigned)G, err, ts);
}

int main(int argc, char **argv) {
    const int N = 400, maxK = 3;
    f32x4 *W; CK(hipMalloc(&W, (size_t)256 << 20)); CK(hipMemset(W, 0, (size_t)256 << 20));
    hipStream_t sA[maxK], sB[maxK];
    f32x4 *act[maxK][2];
    unsigned *counters[maxK], *err;
    unsigned long long *ts;
    CK(hipMalloc(&err, 4)); CK(hipMalloc(&ts, N * 32));
    hipEvent_t ev[maxK], evB[maxK];
    for (int k = 0; k < maxK; ++k) {
        CK(hipStreamCreateWithFlags(&sA[k], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sB[k], hipStreamNonBlocking));
        for (int j = 0; j < 2; ++j) CK(hipMalloc(&act[k][j], (size_t)ACT_F4 * 16));
        CK(hipMalloc(&counters[k], N * 4));
        CK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&evB[k], hipEventDisableTiming));
    }
    struct Shape { const char *name; int wkb, iters; } shapes[] = {{"wide-like (64 KB weights, ~6 us MFMA)", 64, 220}, {"small (16 KB weights, ~1 us MFMA)", 16, 37},
                                                                    {"no weights, no MFMA (hand-off only)", 0, 0}};
    // variants: 0 boundary | 1 overlap, sc1 stores + loads | 2 overlap, plain + release / acquire fences | 3, 4: the kernels of 1, 2 on ONE stream
    // (their recipe's price with kernel boundaries still in place: no overlap)
    const char *vname[] = {"boundary", "overlap/sc1", "overlap/fence", "1-stream/sc1", "1-stream/fence"};
    for (auto &sh : shapes) {
        for (int var = 0; var < 5; ++var) {
            const int mode = var == 0 ? 0 : (var == 1 || var == 3 ? 1 : 2);
            const bool two = var == 1 || var == 2;
            hipGraphExec_t exA[maxK], exB[maxK];
            for (int k = 0; k < maxK; ++k) {
                for (int half = 0; half < (two ? 2 : 1); ++half) {
                    hipGraph_t g;
                    hipStream_t s = half ? sB[k] : sA[k];
                    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                    for (int n = half; n < N; n += two ? 2 : 1) {
                        unsigned long long *tsp = k == 0 ? ts : nullptr;
                        if (mode == 0) launch<0>(s, W, act[k][n & 1], act[k][(n + 1) & 1], counters[k], n, sh.wkb, sh.iters, err, tsp);
                        else if (mode == 1) launch<1>(s, W, act[k][n & 1], act[k][(n + 1) & 1], counters[k], n, sh.wkb, sh.iters, err, tsp);
                        else launch<2>(s, W, act[k][n & 1], act[k][(n + 1) & 1], counters[k], n, sh.wkb, sh.iters, err, tsp);
                    }
                    CK(hipStreamEndCapture(s, &g));
                    CK(hipGraphInstantiate(half ? &exB[k] : &exA[k], g, nullptr, nullptr, 0));
                    CK(hipGraphDestroy(g));
                }
            }
            auto run = [&](int K) -> int {
                for (int k = 0; k < K; ++k) {
                    CK(hipMemsetAsync(counters[k], 0, N * 4, sA[k]));
                    CK(hipMemsetAsync(act[k][0], 0, (size_t)ACT_F4 * 16, sA[k]));   // stage 0 expects zeros
                    CK(hipEventRecord(ev[k], sA[k]));
                    if (two) CK(hipStreamWaitEvent(sB[k], ev[k], 0));
                }
                for (int k = 0; k < K; ++k) {
                    CK(hipGraphLaunch(exA[k], sA[k]));
                    if (two) CK(hipGraphLaunch(exB[k], sB[k]));
                }
                return 0;
            };
            if (run(maxK)) return 1;
            CK(hipDeviceSynchronize());
            for (int K = 1; K <= maxK; K += 2) {
                CK(hipMemset(err, 0, 4));
                const int reps = 3;
                double us = 0;
                for (int r = 0; r < reps; ++r) {
                    auto t0 = std::chrono::steady_clock::now();
                    if (run(K)) return 1;
                    CK(hipDeviceSynchronize());
                    us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
                }
                std::vector<float> h((size_t)ACT_F4 * 4);
                long bad = 0;
                for (int k = 0; k < K; ++k) {
                    CK(hipMemcpy(h.data(), act[k][N & 1], h.size() * 4, hipMemcpyDeviceToHost));
                    for (float x : h) bad += x != (float)N;
                }
                unsigned e = 0; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
                printf("%-40s %-15s K=%d: %6.2f us per stage per chain, aggregate %6.2f | wrong words %ld, poll timeouts %u\n", sh.name, vname[var],
                       K, us / N, us / N / K, bad, e);
                if (K == 1 && var <= 2) {   // timeline of workgroup 0 of a few stages (100 MHz ticks): start, poll done, end, spins
                    std::vector<unsigned long long> hts(N * 4);
                    CK(hipMemcpy(hts.data(), ts, N * 32, hipMemcpyDeviceToHost));
                    for (int n = 200; n < 204; ++n) {
                        printf("    stage %d, workgroup 0: starts %+6.2f us after stage %d's workgroup 0 ended", n, ((double)hts[n * 4] - (double)hts[(n - 1) * 4 + 2]) / 100.0, n - 1);
                        if (var) printf(", counter complete +%.2f us (%llu polls)", ((double)hts[n * 4 + 1] - (double)hts[n * 4]) / 100.0, hts[n * 4 + 3]);
                        printf(", ends +%.2f us\n", ((double)hts[n * 4 + 2] - (double)hts[n * 4]) / 100.0);
                    }
                }
            }
            for (int k = 0; k < maxK; ++k) { (void)hipGraphExecDestroy(exA[k]); if (two) (void)hipGraphExecDestroy(exB[k]); }
        }
    }
    return 0;
}
