// Micro-benchmark: a barrier + data exchange among the 32 workgroups of ONE XCD of a persistent kernel (256 workgroups, 8 XCDs),
// done entirely in that XCD's L2: arrival counter bumped / polled with L2-executed atomics (no sc1), data published with plain
// stores (write-through to L2, s_waitcnt vmcnt(0) before arriving) and read after an L1 invalidate (buffer_inv sc0).
// Compared with the same exchange through agent-scope atomics + fences (what a grid-wide barrier needs: tools/grid_barrier.cpp).
//   hipcc --offload-arch=gfx950 -O3 tools/xcd_barrier.cpp -o tools/xcd_barrier.bin && tools/xcd_barrier.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Sync {
    unsigned slot_ctr[8][32];   // [xcd][0]: slot allocator (own 128-byte line)
    unsigned bar_ctr[8][32];    // [xcd][0]: arrival counter
    unsigned abort_flag, pad[31];
};

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
__device__ __forceinline__ unsigned l2_add_ret(unsigned *p, unsigned v) {   // executed in the XCD's L2 (sc1 = 0), returns the old value
    unsigned r;
    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p), "v"(v) : "memory");
    return r;
}
__device__ __forceinline__ void l2_add(unsigned *p, unsigned v) {
    asm volatile("global_atomic_add %0, %1, off" : : "v"(p), "v"(v) : "memory");
}

template <int MODE>
__device__ __forceinline__ bool xcd_barrier(Sync *s, unsigned xcd, unsigned target) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores are in L2
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        unsigned *c = &s->bar_ctr[xcd][0];
        const unsigned long long t0 = wall_clock64();
        if (MODE == 0) {
            l2_add(c, 1u);
            while (l2_add_ret(c, 0u) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > 20000000ull) { s->abort_flag = 1; ok = false; break; }
            }
        } else {
            __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > 20000000ull) { s->abort_flag = 1; ok = false; break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    __syncthreads();
    if (MODE == 0) asm volatile("buffer_inv sc0" ::: "memory");   // L1 invalidate: later loads come from L2
    return ok;
}

template <int MODE>
__global__ __launch_bounds__(512) void k_persist(float *buf, Sync *s, int iters, int *errs, unsigned *xcc_of_wg) {
    __shared__ unsigned sh[2];
    const int t = threadIdx.x;
    if (t == 0) {
        const unsigned x = xcc_id();
        sh[0] = x;
        sh[1] = __hip_atomic_fetch_add(&s->slot_ctr[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        xcc_of_wg[blockIdx.x] = x;
    }
    __syncthreads();
    const unsigned xcd = sh[0], slot = sh[1];
    if (slot >= 32) return;   // more than 32 workgroups landed on this XCD: not the layout this test is about
    int bad = 0;
    for (int i = 0; i < iters; ++i) {
        float *w = buf + ((size_t)(i & 1) * 8 + xcd) * 32 * 512;
        const float v = (float)(i * 7 + slot);
        if (MODE == 0) w[slot * 512 + t] = v;
        else {
            w[slot * 512 + t] = v;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        }
        if (!xcd_barrier<MODE>(s, xcd, (unsigned)(i + 1) * 32)) return;
        const int src = (slot + 1 + (i % 7) * 3) % 32;
        const float r = w[src * 512 + t];
        if (r != (float)(i * 7 + src)) ++bad;
    }
    if (bad) atomicAdd(errs, bad);
}

int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float *buf; Sync *s; int *errs; unsigned *xw;
    CK(hipMalloc(&buf, 2 * 8 * 32 * 512 * 4)); CK(hipMalloc(&s, sizeof(Sync))); CK(hipMalloc(&errs, 4)); CK(hipMalloc(&xw, 256 * 4));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 4000;
    for (int mode = 0; mode < 2; ++mode) {
        CK(hipMemsetAsync(s, 0, sizeof(Sync), st)); CK(hipMemsetAsync(errs, 0, 4, st)); CK(hipMemsetAsync(buf, 0, 2 * 8 * 32 * 512 * 4, st));
        hipEventRecord(a, st);
        if (mode == 0) hipLaunchKernelGGL(k_persist<0>, dim3(256), dim3(512), 0, st, buf, s, iters, errs, xw);
        else hipLaunchKernelGGL(k_persist<1>, dim3(256), dim3(512), 0, st, buf, s, iters, errs, xw);
        hipEventRecord(b, st);
        CK(hipEventSynchronize(b));
        float ms; hipEventElapsedTime(&ms, a, b);
        Sync hs; int he; unsigned hx[256];
        CK(hipMemcpy(&hs, s, sizeof(hs), hipMemcpyDeviceToHost)); CK(hipMemcpy(&he, errs, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hx, xw, sizeof(hx), hipMemcpyDeviceToHost));
        int rr = 0; for (int i = 0; i < 256; ++i) rr += hx[i] == hx[i % 8];
        printf("mode %d (%s): %.3f us per barrier + exchange, errors %d, abort %u; workgroups per XCD:", mode,
               mode == 0 ? "XCD-local: L2 atomics, buffer_inv sc0" : "agent-scope atomics + fences", ms * 1e3 / iters, he, hs.abort_flag);
        for (int x = 0; x < 8; ++x) printf(" %u", hs.slot_ctr[x][0]);
        printf("; wg i on the XCD of wg i %% 8: %d / 256; xcc of wg 0..7:", rr);
        for (int i = 0; i < 8; ++i) printf(" %u", hx[i]);
        printf("\n");
    }
    return 0;
}
