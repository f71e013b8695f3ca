// Micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 per SIMD — one or two waves per SIMD, 2 or 4 accumulators per wave,
// with and without ds_read_b128 traffic between the MFMAs (the inner loop of the chain's wide kernel).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.cpp -o tools/mfma_rate.bin && tools/mfma_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool LDSR>
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *cyc, int iters) {
    __shared__ f32x4 lds[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = f32x4{1.f, 0.5f, 0.25f, 2.f} * (float)(i & 7);
    __syncthreads();
    f32x4 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 x = lds[lane], w0 = lds[64 + lane], w1 = lds[128 + lane];
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        f32x4 nx = x, nw0 = w0, nw1 = w1;
        if (LDSR) {
            const int o = ((it & 15) * 192 + lane) & 4095;
            nx = lds[o]; nw0 = lds[(o + 64) & 4095]; nw1 = lds[(o + 128) & 4095];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int a = 0; a < NACC; ++a)
                acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32((a & 1) ? w1[e] : w0[e], x[e], acc[a], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int a = 0; a < NACC; ++a)
                acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32((a & 1) ? w0[e] : w1[e], x[e], acc[a], 0, 0, 0);
        x = nx; w0 = nw0; w1 = nw1;
    }
    const unsigned long long t1 = clock64();
    f32x4 s = acc[0];
#pragma unroll
    for (int a = 1; a < NACC; ++a) s += acc[a];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, bool LDSR>
void run(const char *name, int threads) {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    k<NACC, LDSR><<<256, threads>>>(out, cyc, iters);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    k<NACC, LDSR><<<256, threads>>>(out, cyc, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += v; mean /= 256;
    const double mfma_per_wave = (double)iters * 8 * NACC, waves_per_simd = threads / 256.0;
    printf("%-34s %3d threads: %.1f cycles per MFMA per SIMD (%.1f per wave), %.1f TFLOP/s\n", name, threads,
           mean / (mfma_per_wave * waves_per_simd), mean / mfma_per_wave, 256.0 * (threads / 64) * mfma_per_wave * 2048 / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(cyc);
}

int main() {
    run<2, false>("2 acc, registers only", 256);
    run<4, false>("4 acc, registers only", 256);
    run<2, false>("2 acc, registers only", 512);
    run<4, false>("4 acc, registers only", 512);
    run<2, true>("2 acc + 3 ds_read_b128 / 16 MFMA", 256);
    run<4, true>("4 acc + 3 ds_read_b128 / 32 MFMA", 256);
    run<2, true>("2 acc + 3 ds_read_b128 / 16 MFMA", 512);
    run<4, true>("4 acc + 3 ds_read_b128 / 32 MFMA", 512);
    return 0;
}
