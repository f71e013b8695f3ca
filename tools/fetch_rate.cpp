// Per-CU operand fetch rate on MI355X for the two access shapes a K-split MFMA tile can use (tuning aid, no product code):
//   strided     lane (i = lane & 15, g = lane >> 4) reads 16 B at row (r0 + i), column chunk g of a row-major matrix: one wave
//               instruction touches 16 rows x 64 B (what skinny_gemm_f32 does today for both operands)
//   contiguous  lane l reads 16 B at base + 16 l: one wave instruction = 1 KB contiguous (pre-tiled operand)
// Each workgroup (8 waves) pulls `KB` KB into registers (loads issued back to back, 24 per wave in flight) and exits.
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_rate.cpp -o tools/fetch_rate.bin && ./tools/fetch_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <bool STRIDED>
__global__ __launch_bounds__(512) void pull(const float *__restrict__ buf, long wg_stride, int loads_per_wave, int ld, float *sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *base = buf + (long)blockIdx.x * wg_stride;
    f32x4 acc = {0, 0, 0, 0};
    // a wave owns loads_per_wave KB: strided = 16 rows x (loads_per_wave * 64 B) per row block of the matrix
    for (int u0 = 0; u0 < loads_per_wave; u0 += 24) {
        f32x4 v[24];
#pragma unroll
        for (int u = 0; u < 24; ++u) {
            const int uu = u0 + u;
            if (uu < loads_per_wave) {
                const float *p;
                if (STRIDED) p = base + (long)(wave * 16 + (lane & 15)) * ld + uu * 16 + (lane >> 4) * 4;
                else p = base + ((long)(wave * loads_per_wave + uu) * 64 + lane) * 4;
                v[u] = *reinterpret_cast<const f32x4 *>(p);
            } else v[u] = f32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < 24; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[threadIdx.x] = acc[0];
}

int main() {
    const int KBs[] = {96, 192};
    const int WGs[] = {256, 512, 1024};
    float *buf, *sink;
    const size_t total = 512ull << 20;
    CK(hipMalloc(&buf, total)); CK(hipMemset(buf, 0, total)); CK(hipMalloc(&sink, 4096));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int shared = 0; shared < 2; ++shared)
        for (int kb : KBs)
            for (int nwg : WGs)
                for (int strided = 0; strided < 2; ++strided) {
                    const int lpw = kb / 8;                 // 1 KB per wave load, 8 waves
                    const int ld = lpw * 16;                // floats per row for the strided shape (each wave: 16 rows x lpw*64 B)
                    const long wg_stride = shared ? 0 : (long)kb * 256;   // floats
                    auto go = [&]() {
                        if (strided) hipLaunchKernelGGL(pull<true>, dim3(nwg), dim3(512), 0, 0, buf, wg_stride, lpw, ld, sink);
                        else hipLaunchKernelGGL(pull<false>, dim3(nwg), dim3(512), 0, 0, buf, wg_stride, lpw, ld, sink);
                    };
                    go(); CK(hipDeviceSynchronize());
                    CK(hipEventRecord(a, 0));
                    const int it = 20;
                    for (int i = 0; i < it; ++i) go();
                    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
                    float ms; CK(hipEventElapsedTime(&ms, a, b));
                    const double us = ms * 1e3 / it, bytes = (double)nwg * kb * 1024;
                    printf("%s %3d KB/wg %4d wgs %-10s %7.2f us  %6.2f TB/s  %5.1f B/clk/CU (2.4 GHz, 256 CUs)\n", shared ? "same-data " : "distinct  ", kb, nwg,
                           strided ? "strided" : "contiguous", us, bytes / us * 1e-6, bytes / us / 1e3 / 2.4 / 256);
                }
    return 0;
}
