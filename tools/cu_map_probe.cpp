// Which workgroups of a launch share a compute unit?  (speed-only knowledge: HIP promises nothing about placement)
// 256-thread workgroups with 64 KB of LDS (two per CU, like conv_gemm_f32's), each records HW_ID / XCC_ID and spins for a while so
// that the first 512 stay resident together.  Prints, for the first wave of residents, how the block ids of co-resident pairs
// relate, and the order in which an XCD's CUs are filled.
// build: hipcc --offload-arch=gfx950 -O2 tools/cu_map_probe.cpp -o tools/cu_map_probe.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

__global__ __launch_bounds__(256, 2) void probe(unsigned long long *rec, int spin_ticks) {
    __shared__ float pad[16384];
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        while (wall_clock64() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(32);
        rec[blockIdx.x * 4 + 0] = t0;
        rec[blockIdx.x * 4 + 1] = hw;
        rec[blockIdx.x * 4 + 2] = xcc;
        rec[blockIdx.x * 4 + 3] = (unsigned long long)pad[5];
    }
}

int main() {
    const int nb = 1024;
    unsigned long long *d;
    hipMalloc(&d, nb * 4 * sizeof(unsigned long long));
    std::vector<unsigned long long> h(nb * 4);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 0, 0, d, 2000);   // 20 us
        hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    auto cu_of = [&](int b) {
        const unsigned hw = (unsigned)h[b * 4 + 1], xcc = (unsigned)h[b * 4 + 2] & 0xf;
        return (int)((xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15));
    };
    unsigned long long t0 = ~0ull;
    for (int b = 0; b < nb; ++b) t0 = std::min(t0, h[b * 4]);
    std::map<int, std::vector<int>> by_cu;
    int first_wave = 0;
    for (int b = 0; b < nb; ++b)
        if (h[b * 4] - t0 < 500) { by_cu[cu_of(b)].push_back(b); ++first_wave; }   // started within 5 us of the first
    printf("blocks started within 5 us: %d on %zu CUs\n", first_wave, by_cu.size());
    std::map<int, int> diff_hist;
    int n_xcd_mismatch = 0;
    for (auto &kv : by_cu) {
        auto &v = kv.second;
        for (size_t i = 1; i < v.size(); ++i) diff_hist[v[i] - v[0]]++;
        for (int b : v) if ((b & 7) != (kv.first >> 8)) ++n_xcd_mismatch;
    }
    printf("co-resident pairs, block id difference -> count:");
    for (auto &kv : diff_hist) printf("  %d:%d", kv.first, kv.second);
    printf("\nblocks whose id %% 8 is not their XCC id: %d\n", n_xcd_mismatch);
    printf("XCD 0: block id -> (se, sh, cu, wave slot):\n");
    for (int b = 0; b < 8 * 70; b += 8) {
        const unsigned hw = (unsigned)h[b * 4 + 1];
        printf("  %4d:(%u,%u,%2u,w%u)%s", b, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, hw & 15, (b / 8) % 6 == 5 ? "\n" : "");
    }
    printf("\n");
    return 0;
}
