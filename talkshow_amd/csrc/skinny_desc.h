// Problem descriptors of the PixelCNN chain kernels (skinny_gemm.hip: 16/32-row tiles with split-K over the waves;
// skinny_wide.hip: 64 x 64 tiles staged through LDS).  A problem is 64 dwords: every wave fetches it with ONE vector
// load (lane i holds word i) and pulls fields out with v_readlane — one round trip, no scalar-cache misses.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "kernels.h"

namespace ts {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum {   // descriptor word indices
    SD_M = 0, SD_N, SD_CNT, SD_FLAGS, SD_GATED, SD_GX, SD_GY, SD_NSEG,
    SD_W = 8, SD_LDW = 10, SD_BIAS = 11,
    SD_ADD1 = 13, SD_ADD1_STRIDE = 15, SD_ADD1_SHIFT = 16,
    SD_ADD2 = 17, SD_ADD2_STRIDE = 19, SD_ADD2_SHIFT = 20,
    SD_ADD3 = 21, SD_ADD3_STRIDE = 23,
    SD_CLS = 24, SD_CLS_LD = 26,
    SD_OUT = 27, SD_OUT_STRIDE = 29, SD_PRE = 30, SD_PRE_STRIDE = 32,
    SD_SEG = 33,   // per segment: base(2) gidx(2) row_stride gidx_stride row_shift len16
    SD_SEG_WORDS = 8,
    SD_ZERO = 57,
    SD_WTQ = 59,       // K / 16 when the weights are tiled, else 0
    SD_OUT_TW = 60,    // log2 of the tiled view width of out / pre / add1, or 0 (row-major)
    SD_PRE_TW = 61,
    SD_ADD1_TW = 62,
    SD_ITEMS = 63,     // wide kernel: 64 x 64 tiles one workgroup works through (2 for the half-K problems of a launch)
};
enum { SDF_GATE = 1, SDF_RELU = 2, SDF_PRE = 4 };
constexpr int SEG_TILED = 0x10000;   // segment word 6: SEG_TILED | (W / 16) for a tiled dense segment

// float index of element `lin` (= row * row_width + col) of a buffer tiled with view width 2^lw
__device__ __forceinline__ long tiled_index(long lin, int lw) {
    const long m = lin >> lw;
    const int k = (int)(lin - (m << lw));
    return ((((m >> 4) << (lw - 4)) + (k >> 4)) << 8) + ((int)((m & 15) + (((k & 15) >> 2) << 4)) << 2) + (k & 3);
}
struct SkinnyDesc { uint32_t w[64]; };
struct SkinnyDescBatch {
    int start[8];   // first workgroup of problem i (1-D grid over live tiles only); INT_MAX for unused problems
    SkinnyDesc d[SKINNY_MAX_PROBLEMS];
};

// descriptor pointers are rebuilt from integers: tag them as global (address space 1) so the loads are global_load, not flat
typedef __attribute__((address_space(1))) const float gcf;
typedef __attribute__((address_space(1))) float gf;
typedef __attribute__((address_space(1))) const int gci;
typedef __attribute__((address_space(1))) const f32x4 gcf4;
typedef __attribute__((address_space(1))) f32x4 gf4;

// skinny_wide.hip
// trace: start[6..7] of db hold the device pointer of this launch's record block (TS_SKINNY_TRACE)
hipError_t launch_skinny_wide(const SkinnyDescBatch &db, int workgroups, hipStream_t stream, bool trace);

}  // namespace ts
