// Pieces shared by the two engines of conv_gemm_f32 (conv_gemm.hip: global -> VGPR -> LDS staging; conv_gemm_ring.hip: LDS-DMA
// ring): the per-problem operand pointers of a tile and the epilogue.  One definition, so that both engines store the same bits.
#pragma once
#include "kernels.h"

namespace ts {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;   // LDS-qualified: volatile accesses must not fall back to flat

struct ConvTilePtrs {
    const float *x, *w, *bias, *res;
    float *out;
};

// pointers of problem / group `zidx` (batched problems shift every pointer by the problem's offsets)
__device__ __forceinline__ ConvTilePtrs conv_tile_ptrs(const ConvParams &p, const ConvGroup &g, const int zidx) {
    ConvTilePtrs t{g.x, g.w, g.bias, g.res, g.out};
    if (p.zdiv > 0) {
        const int z0 = zidx / p.zdiv, z1 = zidx - z0 * p.zdiv;
        t.x += z0 * p.x_zs0 + z1 * p.x_zs1;
        t.w += z0 * p.w_zs0 + z1 * p.w_zs1;
        t.out += z0 * p.o_zs0 + z1 * p.o_zs1;
        if (t.bias) t.bias += z1 * p.b_zs1;
        if (t.res) t.res += z0 * p.r_zs0 + z1 * p.r_zs1;
    }
    return t;
}

// ---- epilogue: bias (+ residual) + activation, masked store ----
// The MFMA operands are swapped (weights as A, activations as B: the same products in the same k order, the same bits), so an
// accumulator block holds D[channel][row]: lane (li, lh) owns output row m = li and, per group g of 4 registers, the 4
// CONSECUTIVE channels 8 g + 4 lh .. + 3 — bias, residual and output move as 16-byte vectors (4 stores per 32 x 32 block
// instead of 16; the residual values of a block are fetched together, ahead of their use).  Rows / buffers that are not
// 16-byte aligned (the 39- / 90- / 129-wide pose rows) and channel tails take the scalar form.
// (mw, nw): first row / column of this wave's TM x TN blocks of 32 x 32.
template <int TM, int TN>
__device__ __forceinline__ void conv_tile_epilogue(const ConvParams &p, const ConvGroup &g, const ConvTilePtrs &t,
                                                   f32x16 (&acc)[TM][TN], const int mw, const int nw, const int li, const int lh) {
    const float *gbias = t.bias, *gres = t.res;
    float *gout = t.out;
    const bool vec_out = ((p.ldo | g.out_col0) & 3) == 0 && (reinterpret_cast<uintptr_t>(gout) & 15) == 0;
    const bool vec_res = gres && (p.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(gres) & 15) == 0;
    const bool vec_bias = gbias && (reinterpret_cast<uintptr_t>(gbias) & 15) == 0;
    auto activate = [&](float v, float rvv) {
        if (gres && !p.res_after_act) v += rvv;
        if (p.act == 1) v = v >= 0.f ? v : v * 0.2f;
        else if (p.act == 2) v = v > 0.f ? v : 0.f;
        else if (p.act == 3) v = gelu_fast(v);   // kernels.h
        if (gres && p.res_after_act) v += rvv;
        return v;
    };
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = mw + i * 32 + li;
        const bool mok = m < p.M;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nb0 = nw + j * 32 + 4 * lh;
            f32x4 rv[4], bv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nb = nb0 + 8 * q;
                const bool full = nb + 3 < p.N;
                bv[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (gbias) {
                    if (vec_bias && full) bv[q] = *reinterpret_cast<const f32x4 *>(gbias + nb);
                    else
#pragma unroll
                        for (int r = 0; r < 4; ++r) bv[q][r] = nb + r < p.N ? gbias[nb + r] : 0.f;
                }
                rv[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (gres && mok) {
                    const float *rp = gres + (long)m * p.ldr + nb;
                    if (vec_res && full) rv[q] = *reinterpret_cast<const f32x4 *>(rp);
                    else
#pragma unroll
                        for (int r = 0; r < 4; ++r) rv[q][r] = nb + r < p.N ? rp[r] : 0.f;
                }
            }
            if (mok) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int nb = nb0 + 8 * q;
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = activate(acc[i][j][4 * q + r] + bv[q][r], rv[q][r]);
                    float *op = gout + (long)m * p.ldo + g.out_col0 + nb;
                    if (vec_out && nb + 3 < p.N) *reinterpret_cast<f32x4 *>(op) = v;
                    else
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (nb + r < p.N) op[r] = v[r];
                }
            }
        }
    }
}

// the LDS-DMA ring engine (conv_gemm_ring.hip); variant 1 = 128 x 128 on 4 waves of 64 x 64, 9 = 128 x 128 on 8 waves of 32 x 64, 3 = 96 x 128 on 4 waves
// of 96 x 32 (tile ids 31 / 39 / 33), 0 = 9 or 3 by the layer's tile count (conv_gemm_ring_pick)
hipError_t launch_conv_gemm_ring(const ConvParams &p, int variant, hipStream_t stream);
hipError_t launch_conv_gemm_ring_banded(const ConvParams &p, const ConvBands &bd, hipStream_t stream);   // tile id 37: bands (plan_bands) + dealt tiles
bool conv_gemm_ring_takes(const ConvParams &p);   // host: every segment a multiple of the 32-deep stage
int conv_gemm_ring_pick(const ConvParams &p, const ConvBands *bd, const ConvSK *sk);   // host: the plan by tile count: 9 (128 x 128), 3 (96 x 128), 7 (bands, if `bd` is given) or 8 (stream-K band, if `sk` is given)

// grouped many-tap convolution with 48 channels per group (conv_taps48.hip: the wav2vec2 positional convolution); tile id 48
hipError_t launch_conv_taps48(const ConvParams &p, hipStream_t stream);
bool conv_taps48_takes(const ConvParams &p);

}  // namespace ts
