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
        else if (p.act == 3) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
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

// ---- the same epilogue with the stores (and the residual loads) COALESCED through LDS ----
// In the register form above a wave's store instruction touches 32 different output rows with 32 bytes each: the texture path takes
// it apart row by row, and with every workgroup of a round storing at the same time a 128 x 128 tile spent 14-24 us ISSUING its 64 KB
// (tools/ring_trace.py, round 5: the per-tile fixed time that two rounds of loop work had not found).  Here each wave passes its
// accumulators through a private 32-row slab of LDS — 32 x (32 TN) floats, the 16-byte chunk c of row r at position c ^ (r & 7):
// conflict-free for the ds_write_b128 of the MFMA layout (8 consecutive rows, one chunk) and for the ds_read_b128 of the row-major
// read-back (whole rows) — and stores whole rows: 4 rows x 256 B (TN = 2) or 8 rows x 128 B (TN = 1) per instruction, a quarter of the
// instructions.  Bias, residual and activation are applied to the same values in the same order as above: the same bits.
// Needs 16-byte aligned rows (out, residual, bias) and N % 4 == 0: the caller checks conv_tile_staged_ok and otherwise takes the
// register form.  All waves must be past their last read of the operand tiles before the slab (which overlays them) is written.
__device__ __forceinline__ bool conv_tile_staged_ok(const ConvParams &p, const ConvGroup &g, const ConvTilePtrs &t) {
    const bool out_ok = ((p.ldo | g.out_col0 | p.N) & 3) == 0 && (reinterpret_cast<uintptr_t>(t.out) & 15) == 0;
    const bool res_ok = !t.res || ((p.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(t.res) & 15) == 0);
    const bool bias_ok = !t.bias || (reinterpret_cast<uintptr_t>(t.bias) & 15) == 0;
    return out_ok && res_ok && bias_ok;
}

template <int TM, int TN>
__device__ __forceinline__ void conv_tile_epilogue_staged(const ConvParams &p, const ConvGroup &g, const ConvTilePtrs &t,
                                                          f32x16 (&acc)[TM][TN], const int mw, const int nw, const int lane,
                                                          float *slab /* this wave's 32 x 32 TN floats */) {
    static_assert(TN == 1 || TN == 2 || TN == 4, "chunks per row must be a power of two");
    constexpr int CPR = 8 * TN;        // 16-byte chunks per slab row
    constexpr int RPI = 64 / CPR;      // rows per read-back / store instruction
    constexpr int NI = 32 / RPI;
    const float *gbias = t.bias, *gres = t.res;
    float *gout = t.out;
    const int li = lane & 31, lh = lane >> 5;
    const int c4r = lane & (CPR - 1), rr = lane / CPR;
    const int n = nw + c4r * 4;
    const bool nok = n < p.N;          // N % 4 == 0: the lane's 4 channels are in or out together
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (gbias && nok) bv = *reinterpret_cast<const f32x4 *>(gbias + n);
    auto activate = [&](float v, float rvv) {
        if (gres && !p.res_after_act) v += rvv;
        if (p.act == 1) v = v >= 0.f ? v : v * 0.2f;
        else if (p.act == 2) v = v > 0.f ? v : 0.f;
        else if (p.act == 3) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        if (gres && p.res_after_act) v += rvv;
        return v;
    };
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        f32x4 rv[NI];
#pragma unroll
        for (int it = 0; it < NI; ++it) {   // the residual rows of this block row: coalesced, in flight under the LDS round trip
            const int m = mw + i * 32 + it * RPI + rr;
            rv[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (gres && nok && m < p.M) rv[it] = *reinterpret_cast<const f32x4 *>(gres + (long)m * p.ldr + n);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c4 = j * 8 + 2 * q + lh;     // channel j * 32 + 8 q + 4 lh of the slab row
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[i][j][4 * q + r];
                *(lds_f32x4 *)__builtin_assume_aligned(slab + (li * CPR + (c4 ^ (li & 7))) * 4, 16) = v;
            }
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int row = it * RPI + rr;
            const int m = mw + i * 32 + row;
            f32x4 v = *(const lds_f32x4 *)__builtin_assume_aligned(slab + (row * CPR + (c4r ^ (row & 7))) * 4, 16);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = activate(v[r] + bv[r], rv[it][r]);
            if (nok && m < p.M) *reinterpret_cast<f32x4 *>(gout + (long)m * p.ldo + g.out_col0 + n) = v;
        }
    }
}

// launchers of the LDS-DMA ring engine (conv_gemm_ring.hip); `variant` selects stage depth / ring slots / residency (tile ids 31..)
hipError_t launch_conv_gemm_ring(const ConvParams &p, int variant, hipStream_t stream);
// debug: where the trace build of the ring kernel (variant 11) writes its 8 x uint64 per workgroup (nullptr: nowhere)
hipError_t conv_ring_trace_set(unsigned long long *dev_records, int max_records);

}  // namespace ts
