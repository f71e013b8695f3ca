// conv_gemm_split — the same implicit-GEMM convolution as conv_gemm.hip on the bf16 matrix cores with SPLIT fp32 operands:
// an OPT-IN arithmetic plan for the tolerance-only GEMMs of the face generator (wav2vec2 feature convolutions, projections,
// transformer-block GEMMs, LN-conv heads; reference: nets/spg/wav2vec.py:76-143, nets/spg/s2g_face.py:196-224).  Never used by
// the body path (bit-exact code indices) and never the default.
//
//   x = x0 + x1 (+ x2),  x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)          (8 + 8 (+ 8) mantissa bits)
//   x y ~= x0 y0 + x0 y1 + x1 y0                      NP = 2: 3 products, relative error ~2^-16 per product
//   x y ~= ... + x0 y2 + x1 y1 + x2 y0                NP = 3: 6 products, relative error ~2^-23: fp32 grade
// Products of bf16 values are exact in fp32 and v_mfma_f32_32x32x16_bf16 accumulates in fp32 at 16x the rate of the fp32
// MFMA, so the matrix pipe needs 3/16 or 6/16 of conv_gemm_f32's time.  tools/split_bf16_study.py (CPU emulation on the
// reference golden clip): NP = 2 -> hidden state within 7e-5, output within 2e-5 of the reference; NP = 3 -> 4e-6 / 1e-6.
// Activations stay fp32 in HBM: the split happens between the global-load registers and the LDS planes (v_cvt_pk_bf16_f32 +
// one subtraction per extra plane), so layers need not agree on a plan.  Weights are constants: for the 3-product plan they are split
// once, when the plan is selected, into plane images this kernel copies into LDS (BPRE, launch_split_weight_planes at the end of the
// file).  Same segments / taps as conv_gemm_f32; its own tile order (split_tile_of, kernels.h) and GELU (split_erff) — round 4's
// counters on this kernel and what each step bought are in profiles/r04_notes/split_bound.txt.
#include "kernels.h"
#include <cstdint>

namespace ts {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;   // LDS-qualified: volatile accesses must not fall back to flat

// erf for this kernel's GELU epilogue: the odd rational x P(x^2) / Q(x^2) on [-4, 4] (the single-precision form of Eigen / XLA), 12 fused
// multiply-adds + one v_rcp_f32 against ~40 instructions of erff.  Relative error < 5e-7 against float64 erf — a thirtieth of what the three-product
// split itself leaves (1.7e-5 on the output).  conv_gemm_f32 keeps libm's erff: there the epilogue hides under the other workgroup's MFMAs
// (measured neutral, tools/experiments/README.md); here 64 erf per lane are as many VALU instructions as a third of a K = 768 tile's main loop,
// and the kernel is bound by instruction issue.
__device__ __forceinline__ float split_erff(float x) {
    x = fminf(fmaxf(x, -4.0f), 4.0f);
    const float x2 = x * x;
    float p = -2.72614225801306e-10f;
    p = fmaf(p, x2, 2.77068142495902e-08f);
    p = fmaf(p, x2, -2.10102402082508e-06f);
    p = fmaf(p, x2, -5.69250639462346e-05f);
    p = fmaf(p, x2, -7.34990630326855e-04f);
    p = fmaf(p, x2, -2.95459980854025e-03f);
    p = fmaf(p, x2, -1.60960333262415e-02f);
    float q = -1.45660718464996e-05f;
    q = fmaf(q, x2, -2.13374055278905e-04f);
    q = fmaf(q, x2, -1.68282697438203e-03f);
    q = fmaf(q, x2, -7.37332916720468e-03f);
    q = fmaf(q, x2, -1.42647390514189e-02f);
    return (p * x) * __builtin_amdgcn_rcpf(q);
}

// Scheduling hints for one half of a chunk (see the main loop): behind each of the MF MFMAs its share of the VALU work, of the NDS LDS
// operations and one of the NVM global loads.  sched_group_barrier wants literal arguments, hence the recursion.
template <int MASK, int N>
__device__ __forceinline__ void sched_group() {
    if constexpr (N > 0) __builtin_amdgcn_sched_group_barrier(MASK, N, 0);
}
template <int MF, int VPM, int NDS, int NVM, int K = 0>
__device__ __forceinline__ void deal_hints() {
    if constexpr (K < MF) {
        sched_group<0x008, 1>();                                  // one MFMA
        sched_group<0x002, VPM>();                                // VALU
        sched_group<0x080, (K + 1) * NDS / MF - K * NDS / MF>();  // LDS
        sched_group<0x020, (K < NVM ? 1 : 0)>();                  // global load
        deal_hints<MF, VPM, NDS, NVM, K + 1>();
    }
}

template <int BM, int BN, int WM, int WN, int NP, bool BPRE>
__global__ __launch_bounds__(256) void conv_gemm_split_kernel(const ConvParams p) {
    static_assert(!BPRE || NP == 2, "plane images hold two planes");
    constexpr int BK = 32;
    constexpr int LDS_LD = 20;       // dwords per LDS row: 32 bf16 (16 dwords) + 4 of padding -> 80 B pitch, conflict-free for ds_read_b128
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int PA = BM / 32, PB = BN / 32;   // 32-row load passes per operand
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");

    // [buffer][plane][row][k]: plane 0 = bf16(x), 1 = bf16(x - plane 0), 2 = bf16(x - plane 0 - plane 1)
    __shared__ __attribute__((aligned(16))) uint32_t As[2][NP][BM][LDS_LD];
    __shared__ __attribute__((aligned(16))) uint32_t Bs[2][NP][BN][LDS_LD];

    const ConvGroup &g = p.g[p.zdiv > 0 ? 0 : blockIdx.z];
    const float *gx = g.x, *gw = g.w, *gbias = g.bias, *gres = g.res;
    float *gout = g.out;
    if (p.zdiv > 0) {   // batched problems: shift every pointer by this problem's offsets
        const int z0 = blockIdx.z / p.zdiv, z1 = blockIdx.z - z0 * p.zdiv;
        gx += z0 * p.x_zs0 + z1 * p.x_zs1;
        gw += z0 * p.w_zs0 + z1 * p.w_zs1;
        gout += z0 * p.o_zs0 + z1 * p.o_zs1;
        if (gbias) gbias += z1 * p.b_zs1;
        if (gres) gres += z0 * p.r_zs0 + z1 * p.r_zs1;
    }
    const long ldw = p.ldw > 0 ? p.ldw : p.Ktot;
    const int w_rows = p.w_rows > 0 ? p.w_rows : 0x7fffffff;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    // Which tile?  With three bf16 products per fp32 product this kernel moves its operands 2.4 x faster than conv_gemm_f32, and the
    // counters (profiles/r04_notes/split_bound.txt) show what that ran into: 7.6 TB/s over the fabric, L2 hit rate 58 % — in the plain
    // grid a tile's neighbours along N sit on other XCDs and every XCD's L2 fetches its own copy of the operands.  So the launcher asks for a
    // 1-D grid: workgroup ids go round-robin over the 8 XCDs, XCD x takes the x-th contiguous eighth of the tile list, and the list runs
    // through column groups of 8 tiles, rows inside a group, columns fastest — the 64 workgroups resident on an XCD are 8 x 8 tiles that
    // share 8 A and 8 B tiles in its L2, an A tile is fetched by one XCD per column group, a B group once per XCD that touches it.
    // L2 hit rate 0.58 -> 0.82, fabric traffic -64 %, face batch 32.4 -> 29.7 ms; the same tiles, the same bits.
    int tx = blockIdx.x, ty = blockIdx.y;
    if (p.xcd_tiles && !split_tile_of((int)blockIdx.x, (p.M + BM - 1) / BM, (p.N + BN - 1) / BN, p.xcd_tiles, tx, ty)) return;   // xcd_tiles = the column-group width
    const int m0 = tx * BM, n0 = ty * BN;

    // ---- per-thread global-load geometry: row (tid/8) of each 32-row pass, float4 column (tid%8) ----
    // Rows are dealt to the 8-thread groups so that the two rows of a 16-lane group are 4 apart: ds_write_b64 is serviced in contiguous
    // 16-lane groups with banks (a / 4) mod 32, and with the 20-dword pitch rows r and r + 1 share 4 banks (2-way conflict in every
    // group: a third of this kernel's LDS cycles, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.333) while rows r and r + 4 are 16 banks apart.
    const int lg = tid >> 3;
    const int lrow = ((lg & 1) << 2) | ((lg >> 1) & 3) | (lg & 24), lc4 = (tid & 7) * 4;
    long a_rowbase[PA];   // (b*Lin) input row base, or -1 if the output row is out of range
    int a_t[PA];          // t*stride
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        int m = m0 + i * 32 + lrow;
        if (m < p.M) {
            int b = m / p.Lout, t = m - b * p.Lout;
            a_rowbase[i] = (long)b * p.Lin;
            a_t[i] = t * p.stride;
        } else {
            a_rowbase[i] = -1;
            a_t[i] = 0;
        }
    }
    const float *wbase = gw + (long)(n0 + lrow) * ldw + lc4;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int li = lane & 31, lh = lane >> 5;

    // ================= pipelined main loop =================
    // The MFMA pipe is only full if the few dozen non-MFMA instructions of a K chunk fit between the MFMAs, so this loop
    // keeps them few and lets the scheduler spread them:
    //   * per-thread operand POINTERS advance by one chunk (128 B) per iteration; they are recomputed only when the walk
    //     enters the next tap / segment (a wave-uniform branch, every len/BK chunks);
    //   * halo rows, rows beyond M and weight rows beyond w_rows point into a zero buffer instead of being predicated
    //     or selected: every load is unconditional and nothing is patched afterwards;
    //   * segment descriptors live in VGPR lanes (v_readlane): no scalar loads competing with LDS for lgkmcnt;
    //   * MFMA fragments are double-buffered: the fragments of q+1 are read while the MFMAs of q run, and the first
    //     fragments of the next chunk are read right after the barrier, under the last MFMA group of this chunk.
    constexpr int NQ = BK / 16;   // k-steps of 16 per chunk
    int vd = 0, vc0 = 0, vlen = BK, vnt = 1;
    if (lane < 4) {
        vd = g.seg[lane].d;
        vc0 = g.seg[lane].c0;
        vlen = g.seg[lane].len;
        vnt = g.seg[lane].ntap > 1 ? g.seg[lane].ntap : 1;
    }
    const float *zero = p.zero + lc4;
    int s = 0, tap = 0, cc = 0;
    int cur_len = __builtin_amdgcn_readlane(vlen, 0), cur_nt = __builtin_amdgcn_readlane(vnt, 0);
    const float *pa[PA], *pb[PB];
    auto enter_run = [&]() {   // operand pointers of the first chunk of (segment s, tap)
        const int sl = s & 3;
        const int d = __builtin_amdgcn_readlane(vd, sl) + tap;
        const int c0 = __builtin_amdgcn_readlane(vc0, sl);
        cur_len = __builtin_amdgcn_readlane(vlen, sl);
        cur_nt = __builtin_amdgcn_readlane(vnt, sl);
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int it = a_t[i] + d;
            const bool ok = a_rowbase[i] >= 0 && it >= 0 && it < p.Lin;
            pa[i] = ok ? gx + (a_rowbase[i] + it) * p.ldx + c0 + lc4 : zero;
        }
    };
    enter_run();
#pragma unroll
    for (int i = 0; i < PB; ++i) pb[i] = n0 + i * 32 + lrow < w_rows ? wbase + (long)i * 32 * ldw : zero;
    // The A and the B half of a chunk move separately (see the loop below): A walks taps / segments, B walks its rows
    auto advance_a = [&]() {
        cc += 1;
        if (cc * BK >= cur_len) {   // wave-uniform: next tap or next segment
            cc = 0;
            tap += 1;
            if (tap >= cur_nt) {
                tap = 0;
                s += 1;
            }
            enter_run();
        } else {
#pragma unroll
            for (int i = 0; i < PA; ++i) pa[i] += BK;
        }
    };
    auto advance_b = [&](int by) {
#pragma unroll
        for (int i = 0; i < PB; ++i) pb[i] += by;
    };
    f32x4 ra[PA], rb[PB];
    auto load_a = [&](f32x4 (&r)[PA]) {
#pragma unroll
        for (int i = 0; i < PA; ++i) r[i] = *reinterpret_cast<const f32x4 *>(pa[i]);
    };
    auto load_b = [&](f32x4 (&r)[PB]) {
#pragma unroll
        for (int i = 0; i < PB; ++i) r[i] = *reinterpret_cast<const f32x4 *>(pb[i]);
    };
    // fp32 registers -> NP bf16 planes in LDS (8 bytes per plane per thread and row): the split happens HERE, so the operands
    // stay fp32 in HBM and no layer needs to know about the arithmetic plan of its neighbours
    auto split4 = [&](const f32x4 &x, uint2 (&out)[NP]) {
        float r0 = x[0], r1 = x[1], r2 = x[2], r3 = x[3];
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
            // v_cvt_pk_bf16_f32 (round to nearest even), through the conversion builtin so that the scheduler sees a VALU instruction
            const uint32_t p01 = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
            const uint32_t p23 = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{r2, r3}, bf16x2));
            out[pl] = uint2{p01, p23};
            if (pl + 1 < NP) {
                r0 -= __builtin_bit_cast(float, p01 << 16);
                r1 -= __builtin_bit_cast(float, p01 & 0xffff0000u);
                r2 -= __builtin_bit_cast(float, p23 << 16);
                r3 -= __builtin_bit_cast(float, p23 & 0xffff0000u);
            }
        }
    };
    auto store_a = [&](int buf, const f32x4 (&r)[PA]) {
        uint2 sp[NP];
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            split4(r[i], sp);
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<uint2 *>(&As[buf][pl][i * 32 + lrow][lc4 >> 1]) = sp[pl];
        }
    };
    auto store_b = [&](int buf, const f32x4 (&r)[PB]) {
        if constexpr (BPRE) {   // the weights arrive as plane images: the 16 bytes a thread fetched are 8 bf16 of ONE plane
            const int pl = (tid & 7) >> 2, dw = (tid & 3) * 4;
#pragma unroll
            for (int i = 0; i < PB; ++i) *reinterpret_cast<f32x4 *>(&Bs[buf][pl][i * 32 + lrow][dw]) = r[i];
            return;
        }
        uint2 sp[NP];
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            split4(r[i], sp);
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<uint2 *>(&Bs[buf][pl][i * 32 + lrow][lc4 >> 1]) = sp[pl];
        }
    };
    // fragments of one k-step of 16: lane (li, lh) holds k = 8 lh .. 8 lh + 7 of row li, for every plane
    bf16x8 fa[2][NP][TM], fb[2][NP][TN];
    auto read_frags = [&](int buf, int q, int slot) {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[slot][pl][i] = __builtin_bit_cast(bf16x8, *(const volatile lds_u32x4 *)__builtin_assume_aligned(&As[buf][pl][wm * WM + i * 32 + li][q * 8 + lh * 4], 16));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[slot][pl][j] = __builtin_bit_cast(bf16x8, *(const volatile lds_u32x4 *)__builtin_assume_aligned(&Bs[buf][pl][wn * WN + j * 32 + li][q * 8 + lh * 4], 16));
        }
    };
    // x y = (x0 + x1 + x2)(y0 + y1 + y2): the cross terms down to 2^-16 (NP = 2: x0 y0 + x0 y1 + x1 y0) or 2^-24 (NP = 3: + x0 y2 +
    // x1 y1 + x2 y0) of the product, every one an exact product of bf16 values accumulated in fp32; small terms first
    auto mfma_q = [&](int slot) {
#pragma unroll
        for (int sum = NP == 2 ? 1 : 2; sum >= 0; --sum)
#pragma unroll
            for (int pa_ = 0; pa_ <= sum; ++pa_) {
                const int pb_ = sum - pa_;
                if (pa_ >= NP || pb_ >= NP) continue;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[slot][pb_][j], fa[slot][pa_][i], acc[i][j], 0, 0, 0);
            }
    };
    // ================= main loop =================
    // A v_mfma_f32_32x32x16_bf16 occupies the pipe for 32 cycles = 8 issue slots, and a chunk carries 24 of them next to ~170 other
    // instructions per wave (111 VALU of the split, 8 + 16 LDS, 8 loads, scalar bookkeeping).  The round-3 form converted and stored a whole
    // chunk ahead of its first MFMAs (~95 VALU in a row with ONE MFMA in flight, then 12 MFMAs back to back behind the barrier); here the
    // chunk is cut in two equal halves around its barrier and each half carries the same side work:
    //   first half   MFMAs of k-step 0 | B of chunk c + 1: registers -> planes -> LDS, B of chunk c + 2 -> registers | fragments of k-step 1
    //   barrier
    //   second half  MFMAs of k-step 1 | A of chunk c + 2: registers -> planes -> LDS (the buffer chunk c just left), A of chunk c + 3 ->
    //                registers | fragments of k-step 0 of chunk c + 1
    // and inside a half sched_group_barrier deals the side work out behind the MFMAs.  Same products, same order: same bits.  Measured
    // (profiles/r04_notes/split_bound.txt): the even interleave alone is worth 1.5 %; the matrix pipe stays 40 % busy with nothing saturated.
    static_assert(NQ == 2, "two k-steps per chunk");
    constexpr int MF = TM * TN * (NP == 2 ? 3 : 6);           // MFMAs per k-step
    constexpr int VALU_HALF = (PA > PB ? PA : PB) * (NP == 2 ? 12 : 22) + 8;
    constexpr int VPM = (VALU_HALF + MF - 1) / MF;
    const int nchunks = p.Ktot / BK;
    int buf = 0;
    {   // prologue: chunk 0 -> buffer 0, A of chunk 1 -> buffer 1, B of chunk 1 and A of chunk 2 -> registers
        f32x4 ra0[PA], rb0[PB];
        load_a(ra0);
        load_b(rb0);
        if (nchunks > 1) {
            advance_a();
            advance_b(BK);
            load_a(ra);
            load_b(rb);
        }
        store_a(0, ra0);
        store_b(0, rb0);
        if (nchunks > 1) store_a(1, ra);
        if (nchunks > 2) {
            advance_a();
            load_a(ra);
        }
    }
    __syncthreads();
    read_frags(0, 0, 0);
    // One loop body for every chunk, the last ones included: past the end the pointers stop advancing (the last chunk is fetched again),
    // the planes written and the fragments read belong to no chunk and are never multiplied — no tail variants, one register allocation.
    for (int it = 0; it < nchunks; ++it) {
        if (it + 3 < nchunks) advance_a();   // up here: its tap / segment branch must not cut a half in two
        advance_b(it + 2 < nchunks ? BK : 0);
        store_b(buf ^ 1, rb);
        load_b(rb);
        read_frags(buf, 1, 1);
        mfma_q(0);
        deal_hints<MF, VPM, NP * (TM + TN) + NP * PB, PB>();
        __syncthreads();
        read_frags(buf ^ 1, 0, 0);
        store_a(buf, ra);
        load_a(ra);
        mfma_q(1);
        deal_hints<MF, VPM, NP * (TM + TN) + NP * PA, PA>();
        buf ^= 1;
    }

    // ---- epilogue: bias (+ residual) + activation, masked store ----
    // The MFMA operands are swapped (weights as A, activations as B: the same products in the same k order, the same bits), so an
    // accumulator block holds D[channel][row]: lane (li, lh) owns output row m = li and, per group g of 4 registers, the 4
    // CONSECUTIVE channels 8 g + 4 lh .. + 3 — bias, residual and output move as 16-byte vectors (4 stores per 32 x 32 block
    // instead of 16; the residual values of a block are fetched together, ahead of their use).  Rows / buffers that are not
    // 16-byte aligned (the 39- / 90- / 129-wide pose rows) and channel tails take the scalar form.
    const bool vec_out = ((p.ldo | g.out_col0) & 3) == 0 && (reinterpret_cast<uintptr_t>(gout) & 15) == 0;
    const bool vec_res = gres && (p.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(gres) & 15) == 0;
    const bool vec_bias = gbias && (reinterpret_cast<uintptr_t>(gbias) & 15) == 0;
    auto activate = [&](float v, float rvv) {
        if (gres && !p.res_after_act) v += rvv;
        if (p.act == 1) v = v >= 0.f ? v : v * 0.2f;
        else if (p.act == 2) v = v > 0.f ? v : 0.f;
        else if (p.act == 3) v = 0.5f * v * (1.0f + split_erff(v * 0.70710678118654752440f));
        if (gres && p.res_after_act) v += rvv;
        return v;
    };
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * WM + i * 32 + li;
        const bool mok = m < p.M;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nb0 = n0 + wn * WN + j * 32 + 4 * lh;
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


// planes: 2 (three products) or 3 (six products).  Same tile ids as launch_conv_gemm (1: 128 x 128, 2: 64 x 64); 0 = by size.
hipError_t launch_conv_gemm_split(const ConvParams &p_in, int planes, hipStream_t stream) {
    ConvParams p = p_in;
    if (!p.zero) {
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) p.zero = skinny_zero_buffer(dev);
    }
    if (!p.zero || p.g[0].nseg > 4 || p.Ktot > 60000 || p.Ktot % 32 != 0 || (planes != 2 && planes != 3)) return hipErrorInvalidValue;
    dim3 block(256);
    const bool xcd = knobs().split_xcd > 0 && p.ngroups == 1 && p.zdiv == 0;
    p.xcd_tiles = xcd ? knobs().split_xcd : 0;
    auto grid = [&](int bm, int bn) {
        const int mt = (p.M + bm - 1) / bm, nt = (p.N + bn - 1) / bn;
        return xcd ? dim3(((mt * nt + 7) / 8) * 8) : dim3(mt, nt, p.ngroups);
    };
    const long tiles128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * p.ngroups;
    const bool big = tiles128 >= 200;
    if (planes == 2 && p.w_planes) {
        if (big) hipLaunchKernelGGL((conv_gemm_split_kernel<128, 128, 64, 64, 2, true>), grid(128, 128), block, 0, stream, p);
        else hipLaunchKernelGGL((conv_gemm_split_kernel<64, 64, 32, 32, 2, true>), grid(64, 64), block, 0, stream, p);
    } else if (planes == 2) {
        if (big) hipLaunchKernelGGL((conv_gemm_split_kernel<128, 128, 64, 64, 2, false>), grid(128, 128), block, 0, stream, p);
        else hipLaunchKernelGGL((conv_gemm_split_kernel<64, 64, 32, 32, 2, false>), grid(64, 64), block, 0, stream, p);
    } else {
        if (p.w_planes) return hipErrorInvalidValue;
        if (big) hipLaunchKernelGGL((conv_gemm_split_kernel<128, 128, 64, 64, 3, false>), grid(128, 128), block, 0, stream, p);
        else hipLaunchKernelGGL((conv_gemm_split_kernel<64, 64, 32, 32, 3, false>), grid(64, 64), block, 0, stream, p);
    }
    return hipGetLastError();
}

// one thread per 4 consecutive k of a row: the arithmetic of split4 (round to nearest even, exact remainder), written as the image a
// BPRE kernel copies into its LDS planes
__global__ __launch_bounds__(256) void split_weight_planes_kernel(const float *__restrict__ w, uint32_t *__restrict__ planes, long n4) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n4) return;
    const f32x4 x = reinterpret_cast<const f32x4 *>(w)[idx];
    const uint32_t h01 = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{x[0], x[1]}, bf16x2));
    const uint32_t h23 = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{x[2], x[3]}, bf16x2));
    const float r0 = x[0] - __builtin_bit_cast(float, h01 << 16), r1 = x[1] - __builtin_bit_cast(float, h01 & 0xffff0000u);
    const float r2 = x[2] - __builtin_bit_cast(float, h23 << 16), r3 = x[3] - __builtin_bit_cast(float, h23 & 0xffff0000u);
    const uint32_t l01 = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
    const uint32_t l23 = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{r2, r3}, bf16x2));
    const long chunk = idx >> 3;      // 8 threads per chunk of 32 k (K % 32 == 0: chunks never straddle rows)
    const int q = (int)(idx & 7);
    uint32_t *o = planes + chunk * 32 + 2 * q;
    *reinterpret_cast<uint2 *>(o) = uint2{h01, h23};
    *reinterpret_cast<uint2 *>(o + 16) = uint2{l01, l23};
}

hipError_t launch_split_weight_planes(const float *w, float *planes, long rows, int K, hipStream_t stream) {
    if (!w || !planes || rows < 1 || K < 32 || K % 32) return hipErrorInvalidValue;
    const long n4 = rows * (long)K / 4;
    hipLaunchKernelGGL(split_weight_planes_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, w, reinterpret_cast<uint32_t *>(planes), n4);
    return hipGetLastError();
}

}  // namespace ts
