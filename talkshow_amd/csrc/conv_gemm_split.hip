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
// Operands stay fp32 in HBM: the split happens between the global-load registers and the LDS planes (v_cvt_pk_bf16_f32 +
// one subtraction per extra plane), so layers need not agree on a plan.  Same segments / taps / epilogue as conv_gemm_f32.
#include "kernels.h"
#include <cstdint>

namespace ts {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;   // LDS-qualified: volatile accesses must not fall back to flat

template <int BM, int BN, int WM, int WN, int NP>
__global__ __launch_bounds__(256) void conv_gemm_split_kernel(const ConvParams p) {
    constexpr int BK = 32;
    constexpr int LDS_LD = 20;       // dwords per LDS row: 32 bf16 (16 dwords) + 4 of padding -> 80 B pitch, conflict-free for ds_read_b128
    constexpr int KC = 1;
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
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

    // ---- per-thread global-load geometry: row (tid/8) of each 32-row pass, float4 column (tid%8) ----
    const int lrow = tid >> 3, lc4 = (tid & 7) * 4;
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
    auto advance = [&]() {
        cc += 1;
#pragma unroll
        for (int i = 0; i < PB; ++i) pb[i] += BK;
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
    f32x4 ra[PA][KC], rb[PB][KC];
    auto load_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < PA; ++i)
#pragma unroll
            for (int c = 0; c < KC; ++c) ra[i][c] = *reinterpret_cast<const f32x4 *>(pa[i] + c * 32);
#pragma unroll
        for (int i = 0; i < PB; ++i)
#pragma unroll
            for (int c = 0; c < KC; ++c) rb[i][c] = *reinterpret_cast<const f32x4 *>(pb[i] + c * 32);
    };
    // fp32 registers -> NP bf16 planes in LDS (8 bytes per plane per thread and row): the split happens HERE, so the operands
    // stay fp32 in HBM and no layer needs to know about the arithmetic plan of its neighbours
    auto split4 = [&](const f32x4 &x, uint2 (&out)[NP]) {
        float r0 = x[0], r1 = x[1], r2 = x[2], r3 = x[3];
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
            uint32_t p01, p23;
            asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p01) : "v"(r0), "v"(r1));
            asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p23) : "v"(r2), "v"(r3));
            out[pl] = uint2{p01, p23};
            if (pl + 1 < NP) {
                r0 -= __builtin_bit_cast(float, p01 << 16);
                r1 -= __builtin_bit_cast(float, p01 & 0xffff0000u);
                r2 -= __builtin_bit_cast(float, p23 << 16);
                r3 -= __builtin_bit_cast(float, p23 & 0xffff0000u);
            }
        }
    };
    auto store_chunk = [&](int buf) {
        uint2 s[NP];
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            split4(ra[i][0], s);
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<uint2 *>(&As[buf][pl][i * 32 + lrow][lc4 >> 1]) = s[pl];
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            split4(rb[i][0], s);
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<uint2 *>(&Bs[buf][pl][i * 32 + lrow][lc4 >> 1]) = s[pl];
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
    const int nchunks = p.Ktot / BK;
    load_chunk();
    store_chunk(0);
    if (nchunks > 1) {
        advance();
        load_chunk();
    }
    __syncthreads();
    read_frags(0, 0, 0);
    int buf = 0;
    int it = 0;
    for (; it + 2 < nchunks; ++it) {   // steady state: chunk it+1 -> LDS, chunk it+2 -> registers, MFMAs of chunk it
        advance();
        store_chunk(buf ^ 1);
        load_chunk();
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (q + 1 < NQ) {
                read_frags(buf, q + 1, (q + 1) & 1);
            } else {
                __syncthreads();
                read_frags(buf ^ 1, 0, 0);
            }
            mfma_q(q & 1);
            // the LDS writes and the global loads of this iteration must be issued within the first MFMA group: left to
            // itself the scheduler sinks the loads to the end of the iteration and the next one stalls on them
            if (q == 0) __builtin_amdgcn_sched_barrier(0);
        }
        buf ^= 1;
    }
    for (; it < nchunks; ++it) {       // last two chunks: nothing left to load
        const bool has_next = it + 1 < nchunks;
        if (has_next) store_chunk(buf ^ 1);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (q + 1 < NQ) {
                read_frags(buf, q + 1, (q + 1) & 1);
            } else {
                __syncthreads();
                if (has_next) read_frags(buf ^ 1, 0, 0);
            }
            mfma_q(q & 1);
        }
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
        else if (p.act == 3) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
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
    auto grid = [&](int bm, int bn) { return dim3((p.M + bm - 1) / bm, (p.N + bn - 1) / bn, p.ngroups); };
    const long tiles128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * p.ngroups;
    const bool big = tiles128 >= 200;
    if (planes == 2) {
        if (big) hipLaunchKernelGGL((conv_gemm_split_kernel<128, 128, 64, 64, 2>), grid(128, 128), block, 0, stream, p);
        else hipLaunchKernelGGL((conv_gemm_split_kernel<64, 64, 32, 32, 2>), grid(64, 64), block, 0, stream, p);
    } else {
        if (big) hipLaunchKernelGGL((conv_gemm_split_kernel<128, 128, 64, 64, 3>), grid(128, 128), block, 0, stream, p);
        else hipLaunchKernelGGL((conv_gemm_split_kernel<64, 64, 32, 32, 3>), grid(64, 64), block, 0, stream, p);
    }
    return hipGetLastError();
}

}  // namespace ts
