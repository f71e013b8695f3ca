// skinny_gemm_f32 — the per-position GEMM of the autoregressive PixelCNN chain.
//
// One launch = one dependent stage of GatedMaskedConv2d / the logits head evaluated at ONE code position for the
// whole batch of clips (reference: nets/spg/gated_pixelcnn_v2.py:61-87,120-124,137-144): M = B (or 2B) rows,
// N = 256..2048 output channels, K = 256..1536.  The chain is latency bound (≈5k dependent stages per batch), so
// the kernel is built to be SHORT rather than to stream: a workgroup owns 32 output columns for all rows, its
// W (4/8/16) waves split K, every wave issues all of its 16-byte operand loads up front (operands go straight
// from L2 to VGPRs — a weight row is read by exactly one lane, LDS staging would only add a round trip), runs its
// share of v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chains), and the partial 32x32 tiles are summed through LDS in
// a fixed order (deterministic).  Epilogues fuse bias, an additive term (v->h contribution / residual / audio
// term), the class conditioning and the tanh*sigmoid gate: the tile's columns are 16 "tanh" channels followed by
// their 16 "sigmoid" partners, so the gate is one cross-lane exchange (lane ^ 16).
#include <cstdlib>

#include "kernels.h"

namespace ts {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int W>
__global__ __launch_bounds__(W * 64) void skinny_gemm_kernel(const SkinnyBatch batch) {
    __shared__ float red[W][16][64];

    // several INDEPENDENT problems share one launch (e.g. a horizontal-chain stage of this row, the vertical conv of
    // layer l and vert_to_horiz of layer l-1): one kernel boundary instead of three on the dependent chain.
    const SkinnyParams &p = batch.p[blockIdx.z];
    if ((int)blockIdx.x >= p.grid_x || (int)blockIdx.y >= p.grid_y) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int tile = blockIdx.x, mt = blockIdx.y;

    // output column of this lane
    int n;
    if (p.epi == EPI_GATE) {
        const int tiles_per_group = p.gateD >> 4;
        const int group = tile / tiles_per_group, ch0 = (tile - group * tiles_per_group) << 4;
        n = group * 2 * p.gateD + (li >> 4) * p.gateD + ch0 + (li & 15);
    } else {
        n = tile * 32 + li;
    }
    const bool n_ok = n < p.N;
    const float *wrow = p.W + (long)(n_ok ? n : 0) * p.ldw + lh * 4;

    // A row of this lane
    const int m = mt * 32 + li;
    const bool m_ok = m < p.M;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    // epilogue operands (bias / additive terms / class conditioning) do not depend on the GEMM: fetch them now so
    // their L2 latency overlaps the operand loads and MFMAs instead of trailing the reduction
    constexpr int RPW = 16 / W;   // accumulator registers finished by each wave
    float e_add[RPW], e_cls[RPW];
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int r = wave * RPW + rr;
        const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int rowc = row < p.M ? row : 0;
        const int nc = n_ok ? n : 0;
        float a = 0.f;
        if (p.bias) a += p.bias[nc];
        if (p.add1) a += p.add1[(long)(rowc >> p.add1_shift) * p.add1_stride + nc];
        if (p.add2) a += p.add2[(long)(rowc >> p.add2_shift) * p.add2_stride + nc];
        if (p.add3) a += p.add3[(long)rowc * p.add3_stride + nc];
        e_add[rr] = a;
        e_cls[rr] = (p.epi == EPI_GATE && p.clsrow) ? p.clsrow[(long)rowc * p.cls_ld + (nc % p.cls_ld)] : 0.f;
    }

    const int Q = p.Ktot >> 3;
    const int qbeg = (Q * wave) / W, qend = (Q * (wave + 1)) / W;
    int qs = 0;
    for (int s = 0; s < p.nseg; ++s) {
        const SkinnySeg &sg = p.seg[s];
        const int qlen = sg.len >> 3;
        const int lo = qbeg > qs ? qbeg : qs;
        const int hi = qend < qs + qlen ? qend : qs + qlen;
        if (lo < hi) {
            const float *arow = nullptr;
            if (m_ok) {
                if (sg.gidx) {
                    const int gi = sg.gidx[(long)m * sg.gidx_stride];
                    if (gi >= 0) arow = sg.base + (long)gi * sg.row_stride;
                } else if (sg.base) {   // a null dense segment is a block of zero rows (row above the grid)
                    arow = sg.base + (long)(m >> sg.row_shift) * sg.row_stride;
                }
            }
            const float *ap = arow ? arow + (lo - qs) * 8 + lh * 4 : nullptr;
            const float *bp = wrow + lo * 8;
            int q = lo;
            for (; q + 4 <= hi; q += 4) {
                f32x4 a[4], b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    b[u] = *reinterpret_cast<const f32x4 *>(bp + u * 8);
                    a[u] = ap ? *reinterpret_cast<const f32x4 *>(ap + u * 8) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][e], b[u][e], acc, 0, 0, 0);
                bp += 32;
                if (ap) ap += 32;
            }
            for (; q < hi; ++q) {
                f32x4 b = *reinterpret_cast<const f32x4 *>(bp);
                f32x4 a = ap ? *reinterpret_cast<const f32x4 *>(ap) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
                bp += 8;
                if (ap) ap += 8;
            }
        }
        qs += qlen;
    }

    // ---- cross-wave reduction through LDS, fixed summation order ----
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();

#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int r = wave * RPW + rr;
        float v = red[0][r][lane];
#pragma unroll
        for (int w = 1; w < W; ++w) v += red[w][r][lane];

        const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const bool ok = n_ok && row < p.M;
        v += e_add[rr];
        if (p.epi == EPI_GATE) {
            if (p.pre && ok) p.pre[(long)row * p.pre_stride + n] = v;
            v += e_cls[rr];
            const float partner = __shfl_xor(v, 16);
            if ((li & 16) == 0 && ok) {
                const float gate = tanhf(v) * (1.0f / (1.0f + expf(-partner)));
                const int tiles_per_group = p.gateD >> 4;
                const int group = tile / tiles_per_group, ch0 = (tile - group * tiles_per_group) << 4;
                p.out[(long)row * p.out_stride + group * p.gateD + ch0 + (li & 15)] = gate;
            }
        } else {
            if (p.relu) v = v > 0.f ? v : 0.f;
            if (ok) p.out[(long)row * p.out_stride + n] = v;
        }
    }
}



// ---------------------------------------------------------------------------------------------------------------
// 16-column variant on v_mfma_f32_16x16x4_f32: a workgroup owns 16 output columns (8 "tanh" + 8 "sigmoid" partners
// for the gate epilogue) for 32 rows (two 16-row MFMA blocks sharing the B operand).  Twice as many workgroups per
// stage as the 32-column kernel, each with half the weight bytes and half the MFMA burst: the stages are bounded by
// what ONE compute unit can fetch and multiply, so spreading a stage over more CUs shortens it.
// Lane (i = lane & 15, g = lane >> 4) supplies A[row i][k] and B[k][col i] for k = 16 q + 4 g + e, e = 0..3 (one
// 16-byte load per operand per q-step; a K permutation applied identically to A and B).
// ---------------------------------------------------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(W * 64) void skinny16_kernel(const SkinnyBatch batch) {
    __shared__ float red[W][8][64];
    const SkinnyParams &p = batch.p[blockIdx.z];
    if ((int)blockIdx.x >= p.grid_x || (int)blockIdx.y >= p.grid_y) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int tile = blockIdx.x, mt = blockIdx.y;

    int n;
    if (p.epi == EPI_GATE) {
        const int tiles_per_group = p.gateD >> 3;
        const int group = tile / tiles_per_group, ch0 = (tile - group * tiles_per_group) << 3;
        n = group * 2 * p.gateD + (li >> 3) * p.gateD + ch0 + (li & 7);
    } else {
        n = tile * 16 + li;
    }
    const bool n_ok = n < p.N;
    const float *wrow = p.W + (long)(n_ok ? n : 0) * p.ldw + lg * 4;

    const int m0 = mt * 32 + li, m1 = m0 + 16;

    constexpr int RPW = 8 / W;   // accumulator registers (of 8) finished by each wave
    float e_add[RPW], e_cls[RPW];
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int r = wave * RPW + rr;
        const int row = mt * 32 + (r >> 2) * 16 + lg * 4 + (r & 3);
        const int rowc = row < p.M ? row : 0;
        const int nc = n_ok ? n : 0;
        float a = 0.f;
        if (p.bias) a += p.bias[nc];
        if (p.add1) a += p.add1[(long)(rowc >> p.add1_shift) * p.add1_stride + nc];
        if (p.add2) a += p.add2[(long)(rowc >> p.add2_shift) * p.add2_stride + nc];
        if (p.add3) a += p.add3[(long)rowc * p.add3_stride + nc];
        e_add[rr] = a;
        e_cls[rr] = (p.epi == EPI_GATE && p.clsrow) ? p.clsrow[(long)rowc * p.cls_ld + (nc % p.cls_ld)] : 0.f;
    }

    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const int Q = p.Ktot >> 4;
    const int qbeg = (Q * wave) / W, qend = (Q * (wave + 1)) / W;
    int qs = 0;
    for (int s = 0; s < p.nseg; ++s) {
        const SkinnySeg &sg = p.seg[s];
        const int qlen = sg.len >> 4;
        const int lo = qbeg > qs ? qbeg : qs;
        const int hi = qend < qs + qlen ? qend : qs + qlen;
        if (lo < hi) {
            const float *ar0 = nullptr, *ar1 = nullptr;
            if (sg.gidx) {
                if (m0 < p.M) { const int gi = sg.gidx[(long)m0 * sg.gidx_stride]; if (gi >= 0) ar0 = sg.base + (long)gi * sg.row_stride; }
                if (m1 < p.M) { const int gi = sg.gidx[(long)m1 * sg.gidx_stride]; if (gi >= 0) ar1 = sg.base + (long)gi * sg.row_stride; }
            } else if (sg.base) {
                if (m0 < p.M) ar0 = sg.base + (long)(m0 >> sg.row_shift) * sg.row_stride;
                if (m1 < p.M) ar1 = sg.base + (long)(m1 >> sg.row_shift) * sg.row_stride;
            }
            const int koff = (lo - qs) * 16 + lg * 4;
            const float *ap0 = ar0 ? ar0 + koff : nullptr, *ap1 = ar1 ? ar1 + koff : nullptr;
            const float *bp = wrow + lo * 16;
            int q = lo;
            for (; q + 4 <= hi; q += 4) {
                f32x4 a0[4], a1[4], b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    b[u] = *reinterpret_cast<const f32x4 *>(bp + u * 16);
                    a0[u] = ap0 ? *reinterpret_cast<const f32x4 *>(ap0 + u * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
                    a1[u] = ap1 ? *reinterpret_cast<const f32x4 *>(ap1 + u * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u][e], b[u][e], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u][e], b[u][e], acc1, 0, 0, 0);
                    }
                bp += 64;
                if (ap0) ap0 += 64;
                if (ap1) ap1 += 64;
            }
            for (; q < hi; ++q) {
                const f32x4 b = *reinterpret_cast<const f32x4 *>(bp);
                const f32x4 a0 = ap0 ? *reinterpret_cast<const f32x4 *>(ap0) : f32x4{0.f, 0.f, 0.f, 0.f};
                const f32x4 a1 = ap1 ? *reinterpret_cast<const f32x4 *>(ap1) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], b[e], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], b[e], acc1, 0, 0, 0);
                }
                bp += 16;
                if (ap0) ap0 += 16;
                if (ap1) ap1 += 16;
            }
        }
        qs += qlen;
    }

#pragma unroll
    for (int r = 0; r < 4; ++r) {
        red[wave][r][lane] = acc0[r];
        red[wave][4 + r][lane] = acc1[r];
    }
    __syncthreads();

#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int r = wave * RPW + rr;
        float v = red[0][r][lane];
#pragma unroll
        for (int w = 1; w < W; ++w) v += red[w][r][lane];
        const int row = mt * 32 + (r >> 2) * 16 + lg * 4 + (r & 3);
        const bool ok = n_ok && row < p.M;
        v += e_add[rr];
        if (p.epi == EPI_GATE) {
            if (p.pre && ok) p.pre[(long)row * p.pre_stride + n] = v;
            v += e_cls[rr];
            const float partner = __shfl_xor(v, 8);
            if ((li & 8) == 0 && ok) {
                const float gate = tanhf(v) * (1.0f / (1.0f + expf(-partner)));
                const int tiles_per_group = p.gateD >> 3;
                const int group = tile / tiles_per_group, ch0 = (tile - group * tiles_per_group) << 3;
                p.out[(long)row * p.out_stride + group * p.gateD + ch0 + (li & 7)] = gate;
            }
        } else {
            if (p.relu) v = v > 0.f ? v : 0.f;
            if (ok) p.out[(long)row * p.out_stride + n] = v;
        }
    }
}

static int skinny_grid(SkinnyParams &p, int ncol) {
    if (p.Ktot % (ncol == 16 ? 16 : 8) != 0 || p.M <= 0) return -1;
    for (int s = 0; s < p.nseg; ++s)
        if (p.seg[s].len % (ncol == 16 ? 16 : 8) != 0) return -1;
    if (p.epi == EPI_GATE) {
        if (p.gateD % (ncol / 2) != 0 || p.N % (2 * p.gateD) != 0) return -1;
        p.grid_x = p.N / ncol;
    } else {
        p.grid_x = (p.N + ncol - 1) / ncol;
    }
    p.grid_y = (p.M + 31) / 32;
    return 0;
}

hipError_t launch_skinny_batch(const SkinnyParams *const *ps, int n, hipStream_t stream) {
    if (n < 1 || n > SKINNY_MAX_PROBLEMS) return hipErrorInvalidValue;
    SkinnyBatch b;
    int gx = 0, gy = 0, Q = 0;
    // column-tile width: 16 (more, leaner workgroups per stage) unless a shape needs the 8-granular 32-column kernel
    static const int ncol_pref = [] { const char *e = getenv("TS_SKINNY_NT"); return e ? atoi(e) : 16; }();
    int ncol = ncol_pref == 32 ? 32 : 16;
    if (ncol == 16)
        for (int i = 0; i < n; ++i) {
            SkinnyParams t = *ps[i];
            if (skinny_grid(t, 16)) ncol = 32;
        }
    for (int i = 0; i < n; ++i) {
        b.p[i] = *ps[i];
        if (b.p[i].nseg > SKINNY_MAX_SEG || skinny_grid(b.p[i], ncol)) return hipErrorInvalidValue;
        gx = gx > b.p[i].grid_x ? gx : b.p[i].grid_x;
        gy = gy > b.p[i].grid_y ? gy : b.p[i].grid_y;
        Q = Q > b.p[i].Ktot / 8 ? Q : b.p[i].Ktot / 8;
    }
    dim3 grid(gx, gy, n);
    // K is split over the waves of the workgroup; more waves = more loads in flight (lower latency for ONE chain) but a
    // fatter workgroup.  TS_SKINNY_MAXW caps it (tuning).
    static const int maxw = [] { const char *e = getenv("TS_SKINNY_MAXW"); return e ? atoi(e) : 16; }();
    if (ncol == 16) {   // Q counts 8-k steps: K = 8 Q; the 16-column kernel keeps 8 accumulators -> at most 8 waves
        const int W16 = Q >= 32 ? 8 : 4;
        if (W16 >= 8 && maxw >= 8) hipLaunchKernelGGL(skinny16_kernel<8>, grid, dim3(512), 0, stream, b);
        else hipLaunchKernelGGL(skinny16_kernel<4>, grid, dim3(256), 0, stream, b);
        return hipGetLastError();
    }
    int W = Q >= 64 ? 16 : (Q >= 32 ? 8 : 4);
    if (W > maxw) W = maxw;
    if (W >= 16) hipLaunchKernelGGL(skinny_gemm_kernel<16>, grid, dim3(1024), 0, stream, b);
    else if (W >= 8) hipLaunchKernelGGL(skinny_gemm_kernel<8>, grid, dim3(512), 0, stream, b);
    else hipLaunchKernelGGL(skinny_gemm_kernel<4>, grid, dim3(256), 0, stream, b);
    return hipGetLastError();
}

hipError_t launch_skinny_gemm(const SkinnyParams &p, hipStream_t stream) {
    const SkinnyParams *ps[1] = {&p};
    return launch_skinny_batch(ps, 1, stream);
}

}  // namespace ts
