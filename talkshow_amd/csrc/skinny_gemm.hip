// skinny_gemm_f32 — the per-position GEMM of the autoregressive PixelCNN chain.
//
// One launch = one dependent stage of GatedMaskedConv2d / the logits head evaluated at ONE code position for the
// whole batch of clips (reference: nets/spg/gated_pixelcnn_v2.py:61-87,120-124,137-144): M = B (or 2B) rows,
// N = 256..2048 output channels, K = 256..512, up to 6 independent problems per launch.  The chain is latency bound
// (≈2.8k dependent launches per batch), so the kernels are built to be SHORT rather than to stream: a workgroup owns
// a 16- or 32-column tile, its waves split K, operands go straight from L2 to VGPRs (a weight row is read by exactly
// one lane; LDS staging would only add a round trip), v_mfma_f32_{16x16x4,32x32x2}_f32 are exact fp32 fmaf chains,
// and the partial tiles are summed through LDS in a fixed order (deterministic).  Epilogues fuse bias, additive
// terms (v->h contribution / residual / audio term), the class conditioning and the tanh*sigmoid gate: a tile's
// columns are "tanh" channels followed by their "sigmoid" partners, so the gate is one cross-lane exchange.
//
// Three kernels, one contract (bit-identical results):
//   skinny16_fast_kernel  production path: 64-dword problem descriptors, branch-free loads, exact 1-D grid, 16- or
//                         32-row tiles (see the comment above it; tools/skinny_trace.py is its in-kernel profiler)
//   skinny16_kernel       generic 16-column kernel: any K multiple of 16, any segment layout (small / odd models)
//   skinny_gemm_kernel    generic 32-column kernel on the 32x32x2 MFMA: K multiple of 8
#include <cstdlib>
#include <cstring>
#include <cstdint>

#include "kernels.h"
#include "skinny_desc.h"

namespace ts {

typedef float f32x16 __attribute__((ext_vector_type(16)));

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
                const float gate = gate_act(v, partner);
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
                const float gate = gate_act(v, partner);
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


// ---------------------------------------------------------------------------------------------------------------
// Fast path of the 16-column kernel.  A chain stage lasts a few microseconds, so what matters is the number of DEPENDENT
// memory round trips between wave start and the first MFMA.  The generic kernels above read their problem description
// field by field from the kernarg segment (a dozen dependent scalar loads, cold after every kernel boundary) and fetch
// the optional epilogue operands one after the other.  Here
//   * the problem is a 64-dword descriptor: every wave fetches it with ONE vector load (lane i holds word i) and pulls
//     fields out with v_readlane — one round trip, no LDS, no scalar-cache misses;
//   * optional operands never branch: the host points absent ones at a zero buffer (stride 0), rows beyond M are
//     clamped (their results are never stored), so all operand loads of the wave — weights, activations, bias and
//     additive terms — are issued back to back and overlap: a second (and last) round trip;
//   * a wave's share of K is CNT = K / (16 W) <= 4 q-steps inside one segment (host-checked), straight-line code.
// Partition of K over waves and the LDS summation order are those of skinny16_kernel: results are bit-identical.
// ---------------------------------------------------------------------------------------------------------------
template <int V> struct IC { static constexpr int value = V; };

// One q-step (16 k) of a wave: RB row blocks x CB column blocks of 16x16 outputs share RB + CB 16-byte operand loads.
template <int RB, int CB>
__device__ __forceinline__ void skinny16_load_step(gcf *const (&ap)[4], gcf *const (&bp)[4], int u, int astep, int bstep,
                                                   f32x4 (&a)[4], f32x4 (&b)[4]) {
#pragma unroll
    for (int c = 0; c < CB; ++c) b[c] = *reinterpret_cast<gcf4 *>(bp[c] + u * bstep);
#pragma unroll
    for (int r = 0; r < RB; ++r) a[r] = *reinterpret_cast<gcf4 *>(ap[r] + u * astep);
}
template <int RB, int CB>
__device__ __forceinline__ void skinny16_mfma_step(const f32x4 (&a)[4], const f32x4 (&b)[4], f32x4 (&acc)[16]) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int c = 0; c < CB; ++c)
                acc[r * CB + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][e], b[c][e], acc[r * CB + c], 0, 0, 0);
}

// TRACE instrumentation (tools/skinny_trace.py): workgroup (0,0,0), wave 0 stamps the 100 MHz wall clock at five points
constexpr unsigned TRACE_REC = 24;           // u64 per record: 5 stamps, meta, then per wave (loads back, MFMAs done)
constexpr unsigned TRACE_WGS = 512;          // slots per launch
constexpr unsigned TRACE_LAUNCHES = 4096;    // launches traced (launch sequence number modulo this)
constexpr size_t TRACE_SLOTS = (size_t)TRACE_WGS * TRACE_LAUNCHES;
__device__ unsigned long long *g_trace;   // [TRACE_SLOTS][TRACE_REC], allocated by skinny_init when TS_SKINNY_TRACE is set
static unsigned g_trace_seq = 0;
static unsigned long long *g_trace_host = nullptr;   // host copy of the device pointer in g_trace          // host: sequence number handed to the next traced launch

// Tile = RB x CB blocks of 16 x 16 outputs (one v_mfma_f32_16x16x4_f32 accumulator each):
//   (1,1) 16 rows x 16 columns   a single chain whose launch still fits one workgroup per CU: least bytes per CU
//   (2,1) 32 x 16                one batch of <= 32 clips
//   (2,2) 32 x 32, (4,2) 64 x 32, (4,4) 64 x 64   coalesced batches (M >= 64 clips per stage): a launch is then bound by the
//                                operand bytes a CU pulls through its L1; bytes per output 256 / 192 ->
//                                128 / 96 / 64 B.  (4,4) runs one workgroup per CU (128 KB of LDS for the reduction)
// Every block of every tile shape is accumulated in the same order (same K split over the waves, same MFMA, same LDS
// summation order), so a clip's result does not depend on the tile shape its batch happened to get: bit-identical.
template <int W, int RB, int CB, bool TRACE = false>
__global__ __launch_bounds__(W * 64, (RB * CB > 8 ? W / 4 : (RB * CB > 4 ? W / 2 : 1))) void skinny16_fast_kernel(const SkinnyDescBatch batch) {
    constexpr int ROWS = RB * 16;
    constexpr int NBLK = RB * CB;
    constexpr int NREG = NBLK * 4;              // accumulator registers per lane to reduce across the waves
    __shared__ float red[W][NREG][64];
    __shared__ unsigned long long wave_t[TRACE ? W : 1][2];
    unsigned long long tr[5] = {0, 0, 0, 0, 0};
    if (TRACE) tr[0] = wall_clock64();
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;

    // the grid is 1-D over the live tiles of all problems: fetch every problem's descriptor word (independent loads, one
    // round trip) while the scalar unit finds the problem this workgroup belongs to
    int dws[SKINNY_MAX_PROBLEMS];
#pragma unroll
    for (int i = 0; i < SKINNY_MAX_PROBLEMS; ++i) dws[i] = (int)batch.d[i].w[lane];
    // keep these six loads the FIRST thing the wave issues: left alone, hipcc sinks them behind the scalar loads of start[] and the
    // problem search below (it even folds two of them into one load of a selected address), which puts a second round trip in
    // front of the descriptor — 0.3 us per launch, 4 % of a 32-clip chain (the "code layout" swings of round 2 were this)
    __builtin_amdgcn_sched_barrier(0);
    const int bx = blockIdx.x;
    int z = 0, first = 0;
#pragma unroll
    for (int i = 1; i < SKINNY_MAX_PROBLEMS; ++i) {
        const int st = batch.start[i];
        if (bx >= st) { z = i; first = st; }
    }
    int dw = dws[0];
#pragma unroll
    for (int i = 1; i < SKINNY_MAX_PROBLEMS; ++i) dw = z == i ? dws[i] : dw;
    auto I = [&](int k) { return __builtin_amdgcn_readlane(dw, k); };
    auto P = [&](int k) {
        return (gcf *)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(dw, k + 1) << 32) |
                       (uint64_t)(uint32_t)__builtin_amdgcn_readlane(dw, k));
    };
    const int M = I(SD_M), N = I(SD_N), flags = I(SD_FLAGS), gateD = I(SD_GATED);
    const int nmt = (M + ROWS - 1) / ROWS;      // row tiles; tiles of a problem are enumerated column-major
    // scalar integer division costs ~30 instructions on the way to the first operand load: the usual divisors (row tiles per
    // problem, gate tiles per group, class-row width) are powers of two.  Same-box A/B: 37.1 -> 36.4 ms per 256-clip pass.
    auto sdiv = [](int x, int d) { return (d & (d - 1)) == 0 ? x >> __builtin_ctz(d) : x / d; };
    const int tile = sdiv(bx - first, nmt), mt = (bx - first) - tile * nmt;
    if (TRACE) tr[1] = wall_clock64();

    const bool gate = flags & SDF_GATE;
    int n[CB], nc[CB];
    bool n_ok[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) {
        const int t16 = tile * CB + c;
        if (gate) {
            const int tiles_per_group = gateD >> 3;
            const int group = sdiv(t16, tiles_per_group), ch0 = (t16 - group * tiles_per_group) << 3;
            n[c] = group * 2 * gateD + (li >> 3) * gateD + ch0 + (li & 7);
        } else {
            n[c] = t16 * 16 + li;
        }
        n_ok[c] = n[c] < N;
        nc[c] = n_ok[c] ? n[c] : 0;
    }

    // ---- this wave's K range: CNT q-steps (16 k each) inside one segment ----
    const int cnt = I(SD_CNT);
    const int q0 = wave * cnt;
    int sbase = SD_SEG, qs = 0;
    {
        const int nseg = I(SD_NSEG);
        const int l0 = I(SD_SEG + 7);
        if (nseg > 1 && q0 >= l0) {
            sbase = SD_SEG + SD_SEG_WORDS;
            qs = l0;
            const int l1 = I(SD_SEG + SD_SEG_WORDS + 7);
            if (nseg > 2 && q0 >= l0 + l1) {
                sbase = SD_SEG + 2 * SD_SEG_WORDS;
                qs = l0 + l1;
            }
        }
    }
    gcf *base = P(sbase);
    gci *gidx = (gci *)P(sbase + 2);
    const int row_stride = I(sbase + 4), segw = I(sbase + 6);
    const bool a_tiled = segw & SEG_TILED;          // wave-uniform
    const int astep = a_tiled ? 256 : 16;           // floats between consecutive q-steps of this wave's A stream
    const int koff = (q0 - qs) * 16 + lg * 4;
    gcf *ap[4], *bp[4];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int m = mt * ROWS + r * 16 + li;
        const int mc = m < M ? m : M - 1;   // clamped rows: computed, never stored
        if (gidx) {   // wave-uniform: token-embedding gather (one extra round trip, first stage of column 1 only)
            const int g = gidx[(long)mc * I(sbase + 5)];
            ap[r] = (g >= 0 ? base + (long)g * row_stride : P(SD_ZERO)) + koff;
        } else if (a_tiled) {   // fragment (row block, q) is one contiguous KB in lane order
            const int nblk = (M + 15) >> 4;
            int blk = mt * RB + r;
            blk = blk < nblk ? blk : nblk - 1;
            ap[r] = base + ((((long)blk * (segw & 0xffff) + (q0 - qs)) << 6) + lane) * 4;
        } else {
            ap[r] = base + (long)mc * row_stride + koff;
        }
    }
    const int wtq = I(SD_WTQ);
    const int bstep = wtq ? 256 : 16;
#pragma unroll
    for (int c = 0; c < CB; ++c)
        bp[c] = wtq ? P(SD_W) + ((((long)(tile * CB + c) * wtq + q0) << 6) + lane) * 4
                    : P(SD_W) + (long)nc[c] * I(SD_LDW) + q0 * 16 + lg * 4;

    f32x4 acc[16];
#pragma unroll
    for (int k = 0; k < NBLK; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int RPW = NREG >= W ? NREG / W : 1;   // registers finished by each (active) wave
    const bool active = wave * RPW < NREG;          // (1,1) with 8 waves: waves 4..7 only contribute partial sums
    float e_add[RPW], e_cls[RPW];
    const int add1_tw = I(SD_ADD1_TW), out_tw = I(SD_OUT_TW), pre_tw = I(SD_PRE_TW);
    auto epilogue_operands = [&]() {
        if (!active) return;
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int r = wave * RPW + rr;
            const int blk = r >> 2, rb = blk / CB, cb = blk - rb * CB;
            const int row = mt * ROWS + rb * 16 + lg * 4 + (r & 3);
            const int rowc = row < M ? row : 0;
            const int ncol = nc[cb];
            const float t0 = P(SD_BIAS)[ncol];
            const long i1 = (long)(rowc >> I(SD_ADD1_SHIFT)) * I(SD_ADD1_STRIDE) + ncol;
            const float t1 = P(SD_ADD1)[add1_tw ? tiled_index(i1, add1_tw) : i1];
            const float t2 = P(SD_ADD2)[(long)(rowc >> I(SD_ADD2_SHIFT)) * I(SD_ADD2_STRIDE) + ncol];
            const float t3 = P(SD_ADD3)[(long)rowc * I(SD_ADD3_STRIDE) + ncol];
            const int cls_ld = I(SD_CLS_LD);
            const int ccol = (cls_ld & (cls_ld - 1)) == 0 ? (ncol & (cls_ld - 1)) : ncol % cls_ld;
            e_cls[rr] = P(SD_CLS)[(long)rowc * cls_ld + ccol];
            e_add[rr] = ((t0 + t1) + t2) + t3;
        }
    };
    unsigned long long t_loads = 0;
    if (NBLK <= 2) {
        // phase 1: every load of this wave — K operands first (they are waited for first), then the epilogue operands;
        // phase 2: the MFMAs.  Straight-line code per q-step count.
        f32x4 a[8][4], b[8][4];
        auto phase1 = [&](auto cnt_c) {
            constexpr int CNT = decltype(cnt_c)::value;
#pragma unroll
            for (int u = 0; u < CNT; ++u) skinny16_load_step<RB, CB>(ap, bp, u, astep, bstep, a[u], b[u]);
        };
        auto phase2 = [&](auto cnt_c) {
            constexpr int CNT = decltype(cnt_c)::value;
#pragma unroll
            for (int u = 0; u < CNT; ++u) skinny16_mfma_step<RB, CB>(a[u], b[u], acc);
        };
        if (cnt == 8) phase1(IC<8>{});
        else if (cnt == 4) phase1(IC<4>{});
        else if (cnt == 2) phase1(IC<2>{});
        else if (cnt == 1) phase1(IC<1>{});
        else phase1(IC<3>{});
        epilogue_operands();
        __builtin_amdgcn_sched_barrier(0);   // keep the scheduler from sinking loads between the MFMAs
        if (TRACE) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            t_loads = wall_clock64();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (cnt == 8) phase2(IC<8>{});
        else if (cnt == 4) phase2(IC<4>{});
        else if (cnt == 2) phase2(IC<2>{});
        else if (cnt == 1) phase2(IC<1>{});
        else phase2(IC<3>{});
    } else {
        // fat tiles (cnt <= 4, host-checked): (RB + CB) x cnt 16-byte loads would not leave room for a second workgroup on
        // the CU, so the A operand is fetched two q-steps ahead of its use while the weight rows of all steps go out early.
        // Loads are issued STEP-MAJOR (B0 A0 B1 A1 B2 B3): vmcnt retires in order, so the MFMAs of step 0 only wait for the
        // first RB + CB loads and run under the rest of the stream (issued weight-rows-first they waited for 3/4 of it)
        f32x4 a[2][4], b[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (u < cnt) {
#pragma unroll
                for (int c = 0; c < CB; ++c) b[u][c] = *reinterpret_cast<gcf4 *>(bp[c] + u * bstep);
                if (u < 2) {
#pragma unroll
                    for (int r = 0; r < RB; ++r) a[u][r] = *reinterpret_cast<gcf4 *>(ap[r] + u * astep);
                }
            }
        __builtin_amdgcn_sched_barrier(0);
        if (TRACE) t_loads = wall_clock64();
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (u < cnt) {
                skinny16_mfma_step<RB, CB>(a[u & 1], b[u], acc);
                if (u + 2 < cnt) {
#pragma unroll
                    for (int r = 0; r < RB; ++r) a[u & 1][r] = *reinterpret_cast<gcf4 *>(ap[r] + (u + 2) * astep);
                }
                __builtin_amdgcn_sched_barrier(0);   // keep step u's MFMAs ahead of the waits of step u+1
            }
        epilogue_operands();   // issued behind the last MFMAs (not live during the main loop: registers)
    }
    if (TRACE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tr[2] = wall_clock64();
        if (lane == 0) { wave_t[wave][0] = t_loads; wave_t[wave][1] = tr[2]; }
    }

#pragma unroll
    for (int k = 0; k < NBLK; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][k * 4 + r][lane] = acc[k][r];
    __syncthreads();
    if (TRACE) tr[3] = wall_clock64();

    gf *out = (gf *)P(SD_OUT);
    const int out_stride = I(SD_OUT_STRIDE);
#pragma unroll
    for (int rr = 0; rr < (active ? RPW : 0); ++rr) {
        const int r = wave * RPW + rr;
        float v = red[0][r][lane];
#pragma unroll
        for (int w = 1; w < W; ++w) v += red[w][r][lane];
        const int blk = r >> 2, rb = blk / CB, cb = blk - rb * CB;
        const int row = mt * ROWS + rb * 16 + lg * 4 + (r & 3);
        const bool ok = n_ok[cb] && row < M;
        v += e_add[rr];
        if (gate) {
            if ((flags & SDF_PRE) && ok) {
                const long ip = (long)row * I(SD_PRE_STRIDE) + n[cb];
                ((gf *)P(SD_PRE))[pre_tw ? tiled_index(ip, pre_tw) : ip] = v;
            }
            v += e_cls[rr];
            const float partner = __shfl_xor(v, 8);
            if ((li & 8) == 0 && ok) {
                const float g = gate_act(v, partner);
                const int t16 = tile * CB + cb;
                const int tiles_per_group = gateD >> 3;
                const int group = sdiv(t16, tiles_per_group), ch0 = (t16 - group * tiles_per_group) << 3;
                const long io = (long)row * out_stride + group * gateD + ch0 + (li & 7);
                out[out_tw ? tiled_index(io, out_tw) : io] = g;
            }
        } else {
            if (flags & SDF_RELU) v = v > 0.f ? v : 0.f;
            if (ok) {
                const long io = (long)row * out_stride + n[cb];
                out[out_tw ? tiled_index(io, out_tw) : io] = v;
            }
        }
    }
    if (TRACE && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tr[4] = wall_clock64();
        const size_t slot = (size_t)(batch.start[7] % (int)TRACE_LAUNCHES) * TRACE_WGS + bx;   // no atomics: slot = (launch, workgroup)
        if ((unsigned)bx < TRACE_WGS) {
            for (int k = 0; k < 5; ++k) g_trace[(size_t)slot * TRACE_REC + k] = tr[k];
            for (int k = 0; k < W; ++k) {
                g_trace[(size_t)slot * TRACE_REC + 6 + 2 * k] = wave_t[k][0];
                g_trace[(size_t)slot * TRACE_REC + 7 + 2 * k] = wave_t[k][1];
            }
            g_trace[(size_t)slot * TRACE_REC + 5] = ((unsigned long long)z << 48) | ((unsigned long long)tile << 32) |
                                            ((unsigned long long)I(SD_CNT) << 24) | (unsigned)(gridDim.x * gridDim.y * gridDim.z);
        }
    }
}

// ---- host side of the fast path ----
constexpr size_t SKINNY_ZERO_FLOATS = 1 << 16;
static float *g_zero[16] = {};
hipError_t skinny_init(int device) {   // called from ts_ctx_create (never during stream capture)
    if (device < 0 || device >= 16) return hipErrorInvalidDevice;
    if (g_zero[device]) return hipSuccess;
    float *z = nullptr;
    hipError_t e = hipMalloc(&z, SKINNY_ZERO_FLOATS * sizeof(float));
    if (e != hipSuccess) return e;
    e = hipMemset(z, 0, SKINNY_ZERO_FLOATS * sizeof(float));
    if (e != hipSuccess) return e;
    g_zero[device] = z;
    if (knobs().skinny_trace) {
        unsigned long long *t = nullptr;
        e = hipMalloc(&t, TRACE_SLOTS * TRACE_REC * sizeof(unsigned long long));
        if (e != hipSuccess) return e;
        e = hipMemset(t, 0, TRACE_SLOTS * TRACE_REC * sizeof(unsigned long long));
        if (e != hipSuccess) return e;
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &t, sizeof(t));
        if (e != hipSuccess) return e;
        g_trace_host = t;
    }
    return hipSuccess;
}

const float *skinny_zero_buffer(int device) { return device >= 0 && device < 16 ? g_zero[device] : nullptr; }

static inline void put_ptr(SkinnyDesc &d, int k, const void *p) {
    const uint64_t v = (uint64_t)(uintptr_t)p;
    d.w[k] = (uint32_t)v;
    d.w[k + 1] = (uint32_t)(v >> 32);
}
static inline bool fits_i32(long v) { return v >= 0 && v < (1l << 31); }

// false if the problem does not meet the fast kernel's shape constraints
static bool skinny_pack_desc(const SkinnyParams &p, int W, const float *zero, SkinnyDesc &d) {
    if (p.Ktot % (16 * W) != 0 || p.nseg < 1) return false;
    const int cnt = p.Ktot / (16 * W);
    if (cnt < 1 || (cnt > 4 && cnt != 8)) return false;
    if ((size_t)p.N > SKINNY_ZERO_FLOATS || (size_t)p.Ktot > SKINNY_ZERO_FLOATS || p.M < 1 || p.M > 4096) return false;
    if (!fits_i32(p.ldw) || !fits_i32(p.add1_stride) || !fits_i32(p.add2_stride) || !fits_i32(p.add3_stride) ||
        !fits_i32(p.out_stride) || !fits_i32(p.pre_stride))
        return false;
    std::memset(&d, 0, sizeof(d));
    d.w[SD_M] = p.M;
    d.w[SD_N] = p.N;
    d.w[SD_CNT] = cnt;
    d.w[SD_FLAGS] = (p.epi == EPI_GATE ? SDF_GATE : 0) | (p.relu ? SDF_RELU : 0) | (p.pre ? SDF_PRE : 0);
    d.w[SD_GATED] = p.gateD;
    d.w[SD_GX] = p.grid_x;
    d.w[SD_GY] = p.grid_y;
    d.w[SD_NSEG] = p.nseg;
    put_ptr(d, SD_W, p.W);
    d.w[SD_LDW] = (uint32_t)p.ldw;
    put_ptr(d, SD_BIAS, p.bias ? p.bias : zero);
    put_ptr(d, SD_ADD1, p.add1 ? p.add1 : zero);
    d.w[SD_ADD1_STRIDE] = p.add1 ? (uint32_t)p.add1_stride : 0;
    d.w[SD_ADD1_SHIFT] = p.add1 ? p.add1_shift : 0;
    put_ptr(d, SD_ADD2, p.add2 ? p.add2 : zero);
    d.w[SD_ADD2_STRIDE] = p.add2 ? (uint32_t)p.add2_stride : 0;
    d.w[SD_ADD2_SHIFT] = p.add2 ? p.add2_shift : 0;
    put_ptr(d, SD_ADD3, p.add3 ? p.add3 : zero);
    d.w[SD_ADD3_STRIDE] = p.add3 ? (uint32_t)p.add3_stride : 0;
    const bool cls = p.epi == EPI_GATE && p.clsrow;
    put_ptr(d, SD_CLS, cls ? p.clsrow : zero);
    d.w[SD_CLS_LD] = cls ? p.cls_ld : 1;
    if (cls && p.cls_ld < 1) return false;
    put_ptr(d, SD_OUT, p.out);
    d.w[SD_OUT_STRIDE] = (uint32_t)p.out_stride;
    put_ptr(d, SD_PRE, p.pre);
    d.w[SD_PRE_STRIDE] = (uint32_t)p.pre_stride;
    put_ptr(d, SD_ZERO, zero);
    auto lg2 = [](int w) { int l = 0; while ((1 << l) < w) ++l; return (w >= 16 && (1 << l) == w) ? l : -1; };
    if (p.w_tiled) {
        if (p.w_tiled * 16 != p.Ktot) return false;
        d.w[SD_WTQ] = p.w_tiled;
    }
    if (p.out_tiled_w) { if (lg2(p.out_tiled_w) < 0) return false; d.w[SD_OUT_TW] = lg2(p.out_tiled_w); }
    if (p.pre_tiled_w) { if (lg2(p.pre_tiled_w) < 0) return false; d.w[SD_PRE_TW] = lg2(p.pre_tiled_w); }
    if (p.add1_tiled_w) { if (lg2(p.add1_tiled_w) < 0 || p.add1_shift) return false; d.w[SD_ADD1_TW] = lg2(p.add1_tiled_w); }
    for (int s = 0; s < p.nseg; ++s) {
        const SkinnySeg &sg = p.seg[s];
        if (sg.len % (16 * cnt) != 0 || !fits_i32(sg.row_stride) || !fits_i32(sg.gidx_stride) || sg.row_shift != 0) return false;
        if (sg.tiled_w && (sg.gidx || !sg.base || sg.tiled_w % 16 || sg.tiled_w / 16 > 0xffff || sg.len != sg.tiled_w)) return false;
        const int k = SD_SEG + s * SD_SEG_WORDS;
        const bool zero_rows = !sg.gidx && !sg.base;   // null dense segment = rows of zeros
        put_ptr(d, k, zero_rows ? zero : sg.base);
        put_ptr(d, k + 2, sg.gidx);
        d.w[k + 4] = zero_rows ? 0 : (uint32_t)sg.row_stride;
        d.w[k + 5] = (uint32_t)sg.gidx_stride;
        d.w[k + 6] = sg.tiled_w ? (SEG_TILED | (sg.tiled_w / 16)) : 0;
        d.w[k + 7] = sg.len / 16;
    }
    return true;
}

// the wide kernel's shape constraints; *Q = q-steps (16 k) of the problem, 0 for a block of zero rows (epilogue only)
static bool skinny_wide_ok(const SkinnyParams &p, int W, int *Q) {
    auto al16 = [](const void *q) { return ((uintptr_t)q & 15) == 0; };
    if (p.M < 1 || p.N % 64 != 0) return false;
    if (p.epi == EPI_GATE && (p.gateD % 8 != 0 || p.N % (2 * p.gateD) != 0)) return false;
    if (!al16(p.bias) || !al16(p.add1) || !al16(p.add2) || !al16(p.add3) || !al16(p.out) || !al16(p.pre)) return false;
    if ((p.add1 && p.add1_stride % 4) || (p.add2 && p.add2_stride % 4) || (p.add3 && p.add3_stride % 4) || p.out_stride % 4 ||
        (p.pre && p.pre_stride % 4))
        return false;
    if (p.epi == EPI_GATE && p.clsrow && (p.cls_ld % 4 || !al16(p.clsrow))) return false;
    if (p.nseg == 1 && !p.seg[0].gidx && !p.seg[0].base) {
        *Q = 0;
        return true;
    }
    if (!p.w_tiled || p.Ktot % 64 != 0 || p.Ktot % (16 * W) != 0) return false;
    const int cnt = p.Ktot / (16 * W);
    if (cnt != 2 && cnt != 4) return false;
    for (int s = 0; s < p.nseg; ++s) {
        const SkinnySeg &sg = p.seg[s];
        if (sg.len % 64 != 0 || !al16(sg.base)) return false;
        if (!sg.tiled_w && sg.row_stride % 4 != 0) return false;
        if (!sg.gidx && !sg.base) return false;   // zero rows inside a real problem: not needed, not supported
    }
    *Q = p.Ktot / 16;
    return true;
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

bool skinny_descriptor_kernel_enabled();

hipError_t launch_skinny_batch(const SkinnyParams *const *ps, int n, hipStream_t stream) {
    if (n < 1 || n > SKINNY_MAX_PROBLEMS) return hipErrorInvalidValue;
    SkinnyBatch b;
    int gx = 0, gy = 0, Q = 0;
    // column-tile width: 16 (more, leaner workgroups per stage) unless a shape needs the 8-granular 32-column kernel
    const int ncol_pref = knobs().skinny_nt;
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
    // fatter workgroup
    if (ncol == 16) {   // Q counts 8-k steps: K = 8 Q; the 16-column kernel keeps 8 accumulators -> at most 8 waves
        const int W16 = Q >= 32 ? 8 : 4;
        const int variant = knobs().skinny_v;
        int dev = 0;
        if (variant != 0 && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16 && g_zero[dev]) {
            SkinnyDescBatch db;
            bool fast = true;
            // ---- wide path (skinny_wide.hip): a coalesced pass whose launch fills the chip with 64 x 64 full-K tiles ----
            const int wide_min = knobs().wide_min;
            {
                int wM = 0, Qmax = 0, Qp[SKINNY_MAX_PROBLEMS];
                bool wide = wide_min > 0;
                for (int i = 0; i < n && wide; ++i) {
                    wM = wM > b.p[i].M ? wM : b.p[i].M;
                    wide = skinny_wide_ok(b.p[i], W16, &Qp[i]);
                    Qmax = Qmax > Qp[i] ? Qmax : Qp[i];
                }
                if (wide && wM >= 64) {
                    int total = 0, items[SKINNY_MAX_PROBLEMS];
                    for (int i = 0; i < 8; ++i) db.start[i] = 0x7fffffff;
                    // one tile per workgroup while the launch fits one workgroup per CU.  Beyond that, workgroups of equal work
                    // (problems with a fraction of the launch's biggest K get several tiles; an epilogue-only tile counts as 4
                    // q-steps), and as many such units per workgroup as it takes to stay within one round: a workgroup's next
                    // tile is prefetched under the epilogue of the current one, a second ROUND of workgroups pays the start-up again
                    auto count_wgs = [&](int mult) {
                        int t = 0;
                        for (int i = 0; i < n; ++i) {
                            const int cost = Qp[i] > 4 ? Qp[i] : 4;
                            const int eq = Qmax / cost < 1 ? 1 : (Qmax / cost > 4 ? 4 : Qmax / cost);
                            items[i] = mult == 0 ? 1 : eq * mult;
                            const int tiles = (b.p[i].N / 64) * ((b.p[i].M + 63) / 64);
                            db.start[i] = t;
                            t += (tiles + items[i] - 1) / items[i];
                        }
                        return t;
                    };
                    total = count_wgs(0);
                    for (int mult = 1; total > 256 && mult <= 8; ++mult) total = count_wgs(mult);
                    if (total >= wide_min) {
                        for (int i = 0; i < n && wide; ++i) {
                            wide = skinny_pack_desc(b.p[i], W16, g_zero[dev], db.d[i]);
                            if (Qp[i] == 0) db.d[i].w[SD_WTQ] = 0, db.d[i].w[SD_NSEG] = 0;   // zero rows: epilogue only
                            db.d[i].w[SD_ITEMS] = items[i];
                        }
                        for (int i = n; i < SKINNY_MAX_PROBLEMS; ++i) std::memset(&db.d[i], 0, sizeof(SkinnyDesc));
                        if (wide) {
                            db.start[6] = db.start[7] = 0;
                            const int wide_abl = knobs().wide_ablate;
                            db.start[0] = (g_trace_host ? wide_abl : 0) | (knobs().wide_pair ? 8 : 0);   // trace builds only: 2 = no loads, 4 = no MFMAs (problem 0 always starts at workgroup 0)
                            if (g_trace_host) {
                                const uint64_t rec = (uint64_t)(uintptr_t)(g_trace_host + (size_t)(g_trace_seq++ % TRACE_LAUNCHES) * TRACE_WGS * TRACE_REC);
                                db.start[6] = (int)(uint32_t)rec;
                                db.start[7] = (int)(uint32_t)(rec >> 32);
                            }
                            return launch_skinny_wide(db, total, stream, g_trace_host != nullptr);
                        }
                    }
                }
            }
            // Tile shape (rows x columns in blocks of 16).  One batch (M <= 32): 16-row tiles when the launch then still
            // fits one workgroup per CU (less to fetch per CU), else 32 x 16.  Coalesced batches (M >= 64): the biggest of
            // 64 x 32 / 32 x 32 / 32 x 16 that still spreads the launch over about all CUs (fewer operand bytes per output).
            constexpr int half_max = 256, fat_min = 200;
            const int force_shape = knobs().skinny_shape;   // 11, 21, 22, 42 (tests)
            int maxM = 0, maxcnt = 0;
            bool even = true;
            for (int i = 0; i < n; ++i) {
                maxM = maxM > b.p[i].M ? maxM : b.p[i].M;
                const int c = b.p[i].Ktot / (16 * W16);
                maxcnt = maxcnt > c ? maxcnt : c;
                if (b.p[i].grid_x % 2) even = false;
            }
            auto count = [&](int rows, int cb) {
                int t = 0;
                for (int i = 0; i < n; ++i) t += (b.p[i].grid_x / cb) * ((b.p[i].M + rows - 1) / rows);
                return t;
            };
            int RB = 2, CB = 1;
            if (force_shape) {
                RB = force_shape / 10;
                CB = force_shape % 10;
                if (CB > 2) CB = 2;
                if (RB > 4) RB = 4;
                if (CB == 2 && (!even || maxcnt > 4)) CB = 1;
                if (CB == 1 && RB > 2) RB = 2;
            } else if (maxM <= 32) {
                if (count(16, 1) <= half_max) RB = 1;
            } else if (even && maxcnt <= 4) {
                if (count(64, 2) >= fat_min) { RB = 4; CB = 2; }
                else if (count(32, 2) >= fat_min) { RB = 2; CB = 2; }
            }
            const int rows = RB * 16;
            int total = 0;
            for (int i = 0; i < 8; ++i) db.start[i] = 0x7fffffff;
            for (int i = 0; i < n && fast; ++i) {
                fast = skinny_pack_desc(b.p[i], W16, g_zero[dev], db.d[i]);
                db.start[i] = total;
                total += (b.p[i].grid_x / CB) * ((b.p[i].M + rows - 1) / rows);
            }
            for (int i = n; i < SKINNY_MAX_PROBLEMS; ++i) std::memset(&db.d[i], 0, sizeof(SkinnyDesc));
            if (fast) {
                const dim3 grid(total);
                const int trace = knobs().skinny_trace;
                if (trace) db.start[7] = (int)(g_trace_seq++);
                const int shape = RB * 10 + CB;
#define TS_SK_LAUNCH(Wv, R, C)                                                                                         \
    do {                                                                                                               \
        if (trace) hipLaunchKernelGGL((skinny16_fast_kernel<Wv, R, C, true>), grid, dim3(Wv * 64), 0, stream, db); \
        else hipLaunchKernelGGL((skinny16_fast_kernel<Wv, R, C, false>), grid, dim3(Wv * 64), 0, stream, db);     \
    } while (0)
                if (W16 == 8) {
                    if (shape == 11) TS_SK_LAUNCH(8, 1, 1);
                    else if (shape == 21) TS_SK_LAUNCH(8, 2, 1);
                    else if (shape == 22) TS_SK_LAUNCH(8, 2, 2);
                    else TS_SK_LAUNCH(8, 4, 2);
                } else {
                    if (shape == 11) TS_SK_LAUNCH(4, 1, 1);
                    else if (shape == 21) TS_SK_LAUNCH(4, 2, 1);
                    else if (shape == 22) TS_SK_LAUNCH(4, 2, 2);
                    else TS_SK_LAUNCH(4, 4, 2);
                }
#undef TS_SK_LAUNCH
                return hipGetLastError();
            }
        }
        for (int i = 0; i < n; ++i) {   // the generic kernels read row-major operands only
            const SkinnyParams &q = b.p[i];
            bool tiled = q.w_tiled || q.out_tiled_w || q.pre_tiled_w || q.add1_tiled_w;
            for (int sgi = 0; sgi < q.nseg; ++sgi) tiled = tiled || q.seg[sgi].tiled_w;
            if (tiled) return hipErrorInvalidValue;
        }
        if (W16 == 8) hipLaunchKernelGGL(skinny16_kernel<8>, grid, dim3(512), 0, stream, b);
        else hipLaunchKernelGGL(skinny16_kernel<4>, grid, dim3(256), 0, stream, b);
        return hipGetLastError();
    }
    const int W = Q >= 64 ? 16 : (Q >= 32 ? 8 : 4);
    if (W >= 16) hipLaunchKernelGGL(skinny_gemm_kernel<16>, grid, dim3(1024), 0, stream, b);
    else if (W >= 8) hipLaunchKernelGGL(skinny_gemm_kernel<8>, grid, dim3(512), 0, stream, b);
    else hipLaunchKernelGGL(skinny_gemm_kernel<4>, grid, dim3(256), 0, stream, b);
    return hipGetLastError();
}

// copies the TRACE records to the host and resets the counter; returns the number of records
int skinny_trace_read(unsigned long long *out, int max_records) {
    unsigned long long *t = nullptr;
    if (hipMemcpyFromSymbol(&t, HIP_SYMBOL(g_trace), sizeof(t)) != hipSuccess || !t) return -1;
    size_t n = (size_t)(g_trace_seq < TRACE_LAUNCHES ? g_trace_seq : TRACE_LAUNCHES) * TRACE_WGS;
    if (n > (size_t)max_records) n = max_records;
    if (n && hipMemcpy(out, t, n * TRACE_REC * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (hipMemset(t, 0, TRACE_SLOTS * TRACE_REC * sizeof(unsigned long long)) != hipSuccess) return -1;
    return (int)n;   // slots (launch-major); empty slots are all zero
}

void skinny_tile_weights(const float *W, int N, int K, long ldw, int epi, int gateD, float *out) {
    const int nt = (N + 15) / 16, Q = K / 16;
    for (int t = 0; t < nt; ++t)
        for (int li = 0; li < 16; ++li) {
            int n;
            if (epi == EPI_GATE) {   // same column order as the kernel's gate tiles
                const int tiles_per_group = gateD >> 3;
                const int group = t / tiles_per_group, ch0 = (t - group * tiles_per_group) << 3;
                n = group * 2 * gateD + (li >> 3) * gateD + ch0 + (li & 7);
            } else {
                n = t * 16 + li;
            }
            for (int q = 0; q < Q; ++q)
                for (int lg = 0; lg < 4; ++lg)
                    for (int e = 0; e < 4; ++e)
                        out[(((size_t)t * Q + q) * 64 + li + 16 * lg) * 4 + e] = n < N ? W[(size_t)n * ldw + 16 * q + 4 * lg + e] : 0.f;
        }
}

bool skinny_descriptor_kernel_enabled() {
    const Knobs &k = knobs();
    return k.skinny_v != 0 && k.skinny_nt != 32 && k.skinny_tiled;
}

hipError_t launch_skinny_gemm(const SkinnyParams &p, hipStream_t stream) {
    const SkinnyParams *ps[1] = {&p};
    return launch_skinny_batch(ps, 1, stream);
}

}  // namespace ts
