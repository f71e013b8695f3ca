// conv_gemm_f32 — implicit-GEMM 1-D convolution on the gfx950 fp32 matrix cores.
//
// Replaces, for the whole conv stacks of the body path, the PyTorch ops
//   nn.Conv1d k3/s1/p1, k4/s2/p1, k1 ; nn.ConvTranspose1d k4/s2/p1 ; BatchNorm1d(eval) ; LeakyReLU(0.2) ; ReLU ;
//   the `h + x` of Res_CNR_Stack            (reference: nets/spg/vqvae_modules.py:87-212, nets/spg/vqvae_1d.py)
// BatchNorm (and the parallel residual convolution of the down/up layers) is folded into the packed weights on the
// host (models.cpp), so one launch = one reference layer incl. its activation and residual add.
//
// Mapping to CDNA4:
//   * v_mfma_f32_32x32x2_f32: exact fp32 (an fmaf chain per output), 64 cycles / instruction / SIMD = 157 TFLOP/s
//     chip peak — the roof this kernel is measured against.
//   * a workgroup (256 threads = 4 waves, one per SIMD) owns a BM x BN output tile; K advances in chunks of 32
//     floats staged through LDS.  LDS rows are 36 floats wide: a wave's ds_read_b128 (one row per lane, 16 B) then
//     touches 16 distinct 4-bank slots per lane group -> conflict free.
//   * each lane fetches 4 consecutive k with one ds_read_b128 and feeds them to 4 consecutive MFMAs; the two lane
//     halves take k {0..3} and {4..7} of every group of 8 — a permutation of the K order applied identically to A
//     and B, so the product is unchanged and LDS traffic is 1 b128 per 4 MFMA per operand tile.
//   * global -> register -> LDS pipeline two chunks deep: while the MFMAs of chunk i run, chunk i+1 is written to the idle
//     LDS buffer and the 16-byte coalesced loads of chunk i+2 are in flight; one barrier per chunk; MFMA fragments are
//     double-buffered in registers (see the comment at the main loop for how the non-MFMA instruction count is kept low).
//   * convolution taps / stride / transposed-conv phases are "segments": (row shift, channel range) pairs, so only
//     valid taps are multiplied (no zero-insertion for ConvTranspose, no wasted taps for stride 2).
#include "conv_tile.h"
#include <cstdlib>

namespace ts {

// one BM x BN output tile at (m0, n0) of problem / group `zidx`; smem: 2 * (BM + BN) * (BK + 4) floats of LDS
template <int BM, int BN, int WM, int WN, int BK = 32>
__device__ __forceinline__ void conv_tile(const ConvParams &p, const int zidx, const int m0, const int n0, float *smem) {
    constexpr int LDS_LD = BK + 4;   // 36 (68) floats: conflict-free row pitch for ds_read_b128
    constexpr int KC = BK / 32;      // 32-float column blocks per chunk
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int PA = BM / 32, PB = BN / 32;   // 32-row load passes per operand
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");

    float (*As)[BM][LDS_LD] = reinterpret_cast<float (*)[BM][LDS_LD]>(smem);
    float (*Bs)[BN][LDS_LD] = reinterpret_cast<float (*)[BN][LDS_LD]>(smem + 2 * BM * LDS_LD);

    const ConvGroup &g = p.g[p.zdiv > 0 ? 0 : zidx];
    const ConvTilePtrs tp = conv_tile_ptrs(p, g, zidx);
    const float *gx = tp.x, *gw = tp.w;
    const long ldw = p.ldw > 0 ? p.ldw : p.Ktot;
    const int w_rows = p.w_rows > 0 ? p.w_rows : 0x7fffffff;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

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
    constexpr int NQ = BK / 8;
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
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PA; ++i)
#pragma unroll
            for (int c = 0; c < KC; ++c) *reinterpret_cast<f32x4 *>(&As[buf][i * 32 + lrow][lc4 + c * 32]) = ra[i][c];
#pragma unroll
        for (int i = 0; i < PB; ++i)
#pragma unroll
            for (int c = 0; c < KC; ++c) *reinterpret_cast<f32x4 *>(&Bs[buf][i * 32 + lrow][lc4 + c * 32]) = rb[i][c];
    };
    f32x4 fa[2][TM], fb[2][TN];
    auto read_frags = [&](int buf, int q, int slot) {
#pragma unroll
        // volatile: keeps each fragment ONE ds_read_b128.  Left alone the compiler splits the vector load into
        // ds_read2_b32 pieces, and 32-bit reads of a 32-row column conflict 2-way on any 16-byte-aligned row pitch
        // (rows li and li+16 share a bank); the 128-bit read is served 16 lanes at a time and is conflict free.
        for (int i = 0; i < TM; ++i)
            fa[slot][i] = *(const volatile lds_f32x4 *)__builtin_assume_aligned(&As[buf][wm * WM + i * 32 + li][q * 8 + lh * 4], 16);
#pragma unroll
        for (int j = 0; j < TN; ++j)
            fb[slot][j] = *(const volatile lds_f32x4 *)__builtin_assume_aligned(&Bs[buf][wn * WN + j * 32 + li][q * 8 + lh * 4], 16);
    };
    auto mfma_q = [&](int slot) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[slot][j][e], fa[slot][i][e], acc[i][j], 0, 0, 0);
    };
    // ---- the same pieces one at a time, for the hand-ordered steady state below ----
    constexpr int MF = TM * TN * 4;          // MFMAs of one group of 8 k
    constexpr int NR = (PA + PB) * KC;       // staging registers (16 B per thread each)
    constexpr int NF = TM + TN;              // fragments of a group
    auto mfma_one = [&](int slot, int k) {   // k-th MFMA of a group, in mfma_q's order
        const int e = k / (TM * TN), ij = k % (TM * TN), i = ij / TN, j = ij % TN;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[slot][j][e], fa[slot][i][e], acc[i][j], 0, 0, 0);
    };
    auto read_one = [&](int buf, int q, int slot, int f) {
        if (f < TM)
            fa[slot][f] = *(const volatile lds_f32x4 *)__builtin_assume_aligned(&As[buf][wm * WM + f * 32 + li][q * 8 + lh * 4], 16);
        else
            fb[slot][f - TM] = *(const volatile lds_f32x4 *)__builtin_assume_aligned(&Bs[buf][wn * WN + (f - TM) * 32 + li][q * 8 + lh * 4], 16);
    };
    auto move_one = [&](int buf, int r) {    // staging register r: its chunk to LDS, then refilled with the chunk after
        if (r < PA * KC) {
            const int i = r / KC, c = r % KC;
            *reinterpret_cast<f32x4 *>(&As[buf][i * 32 + lrow][lc4 + c * 32]) = ra[i][c];
            ra[i][c] = *reinterpret_cast<const f32x4 *>(pa[i] + c * 32);
        } else {
            const int i = (r - PA * KC) / KC, c = (r - PA * KC) % KC;
            *reinterpret_cast<f32x4 *>(&Bs[buf][i * 32 + lrow][lc4 + c * 32]) = rb[i][c];
            rb[i][c] = *reinterpret_cast<const f32x4 *>(pb[i] + c * 32);
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
    // Steady state, hand ordered (sched_barrier after every step): chunk it+1 -> LDS, chunk it+2 -> registers, MFMAs of chunk it.
    // The matrix pipe gives an older wave priority over a younger one, so the second workgroup of a CU only runs in the first
    // one's gaps, and its four waves (one per SIMD) are tied together by their barrier: a wave's own instruction stream has
    // to keep the pipe fed.  Hence at most one or two side operations behind each MFMA (64 cycles of pipe): the LDS write and
    // global refill of one staging register, or one fragment read; never a run of them with a single MFMA in flight.
    for (; it + 2 < nchunks; ++it) {
        advance();
        __builtin_amdgcn_sched_barrier(0);
        constexpr int OPS0 = NR + NF, PER0 = (OPS0 + MF - 1) / MF;
#pragma unroll
        for (int k = 0; k < MF; ++k) {       // group 0: staging traffic + the fragments of group 1
            mfma_one(0, k);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int o = k * PER0; o < (k + 1) * PER0 && o < OPS0; ++o) {
                if (o < NR) move_one(buf ^ 1, o);
                else read_one(buf, 1, 1, o - NR);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 1; q + 1 < NQ; ++q) {   // middle groups: the fragments of the next group
#pragma unroll
            for (int k = 0; k < MF; ++k) {
                mfma_one(q & 1, k);
                if (k < NF) {
                    __builtin_amdgcn_sched_barrier(0);
                    read_one(buf, q + 1, (q + 1) & 1, k);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        {                                    // last group: barrier half way, then group 0's fragments of the next chunk
            constexpr int H = (MF - NF) < MF / 2 ? (MF - NF) : MF / 2;
#pragma unroll
            for (int k = 0; k < H; ++k) mfma_one((NQ - 1) & 1, k);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = H; k < MF; ++k) {
                mfma_one((NQ - 1) & 1, k);
                if (k - H < NF) {
                    __builtin_amdgcn_sched_barrier(0);
                    read_one(buf ^ 1, 0, 0, k - H);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
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

    conv_tile_epilogue<TM, TN>(p, g, tp, acc, m0 + wm * WM, n0 + wn * WN, li, lh);   // conv_tile.h
}

template <int BM, int BN, int WM, int WN, int BK = 32>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const ConvParams p) {
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * (BK + 4)];
    conv_tile<BM, BN, WM, WN, BK>(p, blockIdx.z, blockIdx.x * BM, blockIdx.y * BN, smem);
}

// Banded launch: the rows of the output are cut into two bands — 128 x 128 tiles where whole rounds of 512 resident workgroups
// fit, the small shape S for what is left, so that the last, partly filled round of a layer is made of short tiles (2 400 tiles
// of 128 x 128 are 4.69 rounds).  Workgroup ids run band by band; inside a band M tiles fastest, then N, then group.
template <int SBM, int SBN, int SWM, int SWN>
__global__ __launch_bounds__(256) void conv_gemm_banded_kernel(const ConvParams p, const ConvBands bands) {
    __shared__ __attribute__((aligned(16))) float smem[2 * (128 + 128) * 36];
    const int big = (int)blockIdx.x < bands.first_small;
    const int local = big ? blockIdx.x : blockIdx.x - bands.first_small;
    const int mt = big ? bands.mt_big : bands.mt_small;
    const int nt = big ? (p.N + 127) / 128 : (p.N + SBN - 1) / SBN;
    const int rest = local / mt, m = local - rest * mt;
    const int z = rest / nt, n = rest - z * nt;
    if (big) conv_tile<128, 128, 64, 64>(p, z, m * 128, n * 128, smem);
    else conv_tile<SBM, SBN, SWM, SWN>(p, z, bands.mt_big * 128 + m * SBM, n * SBN, smem);
}

double conv_gemm_flops(const ConvParams &p) { return 2.0 * p.M * (double)p.N * p.Ktot * p.ngroups; }


static int pick_tile(const ConvParams &p) {
    // Cost model: the 256 CUs pull tiles dynamically, so a launch lasts about ceil(tiles / 256) tile-times on the busiest
    // CU; a tile-time is its MACs over the tile shape's measured intrinsic efficiency (tools/tune_conv.py on 4096^3:
    // 128x128 134 TF, 128x64 / 64x128 129, 64x64 123).  Small / mid-size layers want many small tiles (tail), big ones
    // the 128x128 tile (half the L2->LDS traffic per MAC).  The tall 160x128 / 96x128 tiles (ids 6, 7) stay available for
    // tuning: 160x128 turns the paired 1024-channel VQ layers at batch 32 into one full wave of 240 tiles and wins in
    // isolation (109 vs 102 TF, warm caches), but with one workgroup per CU nothing hides its cold-weight prologue
    // inside the real layer sequence (304-335 us vs 298-310 us for 64x64), so the model does not pick it.
    struct Cand { int id, bm, bn; double eff; };
    static const Cand cands[] = {{1, 128, 128, 1.00}, {2, 64, 64, 0.92}, {3, 128, 64, 0.95}, {4, 64, 128, 0.95}};
    int best = 2;
    double best_cost = 1e300;
    for (const Cand &c : cands) {
        const long tiles = (long)((p.M + c.bm - 1) / c.bm) * ((p.N + c.bn - 1) / c.bn) * p.ngroups;
        const double cost = (double)((tiles + 255) / 256) * c.bm * c.bn / c.eff;
        if (cost < best_cost) { best_cost = cost; best = c.id; }
    }
    return best;
}

// Bands for a layer that the cost model gives to 128 x 128 tiles: big tiles for as many whole rounds of 512 resident workgroups
// as fit, 64 x 128 tiles for the rows that are left (measured against 64 x 64 and 128 x 64: tools/conv_mix_ab.sh).
bool conv_gemm_plan_bands(const ConvParams &p, ConvBands &bd) {
    const int slots = 512;
    const int MT = (p.M + 127) / 128, NT = ((p.N + 127) / 128) * p.ngroups;
    const long total = (long)MT * NT;
    const int rounds = (int)(total / slots);
    if (p.zdiv > 0 || rounds < 1 || total % slots == 0) return false;
    const int mb = (int)((long)rounds * slots / NT);   // M tiles of 128 rows given to the full rounds
    if (mb >= MT) return false;
    bd.mt_big = mb;
    bd.first_small = mb * NT;
    bd.mt_small = (p.M - mb * 128 + 63) / 64;
    bd.total = bd.first_small + bd.mt_small * NT;      // 64 x 128 tiles: as many column tiles as the big ones
    return true;
}

// host-only view of the launch plan (tests): would `launch_conv_gemm(p, 0, ...)` band this layer, and how
bool conv_gemm_band_plan(const ConvParams &p, ConvBands &bd) { return pick_tile(p) == 1 && conv_gemm_plan_bands(p, bd); }

hipError_t launch_conv_gemm(const ConvParams &p_in, int tile, hipStream_t stream) {
    ConvParams p = p_in;
    if (!p.zero) {
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) p.zero = skinny_zero_buffer(dev);
    }
    dim3 block(256);
    auto grid = [&](int bm, int bn) { return dim3((p.M + bm - 1) / bm, (p.N + bn - 1) / bn, p.ngroups); };
    // zero buffer (ts::skinny_init, called by ts_ctx_create): 64 Ki floats; parked pointers walk at most Ktot floats of it
    if (!p.zero || p.g[0].nseg > 4 || p.Ktot > 60000) return hipErrorInvalidValue;
    if (tile == 31 || tile == 33 || tile == 39 || tile == 35 || tile == 36) return launch_conv_gemm_ring(p, tile - 30, stream);   // LDS-DMA ring engine (conv_gemm_ring.hip)
    if (tile == 37) {   // ring engine, banded + dealt (falls back to the dealt plain grid when the layer has no band plan)
        ConvBands bd;
        if (!conv_gemm_plan_bands(p, bd)) return launch_conv_gemm_ring(p, 5, stream);
        return launch_conv_gemm_ring_banded(p, bd, stream);
    }
    if (tile == 38) {   // ring engine, whole tiles for the whole units of 256 + a stream-K band for the rest (falls back to the dealt plain grid)
        ConvSK sk;
        if (!conv_gemm_plan_sk(p, sk)) return launch_conv_gemm_ring(p, 5, stream);
        return launch_conv_gemm_ring_sk(p, sk, stream);
    }
    if (tile == 48 || (tile == 0 && knobs().conv_taps48 && conv_taps48_takes(p))) return launch_conv_taps48(p, stream);   // grouped 48-channel taps, unpadded (conv_taps48.hip)
    // layers that take 128 x 128 tiles run on the ring engine (same-box A/B: profiles/r05_notes/): the face generator's GEMMs, and the paired
    // body + hand layers unless TS_CONV_RING_PAIRED=0 (then: the banded launch below)
    if (tile == 0 && knobs().conv_ring > 0 && (p.ngroups == 1 || knobs().conv_ring_paired) && p.zdiv == 0 && pick_tile(p) == 1 && conv_gemm_ring_takes(p)) {
        const int v = knobs().conv_ring;   // 9: the tile plan by tile count (conv_gemm_ring_pick); tiles dealt to the XCDs unless TS_CONV_DEAL=0
        if (v == 9 && knobs().conv_deal) {
            ConvBands bd;
            // a band plan without a single 128-row block in its big band (more than 512 column tiles: a very wide N x groups) is no plan:
            // the banded launch refuses it, so it must not be offered to the pick
            const bool have = knobs().conv_bands && conv_gemm_plan_bands(p, bd) && bd.mt_big >= 1;
            ConvSK sk;
            const bool have_sk = knobs().conv_sk && p.sk_ok && conv_gemm_plan_sk(p, sk);
            const int pick = conv_gemm_ring_pick(p, have ? &bd : nullptr, have_sk ? &sk : nullptr);
            if (pick == 8 || (have_sk && knobs().conv_sk == 2)) return launch_conv_gemm_ring_sk(p, sk, stream);
            return pick == 7 ? launch_conv_gemm_ring_banded(p, bd, stream) : launch_conv_gemm_ring(p, pick == 3 ? 6 : 5, stream);
        }
        // TS_CONV_RING: 9 = the plan by tile count (default), 8 = the 8-wave 128 x 128 tile, 1 / 3 / 5 / 6 = that variant; anything else is
        // not a variant and means the default plan, not a launch error on every layer
        const int variant = v == 9 ? 0 : (v == 8 ? 9 : ((v == 1 || v == 3 || v == 5 || v == 6) ? v : 0));
        return launch_conv_gemm_ring(p, variant, stream);
    }
    if (tile == 0 && pick_tile(p) == 1) {
        const bool banded = knobs().conv_bands;   // TS_CONV_BANDS=0: plain grid (A/B, tests)
        ConvBands bd;
        if (banded && conv_gemm_plan_bands(p, bd)) {
            hipLaunchKernelGGL((conv_gemm_banded_kernel<64, 128, 32, 64>), dim3(bd.total), block, 0, stream, p, bd);
            return hipGetLastError();
        }
    }
    if (tile == 0) tile = pick_tile(p);
    switch (tile) {
        case 1: hipLaunchKernelGGL((conv_gemm_kernel<128, 128, 64, 64>), grid(128, 128), block, 0, stream, p); break;
        case 2: hipLaunchKernelGGL((conv_gemm_kernel<64, 64, 32, 32>), grid(64, 64), block, 0, stream, p); break;
        case 3: hipLaunchKernelGGL((conv_gemm_kernel<128, 64, 64, 32>), grid(128, 64), block, 0, stream, p); break;
        case 4: hipLaunchKernelGGL((conv_gemm_kernel<64, 128, 32, 64>), grid(64, 128), block, 0, stream, p); break;
        case 5:   // 64x64 with 64-deep chunks (all segment lengths must be multiples of 64)
            hipLaunchKernelGGL((conv_gemm_kernel<64, 64, 32, 32, 64>), grid(64, 64), block, 0, stream, p);
            break;
        // tall tiles, waves side by side along N (each 32 columns x the whole tile height): 160x128 turns the three big
        // layer shapes of the VQ stacks at batch 32 (M*N = 2 x 2400x1024 = 2 x 4800x512 = 2 x 9600x256) into exactly 240
        // tiles, one per CU in a single wave, with the L2->LDS traffic per MAC of the 128x128 tile
        case 6: hipLaunchKernelGGL((conv_gemm_kernel<160, 128, 160, 32>), grid(160, 128), block, 0, stream, p); break;
        case 7: hipLaunchKernelGGL((conv_gemm_kernel<96, 128, 96, 32>), grid(96, 128), block, 0, stream, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace ts
