// conv_gemm_f32, LDS-DMA ring engine — the same implicit-GEMM convolution as conv_gemm.hip (same ConvParams, same segments /
// taps / strides, same MFMA, same k order: BIT-IDENTICAL outputs), with the operand path rebuilt around direct global -> LDS
// loads (global_load_lds_dwordx4) instead of global -> VGPR -> ds_write, and 8 waves per 128 x 128 tile.
//
// Replaces the same PyTorch ops as conv_gemm.hip; it takes every layer that gets 128 x 128 tiles — the plain-GEMM layers of the face
// generator: the wav2vec2 encoder blocks' QKV / out-proj / FFN1 / FFN2 (reference: nets/spg/wav2vec.py:76-143, HF
// Wav2Vec2EncoderLayer), its feature convolutions and heads (nets/spg/s2g_face.py:196-224) — and the paired body + hand layers of
// nets/spg/vqvae_modules.py:87-212 (two problems per launch), for which it has conv_gemm.hip's band plan (conv_ring_banded_kernel).
//
// Mapping to CDNA4:
//   * a stage = 32 consecutive k of the tile's 128 activation rows and 128 weight rows, row-major in LDS with a row pitch of 128
//     bytes — the image a wave's LDS-DMA instruction writes (wave-uniform base + lane x 16 B = 1 KB = 8 rows x 128 B): both
//     operands stay ROW-MAJOR in HBM (no layout change anywhere else), a DMA instruction reads 8 whole cache lines;
//   * conflict-free fragment reads without padding: the 16-byte segment s of row r sits at position s ^ ((r >> 1) & 7) of its row
//     — the permutation is applied to the per-lane SOURCE address of the DMA and to the ds_read_b128 address, never to the
//     destination (which is lane-linear by construction); a 16-lane group of a ds_read_b128 then covers all 64 banks exactly once;
//   * ring of NS = 2 stage slots; stage t + 1 is issued behind the barrier that opens stage t; the barrier that opens stage t + 1
//     sits in the middle of stage t's last MFMA group: one barrier per stage, no staging registers, no ds_write, 94-150 VGPRs;
//   * 8 waves per workgroup (32 x 64 outputs each: 4 x 2 waves), two workgroups per CU: four waves per SIMD, two even when a
//     workgroup is alone on its CU in the tail of a launch;
//   * tiles DEALT to the XCDs (1-D grid, ids round-robin over the 8 XCDs, XCD c takes the c-th contiguous eighth of a tile list
//     ordered by column groups, columns fastest: split_tile_of, kernels.h): the 64 workgroups resident on an XCD share their
//     operand tiles in ITS L2.  Nothing on cache-resident layers (M = 19 200: +0 ... 1.6 %), +7-8 % on the feature convolutions
//     (M up to 1 023 936 rows: 124.7 -> 135.0 TFLOP/s), whose rows otherwise cross the fabric once per column tile.
// What was measured on the way (tools/ring_probe.py, profiles/r05_notes/, tools/experiments/README.md): the main loop runs at the
// matrix pipe's rate (3.45 us per stage against 3.41 us at 2.4 GHz; asymptotically 141 TFLOP/s), also for a workgroup alone on its
// CU; what separates a K = 768 layer (115-128 TFLOP/s) from that is K-independent: ~13 us per launch, ~6 us per further round of
// 512 workgroups, and the partly filled last round (900 tiles of the N = 768 layers: 1.76 rounds cost 1.95).  Persistent workgroups
// with the ring turning across tiles, a staggered start of the two workgroups of a CU, whole-row stores through LDS and 16-deep
// stages with 3-4 ring slots were all built, are all bit-identical, and none of them moves those terms.
#include <type_traits>

#include "conv_tile.h"

namespace ts {

__device__ __forceinline__ void ring_glds16(const float *src, float *lds_dst) {   // lds_dst: wave-uniform; lane i lands at + 16 i bytes
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds_dst, 16, 0, 0);
}

// one BM x BN output tile at (m0, n0) of problem / group `zidx`; smem: 2 * (BM + BN) * 32 floats of LDS (ONE object: a second
// one makes hipcc drain vmcnt before every ds_read)
// Stages [t_begin, t_end) of the tile's K walk (a stage = 32 k; the whole walk is [0, Ktot / 32)); `tail(acc, g, tp, mw, nw, li, lh)` gets the
// accumulators: the epilogue for a whole walk (ring_tile below), a partial-sum store / fix-up for a piece of one (the stream-K band).
template <int BM, int BN, int WM, int WN, class Tail>
__device__ __forceinline__ void ring_tile_range(const ConvParams &p, const int zidx, const int m0, const int n0, float *smem, const int t_begin,
                                                const int t_end, Tail tail) {
    constexpr int BK = 32;                     // stage depth (k); the ring has two slots
    constexpr int SEGS = BK / 4;               // 16-byte segments of a row per stage
    constexpr int RPB = 64 / SEGS;             // rows per LDS-DMA instruction (1 KB)
    constexpr int NW = (BM / WM) * (BN / WN);  // waves per workgroup: 4 (one per SIMD) or 8
    constexpr int NA = BM / RPB / NW, NB = BN / RPB / NW;   // DMA instructions per wave and stage
    constexpr int ND = NA + NB;
    constexpr int STAGE = (BM + BN) * BK;      // floats per slot
    constexpr int NQ = BK / 8;                 // MFMA groups (8 k) per stage
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int MF = TM * TN * 4, NF = TM + TN, H = MF / 2;
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
    static_assert(NA >= 1 && NB >= 1 && NA * RPB * NW == BM && NB * RPB * NW == BN, "tile rows split evenly over the waves' DMA instructions");
    static_assert(NF <= MF - H, "next stage's first fragments fit behind the barrier");

    const ConvGroup &g = p.g[p.zdiv > 0 ? 0 : zidx];
    const ConvTilePtrs tp = conv_tile_ptrs(p, g, zidx);
    const float *gx = tp.x, *gw = tp.w;
    const long ldw = p.ldw > 0 ? p.ldw : p.Ktot;
    const int w_rows = p.w_rows > 0 ? p.w_rows : 0x7fffffff;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    auto swz = [](int row) { return (row >> 1) & 7; };

    // ---- loader: DMA instruction j of this wave covers rows (wave * NA + j) * 8 .. + 7 of the A part (same for B); lane -> row
    // lane / 8 of the block, LDS position lane % 8 of that row, i.e. global segment position ^ f(row) ----
    const int drow = lane >> 3, dpos = lane & (SEGS - 1);
    int a_row[NA], a_t[NA];   // (b * Lin) input row base or -1 if the output row is out of range; t * stride
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int m = m0 + (wave * NA + j) * RPB + drow;
        if (m < p.M) {
            const int b = m / p.Lout, t = m - b * p.Lout;
            a_row[j] = b * p.Lin;
            a_t[j] = t * p.stride;
        } else {
            a_row[j] = -1;
            a_t[j] = 0;
        }
    }
    // column (floats) of this lane's segment in DMA instruction j: position ^ f(row in tile); blocks of 8 rows alternate the top bit of f
    auto dcol = [&](int blk) { return (dpos ^ swz(blk * RPB + drow)) << 2; };

    // segment descriptors live in VGPR lanes (v_readlane): no scalar loads competing with LDS for lgkmcnt
    int vd = 0, vc0 = 0, vlen = BK, vnt = 1;
    if (lane < 4) {
        vd = g.seg[lane].d;
        vc0 = g.seg[lane].c0;
        vlen = g.seg[lane].len;
        vnt = g.seg[lane].ntap > 1 ? g.seg[lane].ntap : 1;
    }
    int s = 0, tap = 0, cc = 0;
    int cur_len = __builtin_amdgcn_readlane(vlen, 0), cur_nt = __builtin_amdgcn_readlane(vnt, 0);
    if (t_begin > 0) {   // seek: (segment, tap, stage inside the tap) of stage t_begin — wave-uniform scalar code, a handful of iterations
        int rem = t_begin;
        for (;;) {
            const int per = cur_len / BK, tot = per * cur_nt;
            if (rem < tot) {
                tap = rem / per;
                cc = rem - tap * per;
                break;
            }
            rem -= tot;
            s += 1;
            cur_len = __builtin_amdgcn_readlane(vlen, s & 3);
            cur_nt = __builtin_amdgcn_readlane(vnt, s & 3);
        }
    }
    const float *pa[NA], *pb[NB];
    auto enter_run = [&]() {   // operand pointers of the first stage of (segment s, tap); halo rows and rows beyond M read zeros
        const int sl = s & 3;
        const int d = __builtin_amdgcn_readlane(vd, sl) + tap;
        const int c0 = __builtin_amdgcn_readlane(vc0, sl);
        cur_len = __builtin_amdgcn_readlane(vlen, sl);
        cur_nt = __builtin_amdgcn_readlane(vnt, sl);
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int it = a_t[j] + d;
            const bool ok = a_row[j] >= 0 && it >= 0 && it < p.Lin;
            pa[j] = (ok ? gx + (long)(a_row[j] + it) * p.ldx + c0 : p.zero) + dcol(wave * NA + j);
        }
    };
    enter_run();
#pragma unroll
    for (int j = 0; j < NA; ++j) pa[j] += cc * BK;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int n = n0 + (wave * NB + j) * RPB + drow;
        pb[j] = (n < w_rows ? gw + (long)n * ldw : p.zero) + dcol(wave * NB + j) + (long)t_begin * BK;
    }
    auto advance = [&]() {   // to the next stage of the K walk
        cc += 1;
#pragma unroll
        for (int j = 0; j < NB; ++j) pb[j] += BK;
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
            for (int j = 0; j < NA; ++j) pa[j] += BK;
        }
    };
    auto dma_one = [&](int slot, int o) {   // DMA instruction o of the stage the pointers stand at, into ring slot `slot`
        float *dst = smem + slot * STAGE;
        if (o < NA) ring_glds16(pa[o], dst + (wave * NA + o) * 256);
        else ring_glds16(pb[o - NA], dst + BM * BK + (wave * NB + (o - NA)) * 256);
    };

    // ---- reader: MFMA lane (li, lh) takes row li of a 32-row block, k = 8 q + 4 lh .. + 3 of group q: segment 2 q + lh ----
    const int li = lane & 31, lh = lane >> 5;
    int fbase[NF];   // float index inside a slot of this lane's fragment of group 0; group q: ^ (q << 3)
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int row = f < TM ? wm * WM + f * 32 + li : wn * WN + (f - TM) * 32 + li;
        const int fs = swz(row);
        fbase[f] = (f < TM ? 0 : BM * BK) + row * BK + ((lh ^ (fs & 1)) << 2) + ((fs >> 1) << 3);
    }
    f32x4 fa[2][TM], fb[2][TN];
    auto read_one = [&](int slot, int q, int set, int f) {
        // volatile + LDS-qualified: one ds_read_b128 per fragment (conv_gemm.hip has the reason)
        const float *src = smem + slot * STAGE + (fbase[f] ^ (q << 3));
        const f32x4 v = *(const volatile lds_f32x4 *)__builtin_assume_aligned(src, 16);
        if (f < TM) fa[set][f] = v;
        else fb[set][f - TM] = v;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto mfma_one = [&](int set, int k) {   // k-th MFMA of a group: e-major, the same order as conv_gemm.hip's mfma_q
        const int e = k / (TM * TN), ij = k % (TM * TN), i = ij / TN, j = ij % TN;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[set][j][e], fa[set][i][e], acc[i][j], 0, 0, 0);
    };
    // every DMA load of this wave has landed and every LDS read it issued is done (two slots: nothing is left in flight across a barrier)
    auto wait_all = [&]() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); };

    const int T = t_end - t_begin;   // stages of this walk
    // ---- prologue: the first stage goes out and lands, its first fragments are read ----
#pragma unroll
    for (int o = 0; o < ND; ++o) dma_one(0, o);
    wait_all();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int f = 0; f < NF; ++f) read_one(0, 0, 0, f);

    // ---- stage t out of ring slot `slot`.  Hand ordered (sched_barrier after every step), at most one or two side operations behind
    // each MFMA (64 cycles of pipe): a DMA issue, or one fragment read — never a run of them with a single MFMA in flight.
    // MORE: stage t + 1 exists: it is issued into the other slot (free since the barrier that opened stage t) behind the first MFMAs ----
    auto stage = [&](auto Mc, const int slot) {
        constexpr bool MORE = decltype(Mc)::value;
        if (MORE) advance();   // pointers -> stage t + 1
        __builtin_amdgcn_sched_barrier(0);
        constexpr int OPS0 = ND + NF, PER0 = (OPS0 + MF - 1) / MF;
#pragma unroll
        for (int q = 0; q + 1 < NQ; ++q) {
#pragma unroll
            for (int k = 0; k < MF; ++k) {
                mfma_one(q & 1, k);
                __builtin_amdgcn_sched_barrier(0);
                if (q == 0) {   // group 0: the refill + the fragments of group 1
#pragma unroll
                    for (int o = k * PER0; o < (k + 1) * PER0 && o < OPS0; ++o) {
                        if (o < ND) {
                            if (MORE) dma_one(slot ^ 1, o);
                        } else {
                            read_one(slot, 1, 1, o - ND);
                        }
                    }
                } else if (k < NF) {
                    read_one(slot, q + 1, (q + 1) & 1, k);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // last group: the barrier that opens stage t + 1 half way, then that stage's first fragments
#pragma unroll
        for (int k = 0; k < H; ++k) mfma_one((NQ - 1) & 1, k);
        __builtin_amdgcn_sched_barrier(0);
        if (MORE) {
            wait_all();                        // this wave's loads of stage t + 1 have landed, its reads of stage t are done
            __builtin_amdgcn_s_barrier();      // ... and everybody's
            asm volatile("" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = H; k < MF; ++k) {
            mfma_one((NQ - 1) & 1, k);
            if (MORE && k - H < NF) {
                __builtin_amdgcn_sched_barrier(0);
                read_one(slot ^ 1, 0, 0, k - H);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    int slot = 0;
    for (int t = 0; t + 1 < T; ++t) {
        stage(std::true_type{}, slot);
        slot ^= 1;
    }
    stage(std::false_type{}, slot);

    tail(acc, g, tp, m0 + wm * WM, n0 + wn * WN, li, lh);
}

// one whole BM x BN output tile: the full K walk, then the epilogue (conv_tile.h)
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void ring_tile(const ConvParams &p, const int zidx, const int m0, const int n0, float *smem) {
    constexpr int TM = WM / 32, TN = WN / 32;
    ring_tile_range<BM, BN, WM, WN>(p, zidx, m0, n0, smem, 0, p.Ktot / 32,
                                    [&](f32x16 (&acc)[TM][TN], const ConvGroup &g, const ConvTilePtrs &tp, int mw, int nw, int li, int lh) {
                                        conv_tile_epilogue<TM, TN>(p, g, tp, acc, mw, nw, li, lh);
                                    });
}

// two workgroups per CU.  XCD = false: grid (row tiles, column tiles, problems).  XCD = true: 1-D grid of 8 ceil(tiles / 8) per problem, tiles
// dealt to the XCDs by split_tile_of (kernels.h): an XCD's 64 resident workgroups are blocks of up to 8 x 8 tiles that share their operand
// tiles in its L2 — the same tiles, the same bits
template <int BM, int BN, int WM, int WN, bool XCD>
__global__ __launch_bounds__(64 * (BM / WM) * (BN / WN), (BM / WM) * (BN / WN) / 2) void conv_ring_kernel(const ConvParams p) {
    __shared__ __attribute__((aligned(1024))) float smem[2 * (BM + BN) * 32];
    int tx = blockIdx.x, ty = blockIdx.y;
    if constexpr (XCD) {
        const int nt = (p.N + BN - 1) / BN;
        if (!split_tile_of((int)blockIdx.x, (p.M + BM - 1) / BM, nt, nt < 8 ? nt : 8, tx, ty)) return;
    }
    ring_tile<BM, BN, WM, WN>(p, blockIdx.z, tx * BM, ty * BN, smem);
}

// Banded + dealt: rows [0, mt_big * 128) in 128 x 128 tiles for the whole rounds of 512 resident workgroups, the rows after them in
// 64 x 128 tiles (conv_gemm.hip's plan_bands: the last, partly filled round of a layer is made of short tiles); per problem each band's
// tiles are dealt to the XCDs (split_tile_of), every region padded to a multiple of 8 ids so that id mod 8 stays the XCD.
// ids: [problem][big band, big8 ids] ..., then [problem][small band, small8 ids] ...
__global__ __launch_bounds__(512, 4) void conv_ring_banded_kernel(const ConvParams p, const ConvBands bd, const int big8, const int small8) {
    __shared__ __attribute__((aligned(1024))) float smem[2 * (128 + 128) * 32];
    const int nt = (p.N + 127) / 128, gw = nt < 8 ? nt : 8;
    int id = blockIdx.x, tx, ty;
    const int nbig = big8 * p.ngroups;
    if (id < nbig) {
        const int z = id / big8;
        if (!split_tile_of(id - z * big8, bd.mt_big, nt, gw, tx, ty)) return;
        ring_tile<128, 128, 32, 64>(p, z, tx * 128, ty * 128, smem);
    } else {
        id -= nbig;
        const int z = id / small8;
        if (!split_tile_of(id - z * small8, bd.mt_small, nt, gw, tx, ty)) return;
        ring_tile<64, 128, 32, 32>(p, z, bd.mt_big * 128 + tx * 64, ty * 128, smem);
    }
}

// ---- stream-K band (the partly filled last round) -------------------------------------------------------------
// A launch of whole tiles costs ceil(tiles / 256) tile times on 256 CUs (a workgroup alone on its CU runs at the full pipe rate): the 900
// tiles of an N = 768 layer at M = 19 200 cost 4 where 3.52 would do.  Here the row tiles that fill whole units of 256 stay whole tiles
// (dealt to the XCDs as before); the rows after them become ONE list of (tile, stage) iterations, cut into `wsk` equal runs, one
// workgroup each — every CU ends up with the same number of MFMA stages.  A run covers the tail of one tile and the head of the next
// (or a slice of one tile, or whole tiles in between).  Every PIECE of a split tile is written to p.sk_ws as a partial accumulator and
// announced on the tile's counter; the workgroup that arrives LAST (whoever that is) sums the pieces in k order — always the same
// order, whatever the arrival order: deterministic run to run — and runs the epilogue.  Nobody ever waits for another workgroup: no
// spinning, nothing to deadlock, and a workgroup slowed down by a neighbour's kernel delays only its own tiles.  The last
// arriver clears the counter, so nothing has to be reset between launches.  NOT bit-identical with the whole-tile plans (a split
// tile is P0 + P1 (+ P2) instead of one running sum), and WHICH tiles are split depends on M: a row's bits depend on the batch it
// rides in.  Only callers that set ConvParams::sk_ok get this plan (the face generator: tolerance-only GEMMs); the body path, whose
// results are bit-identical across pass sizes, never does.
//
// Coherence: the XCDs' L2s are not coherent with each other, and an agent-scope release / acquire fence writes back / invalidates the
// WHOLE L2 (buffer_wbl2 / buffer_inv) — the operand tiles the whole-tile workgroups next door are living on (measured: the band cost 1.2
// tile times instead of 0.5 with fences; and sc1 write-through stores + vmcnt(0) alone are NOT a release: a reader on another XCD saw
// stale memory).  So no tile is ever split ACROSS XCDs: the band's tile list is dealt to the 8 XCDs in whole tiles (XCD c: tiles
// [c Ts / 8, (c + 1) Ts / 8)), each XCD's iterations are cut into wsk / 8 equal runs, and band workgroup q works on XCD q % 8 (workgroup
// ids go round-robin over the XCDs).  All pieces of a tile then meet in ONE L2: plain stores (complete at the L2: vmcnt(0)), sc1 loads
// (past the CU's L1), agent-scope atomics on the counter.
__device__ __forceinline__ void sk_store_partial(f32x16 (&acc)[1][2], float *dst) {
    f32x4 *d = reinterpret_cast<f32x4 *>(dst) + threadIdx.x;   // chunk c of thread tid at [c][tid]: a wave stores 1 KB runs (512 threads per workgroup)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) d[(j * 4 + q) * 512] = f32x4{acc[0][j][4 * q], acc[0][j][4 * q + 1], acc[0][j][4 * q + 2], acc[0][j][4 * q + 3]};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <bool FIRST>
__device__ __forceinline__ void sk_load_partial(f32x16 (&acc)[1][2], const float *src) {
    const float *s_ = src + threadIdx.x * 4;
#pragma unroll
    for (int h = 0; h < 2; ++h) {   // four loads and the wait for them per block (the compiler cannot see an asm load's latency)
        f32x4 v[4];
        const float *b = s_ + h * 4 * 2048;
        asm volatile(
            "global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\tglobal_load_dwordx4 %2, %6, off sc1\n\t"
            "global_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
            : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
            : "v"(b), "v"(b + 2048), "v"(b + 2 * 2048), "v"(b + 3 * 2048)
            : "memory");
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[0][h][4 * q + r] = FIRST ? v[q][r] : acc[0][h][4 * q + r] + v[q][r];
    }
}

__global__ __launch_bounds__(512, 4) void conv_ring_sk_kernel(const ConvParams p, const ConvSK sk) {
    __shared__ __attribute__((aligned(1024))) float smem[2 * (128 + 128) * 32];
    const int nt = (p.N + 127) / 128, gw = nt < 8 ? nt : 8;
    int id = blockIdx.x, tx = 0, ty = 0;
    const int ndp = sk.dp8 * p.ngroups;
    const bool dp = id < ndp;
    const SkRuns R{sk.mt_sk * nt, sk.stages, sk.wsk >> 3};
    // a whole tile is a run of exactly one piece [0, stages) — one instance of the tile code serves both kinds of workgroup
    int z, c = 0, r = 0, it0 = 0, it1 = sk.stages;
    if (dp) {
        z = id / sk.dp8;
        if (!split_tile_of(id - z * sk.dp8, sk.mt_dp, nt, gw, tx, ty)) return;
    } else {
        id -= ndp;
        z = id / sk.wsk;
        const int q = id - z * sk.wsk;
        c = q & 7;                                                  // = this workgroup's XCD (ndp and wsk are multiples of 8)
        r = q >> 3;
        it0 = R.begin(c, r);
        it1 = R.begin(c, r + 1);
    }
    const int j = c * R.w8 + r;
    float *ws = p.sk_ws + (size_t)z * sk.wsk * (2 * 128 * 128);     // two partial slots per band workgroup: [0] its run's first piece, [1] its last
    int *counters = p.sk_flags + z * R.Ts;                          // one arrival counter per band tile
    int hi = it1;
    while (hi > it0) {   // the pieces of [it0, it1), one per tile, last piece first
        int tbase = 0, tile = 0;
        if (!dp) {
            tile = (hi - 1) / sk.stages;
            tbase = tile * sk.stages;
            tx = sk.mt_dp + tile / nt;
            ty = tile % nt;
        }
        const int lo = it0 > tbase ? it0 : tbase;
        ring_tile_range<128, 128, 32, 64>(
            p, z, tx * 128, ty * 128, smem, lo - tbase, hi - tbase,
            [&](f32x16 (&acc)[1][2], const ConvGroup &g, const ConvTilePtrs &tp, int mw, int nw, int li, int lh) {
                if (lo > tbase || hi - tbase < sk.stages) {   // a piece of a split tile
                    sk_store_partial(acc, ws + (size_t)(2 * j + (it0 < tbase ? 1 : 0)) * (128 * 128));
                    __syncthreads();                          // every thread's stores are at the L2 (and every wave is done with the ring)
                    const int rf = R.run_of(c, tbase), rl = R.run_of(c, tbase + sk.stages - 1);
                    int *sh = reinterpret_cast<int *>(smem);
                    if (threadIdx.x == 0) sh[0] = __hip_atomic_fetch_add(counters + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __syncthreads();
                    const bool last = sh[0] == rl - rf;       // the other pieces are all there
                    __syncthreads();                          // (sh[0] is ring memory: read before the next piece's DMA)
                    if (!last) return;
                    if (threadIdx.x == 0) __hip_atomic_store(counters + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the next launch finds it clear
                    sk_load_partial<true>(acc, ws + (size_t)(2 * (c * R.w8 + rf) + (R.begin(c, rf) < tbase ? 1 : 0)) * (128 * 128));
                    for (int rr = rf + 1; rr <= rl; ++rr)     // k order, whoever arrived last
                        sk_load_partial<false>(acc, ws + (size_t)(2 * (c * R.w8 + rr) + (R.begin(c, rr) < tbase ? 1 : 0)) * (128 * 128));
                }
                conv_tile_epilogue<1, 2>(p, g, tp, acc, mw, nw, li, lh);
            });
        hi = lo;
        if (hi > it0) __syncthreads();   // the ring's two slots are reused by the next piece
    }
}

// The band's correctness rests on ONE hardware fact: workgroups of a 1-D grid go to the XCDs round-robin by id, so that band workgroups with
// equal id % 8 share an L2.  ts_ctx_create checks it once per device (a 2 048-workgroup probe reads HW_REG_XCC_ID); where it does not hold
// (another partition mode or part), no layer gets a stream-K plan: whole tiles only.
__global__ void xcc_probe_kernel(int *out) {
    if (threadIdx.x == 0) {
        unsigned v;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
        out[blockIdx.x] = (int)(v & 0xf);
    }
}
static int g_sk_map_ok[16];   // per device: 0 unknown, 1 ids of equal residue mod 8 share an XCD, -1 they do not
hipError_t conv_sk_probe_xcd_map(int device) {
    if (device < 0 || device >= 16 || g_sk_map_ok[device]) return hipSuccess;
    constexpr int N = 2048;
    int *d = nullptr, h[N];
    hipError_t e = hipMalloc(&d, N * sizeof(int));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(xcc_probe_kernel, dim3(N), dim3(64), 0, nullptr, d);
    e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return e;
    bool ok = true;
    for (int i = 8; i < N; ++i) ok = ok && h[i] == h[i & 7];
    g_sk_map_ok[device] = ok ? 1 : -1;
    return hipSuccess;
}
bool conv_sk_supported() {
    int dev = 0;
    return hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16 && g_sk_map_ok[dev] == 1;
}

bool conv_gemm_plan_sk(const ConvParams &p, ConvSK &sk) { return conv_sk_supported() && conv_gemm_plan_sk_shape(p, sk); }

// the plan as a function of the layer's shape alone (host-side tests call this one: no device needed)
bool conv_gemm_plan_sk_shape(const ConvParams &p, ConvSK &sk) {
    if (p.zdiv > 0 || !conv_gemm_ring_takes(p) || p.ngroups < 1 || 256 % p.ngroups) return false;
    const int MT = (p.M + 127) / 128, nt = (p.N + 127) / 128, unit = 256 / p.ngroups;
    const long per = (long)MT * nt;                    // tiles of a problem
    long full = per / unit * unit;                     // ... of which these make whole units of the chip
    if (full == 0 || full == per || unit < 8) return false;
    sk.wsk = unit;
    sk.stages = p.Ktot / 32;
    for (;;) {
        sk.mt_dp = (int)(full / nt);
        sk.mt_sk = MT - sk.mt_dp;
        sk.dp8 = 8 * ((sk.mt_dp * nt + 7) / 8);
        if (sk.mt_dp < 1 || sk.mt_sk < 1) return false;
        const long I = (long)sk.mt_sk * nt * sk.stages;
        if (I * (sk.wsk + 1) >= (1l << 31)) return false;        // the kernel cuts the runs in int arithmetic
        if (I / sk.wsk >= 4 && sk.mt_sk * nt >= 32) return true;   // (and at least 4 tiles per XCD: the band is dealt to the XCDs in whole tiles)
        // a few tiles over whole units (3 600 = 14 x 256 + 16): runs of under 4 stages would be all prologue — the band takes one more
        // unit of tiles instead (runs of a little over one tile each)
        if (full < 2 * unit) return false;
        full -= unit;
    }
}

hipError_t launch_conv_gemm_ring_sk(const ConvParams &p_in, const ConvSK &sk, hipStream_t stream) {
    ConvParams p = p_in;
    if (!p.zero) {
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) p.zero = skinny_zero_buffer(dev);
    }
    if (!p.zero || !conv_gemm_ring_takes(p) || p.zdiv > 0 || sk.mt_dp < 1 || sk.mt_sk < 1 || sk.wsk < 8 || (sk.wsk & 7) || sk.stages != p.Ktot / 32)
        return hipErrorInvalidValue;
    const int nt = (p.N + 127) / 128;
    if (conv_sk_workspace(stream, (size_t)sk.wsk * p.ngroups * 2 * 128 * 128, (size_t)sk.mt_sk * nt * p.ngroups, &p.sk_ws, &p.sk_flags) != 0)
        return hipErrorOutOfMemory;
    hipLaunchKernelGGL(conv_ring_sk_kernel, dim3((unsigned)((sk.dp8 + sk.wsk) * p.ngroups)), dim3(512), 0, stream, p, sk);
    return hipGetLastError();
}

hipError_t launch_conv_gemm_ring_banded(const ConvParams &p_in, const ConvBands &bd, hipStream_t stream) {
    ConvParams p = p_in;
    if (!p.zero) {
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) p.zero = skinny_zero_buffer(dev);
    }
    if (!p.zero || !conv_gemm_ring_takes(p) || p.zdiv > 0 || bd.mt_big < 1 || bd.mt_small < 1) return hipErrorInvalidValue;
    const int nt = (p.N + 127) / 128;
    const int big8 = 8 * ((bd.mt_big * nt + 7) / 8), small8 = 8 * ((bd.mt_small * nt + 7) / 8);
    hipLaunchKernelGGL(conv_ring_banded_kernel, dim3((unsigned)((big8 + small8) * p.ngroups)), dim3(512), 0, stream, p, bd, big8, small8);
    return hipGetLastError();
}

bool conv_gemm_ring_takes(const ConvParams &p) {
    if (p.g[0].nseg > 4 || p.Ktot > 60000 || p.Ktot < 32) return false;
    for (int z = 0; z < (p.zdiv > 0 ? 1 : p.ngroups); ++z)
        for (int i = 0; i < p.g[z].nseg; ++i)
            if (p.g[z].seg[i].len % 32) return false;   // a stage never straddles two taps / segments
    return true;
}

// Which tile plan for a layer?  A launch is rounds of 512 resident workgroups (two per CU); a workgroup alone on its CU runs at the full
// pipe rate (tools/ring_probe.py: 256 tiles on 256 CUs take what 512 take), so a last round that is at most half full costs half a
// round, a fuller one a whole round.  Three plans, cost = rounds x tile height / efficiency:
//   9: 128 x 128 tiles on 8 waves, the fastest per tile;
//   3: 96 x 128 tiles (4 waves of 96 x 32, ~5 % slower per flop) change the tile COUNT: the N = 768 layers of the wav2vec2 blocks (900 tiles
//      of 128 rows = 1.76 rounds -> 2; 1 200 of 96 rows = 2.34 -> 2.5 x 0.75 = 1.875): out-proj 207 -> 200 us, FFN2 750 -> 718 us in a face batch;
//   7: bands (conv_gemm.hip's plan, `bd`): 128 x 128 tiles for the whole rounds, 64 x 128 tiles (~7 % slower per flop) for the rows that
//      are left: FFN1 (3 600 tiles = 7.03 rounds -> 6.98 + half a round of short tiles: 715 -> 693 us) and the paired body + hand layers
//      (2 400 tiles = 4.69 rounds -> 4 + 1.5 short ones: conv stacks of a 256-clip pass 25.7 -> 25.2 ms); ties go to the plain plans.
//   8: whole tiles + stream-K band (`sk`, round 6): cost = the tile list's exact share of the chip + the band's hand-over.
// Measured: profiles/r05_notes/ring_tall_tiles.txt, face_layers_ab.txt, ring_banded_probe.txt; round 6: profiles/r06_notes/stream_k_*.txt.
// In the face pass (profiles/r06_notes/stream_k_in_situ.txt; band forced vs off, per layer): FFN2 (K = 3 072: 96 stages) -5.8 %, the feature
// convolutions of K = 1 536 -0.6 ... -1.6 %, K = 1 024 +1.4 %, QKV (K = 768, 2 700 tiles) -1.1 %, out-proj (K = 768, 900 tiles) +1.7 %, FFN1 +-0:
// what the band costs — two prologues instead of one, the partial's trip through the L2, the arrival — is a fixed number of stages, so its
// share falls with K.
constexpr double SK_OVERHEAD_STAGES = 4.0;    // ... in rounds of 512 tiles: 4 stages' worth, i.e. 4 / (stages per tile)
constexpr double SK_MARGIN = 0.97;            // the band must promise 3 % over the best whole-tile plan
int conv_gemm_ring_pick(const ConvParams &p, const ConvBands *bd, const ConvSK *sk) {
    const long nt = (long)((p.N + 127) / 128) * p.ngroups;
    auto rounds = [](long tiles) {
        const long full = tiles / 512, rest = tiles - full * 512;
        return (double)full + (rest == 0 ? 0.0 : (rest <= 256 ? 0.5 : 1.0));
    };
    const double c128 = rounds((long)((p.M + 127) / 128) * nt) * 128.0;
    const double c96 = rounds((long)((p.M + 95) / 96) * nt) * 96.0 / 0.95;   // layers without a tail: 122.7 vs 130.8, 114 vs 123 TFLOP/s
    double best = c128;
    int pick = 9;
    if (c96 < best) {
        best = c96;
        pick = 3;
    }
    if (bd) {   // the big band is whole rounds but for a few tiles, whose slots the short tiles take
        const double cb = bd->first_small / 512.0 * 128.0 + rounds((long)bd->mt_small * nt) * 64.0 / 0.93;
        if (cb < 0.98 * best) {   // within 2 % the plain plans measure as fast or faster (feature convolutions 5 / 6: 537 vs 532, 277 vs 271 us)
            best = cb;
            pick = 7;
        }
    }
    if (sk) {   // 8: whole tiles for the whole units of 256 + a stream-K band: the chip's share of the tile list, plus what the band's hand-over costs
        const double cs = ((double)((p.M + 127) / 128) * nt / 512.0 + SK_OVERHEAD_STAGES / sk->stages) * 128.0;
        if (cs < SK_MARGIN * best) pick = 8;
    }
    return pick;
}

hipError_t launch_conv_gemm_ring(const ConvParams &p_in, int variant, hipStream_t stream) {
    ConvParams p = p_in;
    if (!p.zero) {
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) p.zero = skinny_zero_buffer(dev);
    }
    if (!p.zero || !conv_gemm_ring_takes(p)) return hipErrorInvalidValue;
    if (variant == 0) variant = conv_gemm_ring_pick(p, nullptr, nullptr);
    if (variant == 10) variant = conv_gemm_ring_pick(p, nullptr, nullptr) == 3 ? 6 : 5;   // the pick (without bands), tiles dealt to the XCDs
    const dim3 grid((p.M + 127) / 128, (p.N + 127) / 128, p.ngroups);
    auto dealt = [&](int bm) { return dim3(8 * (unsigned)(((long)((p.M + bm - 1) / bm) * grid.y + 7) / 8), 1, grid.z); };
    switch (variant) {
        case 1: hipLaunchKernelGGL((conv_ring_kernel<128, 128, 64, 64, false>), grid, dim3(256), 0, stream, p); break;   // 4 waves of 64 x 64
        case 9: hipLaunchKernelGGL((conv_ring_kernel<128, 128, 32, 64, false>), grid, dim3(512), 0, stream, p); break;   // 8 waves of 32 x 64
        case 3: hipLaunchKernelGGL((conv_ring_kernel<96, 128, 96, 32, false>), dim3((p.M + 95) / 96, grid.y, grid.z), dim3(256), 0, stream, p); break;   // 96 x 128: 4 waves of 96 x 32
        case 5: hipLaunchKernelGGL((conv_ring_kernel<128, 128, 32, 64, true>), dealt(128), dim3(512), 0, stream, p); break;   // 9 with the tiles dealt to the XCDs
        case 6: hipLaunchKernelGGL((conv_ring_kernel<96, 128, 96, 32, true>), dealt(96), dim3(256), 0, stream, p); break;     // 3 likewise
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace ts
