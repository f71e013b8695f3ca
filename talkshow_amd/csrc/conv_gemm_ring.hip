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
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void ring_tile(const ConvParams &p, const int zidx, const int m0, const int n0, float *smem) {
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
    for (int j = 0; j < NB; ++j) {
        const int n = n0 + (wave * NB + j) * RPB + drow;
        pb[j] = (n < w_rows ? gw + (long)n * ldw : p.zero) + dcol(wave * NB + j);
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

    const int T = p.Ktot / BK;   // stages
    // ---- prologue: stage 0 goes out and lands, its first fragments are read ----
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

    conv_tile_epilogue<TM, TN>(p, g, tp, acc, m0 + wm * WM, n0 + wn * WN, li, lh);   // conv_tile.h
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
// Measured: profiles/r05_notes/ring_tall_tiles.txt, face_layers_ab.txt, ring_banded_probe.txt.
int conv_gemm_ring_pick(const ConvParams &p, const ConvBands *bd) {
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
        if (cb < 0.98 * best) pick = 7;   // within 2 % the plain plans measure as fast or faster (feature convolutions 5 / 6: 537 vs 532, 277 vs 271 us)
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
    if (variant == 0) variant = conv_gemm_ring_pick(p, nullptr);
    if (variant == 10) variant = conv_gemm_ring_pick(p, nullptr) == 3 ? 6 : 5;   // the pick (without bands), tiles dealt to the XCDs
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
