// conv_gemm_f32, LDS-DMA ring engine — the same implicit-GEMM convolution as conv_gemm.hip (same ConvParams, same segments /
// taps / strides, same MFMA, same k order: BIT-IDENTICAL outputs), with the operand path rebuilt around direct global -> LDS
// loads (global_load_lds_dwordx4) instead of global -> VGPR -> ds_write.
//
// Replaces the same PyTorch ops as conv_gemm.hip; the shapes it is built for are the plain-GEMM layers of the face generator —
// the wav2vec2 encoder blocks' QKV / out-proj / FFN1 / FFN2 (reference: nets/spg/wav2vec.py:76-143, HF Wav2Vec2EncoderLayer) —
// and the k3 conv stacks of nets/spg/vqvae_modules.py:87-212.
//
// Mapping to CDNA4:
//   * a stage = BK (16 or 32) consecutive k of the tile's BM activation rows and BN weight rows, row-major in LDS with a row
//     pitch of BK floats — the image a wave's LDS-DMA instruction writes (wave-uniform base + lane x 16 B = 1 KB = 8 rows x
//     128 B or 16 rows x 64 B): both operands stay ROW-MAJOR in HBM (no layout change anywhere else), a DMA instruction reads
//     whole 128-byte (64-byte) row pieces;
//   * conflict-free fragment reads without padding: the 16-byte segment s of row r sits at position s ^ f(r) of its row
//     (f = (r >> 1) & 7 for 128-byte rows, (r >> 2) & 3 for 64-byte rows) — the permutation is applied to the per-lane SOURCE
//     address of the DMA and to the ds_read_b128 address, never to the destination (which is lane-linear by construction);
//     a 16-lane group of a ds_read_b128 then covers all 64 banks exactly once;
//   * ring of NS stage slots; stage t + NS - 1 is issued behind the barrier that opens stage t; the barrier that opens stage
//     t + 1 sits in the middle of stage t's last MFMA group behind a COUNTED vmcnt (loads of later stages stay in flight across
//     it) — one barrier per stage, never a drained queue, no staging registers, no ds_write;
//   * 4 (or 8) waves per workgroup, each a WM x WN block of 32 x 32 MFMA tiles; 64 KB (BK = 32, NS = 2) .. 32 KB
//     (BK = 16, NS = 2) of LDS and <= 128 .. 168 VGPRs: 2 .. 4 workgroups per CU, whose barriers are independent — a SIMD's
//     matrix pipe is fed by the other workgroups' waves while one waits.
#include <type_traits>

#include "conv_tile.h"

namespace ts {

__device__ __forceinline__ void ring_glds16(const float *src, float *lds_dst) {   // lds_dst: wave-uniform; lane i lands at + 16 i bytes
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds_dst, 16, 0, 0);
}

template <int N> __device__ __forceinline__ void ring_wait_vm() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }

struct RingTile {
    int z, m0, n0;   // problem / group, first output row, first output column
};
// the tiles one workgroup works through, in order
struct RingSeqOne {   // plain grid: one tile per workgroup
    static constexpr bool single = true;
    RingTile t;
    __device__ __forceinline__ int count() const { return 1; }
    __device__ __forceinline__ RingTile tile(int) const { return t; }
};
struct RingSeqStrided {   // persistent grid: tiles first, first + step, ... of an MT x NT x groups tile space (row tiles fastest)
    static constexpr bool single = false;
    int first, step, total, MT, NT, bm, bn;
    __device__ __forceinline__ int count() const { return first < total ? (total - first + step - 1) / step : 0; }
    __device__ __forceinline__ RingTile tile(int i) const {
        const int idx = first + i * step;
        const int r = idx / MT, mt = idx - r * MT;
        const int z = r / NT, nt = r - z * NT;
        return RingTile{z, mt * bm, nt * bn};
    }
};

// The BM x BN output tiles of `seq`, one after the other, as ONE software pipeline: the ring keeps turning across tile boundaries
// (the first stages of tile i + 1 are issued behind the last stages of tile i, its first fragments are read behind the last barrier
// of tile i), so that between the last MFMA of a tile and the first MFMA of the next there is only the epilogue itself — no
// workgroup dispatch, no descriptor fetch, no cold operand round trip.  smem: NS * (BM + BN) * BK floats of LDS (ONE object).
template <int BM, int BN, int WM, int WN, int BK, int NS, class Seq>
__device__ __forceinline__ void ring_tiles(const ConvParams &p, const Seq &seq, float *smem) {
    static_assert(BK == 16 || BK == 32, "stage depth");
    static_assert(NS >= 2 && NS <= 4, "ring slots");
    constexpr int SEGS = BK / 4;               // 16-byte segments of a row per stage
    constexpr int LSEG = BK == 32 ? 3 : 2;
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
    static_assert(NW * 32 * WN <= NS * STAGE, "the epilogue's slabs fit in the ring");

    const int ntiles = Seq::single ? 1 : seq.count();   // workgroup-uniform
    if (ntiles <= 0) return;
    const long ldw = p.ldw > 0 ? p.ldw : p.Ktot;
    const int w_rows = p.w_rows > 0 ? p.w_rows : 0x7fffffff;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    auto swz = [](int row) { return BK == 32 ? (row >> 1) & 7 : (row >> 2) & 3; };

    // ---- loader: DMA instruction j of this wave covers rows (wave * NA + j) * RPB .. + RPB - 1 of the A part (same for B);
    // lane -> row lane / SEGS of the block, LDS position lane % SEGS of that row, i.e. global segment position ^ f(row).
    // The loader's cursor (tile, segment, tap, chunk) runs NS - 1 stages ahead of the MFMAs, into the next tile when this one ends ----
    const int drow = lane >> LSEG, dpos = lane & (SEGS - 1);
    // column (floats) of this lane's segment in DMA instruction j: position ^ f(row in tile); blocks of 8 rows alternate the top bit of f
    auto dcol = [&](int blk) { return (dpos ^ swz(blk * RPB + drow)) << 2; };
    int a_row[NA], a_t[NA];   // (b * Lin) input row base or -1 if the output row is out of range; t * stride
    // segment descriptors live in VGPR lanes (v_readlane): no scalar loads competing with LDS for lgkmcnt
    int vd = 0, vc0 = 0, vlen = BK, vnt = 1, desc_z = -1;
    int s = 0, tap = 0, cc = 0, cur_len = BK, cur_nt = 1;
    const float *gx = nullptr;
    const float *pa[NA], *pb[NB];
    auto enter_run = [&]() {   // operand pointers of the first stage of (segment s, tap)
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
    auto setup = [&](const RingTile &t) {   // cursor -> stage 0 of tile t
        const int zi = p.zdiv > 0 ? 0 : t.z;
        const ConvGroup &g = p.g[zi];
        const ConvTilePtrs tp = conv_tile_ptrs(p, g, t.z);
        gx = tp.x;
        if (desc_z != zi) {   // workgroup-uniform
            desc_z = zi;
            vd = 0, vc0 = 0, vlen = BK, vnt = 1;
            if (lane < 4) {
                vd = g.seg[lane].d;
                vc0 = g.seg[lane].c0;
                vlen = g.seg[lane].len;
                vnt = g.seg[lane].ntap > 1 ? g.seg[lane].ntap : 1;
            }
        }
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int m = t.m0 + (wave * NA + j) * RPB + drow;
            if (m < p.M) {
                const int b = m / p.Lout, tt = m - b * p.Lout;
                a_row[j] = b * p.Lin;
                a_t[j] = tt * p.stride;
            } else {
                a_row[j] = -1;
                a_t[j] = 0;
            }
        }
        s = 0, tap = 0, cc = 0;
        enter_run();
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int n = t.n0 + (wave * NB + j) * RPB + drow;
            pb[j] = (n < w_rows ? tp.w + (long)n * ldw : p.zero) + dcol(wave * NB + j);
        }
    };
    auto advance = [&]() {   // cursor -> the next stage of the K walk
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
    auto dma_one = [&](int slot, int o) {   // DMA instruction o of the stage the cursor stands at, into ring slot `slot`
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
    auto mfma_one = [&](int set, int k) {   // k-th MFMA of a group: e-major, the same order as conv_gemm.hip's mfma_q
        const int e = k / (TM * TN), ij = k % (TM * TN), i = ij / TN, j = ij % TN;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[set][j][e], fa[set][i][e], acc[i][j], 0, 0, 0);
    };

    const int T = p.Ktot / BK;   // stages per tile (a launch with more than one tile per workgroup has T >= NS: host-checked)
    // this wave's DMA loads of a stage have landed once at most ND * (stages issued after it) of its loads are outstanding
    // (other memory operations in the queue — the epilogue's — can only make the wait stricter)
    auto wait_later = [&](int later) {
        if (NS == 2 || later <= 0) ring_wait_vm<0>();
        else if (NS == 3 || later == 1) ring_wait_vm<ND>();
        else ring_wait_vm<2 * ND>();
    };

    RingTile cur = seq.tile(0), nxt = cur;
    // ---- prologue: stages 0 .. NS - 2 of the first tile go out, stage 0 lands, its first fragments are read ----
    setup(cur);
#pragma unroll
    for (int st = 0; st < NS - 1; ++st) {
        if (st < T) {
            if (st > 0) advance();
#pragma unroll
            for (int o = 0; o < ND; ++o) dma_one(st, o);
        }
    }
    wait_later((T < NS - 1 ? T : NS - 1) - 1);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int f = 0; f < NF; ++f) read_one(0, 0, 0, f);

    // ---- one stage out of ring slot `slot`.  Hand ordered (sched_barrier after every step), at most one or two side operations
    // behind each MFMA (64 cycles of pipe): a DMA issue, or one fragment read — never a run of them with a single MFMA in flight.
    // REFILL: the stage NS - 1 ahead exists (in this tile, or — newtile: its stage 0 — in the next) and goes into the slot the
    // previous stage was read from; MORE: the next stage exists (the next tile's first one included) ----
    auto stage = [&](auto Rc, auto Mc, const int slot, const int later, const bool newtile) {
        constexpr bool REFILL = decltype(Rc)::value, MORE = decltype(Mc)::value;
        const int slot_fill = slot == 0 ? NS - 1 : slot - 1, slot_next = slot == NS - 1 ? 0 : slot + 1;
        if (REFILL) {
            if (!Seq::single && newtile) setup(nxt);
            else advance();
        }
        __builtin_amdgcn_sched_barrier(0);
        constexpr int OPS0 = ND + (NQ > 1 ? NF : 0), PER0 = (OPS0 + MF - 1) / MF;
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
                            if (REFILL) dma_one(slot_fill, o);
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
        // last group: the barrier that opens the next stage half way, then that stage's first fragments
#pragma unroll
        for (int k = 0; k < H; ++k) mfma_one((NQ - 1) & 1, k);
        __builtin_amdgcn_sched_barrier(0);
        if (MORE) {
            wait_later(later);                 // ... and every LDS read of this stage is done (lgkmcnt)
            __builtin_amdgcn_s_barrier();      // everybody's loads of the next stage have landed; everybody holds this stage's last fragments
            asm volatile("" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = H; k < MF; ++k) {
            mfma_one((NQ - 1) & 1, k);
            if (MORE && k - H < NF) {
                __builtin_amdgcn_sched_barrier(0);
                read_one(slot_next, 0, 0, k - H);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    using Yes = std::true_type;
    using No = std::false_type;
    auto zero_acc = [&]() {
#pragma unroll
        for (int ii = 0; ii < TM; ++ii)
#pragma unroll
            for (int jj = 0; jj < TN; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ii][jj][r] = 0.f;
    };
    int slot = 0;
    auto turn = [&]() { slot = slot == NS - 1 ? 0 : slot + 1; };
    for (int i = 0; i < ntiles; ++i) {
        const bool last_tile = i + 1 == ntiles;
        if (!last_tile) nxt = seq.tile(i + 1);
        zero_acc();
        // the hot loop: every stage whose refill — the stage NS - 1 ahead, in this tile or (newtile: its first one) in the next —
        // exists; NS - 2 later stages stay in flight across its barrier.  ONE instance of the stage body.
        const int nhot = last_tile ? T - (NS - 1) : T;
        int t = 0;
        for (; t < nhot; ++t) {
            stage(Yes{}, Yes{}, slot, NS - 2, t + NS - 1 == T);
            turn();
        }
        if (last_tile) {   // nothing left to issue: T - 2 - t later stages in flight
            for (; t + 1 < T; ++t) {
                stage(No{}, Yes{}, slot, T - 2 - t, false);
                turn();
            }
            if (t < T) stage(No{}, No{}, slot, 0, false);
        }
        // ---- the tile is complete: its epilogue (the next tile's first stages are landing meanwhile) ----
        const ConvGroup &g = p.g[p.zdiv > 0 ? 0 : cur.z];
        const ConvTilePtrs tp = conv_tile_ptrs(p, g, cur.z);
        if (last_tile && !p.epi_regs && conv_tile_staged_ok(p, g, tp)) {   // workgroup-uniform; the ring is idle: it becomes the slabs
            ring_wait_vm<0>();                 // this wave's last fragment reads are done ...
            __builtin_amdgcn_s_barrier();      // ... and everybody's
            asm volatile("" ::: "memory");
            conv_tile_epilogue_staged<TM, TN>(p, g, tp, acc, cur.m0 + wm * WM, cur.n0 + wn * WN, lane, smem + wave * (32 * WN));
        } else {
            conv_tile_epilogue<TM, TN>(p, g, tp, acc, cur.m0 + wm * WM, cur.n0 + wn * WN, li, lh);   // conv_tile.h
        }
        cur = nxt;
    }
}

// OCC: workgroups per CU the register budget is set for.
// STAGGER (OCC = 2): a launch starts with 512 workgroups at the same instant, two per CU (workgroups b and b + 256 share a CU:
// tools/cu_map_probe.cpp), and equal tiles keep them in lock-step for the whole launch: all 512 reach their epilogue together, the chip
// stores 32 MB at once (~7-8 us in which no matrix pipe has work: the per-round fixed time of tools/ring_probe.py, independent of K),
// then all 512 successors fetch their first operands together.  A phase offset between the two workgroups of a CU, once there,
// persists (whichever is alone runs at the full pipe rate — a single wave per SIMD saturates it — so it does not catch up or fall
// back).  So the second resident of each CU sleeps for half a tile's worth of pipe time before its first tile: its partner computes
// alone meanwhile (nothing is lost), and from then on one of the two always has MFMAs queued while the other stores and restarts.
template <int BM, int BN, int WM, int WN, int BK, int NS, int OCC, bool STAGGER = false>
__global__ __launch_bounds__(64 * (BM / WM) * (BN / WN), OCC * (BM / WM) * (BN / WN) / 4) void conv_ring_kernel(const ConvParams p) {
    __shared__ __attribute__((aligned(1024))) float smem[NS * (BM + BN) * BK];
    if (STAGGER) {
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (lin - 256u < 256u) {   // workgroup-uniform
            // half a tile alone on the CU: (Ktot / 2) k x BM x BN x 2 flops at 256 flops per clock
            const long long target = (long long)(p.Ktot / 2) * (BM * BN * 2 / 256), t0 = clock64();
            while (clock64() - t0 < target) __builtin_amdgcn_s_sleep(32);
        }
    }
    ring_tiles<BM, BN, WM, WN, BK, NS>(p, RingSeqOne{RingTile{(int)blockIdx.z, (int)blockIdx.x * BM, (int)blockIdx.y * BN}}, smem);
}

// persistent form: gridDim.x workgroups (OCC per CU) deal the tiles out round-robin and each runs its share as one pipeline
template <int BM, int BN, int WM, int WN, int BK, int NS, int OCC>
__global__ __launch_bounds__(64 * (BM / WM) * (BN / WN), OCC * (BM / WM) * (BN / WN) / 4) void conv_ring_persistent_kernel(const ConvParams p) {
    __shared__ __attribute__((aligned(1024))) float smem[NS * (BM + BN) * BK];
    const int MT = (p.M + BM - 1) / BM, NT = (p.N + BN - 1) / BN;
    ring_tiles<BM, BN, WM, WN, BK, NS>(p, RingSeqStrided{(int)blockIdx.x, (int)gridDim.x, MT * NT * p.ngroups, MT, NT, BM, BN}, smem);
}

hipError_t conv_ring_trace_set(unsigned long long *, int) { return hipSuccess; }   // (the trace build lives in git history: commit "conv_gemm_f32: LDS-DMA ring engine")

hipError_t launch_conv_gemm_ring(const ConvParams &p_in, int variant, hipStream_t stream) {
    ConvParams p = p_in;
    if (!p.zero) {
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) p.zero = skinny_zero_buffer(dev);
    }
    if (!p.zero || p.g[0].nseg > 4 || p.Ktot > 60000) return hipErrorInvalidValue;
    for (int z = 0; z < (p.zdiv > 0 ? 1 : p.ngroups); ++z)
        for (int i = 0; i < p.g[z].nseg; ++i)
            if (p.g[z].seg[i].len % 32) return hipErrorInvalidValue;
    if (!p_in.epi_regs) p.epi_regs = knobs().conv_staged ? 0 : 1;
    dim3 block(256);
    auto grid = [&](int bm, int bn) { return dim3((p.M + bm - 1) / bm, (p.N + bn - 1) / bn, p.ngroups); };
    switch (variant) {
        case 1: hipLaunchKernelGGL((conv_ring_kernel<128, 128, 64, 64, 32, 2, 2>), grid(128, 128), block, 0, stream, p); break;
        case 2: hipLaunchKernelGGL((conv_ring_kernel<128, 128, 64, 64, 16, 2, 4>), grid(128, 128), block, 0, stream, p); break;
        case 3: hipLaunchKernelGGL((conv_ring_kernel<128, 128, 64, 64, 16, 3, 3>), grid(128, 128), block, 0, stream, p); break;
        case 4: hipLaunchKernelGGL((conv_ring_kernel<128, 128, 64, 64, 16, 4, 2>), grid(128, 128), block, 0, stream, p); break;
        case 5: hipLaunchKernelGGL((conv_ring_kernel<128, 128, 64, 64, 16, 2, 3>), grid(128, 128), block, 0, stream, p); break;
        case 6: hipLaunchKernelGGL((conv_ring_kernel<64, 128, 32, 64, 32, 2, 3>), grid(64, 128), block, 0, stream, p); break;
        case 7: hipLaunchKernelGGL((conv_ring_kernel<64, 64, 32, 32, 32, 2, 4>), grid(64, 64), block, 0, stream, p); break;
        // 8 waves per 128 x 128 tile (64 x 32 each): two waves per SIMD even when a workgroup is alone on its CU (the tail of a launch)
        case 8: hipLaunchKernelGGL((conv_ring_kernel<128, 128, 64, 32, 32, 2, 2>), grid(128, 128), dim3(512), 0, stream, p); break;
        case 9: hipLaunchKernelGGL((conv_ring_kernel<128, 128, 32, 64, 32, 2, 2>), grid(128, 128), dim3(512), 0, stream, p); break;
        case 14: hipLaunchKernelGGL((conv_ring_kernel<128, 128, 64, 64, 32, 2, 2, true>), grid(128, 128), block, 0, stream, p); break;          // 1 + stagger
        case 15: hipLaunchKernelGGL((conv_ring_kernel<128, 128, 32, 64, 32, 2, 2, true>), grid(128, 128), dim3(512), 0, stream, p); break;      // 9 + stagger
        // persistent: 2 workgroups per CU, every workgroup a pipeline over its tiles
        case 12: if (p.Ktot < 64) return hipErrorInvalidValue;
                 hipLaunchKernelGGL((conv_ring_persistent_kernel<128, 128, 64, 64, 32, 2, 2>), dim3(512), block, 0, stream, p); break;
        case 13: if (p.Ktot < 64) return hipErrorInvalidValue;
                 hipLaunchKernelGGL((conv_ring_persistent_kernel<128, 128, 32, 64, 32, 2, 2>), dim3(512), dim3(512), 0, stream, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace ts
