// skinny16_wide_kernel — the chain stage for coalesced passes (M >= 64 clips per stage, the 256-clip operating point).
//
// The split-K kernels of skinny_gemm.hip spread ONE small stage over many CUs: a workgroup owns 16..64 x 16..32 outputs,
// its 8 waves split K, every wave pulls its own operand fragments from L2 and the partial tiles meet in LDS.  At 256
// clips a stage is no longer latency bound but bound by what a CU can pull through its L1 (192 KB of fragments per
// 64 x 32 tile, two tiles per CU) and by the phases in which the matrix pipe idles (8-wave reduction, stragglers).
// Here a workgroup owns a 64 x 64 output tile over the FULL K:
//   * operands are staged through LDS by direct global->LDS loads (global_load_lds_dwordx4): both operands of the
//     chain are stored as per-fragment contiguous KBs in lane order (kernels.h, "tiled"), which is exactly the image an
//     LDS-DMA writes (wave-uniform base + lane * 16) and the image ds_read_b128 reads conflict-free.  8 KB per q-step
//     (16 k) feed 64 MFMAs: 128 B per MFMA instead of 384 (64 x 32 split-K tile) through the CU's L1;
//   * no cross-wave sum: wave (wr, wc) owns row block wr x column blocks {2 wc, 2 wc + 1} and walks all of K.  To stay
//     BIT-IDENTICAL with the split-K kernels the walk keeps their summation structure: K is cut into the same slices
//     (the former waves' shares), each slice is an MFMA chain from zero, slices are added in index order;
//   * the MFMA operands are swapped (weights as A, activations as B: same products, same k order, same bits) so that a
//     lane holds 4 CONSECUTIVE output channels of one clip: every epilogue access is a 16-byte vector;
//   * a 4-slot ring of 32 KB stages (4 q-steps) with counted vmcnt waits: up to 96 KB per CU in flight across the
//     one barrier per stage;
//   * problems with half the K of the launch's biggest get two tiles per workgroup (SD_ITEMS), so that every workgroup
//     of a launch carries about the same work and a launch is about one workgroup per CU.
// Reference arithmetic: nets/spg/gated_pixelcnn_v2.py:61-87,120-124 (one stage of GatedMaskedConv2d / the logits head).
#include <cstdlib>

#include "skinny_desc.h"

namespace ts {

constexpr int WIDE_NS = 4;   // ring slots
constexpr int WIDE_QS = 4;   // q-steps (16 k each) per stage
template <int V> struct IC { static constexpr int value = V; };

__device__ __forceinline__ void glds16(gcf *src, f32x4 *dst) {   // dst: wave-uniform; lane i lands at dst + i
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
}

// TRACE (tools/wide_trace.py): wave 0 of every workgroup stamps the 100 MHz wall clock into the launch's record block
// (pointer in start[6..7]): [0] entry [1] pointers ready [2] loop done [3] epilogue operands back [4] end [5] meta
// [6 + t] barrier of stage t passed (first tile, t < 8) [14] prologue issued
template <bool TRACE>
__global__ __launch_bounds__(512, 2) void skinny16_wide_kernel(const SkinnyDescBatch batch) {
    // ring[slot][q-step][fragment][lane]: fragments 0..3 = the tile's activation row blocks, 4..7 = its weight column blocks
    __shared__ f32x4 lds_all[WIDE_NS * WIDE_QS * 8 * 64 + (TRACE ? 8 : 0)];   // 128 KB: one workgroup per CU (ONE object: a second
                                                                               // one makes hipcc drain vmcnt before every ds_read)
    f32x4(*ring)[WIDE_QS][8][64] = reinterpret_cast<f32x4(*)[WIDE_QS][8][64]>(lds_all);
    unsigned long long *stamps = reinterpret_cast<unsigned long long *>(lds_all + WIDE_NS * WIDE_QS * 8 * 64);   // TRACE: 16 stamps
    const int tid = threadIdx.x;
    unsigned long long clk0 = 0;
#define TS_STAMP(k) do { if (TRACE && tid == 0) stamps[k] = wall_clock64(); } while (0)
    if (TRACE) {
        if (tid < 16) stamps[tid] = 0;
        TS_STAMP(0);
        clk0 = clock64();
    }
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;

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
    auto sdiv = [](int x, int d) { return (d & (d - 1)) == 0 ? x >> __builtin_ctz(d) : x / d; };
    const int M = I(SD_M), flags = I(SD_FLAGS), gateD = I(SD_GATED), cnt = I(SD_CNT), Q = I(SD_WTQ), items = I(SD_ITEMS);
    const int nmt = (M + 63) >> 6, ntile = (I(SD_N) >> 6) * nmt;
    const int T = Q >> 2;   // stages
    const bool pair_order = (batch.start[0] & 8) && nmt == 4;
    const bool gate = flags & SDF_GATE;
    const int wr = wave >> 1, wc = wave & 1;

    // ---- loader role: waves 0..3 stream activation row block `wave`, waves 4..7 weight column block `wave - 4` ----
    gcf *sp0 = nullptr, *sp1 = nullptr, *sp2 = nullptr;
    int st0 = 0, st1 = 0, st2 = 0, l0 = 0, l1 = 0, nsegw = 1;
    auto setup_loader = [&](int tile, int mt) {
        if (wave >= 4) {
            sp0 = P(SD_W) + ((((long)(tile * 4 + wave - 4) * Q) << 6) + lane) * 4;
            st0 = 256;
            l0 = Q;
        } else {
            nsegw = I(SD_NSEG);
            const int m = mt * 64 + wave * 16 + li;
            const int mc = m < M ? m : M - 1;   // clamped rows: computed, never stored
            const int nblk = (M + 15) >> 4;
            int blk = mt * 4 + wave;
            blk = blk < nblk ? blk : nblk - 1;
            auto seg_ptr = [&](int k, int &step) -> gcf * {
                gcf *base = P(k);
                gci *gidx = (gci *)P(k + 2);
                const int segw = I(k + 6);
                if (gidx) {   // token-embedding gather: the row is this lane's, the LDS image is the same fragment
                    const int g = gidx[(long)mc * I(k + 5)];
                    step = 16;
                    return (g >= 0 ? base + (long)g * I(k + 4) : P(SD_ZERO)) + lg * 4;
                }
                if (segw & SEG_TILED) {
                    step = 256;
                    return base + ((((long)blk * (segw & 0xffff)) << 6) + lane) * 4;
                }
                step = 16;
                return base + (long)mc * I(k + 4) + lg * 4;
            };
            sp0 = seg_ptr(SD_SEG, st0);
            l0 = I(SD_SEG + 7);
            if (nsegw > 1) {
                sp1 = seg_ptr(SD_SEG + SD_SEG_WORDS, st1);
                l1 = I(SD_SEG + SD_SEG_WORDS + 7);
            }
            if (nsegw > 2) sp2 = seg_ptr(SD_SEG + 2 * SD_SEG_WORDS, st2);
        }
    };
    bool prefetched = false;   // the tile's first two stages were issued behind the previous tile's last MFMAs

    for (int it = 0; it < items; ++it) {
        const int idx = (bx - first) * items + it;
        if (idx >= ntile) break;
        // tiles of a problem: with four clip blocks (256 clips) as (clip-block pair, column tile, block of the pair) — consecutive workgroups
        // = consecutive XCDs, so a weight tile lives on two XCDs and an activation block on four: 2 W + 4 A bytes over the fabric where the
        // column-major order (the clip blocks of a column tile on four XCDs: 4 W + 2 A = the 41.5 MB per launch of the round-4 counters)
        // moves a quarter more; one chain of 256 clips 31.6 -> 31.1 ms, three in flight unchanged.  TS_SKINNY_WIDE_PAIR=0: column-major
        auto tile_of = [&](int i, int &tl, int &m_) {
            if (pair_order) {
                const int half = I(SD_N) >> 5, h = i >= half ? 1 : 0, r = i - (h ? half : 0);   // 2 x column tiles per half
                tl = r >> 1;
                m_ = 2 * h + (r & 1);
            } else {
                tl = sdiv(i, nmt);
                m_ = i - tl * nmt;
            }
        };
        int tile, mt;
        tile_of(idx, tile, mt);
        const bool abl_noload = TRACE && (batch.start[0] & 2), abl_nomfma = TRACE && (batch.start[0] & 4);
        auto issue = [&](int t) {   // stage t = q-steps 4 t .. 4 t + 3, inside one segment (host-checked)
            if (abl_noload) return;
            const int q0 = t << 2;
            gcf *p = sp0 + (long)q0 * st0;
            int step = st0;
            if (nsegw > 1 && q0 >= l0) {
                p = sp1 + (long)(q0 - l0) * st1;
                step = st1;
                if (nsegw > 2 && q0 >= l0 + l1) {
                    p = sp2 + (long)(q0 - l0 - l1) * st2;
                    step = st2;
                }
            }
            f32x4 *dst = &ring[t & (WIDE_NS - 1)][0][wave][0];
#pragma unroll
            for (int qq = 0; qq < WIDE_QS; ++qq) glds16(p + (long)qq * step, dst + qq * 8 * 64);
        };

        // ---- the first two stages go out before anything else is computed: they are what the first MFMA waits for.
        // (Later stages follow behind the first barrier: sixteen 1 KB loads per wave up front take 1.3 us to issue — the memory
        // pipeline pushes back — and stage 0 queues behind them; loads stream twice as fast as the MFMAs consume them.)
        if (!prefetched) {
            setup_loader(tile, mt);
            if (it == 0) TS_STAMP(1);
            if (T > 0) issue(0);
            if (T > 1) issue(1);
            if (it == 0) TS_STAMP(14);
        }
        prefetched = false;

        // ---- epilogue operands of this wave's two 16 x 16 blocks: activation row m, 4 consecutive channels from n0.
        // Fetched behind the first barrier, ahead of the refill of stages 2..4 (a gate's terms were written by the previous
        // launch: ahead of the first wait they put a cold round trip in front of the first MFMA).  The counted waits count
        // LDS-DMA loads only; a plain load sitting at a later position in the queue can make them stricter than needed,
        // never weaker (a wait for "at most N outstanding" with N = the younger LDS-DMA loads).
        const int mrow = mt * 64 + wr * 16 + li;
        const bool m_ok = mrow < M;
        const int mcl = m_ok ? mrow : 0;
        int n0[2], oc[2];   // first channel in the weight-row numbering / in the (gated) output numbering
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int t16 = tile * 4 + 2 * wc + cb;
            if (gate) {   // a gate tile = 8 "tanh" channels followed by their 8 "sigmoid" partners
                const int tpg = gateD >> 3;
                const int group = sdiv(t16, tpg), ch0 = (t16 - group * tpg) << 3;
                n0[cb] = group * 2 * gateD + (lg >> 1) * gateD + ch0 + (lg & 1) * 4;
                oc[cb] = group * gateD + ch0 + (lg & 1) * 4;
            } else {
                n0[cb] = t16 * 16 + lg * 4;
                oc[cb] = n0[cb];
            }
        }
        const int add1_tw = I(SD_ADD1_TW), out_tw = I(SD_OUT_TW), pre_tw = I(SD_PRE_TW);
        f32x4 t0[2], t1[2], t2[2], t3[2], ecls[2];
        auto epi_load = [&]() {
            const int cls_ld = I(SD_CLS_LD);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                t0[cb] = *reinterpret_cast<gcf4 *>(P(SD_BIAS) + n0[cb]);
                const long i1 = (long)(mcl >> I(SD_ADD1_SHIFT)) * I(SD_ADD1_STRIDE) + n0[cb];
                t1[cb] = *reinterpret_cast<gcf4 *>(P(SD_ADD1) + (add1_tw ? tiled_index(i1, add1_tw) : i1));
                t2[cb] = *reinterpret_cast<gcf4 *>(P(SD_ADD2) + (long)(mcl >> I(SD_ADD2_SHIFT)) * I(SD_ADD2_STRIDE) + n0[cb]);
                t3[cb] = *reinterpret_cast<gcf4 *>(P(SD_ADD3) + (long)mcl * I(SD_ADD3_STRIDE) + n0[cb]);
                const int ccol = (cls_ld & (cls_ld - 1)) == 0 ? (n0[cb] & (cls_ld - 1)) : n0[cb] % cls_ld;
                ecls[cb] = *reinterpret_cast<gcf4 *>(P(SD_CLS) + (long)mcl * cls_ld + ccol);
            }
        };

        // ---- main loop.  A stage's fragments sit in registers (two sets, alternating); the barrier that opens stage t+1
        // falls between the MFMAs of q-steps 1 and 2 of stage t.  The two waves of a SIMD (w, w + 4) take turns behind it: waves
        // 0..3 refill the ring and read stage t+1 from LDS while their partners' MFMAs of q-steps 2, 3 own the matrix pipe, then
        // the roles swap — neither the barrier, the LDS latency nor the load issue leaves the pipe without queued work.  The
        // stage loop is fully unrolled for the shapes of the chain; other stage counts take the plain rolled form ----
        f32x4 tot0 = {0.f, 0.f, 0.f, 0.f}, tot1 = {0.f, 0.f, 0.f, 0.f};
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        int nfold = 0;
        auto fold = [&]() {   // slice sums are added in index order — the split-K kernels' red[0] + red[1] + ...
            if (nfold == 0) { tot0 = acc0; tot1 = acc1; }
            else { tot0 += acc0; tot1 += acc1; }
            ++nfold;
            acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
            acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
        };
        // this wave's loads of stage s have landed once at most 4 * (stages issued after s) LDS-DMA loads are outstanding
        auto wait_stage = [&](int later) {
            if (later >= 3) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
            else if (later == 2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            else if (later == 1) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        };
        auto run = [&](auto Tc, auto Cc) {
            constexpr int TS = decltype(Tc)::value;    // stages
            constexpr int CN = decltype(Cc)::value;    // q-steps per slice
            f32x4 fx[2][WIDE_QS], fw0[2][WIDE_QS], fw1[2][WIDE_QS];
            auto read_stage = [&](int t) {
                const f32x4(*sg)[8][64] = ring[t & (WIDE_NS - 1)];
#pragma unroll
                for (int qq = 0; qq < WIDE_QS; ++qq) {
                    fx[t & 1][qq] = sg[qq][wr][lane];
                    fw0[t & 1][qq] = sg[qq][4 + 2 * wc][lane];
                    fw1[t & 1][qq] = sg[qq][5 + 2 * wc][lane];
                }
            };
            auto mfma_q = [&](int t, int qq) {
                if (abl_nomfma) return;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fw0[t & 1][qq][e], fx[t & 1][qq][e], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fw1[t & 1][qq][e], fx[t & 1][qq][e], acc1, 0, 0, 0);
                }
            };
            wait_stage((TS < 2 ? TS : 2) - 1);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (it == 0) TS_STAMP(6);
            read_stage(0);
#pragma unroll
            for (int t = 0; t < TS; ++t) {
                // issue schedule: stages 0, 1 in the prologue; 2, 3, 4 behind the barrier of stage 0; t + 4 behind the barrier of stage t
                const bool more = t + 1 < TS;
                auto refill = [&]() {
                    if (t == 0) {
                        epi_load();
                        asm volatile("" ::: "memory");
                        if (2 < TS) issue(2);
                        if (3 < TS) issue(3);
                    }
                    if (t + WIDE_NS < TS) issue(t + WIDE_NS);   // into the slot of stage t
                };
                auto open_next = [&]() {   // the barrier that opens stage t+1
                    if (!more) return;
                    const int youngest = t == 0 ? 1 : (t + WIDE_NS - 1 < TS - 1 ? t + WIDE_NS - 1 : TS - 1);
                    wait_stage(youngest - (t + 1));   // ... and every LDS read of stage t is done (lgkmcnt)
                    __builtin_amdgcn_s_barrier();     // everybody's loads of stage t+1 have landed; everybody holds stage t in registers
                    asm volatile("" ::: "memory");
                    if (it == 0 && t + 1 < 8) TS_STAMP(6 + t + 1);
                };
                // The matrix pipe goes to the OLDER wave whenever both have an MFMA ready (tools/mfma_rate.cpp), and an LDS-DMA
                // load takes ~190 cycles to issue against the CU's fetch rate: a wave's refill + LDS reads (~900 cycles) need
                // the partner's MFMAs as cover.  The younger wave therefore crosses the barrier one q-step EARLY, keeping 24
                // MFMAs for the time its partner loads; the older wave then computes 32 in a row while the younger one loads.
                // (sched_barrier: hipcc otherwise sinks the LDS reads to just ahead of their MFMAs — lgkmcnt(0) stalls with no cover)
                if (wave < 4) {
                    mfma_q(t, 0);
                    mfma_q(t, 1);
                    if (CN == 2) fold();
                    open_next();
                    __builtin_amdgcn_sched_barrier(0);
                    refill();
                    if (more) read_stage(t + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_q(t, 2);
                    mfma_q(t, 3);
                    fold();
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    mfma_q(t, 0);
                    open_next();
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_q(t, 1);
                    if (CN == 2) fold();
                    mfma_q(t, 2);
                    mfma_q(t, 3);
                    fold();
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) read_stage(t + 1);
                    refill();
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        auto run_rolled = [&]() {   // stage t is consumed while stages t+1 .. t+3 are in flight; one barrier per stage
            for (int t = 0; t < T; ++t) {   // issue schedule: stages 0, 1 in the prologue; 2 behind the barrier of stage 0; t + 2 behind that of stage t
                const int youngest = t + 1 < T - 1 ? t + 1 : T - 1;
                wait_stage(youngest - t);
                __builtin_amdgcn_s_barrier();   // everybody's loads of stage t have landed; everybody is done reading stage t-1
                asm volatile("" ::: "memory");
                if (it == 0 && t < 8) TS_STAMP(6 + t);
                if (t + 2 < T) issue(t + 2);   // into a slot last read two stages ago
                const f32x4(*sg)[8][64] = ring[t & (WIDE_NS - 1)];
#pragma unroll
                for (int qq = 0; qq < WIDE_QS; ++qq) {
                    const f32x4 xa = sg[qq][wr][lane], w0 = sg[qq][4 + 2 * wc][lane], w1 = sg[qq][5 + 2 * wc][lane];
                    if (!abl_nomfma) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[e], xa[e], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[e], xa[e], acc1, 0, 0, 0);
                        }
                    }
                    if (qq == 3 || (qq == 1 && cnt == 2)) fold();
                }
            }
        };
        if (T == 8 && cnt == 4) run(IC<8>{}, IC<4>{});
        else if (T == 4 && cnt == 2) run(IC<4>{}, IC<2>{});
        else {
            epi_load();
            if (T > 0) run_rolled();
        }
        // the next tile of this workgroup: its first stages go out now and land under this tile's epilogue
        if (it + 1 < items && idx + 1 < ntile) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // the ring is reused: every wave is done reading this tile's stages
            asm volatile("" ::: "memory");
            int tile2, mt2;
            tile_of(idx + 1, tile2, mt2);
            setup_loader(tile2, mt2);
            if (T > 0) issue(0);
            if (T > 1) issue(1);
            prefetched = true;
        }

        if (TRACE && it == 0) {
            TS_STAMP(2);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            TS_STAMP(3);
        }
        // ---- epilogue: D[i = channel lg * 4 + r][j = clip li] ----
        gf *out = (gf *)P(SD_OUT);
        const int out_stride = I(SD_OUT_STRIDE);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            f32x4 v = cb ? tot1 : tot0;
            v += ((t0[cb] + t1[cb]) + t2[cb]) + t3[cb];
            if (gate) {
                if ((flags & SDF_PRE) && m_ok) {
                    const long ip = (long)mrow * I(SD_PRE_STRIDE) + n0[cb];
                    *reinterpret_cast<gf4 *>((gf *)P(SD_PRE) + (pre_tw ? tiled_index(ip, pre_tw) : ip)) = v;
                }
                v += ecls[cb];
                f32x4 partner;
#pragma unroll
                for (int r = 0; r < 4; ++r) partner[r] = __shfl_xor(v[r], 32);
                if (lg < 2 && m_ok) {
                    f32x4 g;
#pragma unroll
                    for (int r = 0; r < 4; ++r) g[r] = gate_act(v[r], partner[r]);
                    const long io = (long)mrow * out_stride + oc[cb];
                    *reinterpret_cast<gf4 *>(out + (out_tw ? tiled_index(io, out_tw) : io)) = g;
                }
            } else {
                if (flags & SDF_RELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
                }
                if (m_ok) {
                    const long io = (long)mrow * out_stride + oc[cb];
                    *reinterpret_cast<gf4 *>(out + (out_tw ? tiled_index(io, out_tw) : io)) = v;
                }
            }
        }
    }
    if (TRACE && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TS_STAMP(4);
        stamps[15] = clock64() - clk0;   // shader cycles entry -> end: with the 100 MHz stamps, the clock this workgroup ran at
        stamps[5] = (1ull << 62) | ((unsigned long long)z << 48) | ((unsigned long long)Q << 32) | ((unsigned long long)items << 24) | gridDim.x;
        unsigned long long *rec = (unsigned long long *)(((uint64_t)(uint32_t)batch.start[7] << 32) | (uint64_t)(uint32_t)batch.start[6]);
        if (rec && blockIdx.x < 512u)
            for (int k = 0; k < 16; ++k) rec[(size_t)blockIdx.x * 24 + k] = stamps[k];
    }
#undef TS_STAMP
}

hipError_t launch_skinny_wide(const SkinnyDescBatch &db, int workgroups, hipStream_t stream, bool trace) {
    if (trace) hipLaunchKernelGGL(skinny16_wide_kernel<true>, dim3(workgroups), dim3(512), 0, stream, db);
    else hipLaunchKernelGGL(skinny16_wide_kernel<false>, dim3(workgroups), dim3(512), 0, stream, db);
    return hipGetLastError();
}

}  // namespace ts
