// chain_persist_kernel — the whole PixelCNN chain of a coalesced pass (75 code rows x 38 dependent stages) as ONE kernel.
//
// Replaces, for passes of 128 / 256 / 384 / 512 clips, the hipGraph of ~2 850 dependent launches that skinny_gemm.hip /
// skinny_wide.hip / vq.hip (sampler) otherwise serve (reference: nets/spg/gated_pixelcnn_v2.py:61-87,120-124,137-165 — the
// gated layers, head and sampling loop of `generate`).
//
// Why it can be one kernel: nothing in the chain mixes clips.  Every stage is out[clip, :] = f(in[clip, :], weights), the
// sampler is per clip, the row cache is per clip.  So the clips are cut into 8 groups, one per XCD, and an XCD takes ITS clips
// through every stage of every row on its own: the only synchronisation is among the workgroups of one XCD, between stages —
// a counter bumped and polled with atomics that execute in that XCD's L2 (no sc1: never leaves the die), data handed over with
// plain stores (write-through to the L2, s_waitcnt vmcnt(0) before arriving) and read after an L1 invalidate (buffer_inv sc0).
// tools/xcd_barrier.cpp: 1.6 us per barrier + dependent exchange, against 23 us with agent-scope atomics and fences, and
// against the ~4 us (dispatch + cold descriptor + cold first operand) a kernel boundary costs the per-launch path.
//
// A stage is the SkinnyDescBatch the per-launch path would have launched (same descriptors, recorded instead of launched:
// launch_skinny_batch / launch_sample with a ChainRecorder set).  A tile is 16 RB clips (the XCD's clips) x 32 columns, K split
// over the stage's 4 or 8 waves with the fixed-order LDS sum of skinny16_fast_kernel — the same products in the same order as
// every other chain kernel: bit-identical results.
#include "skinny_desc.h"
#include "../../include/talkshow_hip.h"

#include <mutex>

namespace ts {

namespace {

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
// atomics without sc1 execute in the L2 of the XCD the wave runs on
__device__ __forceinline__ unsigned l2_add_ret(unsigned *p, unsigned v) {
    unsigned r;
    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p), "v"(v) : "memory");
    return r;
}
__device__ __forceinline__ void l2_add(unsigned *p, unsigned v) {
    asm volatile("global_atomic_add %0, %1, off" : : "v"(p), "v"(v) : "memory");
}

// all workgroups of this XCD have finished the stage (their stores are in the L2); false = timed out (another workgroup of
// the XCD never arrived: the kernel gives up instead of hanging the device)
__device__ __forceinline__ bool xcd_barrier(ChainSync *s, unsigned xcd, unsigned target, unsigned *abort_host) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ unsigned ok_sh;
    if (threadIdx.x == 0) {
        unsigned *c = &s->arrive[xcd][0];
        unsigned ok = 1;
        l2_add(c, 1u);
        const unsigned long long t0 = wall_clock64();
        while (l2_add_ret(c, 0u) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 200000000ull) {   // 2 s
                s->abort_flag = 1;
                __hip_atomic_store(abort_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // pinned host word: the next launch reports it
                ok = 0;
                break;
            }
        }
        ok_sh = ok;
    }
    __syncthreads();
    asm volatile("buffer_inv sc0" ::: "memory");   // L1 invalidate: what the XCD's other workgroups wrote is read from the L2
    return ok_sh != 0;
}

__device__ inline void philox4x32_10_p(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t &o0) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o0 = c0;
}

// the sampler of clip b (vq.hip sample_kernel's arithmetic, statement for statement) by the first 256 threads of the workgroup
__device__ __forceinline__ void chain_sample_clip(const SampleParams &p, const int b, float *sf, int *si, float *s_thr) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool act = tid < 256;
    const float *lg = p.logits + (long)b * p.V;
    const int chunk = (p.V + 255) / 256;
    const int v0 = tid * chunk, v1 = min(v0 + chunk, p.V);
    const bool fast = chunk == 8 && (p.V & 7) == 0;
    float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (fast && act) {
        const f32x4 lo = *reinterpret_cast<const f32x4 *>(lg + v0), hi = *reinterpret_cast<const f32x4 *>(lg + v0 + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { x[k] = lo[k]; x[4 + k] = hi[k]; }
    }
    float best = -INFINITY;
    int bi = 0x7fffffff;
    if (act) {
        if (fast) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (x[k] > best) { best = x[k]; bi = v0 + k; }
        } else {
            for (int v = v0; v < v1; ++v) {
                const float t = lg[v];
                if (t > best) { best = t; bi = v; }
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off);
        const int oi = __shfl_xor(bi, off);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0 && act) { sf[wave] = best; si[wave] = bi; }
    __syncthreads();
    best = sf[0]; bi = si[0];
    for (int w = 1; w < 4; ++w)
        if (sf[w] > best || (sf[w] == best && si[w] < bi)) { best = sf[w]; bi = si[w]; }
    __syncthreads();

    int choice = bi;
    if (p.mode != TS_SAMPLE_GREEDY) {   // wave-uniform
        float u;
        if (p.mode == TS_SAMPLE_UNIFORMS) {
            u = p.uniforms[(long)b * p.u_stride];
        } else {
            const uint64_t seed = p.dyn ? p.dyn[0] : p.seed;
            const uint64_t clip = (uint64_t)((p.dyn ? (int64_t)p.dyn[1] : p.clip_index0) + b);
            uint32_t r;
            philox4x32_10_p(p.position + (p.dyn ? (uint32_t)p.dyn[2] : 0u), (uint32_t)clip, (uint32_t)(clip >> 32), 0u, (uint32_t)seed,
                            (uint32_t)(seed >> 32), r);
            u = (float)(r >> 8) * (1.0f / 16777216.0f);
        }
        float s = 0.f;
        if (act) {
            if (fast) {
#pragma unroll
                for (int k = 0; k < 8; ++k) s += det_expf(x[k] - best);
            } else {
                for (int v = v0; v < v1; ++v) s += det_expf(lg[v] - best);
            }
            sf[tid + 1] = s;
        }
        __syncthreads();
        if (tid == 0) {
            float c = 0.f;
            sf[0] = 0.f;
            for (int t = 1; t <= 256; ++t) { c += sf[t]; sf[t] = c; }   // sf[t] = sum of chunks < t
            *s_thr = u * c;
        }
        __syncthreads();
        const float thr = *s_thr;
        const bool mine = act && (sf[tid] <= thr) && (thr < sf[tid + 1] || tid == 255);
        if (mine && v0 < p.V) {
            float c = sf[tid];
            int k = v1 - 1;
            if (fast) {
                bool found = false;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    c += det_expf(x[j] - best);
                    if (!found && c > thr) { k = v0 + j; found = true; }
                }
            } else {
                for (int v = v0; v < v1; ++v) {
                    c += det_expf(lg[v] - best);
                    if (c > thr) { k = v; break; }
                }
            }
            si[0] = k;
        } else if (mine) {
            si[0] = p.V - 1;
        }
        __syncthreads();
        choice = si[0];
    }
    if (tid == 0) {
        p.tok32[(long)b * p.tok_stride] = choice;
        p.codes[(long)b * p.code_stride] = choice;
    }
    __syncthreads();   // sf / si are reused by the next clip / stage
}

// Operands of one tile as this wave holds them between issue and use.
template <int RB, int CB>
struct TileOps {
    f32x4 a[4][RB], b[4][CB];        // the wave's K slice: up to 4 q-steps (16 k) of activations and weights
    int dw, tile, mt;                // the problem's descriptor word of this lane, column tile, row tile
};

template <int RB, bool PIPE>
__global__ __launch_bounds__(512, PIPE ? 2 : 3) void chain_persist_kernel(const ChainStage *__restrict__ stages, const int nstages, ChainSync *sync, unsigned long long *trace,
                                                                      unsigned *abort_host) {
    constexpr int CB = 2, ROWS = RB * 16, NBLK = RB * CB, NREG = NBLK * 4, EMAX = RB * CB;   // EMAX = NREG / 4 (a W = 4 stage)
    __shared__ float red_raw[8 * NREG * 64];
    __shared__ float sf[256 + 1];
    __shared__ int si[256];
    __shared__ float s_thr;
    __shared__ unsigned who[2];
    float (*red)[NREG][64] = reinterpret_cast<float (*)[NREG][64]>(red_raw);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    if (tid == 0) {
        const unsigned x = xcc_id() & 7;
        who[0] = x;
        who[1] = l2_add_ret(&sync->slot[x][0], 1u);
    }
    __syncthreads();
    const int xcd = who[0], slot = who[1], G = gridDim.x >> 3;
    if (slot >= G) return;   // the grid was not spread evenly over the XCDs (never seen): the others time out and give up

    typedef __attribute__((address_space(1))) const int gcw;
    auto sdiv = [](int x, int d) { return (d & (d - 1)) == 0 ? x >> __builtin_ctz(d) : x / d; };

    // ---- stage headers + descriptors, fetched two stages ahead: lane l holds header word l / descriptor word l.  Named fields,
    //      not an array: hipcc turns a select chain over array elements into a scratch-memory lookup ----
    struct StageRegs { int hdr, d0, d1, d2, d3, d4, d5; };
    static_assert(SKINNY_MAX_PROBLEMS == 6, "StageRegs holds six descriptors");
    auto fetch_stage = [&](int st, StageRegs &r) {
        const int s_ = st < nstages ? st : nstages - 1;
        const ChainStage *S = stages + s_;
        r.hdr = ((gcw *)S)[lane < 12 ? lane : 0];
        r.d0 = ((gcw *)S->d[0].w)[lane];
        r.d1 = ((gcw *)S->d[1].w)[lane];
        r.d2 = ((gcw *)S->d[2].w)[lane];
        r.d3 = ((gcw *)S->d[3].w)[lane];
        r.d4 = ((gcw *)S->d[4].w)[lane];
        r.d5 = ((gcw *)S->d[5].w)[lane];
    };
    auto H = [&](const StageRegs &r, int word) { return __builtin_amdgcn_readlane(r.hdr, word); };

    // ---- tile t of a stage: problem, column tile, row tile ----
    auto decode = [&](const StageRegs &r, int t, TileOps<RB, CB> &o) {
        int z = 0, first = 0;
#pragma unroll
        for (int i = 1; i < SKINNY_MAX_PROBLEMS; ++i) {
            const int s0 = H(r, 4 + i);
            if (t >= s0) { z = i; first = s0; }
        }
        // v_cndmask by hand: hipcc turns a select chain over the six descriptor registers into a scratch-memory lookup
        auto pick = [](int dw, int cand, int zz, int i) {
            asm volatile("v_cmp_eq_u32 vcc, %2, %3\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(dw) : "v"(cand), "v"(zz), "v"(i) : "vcc");
            return dw;
        };
        int dw = r.d0;
        dw = pick(dw, r.d1, z, 1);
        dw = pick(dw, r.d2, z, 2);
        dw = pick(dw, r.d3, z, 3);
        dw = pick(dw, r.d4, z, 4);
        dw = pick(dw, r.d5, z, 5);
        // rows are clips, or (clip, column) pairs clip-major (M = 2 x clips): the XCD's clips are then two row tiles
        const int two = __builtin_amdgcn_readlane(dw, SD_M) >= 2 * 8 * ROWS;
        const int u = t - first;
        o.dw = dw;
        o.tile = two ? u >> 1 : u;
        o.mt = two ? xcd * 2 + (u & 1) : xcd;
    };
#define CH_I(dw_, k_) __builtin_amdgcn_readlane(dw_, k_)
#define CH_P(dw_, k_) ((gcf *)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(dw_, (k_) + 1) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane(dw_, k_)))
    // column of lane li in 16-column block t16 (gate tiles: 8 tanh channels + their 8 sigmoid partners)
    auto column = [&](int dw, int t16) {
        const int flags = CH_I(dw, SD_FLAGS), gateD = CH_I(dw, SD_GATED);
        if (flags & SDF_GATE) {
            const int tiles_per_group = gateD >> 3;
            const int group = sdiv(t16, tiles_per_group), ch0 = (t16 - group * tiles_per_group) << 3;
            return group * 2 * gateD + (li >> 3) * gateD + ch0 + (li & 7);
        }
        return t16 * 16 + li;
    };

    // ---- weights of the wave's K slice (static data: may be issued before the stage barrier) ----
    auto issue_b = [&](TileOps<RB, CB> &o, int W) {
        if (wave >= W) return;
        const int dw = o.dw, cnt = CH_I(dw, SD_CNT), q0 = wave * cnt, wtq = CH_I(dw, SD_WTQ), N = CH_I(dw, SD_N);
        const int bstep = wtq ? 256 : 16;
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            const int t16 = o.tile * CB + c;
            int nn = column(dw, t16);
            nn = nn < N ? nn : 0;
            gcf *bp = wtq ? CH_P(dw, SD_W) + ((((long)t16 * wtq + q0) << 6) + lane) * 4
                          : CH_P(dw, SD_W) + (long)nn * CH_I(dw, SD_LDW) + q0 * 16 + lg * 4;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (u < cnt) o.b[u][c] = *reinterpret_cast<gcf4 *>(bp + u * bstep);
        }
    };
    // ---- activations of the wave's K slice (after the barrier) ----
    auto issue_a = [&](TileOps<RB, CB> &o, int W) {
        if (wave >= W) return;
        const int dw = o.dw, mt = o.mt, cnt = CH_I(dw, SD_CNT), q0 = wave * cnt, M = CH_I(dw, SD_M);
        int sbase = SD_SEG, qs = 0;
        {
            const int nseg = CH_I(dw, SD_NSEG);
            const int l0 = CH_I(dw, SD_SEG + 7);
            if (nseg > 1 && q0 >= l0) {
                sbase = SD_SEG + SD_SEG_WORDS;
                qs = l0;
                const int l1 = CH_I(dw, SD_SEG + SD_SEG_WORDS + 7);
                if (nseg > 2 && q0 >= l0 + l1) {
                    sbase = SD_SEG + 2 * SD_SEG_WORDS;
                    qs = l0 + l1;
                }
            }
        }
        gcf *base = CH_P(dw, sbase);
        gci *gidx = (gci *)CH_P(dw, sbase + 2);
        const int row_stride = CH_I(dw, sbase + 4), segw = CH_I(dw, sbase + 6);
        const bool a_tiled = segw & SEG_TILED;
        const int astep = a_tiled ? 256 : 16;
        const int koff = (q0 - qs) * 16 + lg * 4;
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int m = mt * ROWS + r * 16 + li;
            const int mc = m < M ? m : M - 1;
            gcf *ap;
            if (gidx) {
                const int g = gidx[(long)mc * CH_I(dw, sbase + 5)];
                ap = (g >= 0 ? base + (long)g * row_stride : CH_P(dw, SD_ZERO)) + koff;
            } else if (a_tiled) {
                const int nblk = (M + 15) >> 4;
                int blk = mt * RB + r;
                blk = blk < nblk ? blk : nblk - 1;
                ap = base + ((((long)blk * (segw & 0xffff) + (q0 - qs)) << 6) + lane) * 4;
            } else {
                ap = base + (long)mc * row_stride + koff;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (u < cnt) o.a[u][r] = *reinterpret_cast<gcf4 *>(ap + u * astep);
        }
    };
    // ---- MFMAs of the wave's slice (the order of skinny16_fast_kernel: q-step, k, row block, column block), the fixed-order sum
    //      over the waves through LDS, epilogue ----
    auto finish = [&](const TileOps<RB, CB> &o, int W) {
        const int dw = o.dw, cnt = CH_I(dw, SD_CNT), mt = o.mt;
        const bool kwave = wave < W;
        // epilogue operands of the registers this wave finishes: bias, add1, add2, add3, class row — back by the time the sum is
        float oe[EMAX][5];
        if (kwave) {
            const int M = CH_I(dw, SD_M), N = CH_I(dw, SD_N);
        const int rpw = W == 8 ? NREG / 8 : NREG / 4;
        const int add1_tw = CH_I(dw, SD_ADD1_TW);
#pragma unroll
        for (int rr = 0; rr < EMAX; ++rr)
            if (rr < rpw && rpw * wave < NREG) {
                const int r = wave * rpw + rr;
                const int blk = r >> 2, rb = blk / CB, cb = blk - rb * CB;
                const int row = mt * ROWS + rb * 16 + lg * 4 + (r & 3);
                const int rowc = row < M ? row : 0;
                int ncol = column(dw, o.tile * CB + cb);
                ncol = ncol < N ? ncol : 0;
                oe[rr][0] = CH_P(dw, SD_BIAS)[ncol];
                const long i1 = (long)(rowc >> CH_I(dw, SD_ADD1_SHIFT)) * CH_I(dw, SD_ADD1_STRIDE) + ncol;
                oe[rr][1] = CH_P(dw, SD_ADD1)[add1_tw ? tiled_index(i1, add1_tw) : i1];
                oe[rr][2] = CH_P(dw, SD_ADD2)[(long)(rowc >> CH_I(dw, SD_ADD2_SHIFT)) * CH_I(dw, SD_ADD2_STRIDE) + ncol];
                oe[rr][3] = CH_P(dw, SD_ADD3)[(long)rowc * CH_I(dw, SD_ADD3_STRIDE) + ncol];
                const int cls_ld = CH_I(dw, SD_CLS_LD);
                const int ccol = (cls_ld & (cls_ld - 1)) == 0 ? (ncol & (cls_ld - 1)) : ncol % cls_ld;
                oe[rr][4] = CH_P(dw, SD_CLS)[(long)rowc * cls_ld + ccol];
            }
        }
        if (kwave) {
            f32x4 acc[NBLK];
#pragma unroll
            for (int k = 0; k < NBLK; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (u < cnt) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int r = 0; r < RB; ++r)
#pragma unroll
                            for (int c = 0; c < CB; ++c)
                                acc[r * CB + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[u][r][e], o.b[u][c][e], acc[r * CB + c], 0, 0, 0);
                }
#pragma unroll
            for (int k = 0; k < NBLK; ++k)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave][k * 4 + r][lane] = acc[k][r];
        }
        __syncthreads();
        const int M = CH_I(dw, SD_M), N = CH_I(dw, SD_N), flags = CH_I(dw, SD_FLAGS), gateD = CH_I(dw, SD_GATED);
        const bool gate = flags & SDF_GATE;
        gf *out = (gf *)CH_P(dw, SD_OUT);
        const int out_stride = CH_I(dw, SD_OUT_STRIDE), out_tw = CH_I(dw, SD_OUT_TW), pre_tw = CH_I(dw, SD_PRE_TW);
        const int rpw = W == 8 ? NREG / 8 : NREG / 4;
#pragma unroll
        for (int rr = 0; rr < EMAX; ++rr)
            if (kwave && rr < rpw && rpw * wave < NREG) {
                const int r = wave * rpw + rr;
                float v = red[0][r][lane];
                if (W == 8) {
#pragma unroll
                    for (int w = 1; w < 8; ++w) v += red[w][r][lane];
                } else {
#pragma unroll
                    for (int w = 1; w < 4; ++w) v += red[w][r][lane];
                }
                const int blk = r >> 2, rb = blk / CB, cb = blk - rb * CB;
                const int row = o.mt * ROWS + rb * 16 + lg * 4 + (r & 3);
                const int t16 = o.tile * CB + cb;
                const int ncol = column(dw, t16);
                const bool ok = ncol < N && row < M;
                v += ((oe[rr][0] + oe[rr][1]) + oe[rr][2]) + oe[rr][3];
                if (gate) {
                    if ((flags & SDF_PRE) && ok) {
                        const long ip = (long)row * CH_I(dw, SD_PRE_STRIDE) + ncol;
                        ((gf *)CH_P(dw, SD_PRE))[pre_tw ? tiled_index(ip, pre_tw) : ip] = v;
                    }
                    v += oe[rr][4];
                    const float partner = __shfl_xor(v, 8);
                    if ((li & 8) == 0 && ok) {
                        const float g = tanhf(v) * (1.0f / (1.0f + expf(-partner)));
                        const int tiles_per_group = gateD >> 3;
                        const int group = sdiv(t16, tiles_per_group), ch0 = (t16 - group * tiles_per_group) << 3;
                        const long io = (long)row * out_stride + group * gateD + ch0 + (li & 7);
                        out[out_tw ? tiled_index(io, out_tw) : io] = g;
                    }
                } else {
                    if (flags & SDF_RELU) v = v > 0.f ? v : 0.f;
                    if (ok) {
                        const long io = (long)row * out_stride + ncol;
                        out[out_tw ? tiled_index(io, out_tw) : io] = v;
                    }
                }
            }
        __syncthreads();   // red is reused by the next tile
    };

    // ---- the stage loop: s0 = this stage, s1 = the next one, s2 = the one being fetched ----
    StageRegs s0, s1, s2;
    fetch_stage(0, s0);
    fetch_stage(1, s1);
    // PIPE: the next tile's operands are issued before the current tile is finished, and a stage's first weights before the barrier
    // in front of it (two operand sets: 240 VGPRs).  Without it the kernel needs 125 and leaves room on the CU for a conv_gemm workgroup.
    TileOps<RB, CB> cur, nxt;
    if (PIPE && H(s0, 0) == 0 && slot < H(s0, 2)) {
        decode(s0, slot, cur);
        issue_b(cur, H(s0, 1));
    }
    unsigned epoch = 0;
    // tuning aid (TS_CHAIN_TRACE=1, tools/persist_trace.py): workgroup 0 of XCD 0 stamps the 100 MHz wall clock six times per stage
    const bool tr = trace && xcd == 0 && slot == 0 && tid == 0;
    for (int st = 0; st < nstages; ++st) {
        const int kind = H(s0, 0), W = H(s0, 1), T = H(s0, 2);
        if (tr) { trace[st * 8 + 0] = wall_clock64(); trace[st * 8 + 6] = (unsigned long long)kind << 32 | (unsigned)T; }
        if (kind == 0) {
            if (PIPE) {
                if (slot < T) issue_a(cur, W);   // its weights went out before the barrier
                if (tr) trace[st * 8 + 1] = wall_clock64();
                for (int t = slot; t < T; t += G) {
                    const int tn = t + G;
                    if (tn < T) {   // the next tile's operands fly under this tile's MFMAs, sum and epilogue
                        decode(s0, tn, nxt);
                        issue_b(nxt, W);
                        issue_a(nxt, W);
                    }
                    finish(cur, W);
                    cur = nxt;
                    if (tr && t == slot) trace[st * 8 + 2] = wall_clock64();
                }
            } else {
                if (tr) trace[st * 8 + 1] = wall_clock64();
                for (int t = slot; t < T; t += G) {
                    decode(s0, t, cur);
                    issue_b(cur, W);
                    issue_a(cur, W);
                    finish(cur, W);
                    if (tr && t == slot) trace[st * 8 + 2] = wall_clock64();
                }
            }
        } else {
            const SampleParams &sp = stages[st].sp;
            for (int b = slot; b < ROWS; b += G) chain_sample_clip(sp, xcd * ROWS + b, sf, si, &s_thr);
        }
        if (tr) trace[st * 8 + 3] = wall_clock64();
        // before the barrier: the header + descriptors of stage st + 2, and the weights of this workgroup's first tile of stage st + 1
        fetch_stage(st + 2, s2);
        if (PIPE && st + 1 < nstages && H(s1, 0) == 0 && slot < H(s1, 2)) {
            decode(s1, slot, cur);
            issue_b(cur, H(s1, 1));
        }
        epoch += (unsigned)G;
        if (tr) trace[st * 8 + 4] = wall_clock64();
        if (!xcd_barrier(sync, (unsigned)xcd, epoch, abort_host)) return;
        if (tr) trace[st * 8 + 5] = wall_clock64();
        s0 = s1;
        s1 = s2;
    }
#undef CH_I
#undef CH_P
}

std::mutex g_chain_mu;
unsigned long long *g_ctrace = nullptr;
int g_ctrace_stages = 0;
hipEvent_t g_chain_done[16] = {};
unsigned *g_chain_abort[16] = {};   // pinned host words: set by a kernel that timed out at a barrier
thread_local ChainRecorder *g_recorder = nullptr;

}  // namespace

ChainRecorder *chain_recorder() { return g_recorder; }
void chain_record_set(ChainRecorder *r) { g_recorder = r; }

// One persistent chain at a time per device: its workgroups wait for each other, so two of them sharing the CUs could each
// hold what the other needs.  Launches are chained through an event in submission order (the chains of passes on different
// streams could not overlap anyway: each wants every CU).
hipError_t launch_chain_persist(const ChainStage *stages, int nstages, ChainSync *sync, int clips, int wgs_per_cu, hipStream_t stream) {
    if (!stages || !sync || nstages < 1 || (clips != 128 && clips != 256)) return hipErrorInvalidValue;
    int dev = 0, cus = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 16) return hipErrorInvalidDevice;
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    if (cus % 8 != 0 || cus < 8) return hipErrorInvalidValue;
    const int grid = cus * (wgs_per_cu == 2 ? 2 : 1);
    std::lock_guard<std::mutex> lock(g_chain_mu);
    if (!g_chain_abort[dev]) {
        e = hipHostMalloc((void **)&g_chain_abort[dev], sizeof(unsigned), hipHostMallocMapped);
        if (e != hipSuccess) return e;
        *g_chain_abort[dev] = 0;
    }
    if (*(volatile unsigned *)g_chain_abort[dev]) {   // an earlier launch gave up (2 s at a barrier): its codes are garbage — say so
        *g_chain_abort[dev] = 0;
        return hipErrorLaunchTimeOut;
    }
    if (!g_chain_done[dev]) {
        e = hipEventCreateWithFlags(&g_chain_done[dev], hipEventDisableTiming);
        if (e != hipSuccess) return e;
    } else {
        e = hipStreamWaitEvent(stream, g_chain_done[dev], 0);
        if (e != hipSuccess) return e;
    }
    e = hipMemsetAsync(sync, 0, sizeof(ChainSync), stream);
    if (e != hipSuccess) return e;
    static const bool want_trace = getenv("TS_CHAIN_TRACE") && atoi(getenv("TS_CHAIN_TRACE"));
    unsigned long long *trace = nullptr;
    if (want_trace) {
        if (g_ctrace_stages < nstages) {
            if (g_ctrace) (void)hipFree(g_ctrace);
            e = hipMalloc(&g_ctrace, (size_t)nstages * 8 * sizeof(unsigned long long));
            if (e != hipSuccess) return e;
            g_ctrace_stages = nstages;
        }
        trace = g_ctrace;
    }
    const int rb = clips / 128;
    const bool pipe = wgs_per_cu != 3;   // 3: the lean form (one workgroup per CU, no operand prefetch)
    if (rb == 1 && pipe) hipLaunchKernelGGL((chain_persist_kernel<1, true>), dim3(grid), dim3(512), 0, stream, stages, nstages, sync, trace, g_chain_abort[dev]);
    else if (rb == 1) hipLaunchKernelGGL((chain_persist_kernel<1, false>), dim3(grid), dim3(512), 0, stream, stages, nstages, sync, trace, g_chain_abort[dev]);
    else if (rb == 2 && pipe) hipLaunchKernelGGL((chain_persist_kernel<2, true>), dim3(grid), dim3(512), 0, stream, stages, nstages, sync, trace, g_chain_abort[dev]);
    else if (rb == 2) hipLaunchKernelGGL((chain_persist_kernel<2, false>), dim3(grid), dim3(512), 0, stream, stages, nstages, sync, trace, g_chain_abort[dev]);
    else return hipErrorInvalidValue;
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return hipEventRecord(g_chain_done[dev], stream);
}

// the stamps of the last traced launch: 8 uint64 per stage (start, first operands issued, first tile done, tiles done, prefetch issued,
// barrier passed, kind << 32 | tiles, unused); returns the number of stages
int chain_trace_read(unsigned long long *out, int max_stages) {
    std::lock_guard<std::mutex> lock(g_chain_mu);
    if (!g_ctrace || !out) return -1;
    const int n = g_ctrace_stages < max_stages ? g_ctrace_stages : max_stages;
    if (hipMemcpy(out, g_ctrace, (size_t)n * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return n;
}

}  // namespace ts
