// conv_taps48: grouped convolution with many taps and 48 channels per group, in and out — the positional convolution of the
// wav2vec2 encoder (HF Wav2Vec2PositionalConvEmbedding under nets/spg/wav2vec.py: Conv1d(768, 768, kernel 128, padding 64,
// groups 16) + drop of the last frame + GELU; the residual add of the encoder is fused here as well), one GEMM of
// M = B T rows x 48 columns x K = 128 taps x 48 channels per group.
//
// conv_gemm_f32's tiles carry it as 64-channel windows every 48 channels on 64 x 64 tiles of the 32 x 32 MFMA: a quarter of the
// K walk multiplies zero weights and a quarter of the columns are padding (2.7 ms per face batch of 64 at 113 TFLOP/s of executed,
// 64 of useful work).  Here nothing is padded:
//   * v_mfma_f32_16x16x4_f32 (the same 256 flop / cycle / CU as the 32 x 32 form): 48 = 3 blocks of 16 channels; the weights are the
//     A operand, the activations the B operand, so a lane ends with 4 consecutive channels of one row (16-byte stores);
//   * a stage = ONE tap = 48 consecutive floats (192 B) of each of the tile's 128 activation rows and of the group's 48 weight
//     rows ([48][taps * 48], k contiguous), brought by global_load_lds_dwordx4 into a two-slot ring exactly as conv_gemm_ring.hip
//     does; the next tap of an activation row is the next input row: every lane's source pointer advances by the row stride, rows
//     outside the clip ([0, T): the zero padding of the convolution) read a zero buffer;
//   * LDS rows are 192 B (48 banks): the 16-byte segment s of row r sits at position (s + (r >> 2)) mod 12 — rows r, r + 1, r + 2,
//     r + 3 start 48 banks apart, the rotation separates the four groups of four, and the 16 lanes a ds_read_b128 serves together
//     cover the 64 banks exactly once; as in the ring engine the permutation is applied to the DMA's SOURCE address;
//   * 8 waves of 16 rows x 48 channels per 128-row tile, two workgroups per CU (66 KB of LDS each); 1-D grid, the (group, row
//     tile) list dealt to the XCDs in contiguous eighths: an XCD works on two groups at a time, their weights (1.2 MB each) stay
//     in its L2.
// The k order inside a tap differs from conv_gemm_f32's (k = 16 q + 4 (lane / 16) + e per MFMA e of group q), so results agree
// with it to rounding, not bit for bit; the face goldens bound both.
#include "conv_tile.h"

namespace ts {

namespace {

constexpr int T48_BM = 128, T48_C = 48;
constexpr int T48_A = T48_BM * T48_C;        // floats of a stage's activation block (24 KB = 24 DMA instructions)
constexpr int T48_W = T48_C * T48_C;         // ... of its weight block (9 KB = 9 DMA instructions)
constexpr int T48_STAGE = T48_A + T48_W;

__device__ __forceinline__ void t48_glds16(const float *src, float *lds_dst) {   // lds_dst: wave-uniform; lane i lands at + 16 i bytes
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds_dst, 16, 0, 0);
}

}  // namespace

__global__ __launch_bounds__(512, 4) void conv_taps48_kernel(const ConvParams p) {
    __shared__ __attribute__((aligned(1024))) float smem[2 * T48_STAGE];
    const ConvGroup &g = p.g[0];
    // ---- tile of this workgroup: ids round-robin over the XCDs, XCD c takes the c-th contiguous eighth of the (group, row tile) list ----
    const int mt = (p.M + T48_BM - 1) / T48_BM, total = mt * p.ngroups, per = (total + 7) >> 3;
    const int l0 = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    if (l0 >= total) return;
    const int zidx = l0 / mt, m0 = (l0 - zidx * mt) * T48_BM;
    const ConvTilePtrs tp = conv_tile_ptrs(p, g, zidx);
    const int ntap = g.seg[0].ntap, d0 = g.seg[0].d;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- loader.  DMA instruction gi of a block writes its bytes [1024 gi, 1024 gi + 1024) (lane-linear); byte 16 j of the block is row
    // j / 12, position j % 12, and holds segment (position - (row >> 2)) mod 12 of that row ----
    auto place = [&](int gi, int &row, int &sseg) {
        const int j = gi * 64 + lane;
        row = j / 12;
        const int pos = j - row * 12;
        sseg = (pos + 12 - ((row >> 2) & 3)) % 12;
    };
    const float *pa[3];   // activation pointers of the stage's tap, valid or not
    int lo[3];            // tap index from which row t + d0 + tap is inside the clip; it stays inside for T taps
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        int row, sseg;
        place(wave * 3 + o, row, sseg);
        const int m = m0 + row;
        if (m < p.M) {
            const int b = m / p.Lout, t = m - b * p.Lout;
            pa[o] = tp.x + ((long)b * p.Lin + t + d0) * p.ldx + g.seg[0].c0 + sseg * 4;
            lo[o] = -(t + d0);
        } else {
            pa[o] = tp.x;
            lo[o] = 0x40000000;   // never inside
        }
    }
    const float *pw[2];
    {
        int row, sseg;
        place(wave, row, sseg);
        pw[0] = tp.w + (long)row * p.Ktot + sseg * 4;
        place(8, row, sseg);      // the ninth instruction of the weight block: wave 0 only
        pw[1] = tp.w + (long)row * p.Ktot + sseg * 4;
    }
    const float *zero = p.zero + lane * 4;
    int tap = 0;
    auto dma_one = [&](int slot, int o) {   // DMA instruction o of tap `tap`
        float *dst = smem + slot * T48_STAGE;
        if (o < 3) {
            const bool in = (unsigned)(tap - lo[o]) < (unsigned)p.Lin;
            t48_glds16(in ? pa[o] : zero, dst + (wave * 3 + o) * 256);
        } else if (o == 3) {
            t48_glds16(pw[0], dst + T48_A + wave * 256);
        } else if (wave == 0) {
            t48_glds16(pw[1], dst + T48_A + 8 * 256);
        }
    };
    auto advance = [&]() {
        tap += 1;
#pragma unroll
        for (int o = 0; o < 3; ++o) pa[o] += p.ldx;
        pw[0] += T48_C;
        pw[1] += T48_C;
    };

    // ---- reader: lane (i = lane % 16, kq = lane / 16) takes row i of a block of 16 rows, k = 16 q + 4 kq .. + 3 of group q = segment
    // 4 q + kq, at position (4 q + kq + (i >> 2)) mod 12 — the rotation is the same for activation and weight rows (16 w + i, 16 blk + i) ----
    const int li = lane & 15, kq = lane >> 4;
    int foff[3];   // float offset of this lane's fragment of group q in the activation block; weight block blk: + T48_A - 16 w 48 + 16 blk 48
#pragma unroll
    for (int q = 0; q < 3; ++q) foff[q] = (wave * 16 + li) * T48_C + ((4 * q + kq + (li >> 2)) % 12) * 4;
    const int wrel = T48_A - wave * 16 * T48_C;
    f32x4 fa[3], fw[3][3];   // [group] / [group][channel block]
    auto read_one = [&](int slot, int q, int f) {   // f = 0: activations, 1..3: weight block f - 1
        const float *src = smem + slot * T48_STAGE + foff[q] + (f == 0 ? 0 : wrel + (f - 1) * 16 * T48_C);
        const f32x4 v = *(const volatile lds_f32x4 *)__builtin_assume_aligned(src, 16);
        if (f == 0) fa[q] = v;
        else fw[q][f - 1] = v;
    };
    f32x4 acc[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto mfma_one = [&](int q, int k) {   // k-th of the 12 MFMAs of group q: e-major
        const int e = k / 3, b = k % 3;
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[q][b][e], fa[q][e], acc[b], 0, 0, 0);
    };
    auto wait_all = [&]() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); };

    // ---- prologue: tap 0 lands, its first fragments are read ----
#pragma unroll
    for (int o = 0; o < 5; ++o) dma_one(0, o);
    wait_all();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int f = 0; f < 4; ++f) read_one(0, 0, f);

    // ---- one tap out of ring slot `slot`; MORE: the next tap exists and is issued into the other slot (free since the barrier that
    // opened this one) behind the first MFMAs.  One barrier per tap, in the middle of the last group ----
    auto stage = [&](auto Mc, const int slot) {
        constexpr bool MORE = decltype(Mc)::value;
        if (MORE) advance();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 12; ++k) {   // group 0: the refill (5 instructions) + the fragments of group 1
            mfma_one(0, k);
            __builtin_amdgcn_sched_barrier(0);
            if (k < 5) {
                if (MORE) dma_one(slot ^ 1, k);
            } else if (k < 9) {
                read_one(slot, 1, k - 5);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int k = 0; k < 12; ++k) {   // group 1: the fragments of group 2
            mfma_one(1, k);
            __builtin_amdgcn_sched_barrier(0);
            if (k < 4) read_one(slot, 2, k);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) mfma_one(2, k);
        __builtin_amdgcn_sched_barrier(0);
        if (MORE) {
            wait_all();                       // this wave's loads of the next tap have landed, its reads of this one are done
            __builtin_amdgcn_s_barrier();     // ... and everybody's
            asm volatile("" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 6; k < 12; ++k) {
            mfma_one(2, k);
            if (MORE && k - 6 < 4) {
                __builtin_amdgcn_sched_barrier(0);
                read_one(slot ^ 1, 0, k - 6);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    int slot = 0;
    for (int t = 0; t + 1 < ntap; ++t) {
        stage(std::true_type{}, slot);
        slot ^= 1;
    }
    stage(std::false_type{}, slot);

    // ---- epilogue: lane (li, kq) holds channels 16 blk + 4 kq .. + 3 of row 16 w + li; GELU(acc + bias) (+ residual), as conv_tile_epilogue ----
    const int m = m0 + wave * 16 + li;
    if (m >= p.M) return;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const int c = 16 * b + 4 * kq;
        const f32x4 bv = tp.bias ? *reinterpret_cast<const f32x4 *>(tp.bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 rv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (tp.res) rv = *reinterpret_cast<const f32x4 *>(tp.res + (long)m * p.ldr + c);
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float u = acc[b][r] + bv[r];
            if (tp.res && !p.res_after_act) u += rv[r];
            if (p.act == 3) u = gelu_fast(u);
            if (tp.res && p.res_after_act) u += rv[r];
            v[r] = u;
        }
        *reinterpret_cast<f32x4 *>(tp.out + (long)m * p.ldo + g.out_col0 + c) = v;
    }
}

// batched problems (zdiv) of one segment of 48-channel taps, 48 columns, stride 1, same length in and out, no or GELU activation,
// every pointer and row stride 16-byte aligned
bool conv_taps48_takes(const ConvParams &p) {
    const ConvGroup &g = p.g[0];
    auto al = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    auto al4 = [](long v) { return (v & 3) == 0; };
    return p.zdiv > 0 && p.zdiv == p.ngroups && g.nseg == 1 && g.seg[0].len == T48_C && g.seg[0].ntap >= 1 && p.N == T48_C && p.stride == 1 &&
           p.Lin == p.Lout && p.Lin > 0 && p.M % p.Lout == 0 && (p.act == 0 || p.act == 3) && p.Ktot == g.seg[0].ntap * T48_C && p.ldw == 0 && p.w_rows == 0 &&
           !p.w_planes && al(g.x) && al(g.w) && al(g.out) && al(g.bias) && al(g.res) && al4(p.ldx) && al4(p.ldo) && al4(p.ldr) && al4(g.out_col0) &&
           al4(g.seg[0].c0) && al4(p.x_zs1) && al4(p.w_zs1) && al4(p.o_zs1) && al4(p.b_zs1) && al4(p.r_zs1) && al4(p.x_zs0) && al4(p.w_zs0) && al4(p.o_zs0) &&
           al4(p.r_zs0);
}

hipError_t launch_conv_taps48(const ConvParams &p_in, hipStream_t stream) {
    ConvParams p = p_in;
    if (!p.zero) {
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) p.zero = skinny_zero_buffer(dev);
    }
    if (!p.zero || !conv_taps48_takes(p)) return hipErrorInvalidValue;
    const long total = (long)((p.M + T48_BM - 1) / T48_BM) * p.ngroups;
    hipLaunchKernelGGL(conv_taps48_kernel, dim3(8 * (unsigned)((total + 7) / 8)), dim3(512), 0, stream, p);
    return hipGetLastError();
}

}  // namespace ts
