// Internal kernel-launch interface shared by the .hip kernel files and the host-side model code.
// Everything here is plain structs + launch functions; the public C ABI is include/talkshow_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ts {

// ------------------------------------------------------------------------------------------------
// conv_gemm_f32: 1-D convolution / transposed convolution / pointwise linear as an implicit GEMM on the
// fp32 MFMA (v_mfma_f32_32x32x2_f32).  Activations are NLC ([b][t][c], row stride ld floats, c padded to
// a multiple of 32 with zeros).  One output row m = b*Lout + t gathers, per segment s, `len` channels
// starting at c0 of input row t*stride + d (zero if outside [0, Lin)); the packed weight row n holds the
// segments back to back (Ktot floats, K-contiguous), BatchNorm already folded in.
// ------------------------------------------------------------------------------------------------
struct ConvSeg {
    int d;    // input row shift of the first tap
    int c0;   // first input channel
    int len;  // channels per tap (multiple of 32)
    int ntap; // consecutive taps d, d+1, ... sharing c0/len (0 or 1 = a single tap); K of the segment = ntap*len
};

struct ConvGroup {
    const float *x;     // input  [B*Lin][ldx]
    const float *w;     // packed weights [Npad][Ktot]
    const float *bias;  // [Npad]
    const float *res;   // optional residual [M][ldr] added before the activation
    float *out;         // output [M][ldo], written at columns out_col0 .. out_col0+N-1
    int out_col0;
    int nseg;
    ConvSeg seg[4];
};

struct ConvParams {
    int M, Lout, Lin, stride;
    int ldx, ldo, ldr;
    int N;     // columns stored per group (weights/bias are padded to a multiple of 128 rows)
    int Ktot;
    int act;   // 0 none, 1 LeakyReLU(0.2), 2 ReLU, 3 GELU (erf)
    int ngroups;
    ConvGroup g[4];
    // optional extras (0 = off)
    int res_after_act;   // residual is added after the activation instead of before it
    long ldw;            // weight row stride in floats (0 = Ktot, packed)
    int w_rows;          // number of valid weight rows (0 = padded to the grid); rows beyond read as zero
    // batched mode (zdiv > 0): blockIdx.z = z0*zdiv + z1 indexes independent problems that share g[0]'s geometry;
    // pointers advance by z0*zs0 + z1*zs1 floats
    const float *zero;   // >= 64 KB of zeros (filled by the launcher): where halo / padding operand pointers are parked
    int zdiv;
    long x_zs0, x_zs1, w_zs0, w_zs1, o_zs0, o_zs1, b_zs1, r_zs0, r_zs1;
    int sk_ok;           // the caller accepts the ring engine's stream-K plan: results within fp32 rounding of the whole-tile plans, deterministic,
                         // but a row's bits depend on M (which tiles get split).  Set by the face generator only; the body path never sets it
    float *sk_ws;        // stream-K band of the ring engine (conv_gemm_ring.hip): partial accumulators, two 128 x 128 fp32 blocks per band workgroup,
    int *sk_flags;       // and one arrival counter per band tile (zero between launches); set by launch_conv_gemm_ring_sk
    int w_planes;        // conv_gemm_split, 2 planes only: the weights are plane images already (split_weight_planes): no split of B in the kernel
    int xcd_tiles;       // set by launch_conv_gemm_split (0 or the column-group width): 1-D grid, tiles dealt to the XCDs in blocks that share operands
};

// banded conv_gemm launch (conv_gemm.hip): rows [0, mt_big * 128) in 128 x 128 tiles = workgroups [0, first_small), the rows
// after them in mt_small row blocks of the small tile shape = workgroups [first_small, total)
struct ConvBands {
    int mt_big, mt_small, first_small, total;
};

// stream-K launch of the ring engine (conv_gemm_ring.hip), per problem: row tiles [0, mt_dp) of 128 rows as whole 128 x 128 tiles
// (dp8 dealt workgroup ids), the mt_sk row tiles after them as ONE list of (tile, 32-k stage) iterations cut into wsk equal runs,
// one workgroup each; `stages` = Ktot / 32
struct ConvSK {
    int mt_dp, mt_sk, dp8, wsk, stages;
};
// The band's run arithmetic, one definition for the kernel and the host-side test (ts_debug_conv_sk_run): Ts band tiles of `stages` stages;
// XCD c holds tiles [tlo(c), tlo(c + 1)); its w8 = wsk / 8 runs cut its iterations evenly.  All in band-iteration units (tile * stages + stage).
struct SkRuns {
    int Ts, stages, w8;
    __host__ __device__ int tlo(int c) const { return (int)((long)c * Ts / 8); }
    __host__ __device__ int xcd_of_tile(int tile) const {
        int c = (int)((long)tile * 8 / Ts);
        if (c > 7) c = 7;
        while (c < 7 && tlo(c + 1) <= tile) ++c;
        while (c > 0 && tlo(c) > tile) --c;
        return c;
    }
    __host__ __device__ int begin(int c, int r) const {   // first iteration of run r of XCD c (r == w8: one past its last)
        const int base = tlo(c) * stages, Ic = (tlo(c + 1) - tlo(c)) * stages;
        return base + (int)((long)r * Ic / w8);
    }
    __host__ __device__ int run_of(int c, int x) const {  // the run of XCD c that holds iteration x
        const int base = tlo(c) * stages, Ic = (tlo(c + 1) - tlo(c)) * stages;
        int r = (int)((long)(x - base) * w8 / Ic);
        while (r + 1 < w8 && begin(c, r + 1) <= x) ++r;
        while (r > 0 && begin(c, r) > x) --r;
        return r;
    }
};

hipError_t conv_sk_probe_xcd_map(int device);                   // host, once per device (ts_ctx_create): do workgroup ids of equal residue mod 8 share an XCD?
bool conv_sk_supported();                                       // ... the answer for the current device (false before the probe ran)
bool conv_gemm_plan_sk_shape(const ConvParams &p, ConvSK &sk); // host only: the plan by shape (no device needed)
bool conv_gemm_plan_sk(const ConvParams &p, ConvSK &sk);       // host only: ... where the device supports it: false = the layer has no partly filled last unit worth splitting
hipError_t launch_conv_gemm_ring_sk(const ConvParams &p, const ConvSK &sk, hipStream_t stream);   // tile id 38
// host (models.cpp): per-stream scratch of the stream-K band — ws_floats floats + nflags zeroed ints, grown on demand, dropped with the stream
int conv_sk_workspace(hipStream_t s, size_t ws_floats, size_t nflags, float **ws, int **flags);

bool conv_gemm_band_plan(const ConvParams &p, ConvBands &bd);   // host only: the plan launch_conv_gemm(p, 0, ...) would use
bool conv_gemm_plan_bands(const ConvParams &p, ConvBands &bd);  // host only: the bands of a layer given to 128 x 128 tiles (false: none — under one round, or whole rounds)
// tile: 0 = auto, 1 = 128x128, 2 = 64x64, 3 = 128x64, 4 = 64x128, 5 = 64x64 (BK 64), 6 = 160x128, 7 = 96x128 (for tuning / tests);
// 31 / 39 / 33 = the LDS-DMA ring engine's 128x128 tile with 4 / 8 waves, its 96x128 tile (conv_gemm_ring.hip); 35 / 36 = 39 / 33 with the
// tiles dealt to the XCDs in operand-sharing blocks, 37 = bands + dealt tiles; 48 = conv_taps48.hip (batched problems of 48-channel taps only)
hipError_t launch_conv_gemm(const ConvParams &p, int tile, hipStream_t stream);
bool conv_taps48_takes(const ConvParams &p);   // host: would tile 48 (conv_taps48.hip) take this layer as laid out?
// the same convolution on the bf16 matrix cores with fp32 operands split into `planes` bf16 terms (2: three products, ~2^-16;
// 3: six products, fp32 grade) — conv_gemm_split.hip; an opt-in plan for tolerance-only GEMMs (the face generator)
hipError_t launch_conv_gemm_split(const ConvParams &p, int planes, hipStream_t stream);
// conv_gemm_split's XCD-aware tile order (one definition for the kernel and for the host-side test): workgroup `bid` of a 1-D grid of
// 8 ceil(MT NT / 8) -> tile (tx, ty) of an MT x NT tile grid, false if the workgroup has no tile.  Workgroup ids go round-robin over the 8
// XCDs; XCD x takes the x-th contiguous eighth of a tile list that runs through column groups of GW tiles, rows inside a group, columns fastest.
__host__ __device__ inline bool split_tile_of(int bid, int MT, int NT, int GW, int &tx, int &ty) {
    const int total = MT * NT, per = (total + 7) >> 3;
    const int L = (bid & 7) * per + (bid >> 3);
    if (L >= total) return false;
    const int full = NT / GW;
    int g = L / (MT * GW), r = L - g * MT * GW, gn = GW;
    if (g >= full) {   // the last, narrower column group
        g = full;
        r = L - full * MT * GW;
        gn = NT - GW * full;
    }
    tx = r / gn;
    ty = g * GW + (r - tx * gn);
    return true;
}

// Weights of a layer as the plane images conv_gemm_split builds in LDS — per row and chunk of 32 k: 16 dwords of bf16(x) pairs, then 16
// dwords of bf16(x - bf16(x)) pairs; the same size and pitch as the fp32 matrix (rows x K floats, K % 32 == 0).  Done once per layer when
// the x3 plan is selected: weights are constants, only the activations need splitting per call.
hipError_t launch_split_weight_planes(const float *w, float *planes, long rows, int K, hipStream_t stream);
double conv_gemm_flops(const ConvParams &p);

// ------------------------------------------------------------------------------------------------
// skinny_gemm_f32: out[M x N] = A[M x K] * W[N x K]^T with M = a few tens of rows (the batch of clips at one
// code position).  The A operand is a concatenation of up to 6 segments, each either a dense row-major
// block or rows gathered from a table through an int32 index (token -> embedding row).  Each workgroup owns
// 32 output columns and splits K over its waves; epilogues fuse bias, an additive term, the per-class
// conditioning and the tanh*sigmoid gate.
// ------------------------------------------------------------------------------------------------
struct SkinnySeg {
    const float *base;   // dense: row m at base + (m >> row_shift) * row_stride (+ col offset baked in base); null = zero rows
    const int *gidx;     // gather: row = base + gidx[m * gidx_stride] * row_stride ; negative index -> zero row
    long row_stride;
    long gidx_stride;
    int row_shift;
    int len;             // multiple of 8
    int tiled_w;         // 0 = row-major; else the buffer is TILED with this row width (see SkinnyParams::w_tiled)
};

enum { EPI_LINEAR = 0, EPI_GATE = 1 };
constexpr int SKINNY_MAX_SEG = 3;
constexpr int SKINNY_MAX_PROBLEMS = 6;

struct SkinnyParams {
    int M, N;            // N = number of weight rows (EPI_GATE: 2*gateD per group)
    int nseg, Ktot;
    SkinnySeg seg[SKINNY_MAX_SEG];
    const float *W;      // weight row n at W + n*ldw (K-contiguous, Ktot floats used)
    long ldw;
    const float *bias;   // [N] or null
    const float *add1;   // optional: add1[(m >> add1_shift) * add1_stride + n]
    long add1_stride;
    int add1_shift;
    const float *add2;   // optional second additive term, same indexing scheme
    long add2_stride;
    int add2_shift;
    const float *add3;   // optional third additive term: add3[m * add3_stride + n]
    long add3_stride;
    const float *clsrow; // optional per-row conditioning added AFTER `pre` is stored: clsrow[m * cls_ld + (n % cls_ld)]
    int cls_ld;
    int epi;             // EPI_LINEAR / EPI_GATE
    int relu;            // EPI_LINEAR only
    int gateD;           // EPI_GATE: channels per gate half (columns n and n+gateD pair up inside each 2*gateD group)
    float *out;          // EPI_LINEAR: [M][out_stride] N columns; EPI_GATE: [M][out_stride], N/2 columns
    long out_stride;
    float *pre;          // EPI_GATE optional: pre-activation (acc + bias + add1 + add2, without clsrow) [M][pre_stride]
    long pre_stride;
    int grid_x, grid_y;  // filled by the launcher
    // ---- tiled operand layouts (descriptor kernel only) ----
    // A wave's MFMA operand fragment is 16 rows x 16 k: lane (i = lane & 15, g = lane >> 4) holds k = 16 q + 4 g .. + 3 of row i.
    // Row-major, one wave load touches 16 rows x 64 B — measured at 14 B/clk/CU from a warm L2 against 34-42 B/clk for a
    // contiguous 1 KB (tools/fetch_rate.cpp).  A TILED buffer stores each such fragment contiguously, in lane order:
    //   float index of element (m, k) of a [rows][W] array = (((m >> 4) * (W >> 4) + (k >> 4)) << 8) + (((m & 15) + 16 * ((k & 15) >> 2)) << 2) + (k & 3)
    // rows padded to a multiple of 16, W a power of two >= 16.  Weights: tile t (16 output columns, in the epilogue's
    // column order), q-step q at ((t * (K / 16) + q) << 8), same lane order.
    int w_tiled;         // 0: W is row-major [N][ldw]; else W is the tiled copy and this is K / 16
    int out_tiled_w;     // 0 or the row width of the tiled view `out` is written in (element (row, col) = linear row * out_stride + col)
    int pre_tiled_w;     // same for `pre`
    int add1_tiled_w;    // same for reading `add1`
};

// up to SKINNY_MAX_PROBLEMS INDEPENDENT problems share one launch (blockIdx.z): one kernel boundary on the dependent chain
struct SkinnyBatch {
    SkinnyParams p[SKINNY_MAX_PROBLEMS];
};

// host-side: the tiled copy of a weight matrix W [N][ldw] (K columns used) for epilogue `epi` (column order of EPI_GATE tiles:
// 8 "tanh" channels followed by their 8 "sigmoid" partners); out has ceil(N/16) * (K/16) * 256 floats
void skinny_tile_weights(const float *W, int N, int K, long ldw, int epi, int gateD, float *out);
// false if an environment knob forces the generic kernels (which do not read tiled operands)
bool skinny_descriptor_kernel_enabled();
hipError_t launch_skinny_gemm(const SkinnyParams &p, hipStream_t stream);
hipError_t launch_skinny_batch(const SkinnyParams *const *ps, int n, hipStream_t stream);
// allocates the per-device zero buffer the fast skinny kernel substitutes for absent operands (call once per device,
// outside stream capture; without it the generic kernels are used)
hipError_t skinny_init(int device);
// the per-device zero buffer (256 KB) allocated by skinny_init, or nullptr
const float *skinny_zero_buffer(int device);
// TS_SKINNY_TRACE=1 instrumentation: records of 6 u64 {t_entry, t_desc, t_mfma_done, t_reduced, t_end, cnt<<32|workgroups}
int skinny_trace_read(unsigned long long *out, int max_records);

// ------------------------------------------------------------------------------------------------
// VQ / sampling / glue kernels
// ------------------------------------------------------------------------------------------------
// idx[m] = argmin_j (|x_m|^2 + |e_j|^2) - 2 x_m.e_j   (ties -> lowest j); also writes int32 copy if idx32 != null
hipError_t launch_vq_argmin(const float *x, int ldx, int M, const float *codebook, const float *code_sq, int ncode,
                            int dim, int64_t *idx, long idx_stride, hipStream_t stream);
// code_sq[j] = sum_c e[j][c]^2
hipError_t launch_row_sqnorm(const float *e, int n, int dim, float *out, hipStream_t stream);
// out[m][0..width) = table[idx[m*idx_stride]][0..width); an index outside [0, nrows) gives a row of NaNs
hipError_t launch_gather_rows(const float *table, int ld_table, int nrows, const int64_t *idx, long idx_stride, int M,
                              int width, float *out, int ldo, hipStream_t stream);
// dst[m][0..cpad) = src[m][0..c) then zeros
hipError_t launch_pad_rows(const float *src, int lds, int c, float *dst, int ldd, int cpad, long M, hipStream_t stream);
// (B,Tb,129) body/hand poses + (B,Tf,103) jaw/expression -> (B,Tf,265) full SMPL-X parameter rows (demo.py:207-229, part2full)
hipError_t launch_assemble_full(const float *body, const float *face, const float *lower_pose33, int B, int Tb, int Tf,
                                float *out, hipStream_t stream);
// int64 -> int32 (labels, teacher-forced codes)
hipError_t launch_i64_to_i32(const int64_t *src, int *dst, long n, hipStream_t stream);
// test aid: out[i] = gate_act(v[i], p[i])
hipError_t launch_gate_act(const float *v, const float *p, float *out, long n, hipStream_t stream);
// dst[0..2] = a, b, c, carried by the launch's own arguments (no host buffer has to outlive the call)
hipError_t launch_set_words3(uint64_t *dst, uint64_t a, uint64_t b, uint64_t c, hipStream_t stream);
// measurement aid: n records of (wall ticks since start, wall ticks of the window, shader cycles of the window), 100 MHz wall clock
hipError_t launch_clock_sample(unsigned long long *out, int n, unsigned long long window_ticks, hipStream_t stream);

// ------------------------------------------------------------------------------------------------
// Test / A-B levers (INTEGRATION.md has the table).  Every TS_* environment variable the library knows is read ONCE, by the
// first ts_ctx_create of the process (api.cpp), into this struct; nothing on a launch path calls getenv.  All paths behind
// them are bit-identical on a clip's codes (tests/test_gpu_parity.py::test_alternate_kernel_paths).
// ------------------------------------------------------------------------------------------------
struct Knobs {
    bool conv_bands = true;     // TS_CONV_BANDS=0: big conv layers as one plain grid of 128 x 128 tiles
    bool vq_lds = true;         // TS_VQ_LDS=0: the codebook search reads code rows from L2 per thread instead of LDS-staged tiles (tests, A/B)
    int conv_ring = 9;          // TS_CONV_RING=0|1|3|8|9: single-problem layers that take 128 x 128 tiles on conv_gemm.hip (0) / forced onto the ring engine's 128 x 128 tile with 4 (1) or 8 (8) waves or its 96 x 128 tile (3) / (9, default) 128 x 128 on 8 waves or 96 x 128 by tile count
    bool w2v_moments = true;    // TS_W2V_MOMENTS=0: conv0's GroupNorm statistics from a pass that computes the convolution (512 channels) instead of from the input's second moments (A/B, tests)
    int conv_sk = 1;            // TS_CONV_SK=0: no stream-K band in the ring engine's plans (whole tiles only: the round-5 plans); 2: the band wherever a layer has a plan for one (A/B, tests)
    bool conv_taps48 = true;    // TS_CONV_TAPS48=0: the face generator's grouped positional conv as 64-channel windows on conv_gemm_f32's tiles instead of conv_taps48.hip (A/B, tests)
    bool conv_ring_paired = true;    // TS_CONV_RING_PAIRED=0: paired layers (two problems per launch: body + hands) on conv_gemm.hip's banded launch instead of the ring engine (A/B, tests)
    bool conv_deal = true;      // TS_CONV_DEAL=0: the ring engine's tiles as a plain (row tiles, column tiles) grid instead of dealt to the XCDs in operand-sharing blocks (A/B, tests)
    int split_xcd = 8;          // TS_SPLIT_XCD: column-group width of conv_gemm_split's XCD-aware tile order (0: plain 2-D tile grid)
    bool prof_log = false;      // TS_PROF_LOG=1: one stderr line per conv launch while ts_prof is enabled
    bool no_graph = false;      // TS_NO_GRAPH=1: PixelCNN launches go out eagerly
    int pix_defer_p = -1;       // TS_PIX_DEFER_P: -1 auto (<= 128 clips), 0 / 1 forced
    int skinny_v = 1;           // TS_SKINNY_V=0: generic chain kernels
    int skinny_nt = 16;         // TS_SKINNY_NT=32: 32-column generic kernel
    bool skinny_tiled = true;   // TS_SKINNY_TILED=0: row-major chain operands
    int wide_min = 160;         // TS_SKINNY_WIDE_MIN: workgroups from which a coalesced launch takes the wide kernel (0 never, 1 always)
    int skinny_shape = 0;       // TS_SKINNY_SHAPE=11|21|22|42: forced split-K tile shape
    int skinny_trace = 0;       // TS_SKINNY_TRACE=1: in-kernel clock stamps (tools/skinny_trace.py, tools/wide_trace.py)
    bool wide_pair = true;      // TS_SKINNY_WIDE_PAIR=0: the wide kernel's tiles enumerated column-major instead of (clip-block pair, column tile, block of the pair) (A/B, tests)
    int wide_ablate = 0;        // TS_SKINNY_WIDE_ABLATE=2|4 (trace builds): no loads / no MFMAs in the wide kernel
};
const Knobs &knobs();

// GELU(x) = x / 2 (1 + erf(x / sqrt 2)) for the conv epilogues of the face generator (reference: torch's exact-erf GELU behind the HF wav2vec2
// layers of nets/spg/wav2vec.py).  libm's erff is two branches (|x| < 1: an odd polynomial; else 1 - exp(-poly)) with expf's own range
// reduction: ~45 VALU instructions plus exec-mask juggling per element, and on the ring engine's 8-wave tiles it is NOT hidden — measured
// (tools/gelu_cost.py): +42 us on a 692 us FFN1, +92 us on a 1.04 ms feature-convolution shape.  Here both pieces of the same algorithm with
// the same coefficients run branch-free (v_cndmask) and the exponential is one v_exp_f32: ~22 instructions, |error| <= 9e-8 (1 ulp of erf)
// against float64 over [-6, 6] — the face's distance from the reference does not move (2e-6).
__device__ __forceinline__ float erf_fast(float x) {
    const float a = __builtin_fabsf(x), t = a * a;
    float p = __builtin_fmaf(t, -0.000561801774892956f, 0.004913816228508949f);
    p = __builtin_fmaf(t, p, -0.026707515120506287f);
    p = __builtin_fmaf(t, p, 0.11280010640621185f);
    p = __builtin_fmaf(t, p, -0.37612295150756836f);
    p = __builtin_fmaf(t, p, 0.12837910652160645f);
    const float r1 = __builtin_fmaf(a, p, a);                       // |x| < 1
    float q = __builtin_fmaf(a, 1.699881067906972e-05f, -0.00037867785431444645f);
    q = __builtin_fmaf(a, q, 0.003857815871015191f);
    q = __builtin_fmaf(a, q, -0.024181697517633438f);
    q = __builtin_fmaf(a, q, 0.10666826367378235f);
    q = __builtin_fmaf(a, q, 0.6349332928657532f);
    q = __builtin_fmaf(a, q, 0.12868940830230713f);
    q = __builtin_fmaf(a, q, a);
    const float r2 = 1.0f - __builtin_amdgcn_exp2f(-1.4426950408889634f * q);   // |x| >= 1; exp2 of a large negative number flushes to 0
    return __builtin_copysignf(a < 1.0f ? r1 : r2, x);
}
__device__ __forceinline__ float gelu_fast(float v) { return 0.5f * v * (1.0f + erf_fast(v * 0.70710678118654752440f)); }

// tanh(v) * sigmoid(p) of GatedActivation (gated_pixelcnn_v2.py:16-22) on the hardware exponential and reciprocal (v_exp_f32,
// v_rcp_f32: 1 ulp each): tanh(v) = 1 - 2 / (1 + e^(2v)), sigmoid(p) = 1 / (1 + e^(-p)) — 10 instructions per gate value where
// tanhf + expf + a division took ~45 (8 values per lane in the wide kernel's epilogue: half of its VALU instructions, on the
// dependent chain of 30 launches per code row).  Absolute error < 2e-7 (the gate is O(1)); saturates correctly (e^(2v) = inf -> 1,
// flushed to 0 -> -1).  The RELATIVE error of the tanh factor grows towards v = 0 (1 - 2 / (1 + e^(2v)) cancels: ~6e-8 / |v|), which
// is harmless where the value is used — it is added into O(1) sums by the next layer — and is measured, absolute and relative, by
// tests/test_gpu_parity.py::test_gate_activation_accuracy over the whole input range (ts_debug_gate_act).
// ONE definition for every chain kernel: a clip's bits must not depend on which kernel served its stage.
__device__ __forceinline__ float gate_act(float v, float p) {
    const float ev = __builtin_amdgcn_exp2f(v * 2.88539008177792681f);      // e^(2 v)
    const float ep = __builtin_amdgcn_exp2f(p * -1.44269504088896341f);     // e^(-p)
    const float th = fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + ev), 1.0f);
    return th * __builtin_amdgcn_rcpf(1.0f + ep);
}

// exp(x) for x <= 0 from fp32 multiplies and adds ONLY (no fused multiply-add, no hardware transcendental): every operation is
// one IEEE round-to-nearest fp32 operation, so `oracle/talkshow_oracle.py::det_expf` reproduces it bit for bit on any host and
// the inverse-CDF draw of a given uniform is the same index on the device and in the oracle — not "within one slot".
// Cody-Waite reduction by ln 2 (n * LN2_HI is exact), degree-5 polynomial on [-ln2/2, ln2/2] (Cephes' coefficients), 2^n by
// exponent bits; relative error < 2 ulp.  Arguments below -86 give 0 (2^-124: everything stays a normal number).
__device__ inline float det_expf(float x) {
#pragma clang fp contract(off)
    if (x < -86.0f) return 0.0f;
    const float n = rintf(x * 1.44269504088896341f);
    float r = x - n * 0.693145751953125f;
    r = r - n * 1.42860682030941723212e-6f;
    float q = 1.9875691500e-4f;
    q = q * r + 1.3981999507e-3f;
    q = q * r + 8.3334519073e-3f;
    q = q * r + 4.1665795894e-2f;
    q = q * r + 1.6666665459e-1f;
    q = q * r + 5.0000001201e-1f;
    float y = q * (r * r) + r;
    y = y + 1.0f;
    return y * __int_as_float(((int)n + 127) << 23);
}

struct SampleParams {
    const float *logits;   // row b at logits + b * (logit_stride ? logit_stride : V)
    long logit_stride;     // 0 = V (contiguous rows)
    int B, V;
    int mode;              // TS_SAMPLE_* ; TEACHER_FORCED copies
    const float *uniforms; // element for clip b at uniforms[b * u_stride]
    long u_stride;
    uint64_t seed;
    int64_t clip_index0;
    const uint64_t *dyn;   // optional device words {seed, clip_index0, position base}: the first two override the fields above, the
                           // third is added to `position` (graph replay: a captured chunk serves every chunk of a session)
    uint32_t position;     // Philox counter word: linear position (row*2 + col)
    int *tok32;            // token for clip b written to tok32[b * tok_stride]
    long tok_stride;
    int64_t *codes;        // same position in the int64 output, codes[b * code_stride]
    long code_stride;
    float *logits_copy;    // optional: logits_copy[b * copy_stride + v] = logits[b][v]
    long copy_stride;
};
hipError_t launch_sample(const SampleParams &p, hipStream_t stream);

// ------------------------------------------------------------------------------------------------
// face generator kernels (face.hip)
// ------------------------------------------------------------------------------------------------
hipError_t launch_w2v_conv0(const float *wav, int B, int N, int L0, const float *w, const float *gamma, const float *beta,
                            double2 *part, float2 *stats, float *out, int C, hipStream_t s);
hipError_t launch_layernorm_rows(const float *x, int ldx, long M, int C, const float *gamma, const float *beta,
                                 const float *post_res, int ldr, int relu, float *out, int ldo, hipStream_t s);
hipError_t launch_lerp_ln(const float *x, int B, int Lin, int T, const float *gamma, const float *beta, float *out,
                          hipStream_t s);
hipError_t launch_attention(const float *qkv, int B, int T, int HID, int heads, float scale, float *out, hipStream_t s);
hipError_t launch_fill_id(const float *id, int nc, const float *w, const float *bias, int nj, float *x, int ld, int col0,
                          int B, int T, hipStream_t s);

}  // namespace ts
