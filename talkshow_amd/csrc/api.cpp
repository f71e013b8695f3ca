// Whole-wrapper entry points and the single-operator entry points used by the kernel-level parity tests.
#include <cstring>
#include <vector>
#include "host_common.h"
#include "conv_tile.h"
#include "skinny_desc.h"

using namespace ts;

namespace {
struct BodyWork {
    DevBuf feat;
    DevBuf lat[2];
};
// scratch between the stages of ts_body_pixel_infer (audio feature map, split latents), one set per stream
BodyWork &body_work(hipStream_t s) {
    static StreamWorks<BodyWork> works;
    return works.get(s);
}
}  // namespace

namespace ts {
const Knobs &knobs() {
    static const Knobs k = [] {
        Knobs v;
        auto num = [](const char *name, int dflt) { const char *e = std::getenv(name); return e && e[0] ? std::atoi(e) : dflt; };
        v.conv_bands = num("TS_CONV_BANDS", 1) != 0;
        v.conv_ring = num("TS_CONV_RING", 9);
        v.conv_deal = num("TS_CONV_DEAL", 1) != 0;
        v.conv_ring_paired = num("TS_CONV_RING_PAIRED", 1) != 0;
        v.conv_taps48 = num("TS_CONV_TAPS48", 1) != 0;
        v.conv_sk = num("TS_CONV_SK", 1);
        v.w2v_moments = num("TS_W2V_MOMENTS", 1) != 0;
        v.vq_lds = num("TS_VQ_LDS", 1) != 0;
        v.split_xcd = num("TS_SPLIT_XCD", 8);
        v.prof_log = num("TS_PROF_LOG", 0) != 0;
        if (const char *e = std::getenv("TS_NO_GRAPH")) v.no_graph = e[0] && e[0] != '0';
        v.pix_defer_p = num("TS_PIX_DEFER_P", -1);
        v.skinny_v = num("TS_SKINNY_V", 1);
        v.skinny_nt = num("TS_SKINNY_NT", 16);
        v.skinny_tiled = num("TS_SKINNY_TILED", 1) != 0;
        v.wide_min = num("TS_SKINNY_WIDE_MIN", 160);
        v.skinny_shape = num("TS_SKINNY_SHAPE", 0);
        v.skinny_trace = num("TS_SKINNY_TRACE", 0);
        v.wide_ablate = num("TS_SKINNY_WIDE_ABLATE", 0);
        v.wide_pair = num("TS_SKINNY_WIDE_PAIR", 1) != 0;
        return v;
    }();
    return k;
}
}  // namespace ts

extern "C" {

// Streams for pipelining independent batches.  Created back to back so that ROCclr's round-robin hands consecutive
// streams distinct hardware queues (GPU_MAX_HW_QUEUES); hosts without a stream pool of their own use these.
int ts_stream_create(ts_ctx *ctx, void **out) {
    if (!ctx || !out) return fail("ts_stream_create: null argument");
    TS_HIP(hipSetDevice(ctx->device));
    hipStream_t s = nullptr;
    TS_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out = s;
    return 0;
}
// A stream whose kernels only run on compute units [cu_first, cu_first + cu_count) of the device's CU-mask index space
// (consecutive mask bits are spread round-robin over the 8 XCDs).  Used to keep the latency-bound PixelCNN chain and the
// MFMA-bound conv stacks of different batches off each other's CUs.
int ts_stream_create_cus(ts_ctx *ctx, int cu_first, int cu_count, void **out) {
    if (!ctx || !out) return fail("ts_stream_create_cus: null argument");
    TS_HIP(hipSetDevice(ctx->device));
    hipDeviceProp_t prop;
    TS_HIP(hipGetDeviceProperties(&prop, ctx->device));
    const int ncu = prop.multiProcessorCount;
    if (cu_first < 0 || cu_count < 1 || cu_first + cu_count > ncu) return fail("ts_stream_create_cus: CU range outside the device");
    std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
    for (int i = cu_first; i < cu_first + cu_count; ++i) mask[i / 32] |= 1u << (i % 32);
    hipStream_t s = nullptr;
    TS_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
    *out = s;
    return 0;
}
// Output assembly after both generators (scripts/demo.py:207-229, data_utils/lower_body.py:68-87)
int ts_assemble_full(ts_ctx *ctx, const float *body, int Tb, const float *face, int Tf, int B, const float *lower_pose33,
                     float *out, void *stream) {
    if (!ctx || !body || !face || !lower_pose33 || !out) return fail("ts_assemble_full: null argument");
    if (B < 1 || Tb < 1 || Tf < 1) return fail("ts_assemble_full: empty input");
    MiscScope ms(ctx, (hipStream_t)stream);
    TS_HIP(launch_assemble_full(body, face, lower_pose33, B, Tb, Tf, out, (hipStream_t)stream));
    return 0;
}

// tuning aid (TS_SKINNY_TRACE=1): in-kernel wall-clock stamps of the PixelCNN chain kernel, 6 u64 per record
int ts_debug_skinny_trace(unsigned long long *out, int max_records) {
    if (!out) return -1;
    return ts::skinny_trace_read(out, max_records);
}
// measurement aid: a one-wave kernel on `stream` that records the shader clock the chip runs at, every window_us, n times
int ts_debug_clock_sample(unsigned long long *dev_out, int n, int window_us, void *stream) {
    if (!dev_out || n < 1 || window_us < 1) return fail("ts_debug_clock_sample: bad argument");
    TS_HIP(ts::launch_clock_sample(dev_out, n, (unsigned long long)window_us * 100, (hipStream_t)stream));
    return 0;
}
// Host-only (no GPU): the launch plan of an (M x N, `groups` problems) conv layer — out4 = {row blocks of 128 x 128 tiles, row blocks of
// 64 x 128 tiles, workgroups of the first band, workgroups}; returns 1 if the layer is launched in two bands, 0 for a plain grid
int ts_debug_conv_bands(int M, int N, int groups, int *out4) {
    if (M < 1 || N < 1 || groups < 1 || groups > 4 || !out4) return fail("ts_debug_conv_bands: bad argument") ? -1 : -1;
    ts::ConvParams p;
    std::memset(&p, 0, sizeof(p));
    p.M = M;
    p.N = N;
    p.ngroups = groups;
    ts::ConvBands bd{};
    const bool banded = ts::conv_gemm_band_plan(p, bd);
    out4[0] = bd.mt_big; out4[1] = bd.mt_small; out4[2] = bd.first_small; out4[3] = bd.total;
    return banded ? 1 : 0;
}
// Host-only (no GPU): tile (out2 = {row tile, column tile}) that workgroup `bid` of conv_gemm_split's 1-D grid works on for an MT x NT
// tile grid and column groups of `gw` tiles; returns 1, 0 if that workgroup has no tile, -1 on a bad argument
int ts_debug_split_tile(int bid, int MT, int NT, int gw, int *out2) {
    if (bid < 0 || MT < 1 || NT < 1 || gw < 1 || !out2) return fail("ts_debug_split_tile: bad argument") ? -1 : -1;
    return ts::split_tile_of(bid, MT, NT, gw, out2[0], out2[1]) ? 1 : 0;
}
int ts_debug_tile_weights(const float *W, int N, int K, long ldw, int epi, int gateD, float *out) {
    if (!W || !out || N < 1 || K < 16 || K % 16 || ldw < K) return fail("ts_debug_tile_weights: bad argument");
    if (epi == ts::EPI_GATE && (gateD < 8 || gateD % 8 || N % (2 * gateD))) return fail("ts_debug_tile_weights: gate tiles need gateD % 8 == 0 and N % (2 gateD) == 0");
    ts::skinny_tile_weights(W, N, K, ldw, epi, gateD, out);
    return 0;
}
int ts_stream_destroy(ts_ctx *ctx, void *stream) {
    if (!ctx) return fail("ts_stream_destroy: null ctx");
    TS_HIP(hipStreamSynchronize((hipStream_t)stream));
    drop_stream_everywhere((hipStream_t)stream);   // scratch arenas and captured graphs keyed by this handle
    TS_HIP(hipStreamDestroy((hipStream_t)stream));
    return 0;
}

// s2g_body_pixel.TrainWrapper.infer_on_audio, device part (nets/smplx_body_pixel.py:272-285)
int ts_body_pixel_infer(ts_convnet *ae, ts_pixelcnn *pix, ts_vqvae *vb, ts_vqvae *vh, const float *mfcc,
                        const int64_t *ids, int B, int T, int mode, const float *uniforms, uint64_t seed, int64_t clip0,
                        int64_t *codes, float *poses, void *stream) {
    if (!ae || !pix || !vb || !vh || !mfcc || !ids || !codes || !poses) return fail("ts_body_pixel_infer: null argument");
    hipStream_t s = (hipStream_t)stream;
    const int H = (T / 2) / 2;
    if (H < 1) return fail("ts_body_pixel_infer: clip too short");
    const int aud_dim = convnet_hidden(ae), body_dim = vqvae_in_dim(vb), hand_dim = vqvae_in_dim(vh);
    BodyWork &w = body_work(s);
    TS_TRY(w.feat.ensure((size_t)B * H * aud_dim * sizeof(float)));
    TS_TRY(ts_audioenc_forward(ae, mfcc, B, T, w.feat.f(), s));
    TS_TRY(ts_pixelcnn_generate(pix, ids, w.feat.f(), B, H, mode, uniforms, seed, clip0, codes, nullptr, nullptr, nullptr,
                                0, s));
    // body_latents = latents[..., 0]; hand_latents = latents[..., 1]  (:279-280)
    for (int k = 0; k < 2; ++k) {
        TS_TRY(w.lat[k].ensure((size_t)B * H * sizeof(int64_t)));
        TS_HIP(hipMemcpy2DAsync(w.lat[k].p, sizeof(int64_t), codes + k, 2 * sizeof(int64_t), sizeof(int64_t),
                                (size_t)B * H, hipMemcpyDeviceToDevice, s));
    }
    (void)body_dim;
    (void)hand_dim;
    return ts_vqvae_decode_pair(vb, vh, static_cast<int64_t *>(w.lat[0].p), static_cast<int64_t *>(w.lat[1].p), B, H, poses, s);
}

int ts_op_conv1d(ts_ctx *ctx, const float *x, int B, int Lin, int Cin, const float *w, const float *bias, int Cout,
                 int K, int stride, int pad, int transposed, int act, float *out, void *stream) {
    if (!ctx || !x || !w || !out) return fail("ts_op_conv1d: null argument");
    hipStream_t s = (hipStream_t)stream;
    int kind;
    if (!transposed && stride == 1 && (K == 1 || K == 3) && pad == (K - 1) / 2) kind = 0;
    else if (!transposed && stride == 2 && K == 4 && pad == 1) kind = 1;
    else if (transposed && stride == 2 && K == 4 && pad == 1) kind = 2;
    else return fail("ts_op_conv1d: unsupported geometry");
    std::vector<float> zb(Cout, 0.f);
    ts_tensor t[2];
    t[0].name = "op.weight";
    t[0].data = w;
    t[0].ndim = 3;
    t[0].shape[0] = transposed ? Cin : Cout;
    t[0].shape[1] = transposed ? Cout : Cin;
    t[0].shape[2] = K;
    t[1].name = "op.bias";
    t[1].data = bias ? bias : zb.data();
    t[1].ndim = 1;
    t[1].shape[0] = Cout;
    StateDict sd(t, 2);
    ConvLayer L;
    TS_TRY(pack_conv_layer(sd, "op", "", "", kind, K, Cin, Cout, act, &L));
    DevBuf xin;
    const float *xp = x;
    int ldx = Cin;
    if (Cin % 32 != 0) {
        TS_TRY(xin.ensure((size_t)B * Lin * L.cin_pad * sizeof(float)));
        TS_HIP(launch_pad_rows(x, Cin, Cin, xin.f(), L.cin_pad, L.cin_pad, (long)B * Lin, s));
        xp = xin.f();
        ldx = L.cin_pad;
    }
    ConvParams p;
    conv_layer_params(L, xp, ldx, B, Lin, nullptr, 0, out, Cout, 0, Cout, &p);
    TS_TRY(run_conv(ctx, p, 0, s));
    TS_HIP(hipStreamSynchronize(s));   // temporaries die with this frame
    return 0;
}

// one warm-up launch, then `iters` launches of the layer between two HIP events on `s`: *ms_out = mean launch duration (ms)
static int time_conv_launches(const ts::ConvParams &p, int tile, int iters, float *ms_out, hipStream_t s) {
    hipEvent_t a, b;
    TS_HIP(hipEventCreate(&a));
    TS_HIP(hipEventCreate(&b));
    TS_HIP(ts::launch_conv_gemm(p, tile, s));
    TS_HIP(hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) TS_HIP(ts::launch_conv_gemm(p, tile, s));
    TS_HIP(hipEventRecord(b, s));
    TS_HIP(hipEventSynchronize(b));
    float ms = 0.f;
    TS_HIP(hipEventElapsedTime(&ms, a, b));
    if (ms_out) *ms_out = ms / (iters > 0 ? iters : 1);
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    return 0;
}

// Tuning / roofline entry (not part of the drop-in surface): a stride-1 conv layer (K = 1 or 3, Cin % 32 == 0) with
// weights ALREADY packed on the device as [round128(Cout)][K*Cin] (tap-major), launched `iters` times between two HIP
// events on `stream` with a chosen tile shape (0 = the heuristic used in production).  ms_out = mean launch duration.
int ts_op_conv1d_timed(ts_ctx *ctx, const float *x, int B, int Lin, int Cin, const float *w_packed_dev,
                       const float *bias_dev, int Cout, int K, int tile, int iters, float *out, float *ms_out,
                       void *stream) {
    if (!ctx || !x || !w_packed_dev || !out) return fail("ts_op_conv1d_timed: null argument");
    if (Cin % 32 || (K != 1 && K != 3)) return fail("ts_op_conv1d_timed: unsupported geometry");
    hipStream_t s = (hipStream_t)stream;
    ConvParams p;
    std::memset(&p, 0, sizeof(p));
    p.M = B * Lin;
    p.Lout = p.Lin = Lin;
    p.stride = 1;
    p.ldx = Cin;
    p.ldo = Cout;
    p.N = Cout;
    p.Ktot = K * Cin;
    p.act = 1;
    p.ngroups = 1;
    p.g[0].x = x;
    p.g[0].w = w_packed_dev;
    p.g[0].bias = bias_dev;
    p.g[0].out = out;
    p.g[0].nseg = K;
    for (int k = 0; k < K; ++k) p.g[0].seg[k] = ConvSeg{K == 1 ? 0 : k - 1, 0, Cin};
    return time_conv_launches(p, tile, iters, ms_out, s);
}

// the same for a strided convolution without padding (the wav2vec2 feature convolutions: out[t] = sum_k W_k x[stride t + k]);
// out: (B, (Lin - K) / stride + 1, Cout)
int ts_op_conv1d_strided_timed(ts_ctx *ctx, const float *x, int B, int Lin, int Cin, const float *w_packed_dev,
                               const float *bias_dev, int Cout, int K, int stride, int tile, int iters, float *out,
                               float *ms_out, void *stream) {
    if (!ctx || !x || !w_packed_dev || !out) return fail("ts_op_conv1d_strided_timed: null argument");
    if (Cin % 32 || K < 1 || K > 4 || stride < 1 || Lin < K) return fail("ts_op_conv1d_strided_timed: unsupported geometry");
    hipStream_t s = (hipStream_t)stream;
    ConvParams p;
    std::memset(&p, 0, sizeof(p));
    p.Lin = Lin;
    p.Lout = (Lin - K) / stride + 1;
    p.M = B * p.Lout;
    p.stride = stride;
    p.ldx = Cin;
    p.ldo = Cout;
    p.N = Cout;
    p.Ktot = K * Cin;
    p.act = 3;
    p.ngroups = 1;
    p.g[0].x = x;
    p.g[0].w = w_packed_dev;
    p.g[0].bias = bias_dev;
    p.g[0].out = out;
    p.g[0].nseg = K;
    for (int k = 0; k < K; ++k) p.g[0].seg[k] = ConvSeg{k, 0, Cin};
    return time_conv_launches(p, tile, iters, ms_out, s);
}

// grouped many-tap convolution, 48 channels per group in and out (conv_taps48.hip: the wav2vec2 positional conv): x, res, out
// (B, T, G * 48); w [G][48][ntap * 48] (tap-major, channels contiguous); bias [G * 48]; out = GELU(conv + bias) + res, taps
// -ntap / 2 .. ntap - ntap / 2 - 1, zero padding
int ts_op_conv_taps48_timed(ts_ctx *ctx, const float *x, int B, int T, int G, int ntap, const float *w, const float *bias,
                            const float *res, int iters, float *out, float *ms_out, void *stream) {
    if (!ctx || !x || !w || !out) return fail("ts_op_conv_taps48_timed: null argument");
    if (B < 1 || T < 1 || G < 1 || ntap < 1) return fail("ts_op_conv_taps48_timed: unsupported geometry");
    hipStream_t s = (hipStream_t)stream;
    ConvParams p;
    std::memset(&p, 0, sizeof(p));
    p.M = B * T;
    p.Lout = p.Lin = T;
    p.stride = 1;
    p.ldx = p.ldo = p.ldr = G * 48;
    p.N = 48;
    p.Ktot = ntap * 48;
    p.act = 3;
    p.res_after_act = 1;
    p.ngroups = p.zdiv = G;
    p.x_zs1 = p.o_zs1 = p.r_zs1 = p.b_zs1 = 48;
    p.w_zs1 = 48L * p.Ktot;
    p.g[0].x = x;
    p.g[0].w = w;
    p.g[0].bias = bias;
    p.g[0].res = res;
    p.g[0].out = out;
    p.g[0].nseg = 1;
    p.g[0].seg[0] = ConvSeg{-(ntap / 2), 0, 48, ntap};
    return time_conv_launches(p, 48, iters, ms_out, s);
}

int ts_debug_conv_ring_pick(int M, int N, int groups) {
    if (M < 1 || N < 1 || groups < 1 || groups > 4) return -1;
    ts::ConvParams p;
    std::memset(&p, 0, sizeof(p));
    p.M = M;
    p.N = N;
    p.ngroups = groups;
    ts::ConvBands bd{};
    const bool have = ts::conv_gemm_plan_bands(p, bd);
    const int pick = ts::conv_gemm_ring_pick(p, have ? &bd : nullptr, nullptr);
    return pick == 3 ? 96 : (pick == 7 ? 64 : 128);
}

int ts_debug_conv_sk_plan(int M, int N, int K, int groups, int *out6) {
    if (M < 1 || N < 1 || K < 32 || K % 32 || groups < 1 || groups > 4 || !out6) return -1;
    ts::ConvParams p;
    std::memset(&p, 0, sizeof(p));
    p.M = M;
    p.N = N;
    p.Ktot = K;
    p.ngroups = groups;
    for (int z = 0; z < groups; ++z) {
        p.g[z].nseg = 1;
        p.g[z].seg[0] = ts::ConvSeg{0, 0, K, 1};
    }
    ts::ConvSK sk{};
    if (!ts::conv_gemm_plan_sk_shape(p, sk)) return 0;
    ts::ConvBands bd{};
    const bool have = ts::conv_gemm_plan_bands(p, bd) && bd.mt_big >= 1;
    const int pick = ts::conv_gemm_ring_pick(p, have ? &bd : nullptr, &sk);
    const int o[6] = {sk.mt_dp, sk.mt_sk, sk.dp8, sk.wsk, sk.stages, pick};
    std::memcpy(out6, o, sizeof(o));
    return 1;
}

int ts_debug_conv_sk_supported(void) { return ts::conv_sk_supported() ? 1 : 0; }

int ts_debug_conv_sk_run(int band_tiles, int stages, int band_workgroups, int q, int *out4) {
    if (band_tiles < 8 || stages < 1 || band_workgroups < 8 || (band_workgroups & 7) || q < 0 || q >= band_workgroups || !out4) return -1;
    const ts::SkRuns R{band_tiles, stages, band_workgroups >> 3};
    const int c = q & 7, r = q >> 3;
    out4[0] = R.begin(c, r);
    out4[1] = R.begin(c, r + 1);
    out4[2] = c;
    out4[3] = R.run_of(c, out4[0] < out4[1] ? out4[0] : R.tlo(c) * stages);
    return 0;
}

int ts_debug_gate_act(const float *v_dev, const float *p_dev, float *out_dev, long n, void *stream) {
    if (!v_dev || !p_dev || !out_dev || n < 0) return fail("ts_debug_gate_act: bad argument");
    TS_HIP(ts::launch_gate_act(v_dev, p_dev, out_dev, n, (hipStream_t)stream));
    return 0;
}

int ts_op_vq_argmin(ts_ctx *ctx, const float *x, int M, const float *cb, int ncode, int dim, int64_t *idx, void *stream) {
    if (!ctx || !x || !cb || !idx) return fail("ts_op_vq_argmin: null argument");
    hipStream_t s = (hipStream_t)stream;
    DevBuf sq;
    TS_TRY(sq.ensure((size_t)ncode * sizeof(float)));
    TS_HIP(launch_row_sqnorm(cb, ncode, dim, sq.f(), s));
    TS_HIP(launch_vq_argmin(x, dim, M, cb, sq.f(), ncode, dim, idx, 1, s));
    TS_HIP(hipStreamSynchronize(s));
    return 0;
}

int ts_op_linear(ts_ctx *ctx, const float *x, int M, int K, const float *w, const float *bias, int N, int relu,
                 float *out, void *stream) {
    if (!ctx || !x || !w || !out) return fail("ts_op_linear: null argument");
    if (K % 8 != 0) return fail("ts_op_linear: K must be a multiple of 8");
    hipStream_t s = (hipStream_t)stream;
    DevBuf wd, bd;
    TS_TRY(wd.upload(w, (size_t)N * K * sizeof(float)));
    if (bias) TS_TRY(bd.upload(bias, (size_t)N * sizeof(float)));
    SkinnyParams q;
    std::memset(&q, 0, sizeof(q));
    q.M = M;
    q.N = N;
    q.nseg = 1;
    q.Ktot = K;
    q.seg[0].base = x;
    q.seg[0].row_stride = K;
    q.seg[0].len = K;
    q.W = wd.f();
    q.ldw = K;
    q.bias = bias ? bd.f() : nullptr;
    q.epi = EPI_LINEAR;
    q.relu = relu;
    q.out = out;
    q.out_stride = N;
    TS_TRY(run_skinny(ctx, q, s));
    TS_HIP(hipStreamSynchronize(s));
    return 0;
}

// Tuning entry (not part of the drop-in surface): `iters` DEPENDENT skinny_gemm launches (stage i reads stage i-1's
// output) captured in one hipGraph and replayed; *us_out = microseconds per launch.  M x K activations, N = K outputs
// (linear) or 2K (gate epilogue, so the chain closes on itself); `debug` is unused (kept for ABI stability).
int ts_debug_skinny_chain(ts_ctx *ctx, int M, int K, int gate, int iters, int debug, float *us_out) {
    if (!ctx || !us_out) return fail("ts_debug_skinny_chain: null argument");
    const int N = gate ? 2 * K : K;
    DevBuf w, bias, x0, x1, lab, cls;
    std::vector<float> hw((size_t)N * K), hb(N, 0.01f), hx((size_t)M * K, 0.5f), hc((size_t)M * N, 0.01f);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = ((int)(i * 2654435761u >> 16) % 2001 - 1000) * (1.0f / (1000.f * K));
    std::vector<int> hl(M, 1);
    TS_TRY(w.upload(hw.data(), hw.size() * 4));
    TS_TRY(bias.upload(hb.data(), hb.size() * 4));
    TS_TRY(x0.upload(hx.data(), hx.size() * 4));
    TS_TRY(x1.upload(hx.data(), hx.size() * 4));
    TS_TRY(lab.upload(hl.data(), hl.size() * 4));
    TS_TRY(cls.upload(hc.data(), hc.size() * 4));
    hipStream_t s;
    TS_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipGraph_t g;
    hipGraphExec_t ex;
    TS_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < iters; ++i) {
        SkinnyParams q;
        std::memset(&q, 0, sizeof(q));
        q.M = M;
        q.N = N;
        q.nseg = 1;
        q.Ktot = K;
        q.seg[0].base = (i & 1) ? x1.f() : x0.f();
        q.seg[0].row_stride = K;
        q.seg[0].len = K;
        q.W = w.f();
        q.ldw = K;
        q.bias = bias.f();
        q.epi = gate ? EPI_GATE : EPI_LINEAR;
        q.gateD = K;
        q.clsrow = gate ? cls.f() : nullptr;
        q.cls_ld = N;
        q.out = (i & 1) ? x0.f() : x1.f();
        q.out_stride = K;
        TS_HIP(launch_skinny_gemm(q, s));
    }
    TS_HIP(hipStreamEndCapture(s, &g));
    TS_HIP(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    hipEvent_t a, b;
    TS_HIP(hipEventCreate(&a));
    TS_HIP(hipEventCreate(&b));
    TS_HIP(hipGraphLaunch(ex, s));
    TS_HIP(hipEventRecord(a, s));
    TS_HIP(hipGraphLaunch(ex, s));
    TS_HIP(hipEventRecord(b, s));
    TS_HIP(hipEventSynchronize(b));
    float ms = 0.f;
    TS_HIP(hipEventElapsedTime(&ms, a, b));
    *us_out = ms * 1e3f / iters;
    (void)hipGraphExecDestroy(ex);
    (void)hipGraphDestroy(g);
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    (void)hipStreamDestroy(s);
    return 0;
}

int ts_op_sample(ts_ctx *ctx, const float *logits, int B, int V, int mode, const float *uniforms, int64_t *idx,
                 void *stream) {
    if (!ctx || !logits || !idx) return fail("ts_op_sample: null argument");
    if (mode != TS_SAMPLE_GREEDY && mode != TS_SAMPLE_UNIFORMS) return fail("ts_op_sample: bad mode");
    if (mode == TS_SAMPLE_UNIFORMS && !uniforms) return fail("ts_op_sample: uniforms required");
    hipStream_t s = (hipStream_t)stream;
    DevBuf tok;
    TS_TRY(tok.ensure((size_t)B * sizeof(int)));
    SampleParams sp;
    std::memset(&sp, 0, sizeof(sp));
    sp.logits = logits;
    sp.B = B;
    sp.V = V;
    sp.mode = mode;
    sp.uniforms = uniforms;
    sp.u_stride = 1;
    sp.tok32 = tok.i();
    sp.tok_stride = 1;
    sp.codes = idx;
    sp.code_stride = 1;
    TS_HIP(launch_sample(sp, s));
    TS_HIP(hipStreamSynchronize(s));
    return 0;
}

int ts_op_sample_philox(ts_ctx *ctx, const float *logits, int B, int V, uint64_t seed, int64_t clip_index0, uint32_t position,
                        int64_t *idx, void *stream) {
    if (!ctx || !logits || !idx) return fail("ts_op_sample_philox: null argument");
    hipStream_t s = (hipStream_t)stream;
    DevBuf tok;
    TS_TRY(tok.ensure((size_t)B * sizeof(int)));
    SampleParams sp;
    std::memset(&sp, 0, sizeof(sp));
    sp.logits = logits;
    sp.B = B;
    sp.V = V;
    sp.mode = TS_SAMPLE_PHILOX;
    sp.seed = seed;
    sp.clip_index0 = clip_index0;
    sp.position = position;
    sp.tok32 = tok.i();
    sp.tok_stride = 1;
    sp.codes = idx;
    sp.code_stride = 1;
    TS_HIP(launch_sample(sp, s));
    TS_HIP(hipStreamSynchronize(s));
    return 0;
}

}  // extern "C"
