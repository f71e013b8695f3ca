// Face generator: s2g_face.Generator.forward (nets/spg/s2g_face.py:196-224) over the wav2vec2-base encoder as the
// reference subclasses it (nets/spg/wav2vec.py:76-143), as a launch plan on conv_gemm_f32 + the kernels of face.hip.
//
//   wav (B,N) --conv0+GroupNorm+GELU--> (B,L0,512) --6 strided convs (GELU)--> (B,L6,512)
//     --lerp to `frames` + LayerNorm--> Linear 512->768 --(+ GELU(grouped k128 conv)) LayerNorm-->
//     12 x { QKV GEMM, QK^T (batched GEMM), softmax, P.V (batched GEMM), out-proj (+res), LN, FFN1 GELU, FFN2 (+res), LN }
//     --Linear 768->256 | id_mlp(id) concat--> 3 x {conv k3, LN, +res, ReLU} --> two heads of 3 x {conv k3, LN, ReLU}
//     --1x1 convs--> (B,frames,3+100)
#include <algorithm>

#include "host_common.h"

using namespace ts;

namespace {

struct LNp {
    DevBuf g, b;
};

struct EncLayer {
    ConvLayer qkv, outp, ff1, ff2;
    LNp ln1, ln2;
};

int up(DevBuf &d, const float *p, size_t n) { return d.upload(p, n * sizeof(float)); }

int load_ln(const StateDict &sd, const std::string &k, int c, LNp *o) {
    const float *g = sd.get(k + ".weight", {c}), *b = sd.get(k + ".bias", {c});
    if (!g || !b) return 1;
    TS_TRY(up(o->g, g, c));
    return up(o->b, b, c);
}

int load_linear(const StateDict &sd, const std::string &k, int n, int kk, int act, ConvLayer *L) {
    const float *w = sd.get(k + ".weight", {n, kk}), *b = sd.get(k + ".bias", {n});
    if (!w || !b) return 1;
    TS_TRY(pack_linear_layer(w, kk, b, n, kk, L));
    L->act = act;
    return 0;
}

int load_conv3(const StateDict &sd, const std::string &k, int cout, int cin, int K, ConvLayer *L) {
    const float *w = sd.get(k + ".weight", {cout, cin, K}), *b = sd.get(k + ".bias", {cout});
    if (!w || !b) return 1;
    const int taps3[3] = {-1, 0, 1}, taps1[1] = {0};
    return pack_conv_raw(w, b, cout, cin, K, K == 3 ? taps3 : taps1, 0, L);
}

// generic ConvParams for a packed layer: rows (b, t) with t < Lout, input row t*stride + d
void params_for(const ConvLayer &L, const float *x, int ldx, int B, int Lin, int Lout, int stride, const float *res, int ldr,
                float *out, int ldo, int col0, int nstore, int act, ConvParams *p, bool planes_ok = false) {
    std::memset(p, 0, sizeof(*p));
    p->M = B * Lout;
    p->Lout = Lout;
    p->Lin = Lin;
    p->stride = stride;
    p->ldx = ldx;
    p->ldo = ldo;
    p->ldr = ldr;
    p->N = nstore;
    p->Ktot = L.ktot;
    p->act = act;
    p->ngroups = 1;
    ConvGroup &G = p->g[0];
    G.x = x;
    G.w = L.w.f();
    if (planes_ok && L.wp.p) {   // x3 plan: the layer's weights as plane images (ts_face_set_arith)
        G.w = L.wp.f();
        p->w_planes = 1;
    }
    G.bias = L.bias.f();
    G.res = res;
    G.out = out;
    G.out_col0 = col0;
    G.nseg = L.nseg;
    for (int s = 0; s < L.nseg; ++s) G.seg[s] = L.segs[0][s];
}

}  // namespace

struct ts_face {
    ts_ctx *ctx = nullptr;
    int NL = 12, HID = 768, HEADS = 12, FFN = 3072, C0 = 512, NCLS = 4, POSK = 128, POSG = 16;
    int JAW = 3, CIN = 320;   // identity=False (num_classes 0, the reference's convert_to_6d form): no id channels (CIN 256), 6 jaw values
    DevBuf c0_w, c0_g, c0_b;
    ConvLayer fc[6];
    int fc_k[6] = {3, 3, 3, 3, 2, 2};
    LNp fp_ln;
    ConvLayer fp_proj;
    DevBuf pos_w, pos_b, pos_wp;
    DevBuf pos_wd;   // the same weights unpadded, [group][48][taps * 48] (conv_taps48.hip); empty unless the group width is 48
    int pos_npad = 128, pos_ktot = 0;
    LNp enc_ln;
    std::vector<std::unique_ptr<EncLayer>> layers;
    ConvLayer afm;
    DevBuf id_w, id_b;
    ConvLayer fn[3], fn0res, dec[2][3], fin[2];
    LNp fn_ln[3], dec_ln[2][3];

    int split_planes = 0;   // 0: fp32 MFMA (default, the parity path); 2 / 3: opt-in split-bf16 GEMMs (ts_face_set_arith)

    struct Work {
        DevBuf A, Bf, part, stats, X512, H, H2, TMP, QKV, ATT, FF, X320, Y1, Y2, R, D1, D2;
    };
    StreamWorks<Work> works;
    Work &work(hipStream_t s) { return works.get(s); }
};

namespace ts {
int face_hidden(const ts_face *f) { return f->HID; }
}

extern "C" {

int ts_face_create(ts_ctx *ctx, const ts_tensor *sd_, int n, int n_layers, int num_classes, ts_face **out) {
    if (!ctx || !sd_ || !out) return fail("ts_face_create: null argument");
    TS_HIP(hipSetDevice(ctx->device));
    StateDict sd(sd_, n);
    std::unique_ptr<ts_face> f(new ts_face());
    f->ctx = ctx;
    f->NL = n_layers;
    if (num_classes < 0) return fail("ts_face_create: num_classes < 0");
    f->NCLS = num_classes;
    // Generator(identity=False) (s2g_face.py:142-171, built by smplx_face.py:37-45 when convert_to_6d is set): AudioEncoder without
    // id_mlp (first_net takes the 256 audio channels alone), jaw head each_dim[0] = 6 wide (one joint in the 6-D rotation form)
    f->JAW = num_classes > 0 ? 3 : 6;
    f->CIN = num_classes > 0 ? 320 : 256;
    const int C0 = f->C0, HID = f->HID, FFN = f->FFN;
    const std::string p = "audio_encoder.";
    // ---- feature extractor ----
    const float *w0 = sd.get(p + "feature_extractor.conv_layers.0.conv.weight", {C0, 1, 10});
    const float *g0 = sd.get(p + "feature_extractor.conv_layers.0.layer_norm.weight", {C0});
    const float *b0 = sd.get(p + "feature_extractor.conv_layers.0.layer_norm.bias", {C0});
    if (!w0 || !g0 || !b0) return 1;
    TS_TRY(up(f->c0_w, w0, (size_t)C0 * 10));
    TS_TRY(up(f->c0_g, g0, C0));
    TS_TRY(up(f->c0_b, b0, C0));
    for (int i = 0; i < 6; ++i) {
        const int K = f->fc_k[i];
        const float *w = sd.get(p + "feature_extractor.conv_layers." + std::to_string(i + 1) + ".conv.weight", {C0, C0, K});
        if (!w) return 1;
        const int taps[3] = {0, 1, 2};   // stride 2, no padding: out[t] = sum_k W_k x[2t + k]
        TS_TRY(pack_conv_raw(w, nullptr, C0, C0, K, taps, 3, &f->fc[i]));
    }
    TS_TRY(load_ln(sd, p + "feature_projection.layer_norm", C0, &f->fp_ln));
    TS_TRY(load_linear(sd, p + "feature_projection.projection", HID, C0, 0, &f->fp_proj));
    // ---- positional conv: weight_norm(dim=2) folded; grouped weights padded 48 -> 64 channels per tap ----
    {
        const int G = f->POSG, K = f->POSK, cg = HID / G;
        const std::string q = p + "encoder.pos_conv_embed.conv";
        const float *g = nullptr, *v = nullptr;
        if (sd.m.count(q + ".parametrizations.weight.original0")) {
            g = sd.get(q + ".parametrizations.weight.original0", {1, 1, K});
            v = sd.get(q + ".parametrizations.weight.original1", {HID, cg, K});
        } else {   // transformers 4.22-era checkpoints (the reference's pin), SURVEY.md §0.9
            g = sd.get(q + ".weight_g", {1, 1, K});
            v = sd.get(q + ".weight_v", {HID, cg, K});
        }
        const float *b = sd.get(q + ".bias", {HID});
        if (!g || !v || !b) return 1;
        std::vector<double> nrm(K, 0.0);
        for (size_t i = 0; i < (size_t)HID * cg; ++i)
            for (int k = 0; k < K; ++k) nrm[k] += (double)v[i * K + k] * v[i * K + k];
        for (int k = 0; k < K; ++k) nrm[k] = std::sqrt(nrm[k]);
        const int cpad = 64, npad = f->pos_npad;
        if (cg > cpad || cg % 4) return fail("pos_conv: unsupported group width");
        f->pos_ktot = K * cpad;
        std::vector<float> wp((size_t)G * npad * f->pos_ktot, 0.f), bp((size_t)G * npad, 0.f);
        for (int gi = 0; gi < G; ++gi)
            for (int o = 0; o < cg; ++o) {
                const int oc = gi * cg + o;
                bp[(size_t)gi * npad + o] = b[oc];
                for (int k = 0; k < K; ++k)
                    for (int c = 0; c < cg; ++c)
                        wp[((size_t)gi * npad + o) * f->pos_ktot + (size_t)k * cpad + c] =
                            (float)(g[k] * (v[((size_t)oc * cg + c) * K + k] / nrm[k]));
            }
        TS_TRY(up(f->pos_w, wp.data(), wp.size()));
        TS_TRY(up(f->pos_b, bp.data(), bp.size()));
        if (cg == 48) {
            std::vector<float> wd((size_t)G * cg * K * cg);
            for (int gi = 0; gi < G; ++gi)
                for (int o = 0; o < cg; ++o)
                    for (int k = 0; k < K; ++k)
                        for (int c = 0; c < cg; ++c)
                            wd[(((size_t)gi * cg + o) * K + k) * cg + c] = wp[((size_t)gi * npad + o) * f->pos_ktot + (size_t)k * cpad + c];
            TS_TRY(up(f->pos_wd, wd.data(), wd.size()));
        }
    }
    TS_TRY(load_ln(sd, p + "encoder.layer_norm", HID, &f->enc_ln));
    for (int l = 0; l < n_layers; ++l) {
        const std::string q = p + "encoder.layers." + std::to_string(l) + ".";
        std::unique_ptr<EncLayer> L(new EncLayer());
        const float *wq = sd.get(q + "attention.q_proj.weight", {HID, HID}), *bq = sd.get(q + "attention.q_proj.bias", {HID});
        const float *wk = sd.get(q + "attention.k_proj.weight", {HID, HID}), *bk = sd.get(q + "attention.k_proj.bias", {HID});
        const float *wv = sd.get(q + "attention.v_proj.weight", {HID, HID}), *bv = sd.get(q + "attention.v_proj.bias", {HID});
        if (!wq || !bq || !wk || !bk || !wv || !bv) return 1;
        std::vector<float> w3((size_t)3 * HID * HID), b3((size_t)3 * HID);
        std::memcpy(w3.data(), wq, (size_t)HID * HID * 4);
        std::memcpy(w3.data() + (size_t)HID * HID, wk, (size_t)HID * HID * 4);
        std::memcpy(w3.data() + (size_t)2 * HID * HID, wv, (size_t)HID * HID * 4);
        std::memcpy(b3.data(), bq, HID * 4);
        std::memcpy(b3.data() + HID, bk, HID * 4);
        std::memcpy(b3.data() + 2 * HID, bv, HID * 4);
        TS_TRY(pack_linear_layer(w3.data(), HID, b3.data(), 3 * HID, HID, &L->qkv));
        TS_TRY(load_linear(sd, q + "attention.out_proj", HID, HID, 0, &L->outp));
        TS_TRY(load_ln(sd, q + "layer_norm", HID, &L->ln1));
        TS_TRY(load_linear(sd, q + "feed_forward.intermediate_dense", FFN, HID, 3, &L->ff1));
        TS_TRY(load_linear(sd, q + "feed_forward.output_dense", HID, FFN, 0, &L->ff2));
        TS_TRY(load_ln(sd, q + "final_layer_norm", HID, &L->ln2));
        f->layers.push_back(std::move(L));
    }
    // ---- heads ----
    TS_TRY(load_linear(sd, "audio_feature_map", 256, HID, 0, &f->afm));
    if (num_classes > 0) {
        const float *iw = sd.get("audio_middle.id_mlp.weight", {64, num_classes, 1}), *ib = sd.get("audio_middle.id_mlp.bias", {64});
        if (!iw || !ib) return 1;
        TS_TRY(up(f->id_w, iw, (size_t)64 * num_classes));
        TS_TRY(up(f->id_b, ib, 64));
    }
    const std::string fn = "audio_middle.first_net.conv_layers.";
    // 320 -> 256 channels: the residual branch is a conv; 256 -> 256 (identity=False): nn.Identity, no keys (layers.py:95-96)
    if (f->CIN != 256) TS_TRY(load_conv3(sd, fn + "0.residual_layer.0", 256, f->CIN, 3, &f->fn0res));
    TS_TRY(load_conv3(sd, fn + "0.conv", 256, f->CIN, 3, &f->fn[0]));
    TS_TRY(load_conv3(sd, fn + "1.conv", 256, 256, 3, &f->fn[1]));
    TS_TRY(load_conv3(sd, fn + "2.conv", 256, 256, 3, &f->fn[2]));
    for (int i = 0; i < 3; ++i) TS_TRY(load_ln(sd, fn + std::to_string(i) + ".norm", 256, &f->fn_ln[i]));
    for (int d = 0; d < 2; ++d) {
        const int c = d == 0 ? 64 : 256;
        for (int i = 0; i < 3; ++i) {
            const std::string k = "decoder." + std::to_string(d) + "." + std::to_string(i);
            TS_TRY(load_conv3(sd, k + ".conv", c, i == 0 ? 256 : c, 3, &f->dec[d][i]));
            TS_TRY(load_ln(sd, k + ".norm", c, &f->dec_ln[d][i]));
        }
        TS_TRY(load_conv3(sd, "final_out." + std::to_string(d), d == 0 ? f->JAW : 100, c, 1, &f->fin[d]));
    }
    *out = f.release();
    return 0;
}

void ts_face_destroy(ts_face *f) { delete f; }

// Arithmetic plan of the generator's GEMMs (feature convolutions 1..6, projections, positional conv, transformer-block GEMMs,
// LN-conv heads): 0 = fp32 MFMA (default: what every parity claim is made on), 3 / 6 = split-bf16 with three / six bf16 products
// per fp32 product (conv_gemm_split.hip).  Layer 0 of the feature extractor (K = 10, VALU), the attention products, LayerNorms
// and soft-max stay fp32 in every plan.
int ts_face_set_arith(ts_face *f, int bf16_products) {
    if (!f) return fail("ts_face_set_arith: null argument");
    if (bf16_products != 0 && bf16_products != 3 && bf16_products != 6) return fail("ts_face_set_arith: 0 (fp32), 3 or 6 bf16 products");
    const int planes = bf16_products == 0 ? 0 : (bf16_products == 3 ? 2 : 3);
    // (the plan is published LAST, after the weight plane images exist and the device has finished writing them: a generate call that
    // starts after this function returns sees a complete plan.  Like every entry of a handle, set_arith must not run concurrently
    // with ts_face_generate on the same handle — talkshow_hip.h, thread contract.)
    if (planes == 2 && !f->pos_wp.p) {
        // weights are constants: their two bf16 planes are made once, here, in the layout conv_gemm_split copies into LDS, and only the
        // activations are split per call (the plane images have the size and pitch of the fp32 matrices: + 1 x the weights of HBM)
        TS_HIP(hipSetDevice(f->ctx->device));
        std::vector<ConvLayer *> all;
        for (auto &c : f->fc) all.push_back(&c);
        all.push_back(&f->fp_proj);
        for (auto &L : f->layers)
            for (ConvLayer *c : {&L->qkv, &L->outp, &L->ff1, &L->ff2}) all.push_back(c);
        all.push_back(&f->afm);
        for (auto &c : f->fn) all.push_back(&c);
        if (f->CIN != 256) all.push_back(&f->fn0res);
        for (auto &d : f->dec)
            for (auto &c : d) all.push_back(&c);
        for (auto &c : f->fin) all.push_back(&c);
        for (ConvLayer *c : all) {
            if (!c->w.p || c->ktot % 32) continue;
            TS_TRY(c->wp.ensure(c->w.bytes));
            TS_HIP(launch_split_weight_planes(c->w.f(), c->wp.f(), (long)(c->w.bytes / sizeof(float) / c->ktot), c->ktot, nullptr));
        }
        TS_TRY(f->pos_wp.ensure(f->pos_w.bytes));
        TS_HIP(launch_split_weight_planes(f->pos_w.f(), f->pos_wp.f(), (long)(f->pos_w.bytes / sizeof(float) / f->pos_ktot), f->pos_ktot, nullptr));
        TS_HIP(hipDeviceSynchronize());   // the fills ran on the null stream; generate streams are non-blocking: order by completion
    }
    f->split_planes = planes;
    return 0;
}

// s2g_face.Generator.forward, eval (s2g_face.py:196-224): wav (B,N) fp32, id (B,num_classes) fp32 (one-hot or zeros,
// smplx_face.py:205-208) -> out (B,frames,103); hidden_out optional (B,frames,768) = wav2vec2 last_hidden_state.
// A handle created with num_classes = 0 is Generator(identity=False): id is ignored (may be NULL), out is (B,frames,106).
int ts_face_generate(ts_face *f, const float *wav, int B, int N, int frames, const float *id, float *out, float *hidden_out,
                     void *stream) {
    if (!f || !wav || !out || (!id && f->NCLS > 0)) return fail("ts_face_generate: null argument");
    if (B < 1 || frames < 1) return fail("ts_face_generate: bad shape");
    hipStream_t s = (hipStream_t)stream;
    ts_ctx *ctx = f->ctx;
    const int C0 = f->C0, HID = f->HID, FFN = f->FFN, HEADS = f->HEADS, T = frames;
    int L[7];
    L[0] = (N - 10) / 5 + 1;
    if (N < 10 || L[0] < 1) return fail("ts_face_generate: audio too short");
    for (int i = 0; i < 6; ++i) {
        L[i + 1] = (L[i] - f->fc_k[i]) / 2 + 1;
        if (L[i] < f->fc_k[i]) return fail("ts_face_generate: audio too short (needs >= 400 samples)");
    }
    const long M = (long)B * T;
    ts_face::Work &w = f->work(s);
    const size_t F = sizeof(float);
    const int ntb = (L[0] + 127) / 128;
    TS_TRY(w.A.ensure((size_t)B * L[0] * C0 * F));
    TS_TRY(w.Bf.ensure((size_t)B * L[1] * C0 * F));
    TS_TRY(w.part.ensure((size_t)B * ntb * C0 * sizeof(double2)));
    TS_TRY(w.stats.ensure((size_t)B * C0 * sizeof(float2)));
    TS_TRY(w.X512.ensure(M * C0 * F));
    TS_TRY(w.H.ensure((M * HID + 64) * F));
    TS_TRY(w.H2.ensure(M * HID * F));
    TS_TRY(w.TMP.ensure(M * HID * F));
    TS_TRY(w.QKV.ensure(M * 3 * HID * F));
    TS_TRY(w.ATT.ensure(M * HID * F));
    TS_TRY(w.FF.ensure(M * FFN * F));
    TS_TRY(w.X320.ensure(M * 320 * F));
    const int CIN = f->CIN, JAW = f->JAW, OUTW = f->JAW + 100;
    TS_TRY(w.Y1.ensure(M * 256 * F));
    TS_TRY(w.Y2.ensure(M * 256 * F));
    TS_TRY(w.R.ensure(M * 256 * F));
    TS_TRY(w.D1.ensure(M * 64 * F));
    TS_TRY(w.D2.ensure(M * 64 * F));

    ConvParams p;
    auto conv = [&](const ConvLayer &Ly, const float *x, int ldx, int Bc, int Lin, int Lout, int stride, const float *res,
                    int ldr, float *o, int ldo, int col0, int nstore, int act) -> int {
        params_for(Ly, x, ldx, Bc, Lin, Lout, stride, res, ldr, o, ldo, col0, nstore, act, &p, f->split_planes == 2);
        p.sk_ok = 1;   // tolerance-only GEMMs (<= 1e-4 vs the reference; measured 2e-6): the ring engine may split the last unit's tiles in K
        return run_conv(ctx, p, f->split_planes ? 20 + f->split_planes : 0, s);
    };
    auto ln = [&](const float *x, int C, const LNp &q, const float *post, int relu, float *o) -> int {
        MiscScope ms(ctx, s);
        TS_HIP(launch_layernorm_rows(x, C, M, C, q.g.f(), q.b.f(), post, C, relu, o, C, s));
        return 0;
    };

    // ---- wav2vec2 feature extractor ----
    {
        MiscScope ms(ctx, s);
        TS_HIP(launch_w2v_conv0(wav, B, N, L[0], f->c0_w.f(), f->c0_g.f(), f->c0_b.f(), static_cast<double2 *>(w.part.p),
                                static_cast<float2 *>(w.stats.p), w.A.f(), C0, s));
    }
    float *cur = w.A.f(), *nxt = w.Bf.f();
    for (int i = 0; i < 6; ++i) {
        TS_TRY(conv(f->fc[i], cur, C0, B, L[i], L[i + 1], 2, nullptr, 0, nxt, C0, 0, C0, 3));
        std::swap(cur, nxt);
    }
    {
        MiscScope ms(ctx, s);
        TS_HIP(launch_lerp_ln(cur, B, L[6], T, f->fp_ln.g.f(), f->fp_ln.b.f(), w.X512.f(), s));
    }
    // the grouped positional conv reads 64-channel windows every 48 channels: the last group's window runs 16 floats past
    // the row (zero weights there) — keep the slack after the final row finite
    TS_HIP(hipMemsetAsync(w.H.f() + M * HID, 0, 64 * F, s));
    TS_TRY(conv(f->fp_proj, w.X512.f(), C0, 1, (int)M, (int)M, 1, nullptr, 0, w.H.f(), HID, 0, HID, 0));
    // ---- positional conv embedding (16 groups as batched problems) + residual + LayerNorm ----
    {
        std::memset(&p, 0, sizeof(p));
        const int cg = HID / f->POSG;
        p.M = (int)M;
        p.Lout = p.Lin = T;
        p.stride = 1;
        p.ldx = HID;
        p.ldo = HID;
        p.ldr = HID;
        p.N = cg;
        p.Ktot = f->pos_ktot;
        p.act = 3;
        p.res_after_act = 1;
        p.ngroups = f->POSG;
        p.zdiv = f->POSG;
        p.x_zs1 = cg;
        p.w_zs1 = (long)f->pos_npad * f->pos_ktot;
        p.o_zs1 = cg;
        p.b_zs1 = f->pos_npad;
        p.r_zs1 = cg;
        ConvGroup &G = p.g[0];
        G.x = w.H.f();
        G.w = f->pos_w.f();
        if (f->split_planes == 2 && f->pos_wp.p) {
            G.w = f->pos_wp.f();
            p.w_planes = 1;
        }
        G.bias = f->pos_b.f();
        G.res = w.H.f();
        G.out = w.TMP.f();
        G.nseg = 1;
        G.seg[0] = ConvSeg{-(f->POSK / 2), 0, 64, f->POSK};
        int pos_tile = f->split_planes ? 20 + f->split_planes : 0;
        if (!f->split_planes && f->pos_wd.p && ts::knobs().conv_taps48) {   // unpadded: 48-channel taps, 48 weight rows per group (conv_taps48.hip)
            ConvParams q = p;
            q.Ktot = f->POSK * cg;
            q.w_zs1 = (long)cg * q.Ktot;
            q.g[0].w = f->pos_wd.f();
            q.g[0].seg[0].len = cg;
            // the unpadded weights are ONLY laid out for conv_taps48: take them when that kernel takes the layer (alignment, lengths) and
            // name it explicitly (tile 48 fails loudly on a mismatch); otherwise the padded 64-row weights + the generic engine stay (ADVICE r5:
            // a fall-through to 64 x 64 tiles over 48-row weights would read 16 rows past the last group)
            if (conv_taps48_takes(q)) {
                p = q;
                pos_tile = 48;
            }
        }
        TS_TRY(run_conv(ctx, p, pos_tile, s));
    }
    TS_TRY(ln(w.TMP.f(), HID, f->enc_ln, nullptr, 0, w.H.f()));
    // ---- transformer layers (post-LN) ----
    for (auto &Lp : f->layers) {
        EncLayer &E = *Lp;
        TS_TRY(conv(E.qkv, w.H.f(), HID, 1, (int)M, (int)M, 1, nullptr, 0, w.QKV.f(), 3 * HID, 0, 3 * HID, 0));
        // softmax(Q K^T / 8) V per (clip, head), fused: the scores stay in registers (face.hip::attention_kernel)
        {
            MiscScope ms(ctx, s, FAM_ATTN, 4.0 * B * HEADS * (double)T * T * 64);   // Q K^T and P V: 2 x (2 T^2 d) per (clip, head)
            TS_HIP(launch_attention(w.QKV.f(), B, T, HID, HEADS, 0.125f, w.ATT.f(), s));
        }
        TS_TRY(conv(E.outp, w.ATT.f(), HID, 1, (int)M, (int)M, 1, w.H.f(), HID, w.TMP.f(), HID, 0, HID, 0));
        TS_TRY(ln(w.TMP.f(), HID, E.ln1, nullptr, 0, w.H2.f()));
        TS_TRY(conv(E.ff1, w.H2.f(), HID, 1, (int)M, (int)M, 1, nullptr, 0, w.FF.f(), FFN, 0, FFN, 3));
        TS_TRY(conv(E.ff2, w.FF.f(), FFN, 1, (int)M, (int)M, 1, w.H2.f(), HID, w.TMP.f(), HID, 0, HID, 0));
        TS_TRY(ln(w.TMP.f(), HID, E.ln2, nullptr, 0, w.H.f()));
    }
    if (hidden_out) TS_HIP(hipMemcpyAsync(hidden_out, w.H.f(), M * HID * F, hipMemcpyDeviceToDevice, s));
    // ---- audio_feature_map | id channels ----
    TS_TRY(conv(f->afm, w.H.f(), HID, 1, (int)M, (int)M, 1, nullptr, 0, w.X320.f(), CIN, 0, 256, 0));
    if (f->NCLS > 0) {
        MiscScope ms(ctx, s);
        TS_HIP(launch_fill_id(id, f->NCLS, f->id_w.f(), f->id_b.f(), 64, w.X320.f(), 320, 256, B, T, s));
    }
    // ---- SeqTranslator1D: 3 x {conv, LN, + residual, ReLU} ----
    TS_TRY(conv(f->fn[0], w.X320.f(), CIN, B, T, T, 1, nullptr, 0, w.Y1.f(), 256, 0, 256, 0));
    if (CIN != 256) TS_TRY(conv(f->fn0res, w.X320.f(), CIN, B, T, T, 1, nullptr, 0, w.R.f(), 256, 0, 256, 0));
    TS_TRY(ln(w.Y1.f(), 256, f->fn_ln[0], CIN != 256 ? w.R.f() : w.X320.f(), 1, w.Y2.f()));
    TS_TRY(conv(f->fn[1], w.Y2.f(), 256, B, T, T, 1, nullptr, 0, w.Y1.f(), 256, 0, 256, 0));
    TS_TRY(ln(w.Y1.f(), 256, f->fn_ln[1], w.Y2.f(), 1, w.R.f()));
    TS_TRY(conv(f->fn[2], w.R.f(), 256, B, T, T, 1, nullptr, 0, w.Y1.f(), 256, 0, 256, 0));
    TS_TRY(ln(w.Y1.f(), 256, f->fn_ln[2], w.R.f(), 1, w.Y2.f()));     // feature = Y2
    // ---- jaw head (64 ch) ----
    TS_TRY(conv(f->dec[0][0], w.Y2.f(), 256, B, T, T, 1, nullptr, 0, w.D1.f(), 64, 0, 64, 0));
    TS_TRY(ln(w.D1.f(), 64, f->dec_ln[0][0], nullptr, 1, w.D2.f()));
    TS_TRY(conv(f->dec[0][1], w.D2.f(), 64, B, T, T, 1, nullptr, 0, w.D1.f(), 64, 0, 64, 0));
    TS_TRY(ln(w.D1.f(), 64, f->dec_ln[0][1], nullptr, 1, w.D2.f()));
    TS_TRY(conv(f->dec[0][2], w.D2.f(), 64, B, T, T, 1, nullptr, 0, w.D1.f(), 64, 0, 64, 0));
    TS_TRY(ln(w.D1.f(), 64, f->dec_ln[0][2], nullptr, 1, w.D2.f()));
    TS_TRY(conv(f->fin[0], w.D2.f(), 64, 1, (int)M, (int)M, 1, nullptr, 0, out, OUTW, 0, JAW, 0));
    // ---- expression head (256 ch) ----
    TS_TRY(conv(f->dec[1][0], w.Y2.f(), 256, B, T, T, 1, nullptr, 0, w.Y1.f(), 256, 0, 256, 0));
    TS_TRY(ln(w.Y1.f(), 256, f->dec_ln[1][0], nullptr, 1, w.R.f()));
    TS_TRY(conv(f->dec[1][1], w.R.f(), 256, B, T, T, 1, nullptr, 0, w.Y1.f(), 256, 0, 256, 0));
    TS_TRY(ln(w.Y1.f(), 256, f->dec_ln[1][1], nullptr, 1, w.R.f()));
    TS_TRY(conv(f->dec[1][2], w.R.f(), 256, B, T, T, 1, nullptr, 0, w.Y1.f(), 256, 0, 256, 0));
    TS_TRY(ln(w.Y1.f(), 256, f->dec_ln[1][2], nullptr, 1, w.R.f()));
    TS_TRY(conv(f->fin[1], w.R.f(), 256, 1, (int)M, (int)M, 1, nullptr, 0, out, OUTW, JAW, 100, 0));
    return 0;
}

}  // extern "C"
