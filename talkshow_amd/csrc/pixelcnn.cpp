// GatedPixelCNN (nets/spg/gated_pixelcnn_v2.py) as an INCREMENTAL, row-cached launch plan.
//
// The reference's generate() (:152-177) re-runs the whole 15-layer network over the whole H x 2 grid for each of
// the 2H positions.  Causality (:57-87: layer 0 is mask 'A'; vertical kernels look up, horizontal kernels look left)
// means position (r, j) only needs
//   * the vertical stack at row r: layer 0 sees code rows r-3..r-1, layer l>=1 sees its own input at rows r-1, r;
//   * the horizontal stack at columns <= j of row r.
// So per row we run the vertical stack once for both columns (keeping one previous row per layer), then the
// horizontal chain for column 0 (body code), sample, then column 1 (hand code), sample: 216x fewer FLOPs, same
// arithmetic per position (each conv tap is multiplied exactly once, fp32 fmaf accumulation).
//
// Stage list per row r (every stage = one skinny_gemm launch over all B clips):
//   v0        h_vert_0 = vert_stack_0(E[codes[r-3..r-1]]) ; OV0 = gate(h_vert_0 + c_0[label])
//   fuse_v    XV_1[r]  = fusion_v[:, :D] . OV0 + AEV[r]            (AEV = fusion_v[:, D:] . embedding_aud(aud) + biases, precomputed)
//   v_l       h_vert_l = vert_stack_l(XV_l[r-1], XV_l[r]) ; XV_{l+1}[r] = gate(h_vert_l + c_l)          l = 1..NL-1
//   v2h_l     V2H_l    = vert_to_horiz_l(h_vert_l)                                                       l = 0..NL-1
//   for column j = 0, 1:
//     hg_l    G = gate(V2H_l[j] + horiz_stack_l(XH_l[j-1], XH_l[j]) + c_l)     (layer 0: only column j-1, mask 'A')
//     hr_l    XH_{l+1}[j] = horiz_resid_l(G) (+ XH_l[j] for l >= 1)
//     fuse_h  (after layer 0) XH_1[j] = fusion_h[:, :D] . out_h_0 + AEH[r]
//     head1/2 logits = output_conv(XH_NL[j]) ; sample -> codes[r][j]
#include <algorithm>
#include <cstdlib>
#include <tuple>

#include "host_common.h"

using namespace ts;

struct ts_pixelcnn {
    ts_ctx *ctx = nullptr;
    int V = 0, D = 0, NL = 0, NC = 0, AD = 0, HID = 512;
    DevBuf emb;                                   // [V][D]
    std::vector<std::unique_ptr<DevBuf>> wv, bv;  // vertical: [2*2D][Kv], bias [2*2D] (duplicated per column)
    std::vector<std::unique_ptr<DevBuf>> wv2h, bv2h, wh, bh, cls, wr, br;
    DevBuf fva, fha;                              // fusion_{v,h}[:, :D]  [D][D]
    ConvLayer aud_embed, aud_fv, aud_fh;          // embedding_aud ; fusion_{v,h}[:, D:] (+ fusion bias)
    DevBuf w1, b1, w2, b2;                        // output_conv
    bool use_graph = true;
    // Work set: every buffer the row loop touches + the hipGraph replays of it.  One per stream, so that independent
    // batches can be in flight on different streams against the single weight copy above.
    struct Work {
        int capB = 0, capH = 0;
        DevBuf aud_all, AE, AEV, AEH, tok32, label32, XV, OV0, OVlast, HV, V2H, XH, G, OH0, Y, LG, tfcodes;
        // every pointer inside the captured kernels is one of the buffers above or the staging buffers below, so a graph
        // is valid for any caller pointers; key = (B, H, H0, mode)
        DevBuf codes_int, unif_int, dyn;
        hipStream_t cap_stream = nullptr;
        std::map<std::tuple<int, int, int, int>, hipGraphExec_t> graphs;
        std::map<std::tuple<int, int, int, int>, std::pair<long, double>> graph_stats;   // skinny launches, flops
        void drop_graphs() {
            for (auto &kv : graphs) (void)hipGraphExecDestroy(kv.second);
            graphs.clear();
        }
        ~Work() {
            drop_graphs();
            if (cap_stream) (void)hipStreamDestroy(cap_stream);
        }
    };
    std::map<hipStream_t, std::unique_ptr<Work>> works;
    Work &work(hipStream_t s) {
        auto &w = works[s];
        if (!w) w.reset(new Work());
        return *w;
    }
};

namespace {

int upload_vec(std::vector<std::unique_ptr<DevBuf>> &dst, const std::vector<float> &v) {
    dst.emplace_back(new DevBuf());
    return dst.back()->upload(v.data(), v.size() * sizeof(float));
}

int ensure_work(ts_pixelcnn *p, ts_pixelcnn::Work *w, int B, int Htot) {
    if (B <= w->capB && Htot <= w->capH) return 0;
    const int cb = std::max(B, w->capB), ch = std::max(Htot, w->capH);
    const size_t D = p->D, NL = p->NL, f = sizeof(float);
    TS_TRY(w->aud_all.ensure((size_t)cb * ch * p->AD * f));
    TS_TRY(w->AE.ensure((size_t)cb * ch * D * f));
    TS_TRY(w->AEV.ensure((size_t)cb * ch * D * f));
    TS_TRY(w->AEH.ensure((size_t)cb * ch * D * f));
    TS_TRY(w->tok32.ensure((size_t)cb * ch * 2 * sizeof(int)));
    TS_TRY(w->label32.ensure((size_t)cb * sizeof(int)));
    TS_TRY(w->XV.ensure(NL * 2 * cb * 2 * D * f));
    TS_TRY(w->OV0.ensure((size_t)cb * 2 * D * f));
    TS_TRY(w->OVlast.ensure((size_t)cb * 2 * D * f));
    TS_TRY(w->HV.ensure(NL * cb * 4 * D * f));
    TS_TRY(w->V2H.ensure(NL * cb * 4 * D * f));
    TS_TRY(w->XH.ensure((NL + 1) * 2 * cb * D * f));
    TS_TRY(w->G.ensure((size_t)cb * D * f));
    TS_TRY(w->OH0.ensure((size_t)cb * D * f));
    TS_TRY(w->Y.ensure((size_t)cb * p->HID * f));
    TS_TRY(w->LG.ensure((size_t)cb * p->V * f));
    TS_TRY(w->tfcodes.ensure((size_t)cb * ch * 2 * sizeof(int64_t)));
    TS_TRY(w->codes_int.ensure((size_t)cb * ch * 2 * sizeof(int64_t)));
    TS_TRY(w->unif_int.ensure((size_t)cb * ch * 2 * sizeof(float)));
    TS_TRY(w->dyn.ensure(2 * sizeof(uint64_t)));
    w->drop_graphs();   // buffers moved: captured pointers are stale
    w->capB = cb;
    w->capH = ch;
    return 0;
}

struct RunCfg {
    int B, H, H0, Htot, mode;
    const float *uniforms;
    uint64_t seed;
    int64_t clip0;
    int64_t *codes;     // (B,H,2)
    float *logits;      // (B,H,2,V) or null
    const uint64_t *dyn;   // device {seed, clip0} (graph replay) or null
    ts_pixelcnn::Work *w;
};

SkinnyParams base_params(int M, int N, int epi) {
    SkinnyParams q;
    std::memset(&q, 0, sizeof(q));
    q.M = M;
    q.N = N;
    q.epi = epi;
    return q;
}
void add_dense(SkinnyParams &q, const float *base, long stride, int shift, int len) {
    SkinnySeg &s = q.seg[q.nseg++];
    s.base = base;
    s.gidx = nullptr;
    s.row_stride = stride;
    s.gidx_stride = 0;
    s.row_shift = shift;
    s.len = len;
    q.Ktot += len;
}
void add_gather(SkinnyParams &q, const float *table, long stride, const int *gidx, long gstride, int len) {
    SkinnySeg &s = q.seg[q.nseg++];
    s.base = table;
    s.gidx = gidx;
    s.row_stride = stride;
    s.gidx_stride = gstride;
    s.row_shift = 0;
    s.len = len;
    q.Ktot += len;
}

// vertical stack + v->h projections of row r.  Launch plan (NL + 2 launches): v0 | fuse_v + v2h_0 | v_1 |
// v_2 + v2h_1 | ... | v_{NL-1} + v2h_{NL-2} | v2h_{NL-1}: vert_to_horiz of layer l-1 and the vertical conv of layer l
// both depend only on layer l-1's output, so they share a launch (two independent problems, blockIdx.z).
int vertical_row(ts_pixelcnn *p, const RunCfg &c, int r, hipStream_t s) {
    ts_ctx *ctx = p->ctx;
    const int B = c.B, D = p->D, NL = p->NL, Htot = c.Htot;
    const int *tok = c.w->tok32.i();
    const int *lab = c.w->label32.i();
    auto XV = [&](int l, int par) { return c.w->XV.f() + ((size_t)(l * 2 + par) * B) * 2 * D; };
    auto HV = [&](int l) { return c.w->HV.f() + (size_t)l * B * 4 * D; };
    auto V2H = [&](int l) { return c.w->V2H.f() + (size_t)l * B * 4 * D; };

    auto make_v = [&](int l) {
        SkinnyParams q = base_params(B, 4 * D, EPI_GATE);
        if (l == 0) {
            for (int t = 0; t < 3; ++t) {
                const int rr = r - 3 + t;
                for (int col = 0; col < 2; ++col) {
                    if (rr >= 0) add_gather(q, p->emb.f(), D, tok + (size_t)rr * 2 + col, (long)Htot * 2, D);
                    else add_gather(q, p->emb.f(), D, ctx->neg1.i(), 0, D);   // zero padding above the grid
                }
            }
        } else {
            add_dense(q, r > 0 ? XV(l, (r - 1) & 1) : nullptr, 2 * D, 0, 2 * D);
            add_dense(q, XV(l, r & 1), 2 * D, 0, 2 * D);
        }
        q.W = p->wv[l]->f();
        q.ldw = q.Ktot;
        q.bias = p->bv[l]->f();
        q.cls = p->cls[l]->f();
        q.label = lab;
        q.cls_ld = 2 * D;
        q.gateD = D;
        q.out = l == 0 ? c.w->OV0.f() : (l + 1 < NL ? XV(l + 1, r & 1) : c.w->OVlast.f());
        q.out_stride = 2 * D;
        q.pre = HV(l);
        q.pre_stride = 4 * D;
        return q;
    };
    auto make_fuse_v = [&]() {   // audio fusion in front of layer 1 (gated_pixelcnn_v2.py:137-144)
        SkinnyParams f = base_params(2 * B, D, EPI_LINEAR);
        add_dense(f, c.w->OV0.f(), D, 0, D);
        f.W = p->fva.f();
        f.ldw = D;
        f.add1 = c.w->AEV.f() + (size_t)r * D;
        f.add1_stride = (long)Htot * D;
        f.add1_shift = 1;
        f.out = XV(1, r & 1);
        f.out_stride = D;
        return f;
    };
    auto make_v2h = [&](int l) {   // vert_to_horiz on the pre-gate activations, both columns
        SkinnyParams v = base_params(2 * B, 2 * D, EPI_LINEAR);
        add_dense(v, HV(l), 2 * D, 0, 2 * D);
        v.W = p->wv2h[l]->f();
        v.ldw = 2 * D;
        v.bias = p->bv2h[l]->f();
        v.out = V2H(l);
        v.out_stride = 2 * D;
        return v;
    };

    TS_TRY(run_skinny(ctx, make_v(0), s));
    if (NL == 1) return run_skinny(ctx, make_v2h(0), s);
    TS_TRY(run_skinny2(ctx, make_fuse_v(), make_v2h(0), s));
    TS_TRY(run_skinny(ctx, make_v(1), s));
    for (int l = 2; l < NL; ++l) TS_TRY(run_skinny2(ctx, make_v(l), make_v2h(l - 1), s));
    return run_skinny(ctx, make_v2h(NL - 1), s);
}

// horizontal chain + head + sampler for position (r, j)
int horizontal_pos(ts_pixelcnn *p, const RunCfg &c, int r, int j, hipStream_t s) {
    ts_ctx *ctx = p->ctx;
    const int B = c.B, D = p->D, NL = p->NL, Htot = c.Htot;
    const int *tok = c.w->tok32.i();
    auto V2H = [&](int l) { return c.w->V2H.f() + (size_t)l * B * 4 * D; };
    auto XH = [&](int l, int col) { return c.w->XH.f() + ((size_t)(l * 2 + col) * B) * D; };

    for (int l = 0; l < NL; ++l) {
        SkinnyParams q = base_params(B, 2 * D, EPI_GATE);
        q.ldw = 2 * D;
        if (l == 0) {   // mask 'A': only the column to the left, i.e. the embedding of the code just sampled
            if (j == 1) add_gather(q, p->emb.f(), D, tok + (size_t)r * 2 + 0, (long)Htot * 2, D);
            q.W = p->wh[0]->f();               // tap 0 block
        } else if (j == 0) {
            add_dense(q, XH(l, 0), D, 0, D);
            q.W = p->wh[l]->f() + D;           // tap 1 block (the column itself)
        } else {
            add_dense(q, XH(l, 0), D, 0, D);
            add_dense(q, XH(l, 1), D, 0, D);
            q.W = p->wh[l]->f();
        }
        q.bias = p->bh[l]->f();
        q.add1 = V2H(l) + (size_t)j * 2 * D;
        q.add1_stride = 4 * D;
        q.cls = p->cls[l]->f();
        q.label = c.w->label32.i();
        q.cls_ld = 2 * D;
        q.gateD = D;
        q.out = c.w->G.f();
        q.out_stride = D;
        TS_TRY(run_skinny(ctx, q, s));

        SkinnyParams h = base_params(B, D, EPI_LINEAR);
        add_dense(h, c.w->G.f(), D, 0, D);
        h.W = p->wr[l]->f();
        h.ldw = D;
        h.bias = p->br[l]->f();
        if (l == 0) {
            h.out = c.w->OH0.f();
        } else {
            h.add1 = XH(l, j);
            h.add1_stride = D;
            h.out = XH(l + 1, j);
        }
        h.out_stride = D;
        TS_TRY(run_skinny(ctx, h, s));

        if (l == 0 && NL > 1) {
            SkinnyParams f = base_params(B, D, EPI_LINEAR);
            add_dense(f, c.w->OH0.f(), D, 0, D);
            f.W = p->fha.f();
            f.ldw = D;
            f.add1 = c.w->AEH.f() + (size_t)r * D;
            f.add1_stride = (long)Htot * D;
            f.out = XH(1, j);
            f.out_stride = D;
            TS_TRY(run_skinny(ctx, f, s));
        }
    }
    const float *xfin = NL > 1 ? XH(NL, j) : c.w->OH0.f();
    SkinnyParams h1 = base_params(B, p->HID, EPI_LINEAR);
    add_dense(h1, xfin, D, 0, D);
    h1.W = p->w1.f();
    h1.ldw = D;
    h1.bias = p->b1.f();
    h1.relu = 1;
    h1.out = c.w->Y.f();
    h1.out_stride = p->HID;
    TS_TRY(run_skinny(ctx, h1, s));

    SkinnyParams h2 = base_params(B, p->V, EPI_LINEAR);
    add_dense(h2, c.w->Y.f(), p->HID, 0, p->HID);
    h2.W = p->w2.f();
    h2.ldw = p->HID;
    h2.bias = p->b2.f();
    h2.out = c.w->LG.f();
    h2.out_stride = p->V;
    TS_TRY(run_skinny(ctx, h2, s));

    SampleParams sp;
    std::memset(&sp, 0, sizeof(sp));
    const int ro = r - c.H0;   // row in the caller's (B,H,2) arrays
    sp.logits = c.w->LG.f();
    sp.B = B;
    sp.V = p->V;
    sp.mode = c.mode;
    sp.uniforms = c.uniforms ? c.uniforms + (size_t)ro * 2 + j : nullptr;
    sp.u_stride = (long)c.H * 2;
    sp.seed = c.seed;
    sp.clip_index0 = c.clip0;
    sp.dyn = c.dyn;
    sp.position = (uint32_t)(ro * 2 + j);
    sp.tok32 = c.w->tok32.i() + (size_t)r * 2 + j;
    sp.tok_stride = (long)Htot * 2;
    sp.codes = c.codes + (size_t)ro * 2 + j;
    sp.code_stride = (long)c.H * 2;
    if (c.logits) {
        sp.logits_copy = c.logits + ((size_t)ro * 2 + j) * p->V;
        sp.copy_stride = (long)c.H * 2 * p->V;
    }
    {
        MiscScope ms(ctx, s);
        TS_HIP(launch_sample(sp, s));
    }
    return 0;
}

}  // namespace

extern "C" {

int ts_pixelcnn_create(ts_ctx *ctx, const ts_tensor *sd_, int n, int V, int D, int NL, int NC, int AD, ts_pixelcnn **out) {
    if (!ctx || !sd_ || !out) return fail("ts_pixelcnn_create: null argument");
    if (D % 32 != 0 || AD % 32 != 0) return fail("PixelCNN dim and audio dim must be multiples of 32");
    if (NL < 1) return fail("n_layers must be >= 1");
    TS_HIP(hipSetDevice(ctx->device));
    StateDict sd(sd_, n);
    std::unique_ptr<ts_pixelcnn> p(new ts_pixelcnn());
    p->ctx = ctx;
    p->V = V;
    p->D = D;
    p->NL = NL;
    p->NC = NC;
    p->AD = AD;
    const int D2 = 2 * D;

    const float *emb = sd.get("embedding.weight", {V, D});
    if (!emb) return 1;
    TS_TRY(p->emb.upload(emb, (size_t)V * D * sizeof(float)));

    for (int l = 0; l < NL; ++l) {
        const std::string q = "layers." + std::to_string(l);
        const int kh = l == 0 ? 4 : 2;       // kernel // 2 + 1 rows, kernel = 7 / 3 (gated_pixelcnn_v2.py:34,112-113)
        const int rows = l == 0 ? 3 : 2;     // mask 'A' zeroes the last row of layer 0 (:57-58)
        const float *wv = sd.get(q + ".vert_stack.weight", {D2, D, kh, 3});
        const float *bv = sd.get(q + ".vert_stack.bias", {D2});
        const float *wvh = sd.get(q + ".vert_to_horiz.weight", {D2, D2, 1, 1});
        const float *bvh = sd.get(q + ".vert_to_horiz.bias", {D2});
        const float *wh = sd.get(q + ".horiz_stack.weight", {D2, D, 1, 2});
        const float *bh = sd.get(q + ".horiz_stack.bias", {D2});
        const float *cl = sd.get(q + ".class_cond_embedding.weight", {NC, D2});
        const float *wr = sd.get(q + ".horiz_resid.weight", {D, D, 1, 1});
        const float *br = sd.get(q + ".horiz_resid.bias", {D});
        if (!wv || !bv || !wvh || !bvh || !wh || !bh || !cl || !wr || !br) return 1;
        // vertical: output n = (col j, channel co); k = ((row tap t)*2 + input col c)*D + ci; kernel column c - j + 1
        const int Kv = rows * 2 * D;
        std::vector<float> pv((size_t)2 * D2 * Kv), pb((size_t)2 * D2);
        for (int j = 0; j < 2; ++j)
            for (int co = 0; co < D2; ++co) {
                pb[(size_t)j * D2 + co] = bv[co];
                for (int t = 0; t < rows; ++t)
                    for (int cc = 0; cc < 2; ++cc)
                        for (int ci = 0; ci < D; ++ci)
                            pv[((size_t)j * D2 + co) * Kv + ((size_t)t * 2 + cc) * D + ci] =
                                wv[(((size_t)co * D + ci) * kh + t) * 3 + (cc - j + 1)];
            }
        TS_TRY(upload_vec(p->wv, pv));
        TS_TRY(upload_vec(p->bv, pb));
        TS_TRY(upload_vec(p->wv2h, std::vector<float>(wvh, wvh + (size_t)D2 * D2)));
        TS_TRY(upload_vec(p->bv2h, std::vector<float>(bvh, bvh + D2)));
        // horizontal: row co, k = tap*D + ci ; tap 0 <-> column j-1, tap 1 <-> column j (kernel (1,2), padding (0,1))
        std::vector<float> ph((size_t)D2 * 2 * D);
        for (int co = 0; co < D2; ++co)
            for (int tap = 0; tap < 2; ++tap)
                for (int ci = 0; ci < D; ++ci)
                    ph[(size_t)co * 2 * D + (size_t)tap * D + ci] =
                        (l == 0 && tap == 1) ? 0.f : wh[((size_t)co * D + ci) * 2 + tap];   // mask 'A' (:59)
        TS_TRY(upload_vec(p->wh, ph));
        TS_TRY(upload_vec(p->bh, std::vector<float>(bh, bh + D2)));
        TS_TRY(upload_vec(p->cls, std::vector<float>(cl, cl + (size_t)NC * D2)));
        TS_TRY(upload_vec(p->wr, std::vector<float>(wr, wr + (size_t)D * D)));
        TS_TRY(upload_vec(p->br, std::vector<float>(br, br + D)));
    }
    // audio conditioning
    const float *wa = sd.get("embedding_aud.weight", {D, AD, 1, 1}), *ba = sd.get("embedding_aud.bias", {D});
    const float *wfv = sd.get("fusion_v.weight", {D, D2, 1, 1}), *bfv = sd.get("fusion_v.bias", {D});
    const float *wfh = sd.get("fusion_h.weight", {D, D2, 1, 1}), *bfh = sd.get("fusion_h.bias", {D});
    if (!wa || !ba || !wfv || !bfv || !wfh || !bfh) return 1;
    TS_TRY(pack_linear_layer(wa, AD, ba, D, AD, &p->aud_embed));
    TS_TRY(pack_linear_layer(wfv + D, D2, bfv, D, D, &p->aud_fv));
    TS_TRY(pack_linear_layer(wfh + D, D2, bfh, D, D, &p->aud_fh));
    std::vector<float> fa((size_t)D * D), fb((size_t)D * D);
    for (int o = 0; o < D; ++o)
        for (int i = 0; i < D; ++i) {
            fa[(size_t)o * D + i] = wfv[(size_t)o * D2 + i];
            fb[(size_t)o * D + i] = wfh[(size_t)o * D2 + i];
        }
    TS_TRY(p->fva.upload(fa.data(), fa.size() * sizeof(float)));
    TS_TRY(p->fha.upload(fb.data(), fb.size() * sizeof(float)));
    // head
    const float *w1 = sd.get("output_conv.0.weight", {p->HID, D, 1, 1}), *b1 = sd.get("output_conv.0.bias", {p->HID});
    const float *w2 = sd.get("output_conv.2.weight", {V, p->HID, 1, 1}), *b2 = sd.get("output_conv.2.bias", {V});
    if (!w1 || !b1 || !w2 || !b2) return 1;
    TS_TRY(p->w1.upload(w1, (size_t)p->HID * D * sizeof(float)));
    TS_TRY(p->b1.upload(b1, (size_t)p->HID * sizeof(float)));
    TS_TRY(p->w2.upload(w2, (size_t)V * p->HID * sizeof(float)));
    TS_TRY(p->b2.upload(b2, (size_t)V * sizeof(float)));
    if (const char *e = std::getenv("TS_NO_GRAPH")) p->use_graph = !(e[0] && e[0] != '0');
    *out = p.release();
    return 0;
}
void ts_pixelcnn_destroy(ts_pixelcnn *p) { delete p; }

int ts_pixelcnn_graph_stats(ts_pixelcnn *p, void *stream, int B, int H, int mode, int64_t *launches, double *flops) {
    if (!p) return fail("ts_pixelcnn_graph_stats: null argument");
    auto it = p->works.find((hipStream_t)stream);
    if (it == p->works.end()) return fail("ts_pixelcnn_graph_stats: nothing was run on this stream");
    auto jt = it->second->graph_stats.find(std::make_tuple(B, H, 0, mode));
    if (jt == it->second->graph_stats.end()) return fail("ts_pixelcnn_graph_stats: no captured graph for this shape");
    if (launches) *launches = jt->second.first;
    if (flops) *flops = jt->second.second;
    return 0;
}

int ts_pixelcnn_generate(ts_pixelcnn *p, const int64_t *label, const float *aud, int B, int H, int mode,
                         const float *uniforms, uint64_t seed, int64_t clip0, int64_t *codes, float *logits,
                         const int64_t *pre_codes, const float *pre_aud, int H0, void *stream) {
    if (!p || !label || !aud || !codes) return fail("ts_pixelcnn_generate: null argument");
    if (B < 1 || H < 1 || H0 < 0) return fail("ts_pixelcnn_generate: bad shape");
    if (mode < 0 || mode > TS_TEACHER_FORCED) return fail("ts_pixelcnn_generate: bad mode");
    if (mode == TS_SAMPLE_UNIFORMS && !uniforms) return fail("ts_pixelcnn_generate: uniforms required");
    if (H0 > 0 && (!pre_codes || !pre_aud)) return fail("ts_pixelcnn_generate: prefix pointers required");
    hipStream_t s = (hipStream_t)stream;
    ts_ctx *ctx = p->ctx;
    const int Htot = H0 + H, D = p->D, AD = p->AD;
    ts_pixelcnn::Work *w = &p->work(s);
    TS_TRY(ensure_work(p, w, B, Htot));

    // The row loop is replayed from a hipGraph (host launch cost would otherwise dominate: ~100 dependent tiny
    // launches per row); eager launches remain for the instrumented / logits-returning / teacher-forced paths.
    const bool graph = p->use_graph && !ctx->prof.on && !logits && mode != TS_TEACHER_FORCED;
    RunCfg c{B, H, H0, Htot, mode, uniforms, seed, clip0, codes, logits, nullptr, w};
    if (graph) {
        c.codes = static_cast<int64_t *>(w->codes_int.p);
        c.uniforms = mode == TS_SAMPLE_UNIFORMS ? w->unif_int.f() : nullptr;
        c.dyn = static_cast<const uint64_t *>(w->dyn.p);
    }

    // ---- audio conditioning for every row: AE = embedding_aud(aud); AEV/AEH = fusion_{v,h}[:, D:] . AE + bias ----
    const float *aud_all = aud;
    if (H0 > 0) {
        const size_t f = sizeof(float);
        TS_HIP(hipMemcpy2DAsync(w->aud_all.f(), (size_t)Htot * AD * f, pre_aud, (size_t)H0 * AD * f, (size_t)H0 * AD * f,
                                B, hipMemcpyDeviceToDevice, s));
        TS_HIP(hipMemcpy2DAsync(w->aud_all.f() + (size_t)H0 * AD, (size_t)Htot * AD * f, aud, (size_t)H * AD * f,
                                (size_t)H * AD * f, B, hipMemcpyDeviceToDevice, s));
        aud_all = w->aud_all.f();
    }
    {
        ConvParams q;
        conv_layer_params(p->aud_embed, aud_all, AD, 1, B * Htot, nullptr, 0, w->AE.f(), D, 0, D, &q);
        TS_TRY(run_conv(ctx, q, 0, s));
        conv_layer_params(p->aud_fv, w->AE.f(), D, 1, B * Htot, nullptr, 0, w->AEV.f(), D, 0, D, &q);
        TS_TRY(run_conv(ctx, q, 0, s));
        conv_layer_params(p->aud_fh, w->AE.f(), D, 1, B * Htot, nullptr, 0, w->AEH.f(), D, 0, D, &q);
        TS_TRY(run_conv(ctx, q, 0, s));
    }
    {
        MiscScope ms(ctx, s);
        TS_HIP(launch_i64_to_i32(label, w->label32.i(), B, s));
        // known codes: the continuity prefix, and every position when teacher forced
        if (H0 > 0 || mode == TS_TEACHER_FORCED) {
            int64_t *tf = static_cast<int64_t *>(w->tfcodes.p);
            const size_t e = sizeof(int64_t);
            if (H0 > 0)
                TS_HIP(hipMemcpy2DAsync(tf, (size_t)Htot * 2 * e, pre_codes, (size_t)H0 * 2 * e, (size_t)H0 * 2 * e, B,
                                        hipMemcpyDeviceToDevice, s));
            if (mode == TS_TEACHER_FORCED)
                TS_HIP(hipMemcpy2DAsync(tf + (size_t)H0 * 2, (size_t)Htot * 2 * e, codes, (size_t)H * 2 * e,
                                        (size_t)H * 2 * e, B, hipMemcpyDeviceToDevice, s));
            TS_HIP(launch_i64_to_i32(tf, w->tok32.i(), (long)B * Htot * 2, s));
        }
    }

    auto row_loop = [&](hipStream_t st) -> int {
        for (int r = 0; r < Htot; ++r) {
            TS_TRY(vertical_row(p, c, r, st));
            if (r < H0) continue;                                         // prefix rows only feed the row cache
            if (mode == TS_TEACHER_FORCED && !logits) continue;           // nothing to produce
            for (int j = 0; j < 2; ++j) TS_TRY(horizontal_pos(p, c, r, j, st));
        }
        return 0;
    };
    if (!graph) return row_loop(s);

    if (mode == TS_SAMPLE_UNIFORMS)
        TS_HIP(hipMemcpyAsync(w->unif_int.p, uniforms, (size_t)B * H * 2 * sizeof(float), hipMemcpyDeviceToDevice, s));
    const uint64_t dynh[2] = {seed, (uint64_t)clip0};
    TS_HIP(hipMemcpyAsync(w->dyn.p, dynh, sizeof(dynh), hipMemcpyHostToDevice, s));   // pageable: staged before return
    const auto key = std::make_tuple(B, H, H0, mode);
    auto it = w->graphs.find(key);
    if (it == w->graphs.end()) {
        if (!w->cap_stream) TS_HIP(hipStreamCreateWithFlags(&w->cap_stream, hipStreamNonBlocking));
        hipGraph_t g = nullptr;
        const long l0 = ctx->n_launch[FAM_SKINNY];
        const double f0 = ctx->n_flops[FAM_SKINNY];
        TS_HIP(hipStreamBeginCapture(w->cap_stream, hipStreamCaptureModeThreadLocal));
        const int rc = row_loop(w->cap_stream);
        w->graph_stats[key] = {ctx->n_launch[FAM_SKINNY] - l0, ctx->n_flops[FAM_SKINNY] - f0};
        const hipError_t ec = hipStreamEndCapture(w->cap_stream, &g);
        if (rc != 0) {
            if (g) (void)hipGraphDestroy(g);
            return rc;
        }
        if (ec != hipSuccess) return fail(std::string("hipStreamEndCapture: ") + hipGetErrorString(ec));
        hipGraphExec_t ex = nullptr;
        const hipError_t ei = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (ei != hipSuccess) return fail(std::string("hipGraphInstantiate: ") + hipGetErrorString(ei));
        it = w->graphs.emplace(key, ex).first;
    }
    TS_HIP(hipGraphLaunch(it->second, s));
    TS_HIP(hipMemcpyAsync(codes, w->codes_int.p, (size_t)B * H * 2 * sizeof(int64_t), hipMemcpyDeviceToDevice, s));
    return 0;
}

}  // extern "C"
