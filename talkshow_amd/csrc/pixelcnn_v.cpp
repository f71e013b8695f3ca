// GatedPixelCNN(..., bh_model=False) — the single-stack form of the code predictor (nets/spg/gated_pixelcnn_v2.py:25-150).
//
// With bh_model=False every GatedMaskedConv2d has a vertical kernel one column wide ((kernel // 2 + 1, 1), padding (kernel // 2, 0),
// :37-42) and its forward takes the `else` branch (:80-85): out_v = horiz_resid(gate(vert_stack(x_v) + class)) [+ x_v], out_h = out_v,
// and the logits come from x_v (:147-150).  The grid's columns never mix: position (r, j) sees rows < r of column j only (layer 0 is
// mask A: rows r-3 .. r-1; layers >= 1 see rows r-1, r of the layer below).  So all W columns of a code row are one batch of B * W
// independent rows here, a code row is 2 * n_layers + 3 dependent skinny_gemm launches behind a per-layer row cache (the previous
// input row of every layer >= 1: O(1) state), and the W codes of a row are drawn together — exactly what the reference's
// rows-major, column-by-column loop (:167-176) produces, because a column's logits do not depend on the row's other codes.
// No shipped config uses this form (config/body_pixel.json sets bh_model=true): launches are eager, weights row-major; the tuned
// chain of pixelcnn.cpp is the bh_model=true path.  audio=True adds embedding_aud + fusion_v between layers 0 and 1 (:137-141).
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "host_common.h"

using namespace ts;

struct ts_pixelcnn_v {
    ts_ctx *ctx = nullptr;
    int V = 0, D = 0, NL = 0, NC = 0, AD = 0, HID = 512;
    bool audio = false;
    DevBuf emb;                                        // (V, D)
    std::vector<std::unique_ptr<DevBuf>> Wg, bg, Wr, br, cls;   // per layer: gate conv [2D][taps*D], its bias, horiz_resid [D][D] + bias, class table (NC, 2D)
    DevBuf Wa, ba, Wf, bf;                             // embedding_aud [D][AD], fusion_v [D][2D]
    DevBuf W1, b1, W2, b2;                             // output_conv.0 / .2
    struct Work {
        DevBuf tok, X, G, T, AE, CLS, H1, LG;
        int capM = 0;
    };
    StreamWorks<Work> works;
};

namespace {

SkinnyParams problem(int M, int N, int epi) {
    SkinnyParams q;
    std::memset(&q, 0, sizeof(q));
    q.M = M;
    q.N = N;
    q.epi = epi;
    return q;
}
void dense(SkinnyParams &q, const float *base, long stride, int shift, int len) {
    SkinnySeg &s = q.seg[q.nseg++];
    s.base = base;
    s.row_stride = stride;
    s.row_shift = shift;
    s.len = len;
    q.Ktot += len;
}
void gather(SkinnyParams &q, const float *table, long stride, const int *gidx, int len) {
    SkinnySeg &s = q.seg[q.nseg++];
    s.base = table;
    s.gidx = gidx;
    s.row_stride = stride;
    s.gidx_stride = 1;
    s.len = len;
    q.Ktot += len;
}
int up(DevBuf &b, const std::vector<float> &v) { return b.upload(v.data(), v.size() * sizeof(float)); }

}  // namespace

extern "C" {

int ts_pixelcnn_v_create(ts_ctx *ctx, const ts_tensor *sd_, int n, int V, int D, int NL, int NC, int audio, int AD,
                         ts_pixelcnn_v **out) {
    if (!ctx || !sd_ || !out) return fail("ts_pixelcnn_v_create: null argument");
    if (V < 1 || D < 8 || D % 8 || NL < 1 || NC < 1 || (audio && (AD < 8 || AD % 8)))
        return fail("ts_pixelcnn_v_create: dim (and the audio width) must be multiples of 8");
    TS_HIP(hipSetDevice(ctx->device));
    StateDict sd(sd_, n);
    std::unique_ptr<ts_pixelcnn_v> p(new ts_pixelcnn_v());
    p->ctx = ctx;
    p->V = V; p->D = D; p->NL = NL; p->NC = NC; p->AD = AD; p->audio = audio != 0;
    const float *e = sd.get("embedding.weight", {V, D});
    if (!e) return 1;
    TS_TRY(p->emb.upload(e, (size_t)V * D * sizeof(float)));
    for (int l = 0; l < NL; ++l) {
        const std::string k = "layers." + std::to_string(l) + ".";
        const int kh = l == 0 ? 4 : 2, taps = l == 0 ? 3 : 2;     // mask A: the last kernel row is zeroed on every forward (:57-60)
        const float *w = sd.get(k + "vert_stack.weight", {2 * D, D, kh, 1}), *b = sd.get(k + "vert_stack.bias", {2 * D});
        const float *wr = sd.get(k + "horiz_resid.weight", {D, D, 1, 1}), *brr = sd.get(k + "horiz_resid.bias", {D});
        const float *c = sd.get(k + "class_cond_embedding.weight", {NC, 2 * D});
        if (!w || !b || !wr || !brr || !c) return 1;
        std::vector<float> g((size_t)2 * D * taps * D);
        for (int o = 0; o < 2 * D; ++o)
            for (int t = 0; t < taps; ++t)                          // tap t multiplies input row r - taps + t (+1 for layers >= 1)
                for (int i = 0; i < D; ++i) g[((size_t)o * taps + t) * D + i] = w[((size_t)o * D + i) * kh + t];
        p->Wg.emplace_back(new DevBuf()); TS_TRY(up(*p->Wg.back(), g));
        p->bg.emplace_back(new DevBuf()); TS_TRY(p->bg.back()->upload(b, (size_t)2 * D * sizeof(float)));
        p->Wr.emplace_back(new DevBuf()); TS_TRY(p->Wr.back()->upload(wr, (size_t)D * D * sizeof(float)));
        p->br.emplace_back(new DevBuf()); TS_TRY(p->br.back()->upload(brr, (size_t)D * sizeof(float)));
        p->cls.emplace_back(new DevBuf()); TS_TRY(p->cls.back()->upload(c, (size_t)NC * 2 * D * sizeof(float)));
    }
    if (p->audio) {
        const float *wa = sd.get("embedding_aud.weight", {D, AD, 1, 1}), *ba = sd.get("embedding_aud.bias", {D});
        const float *wf = sd.get("fusion_v.weight", {D, 2 * D, 1, 1}), *bf = sd.get("fusion_v.bias", {D});
        if (!wa || !ba || !wf || !bf) return 1;
        TS_TRY(p->Wa.upload(wa, (size_t)D * AD * sizeof(float)));
        TS_TRY(p->ba.upload(ba, (size_t)D * sizeof(float)));
        TS_TRY(p->Wf.upload(wf, (size_t)D * 2 * D * sizeof(float)));
        TS_TRY(p->bf.upload(bf, (size_t)D * sizeof(float)));
    }
    const float *w1 = sd.get("output_conv.0.weight", {p->HID, D, 1, 1}), *b1 = sd.get("output_conv.0.bias", {p->HID});
    const float *w2 = sd.get("output_conv.2.weight", {V, p->HID, 1, 1}), *b2 = sd.get("output_conv.2.bias", {V});
    if (!w1 || !b1 || !w2 || !b2) return 1;
    TS_TRY(p->W1.upload(w1, (size_t)p->HID * D * sizeof(float)));
    TS_TRY(p->b1.upload(b1, (size_t)p->HID * sizeof(float)));
    TS_TRY(p->W2.upload(w2, (size_t)V * p->HID * sizeof(float)));
    TS_TRY(p->b2.upload(b2, (size_t)V * sizeof(float)));
    *out = p.release();
    return 0;
}

void ts_pixelcnn_v_destroy(ts_pixelcnn_v *p) { delete p; }

int ts_pixelcnn_v_generate(ts_pixelcnn_v *p, const int64_t *label, const float *aud, int B, int H, int W, int mode,
                           const float *uniforms, uint64_t seed, int64_t clip0, int64_t *codes, float *logits,
                           const int64_t *pre_codes, const float *pre_aud, int H0, void *stream) {
    if (!p || !label || !codes) return fail("ts_pixelcnn_v_generate: null argument");
    if (p->audio && (!aud || (H0 > 0 && !pre_aud))) return fail("ts_pixelcnn_v_generate: this network was built with audio=True: audio rows required");
    if (B < 1 || H < 1 || H0 < 0 || (H0 > 0 && !pre_codes)) return fail("ts_pixelcnn_v_generate: bad shape / prefix");
    int wshift = 0;
    while ((1 << wshift) < W) ++wshift;
    if (W < 1 || (1 << wshift) != W || W > 64) return fail("ts_pixelcnn_v_generate: the grid width must be a power of two <= 64");
    if (mode == TS_SAMPLE_UNIFORMS && !uniforms) return fail("ts_pixelcnn_v_generate: uniforms required");
    if ((long)B * W > 4096) return fail("ts_pixelcnn_v_generate: B * W exceeds 4096 rows per stage (the chain kernels' limit): split the call");
    hipStream_t s = (hipStream_t)stream;
    ts_ctx *ctx = p->ctx;
    const int D = p->D, NL = p->NL, V = p->V, M = B * W, HID = p->HID;
    const size_t F = sizeof(float);
    ts_pixelcnn_v::Work &w = p->works.get(s);
    TS_TRY(w.tok.ensure((size_t)4 * M * sizeof(int)));                 // token ring: rows r-3 .. r, slot = row & 3, [slot][b * W + j]
    TS_TRY(w.X.ensure((size_t)(NL + 1) * 2 * M * D * F));             // X[l][row parity][M][D]: input of layer l (l = NL: the stack's output)
    TS_TRY(w.G.ensure((size_t)M * D * F));
    TS_TRY(w.T.ensure((size_t)M * D * F));
    TS_TRY(w.AE.ensure((size_t)B * D * F));
    TS_TRY(w.CLS.ensure((size_t)NL * B * 2 * D * F));
    TS_TRY(w.H1.ensure((size_t)M * HID * F));
    TS_TRY(w.LG.ensure((size_t)M * V * F));
    auto X = [&](int l, int r) { return w.X.f() + ((size_t)l * 2 + (r & 1)) * M * D; };
    {
        MiscScope ms(ctx, s);
        TS_HIP(hipMemsetAsync(w.tok.p, 0xff, (size_t)4 * M * sizeof(int), s));     // -1: rows above the grid gather the zero row
        for (int l = 0; l < NL; ++l)   // class_cond_embedding(label): one (B, 2D) block per layer, added per clip (rows m >> log2 W)
            TS_HIP(launch_gather_rows(p->cls[l]->f(), 2 * D, p->NC, label, 1, B, 2 * D, w.CLS.f() + (size_t)l * B * 2 * D, 2 * D, s));
    }
    const int Htot = H0 + H;
    for (int r = 0; r < Htot; ++r) {
        const bool gen = r >= H0;
        // ---- layer 0 (mask A): embeddings of the three rows above, gate, horiz_resid (no residual: `residual = False if i == 0`) ----
        {
            SkinnyParams q = problem(M, 2 * D, EPI_GATE);
            for (int t = 0; t < 3; ++t) gather(q, p->emb.f(), D, w.tok.i() + (size_t)((r - 3 + t) & 3) * M, D);
            q.W = p->Wg[0]->f(); q.ldw = 3 * D; q.bias = p->bg[0]->f();
            q.add1 = w.CLS.f(); q.add1_stride = 2 * D; q.add1_shift = wshift;
            q.gateD = D; q.out = w.G.f(); q.out_stride = D;
            TS_TRY(run_skinny(ctx, q, s));
            SkinnyParams u = problem(M, D, EPI_LINEAR);
            dense(u, w.G.f(), D, 0, D);
            u.W = p->Wr[0]->f(); u.ldw = D; u.bias = p->br[0]->f();
            u.out = p->audio ? w.T.f() : X(1 < NL ? 1 : NL, r); u.out_stride = D;
            if (NL == 1 && p->audio) u.out = w.T.f();
            TS_TRY(run_skinny(ctx, u, s));
        }
        // ---- audio fusion in front of layer 1 (:137-141): x_v = fusion_v(cat[x_v, embedding_aud(aud)]); one audio row per clip ----
        if (p->audio && NL > 1) {
            const float *arow = gen ? aud + (size_t)(r - H0) * p->AD : pre_aud + (size_t)r * p->AD;
            const long astride = (long)(gen ? H : H0) * p->AD;
            SkinnyParams a = problem(B, D, EPI_LINEAR);
            dense(a, arow, astride, 0, p->AD);
            a.W = p->Wa.f(); a.ldw = p->AD; a.bias = p->ba.f(); a.out = w.AE.f(); a.out_stride = D;
            TS_TRY(run_skinny(ctx, a, s));
            SkinnyParams f = problem(M, D, EPI_LINEAR);
            dense(f, w.T.f(), D, 0, D);
            dense(f, w.AE.f(), D, wshift, D);
            f.W = p->Wf.f(); f.ldw = 2 * D; f.bias = p->bf.f(); f.out = X(1, r); f.out_stride = D;
            TS_TRY(run_skinny(ctx, f, s));
        }
        // ---- layers >= 1 (mask B): rows r-1 and r of the layer's input, gate, horiz_resid + input ----
        for (int l = 1; l < NL; ++l) {
            SkinnyParams q = problem(M, 2 * D, EPI_GATE);
            dense(q, r > 0 ? X(l, r - 1) : nullptr, D, 0, D);          // row -1 is padding
            dense(q, X(l, r), D, 0, D);
            q.W = p->Wg[l]->f(); q.ldw = 2 * D; q.bias = p->bg[l]->f();
            q.add1 = w.CLS.f() + (size_t)l * B * 2 * D; q.add1_stride = 2 * D; q.add1_shift = wshift;
            q.gateD = D; q.out = w.G.f(); q.out_stride = D;
            TS_TRY(run_skinny(ctx, q, s));
            SkinnyParams u = problem(M, D, EPI_LINEAR);
            dense(u, w.G.f(), D, 0, D);
            u.W = p->Wr[l]->f(); u.ldw = D; u.bias = p->br[l]->f();
            u.add3 = X(l, r); u.add3_stride = D;
            u.out = X(l + 1, r); u.out_stride = D;
            TS_TRY(run_skinny(ctx, u, s));
        }
        // ---- logits of the W positions of this row, then their codes (prefix rows: the given codes, no head) ----
        if (gen) {
            const float *xf = NL == 1 ? (p->audio ? w.T.f() : X(NL, r)) : X(NL, r);
            SkinnyParams h1 = problem(M, HID, EPI_LINEAR);
            dense(h1, xf, D, 0, D);
            h1.W = p->W1.f(); h1.ldw = D; h1.bias = p->b1.f(); h1.relu = 1; h1.out = w.H1.f(); h1.out_stride = HID;
            TS_TRY(run_skinny(ctx, h1, s));
            SkinnyParams h2 = problem(M, V, EPI_LINEAR);
            dense(h2, w.H1.f(), HID, 0, HID);
            h2.W = p->W2.f(); h2.ldw = HID; h2.bias = p->b2.f(); h2.out = w.LG.f(); h2.out_stride = V;
            TS_TRY(run_skinny(ctx, h2, s));
        }
        for (int j = 0; j < W; ++j) {
            SampleParams sp;
            std::memset(&sp, 0, sizeof(sp));
            sp.logits = w.LG.f() + (size_t)j * V;
            sp.logit_stride = (long)W * V;
            sp.B = B;
            sp.V = V;
            sp.tok32 = w.tok.i() + (size_t)(r & 3) * M + j;
            sp.tok_stride = W;
            if (gen) {
                const int ro = r - H0;
                sp.mode = mode;
                sp.uniforms = uniforms ? uniforms + (size_t)ro * W + j : nullptr;
                sp.u_stride = (long)H * W;
                sp.seed = seed;
                sp.clip_index0 = clip0;
                sp.position = (uint32_t)(r * W + j);           // absolute grid position, prefix rows counted (as in ts_pixelcnn_generate)
                sp.codes = codes + (size_t)ro * W + j;
                sp.code_stride = (long)H * W;
                if (logits) {
                    sp.logits_copy = logits + ((size_t)ro * W + j) * V;
                    sp.copy_stride = (long)H * W * V;
                }
            } else {
                sp.mode = TS_TEACHER_FORCED;
                sp.codes = const_cast<int64_t *>(pre_codes) + (size_t)r * W + j;
                sp.code_stride = (long)H0 * W;
            }
            MiscScope ms(ctx, s);
            TS_HIP(launch_sample(sp, s));
        }
    }
    return 0;
}

}  // extern "C"
