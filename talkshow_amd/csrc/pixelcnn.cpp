// GatedPixelCNN (nets/spg/gated_pixelcnn_v2.py) as an INCREMENTAL, row-cached launch plan.
//
// The reference's generate() (:152-177) re-runs the whole 15-layer network over the whole H x 2 grid for each of
// the 2H positions.  Causality (:57-87: layer 0 is mask 'A'; vertical kernels look up, horizontal kernels look left)
// means position (r, j) only needs
//   * the vertical stack at row r: layer 0 sees code rows r-3..r-1, layer l>=1 sees its own input at rows r-1, r;
//   * the horizontal stack at columns <= j of row r.
// So per row we run the vertical stack once for both columns, then the horizontal chain for column 0 (body code),
// sample, then column 1 (hand code), sample: 216x fewer FLOPs, every conv tap multiplied exactly once in fp32.
//
// Row-shifted partial sums (keeps every stage's K at <= 2*dim, i.e. its weight slice small and its MFMA burst short):
//   h_vert_l(r) = Wcur_l . XV_l[r] + ( Wprev_l . XV_l[r-1] + b )         the bracket is computed one row EARLIER, in the
//   h_vert_0(r) = W2 . E[r-1] + ( W1 . E[r-2] ) + ( W0 . E[r-3] ) + b     same launch that first sees XV_l[r-1] / E[..]
//
// Launch plan of row r (every entry is ONE skinny_gemm launch over all clips of the pass; "|" separates independent
// problems that share a launch):
//   V0        v0.gate | v0.Q1 | v0.Q0                     gate: OV0 = gate(W2.E[r-1] + Q1[r] + Q0[r] + b + c0)
//   V1        v1.gate | v1.P | v2h_0                      layer 1 reads OV0 directly: fusion_v[:, :D] (gated_pixelcnn_v2.py:137-144)
//                                                         is composed into its two tap matrices on the host, the audio half
//                                                         of the fusion becomes two per-row additive terms precomputed for
//                                                         all rows (XV_1 is never materialised)
//   V_l       v_l.gate | v_l.P | v2h_{l-1}                l = 2..NL-1
//   V_NL      v2h_{NL-1}
//   column j: hg_0, S_1 .. S_{NL-1}, head1', head2, sample
//
// Horizontal chain, one launch per layer.  In the reference a layer is gate(horiz_stack(x_h) + ...) followed by
// horiz_resid (+ x_h), i.e. two dependent linear maps per layer; consecutive linear maps are composed on the host
// (products in fp64, rounded once to fp32 — the same kind of fold as BatchNorm into a conv):
//   S_l   A: XH_l[j]  = Rw_l . G_{l-1} + rb_l + XH_{l-1}[j]                               (Rw_l = horiz_resid_{l-1};
//         B: G_l      = gate( (Wh1_l.Rw_l) . G_{l-1} + Wh1_l . XH_{l-1}[j] + bm_l          l = 1 also folds fusion_h)
//                             + V2H_l[j] + c_l  [+ Wh0_l . XH_l[0] for j = 1, precomputed while column 0 ran] )
//   head1'   relu( (W1.Wr_{NL-1}) . G_{NL-1} + W1 . XH_{NL-1}[j] + b )
// A and B only need (G_{l-1}, XH_{l-1}) so they share a launch: NL + 2 dependent launches per column instead of 2*NL + 3.
// The vertical stack of a row and the column-0 chain of the same row are independent except for V2H_l (ready after
// V_{l+1}); V_k rides in the launch of the column-0 stage that runs one step behind it: V0, V1, then (hg_0 | V2),
// (S_1 | V3), ... = 2 + (NL + 2) + (NL + 2) = 36 skinny launches per row for NL = 15, plus the 2 sampler launches
// (a per-op port would need 2 + 17 + 66).
//
// Rows are absolute indices: a one-shot call runs rows 0..H-1 (after an optional known prefix), a streaming session
// (ts_pixelcnn_stream_*) continues at the row where its previous step stopped, on a row cache that persists in its Work.
#include <algorithm>
#include <cstdlib>
#include <deque>
#include <set>
#include <tuple>

#include "host_common.h"
#include "skinny_desc.h"

using namespace ts;

struct ts_pixelcnn {
    ts_ctx *ctx = nullptr;
    int V = 0, D = 0, NL = 0, NC = 0, AD = 0, HID = 512;
    DevBuf emb;                                              // [V][D]
    std::vector<std::vector<std::unique_ptr<DevBuf>>> wvt;   // vertical taps: layer 0 {W0,W1,W2}, l>=1 {Wprev,Wcur}; each [4D][2D]
    std::vector<std::unique_ptr<DevBuf>> bv;                 // [2*2D] (duplicated per column)
    std::vector<std::unique_ptr<DevBuf>> wv2h, bv2h, wh, bh, cls, wr, br;
    DevBuf fha;                                              // fusion_h[:, :D]  [D][D] (fusion_v's half lives composed in wv1c / wv1p)
    DevBuf wv1c, wv1p;                                       // layer 1 vertical taps with fusion_v[:, :D] composed in  [4D][2D]
    ConvLayer aud_v1c, aud_v1p;                              // the same taps applied to the audio half of the fusion   [4D][D]
    ConvLayer aud_embed, aud_fv, aud_fh;                     // embedding_aud ; fusion_{v,h}[:, D:] (+ fusion bias)
    DevBuf w1, b1, w2, b2;                                   // output_conv
    // composed horizontal maps (see the header): per layer l >= 1
    std::vector<std::unique_ptr<DevBuf>> rw, rb;             // A: Rw_l [D][D], rb_l [D]        (index l, entry 0 unused)
    std::vector<std::unique_ptr<DevBuf>> wB, bm;             // B: [2D][D or 2D], bm_l [2D]
    DevBuf w1m, b1m;                                         // head1': [HID][D or 2D], [HID]
    ConvLayer aud_h1;                                        // Wh1_1 applied to AEH for every row (l = 1 has AEH in place of XH_0)
    bool use_graph = true;
    int defer_p = -1;     // the next-row projections P_l of the vertical stack ride with column 1's launches: -1 auto, 0 never, 1 always
    // tiled operand layouts (kernels.h, SkinnyParams::w_tiled): every weight matrix of the chain gets a tiled twin, and
    // the activation buffers that feed the next stage's GEMM (OV0, XV, HV, G, XH, Y) are written / read tiled
    bool tiled = false;
    struct TiledW {
        DevBuf buf;
        int K = 0, epi = 0;
    };
    std::map<const float *, std::unique_ptr<TiledW>> wtiled;
    // Work set: every buffer the row loop touches + the hipGraph replays of it.  One per stream, so that independent
    // batches can be in flight on different streams against the single weight copy above.
    struct Work {
        int capB = 0, capH = 0;
        DevBuf aud_all, AE, AEV, AEH, AEH1, AV1C, AV1P, tok32, CR, XV, OV0, OVlast, HV, V2H, P, Q1, Q0, XH, G, T0, Y, LG, tfcodes;
        // every pointer inside the captured kernels is one of the buffers above or the staging buffers below, so a graph
        // is valid for any caller pointers; key = (B, H, H0, mode)
        DevBuf codes_int, unif_int, dyn;   // dyn: {seed, clip0, position base} of the call being replayed, written by a kernel ahead of it
        DevBuf cAEH, cAEH1, cAV1C, cAV1P;  // the audio terms of ONE chunk of rows, compact (chunked one-shot calls: see run_chunked)
        hipStream_t cap_stream = nullptr;
        // Captured graphs, least recently used out first: at most GRAPH_CAP per Work.  Keys: (B, H, H0, mode) = a whole one-shot call;
        // (B, Hc, -(1 + phase), mode) = Hc rows of a chunked one-shot call; (B, Hc, 1000 + phase, mode) = a streaming step.
        // At most GRAPH_CAP unpinned graphs + PIN_CAP pinned ones per Work.
        typedef std::tuple<int, int, int, int> Key;
        struct Entry {
            hipGraphExec_t exec;
            uint64_t used;
        };
        static constexpr size_t GRAPH_CAP = 24;
        static constexpr size_t PIN_CAP = 12;                 // shapes a host may pin with ts_pixelcnn_prepare (they are never evicted)
        static constexpr size_t RECENT = 16;                  // window of one-shot calls a shape must recur in to count as hot
        std::map<Key, Entry> graphs;
        std::map<Key, std::pair<long, double>> graph_stats;   // skinny launches, flops
        std::set<Key> pinned;
        std::deque<Key> recent;                               // the last RECENT one-shot (H0 = 0) shapes that found no whole-call graph
        // graphs taken out of the cache while a replay of them may still be queued on the stream: destroyed once the event recorded
        // behind them has completed (no host-side wait on the serving stream)
        struct Retired {
            hipGraphExec_t exec;
            hipEvent_t done;
        };
        std::vector<Retired> retired;
        uint64_t tick = 0;
        long captures = 0;                                    // graphs captured + instantiated on this stream so far
        // A one-shot shape gets its own whole-call graph (~36 H nodes: tens of milliseconds to capture and instantiate) only when it is
        // HOT: pinned by the host, or met for the third time within the last RECENT one-shot calls on this stream.  A pass over a
        // dataset of many distinct lengths (scripts/test_body.py:113-194), however often it is repeated, never qualifies and keeps running
        // on the length-independent chunk graphs; a serving loop on one or a few shapes qualifies on its third call.
        bool hot(const Key &k) {
            if (pinned.count(k)) return true;
            int n = 0;
            for (const Key &r : recent) n += r == k;
            recent.push_back(k);
            if (recent.size() > RECENT) recent.pop_front();
            return n >= 2;
        }
        void reap(bool wait) {
            size_t keep = 0;
            for (size_t i = 0; i < retired.size(); ++i) {
                Retired &r = retired[i];
                if (wait) (void)hipEventSynchronize(r.done);
                if (wait || hipEventQuery(r.done) == hipSuccess) {
                    (void)hipGraphExecDestroy(r.exec);
                    (void)hipEventDestroy(r.done);
                } else {
                    retired[keep++] = r;
                }
            }
            retired.resize(keep);
        }
        void drop_graphs() {
            reap(true);
            for (auto &kv : graphs) (void)hipGraphExecDestroy(kv.second.exec);
            graphs.clear();
            graph_stats.clear();            // `pinned` stays: a pinned shape whose graph went with a buffer growth is re-captured by its next call
        }
        // Two classes share the cache: whole-call graphs of hot one-shot shapes (key field 2 = H0 in [0, 1000): at most WHOLE_CAP
        // unpinned ones) and the length-independent chunk / streaming-step graphs (at most GRAPH_CAP - WHOLE_CAP) — a burst of hot
        // shapes never pushes out the chunk graphs every other call falls back on.  Room for one more graph of `key`'s class: the least
        // recently used unpinned graph of that class leaves the cache; it is destroyed behind an event on `s`.
        static constexpr size_t WHOLE_CAP = 8;
        static bool whole(const Key &k) { return std::get<2>(k) >= 0 && std::get<2>(k) < 1000; }
        int make_room(hipStream_t s, const Key &key) {
            reap(false);
            const bool cls = whole(key);
            const size_t cap = cls ? WHOLE_CAP : GRAPH_CAP - WHOLE_CAP;
            for (;;) {
                size_t n = 0;
                auto lru = graphs.end();
                for (auto it = graphs.begin(); it != graphs.end(); ++it) {
                    if (whole(it->first) != cls || pinned.count(it->first)) continue;
                    ++n;
                    if (lru == graphs.end() || it->second.used < lru->second.used) lru = it;
                }
                if (n < cap) break;
                Retired r{lru->second.exec, nullptr};
                TS_HIP(hipEventCreateWithFlags(&r.done, hipEventDisableTiming));
                TS_HIP(hipEventRecord(r.done, s));
                retired.push_back(r);
                graph_stats.erase(lru->first);
                graphs.erase(lru);
            }
            if (retired.size() > 2 * GRAPH_CAP) reap(true);   // a host that never lets the stream drain: bounded all the same
            return 0;
        }
        ~Work() {
            drop_graphs();
            if (cap_stream) (void)hipStreamDestroy(cap_stream);
        }
    };
    StreamWorks<Work> works;
    Work &work(hipStream_t s) { return works.get(s); }
};

// f3: a generation session whose row cache (one previous row per layer + the layer-0 partial sums + the last code rows)
// persists across calls — O(1) state for any history length (the receptive field is 17 code rows)
struct ts_pixelcnn_stream {
    ts_pixelcnn *p = nullptr;
    int B = 0;
    long rows = 0;              // code rows generated so far = absolute index of the next row
    int max_rows = 0;           // largest chunk a step may bring (buffers are sized once, at open)
    ts_pixelcnn::Work w;
    DevBuf label;
};

namespace {

int upload_vec(std::vector<std::unique_ptr<DevBuf>> &dst, const std::vector<float> &v) {
    dst.emplace_back(new DevBuf());
    return dst.back()->upload(v.data(), v.size() * sizeof(float));
}

// rows per graph of a chunked one-shot call: a multiple of 4 (the period of the row rings), so that every chunk after the first
// starts in the same buffer phase
constexpr int CHUNK_ROWS = 8;

int ensure_work(ts_pixelcnn *p, ts_pixelcnn::Work *w, int B, int Htot) {
    if (B <= w->capB && Htot <= w->capH) return 0;
    const int cb = std::max(B, w->capB), ch = std::max(Htot, w->capH);
    const size_t D = p->D, NL = p->NL, f = sizeof(float);
    const size_t cb16 = round_up(cb, 16);   // tiled buffers hold whole 16-row blocks
    TS_TRY(w->aud_all.ensure((size_t)cb * ch * p->AD * f));
    TS_TRY(w->AE.ensure((size_t)cb * ch * D * f));
    TS_TRY(w->AEV.ensure((size_t)cb * ch * D * f));
    TS_TRY(w->AEH.ensure((size_t)cb * ch * D * f));
    TS_TRY(w->tok32.ensure((size_t)cb * ch * 2 * sizeof(int)));
    TS_TRY(w->CR.ensure(NL * cb * 2 * D * f));
    TS_TRY(w->XV.ensure(NL * 2 * cb16 * 2 * D * f));
    TS_TRY(w->OV0.ensure(cb16 * 2 * D * f));
    TS_TRY(w->OVlast.ensure((size_t)cb * 2 * D * f));
    TS_TRY(w->HV.ensure(NL * cb16 * 4 * D * f));
    TS_TRY(w->V2H.ensure(NL * cb * 4 * D * f));
    TS_TRY(w->P.ensure(NL * 2 * cb * 4 * D * f));
    TS_TRY(w->Q1.ensure((size_t)4 * cb * 4 * D * f));
    TS_TRY(w->Q0.ensure((size_t)4 * cb * 4 * D * f));
    TS_TRY(w->XH.ensure((NL + 1) * 2 * cb16 * D * f));
    TS_TRY(w->G.ensure((size_t)2 * cb16 * D * f));
    TS_TRY(w->T0.ensure(NL * cb * 2 * D * f));
    TS_TRY(w->AEH1.ensure((size_t)cb * ch * 2 * D * f));
    TS_TRY(w->AV1C.ensure((size_t)cb * ch * 4 * D * f));
    TS_TRY(w->AV1P.ensure((size_t)cb * ch * 4 * D * f));
    TS_TRY(w->Y.ensure(cb16 * p->HID * f));
    TS_TRY(w->LG.ensure((size_t)cb * p->V * f));
    TS_TRY(w->tfcodes.ensure((size_t)cb * ch * 2 * sizeof(int64_t)));
    TS_TRY(w->codes_int.ensure((size_t)cb * ch * 2 * sizeof(int64_t)));
    TS_TRY(w->unif_int.ensure((size_t)cb * ch * 2 * sizeof(float)));
    TS_TRY(w->dyn.ensure(3 * sizeof(uint64_t)));
    TS_TRY(w->cAEH.ensure((size_t)cb * CHUNK_ROWS * D * f));
    TS_TRY(w->cAEH1.ensure((size_t)cb * CHUNK_ROWS * 2 * D * f));
    TS_TRY(w->cAV1C.ensure((size_t)cb * CHUNK_ROWS * 4 * D * f));
    TS_TRY(w->cAV1P.ensure((size_t)cb * CHUNK_ROWS * 4 * D * f));
    w->drop_graphs();   // buffers moved: captured pointers are stale
    w->capB = cb;
    w->capH = ch;
    return 0;
}

// Rows are ABSOLUTE indices of the session / call.  One-shot call: rows 0..Htot-1, the first H0 of them a known prefix.
// Streaming step: rows r0..r0+Hc-1 behind a row cache left by the previous steps.
struct RunCfg {
    int B, H, H0, Htot, mode;
    const float *uniforms;
    uint64_t seed;
    int64_t clip0;
    int64_t *codes;        // (B,H,2)
    float *logits;         // (B,H,2,V) or null
    const uint64_t *dyn;   // device {seed, clip0} (graph replay) or null
    ts_pixelcnn::Work *w;
    int R;                 // rows of the token ring tok32[B][R][2]: row rr lives at rr % R (one-shot: R = Htot, no wrap)
    int aud_r0, aud_rows;  // the audio-term buffers AEV / AEH / AEH1 hold rows [aud_r0, aud_r0 + aud_rows)
    int out_r0;            // first row written to `codes` / `logits` / read from `uniforms` (those arrays hold H rows)
    int pos_r0;            // Philox position of row r, column j = pos_base + (r - pos_r0) * 2 + j
    long pos_base;         // (added on the device from a dynamic word when a captured graph is replayed)
    int last_row;          // rows >= last_row are never generated: look-ahead partial sums for them are skipped
    // where the audio terms of rows [aud_r0, aud_r0 + aud_rows) live (the Work's whole-call buffers unless a chunk brings its own)
    const float *aeh = nullptr, *aeh1 = nullptr, *av1c = nullptr, *av1p = nullptr;
    // the caller's codes / uniforms arrays hold out_H rows per clip, of which this run fills rows out_row0 .. out_row0 + H - 1
    int out_H = 0, out_row0 = 0;
    void audio_from(ts_pixelcnn::Work *wk) { aeh = wk->AEH.f(), aeh1 = wk->AEH1.f(), av1c = wk->AV1C.f(), av1p = wk->AV1P.f(); }
};
inline RunCfg one_shot_cfg(int B, int H, int H0, int mode, const float *uniforms, uint64_t seed, int64_t clip0, int64_t *codes,
                           float *logits, ts_pixelcnn::Work *w) {
    const int Htot = H0 + H;
    // Philox position of code (r, j) = 2 r + j with r counted from the first PREFIX row: a call that continues another one
    // behind its codes (`infer(chunk1, pre_latents=chunk0 codes)`) draws the next numbers of the stream, not chunk 0's again
    RunCfg c{B, H, H0, Htot, mode, uniforms, seed, clip0, codes, logits, nullptr, w, Htot, 0, Htot, H0, 0, 0, Htot};
    c.audio_from(w);
    return c;
}

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

struct Slot {   // one launch: up to SKINNY_MAX_PROBLEMS independent problems
    SkinnyParams p[SKINNY_MAX_PROBLEMS];
    int n = 0;
    void add(const SkinnyParams &q) { p[n++] = q; }
};

// the tiled view width of the work buffer that holds `ptr`, or 0 (row-major buffers / foreign pointers)
int tiled_width_of(const ts_pixelcnn *p, const ts_pixelcnn::Work *w, const void *ptr) {
    if (!p->tiled || !ptr) return 0;
    auto in = [&](const DevBuf &b) { return ptr >= b.p && ptr < static_cast<const char *>(b.p) + b.bytes; };
    if (in(w->OV0) || in(w->XV) || in(w->HV)) return 2 * p->D;     // HV is [B][4D], read by v2h as [2B][2D]
    if (in(w->G) || in(w->XH)) return p->D;
    if (in(w->Y)) return p->HID;
    return 0;
}

int launch_slot(ts_pixelcnn *p, ts_pixelcnn::Work *w, const Slot &a, const Slot *b, hipStream_t s) {
    // workgroups are dispatched in problem order: the rider (vertical stack: the big K = 512 problems) goes first, so that
    // whatever does not fit the first round of workgroups is the cheap end of the launch
    SkinnyParams q[SKINNY_MAX_PROBLEMS];
    const SkinnyParams *ps[SKINNY_MAX_PROBLEMS];
    int n = 0;
    if (b)
        for (int i = 0; i < b->n; ++i) q[n++] = b->p[i];
    for (int i = a.n - 1; i >= 0; --i) q[n++] = a.p[i];
    for (int i = 0; i < n; ++i) {
        if (p->tiled) {   // operands by identity: tiled weight twin, tiled activation buffers (see ts_pixelcnn::tiled)
            auto it = p->wtiled.find(q[i].W);
            if (it != p->wtiled.end() && it->second->K == q[i].Ktot && it->second->epi == q[i].epi) {
                q[i].W = it->second->buf.f();
                q[i].w_tiled = q[i].Ktot / 16;
            }
            for (int k = 0; k < q[i].nseg; ++k)
                if (!q[i].seg[k].gidx) q[i].seg[k].tiled_w = tiled_width_of(p, w, q[i].seg[k].base);
            q[i].out_tiled_w = tiled_width_of(p, w, q[i].out);
            q[i].pre_tiled_w = tiled_width_of(p, w, q[i].pre);
            q[i].add1_tiled_w = tiled_width_of(p, w, q[i].add1);
        }
        ps[i] = &q[i];
    }
    return run_skinny_batch(p->ctx, ps, n, s);
}

// vertical stack + v->h projections of row r as NL+2 launch slots
// `deferred` (optional): the P_l problems (only read by row r+1) are handed back instead of riding in slot V_l
void build_vertical(ts_pixelcnn *p, const RunCfg &c, int r, std::vector<Slot> &out, std::vector<SkinnyParams> *deferred = nullptr) {
    const int B = c.B, D = p->D, NL = p->NL, R = c.R;
    ts_pixelcnn::Work *w = c.w;
    const int *tok = w->tok32.i();
    const size_t Bp = round_up(B, 16);   // slices of the tiled buffers are whole 16-row blocks apart
    auto XV = [&](int l, int par) { return w->XV.f() + ((size_t)(l * 2 + par) * Bp) * 2 * D; };
    auto HV = [&](int l) { return w->HV.f() + (size_t)l * Bp * 4 * D; };
    auto V2H = [&](int l) { return w->V2H.f() + (size_t)l * B * 4 * D; };
    auto P = [&](int l, int par) { return w->P.f() + ((size_t)(l * 2 + par) * B) * 4 * D; };
    auto Q = [&](DevBuf &q, int row) { return q.f() + (size_t)(row & 3) * B * 4 * D; };
    auto CR = [&](int l) { return w->CR.f() + (size_t)l * B * 2 * D; };

    auto gate_common = [&](SkinnyParams &q, int l) {
        q.ldw = 2 * D;
        q.clsrow = CR(l);
        q.cls_ld = 2 * D;
        q.gateD = D;
        q.out = l == 0 ? w->OV0.f() : (l + 1 < NL ? XV(l + 1, r & 1) : w->OVlast.f());
        q.out_stride = 2 * D;
        q.pre = HV(l);
        q.pre_stride = 4 * D;
    };
    auto emb_row = [&](SkinnyParams &q, int rr) {   // embeddings of both codes of row rr >= 0
        for (int col = 0; col < 2; ++col) add_gather(q, p->emb.f(), D, tok + (size_t)(rr % R) * 2 + col, (long)R * 2, D);
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

    out.clear();
    {   // V0: layer 0 (mask 'A', kernel rows 0..2 <-> code rows r-3..r-1): newest row here, older rows via Q1 / Q0.
        // Rows above the grid contribute nothing: their terms are simply left out (no zero-filled buffers to rely on).
        Slot s;
        SkinnyParams g = base_params(B, 4 * D, EPI_GATE);
        if (r >= 1) emb_row(g, r - 1);
        else add_dense(g, nullptr, 0, 0, std::min(128, 2 * D));   // top row: bias + conditioning only (a zero segment keeps it on the descriptor kernel)
        g.W = p->wvt[0][2]->f();
        g.bias = p->bv[0]->f();
        if (r >= 2) {
            g.add1 = Q(w->Q1, r);      // W1 . E[r-2], computed by the launch of row r-1
            g.add1_stride = 4 * D;
        }
        if (r >= 3) {
            g.add2 = Q(w->Q0, r);      // W0 . E[r-3], computed by the launch of row r-2
            g.add2_stride = 4 * D;
        }
        gate_common(g, 0);
        s.add(g);
        for (int t = 1; t >= 0 && r >= 1; --t) {   // row r-1's codes as seen from row r+1 (kernel row 1) and r+2 (kernel row 0)
            const int target = r + (2 - t);
            if (target >= c.last_row) continue;
            SkinnyParams q = base_params(B, 4 * D, EPI_LINEAR);
            emb_row(q, r - 1);
            q.W = p->wvt[0][t]->f();
            q.ldw = 2 * D;
            q.out = Q(t == 1 ? w->Q1 : w->Q0, target);
            q.out_stride = 4 * D;
            s.add(q);
        }
        out.push_back(s);
    }
    if (NL == 1) {
        Slot s;
        s.add(make_v2h(0));
        out.push_back(s);
        return;
    }
    for (int l = 1; l < NL; ++l) {   // V_l: v_l.gate | v_l.P | v2h_{l-1}
        Slot s;
        // layer 1 reads the gate output of layer 0 through the composed taps (audio fusion folded in), l >= 2 its own input row
        const float *xin = l == 1 ? w->OV0.f() : XV(l, r & 1);
        const size_t arow = (size_t)(r - c.aud_r0) * 4 * D;
        const long astride = (long)c.aud_rows * 4 * D;
        SkinnyParams g = base_params(B, 4 * D, EPI_GATE);
        add_dense(g, xin, 2 * D, 0, 2 * D);
        g.W = l == 1 ? p->wv1c.f() : p->wvt[l][1]->f();
        if (r > 0) {
            g.add1 = P(l, r & 1);          // Wprev . XV_l[r-1] + bias, computed while row r-1 ran
            g.add1_stride = 4 * D;
        } else {
            g.bias = p->bv[l]->f();        // nothing above the first row
        }
        if (l == 1) {
            g.add2 = c.av1c + arow;        // Wcur_1 . [AEV[r] | AEV[r]]
            g.add2_stride = astride;
        }
        gate_common(g, l);
        s.add(g);
        if (r + 1 < c.last_row) {
            SkinnyParams q = base_params(B, 4 * D, EPI_LINEAR);
            add_dense(q, xin, 2 * D, 0, 2 * D);
            q.W = l == 1 ? p->wv1p.f() : p->wvt[l][0]->f();
            q.ldw = 2 * D;
            q.bias = p->bv[l]->f();
            if (l == 1) {
                q.add1 = c.av1p + arow;        // Wprev_1 . [AEV[r] | AEV[r]]
                q.add1_stride = astride;
            }
            q.out = P(l, (r + 1) & 1);
            q.out_stride = 4 * D;
            if (deferred) deferred->push_back(q);
            else s.add(q);
        }
        s.add(make_v2h(l - 1));
        out.push_back(s);
    }
    {
        Slot s;
        s.add(make_v2h(NL - 1));
        out.push_back(s);
    }
}

// horizontal chain + head of position (r, j): NL + 2 launch slots (the sampler launch follows separately)
void build_horizontal(ts_pixelcnn *p, const RunCfg &c, int r, int j, std::vector<Slot> &out) {
    const int B = c.B, D = p->D, NL = p->NL, R = c.R;
    ts_pixelcnn::Work *w = c.w;
    const int *tok = w->tok32.i();
    auto V2H = [&](int l) { return w->V2H.f() + (size_t)l * B * 4 * D + (size_t)j * 2 * D; };
    const size_t Bp = round_up(B, 16);
    auto XH = [&](int l, int col) { return w->XH.f() + ((size_t)(l * 2 + col) * Bp) * D; };
    auto CR = [&](int l) { return w->CR.f() + (size_t)l * B * 2 * D; };
    auto G = [&](int l) { return w->G.f() + (size_t)(l & 1) * Bp * D; };
    auto T0 = [&](int l) { return w->T0.f() + (size_t)l * B * 2 * D; };
    auto make_t0 = [&](int l) {   // column 0 only: Wh0_l . XH_l[0], consumed by column 1's S_l
        SkinnyParams t = base_params(B, 2 * D, EPI_LINEAR);
        add_dense(t, XH(l, 0), D, 0, D);
        t.W = p->wh[l]->f();      // tap 0 block
        t.ldw = 2 * D;
        t.out = T0(l);
        t.out_stride = 2 * D;
        return t;
    };
    out.clear();
    {   // hg_0 — mask 'A': only the column to the left, i.e. the embedding of the code just sampled
        Slot s;
        SkinnyParams q = base_params(B, 2 * D, EPI_GATE);
        if (j == 1) add_gather(q, p->emb.f(), D, tok + (size_t)(r % R) * 2 + 0, (long)R * 2, D);
        else add_dense(q, nullptr, 0, 0, std::min(128, 2 * D));   // column 0 has nothing to its left: a block of zero rows keeps the launch on
                                                 // the descriptor kernel (K = 0 would drop the whole launch to the generic one)
        q.W = p->wh[0]->f();
        q.ldw = 2 * D;
        q.bias = p->bh[0]->f();
        q.add1 = V2H(0);
        q.add1_stride = 4 * D;
        q.clsrow = CR(0);
        q.cls_ld = 2 * D;
        q.gateD = D;
        q.out = G(0);
        q.out_stride = D;
        s.add(q);
        out.push_back(s);
    }
    for (int l = 1; l < NL; ++l) {   // S_l
        Slot s;
        SkinnyParams a = base_params(B, D, EPI_LINEAR);
        add_dense(a, G(l - 1), D, 0, D);
        a.W = p->rw[l]->f();
        a.ldw = D;
        a.bias = p->rb[l]->f();
        if (l == 1) {
            a.add1 = c.aeh + (size_t)(r - c.aud_r0) * D;
            a.add1_stride = (long)c.aud_rows * D;
        } else {
            a.add1 = XH(l - 1, j);
            a.add1_stride = D;
        }
        a.out = XH(l, j);
        a.out_stride = D;
        s.add(a);

        SkinnyParams g = base_params(B, 2 * D, EPI_GATE);
        add_dense(g, G(l - 1), D, 0, D);
        if (l >= 2) add_dense(g, XH(l - 1, j), D, 0, D);
        g.W = p->wB[l]->f();
        g.ldw = g.Ktot;
        g.bias = p->bm[l]->f();
        g.add1 = V2H(l);
        g.add1_stride = 4 * D;
        if (l == 1) {
            g.add2 = c.aeh1 + (size_t)(r - c.aud_r0) * 2 * D;
            g.add2_stride = (long)c.aud_rows * 2 * D;
        }
        if (j == 1) {
            g.add3 = T0(l);
            g.add3_stride = 2 * D;
        }
        g.clsrow = CR(l);
        g.cls_ld = 2 * D;
        g.gateD = D;
        g.out = G(l);
        g.out_stride = D;
        s.add(g);
        if (j == 0 && l >= 2) s.add(make_t0(l - 1));
        out.push_back(s);
    }
    {   // head1' = relu(output_conv.0(XH_NL)) with XH_NL = horiz_resid_{NL-1}(G_{NL-1}) + XH_{NL-1} composed in
        Slot s;
        SkinnyParams h1 = base_params(B, p->HID, EPI_LINEAR);
        add_dense(h1, G(NL - 1), D, 0, D);
        if (NL >= 2) add_dense(h1, XH(NL - 1, j), D, 0, D);
        h1.W = p->w1m.f();
        h1.ldw = h1.Ktot;
        h1.bias = p->b1m.f();
        h1.relu = 1;
        h1.out = w->Y.f();
        h1.out_stride = p->HID;
        s.add(h1);
        if (j == 0 && NL >= 2) s.add(make_t0(NL - 1));
        out.push_back(s);
    }
    {
        Slot s;
        SkinnyParams h2 = base_params(B, p->V, EPI_LINEAR);
        add_dense(h2, w->Y.f(), p->HID, 0, p->HID);
        h2.W = p->w2.f();
        h2.ldw = p->HID;
        h2.bias = p->b2.f();
        h2.out = w->LG.f();
        h2.out_stride = p->V;
        s.add(h2);
        out.push_back(s);
    }
}

int launch_sampler(ts_pixelcnn *p, const RunCfg &c, int r, int j, hipStream_t s) {
    ts_pixelcnn::Work *w = c.w;
    SampleParams sp;
    std::memset(&sp, 0, sizeof(sp));
    const int ro = r - c.out_r0;   // row in the caller's (B,H,2) arrays
    sp.logits = w->LG.f();
    sp.B = c.B;
    sp.V = p->V;
    sp.mode = c.mode;
    sp.uniforms = c.uniforms ? c.uniforms + (size_t)ro * 2 + j : nullptr;
    sp.u_stride = (long)c.H * 2;
    sp.seed = c.seed;
    sp.clip_index0 = c.clip0;
    sp.dyn = c.dyn;
    sp.position = (uint32_t)((r - c.pos_r0) * 2 + j + (c.dyn ? 0 : c.pos_base));
    sp.tok32 = w->tok32.i() + (size_t)(r % c.R) * 2 + j;
    sp.tok_stride = (long)c.R * 2;
    sp.codes = c.codes + (size_t)ro * 2 + j;
    sp.code_stride = (long)c.H * 2;
    if (c.logits) {
        sp.logits_copy = c.logits + ((size_t)ro * 2 + j) * p->V;
        sp.copy_stride = (long)c.H * 2 * p->V;
    }
    MiscScope ms(p->ctx, s);
    TS_HIP(launch_sample(sp, s));
    return 0;
}

int run_row(ts_pixelcnn *p, const RunCfg &c, int r, bool need_h, hipStream_t s) {
    std::vector<Slot> V, H;
    // Up to 128 clips the column-0 launches (vertical slot riding with the horizontal one) come to 272 workgroups at every
    // tile shape — 16 more than CUs, and the 16 doubled-up CUs set the launch time (5.9 vs 3.6 us at 32 clips).  The P_l
    // projections are only read by the next row: there they ride with column 1's launches (96 -> 160 workgroups).
    std::vector<SkinnyParams> Pd;
    const bool defer = need_h && (p->defer_p < 0 ? c.B <= 128 : p->defer_p > 0);
    build_vertical(p, c, r, V, defer ? &Pd : nullptr);
    if (!need_h) {
        for (auto &sl : V) TS_TRY(launch_slot(p, c.w, sl, nullptr, s));
        return 0;
    }
    build_horizontal(p, c, r, 0, H);
    size_t vi = 0;
    // hg_0 needs V2H_0 (V1); S_k needs V2H_k (V_{k+1}): V0, V1, H0 + V2, H1 + V3, ...
    for (; vi < 2 && vi < V.size(); ++vi) TS_TRY(launch_slot(p, c.w, V[vi], nullptr, s));
    for (size_t k = 0; k < H.size(); ++k) {
        const Slot *ride = nullptr;
        if (vi < V.size() && V[vi].n + H[k].n <= SKINNY_MAX_PROBLEMS) ride = &V[vi++];
        TS_TRY(launch_slot(p, c.w, H[k], ride, s));
    }
    for (; vi < V.size(); ++vi) TS_TRY(launch_slot(p, c.w, V[vi], nullptr, s));   // only if the stack outlasts the chain
    TS_TRY(launch_sampler(p, c, r, 0, s));
    build_horizontal(p, c, r, 1, H);
    size_t pi = 0;
    for (size_t k = 0; k < H.size(); ++k) {
        Slot ride;
        // the last horizontal slots take what is left, so that every P_l has run before the row ends
        while (pi < Pd.size() && H[k].n + ride.n < SKINNY_MAX_PROBLEMS && (ride.n == 0 || Pd.size() - pi > H.size() - 1 - k)) ride.add(Pd[pi++]);
        TS_TRY(launch_slot(p, c.w, H[k], ride.n ? &ride : nullptr, s));
    }
    if (pi < Pd.size()) return fail("pixelcnn: deferred projections left over");
    return launch_sampler(p, c, r, 1, s);
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

    p->wvt.resize(NL);
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
        // one matrix per kernel row t: output n = (col j, channel co); k = (input col c)*D + ci; kernel column c - j + 1
        std::vector<float> pb((size_t)2 * D2);
        for (int j = 0; j < 2; ++j)
            for (int co = 0; co < D2; ++co) pb[(size_t)j * D2 + co] = bv[co];
        for (int t = 0; t < rows; ++t) {
            std::vector<float> pv((size_t)2 * D2 * 2 * D);
            for (int j = 0; j < 2; ++j)
                for (int co = 0; co < D2; ++co)
                    for (int cc = 0; cc < 2; ++cc)
                        for (int ci = 0; ci < D; ++ci)
                            pv[((size_t)j * D2 + co) * 2 * D + (size_t)cc * D + ci] =
                                wv[(((size_t)co * D + ci) * kh + t) * 3 + (cc - j + 1)];
            TS_TRY(upload_vec(p->wvt[l], pv));
        }
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
    TS_TRY(p->fha.upload(fb.data(), fb.size() * sizeof(float)));
    if (NL > 1) {
        // layer 1's vertical taps see XV_1 = fusion_v[:, :D] . OV0 + (fusion_v[:, D:] . AE + b) per column.  Compose (fp64,
        // rounded once): W' = W . blockdiag(F, F) acting on OV0, and Ws = W[:, :D] + W[:, D:] acting on the audio term,
        // which is the same for both columns (the audio map is repeated over the 2 columns, smplx_body_pixel.py:274).
        const float *wv1 = sd.get("layers.1.vert_stack.weight", {D2, D, 2, 3});
        if (!wv1) return 1;
        for (int t = 0; t < 2; ++t) {   // t = 0: row above (Wprev), t = 1: the row itself (Wcur)
            std::vector<double> Wt((size_t)2 * D2 * 2 * D);
            for (int j = 0; j < 2; ++j)
                for (int co = 0; co < D2; ++co)
                    for (int cc = 0; cc < 2; ++cc)
                        for (int ci = 0; ci < D; ++ci)
                            Wt[((size_t)j * D2 + co) * 2 * D + (size_t)cc * D + ci] =
                                wv1[(((size_t)co * D + ci) * 2 + t) * 3 + (cc - j + 1)];
            std::vector<float> Wc((size_t)2 * D2 * 2 * D), Ws((size_t)2 * D2 * D);
            for (int n = 0; n < 2 * D2; ++n)
                for (int cc = 0; cc < 2; ++cc)
                    for (int i = 0; i < D; ++i) {
                        double a = 0.0;
                        for (int m = 0; m < D; ++m) a += Wt[(size_t)n * 2 * D + (size_t)cc * D + m] * (double)fa[(size_t)m * D + i];
                        Wc[(size_t)n * 2 * D + (size_t)cc * D + i] = (float)a;
                    }
            for (int n = 0; n < 2 * D2; ++n)
                for (int i = 0; i < D; ++i)
                    Ws[(size_t)n * D + i] = (float)(Wt[(size_t)n * 2 * D + i] + Wt[(size_t)n * 2 * D + D + i]);
            TS_TRY((t == 1 ? p->wv1c : p->wv1p).upload(Wc.data(), Wc.size() * sizeof(float)));
            TS_TRY(pack_linear_layer(Ws.data(), D, nullptr, 2 * D2, D, t == 1 ? &p->aud_v1c : &p->aud_v1p));
        }
    }
    // head
    const float *w1 = sd.get("output_conv.0.weight", {p->HID, D, 1, 1}), *b1 = sd.get("output_conv.0.bias", {p->HID});
    const float *w2 = sd.get("output_conv.2.weight", {V, p->HID, 1, 1}), *b2 = sd.get("output_conv.2.bias", {V});
    if (!w1 || !b1 || !w2 || !b2) return 1;
    TS_TRY(p->w1.upload(w1, (size_t)p->HID * D * sizeof(float)));
    TS_TRY(p->b1.upload(b1, (size_t)p->HID * sizeof(float)));
    TS_TRY(p->w2.upload(w2, (size_t)V * p->HID * sizeof(float)));
    TS_TRY(p->b2.upload(b2, (size_t)V * sizeof(float)));
    // ---- composed horizontal maps (products in double, rounded once) ----
    {
        auto mat = [&](const std::string &k, int rows_, int cols_, int stride_, int off_) {   // rows_ x cols_ view of a conv weight
            const float *src = sd.m.at(k)->data;
            std::vector<double> m((size_t)rows_ * cols_);
            for (int i = 0; i < rows_; ++i)
                for (int jx = 0; jx < cols_; ++jx) m[(size_t)i * cols_ + jx] = src[((size_t)i * cols_ + jx) * stride_ + off_];
            return m;
        };
        auto vec = [&](const std::string &k, int n_) {
            const float *src = sd.m.at(k)->data;
            return std::vector<double>(src, src + n_);
        };
        auto mm = [](const std::vector<double> &A, const std::vector<double> &Bm, int m_, int k_, int n_) {   // (m x k)(k x n)
            std::vector<double> Cm((size_t)m_ * n_, 0.0);
            for (int i = 0; i < m_; ++i)
                for (int kk = 0; kk < k_; ++kk) {
                    const double a = A[(size_t)i * k_ + kk];
                    const double *brow = &Bm[(size_t)kk * n_];
                    double *crow = &Cm[(size_t)i * n_];
                    for (int jx = 0; jx < n_; ++jx) crow[jx] += a * brow[jx];
                }
            return Cm;
        };
        auto mv = [](const std::vector<double> &A, const std::vector<double> &x, int m_, int k_) {
            std::vector<double> y(m_, 0.0);
            for (int i = 0; i < m_; ++i)
                for (int kk = 0; kk < k_; ++kk) y[i] += A[(size_t)i * k_ + kk] * x[kk];
            return y;
        };
        auto upf = [&](std::vector<std::unique_ptr<DevBuf>> &dst, const std::vector<double> &v) {
            return upload_vec(dst, std::vector<float>(v.begin(), v.end()));
        };
        // entry 0 of the per-layer vectors is a placeholder (layer 0 has no composed stage)
        TS_TRY(upload_vec(p->rw, std::vector<float>(1, 0.f)));
        TS_TRY(upload_vec(p->rb, std::vector<float>(1, 0.f)));
        TS_TRY(upload_vec(p->wB, std::vector<float>(1, 0.f)));
        TS_TRY(upload_vec(p->bm, std::vector<float>(1, 0.f)));
        const std::vector<double> Fha = [&] {   // fusion_h[:, :D]
            const float *src = sd.m.at("fusion_h.weight")->data;
            std::vector<double> m((size_t)D * D);
            for (int i = 0; i < D; ++i)
                for (int jx = 0; jx < D; ++jx) m[(size_t)i * D + jx] = src[(size_t)i * D2 + jx];
            return m;
        }();
        for (int l = 1; l < NL; ++l) {
            const std::string q = "layers." + std::to_string(l), qp = "layers." + std::to_string(l - 1);
            std::vector<double> Rw = mat(qp + ".horiz_resid.weight", D, D, 1, 0), rbv = vec(qp + ".horiz_resid.bias", D);
            if (l == 1) {   // XH_1 = fusion_h[:, :D] . (horiz_resid_0 . G_0 + b) + AEH
                rbv = mv(Fha, rbv, D, D);
                Rw = mm(Fha, Rw, D, D, D);
            }
            const std::vector<double> Wh1 = mat(q + ".horiz_stack.weight", D2, D, 2, 1);   // tap 1: the column itself
            const std::vector<double> Wm = mm(Wh1, Rw, D2, D, D);
            std::vector<double> bmv = mv(Wh1, rbv, D2, D);
            const std::vector<double> bhv = vec(q + ".horiz_stack.bias", D2);
            for (int i = 0; i < D2; ++i) bmv[i] += bhv[i];
            const int KB = l == 1 ? D : 2 * D;
            std::vector<double> WB((size_t)D2 * KB);
            for (int i = 0; i < D2; ++i) {
                for (int jx = 0; jx < D; ++jx) WB[(size_t)i * KB + jx] = Wm[(size_t)i * D + jx];
                if (l >= 2)
                    for (int jx = 0; jx < D; ++jx) WB[(size_t)i * KB + D + jx] = Wh1[(size_t)i * D + jx];
            }
            TS_TRY(upf(p->rw, Rw));
            TS_TRY(upf(p->rb, rbv));
            TS_TRY(upf(p->wB, WB));
            TS_TRY(upf(p->bm, bmv));
            if (l == 1) {
                std::vector<float> wh1f(Wh1.begin(), Wh1.end());
                TS_TRY(pack_linear_layer(wh1f.data(), D, nullptr, D2, D, &p->aud_h1));
            }
        }
        {   // head1' = relu(W1 . (horiz_resid_{NL-1} . G + b + XH_{NL-1}) + b1)
            const std::string ql = "layers." + std::to_string(NL - 1);
            const std::vector<double> W1 = mat("output_conv.0.weight", p->HID, D, 1, 0);
            const std::vector<double> Wr = mat(ql + ".horiz_resid.weight", D, D, 1, 0);
            const std::vector<double> W1r = mm(W1, Wr, p->HID, D, D);
            std::vector<double> bb = mv(W1, vec(ql + ".horiz_resid.bias", D), p->HID, D);
            const std::vector<double> b1v = vec("output_conv.0.bias", p->HID);
            for (int i = 0; i < p->HID; ++i) bb[i] += b1v[i];
            const int K1 = NL >= 2 ? 2 * D : D;
            std::vector<float> wm((size_t)p->HID * K1), bf(bb.begin(), bb.end());
            for (int i = 0; i < p->HID; ++i) {
                for (int jx = 0; jx < D; ++jx) wm[(size_t)i * K1 + jx] = (float)W1r[(size_t)i * D + jx];
                if (NL >= 2)
                    for (int jx = 0; jx < D; ++jx) wm[(size_t)i * K1 + D + jx] = (float)W1[(size_t)i * D + jx];
            }
            TS_TRY(p->w1m.upload(wm.data(), wm.size() * sizeof(float)));
            TS_TRY(p->b1m.upload(bf.data(), bf.size() * sizeof(float)));
        }
    }
    // ---- tiled twins of every weight matrix the chain multiplies with (see ts_pixelcnn::tiled) ----
    p->tiled = skinny_descriptor_kernel_enabled() && D % 128 == 0 && p->HID % 128 == 0 && V % 16 == 0;
    if (p->tiled) {
        auto tile = [&](const DevBuf &rm, int N, int K, long ldw, int epi, int gateD) -> int {
            std::vector<float> host((size_t)N * ldw);
            TS_HIP(hipMemcpy(host.data(), rm.p, host.size() * sizeof(float), hipMemcpyDeviceToHost));
            std::vector<float> t((size_t)((N + 15) / 16) * (K / 16) * 256);
            skinny_tile_weights(host.data(), N, K, ldw, epi, gateD, t.data());
            std::unique_ptr<ts_pixelcnn::TiledW> tw(new ts_pixelcnn::TiledW());
            tw->K = K;
            tw->epi = epi;
            TS_TRY(tw->buf.upload(t.data(), t.size() * sizeof(float)));
            p->wtiled[rm.f()] = std::move(tw);
            return 0;
        };
        const int D4 = 4 * D;
        TS_TRY(tile(*p->wvt[0][2], D4, D2, D2, EPI_GATE, D));                              // V0 gate (the two code rows gathered: K = 2D)
        TS_TRY(tile(*p->wvt[0][0], D4, D2, D2, EPI_LINEAR, 0));                            // Q0 / Q1
        TS_TRY(tile(*p->wvt[0][1], D4, D2, D2, EPI_LINEAR, 0));
        for (int l = 1; l < NL; ++l) {
            if (l >= 2) {
                TS_TRY(tile(*p->wvt[l][1], D4, D2, D2, EPI_GATE, D));
                TS_TRY(tile(*p->wvt[l][0], D4, D2, D2, EPI_LINEAR, 0));
            }
            TS_TRY(tile(*p->wh[l], D2, D, D2, EPI_LINEAR, 0));                             // t0: tap-0 block of horiz_stack
            TS_TRY(tile(*p->rw[l], D, D, D, EPI_LINEAR, 0));
            TS_TRY(tile(*p->wB[l], D2, l == 1 ? D : D2, l == 1 ? D : D2, EPI_GATE, D));
        }
        for (int l = 0; l < NL; ++l) TS_TRY(tile(*p->wv2h[l], D2, D2, D2, EPI_LINEAR, 0));
        TS_TRY(tile(*p->wh[0], D2, D, D2, EPI_GATE, D));                                   // hg_0 of column 1 (K = D gather)
        if (NL > 1) {
            TS_TRY(tile(p->wv1c, D4, D2, D2, EPI_GATE, D));
            TS_TRY(tile(p->wv1p, D4, D2, D2, EPI_LINEAR, 0));
        }
        TS_TRY(tile(p->w1m, p->HID, NL >= 2 ? D2 : D, NL >= 2 ? D2 : D, EPI_LINEAR, 0));
        TS_TRY(tile(p->w2, V, p->HID, p->HID, EPI_LINEAR, 0));
    }
    if (ts::knobs().no_graph) p->use_graph = false;
    if (ts::knobs().pix_defer_p >= 0) p->defer_p = ts::knobs().pix_defer_p;
    *out = p.release();
    return 0;
}
void ts_pixelcnn_destroy(ts_pixelcnn *p) { delete p; }

long ts_pixelcnn_graph_captures(ts_pixelcnn *p, void *stream) {
    if (!p) return -1;
    ts_pixelcnn::Work *w = p->works.find((hipStream_t)stream);
    return w ? w->captures : 0;
}

int ts_debug_pixelcnn_graphs(ts_pixelcnn *p, void *stream) {
    if (!p) return -1;
    ts_pixelcnn::Work *w = p->works.find((hipStream_t)stream);
    return w ? (int)w->graphs.size() : 0;
}

int ts_pixelcnn_graph_stats(ts_pixelcnn *p, void *stream, int B, int H, int mode, int64_t *launches, double *flops) {
    if (!p) return fail("ts_pixelcnn_graph_stats: null argument");
    ts_pixelcnn::Work *w = p->works.find((hipStream_t)stream);
    if (!w) return fail("ts_pixelcnn_graph_stats: nothing was run on this stream");
    auto jt = w->graph_stats.find(std::make_tuple(B, H, 0, mode));
    if (jt == w->graph_stats.end()) return fail("ts_pixelcnn_graph_stats: no captured graph for this shape");
    if (launches) *launches = jt->second.first;
    if (flops) *flops = jt->second.second;
    return 0;
}

}  // extern "C"

namespace {

// audio conditioning of `rows` code rows per clip: AE = embedding_aud(aud); AEV / AEH = fusion_{v,h}[:, D:] . AE + bias;
// AEH1 = layer 1's horiz_stack applied to AEH (gated_pixelcnn_v2.py:137-144) — four conv_gemm launches for all rows at once
int audio_terms(ts_pixelcnn *p, ts_pixelcnn::Work *w, const float *aud, int B, int rows, hipStream_t s) {
    ts_ctx *ctx = p->ctx;
    const int D = p->D, AD = p->AD;
    ConvParams q;
    conv_layer_params(p->aud_embed, aud, AD, 1, B * rows, nullptr, 0, w->AE.f(), D, 0, D, &q);
    TS_TRY(run_conv(ctx, q, 0, s));
    conv_layer_params(p->aud_fv, w->AE.f(), D, 1, B * rows, nullptr, 0, w->AEV.f(), D, 0, D, &q);
    TS_TRY(run_conv(ctx, q, 0, s));
    conv_layer_params(p->aud_fh, w->AE.f(), D, 1, B * rows, nullptr, 0, w->AEH.f(), D, 0, D, &q);
    TS_TRY(run_conv(ctx, q, 0, s));
    if (p->NL > 1) {
        conv_layer_params(p->aud_h1, w->AEH.f(), D, 1, B * rows, nullptr, 0, w->AEH1.f(), 2 * D, 0, 2 * D, &q);
        TS_TRY(run_conv(ctx, q, 0, s));
        conv_layer_params(p->aud_v1c, w->AEV.f(), D, 1, B * rows, nullptr, 0, w->AV1C.f(), 4 * D, 0, 4 * D, &q);
        TS_TRY(run_conv(ctx, q, 0, s));
        conv_layer_params(p->aud_v1p, w->AEV.f(), D, 1, B * rows, nullptr, 0, w->AV1P.f(), 4 * D, 0, 4 * D, &q);
        TS_TRY(run_conv(ctx, q, 0, s));
    }
    return 0;
}

// class conditioning rows: CR[l][b] = class_cond_embedding_l[label[b]]  (h of gated_pixelcnn_v2.py:65)
int class_rows(ts_pixelcnn *p, ts_pixelcnn::Work *w, const int64_t *label, int B, hipStream_t s) {
    const int D = p->D;
    MiscScope ms(p->ctx, s);
    for (int l = 0; l < p->NL; ++l)
        TS_HIP(launch_gather_rows(p->cls[l]->f(), 2 * D, p->NC, label, 1, B, 2 * D, w->CR.f() + (size_t)l * B * 2 * D, 2 * D, s));
    return 0;
}

// Runs rows [r_begin, r_end) of `c`: eagerly, or as a replay of the hipGraph captured for `key` (host launch cost would
// otherwise dominate: ~37 dependent tiny launches per row).  On the graph path the kernels write codes into the Work's
// staging buffer (every pointer inside a captured kernel is a Work buffer, so a graph is valid for any caller pointers);
// the caller's arrays hold c.out_H rows per clip, of which this run covers rows c.out_row0 .. c.out_row0 + c.H - 1.
int run_rows(ts_pixelcnn *p, RunCfg c, int r_begin, int r_end, bool graph, const ts_pixelcnn::Work::Key &key,
             const float *uniforms, int64_t *codes, hipStream_t s, bool capture_only = false) {
    ts_ctx *ctx = p->ctx;
    ts_pixelcnn::Work *w = c.w;
    const int out_H = c.out_H > 0 ? c.out_H : c.H;
    auto row_loop = [&](hipStream_t st) -> int {
        for (int r = r_begin; r < r_end; ++r) {
            const bool need_h = r >= c.out_r0 && !(c.mode == TS_TEACHER_FORCED && !c.logits);   // prefix rows only feed the row cache
            TS_TRY(run_row(p, c, r, need_h, st));
        }
        return 0;
    };
    if (!graph) {
        if (out_H != c.H) return fail("pixelcnn: eager rows write the caller's arrays whole");
        c.codes = codes;
        c.uniforms = uniforms;
        return row_loop(s);
    }
    c.codes = static_cast<int64_t *>(w->codes_int.p);
    c.uniforms = c.mode == TS_SAMPLE_UNIFORMS ? w->unif_int.f() : nullptr;
    c.dyn = static_cast<const uint64_t *>(w->dyn.p);
    if (c.mode == TS_SAMPLE_UNIFORMS && !capture_only)
        TS_HIP(hipMemcpy2DAsync(w->unif_int.p, (size_t)c.H * 2 * sizeof(float), uniforms + (size_t)c.out_row0 * 2,
                                (size_t)out_H * 2 * sizeof(float), (size_t)c.H * 2 * sizeof(float), c.B, hipMemcpyDeviceToDevice, s));
    // the call's sampler words travel as the ARGUMENTS of a one-thread launch (copied when the launch is queued): no host buffer
    // has to stay intact behind the call, so any number of calls may be queued on the stream without a synchronisation
    if (!capture_only) TS_HIP(launch_set_words3(static_cast<uint64_t *>(w->dyn.p), c.seed, (uint64_t)c.clip0, (uint64_t)c.pos_base, s));
    auto it = w->graphs.find(key);
    if (it == w->graphs.end()) {
        TS_TRY(w->make_room(s, key));
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
        ++w->captures;
        it = w->graphs.emplace(key, ts_pixelcnn::Work::Entry{ex, 0}).first;
    }
    if (capture_only) return 0;
    it->second.used = ++w->tick;
    TS_HIP(hipGraphLaunch(it->second.exec, s));
    TS_HIP(hipMemcpy2DAsync(codes + (size_t)c.out_row0 * 2, (size_t)out_H * 2 * sizeof(int64_t), w->codes_int.p,
                            (size_t)c.H * 2 * sizeof(int64_t), (size_t)c.H * 2 * sizeof(int64_t), c.B, hipMemcpyDeviceToDevice, s));
    return 0;
}

// A one-shot call (no prefix) as a sequence of CHUNK_ROWS-row graphs: the reference's own evaluation loop
// (scripts/test_body.py:113-194) feeds clips of arbitrary lengths, and a whole-call graph per length would cost a capture +
// instantiate of ~36 H nodes for every new H and keep them all.  A chunk's graph depends on (B, rows in the chunk, buffer
// phase of its first row, mode) only — the streaming sessions' construction (ts_pixelcnn_stream_step): a 4-row token ring, rows
// indexed absolutely, the Philox position base a device word — so two graphs (first chunk, later chunks) plus one or two for the
// short last chunk serve every clip length.  The audio terms were computed for the whole call; a chunk's rows are copied into the
// compact chunk buffers its graph reads.  Bit-identical to the whole-call graph (same launches, same order, same rings: only the
// look-ahead partial sums of rows past the end are computed and never read).
int run_chunked(ts_pixelcnn *p, ts_pixelcnn::Work *w, int B, int H, int mode, const float *uniforms, uint64_t seed, int64_t clip0,
                int64_t *codes, hipStream_t s) {
    const size_t D = p->D, f = sizeof(float);
    constexpr int RING = 4;
    for (int r0 = 0; r0 < H; r0 += CHUNK_ROWS) {
        const int Hc = std::min(CHUNK_ROWS, H - r0);
        RunCfg c{B, Hc, 0, r0 + Hc, mode, nullptr, seed, clip0, nullptr, nullptr, nullptr, w,
                 RING, r0, Hc, r0, r0, 2l * r0, 0x7fffffff};
        c.aeh = w->cAEH.f(), c.aeh1 = w->cAEH1.f(), c.av1c = w->cAV1C.f(), c.av1p = w->cAV1P.f();
        c.out_H = H;
        c.out_row0 = r0;
        struct { const DevBuf *src; DevBuf *dst; size_t width; } rows[4] = {
            {&w->AEH, &w->cAEH, D}, {&w->AEH1, &w->cAEH1, 2 * D}, {&w->AV1C, &w->cAV1C, 4 * D}, {&w->AV1P, &w->cAV1P, 4 * D}};
        for (auto &m : rows)
            if (p->NL > 1 || m.src == &w->AEH)
                TS_HIP(hipMemcpy2DAsync(m.dst->p, (size_t)Hc * m.width * f, m.src->f() + (size_t)r0 * m.width, (size_t)H * m.width * f,
                                        (size_t)Hc * m.width * f, B, hipMemcpyDeviceToDevice, s));
        const int phase = r0 < 3 ? r0 : 3 + (r0 % 4);
        TS_TRY(run_rows(p, c, r0, r0 + Hc, true, std::make_tuple(B, Hc, -(1 + phase), mode), uniforms, codes, s));
    }
    return 0;
}

}  // namespace

extern "C" {

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
    const int Htot = H0 + H, AD = p->AD;
    ts_pixelcnn::Work *w = &p->work(s);
    TS_TRY(ensure_work(p, w, B, Htot));

    // eager launches remain for the instrumented / logits-returning / teacher-forced paths
    const bool graph = p->use_graph && !ctx->prof.on && !logits && mode != TS_TEACHER_FORCED;
    RunCfg c = one_shot_cfg(B, H, H0, mode, uniforms, seed, clip0, codes, logits, w);

    const float *aud_all = aud;
    if (H0 > 0) {
        const size_t f = sizeof(float);
        TS_HIP(hipMemcpy2DAsync(w->aud_all.f(), (size_t)Htot * AD * f, pre_aud, (size_t)H0 * AD * f, (size_t)H0 * AD * f,
                                B, hipMemcpyDeviceToDevice, s));
        TS_HIP(hipMemcpy2DAsync(w->aud_all.f() + (size_t)H0 * AD, (size_t)Htot * AD * f, aud, (size_t)H * AD * f,
                                (size_t)H * AD * f, B, hipMemcpyDeviceToDevice, s));
        aud_all = w->aud_all.f();
    }
    TS_TRY(audio_terms(p, w, aud_all, B, Htot, s));
    TS_TRY(class_rows(p, w, label, B, s));
    // known codes: the continuity prefix, and every position when teacher forced
    if (H0 > 0 || mode == TS_TEACHER_FORCED) {
        MiscScope ms(ctx, s);
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
    const ts_pixelcnn::Work::Key key = std::make_tuple(B, H, H0, mode);
    // A shape without a whole-call graph runs as chunk graphs (two or three small captures that serve every clip length) until it is
    // hot (Work::hot: pinned by ts_pixelcnn_prepare, or its third sighting among the last 16 one-shot calls of this stream); then it gets
    // its own whole-call graph (one replay per call: the serving loops, bench.py).  The cache is bounded either way.
    if (graph && H0 == 0 && H > CHUNK_ROWS && !w->graphs.count(key) && !w->hot(key))
        return run_chunked(p, w, B, H, mode, uniforms, seed, clip0, codes, s);
    return run_rows(p, c, 0, Htot, graph, key, uniforms, codes, s);
}

// Captures (without running anything) the whole-call graph of a one-shot shape on `stream` and pins it: the first real call of that
// shape is already a single replay, and the graph is never evicted.  Serving hosts call this for their pass shapes at start-up
// (bench.py's warm()); nothing in the reference corresponds (it has no graphs).
int ts_pixelcnn_prepare(ts_pixelcnn *p, int B, int H, int mode, void *stream) {
    if (!p) return fail("ts_pixelcnn_prepare: null argument");
    if (B < 1 || H < 1) return fail("ts_pixelcnn_prepare: bad shape");
    if (mode != TS_SAMPLE_GREEDY && mode != TS_SAMPLE_UNIFORMS && mode != TS_SAMPLE_PHILOX) return fail("ts_pixelcnn_prepare: bad mode");
    TS_HIP(hipSetDevice(p->ctx->device));
    if (!p->use_graph) return 0;                                   // TS_NO_GRAPH: eager launches, nothing to prepare
    hipStream_t s = (hipStream_t)stream;
    ts_pixelcnn::Work *w = &p->work(s);
    TS_TRY(ensure_work(p, w, B, H));
    const ts_pixelcnn::Work::Key key = std::make_tuple(B, H, 0, mode);
    if (!w->pinned.count(key) && w->pinned.size() >= ts_pixelcnn::Work::PIN_CAP) return fail("ts_pixelcnn_prepare: too many pinned shapes on this stream");
    RunCfg c = one_shot_cfg(B, H, 0, mode, nullptr, 0, 0, nullptr, nullptr, w);
    TS_TRY(run_rows(p, c, 0, H, true, key, nullptr, nullptr, s, /*capture_only=*/true));
    w->pinned.insert(key);
    return 0;
}

// ---- streaming generation (reference: the pre_latents / pre_audio prefix of gated_pixelcnn_v2.py:158-165
// and its caller smplx_body_pixel.py:260-269,291-304, which recompute the whole prefix for every chunk) --------------------
int ts_pixelcnn_stream_open(ts_pixelcnn *p, const int64_t *label, int B, int max_chunk_rows, ts_pixelcnn_stream **out) {
    if (!p || !label || !out) return fail("ts_pixelcnn_stream_open: null argument");
    if (B < 1 || max_chunk_rows < 1) return fail("ts_pixelcnn_stream_open: bad shape");
    TS_HIP(hipSetDevice(p->ctx->device));
    std::unique_ptr<ts_pixelcnn_stream> st(new ts_pixelcnn_stream());
    st->p = p;
    st->B = B;
    st->max_rows = max_chunk_rows;
    // everything is allocated here, once: growing a buffer later would drop the row cache it holds
    TS_TRY(ensure_work(p, &st->w, B, std::max(max_chunk_rows, 4)));
    TS_TRY(st->label.ensure((size_t)B * sizeof(int64_t)));
    TS_HIP(hipMemcpy(st->label.p, label, (size_t)B * sizeof(int64_t), hipMemcpyDeviceToDevice));
    *out = st.release();
    return 0;
}

void ts_pixelcnn_stream_close(ts_pixelcnn_stream *st) { delete st; }

int64_t ts_pixelcnn_stream_rows(const ts_pixelcnn_stream *st) { return st ? st->rows : -1; }

int ts_pixelcnn_stream_step(ts_pixelcnn_stream *st, const float *aud, int Hc, int mode, const float *uniforms, uint64_t seed,
                            int64_t clip0, int64_t *codes, void *stream) {
    if (!st || !aud || !codes) return fail("ts_pixelcnn_stream_step: null argument");
    if (Hc < 1) return fail("ts_pixelcnn_stream_step: empty chunk");
    if (Hc > st->max_rows) return fail("ts_pixelcnn_stream_step: chunk longer than max_chunk_rows given to ts_pixelcnn_stream_open");
    if (mode != TS_SAMPLE_GREEDY && mode != TS_SAMPLE_UNIFORMS && mode != TS_SAMPLE_PHILOX)
        return fail("ts_pixelcnn_stream_step: bad mode");
    if (mode == TS_SAMPLE_UNIFORMS && !uniforms) return fail("ts_pixelcnn_stream_step: uniforms required");
    if (st->rows + Hc > (1l << 30)) return fail("ts_pixelcnn_stream_step: row counter overflow");
    ts_pixelcnn *p = st->p;
    ts_ctx *ctx = p->ctx;
    hipStream_t s = (hipStream_t)stream;
    ts_pixelcnn::Work *w = &st->w;
    const int B = st->B, r0 = (int)st->rows;
    constexpr int RING = 4;   // token rows kept: layer 0 looks three code rows up; 4 also is the period of the Q ring
    if (r0 == 0) TS_TRY(class_rows(p, w, static_cast<const int64_t *>(st->label.p), B, s));
    TS_TRY(audio_terms(p, w, aud, B, Hc, s));
    RunCfg c{B, Hc, 0, r0 + Hc, mode, nullptr, seed, clip0, nullptr, nullptr, nullptr, w,
             RING, r0, Hc, r0, r0, 2l * r0, 0x7fffffff};
    c.audio_from(w);
    // a captured chunk is valid for every start row with the same buffer phases (parity of the per-layer row cache, slot
    // in the 4-row rings) and the same set of existing rows above (rows 0..2 have fewer): key on that, not on r0
    const int phase = r0 < 3 ? r0 : 3 + (r0 % 4);
    const bool graph = p->use_graph && !ctx->prof.on;
    TS_TRY(run_rows(p, c, r0, r0 + Hc, graph, std::make_tuple(B, Hc, 1000 + phase, mode), uniforms, codes, s));
    st->rows += Hc;
    return 0;
}

}  // extern "C"
