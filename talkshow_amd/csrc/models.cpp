// Host side of the conv stacks: weight folding / packing, the layer executor, AudioEncoder and VQVAE objects.
//
// Reference structure restated here (as a launch plan, not as code):
//   vqvae_modules.ConvNormRelu      nets/spg/vqvae_modules.py:87-172   conv -> BN(eval) [+ residual conv] -> LeakyReLU(0.2)
//   vqvae_modules.Res_CNR_Stack     nets/spg/vqvae_modules.py:175-212  n x ConvNormRelu, conv, BN, relu(h + x)
//   vqvae_1d.AudioEncoder / Encoder nets/spg/vqvae_1d.py:11-34,66-92
//   vqvae_1d.Decoder                nets/spg/vqvae_1d.py:116-149
//   VectorQuantizerEMA (eval)       nets/spg/vqvae_modules.py:274-286,311-323
#include <cstdio>
#include "host_common.h"

namespace ts {

static thread_local std::string g_err;
void set_error(const std::string &m) { g_err = m; }
const char *last_error() { return g_err.c_str(); }

// registry of every object that owns per-stream scratch (see StreamWorks)
namespace {
std::mutex &scoped_mu() {
    static std::mutex m;
    return m;
}
std::vector<StreamScoped *> &scoped_all() {
    static std::vector<StreamScoped *> v;
    return v;
}
}  // namespace
StreamScoped::StreamScoped() {
    std::lock_guard<std::mutex> g(scoped_mu());
    scoped_all().push_back(this);
}
StreamScoped::~StreamScoped() {
    std::lock_guard<std::mutex> g(scoped_mu());
    auto &v = scoped_all();
    for (size_t i = 0; i < v.size(); ++i)
        if (v[i] == this) {
            v[i] = v.back();
            v.pop_back();
            break;
        }
}
void drop_stream_everywhere(hipStream_t s) {
    std::lock_guard<std::mutex> g(scoped_mu());
    for (StreamScoped *o : scoped_all()) o->drop_stream(s);
}

// ---------------------------------------------------------------------------------------------- profiler
hipEvent_t Profiler::get_event() {
    if (!pool.empty()) {
        hipEvent_t e = pool.back();
        pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
void Profiler::begin(int fam, hipStream_t s) {
    Rec r{get_event(), get_event(), fam};
    (void)hipEventRecord(r.a, s);
    recs.push_back(r);
}
void Profiler::end(hipStream_t s) { (void)hipEventRecord(recs.back().b, s); }
int Profiler::collect() {
    for (auto &r : recs) {
        TS_HIP(hipEventSynchronize(r.b));
        float t = 0.f;
        TS_HIP(hipEventElapsedTime(&t, r.a, r.b));
        ms[r.fam] += t;
        launches[r.fam] += 1;
        if (!r.tag.empty())
            fprintf(stderr, "[ts_prof] %-56s %9.1f us %7.1f TF\n", r.tag.c_str(), t * 1e3, r.flops / (t * 1e-3) / 1e12);
        pool.push_back(r.a);
        pool.push_back(r.b);
    }
    recs.clear();
    return 0;
}
void Profiler::reset() {
    for (int i = 0; i < FAM_COUNT; ++i) ms[i] = 0, flops[i] = 0, launches[i] = 0;
}
Profiler::~Profiler() {
    for (auto &r : recs) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    for (auto e : pool) (void)hipEventDestroy(e);
}

// per-stream scratch of the ring engine's stream-K band (conv_gemm_ring.hip): partial accumulators + flags.  The flags are zeroed when
// the buffer is (re)allocated; after that every flag is cleared by the one workgroup that consumes it.
namespace {
struct SkWork {
    DevBuf ws, flags;
};
StreamWorks<SkWork> &sk_works() {
    static StreamWorks<SkWork> w;
    return w;
}
}  // namespace
int conv_sk_workspace(hipStream_t s, size_t ws_floats, size_t nflags, float **ws, int **flags) {
    SkWork &w = sk_works().get(s);
    TS_TRY(w.ws.ensure(ws_floats * sizeof(float)));
    if (nflags * sizeof(int) > w.flags.bytes) {
        TS_TRY(w.flags.ensure(nflags * sizeof(int)));
        TS_HIP(hipMemsetAsync(w.flags.p, 0, w.flags.bytes, s));
    }
    *ws = w.ws.f();
    *flags = w.flags.i();
    return 0;
}

int run_conv(ts_ctx *ctx, const ConvParams &p, int tile, hipStream_t s) {
    ctx->n_launch[FAM_CONV].fetch_add(1, std::memory_order_relaxed);
    atomic_add(ctx->n_flops[FAM_CONV], conv_gemm_flops(p));
    if (ctx->prof.on) {
        ctx->prof.begin(FAM_CONV, s);
        ctx->prof.flops[FAM_CONV] += conv_gemm_flops(p);
        if (ts::knobs().prof_log) {
            char buf[128];
            snprintf(buf, sizeof(buf), "conv M=%d N=%d K=%d groups=%d z=%d stride=%d Lout=%d", p.M, p.N, p.Ktot, p.ngroups, p.zdiv,
                     p.stride, p.Lout);
            ctx->prof.recs.back().tag = buf;
            ctx->prof.recs.back().flops = conv_gemm_flops(p);
        }
    }
    // tile ids 22 / 23: the opt-in split-bf16 plan with 2 / 3 planes (conv_gemm_split.hip); everything else: fp32 MFMA
    hipError_t e = (tile == 22 || tile == 23) ? launch_conv_gemm_split(p, tile - 20, s) : launch_conv_gemm(p, tile, s);
    if (ctx->prof.on) ctx->prof.end(s);
    if (e != hipSuccess) return fail(std::string("conv_gemm launch: ") + hipGetErrorString(e));
    return 0;
}

int run_skinny(ts_ctx *ctx, const SkinnyParams &p, hipStream_t s) {
    ctx->n_launch[FAM_SKINNY].fetch_add(1, std::memory_order_relaxed);
    atomic_add(ctx->n_flops[FAM_SKINNY], 2.0 * p.M * (double)p.N * p.Ktot);
    if (ctx->prof.on) {
        ctx->prof.begin(FAM_SKINNY, s);
        ctx->prof.flops[FAM_SKINNY] += 2.0 * p.M * (double)p.N * p.Ktot;
    }
    hipError_t e = launch_skinny_gemm(p, s);
    if (ctx->prof.on) ctx->prof.end(s);
    if (e != hipSuccess) return fail(std::string("skinny_gemm launch: ") + hipGetErrorString(e));
    return 0;
}

int run_skinny_batch(ts_ctx *ctx, const SkinnyParams *const *ps, int n, hipStream_t s) {
    double fl = 0;
    for (int i = 0; i < n; ++i) fl += 2.0 * ps[i]->M * (double)ps[i]->N * ps[i]->Ktot;
    ctx->n_launch[FAM_SKINNY].fetch_add(1, std::memory_order_relaxed);
    atomic_add(ctx->n_flops[FAM_SKINNY], fl);
    if (ctx->prof.on) {
        ctx->prof.begin(FAM_SKINNY, s);
        ctx->prof.flops[FAM_SKINNY] += fl;
    }
    hipError_t e = launch_skinny_batch(ps, n, s);
    if (ctx->prof.on) ctx->prof.end(s);
    if (e != hipSuccess) return fail(std::string("skinny_gemm launch: ") + hipGetErrorString(e));
    return 0;
}

// ---------------------------------------------------------------------------------------------- packing
// BatchNorm1d eval folded into the preceding conv: y = conv(x)*a + c with a = gamma / sqrt(var + 1e-5),
// c = (bias - mean) * a + beta.  The parallel residual conv of the down / up ConvNormRelu
// (vqvae_modules.py:122-135,167-171: out = norm(conv(x)); out += residual_layer(x)) has the same geometry, so
// its weights and bias are added into the folded ones: one GEMM per reference layer.
int pack_conv_layer(const StateDict &sd, const std::string &conv_key, const std::string &norm_key,
                    const std::string &res_key, int kind, int k, int cin, int cout, int act, ConvLayer *L) {
    const int K = kind == 0 ? k : 4;
    const bool tr = kind == 2;
    const float *w = tr ? sd.get(conv_key + ".weight", {cin, cout, K}) : sd.get(conv_key + ".weight", {cout, cin, K});
    const float *b = sd.get(conv_key + ".bias", {cout});
    if (!w || !b) return 1;
    std::vector<float> a(cout, 1.f), c(cout, 0.f);
    if (!norm_key.empty()) {
        const float *g = sd.get(norm_key + ".weight", {cout}), *be = sd.get(norm_key + ".bias", {cout});
        const float *mu = sd.get(norm_key + ".running_mean", {cout}), *var = sd.get(norm_key + ".running_var", {cout});
        if (!g || !be || !mu || !var) return 1;
        for (int o = 0; o < cout; ++o) {
            a[o] = g[o] / std::sqrt(var[o] + 1e-5f);
            c[o] = (b[o] - mu[o]) * a[o] + be[o];
        }
    } else {
        for (int o = 0; o < cout; ++o) c[o] = b[o];
    }
    const float *wr = nullptr, *br = nullptr;
    if (!res_key.empty()) {
        wr = tr ? sd.get(res_key + ".weight", {cin, cout, K}) : sd.get(res_key + ".weight", {cout, cin, K});
        br = sd.get(res_key + ".bias", {cout});
        if (!wr || !br) return 1;
    }
    auto W = [&](int o, int i, int kk) -> float {   // folded weight, reference indexing
        const long idx = tr ? ((long)i * cout + o) * K + kk : ((long)o * cin + i) * K + kk;
        float v = w[idx] * a[o];
        if (wr) v += wr[idx];
        return v;
    };

    L->kind = kind;
    L->cin = cin;
    L->cin_pad = round_up(cin, 32);
    L->cout = cout;
    L->cout_pad = round_up(cout, 32);
    L->npad = round_up(cout, 128);
    L->act = act;
    L->name = conv_key;
    const int cp = L->cin_pad;
    // tap list per group: (input row shift d, kernel index kk)
    std::vector<std::pair<int, int>> taps[2];
    if (kind == 0) {
        L->ngroups = 1;
        if (K == 1) taps[0] = {{0, 0}};
        else if (K == 3) taps[0] = {{-1, 0}, {0, 1}, {1, 2}};
        else return fail("pack_conv_layer: unsupported kernel size");
    } else if (kind == 1) {   // Conv1d k4 s2 p1: out[t] = sum_k W_k x[2t + k - 1]
        L->ngroups = 1;
        taps[0] = {{-1, 0}, {0, 1}, {1, 2}, {2, 3}};
    } else {                  // ConvTranspose1d k4 s2 p1: out[2j] = W_1 x[j] + W_3 x[j-1]; out[2j+1] = W_2 x[j] + W_0 x[j+1]
        L->ngroups = 2;
        taps[0] = {{-1, 3}, {0, 1}};
        taps[1] = {{0, 2}, {1, 0}};
    }
    L->nseg = (int)taps[0].size();
    L->ktot = L->nseg * cp;
    std::vector<float> wp((size_t)L->ngroups * L->npad * L->ktot, 0.f), bp((size_t)L->ngroups * L->npad, 0.f);
    for (int g = 0; g < L->ngroups; ++g) {
        for (int s = 0; s < L->nseg; ++s) {
            L->segs[g][s] = ConvSeg{taps[g][s].first, 0, cp};
            for (int o = 0; o < cout; ++o)
                for (int i = 0; i < cin; ++i)
                    wp[((size_t)g * L->npad + o) * L->ktot + (size_t)s * cp + i] = W(o, i, taps[g][s].second);
        }
        for (int o = 0; o < cout; ++o) bp[(size_t)g * L->npad + o] = c[o] + (br ? br[o] : 0.f);
    }
    TS_TRY(L->w.upload(wp.data(), wp.size() * sizeof(float)));
    TS_TRY(L->bias.upload(bp.data(), bp.size() * sizeof(float)));
    return 0;
}

int pack_conv_raw(const float *w, const float *bias, int cout, int cin, int K, const int *taps, int act, ConvLayer *L) {
    if (K > 4) return fail("pack_conv_raw: at most 4 taps");
    L->kind = 0;
    L->cin = cin;
    L->cin_pad = round_up(cin, 32);
    L->cout = cout;
    L->cout_pad = round_up(cout, 32);
    L->npad = round_up(cout, 128);
    L->act = act;
    L->ngroups = 1;
    L->nseg = K;
    const int cp = L->cin_pad;
    L->ktot = K * cp;
    std::vector<float> wp((size_t)L->npad * L->ktot, 0.f), bp(L->npad, 0.f);
    for (int s = 0; s < K; ++s) {
        L->segs[0][s] = ConvSeg{taps[s], 0, cp, 1};
        for (int o = 0; o < cout; ++o)
            for (int i = 0; i < cin; ++i) wp[(size_t)o * L->ktot + (size_t)s * cp + i] = w[((size_t)o * cin + i) * K + s];
    }
    if (bias)
        for (int o = 0; o < cout; ++o) bp[o] = bias[o];
    TS_TRY(L->w.upload(wp.data(), wp.size() * sizeof(float)));
    TS_TRY(L->bias.upload(bp.data(), bp.size() * sizeof(float)));
    return 0;
}

int pack_linear_layer(const float *w, long ldw, const float *bias, int N, int K, ConvLayer *L) {
    L->kind = 0;
    L->cin = K;
    L->cin_pad = round_up(K, 32);
    L->cout = N;
    L->cout_pad = round_up(N, 32);
    L->npad = round_up(N, 128);
    L->act = 0;
    L->ngroups = 1;
    L->nseg = 1;
    L->ktot = L->cin_pad;
    L->segs[0][0] = ConvSeg{0, 0, L->cin_pad};
    std::vector<float> wp((size_t)L->npad * L->ktot, 0.f), bp(L->npad, 0.f);
    for (int o = 0; o < N; ++o) {
        for (int i = 0; i < K; ++i) wp[(size_t)o * L->ktot + i] = w[(long)o * ldw + i];
        if (bias) bp[o] = bias[o];
    }
    TS_TRY(L->w.upload(wp.data(), wp.size() * sizeof(float)));
    TS_TRY(L->bias.upload(bp.data(), bp.size() * sizeof(float)));
    return 0;
}

int conv_layer_params(const ConvLayer &L, const float *x, int ldx, int B, int Lin, const float *res, int ldr,
                      float *out, int ldo, int out_col0, int n_store, ConvParams *p) {
    std::memset(p, 0, sizeof(*p));
    const int Lrows = L.kind == 1 ? Lin / 2 : Lin;   // GEMM rows per clip
    p->M = B * Lrows;
    p->Lout = Lrows;
    p->Lin = Lin;
    p->stride = L.kind == 1 ? 2 : 1;
    p->ldx = ldx;
    p->ldr = ldr;
    p->N = n_store;
    p->Ktot = L.ktot;
    p->act = L.act;
    p->ngroups = L.ngroups;
    for (int g = 0; g < L.ngroups; ++g) {
        ConvGroup &G = p->g[g];
        G.x = x;
        G.w = L.w.f() + (size_t)g * L.npad * L.ktot;
        G.bias = L.bias.f() + (size_t)g * L.npad;
        G.res = res;
        G.out = out;
        G.nseg = L.nseg;
        for (int s = 0; s < L.nseg; ++s) G.seg[s] = L.segs[g][s];
        if (L.kind == 2) {   // two output phases interleave: row (b, 2j + g) of a (B, 2Lin, ldo) buffer
            G.out_col0 = g * ldo + out_col0;
        } else {
            G.out_col0 = out_col0;
        }
    }
    p->ldo = L.kind == 2 ? 2 * ldo : ldo;
    return L.kind == 2 ? 2 * Lin : Lrows;
}

}  // namespace ts

using namespace ts;

// ---------------------------------------------------------------------------------------------- conv trunks
namespace {

struct Stack {   // Res_CNR_Stack
    std::vector<std::unique_ptr<ConvLayer>> layers;   // ConvNormRelu x n
    ConvLayer tail;                                   // conv + BN, then relu(h + x)
};

int pack_stack(const StateDict &sd, const std::string &p, int c, int nres, Stack *st) {
    for (int i = 0; i < nres; ++i) {
        st->layers.emplace_back(new ConvLayer());
        const std::string q = p + "._layers." + std::to_string(i);
        TS_TRY(pack_conv_layer(sd, q + ".conv", q + ".norm", "", 0, 3, c, c, 1, st->layers.back().get()));
    }
    TS_TRY(pack_conv_layer(sd, p + ".conv", p + ".norm", "", 0, 3, c, c, 2, &st->tail));
    return 0;
}

// rotating activation buffers: every stage of a trunk holds B*T*hid/4 floats (L halves when C doubles)
struct Pool {
    DevBuf slab;
    size_t each = 0;
    int live[2] = {-1, -1};
    int ensure(size_t floats_each) {
        each = (floats_each + 63) / 64 * 64;
        return slab.ensure(4 * each * sizeof(float));
    }
    float *buf(int i) const { return slab.f() + (size_t)i * each; }
    int pick(int a, int b = -1, int c = -1) const {
        for (int i = 0; i < 4; ++i)
            if (i != a && i != b && i != c) return i;
        return 0;
    }
};

}  // namespace

struct ts_convnet {   // encoder trunk: project, stack, down, stack, down, stack  [+ pre_vq_conv]
    ts_ctx *ctx = nullptr;
    int in_dim = 0, hid = 0, nres = 0;
    ConvLayer project, down1, down2;
    Stack s1, s2, s3;
    bool has_pre_vq = false;
    ConvLayer pre_vq;
    // scratch is per stream (one weight copy per GPU, one activation arena per stream): independent batches may be in
    // flight on different streams of the same device
    struct Work {
        Pool pool;
        DevBuf xin;   // padded copy of the input when in_dim % 32 != 0
    };
    StreamWorks<Work> works;
    Work &work(hipStream_t s) { return works.get(s); }
};

namespace {

int pack_trunk(ts_ctx *ctx, const StateDict &sd, const std::string &p, int in_dim, int hid, int nres, ts_convnet *n) {
    if (hid % 128 != 0) return fail("num_hiddens must be a multiple of 128 (channel tiles are 32 wide)");
    n->ctx = ctx;
    n->in_dim = in_dim;
    n->hid = hid;
    n->nres = nres;
    TS_TRY(pack_conv_layer(sd, p + "project.conv", p + "project.norm", "", 0, 3, in_dim, hid / 4, 1, &n->project));
    TS_TRY(pack_stack(sd, p + "_enc_1", hid / 4, nres, &n->s1));
    TS_TRY(pack_conv_layer(sd, p + "_down_1.conv", p + "_down_1.norm", p + "_down_1.residual_layer", 1, 4, hid / 4,
                           hid / 2, 1, &n->down1));
    TS_TRY(pack_stack(sd, p + "_enc_2", hid / 2, nres, &n->s2));
    TS_TRY(pack_conv_layer(sd, p + "_down_2.conv", p + "_down_2.norm", p + "_down_2.residual_layer", 1, 4, hid / 2, hid,
                           1, &n->down2));
    TS_TRY(pack_stack(sd, p + "_enc_3", hid, nres, &n->s3));
    return 0;
}

// Runs the same layer of n (1 or 2) structurally identical networks.  Body and hand VQ-VAEs differ only in their weights
// (and in the width of their first / last layer), so wherever the two layers have the same geometry they go out as ONE
// grouped conv_gemm launch (blockIdx.z selects the network): twice the tiles per launch, which is what the 256 CUs need
// at these sizes (608 tiles of 64x64 per network leave a 21 % tail; 1216 leave 5 %).
int run_layer_n(ts_ctx *ctx, int n, const ConvLayer *const *L, const float *const *x, int ldx, int B, int Lin,
                const float *const *res, int ldr, float *const *out, int ldo, const int *col0, const int *nstore,
                hipStream_t s, int *Lout) {
    ConvParams p[2];
    for (int i = 0; i < n; ++i)
        *Lout = conv_layer_params(*L[i], x[i], ldx, B, Lin, res ? res[i] : nullptr, ldr, out[i], ldo, col0 ? col0[i] : 0,
                                  nstore[i], &p[i]);
    bool same = n == 2 && L[0]->kind == L[1]->kind && L[0]->cin_pad == L[1]->cin_pad && L[0]->ktot == L[1]->ktot &&
                L[0]->act == L[1]->act && L[0]->ngroups == L[1]->ngroups && nstore[0] == nstore[1] &&
                2 * L[0]->ngroups <= 4;
    if (same) {
        ConvParams &q = p[0];
        for (int g = 0; g < L[1]->ngroups; ++g) q.g[q.ngroups + g] = p[1].g[g];
        q.ngroups += L[1]->ngroups;
        return run_conv(ctx, q, 0, s);
    }
    for (int i = 0; i < n; ++i) TS_TRY(run_conv(ctx, p[i], 0, s));
    return 0;
}

int run_layer(ts_ctx *ctx, const ConvLayer &L, const float *x, int ldx, int B, int Lin, const float *res, int ldr,
              float *out, int ldo, int col0, int nstore, hipStream_t s, int *Lout) {
    const ConvLayer *Lp[1] = {&L};
    const float *xp[1] = {x}, *rp[1] = {res};
    float *op[1] = {out};
    return run_layer_n(ctx, 1, Lp, xp, ldx, B, Lin, rp, ldr, op, ldo, &col0, &nstore, s, Lout);
}

// Res_CNR_Stack on pool buffer `cur` of each network; returns the index of the output buffer (same for all)
int run_stack_n(ts_ctx *ctx, int n, const Stack *const *st, Pool *const *pool, int cur, int c, int B, int L, hipStream_t s,
                int *out_idx) {
    int h = cur, tmp = 0;
    const int ns[2] = {c, c};
    const ConvLayer *Lp[2];
    const float *xp[2], *rp[2];
    float *op[2];
    for (size_t k = 0; k < st[0]->layers.size(); ++k) {
        const int o = pool[0]->pick(cur, h);
        for (int i = 0; i < n; ++i) { Lp[i] = st[i]->layers[k].get(); xp[i] = pool[i]->buf(h); op[i] = pool[i]->buf(o); }
        TS_TRY(run_layer_n(ctx, n, Lp, xp, c, B, L, nullptr, 0, op, c, nullptr, ns, s, &tmp));
        h = o;
    }
    const int o = pool[0]->pick(cur, h);
    for (int i = 0; i < n; ++i) { Lp[i] = &st[i]->tail; xp[i] = pool[i]->buf(h); rp[i] = pool[i]->buf(cur); op[i] = pool[i]->buf(o); }
    TS_TRY(run_layer_n(ctx, n, Lp, xp, c, B, L, rp, c, op, c, nullptr, ns, s, &tmp));
    *out_idx = o;
    return 0;
}

// encoder trunks of n networks: x[i] (B,T,in_dim_i) with row stride x_ld (0 = in_dim) -> pool buffer holding
// (B,T/4,hid); returns buffer index (same for all) and H
int run_trunk_n(int n, ts_convnet *const *net, const float *const *x, int x_ld, int B, int T, hipStream_t s, int *out_idx,
                int *H) {
    if (T < 4) return fail("sequence too short: need T >= 4 frames");
    ts_ctx *ctx = net[0]->ctx;
    const int hid = net[0]->hid;
    Pool *pool[2];
    const float *xin[2];
    int ldx[2];
    for (int i = 0; i < n; ++i) {
        if (net[i]->hid != hid) return fail("paired networks must have the same width");
        ts_convnet::Work &wk = net[i]->work(s);
        pool[i] = &wk.pool;
        TS_TRY(pool[i]->ensure((size_t)B * T * (hid / 4)));
        xin[i] = x[i];
        ldx[i] = net[i]->in_dim;
        if (net[i]->in_dim % 32 != 0) {
            const int cp = net[i]->project.cin_pad;
            TS_TRY(wk.xin.ensure((size_t)B * T * cp * sizeof(float)));
            MiscScope ms(ctx, s);
            TS_HIP(launch_pad_rows(x[i], x_ld > 0 ? x_ld : net[i]->in_dim, net[i]->in_dim, wk.xin.f(), cp, cp, (long)B * T, s));
            xin[i] = wk.xin.f();
            ldx[i] = cp;
        } else if (x_ld > 0) {
            ldx[i] = x_ld;
        }
    }
    int L = T, tmp = 0, cur = 0, o = 0;
    const ConvLayer *Lp[2];
    const Stack *Sp[2];
    const float *xp[2];
    float *op[2];
    int ns[2];
    // first layer: input widths differ between body and hand (39 / 90 channels) -> separate launches
    for (int i = 0; i < n; ++i)
        TS_TRY(run_layer(ctx, net[i]->project, xin[i], ldx[i], B, L, nullptr, 0, pool[i]->buf(0), hid / 4, 0, hid / 4, s, &tmp));
    for (int i = 0; i < n; ++i) Sp[i] = &net[i]->s1;
    TS_TRY(run_stack_n(ctx, n, Sp, pool, cur, hid / 4, B, L, s, &o));
    cur = o;
    o = pool[0]->pick(cur);
    for (int i = 0; i < n; ++i) { Lp[i] = &net[i]->down1; xp[i] = pool[i]->buf(cur); op[i] = pool[i]->buf(o); ns[i] = hid / 2; }
    TS_TRY(run_layer_n(ctx, n, Lp, xp, hid / 4, B, L, nullptr, 0, op, hid / 2, nullptr, ns, s, &L));
    cur = o;
    for (int i = 0; i < n; ++i) Sp[i] = &net[i]->s2;
    TS_TRY(run_stack_n(ctx, n, Sp, pool, cur, hid / 2, B, L, s, &o));
    cur = o;
    o = pool[0]->pick(cur);
    for (int i = 0; i < n; ++i) { Lp[i] = &net[i]->down2; xp[i] = pool[i]->buf(cur); op[i] = pool[i]->buf(o); ns[i] = hid; }
    TS_TRY(run_layer_n(ctx, n, Lp, xp, hid / 2, B, L, nullptr, 0, op, hid, nullptr, ns, s, &L));
    cur = o;
    for (int i = 0; i < n; ++i) Sp[i] = &net[i]->s3;
    TS_TRY(run_stack_n(ctx, n, Sp, pool, cur, hid, B, L, s, &o));
    *out_idx = o;
    *H = L;
    return 0;
}

int run_trunk(ts_convnet *n, const float *x, int x_ld, int B, int T, hipStream_t s, int *out_idx, int *H) {
    ts_convnet *np[1] = {n};
    const float *xp[1] = {x};
    return run_trunk_n(1, np, xp, x_ld, B, T, s, out_idx, H);
}

}  // namespace

struct ts_vqvae {
    ts_ctx *ctx = nullptr;
    int in_dim = 0, emb = 0, ncode = 0, hid = 0, nres = 0;
    ts_convnet enc;
    DevBuf codebook, code_sq;
    DevBuf aft_table;   // [ncode][hid] = aft_vq_conv(embedding row): Decoder's first layer as a gather table
    ConvLayer aft;      // aft_vq_conv itself: the decoder's first layer on CONTINUOUS latents (vqvae_1d.AE, ncode == 0)
    Stack d1, d2, d3;
    ConvLayer up2, up3, project;
    struct Work {
        Pool pool;
        DevBuf z, lat;   // encoder output (B*H, emb), internal latents (B*H) int64
    };
    StreamWorks<Work> works;
    Work &work(hipStream_t s) { return works.get(s); }
};

namespace {

// n = 1 or 2 VQ-VAEs in lockstep (body + hand): encoder trunk -> pre_vq_conv -> arg-min (-> gather)
int vq_encode_n(int n, ts_vqvae *const *vq, const float *const *poses, int poses_ld, int B, int T, float *const *z_out,
                int64_t *const *lat_out, float *const *q_out, hipStream_t s, int *Hout) {
    ts_ctx *ctx = vq[0]->ctx;
    int idx = 0, H = 0, tmp = 0;
    ts_convnet *nets[2];
    for (int i = 0; i < n; ++i) nets[i] = &vq[i]->enc;
    TS_TRY(run_trunk_n(n, nets, poses, poses_ld, B, T, s, &idx, &H));
    const ConvLayer *Lp[2];
    const float *xp[2];
    float *zp[2];
    int ns[2];
    for (int i = 0; i < n; ++i) {
        ts_vqvae::Work &wk = vq[i]->work(s);
        TS_TRY(wk.z.ensure((size_t)B * H * vq[i]->emb * sizeof(float)));
        zp[i] = (z_out && z_out[i]) ? z_out[i] : wk.z.f();
        Lp[i] = &vq[i]->enc.pre_vq;
        xp[i] = vq[i]->enc.work(s).pool.buf(idx);
        ns[i] = vq[i]->emb;
    }
    TS_TRY(run_layer_n(ctx, n, Lp, xp, vq[0]->hid, B, H, nullptr, 0, zp, vq[0]->emb, nullptr, ns, s, &tmp));
    {
        MiscScope ms(ctx, s);
        for (int i = 0; i < n; ++i) {
            if (vq[i]->ncode == 0) continue;   // auto-encoder flavour (vqvae_1d.AE): no quantiser, z is the result
            TS_HIP(launch_vq_argmin(zp[i], vq[i]->emb, B * H, vq[i]->codebook.f(), vq[i]->code_sq.f(), vq[i]->ncode, vq[i]->emb,
                                    lat_out[i], 1, s));
            if (q_out && q_out[i])
                TS_HIP(launch_gather_rows(vq[i]->codebook.f(), vq[i]->emb, vq[i]->ncode, lat_out[i], 1, B * H, vq[i]->emb, q_out[i], vq[i]->emb, s));
        }
    }
    *Hout = H;
    return 0;
}

int vq_encode_impl(ts_vqvae *vq, const float *poses, int poses_ld, int B, int T, float *z_out, int64_t *lat_out,
                   float *q_out, hipStream_t s, int *Hout) {
    ts_vqvae *vp[1] = {vq};
    const float *pp[1] = {poses};
    float *zp[1] = {z_out}, *qp[1] = {q_out};
    int64_t *lp[1] = {lat_out};
    return vq_encode_n(1, vp, pp, poses_ld, B, T, zp, lp, qp, s, Hout);
}

// n = 1 or 2 decoders in lockstep: first layer (aft_vq table gather for codes, or aft_vq_conv on continuous latents z)
// -> stacks / up-convs -> project into out[.., col0_i .. col0_i+in_dim_i)
int vq_decode_n(int n, ts_vqvae *const *vq, const int64_t *const *lat, const float *const *z, int B, int H, float *out,
                int out_ld, const int *col0, hipStream_t s) {
    ts_ctx *ctx = vq[0]->ctx;
    const int hid = vq[0]->hid;
    Pool *pool[2];
    for (int i = 0; i < n; ++i) {
        if (vq[i]->hid != hid) return fail("paired VQ-VAEs must have the same width");
        pool[i] = &vq[i]->work(s).pool;
        TS_TRY(pool[i]->ensure((size_t)B * H * hid));
        if (z && z[i]) {
            int tmp = 0;
            TS_TRY(run_layer(ctx, vq[i]->aft, z[i], vq[i]->emb, B, H, nullptr, 0, pool[i]->buf(0), hid, 0, hid, s, &tmp));
        } else {
            if (vq[i]->ncode == 0) return fail("this network has no codebook (auto-encoder): decode continuous latents");
            MiscScope ms(ctx, s);
            TS_HIP(launch_gather_rows(vq[i]->aft_table.f(), hid, vq[i]->ncode, lat[i], 1, B * H, hid, pool[i]->buf(0), hid, s));
        }
    }
    int cur = 0, o = 0, L = H;
    const ConvLayer *Lp[2];
    const Stack *Sp[2];
    const float *xp[2];
    float *op[2];
    int ns[2];
    for (int i = 0; i < n; ++i) Sp[i] = &vq[i]->d1;
    TS_TRY(run_stack_n(ctx, n, Sp, pool, cur, hid, B, L, s, &o));
    cur = o;
    o = pool[0]->pick(cur);
    for (int i = 0; i < n; ++i) { Lp[i] = &vq[i]->up2; xp[i] = pool[i]->buf(cur); op[i] = pool[i]->buf(o); ns[i] = hid / 2; }
    TS_TRY(run_layer_n(ctx, n, Lp, xp, hid, B, L, nullptr, 0, op, hid / 2, nullptr, ns, s, &L));
    cur = o;
    for (int i = 0; i < n; ++i) Sp[i] = &vq[i]->d2;
    TS_TRY(run_stack_n(ctx, n, Sp, pool, cur, hid / 2, B, L, s, &o));
    cur = o;
    o = pool[0]->pick(cur);
    for (int i = 0; i < n; ++i) { Lp[i] = &vq[i]->up3; xp[i] = pool[i]->buf(cur); op[i] = pool[i]->buf(o); ns[i] = hid / 4; }
    TS_TRY(run_layer_n(ctx, n, Lp, xp, hid / 2, B, L, nullptr, 0, op, hid / 4, nullptr, ns, s, &L));
    cur = o;
    for (int i = 0; i < n; ++i) Sp[i] = &vq[i]->d3;
    TS_TRY(run_stack_n(ctx, n, Sp, pool, cur, hid / 4, B, L, s, &o));
    int tmp = 0;
    // last layer: output widths differ (39 / 90) -> separate launches into the two column ranges of `out`
    for (int i = 0; i < n; ++i)
        TS_TRY(run_layer(ctx, vq[i]->project, pool[i]->buf(o), hid / 4, B, L, nullptr, 0, out, out_ld, col0[i], vq[i]->in_dim, s, &tmp));
    return 0;
}

int vq_decode_impl(ts_vqvae *vq, const int64_t *lat, int B, int H, float *out, int out_ld, int col0, hipStream_t s) {
    ts_vqvae *vp[1] = {vq};
    const int64_t *lp[1] = {lat};
    return vq_decode_n(1, vp, lp, nullptr, B, H, out, out_ld, &col0, s);
}

}  // namespace

namespace ts {
int convnet_hidden(const ts_convnet *n) { return n->hid; }
int vqvae_in_dim(const ts_vqvae *v) { return v->in_dim; }
}  // namespace ts

// ---------------------------------------------------------------------------------------------- C ABI
extern "C" {

const char *ts_last_error(void) { return ts::last_error(); }
const char *ts_version(void) { return "talkshow_hip 0.1 gfx950 fp32-mfma"; }

int ts_ctx_create(int device, ts_ctx **out) {
    if (!out) return fail("ts_ctx_create: null out");
    int n = 0;
    TS_HIP(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail("ts_ctx_create: no such HIP device " + std::to_string(device));
    TS_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    TS_HIP(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
        return fail(std::string("ts_ctx_create: this library is built for gfx950 (MI355X) only, device is ") + prop.gcnArchName);
    (void)ts::knobs();   // every TS_* test lever is parsed here, once per process; launch paths only read the struct
    std::unique_ptr<ts_ctx> c(new ts_ctx());
    c->device = device;
    const int m1 = -1;
    TS_TRY(c->neg1.upload(&m1, sizeof(int)));
    TS_HIP(ts::skinny_init(device));
    TS_HIP(ts::conv_sk_probe_xcd_map(device));   // the stream-K band's one hardware assumption, checked once (conv_gemm_ring.hip)
    *out = c.release();
    return 0;
}
void ts_ctx_destroy(ts_ctx *ctx) { delete ctx; }

int ts_prof_enable(ts_ctx *ctx, int on) {
    if (!ctx) return fail("null ctx");
    ctx->prof.on = on != 0;
    return 0;
}
int ts_prof_read(ts_ctx *ctx, double *ms_out, int64_t *launches_out, double *flops_out, int reset) {
    return ts_prof_read_n(ctx, 3, ms_out, launches_out, flops_out, reset);
}
int ts_prof_read_n(ts_ctx *ctx, int n_families, double *ms_out, int64_t *launches_out, double *flops_out, int reset) {
    if (!ctx) return fail("null ctx");
    if (n_families < 1 || n_families > FAM_COUNT) return fail("ts_prof_read_n: 1..4 families");
    TS_TRY(ctx->prof.collect());
    for (int i = 0; i < n_families; ++i) {
        if (ms_out) ms_out[i] = ctx->prof.ms[i];
        if (launches_out) launches_out[i] = ctx->prof.launches[i];
        if (flops_out) flops_out[i] = ctx->prof.flops[i];
    }
    if (reset) ctx->prof.reset();
    return 0;
}

int ts_audioenc_create(ts_ctx *ctx, const ts_tensor *sd_, int n, int in_dim, int num_hiddens, int nres, ts_convnet **out) {
    if (!ctx || !sd_ || !out) return fail("ts_audioenc_create: null argument");
    TS_HIP(hipSetDevice(ctx->device));
    StateDict sd(sd_, n);
    std::unique_ptr<ts_convnet> net(new ts_convnet());
    TS_TRY(pack_trunk(ctx, sd, "", in_dim, num_hiddens, nres, net.get()));
    *out = net.release();
    return 0;
}
void ts_convnet_destroy(ts_convnet *net) { delete net; }

int ts_audioenc_forward(ts_convnet *net, const float *mfcc, int B, int T, float *feat, void *stream) {
    if (!net || !mfcc || !feat) return fail("ts_audioenc_forward: null argument");
    hipStream_t s = (hipStream_t)stream;
    int idx = 0, H = 0;
    TS_TRY(run_trunk(net, mfcc, 0, B, T, s, &idx, &H));
    TS_HIP(hipMemcpyAsync(feat, net->work(s).pool.buf(idx), (size_t)B * H * net->hid * sizeof(float), hipMemcpyDeviceToDevice, s));
    return 0;
}

int ts_vqvae_create(ts_ctx *ctx, const ts_tensor *sd_, int n, int in_dim, int emb, int ncode, int hid, int nres,
                    ts_vqvae **out) {
    if (!ctx || !sd_ || !out) return fail("ts_vqvae_create: null argument");
    if (emb % 32 != 0) return fail("embedding_dim must be a multiple of 32");
    TS_HIP(hipSetDevice(ctx->device));
    StateDict sd(sd_, n);
    std::unique_ptr<ts_vqvae> vq(new ts_vqvae());
    vq->ctx = ctx;
    vq->in_dim = in_dim;
    vq->emb = emb;
    vq->ncode = ncode;
    vq->hid = hid;
    vq->nres = nres;
    TS_TRY(pack_trunk(ctx, sd, "encoder.", in_dim, hid, nres, &vq->enc));
    vq->enc.has_pre_vq = true;
    TS_TRY(pack_conv_layer(sd, "encoder.pre_vq_conv", "", "", 0, 1, hid, emb, 0, &vq->enc.pre_vq));
    TS_TRY(pack_conv_layer(sd, "decoder.aft_vq_conv", "", "", 0, 1, emb, hid, 0, &vq->aft));
    if (ncode > 0) {
        const float *cb = sd.get("vq_layer.embeddings", {ncode, emb});
        if (!cb) return 1;
        TS_TRY(vq->codebook.upload(cb, (size_t)ncode * emb * sizeof(float)));
        TS_TRY(vq->code_sq.ensure((size_t)ncode * sizeof(float)));
        TS_HIP(launch_row_sqnorm(vq->codebook.f(), ncode, emb, vq->code_sq.f(), 0));
        TS_TRY(vq->aft_table.ensure((size_t)ncode * hid * sizeof(float)));
        // aft_vq_conv applied to every codebook row once: Decoder.forward's first layer becomes a row gather
        ConvParams p;
        conv_layer_params(vq->aft, vq->codebook.f(), emb, 1, ncode, nullptr, 0, vq->aft_table.f(), hid, 0, hid, &p);
        TS_HIP(launch_conv_gemm(p, 0, 0));
        TS_HIP(hipStreamSynchronize(0));
    }
    TS_TRY(pack_stack(sd, "decoder._dec_1", hid, nres, &vq->d1));
    TS_TRY(pack_conv_layer(sd, "decoder._up_2.conv", "decoder._up_2.norm", "decoder._up_2.residual_layer", 2, 4, hid,
                           hid / 2, 1, &vq->up2));
    TS_TRY(pack_stack(sd, "decoder._dec_2", hid / 2, nres, &vq->d2));
    TS_TRY(pack_conv_layer(sd, "decoder._up_3.conv", "decoder._up_3.norm", "decoder._up_3.residual_layer", 2, 4,
                           hid / 2, hid / 4, 1, &vq->up3));
    TS_TRY(pack_stack(sd, "decoder._dec_3", hid / 4, nres, &vq->d3));
    TS_TRY(pack_conv_layer(sd, "decoder.project", "", "", 0, 1, hid / 4, in_dim, 0, &vq->project));
    *out = vq.release();
    return 0;
}
void ts_vqvae_destroy(ts_vqvae *vq) { delete vq; }

int ts_vqvae_encode(ts_vqvae *vq, const float *poses, int B, int T, float *z, int64_t *lat, float *q, void *stream) {
    if (!vq || !poses) return fail("ts_vqvae_encode: null argument");
    if (vq->ncode > 0 && !lat) return fail("ts_vqvae_encode: latents_dev is required for a network with a codebook");
    if (vq->ncode == 0 && (!z || lat || q)) return fail("ts_vqvae_encode: an auto-encoder (num_embeddings = 0) produces z only");
    int H = 0;
    return vq_encode_impl(vq, poses, 0, B, T, z, lat, q, (hipStream_t)stream, &H);
}

int ts_vqvae_decode(ts_vqvae *vq, const int64_t *lat, int B, int H, float *out, int out_ld, int col0, void *stream) {
    if (!vq || !lat || !out) return fail("ts_vqvae_decode: null argument");
    if (out_ld < col0 + vq->in_dim) return fail("ts_vqvae_decode: out_ld too small");
    return vq_decode_impl(vq, lat, B, H, out, out_ld, col0, (hipStream_t)stream);
}

int ts_vqvae_decode_z(ts_vqvae *vq, const float *z, int B, int H, float *out, int out_ld, int col0, void *stream) {
    if (!vq || !z || !out) return fail("ts_vqvae_decode_z: null argument");
    if (out_ld < col0 + vq->in_dim) return fail("ts_vqvae_decode_z: out_ld too small");
    ts_vqvae *vp[1] = {vq};
    const float *zp[1] = {z};
    const int64_t *lp[1] = {nullptr};
    return vq_decode_n(1, vp, lp, zp, B, H, out, out_ld, &col0, (hipStream_t)stream);
}

int ts_vqvae_forward(ts_vqvae *vq, const float *poses, int B, int T, int64_t *lat, float *out, int out_ld, int col0,
                     void *stream) {
    if (!vq || !poses || !out) return fail("ts_vqvae_forward: null argument");
    if (vq->ncode == 0) return fail("ts_vqvae_forward: auto-encoder handle; use ts_vqvae_encode + ts_vqvae_decode_z");
    hipStream_t s = (hipStream_t)stream;
    int H = 0;
    int64_t *l = lat;
    if (!l) {
        ts_vqvae::Work &wk = vq->work(s);
        TS_TRY(wk.lat.ensure((size_t)B * (T / 4 + 1) * sizeof(int64_t)));
        l = static_cast<int64_t *>(wk.lat.p);
    }
    TS_TRY(vq_encode_impl(vq, poses, 0, B, T, nullptr, l, nullptr, s, &H));
    return vq_decode_impl(vq, l, B, H, out, out_ld, col0, s);
}

int ts_body_vq_infer(ts_vqvae *vb, ts_vqvae *vh, const float *poses, int B, int T, int64_t *codes, float *recon,
                     void *stream) {
    if (!vb || !vh || !poses) return fail("ts_body_vq_infer: null argument");
    if (!codes && !recon) return fail("ts_body_vq_infer: nothing to produce");
    hipStream_t s = (hipStream_t)stream;
    const int db = vb->in_dim, dh = vh->in_dim, ld = db + dh;
    // gt_poses[..., :each_dim[1]] / [..., each_dim[1]:] (smplx_body_vq.py:274-275): strided views of the same rows;
    // body and hand networks run in lockstep (grouped launches)
    ts_vqvae *vqs[2] = {vb, vh};
    const float *pp[2] = {poses, poses + db};
    int64_t *lp[2];
    for (int k = 0; k < 2; ++k) {
        ts_vqvae::Work &wk = vqs[k]->work(s);
        TS_TRY(wk.lat.ensure((size_t)B * (T / 4 + 1) * sizeof(int64_t)));
        lp[k] = static_cast<int64_t *>(wk.lat.p);
    }
    int H = 0;
    TS_TRY(vq_encode_n(2, vqs, pp, ld, B, T, nullptr, lp, nullptr, s, &H));
    if (recon) {
        const int col0[2] = {0, db};
        const int64_t *lc[2] = {lp[0], lp[1]};
        TS_TRY(vq_decode_n(2, vqs, lc, nullptr, B, H, recon, ld, col0, s));
    }
    if (codes)   // codes (B,H,2): column k
        for (int k = 0; k < 2; ++k)
            TS_HIP(hipMemcpy2DAsync(codes + k, 2 * sizeof(int64_t), lp[k], sizeof(int64_t), sizeof(int64_t), (size_t)B * H,
                                    hipMemcpyDeviceToDevice, s));
    return 0;
}

// VQVAE.decode of body and hand latents into the two column ranges of one (B,4H,body+hand) buffer, in lockstep
int ts_vqvae_decode_pair(ts_vqvae *vb, ts_vqvae *vh, const int64_t *lat_body, const int64_t *lat_hand, int B, int H,
                         float *out, void *stream) {
    if (!vb || !vh || !lat_body || !lat_hand || !out) return fail("ts_vqvae_decode_pair: null argument");
    ts_vqvae *vqs[2] = {vb, vh};
    const int64_t *lc[2] = {lat_body, lat_hand};
    const int col0[2] = {0, vb->in_dim};
    return vq_decode_n(2, vqs, lc, nullptr, B, H, out, vb->in_dim + vh->in_dim, col0, (hipStream_t)stream);
}

}  // extern "C"
