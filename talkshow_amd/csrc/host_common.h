// Host-side helpers shared by models.cpp / pixelcnn.cpp / api.cpp: error channel, device buffers, the
// reference-state_dict view, the launch wrappers that feed the per-family profiler.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <map>
#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/talkshow_hip.h"
#include "../../include/talkshow_hip_debug.h"
#include "kernels.h"

namespace ts {

void set_error(const std::string &m);
inline int fail(const std::string &m) {
    set_error(m);
    return 1;
}

#define TS_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t e__ = (expr);                                                                  \
        if (e__ != hipSuccess) return ::ts::fail(std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)
#define TS_TRY(expr)              \
    do {                          \
        int r__ = (expr);         \
        if (r__ != 0) return r__; \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    int ensure(size_t n) {   // grow-only; contents are NOT preserved
        if (n <= bytes) return 0;
        release();
        TS_HIP(hipMalloc(&p, n));
        bytes = n;
        return 0;
    }
    int upload(const void *host, size_t n) {
        TS_TRY(ensure(n));
        TS_HIP(hipMemcpy(p, host, n, hipMemcpyHostToDevice));
        return 0;
    }
    float *f() const { return static_cast<float *>(p); }
    int *i() const { return static_cast<int *>(p); }
};

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// ---- per-stream scratch --------------------------------------------------------------------------------------
// Every model object keeps one Work (activation arena, captured graphs) per HIP stream, created on the stream's first
// call, so independent batches can be in flight on different streams against one weight copy.  Lookups are serialised:
// two host threads may make their first call on two new streams at the same time.  A Work itself is only ever touched
// by the thread driving its stream (contract in talkshow_hip.h: one host thread per stream at a time).
// ts_stream_destroy() evicts the Work of the dying stream from every live object (a later stream could reuse the handle
// value and would otherwise inherit graphs captured for the old one).
struct StreamScoped {
    StreamScoped();
    virtual ~StreamScoped();
    virtual void drop_stream(hipStream_t s) = 0;
    StreamScoped(const StreamScoped &) = delete;
    StreamScoped &operator=(const StreamScoped &) = delete;
};
void drop_stream_everywhere(hipStream_t s);

template <class W>
struct StreamWorks : StreamScoped {
    std::mutex mu;
    std::map<hipStream_t, std::unique_ptr<W>> m;
    W &get(hipStream_t s) {
        std::lock_guard<std::mutex> g(mu);
        auto &w = m[s];
        if (!w) w.reset(new W());
        return *w;
    }
    W *find(hipStream_t s) {
        std::lock_guard<std::mutex> g(mu);
        auto it = m.find(s);
        return it == m.end() ? nullptr : it->second.get();
    }
    void drop_stream(hipStream_t s) override {
        std::lock_guard<std::mutex> g(mu);
        m.erase(s);
    }
};

// relaxed atomic accumulate for the always-on diagnostic counters (several host threads may launch concurrently)
inline void atomic_add(std::atomic<double> &a, double v) {
    double cur = a.load(std::memory_order_relaxed);
    while (!a.compare_exchange_weak(cur, cur + v, std::memory_order_relaxed)) {
    }
}

// view of a reference state_dict, "module." prefixes stripped (nets/smplx_body_pixel.py:119-126)
struct StateDict {
    std::map<std::string, const ts_tensor *> m;
    StateDict(const ts_tensor *t, int n) {
        for (int i = 0; i < n; ++i) {
            std::string k = t[i].name ? t[i].name : "";
            size_t pos;
            while ((pos = k.find("module.")) != std::string::npos) k.erase(pos, 7);
            m[k] = &t[i];
        }
    }
    // returns nullptr (and sets the error) if missing or shape mismatch
    const float *get(const std::string &key, std::initializer_list<int64_t> shape) const {
        auto it = m.find(key);
        if (it == m.end() || !it->second->data) {
            set_error("state_dict: missing key '" + key + "'");
            return nullptr;
        }
        const ts_tensor *t = it->second;
        bool ok = t->ndim == (int)shape.size();
        int i = 0;
        for (int64_t s : shape) {
            if (ok && t->shape[i] != s) ok = false;
            ++i;
        }
        if (!ok) {
            std::string got = "(", want = "(";
            for (int j = 0; j < t->ndim; ++j) got += std::to_string(t->shape[j]) + ",";
            for (int64_t s : shape) want += std::to_string(s) + ",";
            set_error("state_dict: shape mismatch for '" + key + "': got " + got + ") want " + want + ")");
            return nullptr;
        }
        return t->data;
    }
};

// ---- per-family device-time profiler (ts_prof_*) ------------------------------------------------------------
enum { FAM_CONV = 0, FAM_SKINNY = 1, FAM_MISC = 2, FAM_ATTN = 3, FAM_COUNT = 4 };

struct Profiler {
    bool on = false;
    struct Rec {
        hipEvent_t a, b;
        int fam;
        double flops = 0;
        std::string tag;   // TS_PROF_LOG=1: per-launch line printed by collect()
    };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    double ms[FAM_COUNT] = {};
    double flops[FAM_COUNT] = {};
    long launches[FAM_COUNT] = {};
    hipEvent_t get_event();
    void begin(int fam, hipStream_t s);
    void end(hipStream_t s);
    int collect();   // synchronises the recorded events and folds them into ms[]
    void reset();
    ~Profiler();
};

}  // namespace ts

struct ts_ctx {
    int device = 0;
    ts::Profiler prof;
    // always-on launch / algorithmic-flop counters per kernel family (cheap host-side bookkeeping)
    std::atomic<long> n_launch[ts::FAM_COUNT] = {};
    std::atomic<double> n_flops[ts::FAM_COUNT] = {};
    ts::DevBuf neg1;   // a single int32 -1 (gather index meaning "zero row")
};

namespace ts {

int run_conv(ts_ctx *ctx, const ConvParams &p, int tile, hipStream_t s);
int run_skinny(ts_ctx *ctx, const SkinnyParams &p, hipStream_t s);
int run_skinny_batch(ts_ctx *ctx, const SkinnyParams *const *ps, int n, hipStream_t s);
// misc launches are wrapped with this scope guard so they show up under FAM_MISC
struct MiscScope {
    ts_ctx *ctx;
    hipStream_t s;
    MiscScope(ts_ctx *c, hipStream_t st, int fam = FAM_MISC, double flops = 0.0) : ctx(c), s(st) {
        if (ctx->prof.on) {
            ctx->prof.begin(fam, s);
            ctx->prof.flops[fam] += flops;
        }
    }
    ~MiscScope() {
        if (ctx->prof.on) ctx->prof.end(s);
    }
};

// ---- a folded + packed convolution layer -------------------------------------------------------------------
struct ConvLayer {
    int kind = 0;   // 0: stride-1 (k1 / k3), 1: down (k4 s2 p1), 2: up (ConvTranspose k4 s2 p1)
    int cin = 0, cin_pad = 0, cout = 0, cout_pad = 0, npad = 0, ktot = 0;
    int act = 0;
    int ngroups = 1, nseg = 0;
    ConvSeg segs[2][4];
    DevBuf w, bias;   // [ngroups][npad][ktot], [ngroups][npad]
    DevBuf wp;        // face generator, x3 plan only: w as bf16 plane images (launch_split_weight_planes)
    std::string name;
};

// Packs one reference layer: conv `conv_key` (+ BatchNorm `norm_key` folded, + parallel residual conv
// `res_key` folded) into `out`.  kind as above; k = kernel size for kind 0 (1 or 3).
int pack_conv_layer(const StateDict &sd, const std::string &conv_key, const std::string &norm_key,
                    const std::string &res_key, int kind, int k, int cin, int cout, int act, ConvLayer *out);
// generic: pack an (N,K) row-major matrix (+bias) as a k1 ConvLayer (used for the PixelCNN audio precompute)
int pack_linear_layer(const float *w, long ldw, const float *bias, int N, int K, ConvLayer *out);

// raw form: w (cout, cin, K) reference layout, optional bias; taps[k] = input row shift of kernel index k
int pack_conv_raw(const float *w, const float *bias, int cout, int cin, int K, const int *taps, int act, ConvLayer *out);

// Fills ConvParams for `layer` applied to x (B, Lin, ldx) -> out; returns Lout (rows per clip of the output buffer).
// For kind 2 the output buffer has 2*Lin rows of cout_pad (or ldo) floats.
int conv_layer_params(const ConvLayer &layer, const float *x, int ldx, int B, int Lin, const float *res, int ldr,
                      float *out, int ldo, int out_col0, int n_store, ConvParams *p);

// accessors so that api.cpp needs no knowledge of the model structs
int convnet_hidden(const ts_convnet *n);
int vqvae_in_dim(const ts_vqvae *v);

}  // namespace ts
