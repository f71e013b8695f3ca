// C ABI of the evaluation reductions (eval.hip): what evaluation/FGD.py and evaluation/metrics.py of the reference compute
// on the CPU after the hot path, here as streaming reductions on the GPU next to the generated poses.
#include "host_common.h"

using namespace ts;

namespace ts {
int eval_feat_stats_workgroups(long n);
hipError_t launch_feat_stats(const float *x, long n, int D, double *scratch, double *sum_out, double *outer_out, hipStream_t s);
hipError_t launch_l1_total(const float *a, const float *b, long n, double *scratch, double *out, hipStream_t s);
hipError_t launch_body_loss(const float *gt, const float *prs, int B, int T, int J, int Jl, int Tl, double *scratch, double *out3,
                            hipStream_t s);
hipError_t launch_diversity(const float *kps, int bs, long L, double *scratch, double *out, hipStream_t s);
}  // namespace ts

namespace {
struct EvalWork {
    DevBuf scratch;
};
EvalWork &eval_work(hipStream_t s) {
    static StreamWorks<EvalWork> works;
    return works.get(s);
}
}  // namespace

extern "C" {

int ts_eval_feat_stats(ts_ctx *ctx, const float *feat, int64_t n, int D, double *stats, void *stream) {
    if (!ctx || !feat || !stats) return fail("ts_eval_feat_stats: null argument");
    if (n < 1) return fail("ts_eval_feat_stats: no rows");
    if (D != 32 && D != 64 && D != 128) return fail("ts_eval_feat_stats: feature width must be 32, 64 or 128");
    hipStream_t s = (hipStream_t)stream;
    EvalWork &w = eval_work(s);
    const size_t width = (size_t)D + (size_t)D * D;
    TS_TRY(w.scratch.ensure((size_t)eval_feat_stats_workgroups(n) * width * sizeof(double)));
    MiscScope ms(ctx, s);
    TS_HIP(launch_feat_stats(feat, n, D, static_cast<double *>(w.scratch.p), stats, stats + D, s));
    return 0;
}

int ts_eval_l1_total(ts_ctx *ctx, const float *a, const float *b, int64_t n, double *out, void *stream) {
    if (!ctx || !a || !b || !out) return fail("ts_eval_l1_total: null argument");
    if (n < 1) return fail("ts_eval_l1_total: empty input");
    hipStream_t s = (hipStream_t)stream;
    EvalWork &w = eval_work(s);
    TS_TRY(w.scratch.ensure(1024 * sizeof(double)));
    MiscScope ms(ctx, s);
    TS_HIP(launch_l1_total(a, b, n, static_cast<double *>(w.scratch.p), out, s));
    return 0;
}

int ts_eval_body_loss(ts_ctx *ctx, const float *gt, const float *prs, int B, int T, int J, int J_lvd, int T_lvd, double *out3,
                      void *stream) {
    if (!ctx || !gt || !prs || !out3) return fail("ts_eval_body_loss: null argument");
    if (B < 1 || T < 2 || J < 1 || J_lvd < 0 || J_lvd > J || T_lvd < 2 || T_lvd > T) return fail("ts_eval_body_loss: bad shape");
    hipStream_t s = (hipStream_t)stream;
    EvalWork &w = eval_work(s);
    TS_TRY(w.scratch.ensure((size_t)T * 3 * sizeof(double)));
    MiscScope ms(ctx, s);
    TS_HIP(launch_body_loss(gt, prs, B, T, J, J_lvd, T_lvd, static_cast<double *>(w.scratch.p), out3, s));
    return 0;
}

int ts_eval_diversity(ts_ctx *ctx, const float *kps, int bs, int64_t L, double *out, void *stream) {
    if (!ctx || !kps || !out) return fail("ts_eval_diversity: null argument");
    if (bs < 2 || L < 1) return fail("ts_eval_diversity: needs at least two sequences");
    hipStream_t s = (hipStream_t)stream;
    EvalWork &w = eval_work(s);
    TS_TRY(w.scratch.ensure((size_t)bs * (bs - 1) / 2 * sizeof(double)));
    MiscScope ms(ctx, s);
    TS_HIP(launch_diversity(kps, bs, L, static_cast<double *>(w.scratch.p), out, s));
    return 0;
}

}  // extern "C"
