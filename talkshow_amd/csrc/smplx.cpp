// Batched SMPL-X joints / vertices on the device.  Replaces the per-frame float64 CPU loop of
// scripts/demo.py:122-152 and data_utils/get_j.py:20-50 (third-party package smplx ~= 0.1.28: PARITY UNPINNED, the
// algorithm is restated from its publication; oracle/smplx_oracle.py is the float64 checker on synthetic parameters).
//
// Host-side preparation, once per model:
//   * the vertices the joint list needs (extra joints picked from the mesh + the 3 corners of every landmark triangle,
//     ~170 of 10 475) get their own slice of the blend-shape matrix, so joints-only evaluation (the metrics path of
//     scripts/test_body.py) multiplies a 522-row matrix instead of a 31 425-row one;
//   * joint regression is linear in the shape coefficients: J(beta) = J_regressor v_template + (J_regressor shapedirs) beta
//     is folded into a 165-row matrix (float64 products, rounded once);
//   * lbs_weights (V x 55, mostly zeros) become <= KW (bone, weight) pairs per vertex.
#include <algorithm>
#include <cmath>

#include "host_common.h"

using namespace ts;

namespace ts {
hipError_t launch_smplx_pose_prepare(const float *rows, int row_ld, const float *betas, int betas_per_row, int NB, int NE,
                                     int expr_off, const int *src_off, const float *pose_mean, int J, float *rot, float *X,
                                     int Kpad, long N, hipStream_t s);
hipError_t launch_smplx_rigid_chain(const float *rot, const float *jrest, int jr_ld, const int *parents, int J, long N, float *G,
                                    float *A, float *joints, int NJ, hipStream_t s);
hipError_t launch_smplx_skin(const float *vposed, int vp_ld, const float *A, int J, const int *bone, const float *wgt, int KW,
                             int U, long N, float *out, long out_frame_stride, hipStream_t s);
hipError_t launch_smplx_joints_tail(const float *vs, long vs_frame_stride, const int *extra_map, int n_extra, const int *lmk_map,
                                    const float *bary, int n_lmk, int J, float *joints, int NJ, long N, hipStream_t s);
}  // namespace ts

struct ts_smplx {
    ts_ctx *ctx = nullptr;
    int V = 0, J = 0, NB = 0, NE = 0, S = 0, P = 0, Kpad = 0, U = 0, KW = 0, n_extra = 0, n_lmk = 0, NJ = 0;
    bool with_vertices = false;
    ConvLayer blend_sub, blend_full, jdirs;      // [3U | 3V | 3J] x Kpad, bias = template
    DevBuf src_off, pose_mean, parents, bone_sub, wgt_sub, bone_full, wgt_full, extra_map, lmk_map, bary;
    struct Work {
        DevBuf X, rot, jrest, G, A, vposed, vs;
    };
    StreamWorks<Work> works;
};

namespace {

// (bone, weight) lists of the given vertices, padded to KW with (0, 0)
int sparse_weights(const float *lbs, int J, const std::vector<int> &verts, int KW, DevBuf *bone, DevBuf *wgt) {
    std::vector<int> b((size_t)verts.size() * KW, 0);
    std::vector<float> w((size_t)verts.size() * KW, 0.f);
    for (size_t i = 0; i < verts.size(); ++i) {
        int k = 0;
        for (int j = 0; j < J; ++j) {
            const float x = lbs[(size_t)verts[i] * J + j];
            if (x != 0.f) {
                b[i * KW + k] = j;
                w[i * KW + k] = x;
                ++k;
            }
        }
    }
    TS_TRY(bone->upload(b.data(), b.size() * sizeof(int)));
    TS_TRY(wgt->upload(w.data(), w.size() * sizeof(float)));
    return 0;
}

// rows (vertex v, coordinate c) of [shapedirs | posedirs^T] with bias v_template, for the listed vertices
int pack_blend(const float *v_template, const float *shapedirs, const float *posedirs, int V, int S, int P, int Kpad,
               const std::vector<int> &verts, ConvLayer *L) {
    const size_t rows = verts.size() * 3;
    std::vector<float> w(rows * Kpad, 0.f), bias(rows);
    for (size_t i = 0; i < verts.size(); ++i)
        for (int c = 0; c < 3; ++c) {
            float *wr = &w[(i * 3 + c) * Kpad];
            const size_t vc = (size_t)verts[i] * 3 + c;
            for (int k = 0; k < S; ++k) wr[k] = shapedirs[vc * S + k];
            for (int k = 0; k < P; ++k) wr[S + k] = posedirs[(size_t)k * V * 3 + vc];
            bias[i * 3 + c] = v_template[vc];
        }
    return pack_linear_layer(w.data(), Kpad, bias.data(), (int)rows, Kpad, L);
}

}  // namespace

extern "C" {

int ts_smplx_create(ts_ctx *ctx, int V, int J, int n_betas, int n_expr, const float *v_template, const float *shapedirs,
                    const float *posedirs, const float *J_regressor, const int32_t *parents, const float *lbs_weights,
                    const float *pose_mean, const int32_t *pose_src_offset, int n_extra, const int32_t *extra_idx, int n_lmk,
                    const int32_t *lmk_faces, const float *lmk_bary, int with_vertices, ts_smplx **out) {
    if (!ctx || !v_template || !shapedirs || !posedirs || !J_regressor || !parents || !lbs_weights || !pose_mean ||
        !pose_src_offset || !out)
        return fail("ts_smplx_create: null argument");
    if (V < 1 || J < 1 || n_betas < 0 || n_expr < 0 || n_extra < 0 || n_lmk < 0) return fail("ts_smplx_create: bad shape");
    if ((n_extra && !extra_idx) || (n_lmk && (!lmk_faces || !lmk_bary))) return fail("ts_smplx_create: null index array");
    TS_HIP(hipSetDevice(ctx->device));
    std::unique_ptr<ts_smplx> m(new ts_smplx());
    m->ctx = ctx;
    m->V = V;
    m->J = J;
    m->NB = n_betas;
    m->NE = n_expr;
    m->S = n_betas + n_expr;
    m->P = (J - 1) * 9;
    m->Kpad = round_up(m->S + m->P, 32);
    m->n_extra = n_extra;
    m->n_lmk = n_lmk;
    m->NJ = J + n_extra + n_lmk;
    m->with_vertices = with_vertices != 0;
    for (int j = 0; j < J; ++j)
        if (parents[j] >= j || (j > 0 && parents[j] < 0)) return fail("ts_smplx_create: parents must precede their children");
    // needed vertices, de-duplicated, and the maps from joint-list entries into that subset
    std::vector<int> need, extra_map(n_extra), lmk_map((size_t)n_lmk * 3);
    std::map<int, int> slot;
    auto use = [&](int v) -> int {
        auto it = slot.find(v);
        if (it != slot.end()) return it->second;
        const int k = (int)need.size();
        slot[v] = k;
        need.push_back(v);
        return k;
    };
    for (int e = 0; e < n_extra; ++e) {
        if (extra_idx[e] < 0 || extra_idx[e] >= V) return fail("ts_smplx_create: extra joint vertex index out of range");
        extra_map[e] = use(extra_idx[e]);
    }
    for (int l = 0; l < 3 * n_lmk; ++l) {
        if (lmk_faces[l] < 0 || lmk_faces[l] >= V) return fail("ts_smplx_create: landmark vertex index out of range");
        lmk_map[l] = use(lmk_faces[l]);
    }
    if (need.empty()) need.push_back(0);
    m->U = (int)need.size();
    int kw = 1;
    for (int v = 0; v < V; ++v) {
        int k = 0;
        for (int j = 0; j < J; ++j) k += lbs_weights[(size_t)v * J + j] != 0.f;
        kw = std::max(kw, k);
    }
    m->KW = kw;
    TS_TRY(pack_blend(v_template, shapedirs, posedirs, V, m->S, m->P, m->Kpad, need, &m->blend_sub));
    TS_TRY(sparse_weights(lbs_weights, J, need, kw, &m->bone_sub, &m->wgt_sub));
    if (m->with_vertices) {
        std::vector<int> all(V);
        for (int v = 0; v < V; ++v) all[v] = v;
        TS_TRY(pack_blend(v_template, shapedirs, posedirs, V, m->S, m->P, m->Kpad, all, &m->blend_full));
        TS_TRY(sparse_weights(lbs_weights, J, all, kw, &m->bone_full, &m->wgt_full));
    }
    {   // rest joints as a linear map of the shape coefficients (float64 products, one rounding)
        std::vector<double> jd((size_t)J * 3 * m->S, 0.0), jt((size_t)J * 3, 0.0);
        for (int j = 0; j < J; ++j)
            for (int v = 0; v < V; ++v) {
                const double r = J_regressor[(size_t)j * V + v];
                if (r == 0.0) continue;
                for (int c = 0; c < 3; ++c) {
                    jt[(size_t)j * 3 + c] += r * v_template[(size_t)v * 3 + c];
                    const float *sd = shapedirs + ((size_t)v * 3 + c) * m->S;
                    double *dst = &jd[((size_t)j * 3 + c) * m->S];
                    for (int k = 0; k < m->S; ++k) dst[k] += r * sd[k];
                }
            }
        std::vector<float> w((size_t)J * 3 * m->Kpad, 0.f), b((size_t)J * 3);
        for (int r = 0; r < J * 3; ++r) {
            for (int k = 0; k < m->S; ++k) w[(size_t)r * m->Kpad + k] = (float)jd[(size_t)r * m->S + k];
            b[r] = (float)jt[r];
        }
        TS_TRY(pack_linear_layer(w.data(), m->Kpad, b.data(), J * 3, m->Kpad, &m->jdirs));
    }
    TS_TRY(m->src_off.upload(pose_src_offset, (size_t)J * sizeof(int)));
    TS_TRY(m->pose_mean.upload(pose_mean, (size_t)J * 3 * sizeof(float)));
    TS_TRY(m->parents.upload(parents, (size_t)J * sizeof(int)));
    if (n_extra) TS_TRY(m->extra_map.upload(extra_map.data(), extra_map.size() * sizeof(int)));
    if (n_lmk) {
        TS_TRY(m->lmk_map.upload(lmk_map.data(), lmk_map.size() * sizeof(int)));
        TS_TRY(m->bary.upload(lmk_bary, (size_t)n_lmk * 3 * sizeof(float)));
    }
    *out = m.release();
    return 0;
}
void ts_smplx_destroy(ts_smplx *m) { delete m; }
int ts_smplx_num_joints(const ts_smplx *m) { return m ? m->NJ : -1; }

int ts_smplx_forward(ts_smplx *m, const float *betas, int betas_per_row, const float *rows, int row_ld, int expr_off, int64_t N,
                     float *joints, float *verts, void *stream) {
    if (!m || !betas || !rows || !joints) return fail("ts_smplx_forward: null argument");
    if (N < 1 || row_ld < 1) return fail("ts_smplx_forward: bad shape");
    if (N > 0x7fffffff) return fail("ts_smplx_forward: more than 2^31 - 1 frames in one call");
    if (verts && !m->with_vertices) return fail("ts_smplx_forward: model was created without the full-mesh matrices (with_vertices = 0)");
    hipStream_t s = (hipStream_t)stream;
    ts_ctx *ctx = m->ctx;
    ts_smplx::Work &w = m->works.get(s);
    const int J = m->J, Kpad = m->Kpad;
    const size_t F = sizeof(float);
    TS_TRY(w.X.ensure((size_t)N * Kpad * F));
    TS_TRY(w.rot.ensure((size_t)N * J * 9 * F));
    TS_TRY(w.jrest.ensure((size_t)N * J * 3 * F));
    TS_TRY(w.G.ensure((size_t)N * J * 12 * F));
    TS_TRY(w.A.ensure((size_t)N * J * 12 * F));
    {
        MiscScope ms(ctx, s);
        TS_HIP(launch_smplx_pose_prepare(rows, row_ld, betas, betas_per_row, m->NB, m->NE, expr_off, m->src_off.i(),
                                         m->pose_mean.f(), J, w.rot.f(), w.X.f(), Kpad, N, s));
    }
    ConvParams p;
    conv_layer_params(m->jdirs, w.X.f(), Kpad, 1, (int)N, nullptr, 0, w.jrest.f(), J * 3, 0, J * 3, &p);
    TS_TRY(run_conv(ctx, p, 0, s));
    {
        MiscScope ms(ctx, s);
        TS_HIP(launch_smplx_rigid_chain(w.rot.f(), w.jrest.f(), J * 3, m->parents.i(), J, N, w.G.f(), w.A.f(), joints, m->NJ, s));
    }
    if (m->n_extra + m->n_lmk > 0) {
        const int U = m->U;
        TS_TRY(w.vposed.ensure((size_t)N * U * 3 * F));
        TS_TRY(w.vs.ensure((size_t)N * U * 3 * F));
        conv_layer_params(m->blend_sub, w.X.f(), Kpad, 1, (int)N, nullptr, 0, w.vposed.f(), U * 3, 0, U * 3, &p);
        TS_TRY(run_conv(ctx, p, 0, s));
        MiscScope ms(ctx, s);
        TS_HIP(launch_smplx_skin(w.vposed.f(), U * 3, w.A.f(), J, m->bone_sub.i(), m->wgt_sub.f(), m->KW, U, N, w.vs.f(),
                                 (long)U * 3, s));
        TS_HIP(launch_smplx_joints_tail(w.vs.f(), (long)U * 3, m->extra_map.i(), m->n_extra, m->lmk_map.i(), m->bary.f(),
                                        m->n_lmk, J, joints, m->NJ, N, s));
    }
    if (verts) {   // the full mesh, in chunks of frames (the posed-vertex scratch of one chunk stays around 256 MB)
        const int V = m->V;
        const long chunk = std::max<long>(1, std::min<long>(N, (long)(256u << 20) / ((long)V * 3 * F)));
        TS_TRY(w.vposed.ensure((size_t)chunk * V * 3 * F));
        for (long n0 = 0; n0 < N; n0 += chunk) {
            const long nn = std::min(chunk, N - n0);
            conv_layer_params(m->blend_full, w.X.f() + (size_t)n0 * Kpad, Kpad, 1, (int)nn, nullptr, 0, w.vposed.f(), V * 3, 0, V * 3, &p);
            TS_TRY(run_conv(ctx, p, 0, s));
            MiscScope ms(ctx, s);
            TS_HIP(launch_smplx_skin(w.vposed.f(), V * 3, w.A.f() + (size_t)n0 * J * 12, J, m->bone_full.i(), m->wgt_full.f(),
                                     m->KW, V, nn, verts + (size_t)n0 * V * 3, (long)V * 3, s));
        }
    }
    return 0;
}

}  // extern "C"
