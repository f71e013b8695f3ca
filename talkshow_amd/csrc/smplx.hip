// Batched SMPL-X forward kernels: what the reference does per frame on the CPU in float64 through the
// third-party `smplx` package (scripts/demo.py:122-152 get_vertices, data_utils/get_j.py:20-50 get_joints).
//   pose_prepare   265-d TalkSHOW rows -> per-joint rotation matrices (Rodrigues, |r + 1e-8| as smplx.lbs.batch_rodrigues)
//                  + the GEMM operand [betas | expression | (R_j - I) for j >= 1]
//   (conv_gemm)    shape + pose blend shapes as ONE GEMM over the needed vertices; rest joints as a second, tiny one
//   rigid_chain    batch_rigid_transform: one thread walks the kinematic tree of one frame
//   skin           linear blend skinning of the needed vertices with <= KW bones per vertex (sparse lbs_weights)
//   joints_tail    vertex_joint_selector (extra joints = picked vertices) and vertices2landmarks (barycentric)
// All HBM-bound except the blend-shape GEMM, which runs on conv_gemm_f32.
#include "kernels.h"

namespace ts {

// one workgroup (64 threads) per frame
__global__ __launch_bounds__(64) void smplx_pose_prepare(const float *__restrict__ rows, int row_ld, const float *__restrict__ betas,
                                                          int betas_per_row, int NB, int NE, int expr_off, const int *__restrict__ src_off,
                                                          const float *__restrict__ pose_mean, int J, float *__restrict__ rot,
                                                          float *__restrict__ X, int Kpad) {
    const long n = blockIdx.x;
    const int tid = threadIdx.x;
    const float *row = rows + n * row_ld;
    float *x = X + n * Kpad;
    const int S = NB + NE;
    const float *bt = betas + (betas_per_row ? n * NB : 0);
    for (int k = tid; k < NB; k += 64) x[k] = bt[k];
    for (int k = tid; k < NE; k += 64) x[NB + k] = row[expr_off + k];
    for (int k = S + (J - 1) * 9 + tid; k < Kpad; k += 64) x[k] = 0.f;
    for (int j = tid; j < J; j += 64) {
        const float rx = row[src_off[j]] + pose_mean[3 * j], ry = row[src_off[j] + 1] + pose_mean[3 * j + 1],
                    rz = row[src_off[j] + 2] + pose_mean[3 * j + 2];
        const float ex = rx + 1e-8f, ey = ry + 1e-8f, ez = rz + 1e-8f;
        const float angle = sqrtf(ex * ex + ey * ey + ez * ez);
        const float dx = rx / angle, dy = ry / angle, dz = rz / angle;
        const float c = cosf(angle), s = sinf(angle), t = 1.f - c;
        // R = I + s K + (1 - c) K^2,  K = skew(d)
        float R[9];
        R[0] = 1.f + t * (-(dy * dy + dz * dz));
        R[1] = -s * dz + t * (dx * dy);
        R[2] = s * dy + t * (dx * dz);
        R[3] = s * dz + t * (dx * dy);
        R[4] = 1.f + t * (-(dx * dx + dz * dz));
        R[5] = -s * dx + t * (dy * dz);
        R[6] = -s * dy + t * (dx * dz);
        R[7] = s * dx + t * (dy * dz);
        R[8] = 1.f + t * (-(dx * dx + dy * dy));
        float *r = rot + (n * J + j) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) r[k] = R[k];
        if (j >= 1) {
            float *f = x + S + (j - 1) * 9;
#pragma unroll
            for (int k = 0; k < 9; ++k) f[k] = R[k] - ((k == 0 || k == 4 || k == 8) ? 1.f : 0.f);
        }
    }
}

// one thread per frame: G_j = G_parent . [R_j | J_j - J_parent]; A_j = [G_j.R | G_j.t - G_j.R J_j]; joints = G_j.t
__global__ void smplx_rigid_chain(const float *__restrict__ rot, const float *__restrict__ jrest, int jr_ld,
                                  const int *__restrict__ parents, int J, long N, float *__restrict__ G, float *__restrict__ A,
                                  float *__restrict__ joints, int NJ) {
    const long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float *jr = jrest + n * jr_ld;
    for (int j = 0; j < J; ++j) {
        const float *R = rot + (n * J + j) * 9;
        const int p = parents[j];
        float t[3] = {jr[3 * j], jr[3 * j + 1], jr[3 * j + 2]};
        float g[12];
        if (p < 0) {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                g[4 * r] = R[3 * r];
                g[4 * r + 1] = R[3 * r + 1];
                g[4 * r + 2] = R[3 * r + 2];
                g[4 * r + 3] = t[r];
            }
        } else {
            const float *gp = G + (n * J + p) * 12;
            const float rel[3] = {t[0] - jr[3 * p], t[1] - jr[3 * p + 1], t[2] - jr[3 * p + 2]};
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float a = gp[4 * r], b = gp[4 * r + 1], c = gp[4 * r + 2];
                g[4 * r] = a * R[0] + b * R[3] + c * R[6];
                g[4 * r + 1] = a * R[1] + b * R[4] + c * R[7];
                g[4 * r + 2] = a * R[2] + b * R[5] + c * R[8];
                g[4 * r + 3] = a * rel[0] + b * rel[1] + c * rel[2] + gp[4 * r + 3];
            }
        }
        float *go = G + (n * J + j) * 12, *ao = A + (n * J + j) * 12, *jo = joints + (n * NJ + j) * 3;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            go[4 * r] = g[4 * r];
            go[4 * r + 1] = g[4 * r + 1];
            go[4 * r + 2] = g[4 * r + 2];
            go[4 * r + 3] = g[4 * r + 3];
            ao[4 * r] = g[4 * r];
            ao[4 * r + 1] = g[4 * r + 1];
            ao[4 * r + 2] = g[4 * r + 2];
            ao[4 * r + 3] = g[4 * r + 3] - (g[4 * r] * t[0] + g[4 * r + 1] * t[1] + g[4 * r + 2] * t[2]);
            jo[r] = g[4 * r + 3];
        }
    }
}

// out[n][u] = sum_k w[u][k] * A[n][bone[u][k]] . [v_posed[n][u]; 1]
__global__ void smplx_skin(const float *__restrict__ vposed, int vp_ld, const float *__restrict__ A, int J,
                           const int *__restrict__ bone, const float *__restrict__ wgt, int KW, int U, float *__restrict__ out,
                           long out_frame_stride) {
    const long n = blockIdx.y;
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U) return;
    const float *v = vposed + n * vp_ld + 3 * u;
    const float vx = v[0], vy = v[1], vz = v[2];
    float T[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = 0.f;
    for (int k = 0; k < KW; ++k) {
        const float w = wgt[(long)u * KW + k];
        if (w == 0.f) continue;
        const float *a = A + (n * J + bone[(long)u * KW + k]) * 12;
#pragma unroll
        for (int q = 0; q < 12; ++q) T[q] += w * a[q];
    }
    float *o = out + n * out_frame_stride + 3 * u;
    o[0] = T[0] * vx + T[1] * vy + T[2] * vz + T[3];
    o[1] = T[4] * vx + T[5] * vy + T[6] * vz + T[7];
    o[2] = T[8] * vx + T[9] * vy + T[10] * vz + T[11];
}

// joints[n][J + e] = vs[n][extra_map[e]];  joints[n][J + n_extra + l] = sum_f bary[l][f] * vs[n][lmk_map[l][f]]
__global__ void smplx_joints_tail(const float *__restrict__ vs, long vs_frame_stride, const int *__restrict__ extra_map, int n_extra,
                                  const int *__restrict__ lmk_map, const float *__restrict__ bary, int n_lmk, int J,
                                  float *__restrict__ joints, int NJ) {
    const long n = blockIdx.x;
    const float *v = vs + n * vs_frame_stride;
    float *jo = joints + n * NJ * 3;
    for (int e = threadIdx.x; e < n_extra + n_lmk; e += blockDim.x) {
        float x, y, z;
        if (e < n_extra) {
            const float *s = v + 3 * extra_map[e];
            x = s[0], y = s[1], z = s[2];
        } else {
            const int l = e - n_extra;
            x = y = z = 0.f;
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                const float *s = v + 3 * lmk_map[3 * l + f];
                const float b = bary[3 * l + f];
                x += b * s[0];
                y += b * s[1];
                z += b * s[2];
            }
        }
        jo[(J + e) * 3] = x;
        jo[(J + e) * 3 + 1] = y;
        jo[(J + e) * 3 + 2] = z;
    }
}

hipError_t launch_smplx_pose_prepare(const float *rows, int row_ld, const float *betas, int betas_per_row, int NB, int NE,
                                     int expr_off, const int *src_off, const float *pose_mean, int J, float *rot, float *X,
                                     int Kpad, long N, hipStream_t s) {
    hipLaunchKernelGGL(smplx_pose_prepare, dim3((unsigned)N), dim3(64), 0, s, rows, row_ld, betas, betas_per_row, NB, NE, expr_off,
                       src_off, pose_mean, J, rot, X, Kpad);
    return hipGetLastError();
}
hipError_t launch_smplx_rigid_chain(const float *rot, const float *jrest, int jr_ld, const int *parents, int J, long N, float *G,
                                    float *A, float *joints, int NJ, hipStream_t s) {
    hipLaunchKernelGGL(smplx_rigid_chain, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, s, rot, jrest, jr_ld, parents, J, N, G, A,
                       joints, NJ);
    return hipGetLastError();
}
hipError_t launch_smplx_skin(const float *vposed, int vp_ld, const float *A, int J, const int *bone, const float *wgt, int KW,
                             int U, long N, float *out, long out_frame_stride, hipStream_t s) {
    // frames ride on grid.y (limit 65 535): longer runs (256 clips x 300 frames) go out in slices
    for (long n0 = 0; n0 < N; n0 += 65535) {
        const long nn = N - n0 < 65535 ? N - n0 : 65535;
        hipLaunchKernelGGL(smplx_skin, dim3((U + 127) / 128, (unsigned)nn), dim3(128), 0, s, vposed + n0 * vp_ld, vp_ld,
                           A + n0 * J * 12, J, bone, wgt, KW, U, out + n0 * out_frame_stride, out_frame_stride);
    }
    return hipGetLastError();
}
hipError_t launch_smplx_joints_tail(const float *vs, long vs_frame_stride, const int *extra_map, int n_extra, const int *lmk_map,
                                    const float *bary, int n_lmk, int J, float *joints, int NJ, long N, hipStream_t s) {
    hipLaunchKernelGGL(smplx_joints_tail, dim3((unsigned)N), dim3(64), 0, s, vs, vs_frame_stride, extra_map, n_extra, lmk_map, bary,
                       n_lmk, J, joints, NJ);
    return hipGetLastError();
}

}  // namespace ts
