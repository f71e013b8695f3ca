// Evaluation reductions on the device: the heavy parts of the reference's CPU metrics code.
//
//   feat_stats      count / sum x / sum x x^T of feature rows in float64     evaluation/FGD.py:131-146  (np.mean, np.cov -> Frechet distance)
//   l1_rows         sum over rows of |a - b|_1                                evaluation/FGD.py:153-158  (feat_dist)
//   body_loss       LVD, L2 "error" and variance "diverse" of joint tracks    scripts/test_body.py:98-110, evaluation/metrics.py:27-36,79-84
//   diversity       mean |seq_i - seq_j| over all pairs                       evaluation/metrics.py:96-109
//
// All of them are HBM-bound streaming reductions; every one is TWO launches — per-workgroup partials in a fixed
// order, then one workgroup summing the partials in index order — so results do not depend on scheduling (no float
// atomics).  Accumulation is float64 like the numpy / torch.float64 code they replace.
#include "kernels.h"

namespace ts {

namespace {

constexpr int FS_ROWS = 128;   // feature rows staged per workgroup iteration

// partial[wg] = { sum_d x_d (D doubles), sum x_i x_j (D*D doubles) } over the rows this workgroup owns (strided chunks)
template <int D>
__global__ __launch_bounds__(256) void feat_stats_partial(const float *__restrict__ x, long n, double *__restrict__ part) {
    __shared__ float rows[FS_ROWS][D + 1];
    const int tid = threadIdx.x;
    constexpr int TI = D / 16;                   // each thread owns a TI x TI block of the outer product (256 threads = 16 x 16)
    const int bi = (tid >> 4) * TI, bj = (tid & 15) * TI;
    double acc[TI][TI];
    double s = 0.0;                              // threads 0..D-1 also carry the column sums
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TI; ++b) acc[a][b] = 0.0;
    for (long r0 = (long)blockIdx.x * FS_ROWS; r0 < n; r0 += (long)gridDim.x * FS_ROWS) {
        const int nr = (int)((n - r0) < FS_ROWS ? (n - r0) : FS_ROWS);
        for (int e = tid; e < nr * D; e += 256) rows[e / D][e % D] = x[r0 * D + e];    // coalesced: rows are contiguous
        __syncthreads();
        for (int r = 0; r < nr; ++r) {
            double xi[TI], xj[TI];
#pragma unroll
            for (int a = 0; a < TI; ++a) { xi[a] = rows[r][bi + a]; xj[a] = rows[r][bj + a]; }
#pragma unroll
            for (int a = 0; a < TI; ++a)
#pragma unroll
                for (int b = 0; b < TI; ++b) acc[a][b] += xi[a] * xj[b];
            if (tid < D) s += rows[r][tid];
        }
        __syncthreads();
    }
    double *p = part + (size_t)blockIdx.x * (D + D * D);
    if (tid < D) p[tid] = s;
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TI; ++b) p[D + (bi + a) * D + bj + b] = acc[a][b];
}

// out[k] = sum over workgroups (ascending) of part[wg][k]
__global__ void sum_partials(const double *__restrict__ part, int nwg, int width, double *__restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= width) return;
    double s = 0.0;
    for (int w = 0; w < nwg; ++w) s += part[(size_t)w * width + k];
    out[k] = s;
}

__device__ __forceinline__ double block_sum(double v, double *sh) {   // deterministic tree over 256 threads
    const int tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) sh[tid] += sh[tid + o];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

// part[wg] = sum over this workgroup's elements of |a - b|  (rows of width D flattened: the row sums add up to the same total)
__global__ __launch_bounds__(256) void l1_partial(const float *__restrict__ a, const float *__restrict__ b, long n,
                                                  double *__restrict__ part) {
    __shared__ double sh[256];
    double s = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += fabs((double)a[i] - (double)b[i]);
    const double t = block_sum(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// body_loss partials.  One workgroup per time step t (T workgroups); part[t] = {lvd_t, err_t, var_t}:
//   lvd_t (t < Tl-1): sum_b sum_{j<Jl} | |p[b,t+1,j]-p[b,t,j]|_2 - |g[t+1,j]-g[t,j]|_2 |          metrics.py:73-84 (non-symmetrical, unweighted)
//   err_t: sum_b sum_j |g[t,j] - p[b,t,j]|_2                                                      test_body.py:103
//   var_t: sum_j | var_b(p[:,t,j,:]) |_2   (unbiased variance over the B samples)                 test_body.py:106
__global__ __launch_bounds__(256) void body_loss_partial(const float *__restrict__ gt, const float *__restrict__ prs, int B,
                                                         int T, int J, int Jl, int Tl, double *__restrict__ part) {
    __shared__ double sh[256];
    const int t = blockIdx.x, tid = threadIdx.x;
    double lvd = 0.0, err = 0.0, var = 0.0;
    for (int e = tid; e < B * J; e += 256) {
        const int b = e / J, j = e - b * J;
        const float *p = prs + (((size_t)b * T + t) * J + j) * 3, *g = gt + ((size_t)t * J + j) * 3;
        const double dx = (double)g[0] - p[0], dy = (double)g[1] - p[1], dz = (double)g[2] - p[2];
        err += sqrt(dx * dx + dy * dy + dz * dz);
        if (j < Jl && t + 1 < Tl) {
            const float *p1 = p + (size_t)J * 3, *g1 = g + (size_t)J * 3;
            const double pvx = (double)p1[0] - p[0], pvy = (double)p1[1] - p[1], pvz = (double)p1[2] - p[2];
            const double gvx = (double)g1[0] - g[0], gvy = (double)g1[1] - g[1], gvz = (double)g1[2] - g[2];
            lvd += fabs(sqrt(pvx * pvx + pvy * pvy + pvz * pvz) - sqrt(gvx * gvx + gvy * gvy + gvz * gvz));
        }
    }
    for (int j = tid; j < J; j += 256) {
        double v2 = 0.0;
        for (int c = 0; c < 3; ++c) {
            double m = 0.0;
            for (int b = 0; b < B; ++b) m += prs[(((size_t)b * T + t) * J + j) * 3 + c];
            m /= B;
            double q = 0.0;
            for (int b = 0; b < B; ++b) {
                const double d = prs[(((size_t)b * T + t) * J + j) * 3 + c] - m;
                q += d * d;
            }
            q = B > 1 ? q / (B - 1) : 0.0 / 0.0;        // torch.var of one sample is nan
            v2 += q * q;
        }
        var += sqrt(v2);
    }
    const double a = block_sum(lvd, sh), b2 = block_sum(err, sh), c2 = block_sum(var, sh);
    if (tid == 0) {
        part[(size_t)t * 3 + 0] = a;
        part[(size_t)t * 3 + 1] = b2;
        part[(size_t)t * 3 + 2] = c2;
    }
}

// one workgroup per pair (i < j): part[pair] = sum_k |kps[i][k] - kps[j][k]|
__global__ __launch_bounds__(256) void diversity_partial(const float *__restrict__ kps, int bs, long L, double *__restrict__ part) {
    __shared__ double sh[256];
    // unrank the pair index: row i holds bs-1-i pairs
    int i = 0, rem = blockIdx.x;
    while (rem >= bs - 1 - i) { rem -= bs - 1 - i; ++i; }
    const int j = i + 1 + rem;
    const float *a = kps + (size_t)i * L, *b = kps + (size_t)j * L;
    double s = 0.0;
    for (long k = threadIdx.x; k < L; k += 256) s += fabs((double)a[k] - (double)b[k]);
    const double t = block_sum(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

}  // namespace

int eval_feat_stats_workgroups(long n) {
    const long w = (n + FS_ROWS - 1) / FS_ROWS;
    return (int)(w < 1 ? 1 : (w > 1024 ? 1024 : w));
}

hipError_t launch_feat_stats(const float *x, long n, int D, double *scratch, double *sum_out, double *outer_out, hipStream_t s) {
    const int nwg = eval_feat_stats_workgroups(n);
    if (D == 64) hipLaunchKernelGGL(feat_stats_partial<64>, dim3(nwg), dim3(256), 0, s, x, n, scratch);
    else if (D == 32) hipLaunchKernelGGL(feat_stats_partial<32>, dim3(nwg), dim3(256), 0, s, x, n, scratch);
    else if (D == 128) hipLaunchKernelGGL(feat_stats_partial<128>, dim3(nwg), dim3(256), 0, s, x, n, scratch);
    else return hipErrorInvalidValue;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int width = D + D * D;
    // sum_out and outer_out are one contiguous [D + D*D] block on the caller's side (checked by the caller)
    (void)outer_out;
    hipLaunchKernelGGL(sum_partials, dim3((width + 255) / 256), dim3(256), 0, s, scratch, nwg, width, sum_out);
    return hipGetLastError();
}

hipError_t launch_l1_total(const float *a, const float *b, long n, double *scratch, double *out, hipStream_t s) {
    const long w = (n + 256 * 16 - 1) / (256 * 16);
    const int nwg = (int)(w < 1 ? 1 : (w > 1024 ? 1024 : w));
    hipLaunchKernelGGL(l1_partial, dim3(nwg), dim3(256), 0, s, a, b, n, scratch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sum_partials, dim3(1), dim3(256), 0, s, scratch, nwg, 1, out);
    return hipGetLastError();
}

hipError_t launch_body_loss(const float *gt, const float *prs, int B, int T, int J, int Jl, int Tl, double *scratch, double *out3,
                            hipStream_t s) {
    hipLaunchKernelGGL(body_loss_partial, dim3(T), dim3(256), 0, s, gt, prs, B, T, J, Jl, Tl, scratch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sum_partials, dim3(1), dim3(256), 0, s, scratch, T, 3, out3);
    return hipGetLastError();
}

hipError_t launch_diversity(const float *kps, int bs, long L, double *scratch, double *out, hipStream_t s) {
    const int pairs = bs * (bs - 1) / 2;
    hipLaunchKernelGGL(diversity_partial, dim3(pairs), dim3(256), 0, s, kps, bs, L, scratch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sum_partials, dim3(1), dim3(256), 0, s, scratch, pairs, 1, out);
    return hipGetLastError();
}

}  // namespace ts
